"""CPU-side checks of the drop-in boundary: the C-ABI library loads and
exports every symbol include/pylda_hip.h declares (no compute calls)."""
import os
import re

import pytest

from conftest import ROOT


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "pylda_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pylda_[a-z_]+)\s*\(", text)))


def test_header_and_binding_list_the_same_symbols():
    from pylda_amd import _capi
    assert declared_symbols() == sorted(_capi.SIGNATURES)


def test_library_exports_every_declared_symbol():
    from pylda_amd import _capi, build
    build.build(verbose=False)
    lib = _capi.load()
    for name in declared_symbols():
        assert hasattr(lib, name), name
    assert b"gfx950" in lib.pylda_version()


def test_create_fails_loudly_without_a_gpu():
    from pylda_amd import _capi
    if _capi.device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(_capi.PyldaError) as e:
        _capi.Context(4, 10)
    assert e.value.status == -2 and "no CPU fallback" in str(e.value)
