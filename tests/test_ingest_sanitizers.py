"""the host-side C++ of the path under ASan + UBSan.  The native parser
(pylda_amd/csrc/ingest.cpp, pure host code) is compiled with g++ -fsanitize=address,undefined
together with tests/native/ingest_fuzz.cpp and run on structured and random inputs."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_native_ingest_under_asan_ubsan(tmp_path):
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("g++ not available")
    exe = str(tmp_path / "ingest_fuzz")
    cmd = [gxx, "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
           "-fno-omit-frame-pointer", os.path.join(ROOT, "pylda_amd", "csrc", "ingest.cpp"),
           os.path.join(ROOT, "tests", "native", "ingest_fuzz.cpp"), "-o", exe]
    build = subprocess.run(cmd, capture_output=True, text=True)
    if build.returncode != 0 and "sanitize" in build.stderr and "cannot find" in build.stderr:
        pytest.skip("sanitizer runtimes not installed: " + build.stderr.splitlines()[0])
    assert build.returncode == 0, build.stderr
    run = subprocess.run([exe], capture_output=True, text=True, timeout=300,
                         env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0"))
    assert run.returncode == 0, run.stdout + run.stderr
    assert "ingest sanitizer run: ok" in run.stdout
