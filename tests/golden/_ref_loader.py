"""Container-only loader for the Python-2 reference (never travels to the GPU box).

Imports /root/reference/{inferencer,variational_bayes}.py through an in-memory
lib2to3 translation.  Nothing is written next to the
reference and no reference source is copied into this repository: the
translated text lives only in this process.  Used by make_golden.py to
generate the committed .npz fixtures and by tests that are skipped when
/root/reference is absent.
"""
import os
import sys
import types
import warnings

REFERENCE_ROOT = os.environ.get("PYLDA_REFERENCE_ROOT", "/root/reference")


def reference_available():
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "variational_bayes.py"))


def load_reference():
    """Return (inferencer_module, variational_bayes_module) of the reference."""
    if "variational_bayes" in sys.modules and getattr(
            sys.modules["variational_bayes"], "_pylda_ref", False):
        return sys.modules["inferencer"], sys.modules["variational_bayes"]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        from lib2to3 import refactor
    import numpy  # noqa: F401
    import scipy
    import scipy.special
    import scipy.misc

    # shims for modules/functions the py2-era code expects
    sys.modules.setdefault("nltk", types.ModuleType("nltk"))
    if not hasattr(scipy.misc, "logsumexp"):
        scipy.misc.logsumexp = scipy.special.logsumexp

    fixers = [f for f in refactor.get_fixers_from_package("lib2to3.fixes")
              if not f.endswith("fix_import")]
    tool = refactor.RefactoringTool(fixers)
    mods = []
    for name in ("inferencer", "variational_bayes"):
        path = os.path.join(REFERENCE_ROOT, name + ".py")
        with open(path, "r") as fh:
            src = fh.read() + "\n"
        tree = tool.refactor_string(src, path)
        mod = types.ModuleType(name)
        mod.__file__ = path
        mod._pylda_ref = True
        sys.modules[name] = mod
        exec(compile(str(tree), path, "exec"), mod.__dict__)
        mods.append(mod)
    return tuple(mods)
