"""the host-side planner under ASan + UBSan.  pylda_amd/csrc/host_plan.cpp holds the index arithmetic
the kernels rely on - launch classes per document length, the segment cut of the postings, rounds under a byte budget,
the XCD execution order, the sweep's term dealing - as pure functions without HIP, so it compiles with plain g++
-fsanitize=address,undefined together with tests/native/planner_fuzz.cpp and runs on random corpora on the CPU."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_planner_under_asan_ubsan(tmp_path):
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("g++ not available")
    exe = str(tmp_path / "planner_fuzz")
    cmd = [gxx, "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-pthread",
           "-fno-omit-frame-pointer", os.path.join(ROOT, "pylda_amd", "csrc", "host_plan.cpp"),
           os.path.join(ROOT, "tests", "native", "planner_fuzz.cpp"), "-o", exe]
    build = subprocess.run(cmd, capture_output=True, text=True)
    if build.returncode != 0 and "sanitize" in build.stderr and "cannot find" in build.stderr:
        pytest.skip("sanitizer runtimes not installed: " + build.stderr.splitlines()[0])
    assert build.returncode == 0, build.stderr
    run = subprocess.run([exe, "250"], capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0"))
    assert run.returncode == 0, run.stdout + run.stderr
    assert "planner sanitizer run: ok" in run.stdout


def test_no_long_functions_left_in_the_planner_sources():
    """build_postings was one 320-line function.  The planner and its two callers now hold no
    function above ~80 lines (the sources open and close a function body with a brace in column 0)."""
    for name in ("host_plan.cpp", "plan.hip", "sstats_gather.hip"):
        lines = open(os.path.join(ROOT, "pylda_amd", "csrc", name)).read().splitlines()
        opened = None
        for i, line in enumerate(lines):
            if line == "{":
                opened = i
            elif line.startswith("}") and opened is not None:
                assert i - opened <= 90, "%s: %d lines in `%s`" % (name, i - opened, lines[opened - 1].strip())
                opened = None
