"""Registers and scratch of the document kernels, from the compiler's own assembly (no GPU): the quad kernels
(estep_quad.h) sit AT the 256-register limit, where one more live value makes the allocator spill tile rows inside
the inner loop (round 6: a vector register holding the uniform live-topic count cost the 209-224-term class two
rows of its tile per iteration).  The hot path - basic blocks inside a loop that carry >= 16 fp64 multiply-adds -
is held to the scratch traffic recorded here; the live-topic kernels (estep_compact.h) to none at all in theirs.  A change that
raises a figure has to lower it again or say why."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

# scratch instructions allowed in the hot blocks of an instantiation the planner can select (host_plan.cpp quad_geom_for);
# everything not listed: 0.  Round 5 shipped 1 / 1 / 1 / 1 / 3 in <16,10,4,0> <32,10,3,0> <32,10,4,0> <16,9,4,2> <16,9,4,3>.
QUAD_HOT_SCRATCH_CEILING = {
    "<16, 10, 4, 0, true>": 1, "<32, 10, 4, 0, true>": 2, "<16, 9, 4, 2, true>": 6, "<16, 9, 4, 3, true>": 7, "<32, 8, 4, 4, true>": 1,
    # ... and without the hand-over (launch_quad_dense.hip: the corpora that hand nothing over run round 5's loop)
    "<16, 10, 4, 0, false>": 1, "<32, 10, 4, 0, false>": 1, "<16, 9, 4, 2, false>": 1, "<16, 9, 4, 3, false>": 3,
}


def _analyse(source):
    import kernel_resources as kr
    path = kr.compile_to_asm(os.path.join(ROOT, "pylda_amd", "csrc", source))
    lines = open(path).read().splitlines()
    res = kr.resources(lines, "estep")
    names = kr.demangle(list(res))
    hot = {names[k]: sum(n for _, _, n in blocks) for k, blocks in kr.hot_scratch(lines, "estep").items()}
    return {names[k]: v for k, v in res.items()}, hot


@pytest.mark.parametrize("source", ["launch_quad.hip", "launch_quad_dense.hip"])
def test_quad_kernels_keep_their_tile_in_registers(source):
    res, hot = _analyse(source)
    assert len(res) == 16, sorted(res)
    for name, info in res.items():
        assert info["NumVgprs"] <= 256 and info["Occupancy"] >= 2, (name, info)      # two wavefronts per SIMD
        shape = name[name.index("<"):name.index(">") + 1]
        ceiling = QUAD_HOT_SCRATCH_CEILING.get(shape, 0)
        assert hot.get(name, 0) <= ceiling, "%s: %d scratch instructions in the hot blocks (ceiling %d)" % (name, hot.get(name, 0), ceiling)


def test_live_topic_kernels_have_no_scratch_in_their_loops():
    res, hot = _analyse("launch_compact.hip")
    assert len(res) == 16, sorted(res)                   # lane shapes 1 .. 8 term slots per lane, one or two wavefronts per document
    for name, info in res.items():
        assert info["NumVgprs"] <= 256 and info["Occupancy"] >= 2, (name, info)      # two wavefronts per SIMD
        assert info["ScratchSize"] <= 128, (name, info)                                # (a body's tile load, prologue / epilogue only)
    assert not hot, hot
