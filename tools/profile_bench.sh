#!/bin/bash
# Profile bench.py on the GPU box: kernel-trace stats + PMC passes (separately,
# as gpurun requires).  Usage: tools/profile_bench.sh <tag> [bench args...]
# Outputs land in gpurun_out/prof_<tag>/ ; copy the summaries into profiles/.
set -u
TAG=${1:-run}; shift || true
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-ap-extra $*"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python bench.py $ARGS > $OUT/bench_trace.json 2> $OUT/trace.err
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $OUT/pmc1 -o pmc -- python bench.py $ARGS > /dev/null 2> $OUT/pmc1.err
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE -d $OUT/pmc2 -o pmc -- python bench.py $ARGS > /dev/null 2> $OUT/pmc2.err
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc3 -o pmc -- python bench.py $ARGS > /dev/null 2> $OUT/pmc3.err
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc4 -o pmc -- python bench.py $ARGS > /dev/null 2> $OUT/pmc4.err
python tools/summarize_prof.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
