#!/bin/bash
# Profile bench.py on the GPU box: kernel-trace stats + PMC passes (separately,
# as gpurun requires).  Usage: tools/profile_bench.sh <tag> [bench args...]
# Raw rocprofv3 output stays in /tmp on the box; the condensed summary and the
# stats CSVs are copied to gpurun_out/prof_<tag>/ (copy what should be judged
# into profiles/).
set -u
TAG=${1:-run}; shift || true
RAW=/tmp/prof_$TAG
OUT=gpurun_out/prof_$TAG
rm -rf $RAW; mkdir -p $RAW $OUT
export TMPDIR=/tmp
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-extras $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/trace -o trace -- python bench.py $ARGS > $OUT/bench_trace.json 2> $RAW/trace.err
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $RAW/pmc1 -o pmc -- python bench.py $ARGS > /dev/null 2> $RAW/pmc1.err
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE --output-format csv -d $RAW/pmc2 -o pmc -- python bench.py $ARGS > /dev/null 2> $RAW/pmc2.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $RAW/pmc3 -o pmc -- python bench.py $ARGS > /dev/null 2> $RAW/pmc3.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $RAW/pmc4 -o pmc -- python bench.py $ARGS > /dev/null 2> $RAW/pmc4.err
find $RAW -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
python tools/summarize_prof.py $RAW > $OUT/summary.txt 2>&1
cp $RAW/traffic.json $OUT/traffic.json 2>/dev/null
tail -3 $RAW/*.err | grep -iE "error|fail" | head
cat $OUT/summary.txt
