#!/bin/bash
# Profile any command on the GPU box: kernel-trace stats + SQ counter passes (separately, as gpurun
# requires).  Usage: tools/profile_cmd.sh <tag> <command...>
# The condensed summary and the stats CSV are copied to gpurun_out/prof_<tag>/.
set -u
TAG=${1:-run}; shift
RAW=/tmp/prof_$TAG
OUT=gpurun_out/prof_$TAG
rm -rf $RAW; mkdir -p $RAW $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/trace -o trace -- "$@" > $OUT/stdout_trace.txt 2> $RAW/trace.err
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $RAW/pmc1 -o pmc -- "$@" > /dev/null 2> $RAW/pmc1.err
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE --output-format csv -d $RAW/pmc2 -o pmc -- "$@" > /dev/null 2> $RAW/pmc2.err
find $RAW -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
python tools/summarize_prof.py $RAW > $OUT/summary.txt 2>&1
tail -3 $RAW/*.err | grep -iE "error|fail" | head
cat $OUT/summary.txt
