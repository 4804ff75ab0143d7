#!/bin/bash
# The LAST act of a round, on the GPU box (gpurun -- 'bash tools/finalize_round.sh r05'): rocprofv3 kernel-trace + SQ + FETCH_SIZE /
# WRITE_SIZE passes of cfg 4, cfg 3, cfg 5 and associated-press with the kernels as they are NOW, the traffic files bench.py ties
# to the hash of the device headers, then the default bench line (which therefore carries roofline.traffic), the
# --steps 20 --warmup 5 line and the GPU suite.  Everything lands under gpurun_out/; copy into profiles/ with
#   bash tools/finalize_round.sh --collect r05        (here, after the call has merged gpurun_out/)
TAG=${1:-rXX}
if [ "$TAG" = "--collect" ]; then
    set -euo pipefail      # a missing gpurun_out file must not leave a stale profiles/traffic_*.json behind
    TAG=$2
    for n in cfg4 cfg3; do
        cp gpurun_out/prof_${TAG}_$n/summary.txt profiles/${TAG}_${n}_rocprof_summary.txt
        cp gpurun_out/prof_${TAG}_$n/kernel_stats.csv profiles/${TAG}_${n}_kernel_stats.csv
    done
    cp gpurun_out/prof_${TAG}_nips/summary.txt profiles/${TAG}_nips_k500_rocprof_summary.txt
    cp gpurun_out/prof_${TAG}_nips/kernel_stats.csv profiles/${TAG}_nips_k500_kernel_stats.csv
    cp gpurun_out/prof_${TAG}_ap/summary.txt profiles/${TAG}_ap_k10_rocprof_summary.txt
    cp gpurun_out/prof_${TAG}_ap/kernel_stats.csv profiles/${TAG}_ap_k10_kernel_stats.csv
    cp gpurun_out/prof_${TAG}_cfg4/traffic.json profiles/traffic_synth1m.json
    cp gpurun_out/prof_${TAG}_cfg3/traffic.json profiles/traffic_synth100k.json
    cp gpurun_out/bench_final.json profiles/${TAG}_bench_default.json
    cp gpurun_out/bench_steps20.json profiles/${TAG}_bench_steps20_warmup5.json
    python -c "import json, bench; t = json.load(open('profiles/traffic_synth1m.json')); print('traffic hash', t['kernel_source_hash'], 'tree', bench.kernel_source_hash())"
    exit 0
fi
bash tools/profile_bench.sh ${TAG}_cfg4 --workload synth1m > gpurun_out/prof_${TAG}_cfg4.log 2>&1
bash tools/profile_bench.sh ${TAG}_cfg3 --workload synth100k > gpurun_out/prof_${TAG}_cfg3.log 2>&1
bash tools/profile_cmd.sh ${TAG}_nips python bench.py --workload nips --no-cpu-baseline --no-extras --steps 5 --warmup 3 > gpurun_out/prof_${TAG}_nips.log 2>&1
bash tools/profile_cmd.sh ${TAG}_ap python bench.py --workload ap --no-cpu-baseline --no-extras --steps 20 --warmup 5 > gpurun_out/prof_${TAG}_ap.log 2>&1
cp gpurun_out/prof_${TAG}_cfg4/traffic.json profiles/traffic_synth1m.json
cp gpurun_out/prof_${TAG}_cfg3/traffic.json profiles/traffic_synth100k.json
( time python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err ) 2>&1 | tail -3
python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > gpurun_out/bench_steps20.json 2>/dev/null
python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|rror" | head -5
