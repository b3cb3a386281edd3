#!/bin/bash
# round 3, GPU call 3: bench latency probe, attention QT=1 (3 waves / SIMD) A/B, pipelined per-frame runner, SQ counters of the attention kernel
R=$(pwd); O=$R/gpurun_out/r03_c; mkdir -p $O
python -c "
import sys; sys.path.insert(0,'tests'); sys.path.insert(0,'.')
from superslam_amd.weights import *; import os
os.makedirs('/tmp/w',exist_ok=True)
save_safetensors(make_superpoint_weights(0),'/tmp/w/sp.safetensors'); save_safetensors(make_lightglue_weights(1),'/tmp/w/lg.safetensors')"
for qt in 0 1; do for ks in 0 1 2; do
  SUPERSLAM_HIP_ATTN_QT=$qt SUPERSLAM_HIP_ATTN_KS=$ks timeout 300 python scripts/lg_stage_times.py 64 600 2>&1 | tail -1 | sed "s/^/QT=$qt KS=$ks /" >> $O/attn_qt.txt
  SUPERSLAM_HIP_ATTN_QT=$qt SUPERSLAM_HIP_ATTN_KS=$ks timeout 300 python scripts/lg_call_time.py 64 600 20 2>&1 | tail -1 | sed "s/^/QT=$qt KS=$ks /" >> $O/attn_qt.txt
done; done
cat $O/attn_qt.txt
B=superslam_amd/lib/frontend_benchmark
for args in "" "--no-pipeline" "--keyframe-match" "--keyframe-match --no-pipeline" "--no-ring"; do
  echo "== frontend_benchmark --synthetic 300 $args" >> $O/frontend_benchmark.txt
  timeout 120 $B --sp /tmp/w/sp.safetensors --lg /tmp/w/lg.safetensors --synthetic 300 $args 2>&1 | grep -E "pipelined|per-frame|throughput|stereo matches|keyframe" >> $O/frontend_benchmark.txt
done
cat $O/frontend_benchmark.txt
BENCH_LATENCY_PROBE=1 timeout 900 python bench.py --no-cpu-baseline > $O/bench_probe.json 2> $O/bench_probe.err; echo "bench rc=$?"
python -c "
import json; j=json.load(open('$O/bench_probe.json')); print(j['value'], j.get('_latency_probe'), j['latency_ms_single_pair'])"
timeout 400 scripts/pmc_sq.sh 64 gpurun_out/r03_c/pmc_sq_raw.txt; python scripts/pmc_sq_table.py $O/pmc_sq_raw.txt > $O/pmc_sq_P64.txt; rm -f $O/pmc_sq_raw.txt; head -12 $O/pmc_sq_P64.txt
ls -la $O
