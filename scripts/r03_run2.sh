#!/bin/bash
# round 3, GPU call 2: XCD-aware attention mapping (A/B), softmax variant 2 default, pipelined per-frame runner, latency diagnosis, P sweep
R=$(pwd); O=$R/gpurun_out/r03_b; mkdir -p $O
for cfg in "1 0" "2 0" "1 1" "2 1"; do set -- $cfg
  SUPERSLAM_HIP_ATTN_V=$1 SUPERSLAM_HIP_ATTN_XCD=$2 timeout 300 python scripts/lg_stage_times.py 64 600 2>&1 | tail -1 | sed "s/^/V=$1 XCD=$2 /" >> $O/attn_ab.txt
  SUPERSLAM_HIP_ATTN_V=$1 SUPERSLAM_HIP_ATTN_XCD=$2 timeout 300 python scripts/lg_call_time.py 64 600 20 2>&1 | tail -1 | sed "s/^/V=$1 XCD=$2 /" >> $O/attn_ab.txt
done
cat $O/attn_ab.txt
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu.log
timeout 200 python scripts/diag_latency.py > $O/diag_latency.txt 2>&1; cat $O/diag_latency.txt
for P in 64 96 128; do CH=$((768 / P)); timeout 300 python bench.py --headline-only --pairs $P --chunks $CH --steps 10 --warmup 2 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('P=$P chunks=$CH', j['value'], 'pairs/s', j['ms_per_step'], 'ms/step')" >> $O/p_sweep.txt; done
cat $O/p_sweep.txt
ls -la $O
