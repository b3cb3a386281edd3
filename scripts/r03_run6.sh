#!/bin/bash
R=$(pwd); O=$R/gpurun_out/r03_f; mkdir -p $O
for rep in 1 2; do for v in 2 3; do
  SUPERSLAM_HIP_ATTN_V=$v timeout 300 python scripts/lg_stage_times.py 64 600 2>&1 | tail -1 | sed "s/^/V=$v /" >> $O/attn_v3.txt
  SUPERSLAM_HIP_ATTN_V=$v timeout 300 python scripts/lg_call_time.py 64 600 20 2>&1 | tail -1 | sed "s/^/V=$v /" >> $O/attn_v3.txt
done; done
SUPERSLAM_HIP_ATTN_V=3 timeout 300 python scripts/lg_stage_times.py 1 600 2>&1 | tail -1 | sed "s/^/V=3 P=1 /" >> $O/attn_v3.txt
cat $O/attn_v3.txt
SUPERSLAM_HIP_ATTN_V=3 timeout 600 python -m pytest tests/test_gpu_lightglue_layers.py tests/test_gpu_parity.py tests/test_gpu_bench_batch_parity.py tests/test_gpu_alt_paths.py -m gpu -x -q > $O/pytest_v3.log 2>&1; echo "V=3 pytest rc=$?"; tail -2 $O/pytest_v3.log
