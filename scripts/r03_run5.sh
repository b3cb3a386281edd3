#!/bin/bash
# round 3, GPU call 5: LDS-ring attention kernel A/B (+ 3 waves per SIMD build), parity
R=$(pwd); O=$R/gpurun_out/r03_e; mkdir -p $O
for cfg in "0 2" "1 2" "1 3" "2 2" "2 3"; do set -- $cfg
  SUPERSLAM_HIP_ATTN_RING=$1 SUPERSLAM_HIP_ATTN_WPS=$2 timeout 300 python scripts/lg_stage_times.py 64 600 2>&1 | tail -1 | sed "s/^/RING=$1 WPS=$2 /" >> $O/attn_ring.txt
  SUPERSLAM_HIP_ATTN_RING=$1 SUPERSLAM_HIP_ATTN_WPS=$2 timeout 300 python scripts/lg_call_time.py 64 600 20 2>&1 | tail -1 | sed "s/^/RING=$1 WPS=$2 /" >> $O/attn_ring.txt
done
cat $O/attn_ring.txt
for cfg in "1 2" "1 3" "2 3"; do set -- $cfg
  SUPERSLAM_HIP_ATTN_RING=$1 SUPERSLAM_HIP_ATTN_WPS=$2 timeout 600 python -m pytest tests/test_gpu_lightglue_layers.py tests/test_gpu_parity.py tests/test_gpu_bench_batch_parity.py tests/test_gpu_alt_paths.py -m gpu -x -q > $O/pytest_ring$1_wps$2.log 2>&1; echo "RING=$1 WPS=$2 pytest rc=$?" | tee -a $O/attn_ring.txt; tail -2 $O/pytest_ring$1_wps$2.log
done
