#!/bin/bash
R=$(pwd); O=$R/gpurun_out/pairs; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_alt_paths.py tests/test_gpu_bench_batch_parity.py tests/test_gpu_parity.py tests/test_gpu_stress_shapes.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
for pr in 1 0; do
  SUPERSLAM_HIP_CONV128_PAIRS=$pr python bench.py --headline-only --steps 10 --warmup 2 2>/dev/null | python -c "
import sys, json
b = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('pairs=$pr', 'headline', b['value'], 'ms/step', b['ms_per_step'])"
done
for pr in 1 0; do
  SUPERSLAM_HIP_CONV128_PAIRS=$pr python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
b = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('pairs=$pr', b['value'], {k: b['insitu_launch_ms'][k] for k in ('conv3b+pool', 'conv4a', 'conv4b', 'convPa')}, {k: b['layer_ms'][k] for k in ('conv4a', 'conv4b', 'convPa')}, b['self_check']['ok'], b['self_check']['kp_bit_identical'])"
done
