#!/bin/bash
R=$(pwd); O=$R/gpurun_out/u; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_lightglue_layers.py tests/test_gpu_parity.py tests/test_gpu_bench_batch_parity.py tests/test_gpu_stress_shapes.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
python scripts/dev/lg_ab.py --pairs 1 --tag p1 2>&1 | tail -1 | tee $O/lg_p1.txt
python scripts/dev/lg_ab.py --pairs 64 --tag p64 2>&1 | tail -1 | tee -a $O/lg_p1.txt
SUPERSLAM_HIP_LG_ASSIGN=matrix python scripts/dev/lg_ab.py --pairs 64 --tag p64_matrix 2>&1 | tail -1 | tee -a $O/lg_p1.txt
