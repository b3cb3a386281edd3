mkdir -p gpurun_out
O=gpurun_out/r04_w_wino.txt
python scripts/sp_layer_times.py 64 2>&1 | grep -v amdgpu > $O
SUPERSLAM_HIP_CONV64=wino python scripts/sp_layer_times.py 64 2>&1 | grep -v amdgpu >> $O
cat $O
SUPERSLAM_HIP_CONV64=wino timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bench_batch_parity.py -x -q 2>&1 | tail -6
