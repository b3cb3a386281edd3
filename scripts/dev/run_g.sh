mkdir -p gpurun_out
O=gpurun_out/r04_g_latency_prefetch.txt
SUPERSLAM_HIP_LG_PREFETCH=0 python scripts/dev/lg_ab.py --pairs 1 --reps 200 --save /tmp/ref1.npz --tag p1_noprefetch > $O 2>&1
python scripts/dev/lg_ab.py --pairs 1 --reps 200 --ref /tmp/ref1.npz --tag p1_prefetch >> $O 2>&1
SUPERSLAM_HIP_LG_PREFETCH=0 python scripts/dev/lg_ab.py --pairs 2 --reps 100 --tag p2_noprefetch >> $O 2>&1
python scripts/dev/lg_ab.py --pairs 2 --reps 100 --tag p2_prefetch >> $O 2>&1
grep -v amdgpu.ids $O
timeout 900 python -m pytest tests/test_gpu_lightglue_layers.py tests/test_gpu_parity.py -x -q 2>&1 | tail -3
