mkdir -p gpurun_out
O=gpurun_out/r04_i_latency_inflight16.txt
python scripts/dev/lg_ab.py --pairs 1 --reps 200 --tag p1 > $O 2>&1
python scripts/dev/lg_ab.py --pairs 2 --reps 100 --tag p2 >> $O 2>&1
python scripts/dev/lg_ab.py --pairs 8 --reps 50 --tag p8 >> $O 2>&1
python scripts/dev/lg_ab.py --pairs 16 --reps 50 --tag p16 >> $O 2>&1
grep -v amdgpu.ids $O
timeout 900 python -m pytest tests/test_gpu_lightglue_layers.py tests/test_gpu_parity.py -x -q 2>&1 | tail -3
