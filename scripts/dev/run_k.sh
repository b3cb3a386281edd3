mkdir -p gpurun_out
O=gpurun_out/r04_k_latency_epilogue.txt
python scripts/dev/lg_ab.py --pairs 1 --reps 200 --tag p1 > $O 2>&1
python scripts/dev/lg_ab.py --pairs 8 --reps 50 --tag p8 >> $O 2>&1
SSHIP_FFN_TRACE=1 SSHIP_FFN_TRACE_IT=0 python scripts/dev/lg_ab.py --pairs 1 --reps 1 --tag trace 2>&1 | grep -m 2 "ffn trace" >> $O
grep -v amdgpu.ids $O
timeout 900 python -m pytest tests/test_gpu_lightglue_layers.py tests/test_gpu_parity.py -x -q 2>&1 | tail -3
