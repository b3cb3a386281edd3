#!/bin/bash
R=$(pwd); O=$R/gpurun_out/r; mkdir -p $O
python scripts/dev/lg_ab.py --pairs 1 --tag p1_deep 2>&1 | tail -1 | tee $O/lg_p1.txt
python scripts/dev/lg_ab.py --pairs 2 --tag p2_deep 2>&1 | tail -1 | tee -a $O/lg_p1.txt
python scripts/dev/lg_ab.py --pairs 4 --tag p4_deep 2>&1 | tail -1 | tee -a $O/lg_p1.txt
SSHIP_FFN_TRACE=1 SSHIP_FFN_TRACE_IT=0 python scripts/dev/lg_ab.py --pairs 1 --reps 2 2>&1 | grep "ffn trace" | sed -n "5,8p" | tee $O/trace.txt
timeout 600 python -m pytest tests/test_gpu_lightglue_layers.py -x -q 2>&1 | tail -3
