mkdir -p gpurun_out
O=gpurun_out/r04_f_zero.txt
SUPERSLAM_HIP_LG_SPLIT=1 python scripts/dev/lg_ab.py --pairs 64 --tag base_nosplit > $O 2>&1
SUPERSLAM_HIP_LG_SPLIT=1 python scripts/dev/lg_ab.py --pairs 64 --zero-weights --tag base_nosplit_zero >> $O 2>&1
SUPERSLAM_HIP_FFN=16 SUPERSLAM_HIP_LG_SPLIT=1 python scripts/dev/lg_ab.py --pairs 64 --tag ffn16_nosplit >> $O 2>&1
SUPERSLAM_HIP_FFN=16 SUPERSLAM_HIP_LG_SPLIT=1 python scripts/dev/lg_ab.py --pairs 64 --zero-weights --tag ffn16_nosplit_zero >> $O 2>&1
SUPERSLAM_HIP_FFN=16 SUPERSLAM_HIP_LG_SPLIT=1 SSHIP_FFN_TRACE=1 python scripts/dev/lg_ab.py --pairs 64 --reps 1 --zero-weights --tag trace 2>&1 | grep -m 2 "ffn16 trace" >> $O
grep -v amdgpu.ids $O
