mkdir -p gpurun_out
O=gpurun_out/r04_c_lgab.txt
python scripts/dev/lg_ab.py --pairs 64 --save /tmp/ref.npz --tag base > $O 2>&1
SUPERSLAM_HIP_FFN=16 SUPERSLAM_HIP_LG_SPLIT=1 python scripts/dev/lg_ab.py --pairs 64 --ref /tmp/ref.npz --tag ffn16v2_nosplit >> $O 2>&1
SUPERSLAM_HIP_FFN=16 python scripts/dev/lg_ab.py --pairs 64 --ref /tmp/ref.npz --tag ffn16v2_split2 >> $O 2>&1
SUPERSLAM_HIP_LIBRARY=$PWD/superslam_amd/lib/variants/nopk.so SUPERSLAM_HIP_FFN=16 SUPERSLAM_HIP_LG_SPLIT=1 python scripts/dev/lg_ab.py --pairs 64 --ref /tmp/ref.npz --tag ffn16v2_nopk_nosplit >> $O 2>&1
SUPERSLAM_HIP_FFN=16 SUPERSLAM_HIP_LG_SPLIT=1 SSHIP_FFN_TRACE=1 python scripts/dev/lg_ab.py --pairs 64 --reps 1 --tag trace 2>&1 | grep -m 4 "ffn16 trace" >> $O
grep -v amdgpu.ids $O
SUPERSLAM_HIP_FFN=16 timeout 600 python -m pytest tests/test_gpu_lightglue_layers.py -x -q 2>&1 | tail -3
