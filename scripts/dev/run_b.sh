set -x
mkdir -p gpurun_out
python scripts/dev/lg_ab.py --pairs 64 --save /tmp/ref.npz --tag base > gpurun_out/r04_b_lgab.txt 2>&1
SUPERSLAM_HIP_FFN=16 SUPERSLAM_HIP_LG_SPLIT=1 python scripts/dev/lg_ab.py --pairs 64 --ref /tmp/ref.npz --tag ffn16_nosplit >> gpurun_out/r04_b_lgab.txt 2>&1
SUPERSLAM_HIP_FFN=16 python scripts/dev/lg_ab.py --pairs 64 --ref /tmp/ref.npz --tag ffn16_split2 >> gpurun_out/r04_b_lgab.txt 2>&1
SUPERSLAM_HIP_LG_SPLIT=1 python scripts/dev/lg_ab.py --pairs 64 --ref /tmp/ref.npz --tag base_nosplit >> gpurun_out/r04_b_lgab.txt 2>&1
SUPERSLAM_HIP_FFN=16 SUPERSLAM_HIP_LG_SPLIT=1 SSHIP_FFN_TRACE=1 python scripts/dev/lg_ab.py --pairs 64 --reps 1 --tag trace 2>&1 | grep -m 12 "ffn16 trace" >> gpurun_out/r04_b_lgab.txt
cat gpurun_out/r04_b_lgab.txt
SUPERSLAM_HIP_FFN=16 timeout 600 python -m pytest tests/test_gpu_lightglue_layers.py -x -q 2>&1 | tail -8
