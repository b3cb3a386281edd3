mkdir -p gpurun_out
V=superslam_amd/lib/variants
PMC_EXTRA="--library $V/nms_noxcd.so" bash scripts/pmc_traffic.sh 64 gpurun_out/pmc_noxcd.json > gpurun_out/pmc_noxcd.log 2>&1
bash scripts/pmc_traffic.sh 64 gpurun_out/pmc_xcd.json > gpurun_out/pmc_xcd.log 2>&1
python - <<'PY'
import json
for t in ("noxcd","xcd"):
    j=json.load(open(f"gpurun_out/pmc_{t}.json"))
    for k,v in j["kernels"].items():
        if "nms" in k or "convpb" in k or "topk" in k or "desc_head" in k: print(t, k[:60], v["launches"], "fetch x2 MB", round(2*v["FETCH_SIZE_KB_mean"]/1e3,1), "write MB", round(v["WRITE_SIZE_KB_mean"]/1e3,1), "total MB", round(v["hbm_bytes_per_launch"]/1e6,1))
PY
