mkdir -p gpurun_out
V=$(pwd)/superslam_amd/lib/variants
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bench_batch_parity.py tests/test_gpu_alt_paths.py -q -x > gpurun_out/pytest_k.log 2>&1; tail -3 gpurun_out/pytest_k.log
O=gpurun_out/energy_k.jsonl; : > $O
for rep in 1 2; do
python scripts/dev/stage_energy.py --tag rowmajor --library $V/rowmajor.so --sp 1,2,3,4,5,6,8 --calls fe 2>/dev/null | tail -1 >> $O
python scripts/dev/stage_energy.py --tag colmajor --sp 1,2,3,4,5,6,8 --calls fe 2>/dev/null | tail -1 >> $O
done
python - <<'PY'
import json
for l in open('gpurun_out/energy_k.jsonl'):
    j=json.loads(l); print(j["tag"])
    for r in j["rows"]: print(f'   {r["stage"]:22s} {r["launch_us"]:9.1f} us  {r["avg_W"]:7.1f} W  {r["sclk_MHz"]:6.0f} MHz  {r["joules_per_launch"]:.4f} J')
PY
PMC_EXTRA="--library $V/rowmajor.so" bash scripts/pmc_traffic.sh 64 gpurun_out/pmc_rowmajor.json > gpurun_out/pmc_rowmajor.log 2>&1
bash scripts/pmc_traffic.sh 64 gpurun_out/pmc_colmajor.json > gpurun_out/pmc_colmajor.log 2>&1
python - <<'PY'
import json
for t in ("rowmajor","colmajor"):
    j=json.load(open(f"gpurun_out/pmc_{t}.json"))
    for k,v in sorted(j["kernels"].items()):
        if "conv3x3" in k: print(t, k[:60], v["launches"], "fetch x2 MB", round(2*v["FETCH_SIZE_KB_mean"]/1e3,1), "write MB", round(v["WRITE_SIZE_KB_mean"]/1e3,1))
PY
