mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_alt_paths.py tests/test_gpu_bench_batch_parity.py -x -q 2>&1 | tail -3
python bench.py --steps 5 --warmup 2 > gpurun_out/r04_l_bench.json 2> gpurun_out/r04_l_bench.err
python - <<'PY'
import json
j = json.load(open("gpurun_out/r04_l_bench.json"))
print("value", j["value"], "ms/step", j["ms_per_step"], "self_check", j["self_check"]["ok"], j["self_check"]["mscores_maxd"], j["self_check"]["mutual_flips"])
print("single_pair_protocol", j["single_pair_protocol"])
print("latency", j.get("latency_ms_single_pair"), "deployed", j.get("deployed_unit"))
print("stage_ms", j["stage_ms"])
for e in j["roofline_hbm"]:
    print(e["kernel"][:40], e["launch_ms"])
print("lightglue_mfma", j["lightglue_mfma"])
PY
