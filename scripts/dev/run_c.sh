mkdir -p gpurun_out
O=gpurun_out/energy_g.jsonl; : > $O
for P in 2 4 8 16 64; do
python scripts/dev/stage_energy.py --tag P$P --pairs $P --sp 1,2,3,4,5,6,8 2>/dev/null | tail -1 >> $O
done
python - <<'PY'
import json
for l in open('gpurun_out/energy_g.jsonl'):
    j=json.loads(l); P=int(j["tag"][1:]); print(j["tag"])
    for r in j["rows"]: print(f'   {r["stage"]:22s} {r["launch_us"]:9.1f} us  {r["avg_W"]:7.1f} W  {r["sclk_MHz"]:6.0f} MHz  {r["joules_per_launch"]:.4f} J   per image: {r["launch_us"]/(2*P):8.2f} us {r["joules_per_launch"]/(2*P)*1e3:8.3f} mJ')
PY
