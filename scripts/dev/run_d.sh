mkdir -p gpurun_out
R=$PWD
cd /tmp && export TMPDIR=/tmp
for v in base ffn16; do
  if [ $v = ffn16 ]; then export SUPERSLAM_HIP_FFN=16 SUPERSLAM_HIP_LG_SPLIT=1; else export SUPERSLAM_HIP_LG_SPLIT=1; fi
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --kernel-trace -d /tmp/pmc_$v -o t -- python $R/scripts/dev/lg_ab.py --pairs 64 --reps 3 --tag $v > /tmp/pmc_$v.log 2>&1
  python - $v <<'PY'
import sqlite3, sys, glob
v = sys.argv[1]
db = sqlite3.connect(glob.glob(f"/tmp/pmc_{v}/**/t_results.db", recursive=True)[0])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
# counters + durations per kernel
cols = [r[1] for r in db.execute("pragma table_info('counters_collection')")]
ci = {k: i for i, k in enumerate(cols)}
agg = {}
for r in db.execute("select * from counters_collection"):
    name = str(r[ci.get('kernel_name', ci.get('name', 0))]).split('(')[0][:60]
    d = agg.setdefault(name, {})
    d.setdefault(r[ci['counter_name']], []).append(float(r[ci['value']]))
    if 'start' in ci and 'end' in ci and r[ci['counter_name']] == 'GRBM_GUI_ACTIVE':
        d.setdefault('dur_ns', []).append(float(r[ci['end']]) - float(r[ci['start']]))
print(v, 'columns', cols)
for k, d in sorted(agg.items()):
    if 'ffn' not in k and 'attention' not in k and 'conv' not in k: continue
    g = d.get('GRBM_GUI_ACTIVE', [])
    du = d.get('dur_ns', [])
    if g and du:
        import statistics
        print(f"{v:6s} {k:60s} n={len(g):3d} gui_active={statistics.mean(g):12.0f} dur_us={statistics.mean(du)/1e3:8.1f} clock_GHz={statistics.mean(g)/statistics.mean(du):.3f}")
    else:
        print(v, k, {kk: (len(vv), sum(vv)/len(vv)) for kk, vv in d.items()})
PY
done > $R/gpurun_out/r04_d_clock.txt 2>&1
cat $R/gpurun_out/r04_d_clock.txt | tail -30
