#!/bin/bash
R=$(pwd); O=$R/gpurun_out/r03_d; mkdir -p $O
BENCH_LATENCY_PROBE=1 timeout 900 python bench.py --no-cpu-baseline > $O/bench_probe.json 2> $O/bench_probe.err; echo "bench rc=$?"; tail -3 $O/bench_probe.err
python -c "
import json; j=json.load(open('$O/bench_probe.json')); print(j['value'], j.get('_latency_probe'), j['latency_ms_single_pair'], j['lightglue_mfma'])"
