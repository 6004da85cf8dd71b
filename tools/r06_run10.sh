#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out
timeout 1500 python bench.py > $O/r06_bench_full_a.json 2> $O/r06_bench_full_a.err
python - <<'PY'
import json
o=json.loads([l for l in open('gpurun_out/r06_bench_full_a.json') if l.startswith('{')][-1])
print(round(o['value'],1), round(o['ms_per_step'],3), o['roofline']['kernel'], round(o['roofline']['frac'],3), o['cpu_baseline']['value'])
for k,v in o['config'].items():
    if k.endswith('_summary'): print(k, v)
PY
timeout 600 python -m pytest tests/test_gpu_resume.py -q > $O/r06_t1.log 2>&1; tail -n 4 $O/r06_t1.log
