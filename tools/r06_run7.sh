#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out
one() { tag=$1; shift; env "$@" timeout 600 python bench.py --no-cpu-baseline --no-also --no-roofline --steps 20 --repeats 3 2>/dev/null | grep '^{' | python -c "
import json,sys
for l in sys.stdin:
    o=json.loads(l); print('$tag', round(o['value'],1), round(o['ms_per_step'],3), o['config'].get('deferred_share'), o['config'].get('deferred_images'), o.get('step_schedule',{}).get('autotune_ms_per_step_by_share'))
"; }
: > $O/r06_defer_attn_wgs.txt
for w in 256 192 176 160 128; do one wgs$w SR_DEFER_ATTN_WGS=$w >> $O/r06_defer_attn_wgs.txt; done
one wgs256_again SR_DEFER_ATTN_WGS=256 >> $O/r06_defer_attn_wgs.txt
cat $O/r06_defer_attn_wgs.txt
timeout 600 python -m pytest tests/test_gpu_resume.py tests/test_gpu_bench_launch.py -q -k "resume or forced_rank" > $O/r06_t1.log 2>&1; tail -n 5 $O/r06_t1.log
