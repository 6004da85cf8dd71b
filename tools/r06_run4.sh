#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out
run() { tag=$1; shift; env "$@" timeout 600 python bench.py --force-dp --no-cpu-baseline --no-also --no-roofline --no-allreduce-ab --steps 20 --repeats 3 2>/dev/null | grep '^{' | python -c "
import json,sys
for l in sys.stdin:
    o=json.loads(l); print('$tag', round(o['ms_per_step'],3), o['config'].get('deferred_share'), o.get('step_schedule',{}).get('autotune_ms_per_step_by_share'), o.get('grad_exchange',{}).get('chosen'))
"; }
run default X=1 > $O/r06_hwq.txt
run hwq8 GPU_MAX_HW_QUEUES=8 >> $O/r06_hwq.txt
run hwq2 GPU_MAX_HW_QUEUES=2 >> $O/r06_hwq.txt
run pinned_allreduce SR_GRAD_EXCHANGE=allreduce >> $O/r06_hwq.txt
run pinned_overlap SR_GRAD_EXCHANGE=overlap >> $O/r06_hwq.txt
run pinned_overlap_hwq8 SR_GRAD_EXCHANGE=overlap GPU_MAX_HW_QUEUES=8 >> $O/r06_hwq.txt
cat $O/r06_hwq.txt
