#!/bin/bash
# round 6, second GPU call: data-parallel tests on RCCL / gloo, resume, step graphs, K = 0 regime with the stage-1 update under the backward
cd "$(dirname "$0")/.."
O=gpurun_out
timeout 900 python tools/rccl_one_rank_check.py > $O/r06_rccl_one_rank.txt 2>&1
timeout 2400 python -m pytest tests/test_gpu_rccl_one_rank.py tests/test_gpu_dp_overlap.py tests/test_gpu_resume.py tests/test_gpu_stepgraph.py -q > $O/r06_t1.log 2>&1; echo "rc=$?" >> $O/r06_t1.log
timeout 1500 python -m pytest tests/test_gpu_srflexmatch.py tests/test_gpu_wrn.py -q -k "not mask_identity" > $O/r06_t2.log 2>&1; echo "rc=$?" >> $O/r06_t2.log
timeout 400 python bench.py --regime pre --no-cpu-baseline --no-also --steps 20 > $O/r06_k0.txt 2>&1
timeout 400 python bench.py --no-cpu-baseline --no-also --steps 20 > $O/r06_head.txt 2>&1
timeout 600 python bench.py --force-dp --no-cpu-baseline --no-also --no-roofline --steps 20 > $O/r06_force_dp.txt 2>&1
SR_HIP_GRAPH=1 timeout 600 python bench.py --force-dp --no-cpu-baseline --no-also --no-roofline --no-allreduce-ab --steps 20 > $O/r06_force_dp_graph.txt 2>&1
tail -n 6 $O/r06_t1.log; tail -n 6 $O/r06_t2.log
for f in k0 head force_dp force_dp_graph; do grep '^{' $O/r06_$f.txt | python -c "
import json,sys
for l in sys.stdin:
    o=json.loads(l); print('$f', round(o['value'],1), round(o['ms_per_step'],3), o['config'].get('hip_graph'), o.get('rccl_ranks'), o.get('allreduce_ms_per_step'), o.get('grad_exchange'), o['host_enqueue_ms_per_step']['max'])
"; done
