#!/bin/bash
# round 6, first GPU call: the new data-parallel tests on RCCL, the asynchronous deferred rows (test + A/B)
cd "$(dirname "$0")/.."
O=gpurun_out
timeout 1500 python -m pytest tests/test_gpu_rccl_one_rank.py tests/test_gpu_dp_overlap.py -x -q > $O/r06_t1.log 2>&1; echo "rc=$?" >> $O/r06_t1.log
timeout 600 python tools/rccl_one_rank_check.py > $O/r06_rccl_one_rank.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_srflexmatch.py -x -q -k "asynchronous or every_pass or autotune or full_size" > $O/r06_t2.log 2>&1; echo "rc=$?" >> $O/r06_t2.log
timeout 900 tools/ab_env.sh SR_ASYNC_DEFER 0 1 --steps 20 --warmup 3 --repeats 3 > $O/r06_async_ab.txt 2>&1
SR_MAIN_PRIO=1 timeout 600 tools/ab_env.sh SR_ASYNC_DEFER 0 1 --steps 20 --warmup 3 --repeats 3 > $O/r06_async_ab_prio.txt 2>&1
SR_PHASES=1 SR_ASYNC_DEFER=1 timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-also --steps 20 --repeats 2 > $O/r06_async_phases.txt 2>&1
tail -5 $O/r06_t1.log $O/r06_t2.log; cat $O/r06_async_ab.txt $O/r06_async_ab_prio.txt; grep -i phases $O/r06_async_phases.txt
