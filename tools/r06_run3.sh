#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out
SR_PHASES=1 timeout 600 python tools/dp_cost_probe.py > $O/r06_dp_cost_probe.txt 2>&1
timeout 900 python tools/rccl_one_rank_check.py > $O/r06_rccl_one_rank.txt 2>&1
timeout 1500 python -m pytest tests/test_gpu_rccl_one_rank.py tests/test_gpu_resume.py -q > $O/r06_t1.log 2>&1; echo "rc=$?" >> $O/r06_t1.log
grep -v "amdgpu.ids\|socket.cpp" $O/r06_dp_cost_probe.txt; tail -n 8 $O/r06_t1.log; grep "^auto\|resume\[" $O/r06_rccl_one_rank.txt $O/r06_t1.log | cut -c1-300
