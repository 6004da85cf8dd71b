#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out
export SR_DEFER_FRACTION=0.5
tools/ab_env.sh SR_GRAD_PRIO 0 1 --steps 20 --warmup 3 --repeats 3 > $O/r06_grad_prio_ab.txt 2>&1
for v in 0 1; do SR_GRAD_PRIO=$v SR_PHASES=1 python bench.py --no-cpu-baseline --no-roofline --no-also --steps 20 --repeats 2 2>&1 | grep -i "^phases" >> $O/r06_grad_prio_ab.txt; done
grep -v amdgpu $O/r06_grad_prio_ab.txt
