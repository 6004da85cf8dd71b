#!/bin/bash
cd "$(dirname "$0")/.."
bash tools/ab_env.sh SR_WT_ON_MAIN 0 1 --steps 20 --warmup 3 --repeats 3 2>/dev/null
for v in 0 1; do echo "SR_WT_ON_MAIN=$v"; SR_WT_ON_MAIN=$v SR_PHASES=1 python bench.py --no-also --no-cpu-baseline --no-roofline --repeats 2 2>&1 | grep -i "^phases" | tail -1; done
bash tools/ab_env.sh SR_WT_ON_MAIN 0 1 --net bert --steps 6 --warmup 2 --repeats 2 2>/dev/null
bash tools/ab_env.sh SR_WT_ON_MAIN 0 1 --img 224 --steps 20 --warmup 3 --repeats 2 2>/dev/null
