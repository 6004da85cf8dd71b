#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "gemm_tn" > $O/r06_t21a.log 2>&1; tail -n 12 $O/r06_t21a.log
timeout 600 python tools/gemm_tn_pp_bench.py 2>&1 | grep -v amdgpu.ids | tail -n 20
timeout 1500 python -m pytest tests/test_gpu_vit.py tests/test_gpu_srflexmatch.py -q -x > $O/r06_t21b.log 2>&1; tail -n 5 $O/r06_t21b.log
bash tools/ab_env.sh SR_DW_PP 0 1 --steps 20 --warmup 3 --repeats 3 2>/dev/null
bash tools/ab_env.sh SR_DW_PP 0 1 --steps 20 --warmup 3 --repeats 3 --regime pre 2>/dev/null
