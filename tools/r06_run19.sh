#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "ln_gemm or gemm_tn" > $O/r06_t19a.log 2>&1; tail -n 12 $O/r06_t19a.log
timeout 1500 python -m pytest tests/test_gpu_vit.py tests/test_gpu_srflexmatch.py -q -x > $O/r06_t19b.log 2>&1; tail -n 12 $O/r06_t19b.log
bash tools/ab_env.sh SR_FUSED_LN_GEMM 0 1 --steps 20 --warmup 3 --repeats 3 2>/dev/null
bash tools/ab_env.sh SR_FUSED_LN_GEMM 0 1 --steps 20 --warmup 3 --repeats 3 --regime pre 2>/dev/null
