#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -k "gemm_tn" > $O/r06_t17.log 2>&1; tail -n 15 $O/r06_t17.log
timeout 600 python tools/gemm_tn_pp_bench.py 2>&1 | tail -n 20
