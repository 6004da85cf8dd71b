#!/bin/bash
cd "$(dirname "$0")/.."
timeout 900 python -m pytest tests/test_gpu_bert.py tests/test_gpu_kernels.py -q -k "attention" 2>&1 | tail -n 3
python tools/attn_drop_probe.py 2>&1 | grep bert
