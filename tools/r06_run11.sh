#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_bert.py tests/test_gpu_kernels.py -q -k "attention" > $O/r06_t1.log 2>&1; tail -n 6 $O/r06_t1.log
echo "== paired-tile forward"; python tools/attn_drop_probe.py 2>&1 | grep bert
echo "== whole-row forward"; SRHIP_ATTN_FWD_WHOLE_ROW=1 python tools/attn_drop_probe.py 2>&1 | grep bert
