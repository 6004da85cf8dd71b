#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out
timeout 900 python tools/rccl_one_rank_check.py > $O/r06_rccl_one_rank.txt 2>&1
timeout 1800 python -m pytest tests/test_gpu_srflexmatch.py -q -s -k "mask_identity" > $O/r06_mask_identity.log 2>&1; echo "rc=$?" >> $O/r06_mask_identity.log
timeout 3000 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_srflexmatch.py::test_end_to_end_mask_identity_over_the_reference_sweep > $O/r06_gpu_suite.log 2>&1; echo "rc=$?" >> $O/r06_gpu_suite.log
grep "MASK_IDENTITY\|passed\|failed" $O/r06_mask_identity.log | cut -c1-900; tail -n 12 $O/r06_gpu_suite.log; grep "^schedule\|^auto" $O/r06_rccl_one_rank.txt | cut -c1-400
