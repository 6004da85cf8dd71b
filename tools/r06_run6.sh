#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out
timeout 1800 python -m pytest tests/test_gpu_srflexmatch.py -q -s -k "mask_identity or full_size_reference" > $O/r06_mask_identity.log 2>&1; echo "rc=$?" >> $O/r06_mask_identity.log
timeout 4000 python -m pytest tests -m gpu -q --deselect tests/test_gpu_srflexmatch.py::test_end_to_end_mask_identity_over_the_reference_sweep > $O/r06_gpu_suite.log 2>&1; echo "rc=$?" >> $O/r06_gpu_suite.log
grep "MASK_IDENTITY\|passed\|failed\|full trace" $O/r06_mask_identity.log | cut -c1-1200; tail -n 15 $O/r06_gpu_suite.log
