python -m pytest tests -m gpu -q --tb=short 2>&1 > gpurun_out/r03r_tests_full.txt; tail -5 gpurun_out/r03r_tests_full.txt
