python -m pytest tests -m gpu -q -x --durations=25 2>&1 | tail -45 > gpurun_out/r03p_tests.txt
( time python bench.py ) 2> gpurun_out/r03p_bench.err | tail -1 > gpurun_out/r03p_bench.json
tail -4 gpurun_out/r03p_bench.err
cat gpurun_out/r03p_tests.txt
