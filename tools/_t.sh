python -m pytest tests/test_gpu_bench_launch.py -q -x 2>&1 | tail -5
( time python bench.py ) 2>&1 | tail -4 | cut -c1-300
