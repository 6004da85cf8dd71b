python -m pytest tests/test_gpu_wrn.py tests/test_gpu_dp_overlap.py -q -x 2>&1 | tail -2
python bench.py --net wrn --bu 64 --steps 10 --warmup 4 --repeats 3 --no-cpu-baseline --no-roofline --no-also 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
