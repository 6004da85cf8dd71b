python -m pytest tests/test_gpu_wrn.py tests/test_gpu_dp_overlap.py -q -x 2>&1 | tail -25
python bench.py --net wrn --bu 64 --steps 10 --warmup 4 --repeats 3 --no-cpu-baseline --no-roofline --no-also 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
python bench.py --gpus 2 --net wrn --bu 64 --steps 4 --warmup 2 --repeats 1 --no-cpu-baseline --no-roofline --no-also 2>&1 | tail -1 | cut -c1-400
