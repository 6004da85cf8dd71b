python -m pytest tests/test_gpu_wrn.py -q -x 2>&1 | tail -5
python tools/wrn_conv_bench.py 2>&1 | grep -v amdgpu.ids | cut -c1-150
for g in 0 1; do SR_WRN_GRAPH=$g python bench.py --net wrn --bu 64 --steps 10 --warmup 4 --repeats 3 --no-cpu-baseline --no-roofline --no-also 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('graph $g', d['value'], d['ms_per_step'])"; done
