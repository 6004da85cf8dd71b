python -m pytest tests/test_gpu_wrn.py -q -x 2>&1 | tail -15
bash tools/prof.sh r03s_wrn --net wrn --bu 64 --steps 6 --warmup 2 --repeats 1 --no-cpu-baseline --no-roofline --no-also > /dev/null 2>&1
head -12 gpurun_out/r03s_wrn.stats.txt | cut -c1-150
grep -h '^{"metric"' gpurun_out/r03s_wrn.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
