python -m pytest tests -m gpu -q --durations=15 --deselect tests/test_gpu_attn_block.py --deselect tests/test_gpu_augment.py 2>&1 | tail -30 > gpurun_out/r03q_tests.txt
bash tools/prof.sh r03q_wrn --net wrn --bu 64 --steps 6 --warmup 2 --repeats 1 --no-cpu-baseline --no-roofline --no-also > /dev/null 2>&1
head -14 gpurun_out/r03q_wrn.stats.txt | cut -c1-150
grep -h '^{"metric"' gpurun_out/r03q_wrn.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
cat gpurun_out/r03q_tests.txt
