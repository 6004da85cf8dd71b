python -m pytest tests/test_gpu_wrn.py tests/test_gpu_stepgraph.py -q 2>&1 | tail -8 > gpurun_out/r03n_tests.txt
for g in 0 1; do SR_WRN_GRAPH=$g python bench.py --net wrn --bu 64 --steps 10 --warmup 3 --repeats 3 --no-cpu-baseline --no-roofline --no-also 2>&1 | tail -1 > gpurun_out/r03n_wrn_g$g.json; done
cat gpurun_out/r03n_tests.txt; for g in 0 1; do python -c "
import json,sys; d=json.loads(open('gpurun_out/r03n_wrn_g$g.json').read()); print($g, d['value'], d['ms_per_step'], d['config'].get('workload'))"; done
