#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out
tools/ab_env.sh SR_SPLIT_CHAIN 0 1 --steps 20 --warmup 3 --repeats 3 > $O/r06_split_chain_ab.txt 2>&1
tools/ab_env.sh SR_SPLIT_CHAIN 0 1 --regime pre --steps 20 --warmup 3 --repeats 3 > $O/r06_split_chain_ab_k0.txt 2>&1
SR_HIP_GRAPH=1 tools/ab_env.sh SR_SPLIT_CHAIN 0 1 --regime pre --steps 20 --warmup 3 --repeats 3 > $O/r06_split_chain_ab_k0_graph.txt 2>&1
SR_PHASES=1 SR_SPLIT_CHAIN=1 timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-also --steps 20 --repeats 2 2>&1 | grep -i "phases" > $O/r06_split_chain_phases.txt
SR_PHASES=1 SR_SPLIT_CHAIN=1 timeout 300 python bench.py --regime pre --no-cpu-baseline --no-roofline --no-also --steps 20 --repeats 2 2>&1 | grep -i "phases" >> $O/r06_split_chain_phases.txt
timeout 1200 python -m pytest tests/test_gpu_srflexmatch.py tests/test_gpu_vit.py tests/test_gpu_resume.py tests/test_gpu_stepgraph.py -q -x > $O/r06_t1.log 2>&1; echo "rc=$?" >> $O/r06_t1.log
grep -v amdgpu.ids $O/r06_split_chain_ab.txt $O/r06_split_chain_ab_k0.txt $O/r06_split_chain_ab_k0_graph.txt; cat $O/r06_split_chain_phases.txt; tail -n 6 $O/r06_t1.log
