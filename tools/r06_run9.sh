#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out
SR_HIP_GRAPH=1 tools/ab_env.sh SR_SPLIT_CHAIN 0 k0 --regime pre --steps 20 --warmup 3 --repeats 3 > $O/r06_split_chain_ab_k0_graph.txt 2>&1
SR_HIP_GRAPH=auto tools/ab_env.sh SR_SPLIT_CHAIN 0 k0 --regime pre --steps 20 --warmup 3 --repeats 3 > $O/r06_split_chain_ab_k0_auto.txt 2>&1
grep -v amdgpu.ids $O/r06_split_chain_ab_k0_graph.txt $O/r06_split_chain_ab_k0_auto.txt
