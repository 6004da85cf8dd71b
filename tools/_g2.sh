cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_srflexmatch.py tests/test_gpu_vit.py tests/test_gpu_stepgraph.py tests/test_gpu_dp_overlap.py -x -q 2>&1 | tail -8 > gpurun_out/g2_pytest.txt
B="python bench.py --no-also --no-cpu-baseline --no-roofline --repeats 3"
for cfg in "SR_EARLY_SUP_BWD=0" "SR_EARLY_SUP_BWD=1" "SR_EARLY_SUP_BWD=1 SR_EARLY_SUP_STREAM=1" "SR_EARLY_SUP_BWD=0" "SR_EARLY_SUP_BWD=1"; do
  for reg in sr pre; do
    echo "== $cfg regime=$reg" >> gpurun_out/g2_ab.txt
    env $cfg SR_PHASES=1 $B --regime $reg 2>> gpurun_out/g2_ab.txt | python -c "import sys,json; o=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(o['value'], o['ms_per_step'], o['repeats']['ms_per_step'], o['config'].get('deferred_share'), o['config'].get('deferred_images'))" >> gpurun_out/g2_ab.txt
  done
done
env SR_EARLY_SUP_BWD=1 $B --img 224 | python -c "import sys,json; o=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('224 early', o['value'], o['ms_per_step'])" >> gpurun_out/g2_ab.txt
env SR_EARLY_SUP_BWD=0 $B --img 224 | python -c "import sys,json; o=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('224 joint', o['value'], o['ms_per_step'])" >> gpurun_out/g2_ab.txt
bash tools/prof.sh g2_roof --steps 20 --warmup 5 --repeats 1 --no-cpu-baseline --no-also
grep -h '^{"metric"' gpurun_out/g2_roof.log > gpurun_out/g2_roof_bench.json
python tools/check_roofline_vs_rocprof.py gpurun_out/g2_roof_bench.json gpurun_out/g2_roof.stats.txt > gpurun_out/g2_roofcheck.txt 2>&1
cat gpurun_out/g2_pytest.txt gpurun_out/g2_ab.txt gpurun_out/g2_roofcheck.txt
