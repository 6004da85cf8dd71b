cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/g1_pytest.txt
python bench.py --no-also --no-cpu-baseline > gpurun_out/g1_bench.json 2> gpurun_out/g1_bench.err
bash tools/prof.sh g1_headline --steps 20 --warmup 5 --repeats 1 --no-cpu-baseline --no-roofline --no-also
python tools/check_roofline_vs_rocprof.py gpurun_out/g1_bench.json gpurun_out/g1_headline.stats.txt > gpurun_out/g1_roofcheck.txt 2>&1
cat gpurun_out/g1_pytest.txt gpurun_out/g1_roofcheck.txt; tail -c 1500 gpurun_out/g1_bench.json
