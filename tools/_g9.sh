cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_wrn.py -q -x 2>&1 | tail -6 > gpurun_out/g9_pytest.txt
cat gpurun_out/g9_pytest.txt
for s in 103 147 123 121; do
  echo "######## candidate $s" >> gpurun_out/g9_cand.txt
  python tools/full_trace_diag.py gpurun_in_full_trace_$s.npz 2>&1 | grep -v amdgpu.ids | grep -E "^==|<--|loss|gradient|reward" >> gpurun_out/g9_cand.txt
done
cat gpurun_out/g9_cand.txt
bash tools/prof.sh g9_wrn --steps 8 --warmup 3 --repeats 1 --no-cpu-baseline --no-roofline --no-also --net wrn --bu 64
grep -h '^{"metric"' gpurun_out/g9_wrn.log | python -c "import json,sys; o=json.loads(sys.stdin.read()); print(o['value'], o['ms_per_step'])"
head -16 gpurun_out/g9_wrn.stats.txt | cut -c1-75,100-140
python bench.py --net wrn --bu 64 --steps 8 --warmup 3 --repeats 3 --no-cpu-baseline --no-also 2>/dev/null | python -c "
import json,sys; o=json.loads(sys.stdin.read().strip().splitlines()[-1]); rf=o['roofline']
print('%.0f img/s %.3f ms | %s %.1f us x %d frac %.3f %s' % (o['value'], o['ms_per_step'], rf['kernel'], rf['avg_launch_us'], rf['launches'], rf['frac'], rf['bound']))"
