cd $GRAFT_REPO_ROOT
cp semireward_amd/libsrhip_B.so semireward_amd/libsrhip.so
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_vit.py tests/test_gpu_srflexmatch.py tests/test_gpu_w2v.py tests/test_gpu_bert.py tests/test_gpu_driver_contract.py tests/test_gpu_dp_overlap.py -x -q 2>&1 | tail -8 > gpurun_out/g3_pytest.txt
cat gpurun_out/g3_pytest.txt
for i in 1 2 3; do for v in A B; do
  cp semireward_amd/libsrhip_$v.so semireward_amd/libsrhip.so
  for reg in sr pre; do
  echo -n "$v $reg " >> gpurun_out/g3_ab.txt; python bench.py --no-cpu-baseline --no-roofline --no-also --repeats 3 --regime $reg | python -c "import json,sys; o=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f  %.3f ms  share %s' % (o['value'], o['ms_per_step'], o['config'].get('deferred_share')))" >> gpurun_out/g3_ab.txt
  done
done; done
cp semireward_amd/libsrhip_B.so semireward_amd/libsrhip.so
python bench.py --no-cpu-baseline --no-also --repeats 1 > gpurun_out/g3_bench_B.json 2>/dev/null
python - <<'PY' >> gpurun_out/g3_ab.txt
import json
a=json.load(open("gpurun_out/g3_bench_B.json")); r=a["roofline"]
print("B roofline", r["kernel"], r["avg_launch_us"], r["frac"])
for o in r["other_kernels"]: print("  %-36s %6.1f us x %4d  %.3f %s  %.3f ms/step"%(o["kernel"],o["avg_launch_us"],o["launches"],o["frac"],o["bound"],o["ms_per_step_in_kernel"]))
PY
cat gpurun_out/g3_ab.txt
