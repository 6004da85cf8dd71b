cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q 2>&1 | tail -12 > gpurun_out/g10_pytest.txt
cat gpurun_out/g10_pytest.txt
python bench.py --net wrn --bu 64 --steps 8 --warmup 3 --repeats 3 --no-cpu-baseline --no-also 2>/dev/null | python -c "
import json,sys; o=json.loads(sys.stdin.read().strip().splitlines()[-1]); rf=o['roofline']
print('%.0f img/s %.3f ms | %s %.1f us x %d frac %.3f %s' % (o['value'], o['ms_per_step'], rf['kernel'], rf['avg_launch_us'], rf['launches'], rf['frac'], rf['bound']))"
