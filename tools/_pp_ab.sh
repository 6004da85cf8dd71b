#!/bin/bash
# legs under the two-wave-group GEMM (default) and the lockstep persistent kernel (SRHIP_GEMM=bigold), alternating on one box
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu 2>&1 | tail -3
for i in 1 2; do for m in default bigold; do
  for net in bert wave2vec; do
    echo -n "gemm=$m $net: "; if [ $m = default ]; then unset SRHIP_GEMM; else export SRHIP_GEMM=$m; fi
    timeout 600 python bench.py --net $net --steps 5 --warmup 2 --repeats 3 --no-cpu-baseline --no-roofline --no-also | python -c "import json,sys; o=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f %s  %.3f ms' % (o['value'], o['unit'], o['ms_per_step']))"
  done
  echo -n "gemm=$m vit bu64: "; timeout 600 python bench.py --bu 64 --steps 5 --warmup 2 --repeats 3 --no-cpu-baseline --no-roofline --no-also | python -c "import json,sys; o=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f %s  %.3f ms' % (o['value'], o['unit'], o['ms_per_step']))"
  echo -n "gemm=$m vit headline: "; timeout 600 python bench.py --steps 20 --warmup 3 --repeats 3 --no-cpu-baseline --no-roofline --no-also | python -c "import json,sys; o=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f %s  %.3f ms' % (o['value'], o['unit'], o['ms_per_step']))"
done; done
