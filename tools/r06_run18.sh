#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out
timeout 1800 python -m pytest tests/test_gpu_bert.py tests/test_gpu_w2v.py -q -x > $O/r06_t18.log 2>&1; tail -n 6 $O/r06_t18.log
one() { python bench.py --net $1 --no-cpu-baseline --no-roofline --no-also --steps 6 --warmup 2 --repeats 3 2>/dev/null | grep '^{' | python -c "import json,sys; o=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f %s  %.3f ms' % (o['value'], o['unit'], o['ms_per_step']))"; }
for i in 1 2; do echo -n "bert  "; one bert; echo -n "wave2vec  "; one wave2vec; done
echo -n "hubert  "; one hubert
