#!/bin/bash
# Same-box A/B of an environment switch:  tools/ab_env.sh VAR A_VALUE B_VALUE [bench args]   (GPU box; 3 alternating runs each, value + ms_per_step)
var=$1; a=$2; b=$3; shift 3
for i in 1 2 3; do for v in "$a" "$b"; do
  echo -n "$var=$v "; env $var=$v python bench.py --no-cpu-baseline --no-roofline --no-also "$@" | python -c "import json,sys; o=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f %s  %.3f ms' % (o['value'], o['unit'], o['ms_per_step']))"
done; done
