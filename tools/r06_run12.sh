#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out
one() { python bench.py --net bert --no-cpu-baseline --no-roofline --no-also --steps 6 --warmup 2 --repeats 3 2>/dev/null | grep '^{' | python -c "import json,sys; o=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f %s  %.3f ms' % (o['value'], o['unit'], o['ms_per_step']))"; }
for i in 1 2 3; do
  echo -n "paired-tile forward  "; one
  echo -n "whole-row forward    "; SRHIP_ATTN_FWD_WHOLE_ROW=1 one
done
