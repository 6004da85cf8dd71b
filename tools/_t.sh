SR_BENCH_LEGS=vit_s16_224 SR_PHASES=1 python bench.py --gpus 2 --steps 6 --warmup 2 --repeats 2 --no-allreduce-ab --no-roofline 2>gpurun_out/r03_224x2.err | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step']); print(d['also'])"
grep -i "phases\|Error\|warn" gpurun_out/r03_224x2.err | cut -c1-400 | head
