#!/bin/bash
# usage (GPU box): tools/timeline.sh <tag> [bench args] -> gpurun_out/<tag>.timeline.txt : phases of ONE steady-state step
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/$tag -o r -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-roofline "$@" > $GRAFT_REPO_ROOT/gpurun_out/$tag.log 2>&1
python - <<PY
import sqlite3, glob, re
db = glob.glob("$GRAFT_REPO_ROOT/gpurun_out/$tag/*.db")[0]
c = sqlite3.connect(db)
rows = list(c.execute("select name, stream_id, start, end, grid_x * grid_y * grid_z / (workgroup_x * workgroup_y * workgroup_z), workgroup_x * workgroup_y * workgroup_z, lds_size from kernels order by start"))
ends = [i for i, r in enumerate(rows) if "adamw_flat" in r[0]]
a, b = ends[-3] + 1, ends[-2] + 1
step = rows[a:b]
t0 = step[0][2]
short = lambda n: re.sub(r"\(anonymous namespace\)::|^void ", "", n)[:44]
with open("$GRAFT_REPO_ROOT/gpurun_out/$tag.timeline.txt", "w") as f:
    f.write("step wall (first start -> last end): %.1f us, %d kernels\n" % ((step[-1][3] - t0) / 1e3, len(step)))
    streams = sorted(set(r[1] for r in step))
    for s in streams:
        ks = [r for r in step if r[1] == s]
        busy = sum(r[3] - r[2] for r in ks) / 1e3
        f.write("stream %s: %d kernels, busy %.1f us, span %.1f .. %.1f us\n" % (s, len(ks), busy, (ks[0][2] - t0) / 1e3, (ks[-1][3] - t0) / 1e3))
    f.write("\n%10s %10s %8s %6s  kernel\n" % ("start_us", "end_us", "dur_us", "stream"))
    for n, s, st, en, wgs, wgsz, lds in step:
        f.write("%10.1f %10.1f %8.1f %6s  %5d wg x %4d thr lds %6d  %s\n" % ((st - t0) / 1e3, (en - t0) / 1e3, (en - st) / 1e3, s, wgs, wgsz, lds or 0, short(n)))
    # CU demand over time: a kernel asks for min(256, workgroups / workgroups-per-CU) CUs while it runs (workgroups per CU from its LDS and thread
    # count: a coarse estimate -- registers are not in the trace); bins of 50 us
    def per_cu(wgsz, lds):
        by_lds = 160 * 1024 // max(lds or 1, 1)
        by_thr = 2048 // max(wgsz, 64)
        return max(1, min(by_lds, by_thr, 8))
    T = (step[-1][3] - t0) / 1e3
    nb = int(T // 50) + 1
    dem = [0.0] * nb
    for n, s, st, en, wgs, wgsz, lds in step:
        cus = min(256.0, wgs / per_cu(wgsz, lds))
        a_, b_ = (st - t0) / 1e3, (en - t0) / 1e3
        for i in range(int(a_ // 50), min(nb, int(b_ // 50) + 1)):
            ov = max(0.0, min(b_, (i + 1) * 50) - max(a_, i * 50))
            dem[i] += cus * ov / 50.0
    f.write("\nCU demand per 50-us bin (sum over running kernels, capped at 256 per kernel; > 256 = oversubscribed, < 256 = idle CUs):\n")
    for i, d in enumerate(dem):
        f.write("%7d us  %6.0f  %s\n" % (i * 50, d, "#" * int(min(d, 512) / 8)))
    f.write("mean demand %.0f CUs; bins below 200 CUs: %d of %d\n" % (sum(dem) / nb, sum(1 for d in dem if d < 200), nb))
PY
rm -rf $GRAFT_REPO_ROOT/gpurun_out/$tag
