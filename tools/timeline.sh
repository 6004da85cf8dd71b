#!/bin/bash
# usage (GPU box): tools/timeline.sh <tag> [bench args] -> gpurun_out/<tag>.timeline.txt : phases of ONE steady-state step
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/$tag -o r -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-roofline "$@" > $GRAFT_REPO_ROOT/gpurun_out/$tag.log 2>&1
python - <<PY
import sqlite3, glob, re
db = glob.glob("$GRAFT_REPO_ROOT/gpurun_out/$tag/*.db")[0]
c = sqlite3.connect(db)
rows = list(c.execute("select name, stream_id, start, end from kernels order by start"))
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
    for n, s, st, en in step:
        f.write("%10.1f %10.1f %8.1f %6s  %s\n" % ((st - t0) / 1e3, (en - t0) / 1e3, (en - st) / 1e3, s, short(n)))
PY
rm -rf $GRAFT_REPO_ROOT/gpurun_out/$tag
