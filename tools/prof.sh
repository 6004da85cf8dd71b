#!/bin/bash
# usage: tools_prof.sh <tag> [bench args]   (runs on the GPU box; writes gpurun_out/<tag>.stats.txt)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
# (the profiler multiplies the host's enqueue time: SR_HIP_GRAPH=auto would switch the step to graph replay under it; profile the eager launches the driver's run makes)
export SR_HIP_GRAPH=0
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/$tag -o r -- python $GRAFT_REPO_ROOT/bench.py "$@" > $GRAFT_REPO_ROOT/gpurun_out/$tag.log 2>&1
python - <<PY
import sqlite3,glob
db=glob.glob("$GRAFT_REPO_ROOT/gpurun_out/$tag/*.db")[0]
c=sqlite3.connect(db)
rows=list(c.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))
with open("$GRAFT_REPO_ROOT/gpurun_out/$tag.stats.txt","w") as f:
    for n,cl,t,a,p in rows[:70]:
        f.write("%-100s %6d %10.0f %9.1f %6.2f\n"%(n[:100],cl,t,a,p))
    f.write("TOTAL_US %.0f\n"%sum(r[2] for r in rows))
# every dispatch's duration of the ten kernels with the most time, in dispatch order (tools/check_roofline_vs_rocprof.py compares the bench line's
# live per-launch time with the profiler's average over THE SAME launches -- the last N of the process, the roofline pass)
import json
top=[r[0] for r in rows[:10]]
dur={n:[d/1e3 for (d,) in c.execute("select (end-start) from kernels where name=? order by start",(n,))] for n in top}
json.dump(dur,open("$GRAFT_REPO_ROOT/gpurun_out/$tag.durations.json","w"))
cols=[r[1] for r in c.execute("pragma table_info(kernels)")]
with open("$GRAFT_REPO_ROOT/gpurun_out/$tag.dispatch.txt","w") as f:
    f.write(" ".join(cols)+"\n")
    q=list(c.execute("select name, grid_x, grid_y, workgroup_x, (end-start) as dur, start from kernels order by start"))
    q=q[-int(len(q)/7):]
    for n,gx,gy,wx,d,st in q:
        f.write("%-60s grid=%7d x %3d wg=%4d dur_us=%9.1f\n"%(n[:60],gx,gy,wx,d/1e3))
PY
rm -rf $GRAFT_REPO_ROOT/gpurun_out/$tag
