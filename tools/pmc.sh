#!/bin/bash
# usage: tools/pmc.sh <tag> "<counters>" <python args...>   -> gpurun_out/<tag>.pmc.txt (per-kernel averages)
tag=$1; shift; ctr=$1; shift
cd /tmp && export TMPDIR=/tmp
export SR_HIP_GRAPH=0          # eager launches, as in the driver's run (see tools/prof.sh)
rocprofv3 --pmc $ctr -d $GRAFT_REPO_ROOT/gpurun_out/$tag -o r -- python "$@" > $GRAFT_REPO_ROOT/gpurun_out/$tag.log 2>&1
python - <<PY
import sqlite3,glob,collections
db=glob.glob("$GRAFT_REPO_ROOT/gpurun_out/$tag/*.db")[0]
c=sqlite3.connect(db)
cols=[r[1] for r in c.execute("pragma table_info(counters_collection)")]
rows=list(c.execute("select * from counters_collection"))
ki=cols.index("kernel_name") if "kernel_name" in cols else None
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    d=dict(zip(cols,r))
    agg[d.get("kernel_name","?")[:70]][d.get("counter_name","?")].append(d.get("value",0))
with open("$GRAFT_REPO_ROOT/gpurun_out/$tag.pmc.txt","w") as f:
    f.write(" ".join(cols)+"\n")
    for k,v in agg.items():
        f.write(k+"\n")
        for cn,vals in sorted(v.items()):
            f.write("   %-28s n=%5d mean=%16.1f max=%16.1f\n"%(cn,len(vals),sum(vals)/len(vals),max(vals)))
PY
rm -rf $GRAFT_REPO_ROOT/gpurun_out/$tag
