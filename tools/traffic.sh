#!/bin/bash
# usage (on the GPU box): tools/traffic.sh <tag> [bench args]  -> gpurun_out/<tag>_hbm_traffic.json
# HBM bytes per launch and kernel from the L2 fabric-side counters, collected as MI355X_MICROARCH.md (HBM section) prescribes:
# FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes (TCC slots), FETCH_SIZE doubled on gfx950, WRITE_SIZE as reported.
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
export SR_HIP_GRAPH=0          # eager launches, as in the driver's run (see tools/prof.sh)
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c -d $GRAFT_REPO_ROOT/gpurun_out/${tag}_$c -o r -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline "$@" > $GRAFT_REPO_ROOT/gpurun_out/${tag}_$c.log 2>&1
done
python - <<PY
import sqlite3, glob, collections, json, re
out = collections.OrderedDict()
vals = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    db = glob.glob("$GRAFT_REPO_ROOT/gpurun_out/${tag}_%s/*.db" % c)[0]
    con = sqlite3.connect(db)
    cols = [r[1] for r in con.execute("pragma table_info(counters_collection)")]
    agg = collections.defaultdict(list)
    for r in con.execute("select * from counters_collection"):
        d = dict(zip(cols, r))
        if d.get("counter_name") != c:
            continue
        name = d.get("kernel_name", "?")
        m = re.search(r"(\w+_kernel(?:<[^>]*>)?)", name)
        agg[m.group(1) if m else name[:60]].append(d.get("value", 0.0))
    vals[c] = agg
for k in vals["FETCH_SIZE"]:
    f, w = vals["FETCH_SIZE"][k], vals["WRITE_SIZE"].get(k, [0.0])
    fk, wk = sum(f) / len(f), sum(w) / len(w)
    out[k] = {"launches_sampled": len(f), "FETCH_SIZE_KB_raw": round(fk, 1), "WRITE_SIZE_KB_raw": round(wk, 1),
              "hbm_read_bytes_per_launch": 2.0 * fk * 1024, "hbm_write_bytes_per_launch": wk * 1024,
              "hbm_bytes_per_launch": 2.0 * fk * 1024 + wk * 1024,
              "correction": "FETCH_SIZE x2 (gfx950 rocprofv3 counts 128-B requests at 64 B, MI355X_MICROARCH.md HBM section); WRITE_SIZE as reported (uncalibrated)"}
out = collections.OrderedDict(sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches_sampled"]))
json.dump(out, open("$GRAFT_REPO_ROOT/gpurun_out/${tag}_hbm_traffic.json", "w"), indent=1)
PY
rm -rf $GRAFT_REPO_ROOT/gpurun_out/${tag}_FETCH_SIZE $GRAFT_REPO_ROOT/gpurun_out/${tag}_WRITE_SIZE
