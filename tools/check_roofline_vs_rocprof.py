"""usage: check_roofline_vs_rocprof.py <bench line json> <rocprofv3 stats txt (tools/prof.sh)> [tolerance, default 0.03]
The roofline object of bench.py times the dominant kernel with start / stop events bound to each dispatch (csrc/prof.hip); rocprofv3 --kernel-trace
--stats averages the same kernel's execution time over the same command.  The two must agree, and frac x peak x avg must reproduce the per-launch
work: prints both, exits 1 when they differ by more than the tolerance."""
import json
import sys

line = json.load(open(sys.argv[1]))
tol = float(sys.argv[3]) if len(sys.argv) > 3 else 0.03
rf = line["roofline"]
want = rf["kernel"].replace(" ", "")
avg = None
for row in open(sys.argv[2]):
    if row.startswith("TOTAL_US"):
        continue
    name, calls, total, a, pct = row.rsplit(None, 4)
    if want in name.replace(" ", ""):
        avg = float(a)
        break
if avg is None:
    print("kernel %s not in %s" % (rf["kernel"], sys.argv[2]))
    sys.exit(1)
dev = rf["avg_launch_us"] / avg - 1.0
work = rf["flop_per_launch"] if rf["bound"] == "mfma" else rf["algorithmic_bytes_per_launch"]
unit = 1e12 if rf["bound"] == "mfma" else 1e9
rep = rf["frac"] * rf["peak"] * unit * rf["avg_launch_us"] * 1e-6 / work - 1.0
frac_prof = work / (avg * 1e-6) / unit / rf["peak"]
print("kernel %s\n  bench.py roofline: %.2f us per launch (event pair around the call: %.2f us), frac %.4f\n  rocprofv3 --stats: %.2f us per launch -> frac %.4f\n"
      "  deviation of the live measurement from the profiler: %+.2f %% (tolerance %.0f %%); frac x peak x avg vs work per launch: %+.3f %%" % (
          rf["kernel"], rf["avg_launch_us"], rf.get("avg_launch_us_event_pair", float("nan")), rf["frac"], avg, frac_prof, 100 * dev, 100 * tol,
          100 * rep))
sys.exit(0 if abs(dev) <= tol and abs(rep) <= tol else 1)
