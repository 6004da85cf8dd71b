"""usage: check_roofline_vs_rocprof.py <bench line json> <rocprofv3 stats txt (tools/prof.sh)> [<durations json (tools/prof.sh)>] [tolerance, default 0.03]
The roofline object of bench.py times the dominant kernel with start / stop events bound to each dispatch (csrc/prof.hip).  When the bench line and
the profile come from the SAME process (bench.py run under rocprofv3 with its roofline pass on) the profiler's durations of the same launches -- the
last ``roofline.launches`` dispatches of that kernel in the process -- must agree with it: that is the check (exit 1 outside [-3 %, +6 %]).
The live figure may only err HIGH: the start event of hipExtLaunchKernel is a marker in front of the dispatch, so the pair also spans the 3-4 us
between that marker and the kernel's first wave -- roofline.frac is conservative by that much, never optimistic.  The
average over ALL launches of the process (the --stats line: tuning steps with other launch sizes included) and, for a line from a separate
unprofiled run, the profiler's slow-down of the whole step (rocprofv3 serialises part of the two-stream overlap: ~6.2 vs 4.9 ms per step) are
printed beside it so that nobody compares unlike populations."""
import json
import sys

line = json.load(open(sys.argv[1]))
durs = json.load(open(sys.argv[3])) if len(sys.argv) > 3 and sys.argv[3].endswith(".json") else None
tol = float(sys.argv[-1]) if sys.argv[-1].replace(".", "").isdigit() else 0.03
rf = line["roofline"]
want = rf["kernel"].replace(" ", "")
avg_all = None
for row in open(sys.argv[2]):
    if row.startswith("TOTAL_US"):
        continue
    name, calls, total, a, pct = row.rsplit(None, 4)
    if want in name.replace(" ", ""):
        avg_all = float(a)
        break
if avg_all is None:
    print("kernel %s not in %s" % (rf["kernel"], sys.argv[2]))
    sys.exit(1)
work = rf["flop_per_launch"] if rf["bound"] == "mfma" else rf["algorithmic_bytes_per_launch"]
unit = 1e12 if rf["bound"] == "mfma" else 1e9
rep = rf["frac"] * rf["peak"] * unit * rf["avg_launch_us"] * 1e-6 / work - 1.0
print("kernel %s\n  bench.py roofline (dispatch-bound events, %d launches of the roofline pass): %.2f us per launch, frac %.4f   [event pair around the call: %.2f us]"
      % (rf["kernel"], rf["launches"], rf["avg_launch_us"], rf["frac"], rf.get("avg_launch_us_event_pair", float("nan"))))
print("  rocprofv3 --stats, ALL launches of the process: %.2f us per launch -> frac %.4f" % (avg_all, work / (avg_all * 1e-6) / unit / rf["peak"]))
ok = abs(rep) <= tol
print("  frac x peak x avg vs work per launch: %+.3f %%" % (100 * rep))
if durs is not None:
    key = next((k for k in durs if want in k.replace(" ", "")), None)
    same = durs[key][-rf["launches"]:] if key else []
    if len(same) == rf["launches"]:
        avg_same = sum(same) / len(same)
        dev = rf["avg_launch_us"] / avg_same - 1.0
        print("  rocprofv3, the SAME %d launches (the last ones of the process): %.2f us per launch -> frac %.4f; live measurement deviates by %+.2f %% "
              "(accepted: -%.0f %% .. +%.0f %%; the live pair includes the dispatch gap in front of the kernel)"
              % (len(same), avg_same, work / (avg_same * 1e-6) / unit / rf["peak"], 100 * dev, 100 * tol, 200 * tol))
        ok = ok and -tol <= dev <= 2 * tol
    else:
        print("  (durations file has %d launches of the kernel, the roofline pass %d: not the same process?)" % (len(same), rf["launches"]))
        ok = False
sys.exit(0 if ok else 1)
