"""A/B on one box: the K + 1 forwards of x_ulb_w of an SRPseudoLabel step (WRN-28-2, 64 / 64) as one launch train per pass vs sharing their launches
(WideResNet.forward_passes).  GPU box: python tools/wrn_pass_ab.py"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = ("import sys; sys.argv = ['bench.py', '--net', 'wrn', '--bu', '64', '--steps', '8', '--warmup', '3', '--repeats', '3', '--no-cpu-baseline', '--no-also'];"
        "import semireward_amd.algorithms.srpseudolabel as S; S._SHARE_PASS_LAUNCHES = %s; import runpy; runpy.run_path('bench.py', run_name='__main__')")
for i in range(2):
    for share in (False, True):
        r = subprocess.run([sys.executable, "-c", code % share], cwd=ROOT, capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if not line:
            print("share=%s FAILED" % share, r.stderr[-1500:]); continue
        o = json.loads(line[-1]); rf = o.get("roofline", {})
        print("share_pass_launches=%-5s  %.0f img/s  %.3f ms/step   %s %.1f us x %d  frac %.3f (%s)" % (
            share, o["value"], o["ms_per_step"], rf.get("kernel"), rf.get("avg_launch_us", 0), rf.get("launches", 0), rf.get("frac", 0), rf.get("bound")), flush=True)
