"""Four runs of the same 16 SRFlexMatch steps (two eager, two HIP-graph replayed) side by side: shows where fp32-atomic ordering makes any two
runs part ways (a 0/1 mask flips), which bounds what tests/test_gpu_stepgraph.py may compare.  GPU box: python tools/stepgraph_diag.py"""
import sys, types
sys.path[:0] = ["tests", "."]
import pytest, torch, numpy as np
import test_gpu_stepgraph as T
class MP:
    def setattr(self, o, n, v): setattr(o, n, v)
mp = MP()
runs = {}
for name, g in (("eager_a", False), ("eager_b", False), ("graph_a", True), ("graph_b", True)):
    _, sg, r = T._run(g, 30008, 16, mp)
    runs[name] = r
    print(name, "replays", getattr(sg, "replays", None))
    for i, x in enumerate(r):
        print("  %2d" % i, " ".join("%9.6f" % v for v in x["loss"]), "maxr %.6f" % x["maxr"])
