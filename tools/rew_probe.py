"""Standalone time of the rewarder scoring (two launches) for the K-pass call (8 groups of 8 rows) and the single-group call."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from semireward_amd.algorithms.semireward import Rewarder, label_dim
from tools.microbench import timeit

dev = "cuda:0"
rw = Rewarder(label_dim(100), 128, 384, device=dev)
for groups in (1, 8):
    f = torch.randn(groups * 8, 384, device=dev)
    y = torch.randint(0, 100, (groups * 8,), device=dev)
    print("groups=%d  %.1f us (embed + score launches, HIP events)" % (groups, timeit(lambda: rw.score(f, y, groups=groups), reps=100)))
