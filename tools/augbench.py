"""Throughput of srhip_augment on synthetic uint8 batches (GPU box)."""
import sys, time, numpy as np, torch
sys.path.insert(0, ".")
from semireward_amd.data.augment import GpuAugment
for S, pad, B in ((32, 4, 4096), (96, 12, 1024), (224, 28, 512)):
    aug = GpuAugment(S, pad, (0.5, 0.5, 0.5), (0.25, 0.25, 0.25), n_ops=3, device="cuda:0", seed=1)
    src = torch.randint(0, 256, (B, S, S, 3), dtype=torch.uint8, device="cuda:0")
    d = aug.draw(B, True)
    t0 = time.perf_counter(); ip, dp = aug.pack(d); t_pack = time.perf_counter() - t0
    for _ in range(3):
        aug(src, True, draws=d)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        aug(src, True, draws=d)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print("strong %dx%d B=%d: %.3f ms/batch incl. host packing (pack alone %.1f ms) -> %.0f img/s" % (S, S, B, ms, 1e3 * t_pack, B / ms * 1e3))
