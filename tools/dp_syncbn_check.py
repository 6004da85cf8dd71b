"""SyncBatchNorm property of the WideResNet engine under data parallel (the reference converts its BatchNorms under DDP, core/utils/misc.py:55):
two ranks with half of a batch each == one rank with the whole batch -- logits of the rank's rows, running statistics, and the SUM over the
ranks of the parameter gradients (the ranks exchange every BatchNorm's sums in the forward and the two column sums of its backward).
Run: python -m torch.distributed.run --nproc-per-node 2 tools/dp_syncbn_check.py   (gloo on one GPU is enough: SR_DIST_BACKEND=gloo)"""
import os
import sys
import zlib

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from semireward_amd.distributed import DataParallel      # noqa: E402
from semireward_amd.nets import wrn                      # noqa: E402

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0 if os.environ.get("SR_DIST_BACKEND", "nccl") == "gloo" else int(os.environ.get("LOCAL_RANK", "0")))
dist.init_process_group(os.environ.get("SR_DIST_BACKEND", "nccl"), rank=rank, world_size=world)
dev = torch.device("cuda", torch.cuda.current_device())
rng = np.random.Generator(np.random.PCG64(11))
B, HW, C = 4 * world, 16, 10
x = torch.from_numpy(rng.standard_normal((B, 3, HW, HW)).astype(np.float32)).to(dev)
dl = torch.from_numpy((rng.standard_normal((B, C)) / B).astype(np.float32)).to(dev)


def build():
    m = wrn.WideResNet(num_classes=C, depth=10, widen_factor=2, first_stride=1, device=dev)
    m.init_weights(seed=3)
    for n, _ in m.names_shapes:                           # BatchNorm scales / shifts away from 1 / 0 so that they matter
        if "bn" in n:
            g = torch.Generator().manual_seed(zlib.crc32(n.encode()) % 100000)        # (hash(str) differs from process to process)
            m.view(n).copy_((1.0 if n.endswith("weight") else 0.0) + 0.2 * torch.randn(m.view(n).shape, generator=g))
    m.refresh_operands()
    m.train()
    return m


full = build()                                            # the whole batch on one rank (every rank computes it for itself)
lg_f, ft_f, ctx_f = full.forward_features(x, save=True, update_stats=True, tag="full")
full.backward(ctx_f, dl)
part = build()
part.dp = DataParallel(world, rank)
assert part.stat_ranks == world
sl = slice(rank * 4, rank * 4 + 4)
lg_p, ft_p, ctx_p = part.forward_features(x[sl].contiguous(), save=True, update_stats=True, tag="part")
part.backward(ctx_p, dl[sl].contiguous())
g = part.grad.clone()
dist.all_reduce(g)                                        # SUM over the ranks (the engine's exchange; 1 / world rides in the optimizer launch)
torch.cuda.synchronize()
rel = lambda a, b: float((a - b).double().norm() / (b.double().norm() + 1e-30))   # noqa: E731
e_lg, e_ft, e_g = rel(lg_p, lg_f[sl]), rel(ft_p, ft_f[sl]), rel(g, full.grad)
e_rs = max(rel(part.buffers[k].float(), full.buffers[k].float()) for k in full.buffers if "running" in k)
nbt = all(int(part.buffers[k]) == int(full.buffers[k]) == 1 for k in full.buffers if k.endswith("num_batches_tracked"))
# without the exchange the half-batch statistics are different statistics: the same comparison must then FAIL (the test's power)
loc = build()
lg_l, _, _ = loc.forward_features(x[sl].contiguous(), save=False, update_stats=False, tag="loc")
e_loc = rel(lg_l, lg_f[sl])
ok = e_lg < 2e-3 and e_ft < 2e-3 and e_g < 5e-3 and e_rs < 1e-5 and nbt and e_loc > 10 * max(e_lg, 1e-6)
print("rank %d: logits %.2e feat %.2e grad(sum over ranks) %.2e running stats %.2e | per-rank statistics would give %.2e | syncbn == whole batch: %s"
      % (rank, e_lg, e_ft, e_g, e_rs, e_loc, ok), flush=True)
assert ok
# unequal per-rank batches (a last partial batch): every rank enters the same collectives (the row counts ride on the first statistics exchange of
# the forward) and EVERY rank gets the error -- neither a hang nor silently skewed statistics.  One rank sees a batch size it has used before, the
# other a new one: the case a per-batch-size cache of a separate check would turn into mismatched collectives.
bad = x[:4 if rank == 0 else 3].contiguous()
try:
    part.forward_features(bad, save=False, update_stats=False, tag="bad")      # (SRHIP_CHECK_ARGS=1, the test suite: raised inside the forward)
    part.check_equal_rows()
    caught = False
except RuntimeError as e:
    caught = "per-rank batches" in str(e)
part.forward_features(x[sl].contiguous(), save=False, update_stats=False, tag="part2")      # equal again: the flag was cleared, no error
part.check_equal_rows()
print("rank %d: unequal per-rank batches raise on every rank: %s" % (rank, caught), flush=True)
assert caught
# the same on the shared-launch path (passes > 1: the K + 1 frozen forwards of an SRPseudoLabel step, its default): the counts ride on the first
# accumulator block of the pass arena
try:
    part.forward_features(bad, save=False, update_stats=False, tag="badp", passes=3)
    part.check_equal_rows()
    caught_p = False
except RuntimeError as e:
    caught_p = "per-rank batches" in str(e)
lg_3, _, _ = part.forward_features(x[sl].contiguous(), save=False, update_stats=False, tag="part3", passes=3)
part.check_equal_rows()
torch.cuda.synchronize()
same = all(torch.equal(lg_3[:4], lg_3[4 * g_:4 * g_ + 4]) for g_ in (1, 2))        # three passes of one batch under frozen statistics: identical
print("rank %d: unequal per-rank batches raise on every rank (shared launches): %s" % (rank, caught_p and same), flush=True)
assert caught_p and same
dist.barrier()
dist.destroy_process_group()
