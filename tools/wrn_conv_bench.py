"""Times srhip_wrn_conv_bn on the WRN-28-2 layer shapes of the classic_cv batch (64 images of 32x32), with and without the statistics
epilogue, against the unfused chain (bn_fwd + im2col + gemm_nt).  GPU box: python tools/wrn_conv_bench.py"""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from semireward_amd import ops  # noqa: E402

DEV = "cuda:0"


def timeit(fn, n=100):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000.0 / n


for (B, H, Cin, Cout, ks, stride) in [(64, 32, 32, 32, 3, 1), (64, 32, 16, 32, 3, 1), (64, 32, 32, 64, 3, 2), (64, 16, 64, 64, 3, 1),
                                       (64, 16, 64, 128, 3, 2), (64, 8, 128, 128, 3, 1), (64, 32, 32, 64, 1, 2)]:
    rng = np.random.Generator(np.random.PCG64(1))
    T = lambda a: torch.from_numpy(np.asarray(a, dtype=np.float32)).to(DEV).contiguous()   # noqa: E731
    rows_in = B * H * H
    x = T(rng.standard_normal((rows_in, Cin)))
    gam, bet = T(np.ones(Cin)), T(np.zeros(Cin))
    mean, invstd = T(np.zeros(Cin)), T(np.ones(Cin))
    K, Kp = Cin * ks * ks, (Cin * ks * ks + 31) // 32 * 32
    Ho = (H + 2 * (ks // 2) - ks) // stride + 1
    rows = B * Ho * Ho
    Wb = (torch.randn(Cout, Kp, device=DEV) * 0.05).to(torch.bfloat16)
    y = torch.empty(rows, Cout, device=DEV)
    res = torch.randn(rows, Cout, device=DEV)
    ws = torch.zeros(ops.bn_ws_doubles(), dtype=torch.float64, device=DEV)
    om, oi = torch.empty(Cout, device=DEV), torch.empty(Cout, device=DEV)
    rm, rv = torch.zeros(Cout, device=DEV), torch.ones(Cout, device=DEV)
    acc_in = torch.zeros(ops.bn_acc_doubles(Cin), dtype=torch.float64, device=DEV)
    acc_in.view(16, 2 * Cin)[:, Cin:] = rows_in / 16.0                 # mean 0, variance 1
    acc_out = torch.zeros(ops.bn_acc_doubles(Cout), dtype=torch.float64, device=DEV)
    pm, pi = torch.empty(Cin, device=DEV), torch.empty(Cin, device=DEV)
    t_stats = timeit(lambda: ops.wrn_conv_bn(x, 3, None, acc_in, gam, bet, 1e-5, 0.1, Wb, res, y, B, H, H, Cin, Cout, ks, stride, Kp,
                                             publish=(pm, pi), running=(rm[:Cin] if Cin <= Cout else None, rv[:Cin] if Cin <= Cout else None),
                                             momentum=0.001, update_running=False, acc_out=acc_out))
    t_fold = timeit(lambda: ops.wrn_conv_bn(x, 3, None, acc_in, gam, bet, 1e-5, 0.1, Wb, res, y, B, H, H, Cin, Cout, ks, stride, Kp))
    t_sums = timeit(lambda: ops.wrn_conv_bn(x, 0, (mean, invstd), None, gam, bet, 1e-5, 0.1, Wb, res, y, B, H, H, Cin, Cout, ks, stride, Kp, acc_out=acc_out))
    t_plain = timeit(lambda: ops.wrn_conv_bn(x, 0, (mean, invstd), None, gam, bet, 1e-5, 0.1, Wb, res, y, B, H, H, Cin, Cout, ks, stride, Kp))
    t_raw = timeit(lambda: ops.wrn_conv_bn(x, 2, None, None, None, None, 0.0, 0.1, Wb, None, y, B, H, H, Cin, Cout, ks, stride, Kp))
    act = torch.empty(rows_in, Cin, dtype=torch.bfloat16, device=DEV)
    col = torch.empty(rows, Kp, dtype=torch.bfloat16, device=DEV)
    sm, si = torch.empty(Cin, device=DEV), torch.empty(Cin, device=DEV)

    def chain():
        ops.bn_fwd(x, gam, bet, 1e-5, 0.1, 0.001, True, False, rm[:Cin] if Cin <= Cout else torch.zeros(Cin, device=DEV), rv[:Cin] if Cin <= Cout else torch.ones(Cin, device=DEV),
                   sm, si, act, None, ws, rows_in, Cin)
        ops.im2col(act, col, B, H, H, Cin, ks, stride, Kp)
        ops.gemm_nt(ops.EPI_RESID_F32, col, Wb, y, rows, Cout, Kp, aux_in=res, ldaux=Cout)
    t_chain = timeit(chain)
    flop = 2.0 * rows * K * Cout
    print("B%d %dx%d Cin %3d Cout %3d k%d s%d: fold only %6.1f | sums only %6.1f | fold + sums %6.1f us | given stats, no sums %6.1f us | raw input %6.1f us | unfused chain %6.1f us | %5.1f TF/s, input %.1f MB"
          % (B, H, H, Cin, Cout, ks, stride, t_fold, t_sums, t_stats, t_plain, t_raw, t_chain, flop / t_plain * 1e-6, rows_in * Cin * 4 / 1e6), flush=True)
