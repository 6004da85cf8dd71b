"""Micro-benchmarks of single libsrhip kernels on the GPU box (HIP-event timing, median of reps)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from semireward_amd import ops

DEV = "cuda:0"


def timeit(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


def attn(N=257, H=6):
    for B in (7, 43, 86, 128, 200):
        D = H * 64
        qkv = torch.randn(B * N, 3 * D, device=DEV).to(torch.bfloat16)
        out = torch.empty(B * N, D, dtype=torch.bfloat16, device=DEV)
        t = timeit(lambda: ops.attn_fwd(qkv, out, None, B, N, H, 0.125))
        fl = 4.0 * N * N * 64 * B * H
        print("attn_fwd B=%4d WGs=%5d  %8.1f us  %7.1f TF/s" % (B, B * H, t, fl / t / 1e6), flush=True)


def gemm():
    for (M, N, K, epi, name) in [(51400, 1152, 384, ops.EPI_BF16, "qkv"), (51400, 384, 384, ops.EPI_RESID_F32, "proj"),
                                 (51400, 1536, 384, ops.EPI_GELU_BF16, "fc1"), (51400, 384, 1536, ops.EPI_RESID_F32, "fc2"),
                                 (51400, 1536, 384, ops.EPI_BF16, "fc1-nogelu"), (8192, 8192, 8192, ops.EPI_BF16, "big"),
                                 (13952, 2304, 768, ops.EPI_BF16, "bert qkv"), (13952, 768, 768, ops.EPI_RESID_F32, "bert proj"),
                                 (13952, 3072, 768, ops.EPI_GELU_BF16, "bert fc1"), (13952, 768, 3072, ops.EPI_RESID_F32, "bert fc2"),
                                 (13952, 3072, 768, ops.EPI_BF16, "bert fc1-nogelu"), (4096, 4096, 4096, ops.EPI_BF16, "4k")]:
        A = torch.randn(M, K, device=DEV).to(torch.bfloat16)
        Bm = (torch.randn(N, K, device=DEV) * 0.05).to(torch.bfloat16)
        bias = torch.randn(N, device=DEV)
        C = torch.zeros(M, N, dtype=torch.float32 if epi == ops.EPI_RESID_F32 else torch.bfloat16, device=DEV)
        t = timeit(lambda: ops.gemm_nt(epi, A, Bm, C, M, N, K, bias=bias), reps=10)
        print("gemm %-10s M=%6d N=%5d K=%5d  %8.1f us  %7.1f TF/s" % (name, M, N, K, t, 2.0 * M * N * K / t / 1e6), flush=True)


def gemm_small():
    """The M = 4112-row products of the backward (16 gradient-carrying images)."""
    for (M, N, K, epi, name) in [(4112, 384, 1152, ops.EPI_BF16, "dqkv->dx"), (4112, 384, 384, ops.EPI_BF16, "dproj->do"),
                                 (4112, 1536, 384, ops.EPI_DGELU_BF16, "dfc2->dh"), (4112, 384, 1536, ops.EPI_BF16, "dfc1->dx"),
                                 (4112, 1152, 384, ops.EPI_BF16, "qkv fwd"), (4112, 1536, 384, ops.EPI_GELU_BF16, "fc1 fwd"),
                                 (4112, 384, 1536, ops.EPI_RESID_F32, "fc2 fwd")]:
        A = torch.randn(M, K, device=DEV).to(torch.bfloat16)
        Bm = (torch.randn(N, K, device=DEV) * 0.05).to(torch.bfloat16)
        bias = torch.randn(N, device=DEV)
        C = torch.zeros(M, N, dtype=torch.float32 if epi == ops.EPI_RESID_F32 else torch.bfloat16, device=DEV)
        aux = torch.randn(M, N, device=DEV).to(torch.bfloat16) if epi == ops.EPI_DGELU_BF16 else None
        t = timeit(lambda: ops.gemm_nt(epi, A, Bm, C, M, N, K, bias=bias, aux_in=aux), reps=20)
        print("gemm %-10s M=%6d N=%5d K=%5d  %8.1f us  %7.1f TF/s" % (name, M, N, K, t, 2.0 * M * N * K / t / 1e6), flush=True)


def mlp():
    """Fused LN2+fc1+GELU+fc2+residual vs the three unfused launches (inference rows of the reference batch)."""
    M, D, Hd = int(os.environ.get("MB_ROWS", "51400")), 384, 1536      # MB_ROWS=26985: the 105 read images of a step, one round of tiles
    x = torch.randn(M, D, device=DEV)
    g, b = torch.rand(D, device=DEV) + 0.5, torch.randn(D, device=DEV) * 0.1
    W1 = (torch.randn(Hd, D, device=DEV) * 0.05).to(torch.bfloat16)
    W2 = (torch.randn(D, Hd, device=DEV) * 0.02).to(torch.bfloat16)
    b1, b2 = torch.randn(Hd, device=DEV) * 0.1, torch.randn(D, device=DEV) * 0.1
    ln = torch.empty(M, D, dtype=torch.bfloat16, device=DEV)
    h = torch.empty(M, Hd, dtype=torch.bfloat16, device=DEV)

    def unfused():
        ops.layernorm_fwd(x, g, b, 1e-6, ln, None, None, M, D)
        ops.gemm_nt(ops.EPI_GELU_BF16, ln, W1, h, M, Hd, D, bias=b1)
        ops.gemm_nt(ops.EPI_RESID_F32, h, W2, x, M, D, Hd, bias=b2)
    t0 = timeit(unfused, reps=10)
    x.normal_()
    t1 = timeit(lambda: ops.mlp_fused(x, g, b, 1e-6, W1, b1, W2, b2, None, 0, M, D, Hd), reps=10)
    if os.environ.get("SRHIP_TUNING_BUILD"):
        for dbg in (1, 2, 3, 7, 11, 15, 100, 101):
            os.environ["SRHIP_MLP_DEBUG"] = str(dbg)
            t = timeit(lambda: ops.mlp_fused(x, g, b, 1e-6, W1, b1, W2, b2, None, 0, M, D, Hd), reps=5)
            print("  debug=%3d (1 no GELU, 2 no DMA, 4 no ds_read, 8 no MFMA; 100: GS=2, 101: GS=1): %8.1f us" % (dbg, t), flush=True)
        os.environ.pop("SRHIP_MLP_DEBUG")
    fl = 4.0 * M * D * Hd
    print("mlp unfused %8.1f us %7.1f TF/s | fused %8.1f us %7.1f TF/s" % (t0, fl / t0 / 1e6, t1, fl / t1 / 1e6), flush=True)


def attn_block():
    """Fused norm1 + qkv + attention (one workgroup per image) vs the three launches it replaces, inference rows of a step."""
    D, H = 384, 6
    for N in (257, 197):
        for B in (73, 127, 200):
            M = B * N
            x = torch.randn(M, D, device=DEV)
            g, b = torch.rand(D, device=DEV) + 0.5, torch.randn(D, device=DEV) * 0.1
            W = (torch.randn(3 * D, D, device=DEV) * 0.05).to(torch.bfloat16)
            bq = torch.randn(3 * D, device=DEV) * 0.1
            ln = torch.empty(M, D, dtype=torch.bfloat16, device=DEV)
            qkv = torch.empty(M, 3 * D, dtype=torch.bfloat16, device=DEV)
            out = torch.empty(M, D, dtype=torch.bfloat16, device=DEV)
            qx = torch.empty(B, 3 * D, dtype=torch.bfloat16, device=DEV)

            def unfused():
                ops.layernorm_fwd(x, g, b, 1e-6, ln, None, None, M, D)
                ops.gemm_nt(ops.EPI_BF16, ln, W, qkv, M, 3 * D, D, bias=bq)
                ops.attn_fwd(qkv, out, None, B, N, H, 0.125)

            def fused():
                ops.layernorm_fwd(x, g, b, 1e-6, ln, None, None, M, D)
                ops.attn_block_fused(ln, W, bq, out, B, N, D, H, 0.125, qkv_extra=qx)
            t0 = timeit(unfused, reps=20)
            t1 = timeit(fused, reps=20)
            if os.environ.get("SRHIP_TUNING_BUILD") and B == 200:
                only = lambda: ops._call("srhip_attn_block_fused", ln.data_ptr(), W.data_ptr(), bq.data_ptr(), qx.data_ptr(), out.data_ptr(), None, B, N, D, H, 0.125, None)   # noqa: E731
                print("   kernel alone %7.1f us" % timeit(only, reps=20), flush=True)
                for dbg in (1, 2, 3, 4, 6, 7, 8, 16):
                    os.environ["SRHIP_AB_DEBUG"] = str(dbg)
                    print("   debug=%d (1 no attention, 2 no projection MFMAs, 4 no DMA, 8 no pass for query 256, 16 no v_exp): %7.1f us" % (dbg, timeit(only, reps=10)), flush=True)
                os.environ.pop("SRHIP_AB_DEBUG")
            fl = 2.0 * M * 3 * D * D + 4.0 * B * H * N * N * 64
            print("attn half N=%3d B=%3d: three launches %7.1f us %6.1f TF/s | fused %7.1f us %6.1f TF/s" % (N, B, t0, fl / t0 / 1e6, t1, fl / t1 / 1e6),
                  flush=True)


def mlp_proj():
    """proj GEMM (+ residual) + fused MLP as two launches vs srhip_mlp_fused_proj, launch sizes of a step."""
    D, Hd, N = 384, 1536, 257
    for B in (73, 127, 200):
        M = B * N
        x = torch.randn(M, D, device=DEV)
        ao = torch.randn(M, D, device=DEV).to(torch.bfloat16)
        Wp = (torch.randn(D, D, device=DEV) * 0.05).to(torch.bfloat16)
        bp = torch.randn(D, device=DEV) * 0.1
        g, b = torch.rand(D, device=DEV) + 0.5, torch.randn(D, device=DEV) * 0.1
        W1 = (torch.randn(Hd, D, device=DEV) * 0.05).to(torch.bfloat16)
        W2 = (torch.randn(D, Hd, device=DEV) * 0.02).to(torch.bfloat16)
        b1, b2 = torch.randn(Hd, device=DEV) * 0.1, torch.randn(D, device=DEV) * 0.1

        def two():
            ops.gemm_nt(ops.EPI_RESID_F32, ao, Wp, x, M, D, D, bias=bp)
            ops.mlp_fused(x, g, b, 1e-6, W1, b1, W2, b2, None, 0, M, D, Hd)
        t0 = timeit(two, reps=10)
        x.normal_()
        t1 = timeit(lambda: ops.mlp_fused_proj(x, ao, Wp, bp, None, g, b, 1e-6, W1, b1, W2, b2, None, 0, M, D, Hd), reps=10)
        t2 = timeit(lambda: ops.mlp_fused(x, g, b, 1e-6, W1, b1, W2, b2, None, 0, M, D, Hd), reps=10)
        x.normal_()
        fl = 4.0 * M * D * Hd + 2.0 * M * D * D
        print("proj+mlp B=%3d: two launches %7.1f us | fused %7.1f us %6.1f TF/s | (mlp alone %7.1f us)" % (B, t0, t1, fl / t1 / 1e6, t2), flush=True)


def attn_block_alone():
    """srhip_attn_block_fused alone at the launch sizes of a step: A/B of SRHIP_ATTN_SPREAD (run the process once per setting)."""
    D, H = 384, 6
    for N, Bs in ((257, (95, 105)), (197, (95, 105))):
        for B in Bs:
            M = B * N
            ln = torch.randn(M, D, device=DEV).to(torch.bfloat16)
            W = (torch.randn(3 * D, D, device=DEV) * 0.05).to(torch.bfloat16)
            bq = torch.randn(3 * D, device=DEV) * 0.1
            out = torch.empty(M, D, dtype=torch.bfloat16, device=DEV)
            qx = torch.zeros(B, 3 * D, dtype=torch.bfloat16, device=DEV)
            ts = [timeit(lambda: ops.attn_block_fused(ln, W, bq, out, B, N, D, H, 0.125, qkv_extra=qx if N == 257 else None), reps=20) for _ in range(3)]
            print("SRHIP_ATTN_SPREAD=%s  attn_block N=%d B=%3d: %s us" % (os.environ.get("SRHIP_ATTN_SPREAD", "0"), N, B, " / ".join("%.1f" % t for t in ts)), flush=True)


def mlp_rows16():
    """srhip_mlp_fused_proj (+ next norm1) alone, launch sizes of a step: A/B of SRHIP_MLP_SPREAD (run the process once per setting)."""
    D, Hd, N = 384, 1536, 257
    for B in (95, 105, 200):
        M = B * N
        x = torch.randn(M, D, device=DEV)
        ao = torch.randn(M, D, device=DEV).to(torch.bfloat16)
        Wp = (torch.randn(D, D, device=DEV) * 0.05).to(torch.bfloat16)
        bp = torch.randn(D, device=DEV) * 0.1
        g, b = torch.rand(D, device=DEV) + 0.5, torch.randn(D, device=DEV) * 0.1
        W1 = (torch.randn(Hd, D, device=DEV) * 0.05).to(torch.bfloat16)
        W2 = (torch.randn(D, Hd, device=DEV) * 0.02).to(torch.bfloat16)
        b1, b2 = torch.randn(Hd, device=DEV) * 0.1, torch.randn(D, device=DEV) * 0.1
        ln = torch.empty(M, D, dtype=torch.bfloat16, device=DEV)
        ts = []
        for _ in range(3):
            ts.append(timeit(lambda: ops.mlp_fused_proj(x, ao, Wp, bp, None, g, b, 1e-6, W1, b1, W2, b2, None, 0, M, D, Hd, ln_next=ln, next_gamma=g, next_beta=b), reps=20))
            x.normal_()
        print("SRHIP_MLP_SPREAD=%s  proj+mlp+ln B=%3d: %s us" % (os.environ.get("SRHIP_MLP_SPREAD", "0"), B, " / ".join("%.1f" % t for t in ts)), flush=True)


def gemm_qkv5():
    M, N, K = 51400, 1152, 384
    A = torch.randn(M, K, device=DEV).to(torch.bfloat16)
    Bm = (torch.randn(N, K, device=DEV) * 0.05).to(torch.bfloat16)
    bias = torch.randn(N, device=DEV)
    C = torch.zeros(M, N, dtype=torch.bfloat16, device=DEV)
    for _ in range(5):
        ops.gemm_nt(ops.EPI_BF16, A, Bm, C, M, N, K, bias=bias)
    torch.cuda.synchronize()


def attn200():
    B, N, H = 200, 257, 6
    qkv = torch.randn(B * N, 3 * H * 64, device=DEV).to(torch.bfloat16)
    out = torch.empty(B * N, H * 64, dtype=torch.bfloat16, device=DEV)
    for _ in range(5):
        ops.attn_fwd(qkv, out, None, B, N, H, 0.125)
    torch.cuda.synchronize()


def attn_bwd(N=257, H=6):
    """Attention backward of the gradient rows (16 images) and of a full launch."""
    for B in (16, 43, 200):
        D = H * 64
        qkv = torch.randn(B * N, 3 * D, device=DEV).to(torch.bfloat16)
        out = torch.empty(B * N, D, dtype=torch.bfloat16, device=DEV)
        lse = torch.empty(B, H, N, device=DEV)
        ops.attn_fwd(qkv, out, lse, B, N, H, 0.125)
        dout = torch.randn(B * N, D, device=DEV).to(torch.bfloat16)
        dqkv = torch.empty_like(qkv)
        delta = torch.empty(B, H, N, device=DEV)
        t = timeit(lambda: ops.attn_bwd(qkv, out, dout, lse, dqkv, delta, B, N, H, 0.125), reps=50)
        print("attn_bwd B=%4d  %8.1f us  %7.1f TF/s" % (B, t, 10.0 * N * N * 64 * B * H / t / 1e6), flush=True)


def attn_bert(N=512, H=12):
    """BERT-base attention (padding mask + probability dropout), forward and backward, 8 and 40 sequences."""
    for B in (8, 40):
        D = H * 64
        qkv = torch.randn(B * N, 3 * D, device=DEV).to(torch.bfloat16)
        out = torch.empty(B * N, D, dtype=torch.bfloat16, device=DEV)
        lse = torch.empty(B, H, N, device=DEV)
        kl = torch.full((B,), N, dtype=torch.int32, device=DEV)
        dr = ops.Drop(1234, 0, 0.1)
        t = timeit(lambda: ops.attn_masked_fwd(qkv, out, lse, kl, B, N, H, 0.125, dr), reps=30)
        print("attn_masked_fwd B=%3d  %8.1f us  %7.1f TF/s" % (B, t, 4.0 * N * N * 64 * B * H / t / 1e6), flush=True)
        dout = torch.randn(B * N, D, device=DEV).to(torch.bfloat16)
        dqkv = torch.empty_like(qkv)
        delta = torch.empty(B, H, N, device=DEV)
        t = timeit(lambda: ops.attn_masked_bwd(qkv, out, dout, lse, dqkv, delta, kl, B, N, H, 0.125, dr), reps=30)
        print("attn_masked_bwd B=%3d  %8.1f us  %7.1f TF/s" % (B, t, 10.0 * N * N * 64 * B * H / t / 1e6), flush=True)


if __name__ == "__main__":
    for a in sys.argv[1:]:
        globals()[a]()
