"""CPU-only checks: the C-ABI library loads and exports every symbol include/srhip.h declares (no compute calls
without a GPU), host-side logic (chunk table, layer-decay table, scheduler, pass plan, registry, DeferredScalar)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "srhip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(srhip_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from semireward_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        from semireward_amd.build import build
        build()
    h = ctypes.CDLL(_lib.LIB_PATH)
    names = header_functions()
    assert len(names) >= 30
    for n in names:
        assert hasattr(h, n), n
    # the Python binding table covers exactly the header
    assert sorted(_lib.SIGNATURES.keys()) == names
    lib = _lib.lib()
    assert lib.srhip_rewarder_param_count(384, 100) == 136962          # == reference Rewarder(100,128,384)
    assert lib.srhip_generator_param_count(384) == 139777              # == reference Generator(384)
    assert lib.srhip_rewarder_ws_floats(1, 8) > 0 and lib.srhip_rewarder_t_floats(384) == 128 * 384 + 2 * 256 * 128 + 64 * 128


def test_invalid_arguments_return_error_codes_without_gpu():
    """Argument validation happens before any launch, so it is testable on CPU."""
    from semireward_amd import _lib
    lib = _lib.lib()
    assert lib.srhip_gemm_nt(0, None, 64, None, 64, None, 64, 16, 16, 100, None, None, 0, None, None, 0, 1.0, 0.0, None) == -1
    assert lib.srhip_attn_fwd(None, None, None, 1, 1000, 6, 0.125, None) == -1
    assert lib.srhip_layernorm_fwd(None, None, None, 1e-6, None, None, None, 4, 100, None) == -1
    assert lib.srhip_rewarder_fwd(None, None, None, None, None, None, 2, 8, 384, 100, 1, None) == -1     # save needs G == 1
    with pytest.raises(RuntimeError):
        _lib.check(-1, "x")


def test_missing_library_fails_loudly(monkeypatch):
    from semireward_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libsrhip.so")
    with pytest.raises(RuntimeError, match="only compute path"):
        _lib.lib()


def test_optimizer_host_tables(golden):
    from oracle import vit_ref as V
    from semireward_amd.nets import vit
    from semireward_amd.optim import build_chunk_table, cosine_with_warmup, layer_decay_hparams
    g = golden("optim")
    cfg = vit.VitConfig(img_size=32, patch_size=2, embed_dim=384, depth=12, num_heads=6, num_classes=100, drop_path_rate=0.2)
    ns = vit.param_names_shapes(cfg)
    assert [(n, tuple(s)) for n, s in ns] == [(n, tuple(s)) for n, s in V.param_shapes(V.VitCfg(num_classes=100, **V.VIT_SMALL_P2_32))]
    assert sum(int(np.prod(s)) for _, s in ns) == 21436900
    hp = dict(zip([n for n, _ in ns], layer_decay_hparams(ns, 12, 5e-4, 5e-4, 0.5)))
    for n, lr, wd in zip(g["small/names"], g["small/lr"], g["small/wd"]):          # the reference's 28 param groups
        assert hp[str(n)][0] == pytest.approx(float(lr), rel=1e-12) and hp[str(n)][1] == float(wd), n
    for s, f in zip(g["sched/steps"], g["sched/factor"]):
        assert cosine_with_warmup(int(s), 204800, 5120) == pytest.approx(float(f), rel=1e-12, abs=1e-15)
    sizes = [int(np.prod(s)) for _, s in ns]
    t = build_chunk_table(sizes).numpy()
    assert t[:, 1].sum() == 21436900 and t[:, 1].max() <= 4096 and t[:, 1].min() > 0
    starts = np.cumsum([0] + sizes)
    for off, ln, tid, _ in t[::97]:
        assert starts[tid] <= off and off + ln <= starts[tid + 1]          # a chunk never straddles two tensors


def test_pass_plan_and_registry():
    from semireward_amd.algorithms import ALGORITHMS
    from inspect import signature
    from semireward_amd.algorithms.srflexmatch import SRFixMatch, SRFlexMatch, _Plan
    assert ALGORITHMS["srflexmatch"] is SRFlexMatch and ALGORITHMS["srfixmatch"] is SRFixMatch
    # train_step signatures ARE the batch schema (algorithmbase.py:287-306)
    assert list(signature(SRFlexMatch.train_step).parameters)[1:] == ["x_lb", "y_lb", "idx_ulb", "x_ulb_w", "x_ulb_s"]
    assert list(signature(SRFixMatch.train_step).parameters)[1:] == ["x_lb", "y_lb", "x_ulb_w", "x_ulb_s"]
    names = [a.name for a in SRFlexMatch.get_argument()]
    assert names == ["--hard_label", "--T", "--p_cutoff", "--thresh_warmup", "--start_timing", "--feature_dim", "--sr_lr",
                     "--N_k", "--sr_ema", "--sr_ema_m"]                      # srflexmatch.py:233-246
    from semireward_amd.algorithms.srfreematch import SRFreeMatch
    from semireward_amd.algorithms.srpseudolabel import SRPseudoLabel
    assert ALGORITHMS["srpseudolabel"] is SRPseudoLabel and ALGORITHMS["srfreematch"] is SRFreeMatch
    assert list(signature(SRFreeMatch.train_step).parameters)[1:] == ["x_lb", "y_lb", "x_ulb_w", "x_ulb_s"]
    from semireward_amd.algorithms.srsoftmatch import SRSoftMatch
    assert ALGORITHMS["srsoftmatch"] is SRSoftMatch                         # all five registry keys of the reference
    assert list(signature(SRSoftMatch.train_step).parameters)[1:] == ["x_lb", "y_lb", "x_ulb_w", "x_ulb_s"]
    assert [a.name for a in SRSoftMatch.get_argument()] == [
        "--hard_label", "--T", "--dist_align", "--dist_uniform", "--ema_p", "--n_sigma", "--per_class", "--start_timing", "--feature_dim",
        "--sr_lr", "--N_k", "--sr_ema", "--sr_ema_m"]                        # srsoftmatch.py:243-258
    pf = _Plan.cat_passes(8, 8, 8, "cpu", extra_pass0_strong=True)
    assert pf.grad_cols.tolist() == list(range(8)) + list(range(16, 24)) + [8 * 24 + j for j in range(16, 24)]
    assert list(signature(SRPseudoLabel.train_step).parameters)[1:] == ["x_lb", "y_lb", "x_ulb_w"]
    for nl, nu, K in [(8, 8, 8), (8, 8, 0), (4, 4, 20), (3, 5, 2)]:
        p = _Plan.cat_passes(nl, nu, K, "cpu")
        Bt = nl + 2 * nu
        allc = torch.cat([p.grad_cols, p.inf_cols]).sort().values
        assert torch.equal(allc, torch.arange((K + 1) * Bt))                 # every (pass, image) row exactly once
        assert p.grad_cols.numel() == nl + nu
        assert p.grad_cols[:nl].tolist() == list(range(nl))                  # pass 0, labelled rows (sup loss)
        assert p.grad_cols[nl:].tolist() == [K * Bt + j for j in range(nl + nu, Bt)]   # last pass, strong rows
        assert torch.equal(p.grad_img.long(), p.grad_cols % Bt) and torch.equal(p.inf_img.long(), p.inf_cols % Bt)


def test_deferred_scalar_and_sr_decay():
    from semireward_amd.core.algorithmbase import AlgorithmBase, DeferredScalar
    d = DeferredScalar(torch.tensor(1.5))
    assert float(d) == 1.5 and "%.2f" % d == "1.50" and d + 1 == 2.5 and d.item() == 1.5

    class A:
        num_train_iter = 204800
    for it, k in [(20001, 11), (20480, 11), (22755, 10), (25600, 9), (25601, 8), (204799, 8)]:
        A.it = it
        assert AlgorithmBase.sr_decay(A) == k


def test_classification_metrics_match_sklearn():
    """evaluate() host metrics (algorithmbase.py:419-423) against the sklearn calls the reference makes."""
    from sklearn.metrics import accuracy_score, balanced_accuracy_score, f1_score, precision_score, recall_score
    from semireward_amd.core.algorithmbase import AlgorithmBase
    rng = np.random.Generator(np.random.PCG64(5))
    for C, n in [(10, 200), (100, 300), (3, 7)]:
        yt, yp = rng.integers(0, C, n), rng.integers(0, C, n)
        yp[: n // 3] = yt[: n // 3]
        m = AlgorithmBase.classification_metrics(yt, yp)
        assert m["top-1-acc"] == pytest.approx(accuracy_score(yt, yp))
        assert m["balanced_acc"] == pytest.approx(balanced_accuracy_score(yt, yp))
        assert m["precision"] == pytest.approx(precision_score(yt, yp, average="macro", zero_division=0))
        assert m["recall"] == pytest.approx(recall_score(yt, yp, average="macro", zero_division=0))
        assert m["F1"] == pytest.approx(f1_score(yt, yp, average="macro", zero_division=0))


def test_plan_defers_only_unread_rows():
    """Row bookkeeping of one step: gradient rows, rows that are read (weak rows of every pass), rows nothing reads (deferred to the second
    stream), rows never computed (labelled rows of passes >= 1 under use_cat False); the deferred launch is sized to whole rounds of row tiles."""
    from semireward_amd.algorithms.srflexmatch import _Plan
    nl, nu, K = 8, 8, 8
    Bt = nl + 2 * nu
    p = _Plan.cat_passes(nl, nu, K, "cpu", defer_unread=True, rows_per_col=257)
    grad, read, rest = set(p.grad_cols.tolist()), set(p.inf_cols.tolist()), set(p.rest_cols.tolist())
    assert grad == set(range(nl)) | {K * Bt + j for j in range(nl + nu, Bt)}
    assert grad | read | rest == set(range((K + 1) * Bt)) and not (grad & read) and not (read & rest) and not (grad & rest)
    weak = {k * Bt + j for k in range(K + 1) for j in range(nl, nl + nu)}
    # the weak rows of every pass are read; of the 128 columns nothing reads a little under half of the 200 inference columns is deferred
    # (SR_DEFER_FRACTION, default 0.475 -> 95), the others ride in the read launch
    from semireward_amd.algorithms import srflexmatch as S
    assert weak <= read and len(rest) == int(S._DEFER_FRACTION * 200) and len(read - weak) == 128 - len(rest)
    assert -(-len(rest) * 257 // 128) <= 256
    q = _Plan.cat_passes(nl, nu, K, "cpu", lb_every_pass=False, defer_unread=True)
    done = set(q.grad_cols.tolist()) | set(q.inf_cols.tolist()) | set(q.rest_cols.tolist())
    assert done == set(range((K + 1) * Bt)) - {k * Bt + j for k in range(1, K + 1) for j in range(nl)}          # use_cat False: no x_lb after pass 0
    r = _Plan.cat_passes(nl, nu, K, "cpu")                          # no deferral: every inference column is "read"
    assert r.rest_cols.numel() == 0 and r.inf_cols.numel() == (K + 1) * Bt - 2 * nl


def test_host_side_random_inputs_match_oracle():
    """Host logic shared with the oracle by construction, not by import: dropout site keys, SpecAugment spans, optimizer groups of the BERT /
    Wav2Vec2 / HuBERT matchers, Pillow's Python-layer affine coefficients."""
    from oracle import augment_ref as A
    from oracle import bert_ref as BR
    from oracle import optim_ref as O
    from oracle import w2v2_ref as WR
    from semireward_amd import ops
    from semireward_amd.data import augment as GA
    from semireward_amd.nets import bert, hubert, wave2vec
    from semireward_amd.optim import layer_decay_hparams
    for seed, site in [(0, 0), ((71 << 32) + 5, 3), (2 ** 64 - 1, BR.SITE_HEAD)]:
        assert ops.site_key(seed, site) == BR.site_key(seed, site)
    for seed, B, T_ in [(5, 3, 19), (6, 4, 199)]:
        assert np.array_equal(wave2vec.spec_augment_mask(np.random.Generator(np.random.PCG64(seed)), B, T_, 0.05, 10, 2),
                              WR.spec_augment_mask(seed, B, T_, 0.05, 10, 2))
    # optimizer groups (no device memory is touched: only the name tables)
    cfg = BR.BertCfg(num_classes=4, **BR.BERT_TINY_TEST)
    names = bert.param_names_shapes(bert.BertConfig(num_classes=4, **BR.BERT_TINY_TEST))
    assert sorted(n for n, _ in names) == sorted(n for n, _ in BR.param_shapes(cfg))
    fake = type("M", (), {"names_shapes": names, "cfg": bert.BertConfig(num_classes=4, **BR.BERT_TINY_TEST)})()
    ids = bert.ClassificationBert.layer_ids(fake)
    hp = dict(zip([n for n, _ in names], layer_decay_hparams(names, 2, 5e-4, 5e-4, 0.65, no_weight_decay=(), layer_ids=ids,
                                                              frozen=bert.ClassificationBert.frozen_params)))
    want = O.bert_param_hparams(BR.param_shapes(cfg), cfg.layers, 5e-4, 5e-4, 0.65)
    assert all(hp[n] == pytest.approx(want[n], rel=1e-12) for n in want)
    wcfg = WR.W2vCfg(num_classes=4, **WR.W2V_TINY_TEST)
    wnames = wave2vec.param_names_shapes(wave2vec.W2vConfig(num_classes=4, **WR.W2V_TINY_TEST))
    fake = type("M", (), {"names_shapes": wnames, "cfg": wave2vec.W2vConfig(num_classes=4, **WR.W2V_TINY_TEST)})()
    for cls, hub in ((wave2vec.ClassificationWave2Vec, False), (hubert.ClassificationHubert, True)):
        ids = cls.layer_ids(fake)
        hp = dict(zip([n for n, _ in wnames], layer_decay_hparams(wnames, 2, 5e-4, 5e-4, 0.75, no_weight_decay=(), layer_ids=ids)))
        want = O.w2v_param_hparams(WR.param_shapes(wcfg), wcfg.layers, 5e-4, 5e-4, 0.75, hubert=hub)
        assert all(hp[n] == pytest.approx(want[n], rel=1e-12) for n in want), cls
    # Pillow's Python-layer math for the geometric ops: fixed-point coefficients / float64 walk origins
    assert GA.OPS == A.OPS and GA.RANGES == A.RANGES
    for S in (32, 96):
        for op, v in ((7, 17.3), (7, -29.9), (9, 0.21), (10, -0.13), (12, 0.27), (13, -0.3)):
            a = GA._affine_matrix(op, v, S)
            want = A.rotate_matrix(v, S, S) if A.OPS[op] == "Rotate" else {"ShearX": (1, v, 0, 0, 1, 0), "ShearY": (1, 0, 0, v, 1, 0),
                                                                          "TranslateX": (1, 0, v * S, 0, 1, 0), "TranslateY": (1, 0, 0, 0, 1, v * S)}[A.OPS[op]]
            assert list(a) == list(want)


def test_hot_kernels_keep_their_register_budget(tmp_path):
    """The fused MLP kernel runs two waves per SIMD at 240-odd VGPRs: one more live value and hipcc spills to scratch (measured: -5 % on the
    whole step).  Compile it with the resource remarks on and fail on any scratch use; the 128x128 GEMM must keep three waves per SIMD."""
    import re
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "semireward_amd", "csrc")

    def remarks(src):
        r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(root, "include"), "-I" + csrc, "-c",
                            "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage", "-o", str(tmp_path / "o.o"), os.path.join(csrc, src)],
                           capture_output=True, text=True, cwd=str(tmp_path))
        assert r.returncode == 0, r.stderr[-2000:]
        out, cur = {}, None
        for line in r.stderr.splitlines():
            m = re.search(r"Function Name: (\S+)", line)
            if m:
                cur = out.setdefault(m.group(1), {})
            m = re.search(r"remark:\s+([A-Za-z \[\]/]+): (\d+)", line)
            if m and cur is not None:
                cur[m.group(1).strip()] = int(m.group(2))
        return out
    mlp = {k: v for k, v in remarks("mlp_fused.hip").items() if "mlp_fused_kernelILi384ELi0ELi4" in k}
    assert mlp, "shipped instantiation not found"
    for k, v in mlp.items():
        assert v["ScratchSize [bytes/lane]"] == 0 and v["VGPRs Spill"] == 0 and v["Occupancy [waves/SIMD]"] >= 2, (k, v)
    # the 256 x 256 weight-gradient kernel: 256 registers, nothing spilled inside its K loop (the 16 bytes of scratch are three epilogue addresses)
    tn = {k: v for k, v in remarks("gemm_tn.hip").items() if "gemm_tn_pp_kernel" in k}
    assert len(tn) == 1
    for k, v in tn.items():
        assert v["ScratchSize [bytes/lane]"] <= 16 and v["VGPRs Spill"] <= 3 and v["Occupancy [waves/SIMD]"] >= 2, (k, v)
    gr = remarks("gemm.hip")
    gemm = {k: v for k, v in gr.items() if "gemm_nt_kernel" in k}
    assert len(gemm) == 6                 # five public epilogues + the residual that owes its LayerNorm (an instantiation of its own: as a run-time
    for k, v in gemm.items():             # switch inside the plain residual kernels it spilled both of them)
        assert v["ScratchSize [bytes/lane]"] == 0 and v["Occupancy [waves/SIMD]"] >= 3, (k, v)
    pp = {k: v for k, v in gr.items() if "gemm_pp_kernel" in k}
    assert len(pp) == 10
    for k, v in pp.items():
        assert v["ScratchSize [bytes/lane]"] == 0 and v["VGPRs Spill"] == 0, (k, v)


def test_torch_group_order_of_sgd_matches_the_reference(golden):
    """optim.torch_group_order(model, 1.0) -- the running index -> name map FusedSGD._load_torch_state relies on -- against the group order
    the reference's get_optimizer produced for the tiny WideResNet (ema.npz wrn/opt/*, written by oracle/gen_golden.py gen_ema)."""
    import types
    from oracle import wrn_ref as W
    from semireward_amd.optim import torch_group_order
    g = golden("ema")
    wcfg = W.WrnCfg(num_classes=10, **W.WRN_TINY_TEST)
    m = types.SimpleNamespace(names_shapes=[(n, tuple(s)) for n, s in W.param_shapes(wcfg)])
    order = torch_group_order(m, 1.0)
    assert [len(x) for x in order] == [int(v) for v in g["wrn/opt/group_sizes"]]
    assert [n for grp in order for n in grp] == [str(n) for n in g["wrn/opt/names_by_index"]]
    assert float(g["wrn/opt/group_wd"][0]) == 0.0 and float(g["wrn/opt/group_wd"][1]) > 0.0


def test_step_plan_partitions_every_column_for_every_deferred_share():
    """srflexmatch._Plan: whatever share of the inference rows a tuner candidate defers, the (gradient | read | deferred) launches partition the
    (pass, image) columns of the step, the rows the step reads are never deferred, and a deferred launch below the fused kernels' launch size is
    folded into the read launch."""
    from semireward_amd.algorithms.srflexmatch import _DeferTuner, _Plan
    from semireward_amd.nets import vit
    nl = nu = 8
    for K, N in ((8, 257), (8, 197), (11, 257), (8, 17)):
        Bt = nl + 2 * nu
        read = {k * Bt + j for k in range(K + 1) for j in range(nl, nl + nu)}
        sizes = set()
        for f in (None,) + _DeferTuner.CANDIDATES:
            p = _Plan.cat_passes(nl, nu, K, "cpu", defer_unread=True, rows_per_col=N, defer_fraction=f)
            g, i, r = (set(t.tolist()) for t in (p.grad_cols, p.inf_cols, p.rest_cols))
            assert g | i | r == set(range((K + 1) * Bt)) and not (g & i or g & r or i & r)
            assert read <= i and len(g) == nl + nu
            assert p.perm_cols.tolist() == p.grad_cols.tolist() + p.inf_cols.tolist() + p.rest_cols.tolist()
            assert len(r) == 0 or len(r) * N >= vit._FUSED_MLP_MIN_ROWS
            if f is not None and 0 < f < 1 and r:
                assert len(r) <= int(f * (len(i) + len(r))) + 1
            sizes.add(len(r))
        assert (len(sizes) > 2) == (N > 100)          # tiny backbones: nothing to tune (everything rides in the read launch)


def test_wide_resnet_refuses_unsupported_widths_at_construction():
    """Every block convolution runs as srhip_wrn_conv_bn (no generic chain behind it): a width the launch is not built for is an error with the
    reason when the model is BUILT (the reference's wrn_28_8 / widen_factor 4: Cin = 256), not SR_EINVAL in the middle of a forward."""
    from semireward_amd.nets import wrn
    with pytest.raises(NotImplementedError, match="Cin 256"):
        wrn.WideResNet(num_classes=10, depth=10, widen_factor=4, device="cpu")
    m = wrn.WideResNet(num_classes=10, depth=10, widen_factor=2, device="cpu")       # wrn_tiny_test / wrn_28_2 widths
    assert m.convs["block3.layer.0.conv2.weight"]["cin"] == 128


def test_bench_headline_keys_are_the_reference_yaml():
    """bench.py:NS (the headline configuration) against the namespace the reference's get_config() produces for
    config/SemiReward/usb_cv/flexmatch/flexmatch_cifar100_200_0.yaml (tests/golden/sr_configs.json); the three BASELINE configurations without a
    reference yaml load from configs/*.yaml through the same loader."""
    import glob
    import json
    import bench
    from semireward_amd import config
    ref = json.load(open(os.path.join(ROOT, "tests", "golden", "sr_configs.json")))["configs"]["usb_cv/flexmatch/flexmatch_cifar100_200_0.yaml"]
    for k, v in bench.NS.items():
        if k == "ulb_dest_len":           # not a yaml key: train.py sets it to len(unlabeled set) at run time (50 000 = CIFAR-100's training set)
            assert v == 50000
            continue
        assert ref[k] == v, (k, ref[k], v)
    assert ref["net"] == "vit_small_patch2_32" and ref["batch_size"] == 8 and ref["uratio"] == 1
    files = sorted(glob.glob(os.path.join(ROOT, "configs", "*.yaml")))
    assert len(files) == 3
    seen = {}
    for f in files:
        a = config.get_config(f)
        seen[a.algorithm] = a
        for k in ("start_timing", "feature_dim", "sr_lr", "N_k", "sr_ema", "sr_ema_m"):      # the SR keys every config/SemiReward yaml carries
            assert hasattr(a, k), (f, k)
        assert isinstance(a.lr, float) and a.amp is False
    assert seen["srpseudolabel"].net == "wrn_28_2" and seen["srpseudolabel"].feature_dim == 128 and seen["srpseudolabel"].num_classes == 100
    assert seen["srsoftmatch"].net == "bert_base_uncased" and seen["srsoftmatch"].num_classes == 2 and seen["srsoftmatch"].dataset == "aclImdb"
    assert seen["srfreematch"].net == "wave2vecv2_base" and seen["srfreematch"].num_classes == 10 and seen["srfreematch"].ema_p == 0.999


def test_defer_tuner_discards_a_window_with_an_evaluation_in_it(monkeypatch):
    """_DeferTuner times a step as the gap between consecutive step-start events; AlgorithmBase.evaluate / save_model call ``invalidate`` so that a
    gap which contains an evaluation is not taken for a step time -- the candidate restarts its window."""
    import torch
    from semireward_amd.algorithms import srflexmatch as SF

    class FakeEvent:
        clock, cost = 0.0, 1.0

        def __init__(self, enable_timing=True):
            self.t = None

        def record(self):
            FakeEvent.clock += FakeEvent.cost
            self.t = FakeEvent.clock

        def synchronize(self):
            pass

        def elapsed_time(self, other):
            return other.t - self.t

    monkeypatch.setattr(torch.cuda, "Event", FakeEvent)
    t = SF._DeferTuner([0.4, 0.5], refine=False)
    seq = []
    for i in range(40):
        if t.done:
            break
        f = t.fraction()
        seq.append(f)
        if i == 2:                       # an evaluation ran between two steps of the first candidate
            FakeEvent.clock += 100.0
            t.invalidate()
        FakeEvent.cost = 1.0 + abs(f - 0.5)
    assert t.best == 0.5 and t.report == {"0.400": 1.1, "0.500": 1.0}, (seq, t.report)       # the 100 ms gap is in neither median
    assert seq.count(0.4) == 3 + (SF._DeferTuner.WARM + SF._DeferTuner.TIMED)


def test_gemm_tile_dispatch_is_pinned_per_shape_family(monkeypatch):
    """srhip_gemm_nt_plan (host logic of srhip_gemm_nt, no launch): which tile kernel every product family of the legs goes to.  A dispatch rule
    written for the D = 768 legs once caught the ViT-S gradient-row products as well (99 workgroups of 128 x 128 instead of 390 of 64 x 64:
    12.8 -> 31 us per launch, found only in the round's rocprofv3 stats) -- this table makes such a change visible on the CPU."""
    from semireward_amd import ops
    for v in ("SRHIP_GEMM", "SRHIP_BIG_MIN_ROUNDS", "SRHIP_SMALL_MAX_GRID"):
        assert v not in os.environ, "tuning switch set: the table below is the default dispatch"
    want = {
        # ViT-S (D = 384), the 16 gradient images of a step (4112 rows): latency-bound launches -> 64 x 64 tiles wherever 128 x 128 would not fill the chip
        ("vit grad fc2 / proj fwd", ops.EPI_RESID_F32, 4112, 384, 1536): "small64", ("vit grad proj", ops.EPI_RESID_F32, 4112, 384, 384): "small64",
        ("vit grad fc1^T dX", ops.EPI_BF16, 4112, 384, 1536): "small64", ("vit grad qkv^T dX", ops.EPI_BF16, 4112, 384, 1152): "small64",
        ("vit grad fc1", ops.EPI_GELU_BF16, 4112, 1536, 384): "tile128", ("vit grad fc2^T dGELU", ops.EPI_DGELU_BF16, 4112, 1536, 384): "tile128",
        ("vit grad qkv", ops.EPI_BF16, 4112, 1152, 384): "tile128", ("vit 257th-token rows", ops.EPI_BF16, 105, 1152, 384): "small64",
        # ViT-S inference launches when the fused kernels are off: persistent kernel from 3 rounds of 256 x 256 tiles on
        ("vit qkv 200 images", ops.EPI_BF16, 51400, 1152, 384): "pp256", ("vit qkv 127 images", ops.EPI_BF16, 32639, 1152, 384): "tile128",
        ("vit proj 200 images", ops.EPI_RESID_F32, 51400, 384, 384): "tile128",
        # D = 768 (BERT / Wav2Vec2 / HuBERT): 256 x 256 kernel from 0.5 rounds of tiles on, 128 x 128 (not 64 x 64) below
        ("bert qkv", ops.EPI_BF16, 13952, 2304, 768): "pp256", ("bert proj", ops.EPI_RESID_F32, 13952, 768, 768): "pp256",
        ("bert fc1", ops.EPI_GELU_BF16, 13952, 3072, 768): "pp256", ("bert fc2", ops.EPI_RESID_F32, 13952, 768, 3072): "pp256",
        ("bert grad fc2", ops.EPI_RESID_F32, 4096, 768, 3072): "tile128", ("bert grad fc1", ops.EPI_GELU_BF16, 4096, 3072, 768): "pp256",
        ("w2v fc1", ops.EPI_GELU_BF16, 5373, 3072, 768): "pp256", ("w2v fc2", ops.EPI_RESID_F32, 5373, 768, 3072): "tile128",
        # fp32-accumulating products (weight gradients outside the grouped launch) never leave the 128 x 128 kernel (split-K lives there)
        ("dW small", ops.EPI_F32, 384, 1536, 4160): "tile128",
    }
    got = {k: ops.gemm_nt_plan(k[1], k[2], k[3], k[4], beta=1.0 if k[1] == ops.EPI_F32 else 0.0) for k in want}
    assert got == want, {k[0]: (got[k], want[k]) for k in want if got[k] != want[k]}
    with pytest.raises(RuntimeError):
        ops.gemm_nt_plan(ops.EPI_BF16, 128, 128, 100)          # K % 32 != 0, as srhip_gemm_nt refuses it
    # the run-time threshold a step with K > 0 sets (srhip_gemm_small_max_grid): the gradient-row products move to 128 x 128 tiles, the tiny
    # 257th-token launch keeps its 64 x 64 tiles, nothing else moves; back at the default the table above holds again
    ops.gemm_small_max_grid(ops.GEMM_SMALL_CONTENDED)
    try:
        moved = {k[0] for k in want if ops.gemm_nt_plan(k[1], k[2], k[3], k[4], beta=1.0 if k[1] == ops.EPI_F32 else 0.0) != want[k]}
        assert moved == {"vit grad fc2 / proj fwd", "vit grad proj", "vit grad fc1^T dX", "vit grad qkv^T dX"}, moved
        assert ops.gemm_nt_plan(ops.EPI_RESID_F32, 4112, 384, 1536) == "tile128"
    finally:
        ops.gemm_small_max_grid(ops.GEMM_SMALL_ALONE)
    assert {k: ops.gemm_nt_plan(k[1], k[2], k[3], k[4], beta=1.0 if k[1] == ops.EPI_F32 else 0.0) for k in want} == want


def test_weight_gradient_table_of_the_persistent_kernel_is_balanced_on_the_host():
    """ops.tn_pp_plan / tn_pp_efficiency (host logic of srhip_gemm_tn_grouped_pp_f32, no launch): the kernel's walk is static -- workgroup w of
    256 takes tiles w, w + 256, ... -- so the table slices the LAST problems along the token axis when a few tiles past a full round would leave
    most workgroups idle for one whole tile; slices are multiples of the 64-token K-tile; tables whose tiles 256 x 256 covers badly are left to the
    128 x 128 kernel by the encoders (efficiency < 0.85)."""
    from semireward_amd import ops
    D, I, K = 768, 3072, 8192
    layer = [(0, 0, 0, 0, D, I, K), (0, 0, 0, 0, I, D, K), (0, 0, 0, 0, D, D, K), (0, 0, 0, 0, 3 * D, D, K)]
    bert = layer * 12                                          # 12 x 108 = 1296 tiles = 5 rounds + 16
    assert ops.tn_pp_efficiency(bert) == 1.0
    plan = ops.tn_pp_plan(bert)
    sliced = [(pr[4:], sl) for pr, sl in plan if sl]
    assert sliced == [((3 * D, D, K), 1024)], sliced           # the last problem (27 tiles >= the 16 of the tail) in 8 slices of 1024 tokens
    assert [pr for pr, _ in plan] == [tuple(p) for p in bert]  # order kept
    # worst workgroup: 5 whole tiles + one slice instead of 6 whole tiles
    tiles = []
    for (A, B, C, db, M, N, Kp), sl in plan:
        n = ((M + 255) // 256) * ((N + 255) // 256)
        tiles += [min(sl, Kp - k0) for k0 in range(0, Kp, sl) for _ in range(n)] if sl else [Kp] * n
    load = [sum(tiles[w::256]) for w in range(256)]
    assert max(load) == 5 * K + 1024 and sum(tiles) == 1296 * K
    # one layer alone (108 tiles < 256 workgroups), a whole number of rounds, a long tail: table order, nothing sliced
    assert all(sl == 0 for _, sl in ops.tn_pp_plan(layer))
    assert all(sl == 0 for _, sl in ops.tn_pp_plan([(0, 0, 0, 0, 4096, 4096, K)]))              # 256 tiles
    assert all(sl == 0 for _, sl in ops.tn_pp_plan([(0, 0, 0, 0, 4096, 4096, K), (0, 0, 0, 0, 4096, 3072, K)]))   # 256 + 192: tail > 0.7 round
    # short token axes are not sliced below 512 tokens; slices are multiples of 64
    for pr, sl in ops.tn_pp_plan([(0, 0, 0, 0, D, I, 3184)] * 8 + [(0, 0, 0, 0, D, D, 3184)]):
        assert sl == 0 or (sl % 64 == 0 and sl >= 512)
    assert all(sl == 0 for _, sl in ops.tn_pp_plan([(0, 0, 0, 0, D, I, 600)] * 8 + [(0, 0, 0, 0, D, D, 600)]))
    # ViT-S: 384 = 256 + 128 in both directions -- 71 % of the tile area is inside the problems: stays on the 128 x 128 kernel
    vit = [(0, 0, 0, 0, 384, 1536, 4112), (0, 0, 0, 0, 1536, 384, 4112), (0, 0, 0, 0, 384, 384, 4112), (0, 0, 0, 0, 1152, 384, 4112)]
    assert 0.70 < ops.tn_pp_efficiency(vit) < 0.72

