"""Edge cases of the score filter / masked loss kernels against the CPU oracle (through the C ABI; needs a MI355X): ties, values exactly
on a threshold, empty and full masks, single-row and ragged batches, a saturated selected_label table, extreme logits."""
import os
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import hooks_ref as H          # noqa: E402
from oracle import semireward_ref as S     # noqa: E402
from semireward_amd import ops             # noqa: E402

DEV = "cuda:0"
F32 = np.float32


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def _flex_engine(C, U, warm):
    sel = torch.full((U,), -1, dtype=torch.int64, device=DEV)
    hist = torch.zeros(C + 1, dtype=torch.int32, device=DEV)
    acc = torch.zeros(C, device=DEV)
    ops.flexmatch_rebuild_hist(sel, hist, U, C)
    return sel, hist, acc


def _flex_step(state, eng, probs, idx, C, U, warm, p_cutoff=0.95):
    sel, hist, acc = eng
    Bu = probs.shape[0]
    mp, mi = torch.empty(Bu, device=DEV), torch.empty(Bu, dtype=torch.int64, device=DEV)
    mask = torch.empty(Bu, device=DEV)
    ops.row_max(_dev(probs), True, None, mp, mi, Bu, C)
    ops.flexmatch_mask(mp, mi, _dev(idx), p_cutoff, sel, hist, acc, mask, Bu, C, U, warm)
    ref = state.masking(probs, idx, p_cutoff)
    assert np.array_equal(mi.cpu().numpy(), probs.argmax(-1))
    assert np.array_equal(mask.cpu().numpy(), ref)
    assert np.array_equal(acc.cpu().numpy().view(np.uint32), state.classwise_acc.view(np.uint32))
    assert np.array_equal(sel.cpu().numpy(), state.selected_label)
    return mask.cpu().numpy()


def _probs_with_max(rows, C, seed=0):
    """Rows of a probability table whose maximum is EXACTLY the requested fp32 value at the requested class (the rest spread evenly)."""
    out = np.zeros((len(rows), C), dtype=F32)
    for r, (cls, mx) in enumerate(rows):
        mx = F32(mx)
        rest = (F32(1.0) - mx) / F32(C - 1)
        out[r, :] = min(rest, np.nextafter(mx, F32(0)))
        out[r, cls] = mx
    return out


@pytest.mark.parametrize("warm", [True, False])
def test_flexmatch_threshold_boundaries_and_ties(warm):
    C, U = 10, 128
    cut = F32(0.95)
    below, above = np.nextafter(cut, F32(0)), np.nextafter(cut, F32(1))
    state, eng = H.FlexMatchState(U, C, warm), _flex_engine(C, U, warm)
    # step 1: values exactly on / one ulp either side of p_cutoff; a two-way tie of the maximum (first index wins)
    probs = _probs_with_max([(3, cut), (3, below), (4, above), (0, 0.5), (9, 1.0), (2, 0.96)], C)
    probs[3, 7] = probs[3, 0]                          # tie between class 0 and class 7
    _flex_step(state, eng, probs, np.array([5, 6, 7, 8, 9, 10], dtype=np.int64), C, U, warm)
    # step 2: classwise_acc is now non-zero -> per-class thresholds 0.95 * acc / (2 - acc); rows sitting exactly on them
    acc = state.classwise_acc.copy()
    thr = cut * (acc / (F32(2.0) - acc))
    rows = [(c, max(thr[c], F32(0.2))) for c in (3, 4, 9, 2)] + [(c, np.nextafter(max(thr[c], F32(0.2)), F32(0))) for c in (3, 4, 9, 2)]
    _flex_step(state, eng, _probs_with_max(rows, C), np.arange(20, 28, dtype=np.int64), C, U, warm)
    # step 3: a single-row batch, then a ragged one (65 rows: one past a wavefront), re-selecting already selected indices
    _flex_step(state, eng, _probs_with_max([(1, 0.99)], C), np.array([5], dtype=np.int64), C, U, warm)
    g = np.random.Generator(np.random.PCG64(7))
    p = g.dirichlet(np.full(C, 0.05), size=65).astype(F32)
    _flex_step(state, eng, p, g.permutation(U)[:65].astype(np.int64), C, U, warm)      # (unique within a batch, as the sampler draws them)


def test_flexmatch_saturated_table_stops_updating():
    """utils.py:26: once one label (or the unused bucket) covers the whole table the accuracies are frozen."""
    C, U = 4, 8
    state, eng = H.FlexMatchState(U, C, True), _flex_engine(C, U, True)
    # nothing selected in the first step: the unused bucket still holds all U entries -> no update, mask from acc = 0 (threshold 0)
    m = _flex_step(state, eng, _probs_with_max([(0, 0.5)] * 4, C), np.arange(4, dtype=np.int64), C, U, True)
    assert m.all() and not state.classwise_acc.any()
    # every entry selected with the same class -> count == U -> frozen again
    _flex_step(state, eng, _probs_with_max([(2, 0.99)] * 8, C), np.arange(8, dtype=np.int64), C, U, True)
    frozen = state.classwise_acc.copy()
    _flex_step(state, eng, _probs_with_max([(2, 0.99)] * 8, C), np.arange(8, dtype=np.int64), C, U, True)
    assert np.array_equal(frozen, state.classwise_acc)
    # one entry moves to another class: updates resume
    _flex_step(state, eng, _probs_with_max([(1, 0.99)], C), np.array([3], dtype=np.int64), C, U, True)
    assert not np.array_equal(frozen, state.classwise_acc)


def test_fixed_mask_boundary():
    cut = F32(0.95)
    mp = np.array([cut, np.nextafter(cut, F32(0)), np.nextafter(cut, F32(1)), 0.0, 1.0], dtype=F32)
    out = torch.empty(5, device=DEV)
    ops.fixed_mask(_dev(mp), float(cut), out, 5)
    assert np.array_equal(out.cpu().numpy(), (mp >= cut).astype(F32))
    assert out.cpu().tolist() == [1.0, 0.0, 1.0, 0.0, 1.0]


@pytest.mark.parametrize("groups,B", [(1, 1), (1, 8), (8, 8), (3, 65)])
def test_reward_mask2_ties_and_single_rows(groups, B):
    """mask2 = reward >= mean (srflexmatch.py:100-101): all-equal rewards sit ON the mean -> every row passes; per group of B rows."""
    g = np.random.Generator(np.random.PCG64(groups * 100 + B))
    r = g.uniform(0, 1, size=(groups, B)).astype(F32)
    r[0, :] = F32(0.625)                                # a constant group: mean == every element exactly
    if groups > 1:
        r[1, :] = np.where(np.arange(B) % 2 == 0, F32(0.25), F32(0.75))    # half below / half above an exactly representable mean
    m2, mean = torch.empty(groups * B, device=DEV), torch.empty(groups, device=DEV)
    ops.reward_mask2(_dev(r.reshape(-1)), m2, mean, groups, B)
    got = m2.cpu().numpy().reshape(groups, B)
    assert got[0].all()
    if groups > 1 and B % 2 == 0:
        assert np.array_equal(got[1], (np.arange(B) % 2 == 1).astype(F32))
    for k in range(groups):
        ref = S.reward_mask2(torch.from_numpy(r[k])).numpy()
        away = np.abs(r[k] - r[k].mean(dtype=np.float64)) > 1e-6          # summation order only matters within round-off of the mean
        assert np.array_equal(got[k][away], ref[away]), k
    # mean_in overrides the local mean (data-parallel global threshold)
    ops.reward_mask2(_dev(r.reshape(-1)), m2, None, groups, B, mean_in=_dev(np.full(groups, 2.0, dtype=F32)))
    assert not m2.cpu().numpy().any()


@pytest.mark.parametrize("B,C", [(1, 2), (8, 100), (65, 10)])
def test_masked_ce_empty_full_and_extreme(B, C):
    g = np.random.Generator(np.random.PCG64(B * 1000 + C))
    lg = (4 * g.standard_normal((B, C))).astype(F32)
    lg[0, :] = 0.0
    lg[0, 0], lg[0, C - 1] = 80.0, -80.0               # saturated softmax: log-sum-exp must not overflow
    y = g.integers(0, C, size=B).astype(np.int64)
    y[0] = C - 1                                        # the target is the -80 logit: loss 160, gradient -1 / B there
    tl, ty = torch.from_numpy(lg), torch.from_numpy(y)
    loss, dl = torch.empty(1, device=DEV), torch.empty(B, C, device=DEV)
    zero, one = torch.zeros(B, device=DEV), torch.ones(B, device=DEV)
    # empty mask: exactly zero loss and gradient (consistency.py:38-45 multiplies before the mean)
    ops.masked_ce(_dev(lg), _dev(y), zero, one, 1.0, loss, dl, B, C)
    assert float(loss) == 0.0 and not dl.cpu().numpy().any()
    ops.masked_ce(_dev(lg), _dev(y), one, zero, 1.0, loss, dl, B, C)
    assert float(loss) == 0.0 and not dl.cpu().numpy().any()
    # full masks == no masks == the oracle's mean cross entropy, finite everywhere
    ref = float(H.consistency_loss(tl, ty, torch.ones(B), torch.ones(B)))
    ops.masked_ce(_dev(lg), _dev(y), one, one, 1.0, loss, dl, B, C)
    full = float(loss)
    d_full = dl.cpu().numpy().copy()
    ops.masked_ce(_dev(lg), _dev(y), None, None, 1.0, loss, dl, B, C)
    assert float(loss) == full and np.array_equal(dl.cpu().numpy(), d_full)
    assert np.isfinite(d_full).all() and abs(full - ref) <= 2e-6 * max(1.0, abs(ref))
    tg = tl.clone().requires_grad_(True)
    H.consistency_loss(tg, ty, torch.ones(B), torch.ones(B)).backward()
    np.testing.assert_allclose(d_full, tg.grad.numpy(), rtol=2e-5, atol=1e-8)
    # a one-row mask: only that row carries gradient, scaled by grad_scale / B
    one_row = torch.zeros(B, device=DEV)
    one_row[B - 1] = 1.0
    ops.masked_ce(_dev(lg), _dev(y), one_row, None, 0.5, loss, dl, B, C)
    d = dl.cpu().numpy()
    assert not d[:B - 1].any()
    m = torch.zeros(B)
    m[B - 1] = 1.0
    tg = tl.clone().requires_grad_(True)
    H.consistency_loss(tg, ty, m).backward()
    np.testing.assert_allclose(d[B - 1], 0.5 * tg.grad.numpy()[B - 1], rtol=2e-5, atol=1e-8)


def test_flexmatch_passes_in_one_launch_equal_sequential_calls():
    """srhip_flexmatch_mask_passes == the same passes as separate srhip_flexmatch_mask launches (masks and state bit-identical), and both
    equal the oracle; the passes share idx_ulb and later passes re-select indices the earlier ones selected."""
    C, U, nu, P = 10, 64, 8, 9
    g = np.random.Generator(np.random.PCG64(21))
    idx = g.permutation(U)[:nu].astype(np.int64)
    probs = [g.dirichlet(np.full(C, 0.03), size=nu).astype(F32) for _ in range(P)]
    for warm in (True, False):
        state = H.FlexMatchState(U, C, warm)
        want = [state.masking(pr, idx, 0.95) for pr in probs]
        mp = _dev(np.concatenate([pr.max(-1) for pr in probs]))
        mi = _dev(np.concatenate([pr.argmax(-1) for pr in probs]).astype(np.int64))
        sel, hist, acc = _flex_engine(C, U, warm)
        mask = torch.empty(P * nu, device=DEV)
        ops.flexmatch_mask_passes(mp, mi, _dev(idx), 0.95, sel, hist, acc, mask, P, nu, C, U, warm)
        assert np.array_equal(mask.cpu().numpy().reshape(P, nu), np.stack(want))
        assert np.array_equal(sel.cpu().numpy(), state.selected_label)
        assert np.array_equal(acc.cpu().numpy().view(np.uint32), state.classwise_acc.view(np.uint32))
        sel2, hist2, acc2 = _flex_engine(C, U, warm)
        m2 = torch.empty(P * nu, device=DEV)
        for k in range(P):
            ops.flexmatch_mask(mp[k * nu:(k + 1) * nu], mi[k * nu:(k + 1) * nu], _dev(idx), 0.95, sel2, hist2, acc2, m2[k * nu:(k + 1) * nu], nu, C, U, warm)
        assert torch.equal(mask, m2) and torch.equal(sel, sel2) and torch.equal(acc, acc2) and torch.equal(hist, hist2)


@pytest.mark.parametrize("general", [False, True])
def test_flexmatch_duplicate_indices_and_multi_pass_state(general, monkeypatch):
    """idx_ulb with the SAME index several times in one batch (the sampler's concatenated permutations can meet): selected_label takes the
    LAST selected copy (the reference's CPU index_put) and the incremental histogram stays equal to the reference's recount -- over several
    calls and inside one multi-pass launch; both kernel variants (state in LDS / general path with atomicExch)."""
    if general:
        monkeypatch.setenv("SRHIP_FLEXMATCH_GENERAL", "1")
    C, U, Bu, P = 7, 40, 24, 5
    rng = np.random.Generator(np.random.PCG64(31))
    state, (sel, hist, acc) = H.FlexMatchState(U, C, True), _flex_engine(C, U, True)
    for call in range(4):
        idx = rng.integers(0, 12, size=Bu).astype(np.int64)              # 24 draws from 12 indices: many duplicates
        probs = np.stack([_probs_with_max([(int(rng.integers(C)), float(rng.choice([0.5, 0.94, 0.96, 0.99]))) for _ in range(Bu)], C)
                          for _ in range(P)])
        mp, mi = torch.empty(P * Bu, device=DEV), torch.empty(P * Bu, dtype=torch.int64, device=DEV)
        ops.row_max(_dev(probs.reshape(P * Bu, C)), True, None, mp, mi, P * Bu, C)
        mask = torch.empty(P * Bu, device=DEV)
        ops.flexmatch_mask_passes(mp, mi, _dev(idx), 0.95, sel, hist, acc, mask, P, Bu, C, U, True)
        if not general:            # last-wins needs the ordered LDS kernel; the atomic path keeps hist == recount for any winner
            want = np.stack([state.masking(probs[p], idx, 0.95) for p in range(P)])
            assert np.array_equal(mask.cpu().numpy().reshape(P, Bu), want), call
            assert np.array_equal(sel.cpu().numpy(), state.selected_label)
            assert np.array_equal(acc.cpu().numpy().view(np.uint32), state.classwise_acc.view(np.uint32))
        s = sel.cpu().numpy()
        recount = np.bincount(np.where(s < 0, C, s), minlength=C + 1)
        assert np.array_equal(hist.cpu().numpy(), recount), (call, hist.cpu().numpy(), recount)
    ops.check_label_errors()


def test_out_of_range_indices_and_labels_raise_like_the_reference():
    """nn.Embedding / F.one_hot / index_put raise in the reference for an index outside its table; the kernels stay memory safe, set a device
    flag and ops.check_label_errors() raises IndexError at the next host check."""
    from semireward_amd.algorithms.semireward import Generator, Rewarder, cosine_target
    C, U, Bu = 5, 16, 4
    ops.check_label_errors()
    # FlexMatch: idx_ulb beyond the selected_label table
    sel, hist, acc = _flex_engine(C, U, True)
    probs = _probs_with_max([(1, 0.99)] * Bu, C)
    mp, mi = torch.empty(Bu, device=DEV), torch.empty(Bu, dtype=torch.int64, device=DEV)
    ops.row_max(_dev(probs), True, None, mp, mi, Bu, C)
    mask = torch.empty(Bu, device=DEV)
    guard = torch.full((U + 64,), -1, dtype=torch.int64, device=DEV)
    ops.flexmatch_mask(mp, mi, _dev(np.array([0, U, 3, -2], dtype=np.int64)), 0.95, guard[:U], hist, acc, mask, Bu, C, U, True)
    assert int((guard[U:] != -1).sum()) == 0 and int((guard[:U] == 1).sum()) == 2      # nothing written past the table; the 2 valid rows landed
    with pytest.raises(IndexError, match="idx_ulb"):
        ops.check_label_errors()
    ops.check_label_errors()                                                           # reset by the failed check
    # Rewarder: label >= label_dim
    rew = Rewarder(100, 128, 32, device=DEV)
    feats = torch.randn(Bu, 32, device=DEV)
    r = rew.score(feats, torch.tensor([0, 99, 100, 7], dtype=torch.int64, device=DEV))
    assert bool(torch.isfinite(r).all())
    with pytest.raises(IndexError, match="Embedding"):
        ops.check_label_errors()
    # Generator: huge features -> output far above any class (still finite): label stored, one_hot of the SR target refuses it
    gen = Generator(32, device=DEV)
    gen.flat.abs_(); gen.prepare()
    _, lab = gen.forward_with_labels(torch.full((Bu, 32), 10.0, device=DEV))
    assert int(lab.min()) > 1000
    cosine_target(lab, torch.zeros(Bu, dtype=torch.int64, device=DEV), C)
    with pytest.raises(IndexError, match="one_hot"):
        ops.check_label_errors()
    # ... and a NaN feature: .long() of NaN is garbage in torch; here label -1 + flag
    _, lab = gen.forward_with_labels(torch.full((Bu, 32), float("nan"), device=DEV))
    assert int(lab.max()) == -1
    with pytest.raises(IndexError, match="NaN"):
        ops.check_label_errors()
    ops.check_label_errors()


def test_flexmatch_without_warmup_before_any_selection_keeps_the_state():
    """thresh_warmup False and not one row over the cut-off yet: every entry of selected_label is -1, so max(Counter) == ulb_dest_len and the
    reference's update() does nothing (srflexmatch/utils.py:27) -- the kernels must leave classwise_acc at 0 (no 0 / 0 = NaN from the
    "without -1" denominator of :31-35), in the LDS kernel and the general one, and the oracle agrees."""
    C, U, Bu = 5, 16, 4
    ops.check_label_errors()
    for general in (False, True):
        if general:
            os.environ["SRHIP_FLEXMATCH_GENERAL"] = "1"
        try:
            sel, hist, acc = _flex_engine(C, U, False)
            st = H.FlexMatchState(U, C, False)
            probs = _probs_with_max([(1, 0.5)] * Bu, C)                    # nothing reaches p_cutoff
            mp, mi = torch.empty(Bu, device=DEV), torch.empty(Bu, dtype=torch.int64, device=DEV)
            ops.row_max(_dev(probs), True, None, mp, mi, Bu, C)
            mask = torch.empty(Bu, device=DEV)
            idx = np.arange(Bu, dtype=np.int64)
            for _ in range(2):
                ops.flexmatch_mask(mp, mi, _dev(idx), 0.95, sel, hist, acc, mask, Bu, C, U, False)
                want = st.masking(probs, idx, 0.95)
                assert np.array_equal(mask.cpu().numpy(), want)
                assert bool(torch.isfinite(acc).all()) and np.array_equal(acc.cpu().numpy(), st.classwise_acc)
            assert float(acc.abs().max()) == 0.0 and int(hist[C]) == U
            ops.check_label_errors()
        finally:
            os.environ.pop("SRHIP_FLEXMATCH_GENERAL", None)


def test_flexmatch_state_larger_than_lds_takes_the_general_kernel():
    """C = 8192 classes x B = 4096 rows would ask the state-in-LDS kernel for 163 844 B (> 160 KiB): the call must fall back to the general
    kernel instead of failing the launch, with the same masks as the numpy oracle."""
    from oracle import hooks_ref as H
    C, U, Bu = 8192, 20000, 4096
    rng = np.random.Generator(np.random.PCG64(5))
    sel, hist, acc = _flex_engine(C, U, True)
    mpn = rng.random(Bu).astype(np.float32)
    mpn[::3] = 0.99
    min_ = rng.integers(0, C, size=Bu, dtype=np.int64)
    idx = rng.permutation(U)[:Bu].astype(np.int64)
    mask = torch.empty(Bu, device=DEV)
    ops.flexmatch_mask(_dev(mpn), _dev(min_), _dev(idx), 0.95, sel, hist, acc, mask, Bu, C, U, True)
    torch.cuda.synchronize()
    st = H.FlexMatchState(U, C, True)
    probs = np.zeros((Bu, C), np.float32)
    probs[np.arange(Bu), min_] = mpn
    want = st.masking(probs, idx, 0.95)
    assert np.array_equal(mask.cpu().numpy(), want)
    assert np.array_equal(sel.cpu().numpy(), st.selected_label)
    assert np.array_equal(acc.cpu().numpy().view(np.uint32), st.classwise_acc.view(np.uint32))
    ops.check_label_errors()
