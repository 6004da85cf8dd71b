"""SRFlexMatch / SRFixMatch (FlexMatch|FixMatch + SemiReward) on the HIP engine -- registered under the reference's keys.

Reference: semilearn/algorithms/srflexmatch/srflexmatch.py (train_step :107-217, data_generator :72-104).
Semantics reproduced exactly: control-flow thresholds on ``it``, K = sr_decay() extra
backbone passes whose masking calls mutate the FlexMatch state, only the LAST pass's loss survives, per-rank reward
mean, inert generator, no-op max_reward filter, stage-1 / stage-2 rewarder updates.

What is restructured for MI355X -- results are unchanged because ViT rows are independent (no BatchNorm):
  * the 1+K passes over the SAME 24 images are ONE batched forward of (1+K)*Bt rows, each (pass, image) row with
    its own DropPath draw, instead of 1+K sequential launches trains at Bt=24;
  * only rows whose logits enter the loss carry a backward graph: x_lb of pass 0 (sup loss) and x_ulb_s of the last
    pass (unsup loss).  The reference back-propagates two full 24-row graphs whose other rows receive exactly zero
    gradient; pruning them is an algebraic identity, not an approximation;
  * softmax/argmax of all passes is one launch, the K rewarder scorings are one grouped launch, the FlexMatch
    state updates stay sequential (they are order dependent) but never leave the device.
"""
import os

import torch

from .. import ops
from ..nets import vit as _vit
from ..core.algorithmbase import AlgorithmBase, DeferredScalar
from ..core.registry import ALGORITHMS
from .hooks import FixedThresholdingHook, FlexMatchThresholdingHook, PseudoLabelingHook
from .semireward import FlatAdam, Generator, Rewarder, cosine_target, label_dim
from .utils import SSL_Argument, str2bool

_PHASES = os.environ.get("SR_PHASES", "0") != "0"
_SCORE_ON_SIDE = True         # the max_reward scoring launch runs on the side stream under the backward (DESIGN: -55 us of serial tail)
# Share of the inference images that run on the second stream (see _Plan).  SR_DEFER_FRACTION=<f> pins it; otherwise it is TUNED per
# (batch, K, backbone) in the first steps of a regime (_DeferTuner).
_DEFER_SEED = 0.475           # measured optimum of ViT-S/2 at 8 / 8 / 8, K = 8 on one MI355X; the tuner's middle candidate
_DEFER_FIXED = float(os.environ["SR_DEFER_FRACTION"]) if "SR_DEFER_FRACTION" in os.environ else None
_DEFER_AUTOTUNE = _DEFER_FIXED is None
_DEFER_FRACTION = _DEFER_FIXED if _DEFER_FIXED is not None else _DEFER_SEED


class _DeferTuner:
    """Start-up autotune of the deferred share.  Where the optimum lies depends on how the two launch trains of a step pack the chip: the
    workgroups of the row-streaming kernels own a CU for ~100 us, so while a deferred launch fills the chip every small launch of the critical
    chain (masks, losses, the backward of the gradient rows) waits for one of them to retire; too few deferred rows and the read launch alone
    delays that chain.  The balance moves with the token count, the batch, K and the CU count, and it is sharp (95 deferred images
    1542 img/s, 98 -> 1469), so it is MEASURED: every candidate share runs WARM + TIMED real training steps (nothing is thrown away; the split
    does not change a single result -- rows are independent), step time = HIP events at consecutive step starts (one host synchronisation per
    candidate); after the coarse candidates the two neighbours of the winner are measured as well, and the median-fastest share is kept for the
    rest of the run.  Cached per plan key."""
    CANDIDATES = (0.35, 0.42, _DEFER_SEED, 0.53, 0.58, 1.0)      # 1.0 = every row nothing reads (clipped to what may be deferred)
    REFINE = 0.025
    WARM, TIMED = 1, 3

    def __init__(self, fractions, refine=True, agree=None):
        self.queue = list(fractions)        # candidates still to run
        self.refine = refine
        # data parallel: every rank runs its own tuner over the same steps, and which candidates are queued next (hence how many tuning steps
        # there are, hence how many gradient all-reduces) hangs on the measured times -- agree(ms) = the maximum over the ranks, so that all
        # ranks see the same table, take the same decisions and keep the same share (the step of the slowest rank is the job's step anyway)
        self.agree = agree
        self.results = {}                   # share -> median ms per step
        self.cur, self.marks = None, []
        self.best = self.report = None

    @property
    def done(self):
        return self.best is not None

    def invalidate(self):
        """Something other than training steps ran on the device (evaluation, a checkpoint): the gap between the last two step starts is not
        a step time.  The current candidate starts its WARM + TIMED steps over (every rank calls this at the same iteration)."""
        self.marks = []

    def fraction(self):
        """Share for the step that starts now (records the step-start event while tuning)."""
        if self.best is not None:
            return self.best
        per = self.WARM + self.TIMED
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        self.marks.append(e)
        if self.cur is not None and len(self.marks) == per + 1:          # the current candidate's steps are marks[0] .. marks[per]
            e.synchronize()
            ms = sorted(self.marks[j].elapsed_time(self.marks[j + 1]) for j in range(self.WARM, per))
            self.results[self.cur] = self.agree(ms[len(ms) // 2]) if self.agree is not None else ms[len(ms) // 2]
            self.marks, self.cur = self.marks[-1:], None                 # this event also starts the next candidate's first step
        if self.cur is None:
            if not self.queue and self.refine and self.results:
                b = min(self.results, key=self.results.get)
                self.queue = [f for f in (round(b - self.REFINE, 4), round(b + self.REFINE, 4)) if 0.05 < f < 0.95 and b < 1.0 and f not in self.results]
                self.refine = False
            if not self.queue:
                self.best = min(self.results, key=self.results.get)
                self.report = {"%.3f" % f: round(m, 4) for f, m in sorted(self.results.items())}
                self.marks = []
                return self.best
            self.cur = self.queue.pop(0)
        return self.cur


class _Plan:
    """Row bookkeeping of one step: every column is one (pass, image) row of the batched forward; ``grad_cols`` are the
    rows whose logits enter the loss (they are run with activations kept), all other rows run in inference mode."""

    def __init__(self, cols_img, grad_cols, device, skip_cols=(), read_cols=None, rows_per_col=None, defer_fraction=None):
        """read_cols: the inference columns whose logits / features the step actually READS (weak rows of every pass).  The others are
        computed because the reference computes them (the strong and labelled rows of the passes whose loss is thrown away) -- same
        launches, but nothing waits for them: they run on the second stream behind the gradient rows (``rest``)."""
        gset = set(grad_cols) | set(skip_cols)
        inf_all = [c for c in range(len(cols_img)) if c not in gset]
        rset = set(inf_all) if read_cols is None else set(read_cols)
        inf_cols = [c for c in inf_all if c in rset]
        rest_cols = [c for c in inf_all if c not in rset]
        if rest_cols and rows_per_col:
            # tile quantisation of the deferred launch: its row-streaming kernels own a CU per 128-row tile (fused MLP: 147 KB of LDS), so
            # 128 images x 257 tokens = 257 tiles would be TWO rounds on 256 CUs, the second one a single tile.  A few columns more in the
            # launch that is read anyway bring it back to whole rounds.
            over = (-(-len(rest_cols) * rows_per_col // 128)) % 256
            if 0 < over <= 8:
                nmove = min(-(-over * 128 // rows_per_col), len(rest_cols) - 1)
                inf_cols, rest_cols = sorted(inf_cols + rest_cols[-nmove:]), rest_cols[:-nmove]
            # How many of the inference images are deferred (SR_DEFER_FRACTION).  The workgroups of the row-streaming kernels own a CU
            # outright (147 KB of LDS, all registers) for ~100 us: while a deferred launch fills the chip, every small launch of the critical
            # chain (masks, rewards, losses, the backward of the 16 gradient images) waits for one of them to retire.  Deferring a little
            # under half of the inference images -- instead of all 127 nothing reads -- keeps both launch trains at one round and leaves the
            # deferred one ~64 CUs short of the chip.  Measured on one box, ViT-S/2 (rest images -> img/s): 127 -> 1324, 111 -> 1373,
            # 95 -> 1398, 91 -> 1398, 87 -> 1324; ViT-S/16 at 224: 127 -> 1473, 103 -> 1646, 93 -> 1670, 83 (folded, below) -> 1396.
            # defer_fraction None = the untuned rule (share _DEFER_FRACTION while the deferred launch is a single round of tiles); a tuner's
            # candidate applies at any size
            frac = _DEFER_FRACTION if defer_fraction is None else defer_fraction
            tiles = -(-len(rest_cols) * rows_per_col // 128)
            keep = int(frac * (len(inf_cols) + len(rest_cols)))
            if 0.0 < frac < 1.0 and (tiles <= 256 or defer_fraction is not None) and 0 < keep < len(rest_cols):
                nmove = len(rest_cols) - keep
                inf_cols, rest_cols = sorted(inf_cols + rest_cols[-nmove:]), rest_cols[:-nmove]
        if rest_cols and rows_per_col and len(rest_cols) * rows_per_col < _vit._FUSED_MLP_MIN_ROWS:
            # a deferred launch below the size from which the fused row-streaming kernels are used (nets/vit.py _FUSED_MLP_MIN_ROWS) would
            # run different kernels than the same rows do inside a large launch: it rides in the launch that is read (elide mode: 8 images)
            inf_cols, rest_cols = sorted(inf_cols + rest_cols), []
        t = lambda v, dt: torch.tensor(v, dtype=dt, device=device)   # noqa: E731
        self.grad_cols, self.inf_cols, self.rest_cols = t(list(grad_cols), torch.int64), t(inf_cols, torch.int64), t(rest_cols, torch.int64)
        # the columns in launch order (gradient | read | deferred): a DropPath table drawn in THIS column order hands every launch train a slice
        self.perm_cols = t(list(grad_cols) + list(inf_cols) + list(rest_cols), torch.int64)
        self.grad_img = t([cols_img[c] for c in grad_cols], torch.int32)
        self.inf_img = t([cols_img[c] for c in inf_cols], torch.int32)
        self.rest_img = t([cols_img[c] for c in rest_cols], torch.int32)
        self.ncols = len(cols_img)

    @classmethod
    def cat_passes(cls, nl, nu, K, device, extra_pass0_strong=False, lb_every_pass=True, defer_unread=False, rows_per_col=None,
                   elide_unread=False, defer_fraction=None):
        """use_cat layout of SRFlexMatch / SRFixMatch: every pass is cat(x_lb, x_ulb_w, x_ulb_s); gradients flow from the
        labelled rows of pass 0 and the strong rows of the last pass.  lb_every_pass=False (use_cat=False, the usb_nlp / usb_audio
        configs): data_generator forwards only x_ulb_s and x_ulb_w (srflexmatch.py:83-90), so the labelled columns of the passes
        1..K are never computed (nor read)."""
        Bt = nl + 2 * nu
        cols_img = [j for _ in range(K + 1) for j in range(Bt)]
        grad = list(range(nl)) + ([j for j in range(nl + nu, Bt)] if extra_pass0_strong else []) + \
            [K * Bt + j for j in range(nl + nu, Bt)]
        skip = [] if lb_every_pass else [k * Bt + j for k in range(1, K + 1) for j in range(nl)]
        if elide_unread:
            # OPT-IN extension (not the reference's work): the rows of the passes 1..K whose outputs nothing reads -- labelled rows, and the
            # strong rows of all but the last pass -- are not computed at all.  Rows are independent in the LayerNorm backbones (no batch
            # statistics), so every number the step produces is unchanged; pass 0 stays complete (its features are returned in feat_dict).
            skip = sorted(set(skip) | {k * Bt + j for k in range(1, K + 1) for j in range(nl)} |
                          {k * Bt + j for k in range(1, K) for j in range(nl + nu, Bt)})
        read = [k * Bt + j for k in range(K + 1) for j in range(nl, nl + nu)] if defer_unread else None      # the weak rows
        p = cls(cols_img, grad, device, skip, read, rows_per_col, defer_fraction)
        p.P, p.Bt = K + 1, Bt
        return p


class SRConsistencyBase(AlgorithmBase):
    """Shared step of the confidence-threshold SemiReward algorithms (SRFlexMatch, SRFixMatch): the reference classes
    differ only in their MaskingHook and in whether ``train_step`` receives ``idx_ulb``."""

    def __init__(self, args, net_builder, tb_log=None, logger=None):
        super().__init__(args, net_builder, tb_log, logger)
        self._init_thresholds(args)
        self.N_k = args.N_k
        # sr_ema != 0 selects EMARewarder in the reference (srflexmatch.py:49-50); its forward is identical and its EMA
        # dict is never read, so one Rewarder class covers both.
        self.rewarder = Rewarder(label_dim(self.num_classes), 128, args.feature_dim, device=self.device)
        self.generator = Generator(args.feature_dim, device=self.device)
        self.start_timing = args.start_timing
        self.rewarder_optimizer = FlatAdam(self.rewarder, args.sr_lr)
        self.generator_optimizer = None       # generator grads are None in the reference -> its Adam step is a no-op (A.1)
        self.max_reward = torch.full((), -float("inf"), device=self.device)
        self.dp.broadcast_params(self.model, self.rewarder, self.generator)
        # data parallel: which gradient exchange runs (one all-reduce / reduce-scatter + all-gather, after or under the backward) is measured on
        # the live backend in the first steps and agreed between the ranks (distributed.ExchangeTuner; SR_GRAD_EXCHANGE pins it)
        self._plans = {}
        self._untuned = set()                  # plan keys created while the exchange selection was still measuring steps: tuned once it has settled
        self._tuners = {}                      # plan key -> (_DeferTuner, {share: _Plan}) while the deferred share of that regime is being tuned
        self.defer_share = None                # a share handed in (e.g. the one an earlier leg of a bench was tuned to): no tuning steps
        self.defer_report = {}                 # plan key -> what the tuner measured and chose
        # gradient-row forward on a second HIP stream
        # (args.overlap_grad_rows = False / args.defer_unread_rows = False: the serial schedule, one stream -- tests and A/B runs)
        self.overlap_grad_rows = bool(getattr(args, "overlap_grad_rows", True)) and torch.cuda.is_available()
        # (a stream SEEN to execute beside the constructing stream: two HIP streams may share a hardware queue, ops.concurrent_stream)
        self._side_stream = ops.concurrent_stream(self.device) if self.overlap_grad_rows else None
        self.dp.side_stream = self._side_stream          # (the exchange's communication stream is chosen beside both)
        self.dp.attach(self.model)
        # rows nothing downstream reads (see _Plan) go behind the gradient rows on the second stream; the step end waits for them
        self.defer_unread_rows = self.overlap_grad_rows and bool(getattr(args, "defer_unread_rows", True))
        # opt-in: do not compute the rows nothing reads (see _Plan.cat_passes); never on by default -- the reference computes them
        self.elide_unread_rows = bool(getattr(args, "elide_unread_rows", os.environ.get("SR_ELIDE_UNREAD_ROWS", "0") != "0"))
        self._rest_done = None
        self._grad_pending = None
        self._phases = []
        self.inject_droppath = None            # tests: list of [depth,2,Bt] tensors, one per pass
        self.trace = None                      # tests: dict filled with per-pass intermediates when not None

    def _init_thresholds(self, args):
        raise NotImplementedError

    def _masks(self, max_probs, max_idx, idx_ulb, P, nu, weak_logits):
        """Per-pass confidence masks [P lists of nu] given the row-max of every pass's weak logits."""
        raise NotImplementedError

    def _fairness(self, logits_s0, mask0, dl_into):
        """Optional extra loss on the pass-0 strong logits (FreeMatch).  Returns a 0-d loss tensor or None; adds its gradient
        (already scaled) into ``dl_into`` when that is not None, else returns (loss, new dlogits)."""
        return None, None


    # ---- batched multi-pass forward -----------------------------------------------------------------
    def _forward_plan(self, imgs, pl, droppath_cols=None):
        """Runs every column of the plan through the backbone (inference rows in one or more no-save launches-trains, grad
        rows with activations kept).  Returns (logits [ncols,C], feats [ncols,D], ctx)."""
        m = self.model
        C, D = self.num_classes, m.cfg.embed_dim
        ng_, ni_ = pl.grad_cols.numel(), pl.inf_cols.numel()
        capturing = torch.cuda.is_current_stream_capturing()      # (a captured step owns its allocations: no record_stream bookkeeping)
        if droppath_cols is not None:
            dp_all = droppath_cols.to(self.device)                                                   # [depth,2,ncols]
            sel = lambda cols, a: dp_all.index_select(2, cols).contiguous()                          # noqa: E731
        elif m.training and m.cfg.drop_path_rate > 0:
            if getattr(m, "droppath_by_cols", False):
                # ONE launch: the columns of the draw in launch order; a launch train's table is the slice [a, a + len(cols))
                dp_all = m.make_droppath(pl.ncols, cols=pl.perm_cols)
                sel = lambda cols, a: dp_all[:, :, a:a + cols.numel()]                               # noqa: E731
            else:
                dp_all = m.make_droppath(pl.ncols)
                sel = lambda cols, a: dp_all.index_select(2, cols).contiguous()                      # noqa: E731
        else:
            dp_all = None
            sel = lambda cols, a: None                                                               # noqa: E731
        scatter = getattr(m, "scatter_outputs", False)
        logits = torch.empty(pl.ncols, C, dtype=torch.float32, device=self.device)
        feats = torch.empty(pl.ncols, D, dtype=torch.float32, device=self.device)
        # The gradient-carrying rows (16 of 216 images at the reference batch) run on a SECOND HIP stream: their launches are
        # 100-400 workgroups of latency-bound work (14-28 us each, 1.6 ms per step back to back) that fit beside the tails of
        # the 200-image inference launches.  Both forwards only read the parameters; they write disjoint workspaces.
        # Host order matters: the 200-image inference launches are enqueued FIRST (6 ms of GPU work in ~60 launches), the ~90
        # small launches of the gradient rows (1.3 ms of host enqueue time) after them on the other stream, gated only by an
        # event recorded before either -- enqueued the other way round the main stream sat idle while the host was still feeding
        # the side stream (rocprof timeline: a 1.28 ms hole).
        main = torch.cuda.current_stream()
        side = self._side_stream if self.overlap_grad_rows else None
        dp_grad = sel(pl.grad_cols, 0)
        ni, nr = pl.inf_cols.numel(), pl.rest_cols.numel()
        chunks = [(pl.inf_cols, pl.inf_img, ng_)] if ni else []
        if side is None and nr:
            chunks.append((pl.rest_cols, pl.rest_img, ng_ + ni))
        dps = [sel(cols, a) for cols, _, a in chunks]
        dp_rest = sel(pl.rest_cols, ng_ + ni) if (side is not None and nr) else None
        chunks = [(cols, imgi) for cols, imgi, _ in chunks]
        if side is not None:
            ready = torch.cuda.Event()
            ready.record(main)                       # parameters, images, DropPath draws are final here
        for (cols, imgi), dpi in zip(chunks, dps):
            if scatter:                  # the head writes the rows of the step's tables itself
                m.forward_features(imgs, imgi, dpi, save=False, out=(logits, feats, cols))
            else:
                lg, ft, _ = m.forward_features(imgs, imgi, dpi, save=False)
                logits.index_copy_(0, cols, lg)
                feats.index_copy_(0, cols, ft)
        if side is not None:
            side.wait_event(ready)
            # tensors allocated on the main stream that the second stream keeps reading after this function returns: tell the caching
            # allocator (a freed DropPath table was handed to the next main-stream allocation while the deferred rows still read it)
            for t_ in (logits, feats, dp_grad, dp_rest, imgs, getattr(imgs, "ids", None), getattr(imgs, "key_len", None),
                       getattr(imgs, "seq_len", None)):
                if torch.is_tensor(t_) and not capturing:
                    t_.record_stream(side)
            with torch.cuda.stream(side), ops.stream_scope():
                if hasattr(m, "ensure_transposed"):
                    m.ensure_transposed()          # backward-only operands of the new parameters: here they delay nothing
                lg_g, ft_g, ctx = m.forward_features(imgs, pl.grad_img, dp_grad, save=True,
                                                     **(dict(out=(logits, feats, pl.grad_cols)) if scatter else {}))
                grad_done = torch.cuda.Event()
                grad_done.record(side)
                if _PHASES:
                    self._phase_mark("side:grad_rows_forwarded")
            with torch.cuda.stream(side), ops.stream_scope():
                if nr:
                    # Rows whose outputs nothing reads before the step ends (strong / labelled rows of the passes whose loss the
                    # reference discards): 60 % of the forward work, off the critical path.  The masks, losses and the latency-bound
                    # backward of the 16 gradient images (small launches that leave most CUs idle) run on the main stream meanwhile.
                    if scatter:
                        m.forward_features(imgs, pl.rest_img, dp_rest, save=False, buftag="r", out=(logits, feats, pl.rest_cols))
                    else:
                        lg_r, ft_r, _ = m.forward_features(imgs, pl.rest_img, dp_rest, save=False, buftag="r")
                        logits.index_copy_(0, pl.rest_cols, lg_r)
                        feats.index_copy_(0, pl.rest_cols, ft_r)
                    self._rest_done = torch.cuda.Event()
                    self._rest_done.record(side)
                    if _PHASES:
                        self._phase_mark("side:deferred_rows_done")
            # The gradient rows are joined LATER (_join_grad): masks, pseudo labels and reward scores only read the weak rows of the launch
            # above, so that chain (~0.3 ms of tiny sequential launches) runs while the second stream still works on the gradient rows.
            self._grad_pending = (grad_done, lg_g, ft_g, logits, feats, None if scatter else pl.grad_cols)
            return logits, feats, ctx
        if scatter:
            _, _, ctx = m.forward_features(imgs, pl.grad_img, dp_grad, save=True, out=(logits, feats, pl.grad_cols))
            return logits, feats, ctx
        lg_g, ft_g, ctx = m.forward_features(imgs, pl.grad_img, dp_grad, save=True)
        logits.index_copy_(0, pl.grad_cols, lg_g)
        feats.index_copy_(0, pl.grad_cols, ft_g)
        return logits, feats, ctx

    def _join_grad(self):
        """Logits / features of the gradient rows become visible on the step's stream."""
        if self._grad_pending is not None:
            grad_done, lg_g, ft_g, logits, feats, cols = self._grad_pending
            self._grad_pending = None
            main = torch.cuda.current_stream()
            main.wait_event(grad_done)
            if cols is None:              # the gradient rows' head wrote the step's tables itself
                return
            lg_g.record_stream(main)
            ft_g.record_stream(main)
            logits.index_copy_(0, cols, lg_g)
            feats.index_copy_(0, cols, ft_g)

    def _step_scope(self):
        return ops.stream_scope()

    def _phase_mark(self, name):
        """Tuning aid (SR_PHASES=1): GPU timestamps (events on the step's stream) + host timestamps of the phases of train_step."""
        import time
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        self._phases.append((name, e, time.perf_counter()))

    def phase_report(self):
        torch.cuda.synchronize()
        rows, out = self._phases, {}
        i = 0
        while i < len(rows):
            if rows[i][0] == "start":
                j = i + 1
                while j < len(rows) and rows[j][0] != "start":
                    n = rows[j][0]
                    g, h = out.setdefault(n, ([], []))
                    g.append(rows[i][1].elapsed_time(rows[j][1])); h.append(1e3 * (rows[j][2] - rows[i][2]))
                    j += 1
                i = j
            else:
                i += 1
        return {k: (sum(g[-10:]) / len(g[-10:]), sum(h[-10:]) / len(h[-10:])) for k, (g, h) in out.items()}

    def _join_deferred(self):
        """The step is complete (and the parameters may change) only when the deferred rows are done."""
        if self._rest_done is not None:
            torch.cuda.current_stream().wait_event(self._rest_done)
            self._rest_done = None

    graph_safe = False         # core/stepgraph.py: the step keeps no per-step state in Python objects (set by the subclasses that qualify)

    def step_variant(self):
        """What of the step's control flow depends on ``it`` (srflexmatch.py:147, :154-208): K = sr_decay() and which SemiReward update runs."""
        it = self.it
        K = self.sr_decay() if it > self.start_timing else 0
        sr = 0 if it <= 0 else (1 if it < self.start_timing else (3 if (it % self.N_k == 0 and it > self.start_timing) else 2))
        return (K, sr)

    fairness_rows = False      # FreeMatch: the pass-0 strong rows also carry a gradient
    masks_read_labelled_rows = False      # SoftMatch with the 'model' alignment target: the masks need the labelled logits of pass 0
    masks_need_weak_logits = True         # the thresholding hook reads the weak logits themselves (FreeMatch, SoftMatch), not only max / argmax

    def _tokens(self, x):
        from ..nets.bert import TokenBatch
        return x if isinstance(x, TokenBatch) else TokenBatch.from_dict(x, self.device)

    def _token_cat(self, batches):
        from ..nets.bert import TokenBatch
        return TokenBatch.cat(batches)

    def _make_plan(self, nl, nu, K, defer_fraction=None):
        return _Plan.cat_passes(nl, nu, K, self.device, extra_pass0_strong=self.fairness_rows and K > 0,
                                lb_every_pass=bool(self.use_cat), defer_unread=self.defer_unread_rows,
                                rows_per_col=getattr(self.model.cfg, "num_tokens", None),
                                elide_unread=self.elide_unread_rows, defer_fraction=defer_fraction)

    def _forward_passes(self, imgs, nl, nu, K):
        key = (nl, nu, K, bool(self.use_cat), self.elide_unread_rows)
        if key in self._untuned and self.dp.settled:
            self._untuned.discard(key)
            del self._plans[key]
        if key not in self._plans:
            if self.elide_unread_rows and not getattr(self.model, "rows_independent", False):
                raise ValueError("elide_unread_rows needs a backbone without batch statistics (ViT / BERT / Wav2Vec2 engines)")
            self._plans[key] = self._make_plan(nl, nu, K, defer_fraction=self.defer_share)
            if not self.dp.settled:
                self._untuned.add(key)               # (two tuners varying the step at once would time each other)
            elif _DEFER_AUTOTUNE and self.defer_share is None and self.defer_unread_rows and self._plans[key].rest_cols.numel() > 0:
                # candidates that give distinct (read | deferred) splits; a split whose deferred launch would fall below the fused kernels'
                # launch size is folded by _Plan and drops out here
                cand, seen = {}, set()
                # a regime that differs only in K (sr_decay() walks through dozens of values early in a run) starts from the share its
                # neighbour was tuned to and measures that share and its two neighbours only: 12 tuning steps instead of ~32 per new K
                prev = [(abs(k_[2] - K), r["chosen"]) for k_, r in self.defer_report.items() if (k_[0], k_[1], k_[3], k_[4]) == (nl, nu, key[3], key[4])]
                seeds = _DeferTuner.CANDIDATES
                if prev:
                    b_ = min(prev)[1]
                    seeds = [f_ for f_ in (round(b_ - _DeferTuner.REFINE, 4), b_, round(b_ + _DeferTuner.REFINE, 4)) if 0.05 < f_ <= 1.0]
                for f in seeds:
                    p_ = self._make_plan(nl, nu, K, defer_fraction=f)
                    n_ = int(p_.rest_cols.numel())
                    if n_ > 0 and n_ not in seen:
                        seen.add(n_)
                        cand[f] = p_
                if len(cand) > 1:
                    agree = (lambda ms: self.dp.max_over_ranks(ms, self.device)) if self.dp.active else None
                    # a regime abandoned mid-tuning (K or the batch shape moved on before its windows finished) will not come back to finish:
                    # its tuner and candidate plans go (they would keep StepGraph in eager mode and hold device tensors for the rest of the run)
                    # (its plan goes with it: the regime is tuned from scratch if it does come back -- a partial last batch once per epoch --
                    # instead of running on the untuned seed share for the rest of the run)
                    for k_ in [k_ for k_ in self._tuners if k_ != key]:
                        del self._tuners[k_]
                        self._plans.pop(k_, None)
                    self._tuners[key] = (_DeferTuner(cand.keys(), refine=not prev, agree=agree), cand)
                elif prev:
                    # the neighbour's candidates collapse to one split at this size: that split, not the untuned rule
                    self._plans[key] = self._make_plan(nl, nu, K, defer_fraction=min(prev)[1])
        tn = self._tuners.get(key)
        if tn is not None:
            tuner, cand = tn
            f = tuner.fraction()
            if f not in cand:                  # a neighbour of the coarse winner (second pass)
                cand[f] = self._make_plan(nl, nu, K, defer_fraction=f)
            self._plans[key] = cand[f]
            if tuner.done:
                self.defer_report[key] = dict(chosen=tuner.best, deferred_images=int(self._plans[key].rest_cols.numel()), ms_per_step=tuner.report)
                del self._tuners[key]
        pl = self._plans[key]
        dpc = torch.cat([d for d in self.inject_droppath[:pl.P]], dim=2) if self.inject_droppath is not None else None
        logits, feats, ctx = self._forward_plan(imgs, pl, dpc)
        return logits.view(pl.P, pl.Bt, -1), feats.view(pl.P, pl.Bt, -1), ctx

    def _sr_update(self, feats, gen_labels, ref_labels):
        """srflexmatch.py:179-193 / :194-208: target, MSE(r,1) + MSE(r,t), both backward() into the rewarder, Adam."""
        feats = feats.contiguous()
        target = cosine_target(gen_labels, ref_labels, self.num_classes)
        self.rewarder.train()
        self.rewarder.score(feats, gen_labels, groups=1, save_for_bwd=True)
        losses = torch.empty(2, dtype=torch.float32, device=self.device)
        self.rewarder.backward_mse(feats, gen_labels, target, losses)
        if self.dp.active:
            self.dp.all_reduce_flat(self.rewarder.grad).mul_(1.0 / self.world_size)
        self.rewarder_optimizer.step()
        if ops._CHECK_ARGS and not torch.cuda.is_current_stream_capturing():     # test suite: surface an out-of-range label at the update itself
            ops.check_label_errors()        # (production: AlgorithmBase.train checks at the logging cadence and before a checkpoint -- the check synchronises)
        if self.trace is not None:
            self.trace.update(sr_target=target, sr_losses=losses)

    def _train_step(self, x_lb, y_lb, idx_ulb, x_ulb_w, x_ulb_s):
        it = self.it
        if getattr(self.model, "takes_tokens", False):
            # usb_nlp: x_* are {'input_ids','attention_mask'} dicts, each batch padded to its own longest row (nlp_collactor.py:63-69);
            # the reference forwards them in separate model calls (use_cat=False, :118-128) -- here they share the batched launches,
            # filled up to the longest of the three (TokenBatch.cat: identical results, see nets/bert.py)
            if self.use_cat:
                raise ValueError("use_cat: True with token batches: batches of different padded lengths cannot be torch.cat'ed "
                                 "(the reference fails in torch.cat, srsoftmatch.py:118); the usb_nlp / usb_audio configs set use_cat False")
            tb = [self._tokens(x) for x in (x_lb, x_ulb_w, x_ulb_s)]
            nl, nu = tb[0].S, tb[1].S
            imgs = self._token_cat(tb)
        else:
            nl, nu = y_lb.shape[0], x_ulb_w.shape[0]
            imgs = torch.cat((x_lb, x_ulb_w, x_ulb_s)).contiguous()                              # :113
        K = self.sr_decay() if it > self.start_timing else 0                                     # :147, :75
        # tile choice of the small GEMMs of the gradient rows: with K > 0 the inference launches of the K passes own most CUs while they run
        # (fewer, fatter workgroups win: 4.86 vs 4.95 ms per step); with K = 0 the chain has the chip to itself (64 x 64 tiles: 3.21 vs 3.57 ms)
        # (process-wide library setting: restored when the step returns -- train_step's wrapper -- so that evaluate(), the EMA forward and
        # anything else that runs between steps picks its tiles independently of which step ran last)
        ops.gemm_small_max_grid(ops.GEMM_SMALL_CONTENDED if (K > 0 and self.defer_unread_rows) else ops.GEMM_SMALL_ALONE)
        ph = self._phase_mark if _PHASES else (lambda name: None)
        ph("start")
        L, Fe, ctx = self._forward_passes(imgs, nl, nu, K)
        ph("forward_joined")
        P, C = K + 1, self.num_classes
        # softmax + max/argmax of the weak logits of ALL passes: one launch (compute_prob :135, argmax :142-146)
        # (the weak rows of pass p are rows p * Bt + nl .. + nu of the logits table: read in place; only the thresholding hooks that want the
        # logits themselves -- FreeMatch, SoftMatch -- get a gathered copy)
        Bt = L.shape[1]
        mp = torch.empty(P * nu, dtype=torch.float32, device=self.device)
        mi = torch.empty(P * nu, dtype=torch.int64, device=self.device)
        if self.masks_need_weak_logits:
            Lw = L[:, nl:nl + nu].reshape(P * nu, C)
            ops.row_max(Lw, False, None, mp, mi, P * nu, C)
        else:
            Lw = None
            ops.row_max_strided(L, nl, False, None, mp, mi, P * nu, C, nu, Bt)
        self._lb_logits0 = L[0, :nl]                       # SoftMatch's 'model' alignment target reads the labelled rows of pass 0
        if self.masks_read_labelled_rows:
            self._join_grad()
        masks = self._masks(mp, mi, idx_ulb, P, nu, Lw)
        pl0 = mi[:nu]
        if K > 0:
            self.rewarder.eval()                                                                  # :74
            # the weak rows of passes 1 .. K, scored where they are in the feature table (group g = rows (g + 1) * Bt + nl .. + nu)
            reward = self.rewarder.score_in_place(Fe, Bt + nl, Bt, nu, mi[nu:], groups=K)          # :99
            mask2 = torch.empty_like(reward)
            mean_in = self.dp.reward_means(reward, K).contiguous() if (self.dp.global_reward_threshold and self.dp.active) else None
            ops.reward_mask2(reward, mask2, None, K, nu, mean_in=mean_in)                          # :100-101
        self._join_grad()                                  # from here on the labelled / strong rows of the loss are read
        # the upstream gradient of the gradient rows (labelled rows of pass 0 | strong rows of the last pass) is ONE buffer the two loss
        # launches fill (the fairness rows of FreeMatch with K > 0 sit between them: that case still concatenates)
        n_strong = L.shape[1] - nl - nu
        two_blocks = not (self.fairness_rows and K > 0)
        dl_buf = torch.empty(nl + n_strong, C, dtype=torch.float32, device=self.device) if two_blocks else None
        sup_loss, dl_lb = self.ce_loss(L[0, :nl], y_lb, reduction="mean", dl_out=dl_buf[:nl] if two_blocks else None)          # :132
        if K > 0:
            plK, mK, m2K = mi[K * nu:], masks[K], mask2[(K - 1) * nu:]
            unsup_loss, dl_s = self.consistency_loss(L[K, nl + nu:], plK, "ce", mask=mK, mask2=m2K, grad_scale=self.lambda_u,
                                                     dl_out=dl_buf[nl:] if two_blocks else None)                                # :102
        else:
            reward = mask2 = None
            unsup_loss, dl_s = self.consistency_loss(L[0, nl + nu:], pl0, "ce", mask=masks[0], grad_scale=self.lambda_u,
                                                     dl_out=dl_buf[nl:] if two_blocks else None)                                # :152
        # optional fairness term on the pass-0 strong rows (FreeMatch): same rows as dl_s when K == 0, extra grad rows otherwise
        if K > 0:
            ent_loss, dl_e = self._fairness(L[0, nl + nu:], masks[0], None)
            dl_all = (dl_lb, dl_e, dl_s) if dl_e is not None else (dl_lb, dl_s)
        else:
            ent_loss, _ = self._fairness(L[0, nl + nu:], masks[0], dl_s)
            dl_all = (dl_lb, dl_s)
        # ---- backbone backward: only the rows with a non-zero upstream gradient (see module docstring)
        ph("losses")
        fx, fw0 = Fe[0, :nl], Fe[0, nl:nl + nu]
        # :166-170 (the filter is a no-op, A.2): reward.mean() and the running maximum ride in ONE scoring launch of one workgroup (~55 us).  It
        # reads the pass-0 weak features and the rewarder only, so it runs on the side stream UNDER the backward instead of behind it (behind it,
        # it was 55 us of the step's serial tail: wgrad -> this -> AdamW); the step's stream waits for it before anything touches the rewarder
        score_done = None
        if it >= self.start_timing and it > 0:                                                    # :163
            side = self._side_stream if (self.overlap_grad_rows and _SCORE_ON_SIDE and not torch.cuda.is_current_stream_capturing()) else None
            fw0c = fw0.contiguous()
            if side is None:
                self.rewarder.score(fw0c, pl0, max_reward=self.max_reward)
            else:
                main = torch.cuda.current_stream()
                ready = torch.cuda.Event()
                ready.record(main)
                with torch.cuda.stream(side):
                    side.wait_event(ready)
                    with ops.stream_scope():
                        self.rewarder.score(fw0c, pl0, max_reward=self.max_reward)
                    score_done = torch.cuda.Event()
                    score_done.record(side)
        # Stage 1 (0 < it < start_timing, :156-159 / :194-208): the rewarder learns from the LABELLED features and labels -- nothing the backbone's
        # backward produces -- and nothing else of this step reads the rewarder (K = 0: no scoring).  ~0.23 ms of small dependent launches that
        # sat behind the backward on the step's critical path (7 % of the K = 0 step) run on the side stream UNDER the backward instead.
        stage1_done = None
        if 0 < it < self.start_timing:
            side = self._side_stream if (self.overlap_grad_rows and _SCORE_ON_SIDE) else None
            if side is None:
                stage1_after = True
            else:
                stage1_after = False
                main = torch.cuda.current_stream()
                ready = torch.cuda.Event()
                ready.record(main)                          # the labelled features (gradient rows joined above) and labels are final here
                if torch.is_tensor(y_lb) and not torch.cuda.is_current_stream_capturing():
                    y_lb.record_stream(side)
                with torch.cuda.stream(side):
                    side.wait_event(ready)
                    with ops.stream_scope():
                        gen = self.generator.forward_with_labels(fx.contiguous())[1]              # :158-159
                        self._sr_update(fx, gen, y_lb)                                            # :194-208
                    stage1_done = torch.cuda.Event()
                    stage1_done.record(side)
        self.model.backward(ctx, dl_buf if (two_blocks and len(dl_all) == 2) else torch.cat(dl_all))
        if score_done is not None:
            torch.cuda.current_stream().wait_event(score_done)
        if stage1_done is not None:
            torch.cuda.current_stream().wait_event(stage1_done)
        ph("backward")
        # ---- rewarder / generator training (:154-208)
        if it > 0:
            if it >= self.start_timing:                                                           # :163
                if it % self.N_k == 0 and it > self.start_timing:                                 # :173
                    self.max_reward.fill_(-float("inf"))                                            # (in place: the buffer of a captured step)
                    gen2 = self.generator.forward_with_labels(fw0.contiguous())[1]                # :177-178
                    self._sr_update(fw0, gen2, pl0)
            elif stage1_after:
                gen = self.generator.forward_with_labels(fx.contiguous())[1]                      # :158-159
                self._sr_update(fx, gen, y_lb)                                                    # :194-208
        total_loss = torch.add(sup_loss, unsup_loss, alpha=self.lambda_u)                         # :210 (one launch)
        if ent_loss is not None:
            total_loss = total_loss + self.lambda_e * ent_loss                                    # srfreematch.py:220
        if self.trace is not None:
            self.trace.update(K=K, masks=masks, max_probs=mp, pseudo=mi, reward=reward, mask2=mask2, logits=L, feats=Fe)
        ph("sr_update")
        self._join_deferred()
        ph("deferred_joined")
        feat_dict = {"x_lb": fx, "x_ulb_w": fw0, "x_ulb_s": Fe[0, nl + nu:]}
        out_dict = self.process_out_dict(loss=total_loss, feat=feat_dict)
        log_dict = self.process_log_dict(sup_loss=DeferredScalar(sup_loss), unsup_loss=DeferredScalar(unsup_loss),
                                         total_loss=DeferredScalar(total_loss), util_ratio=DeferredScalar(lambda m=masks[0]: m.mean()))
        return out_dict, log_dict

    def _sr_save(self, d):
        # n4: the reference forgets the SR state on resume; keep it under extra keys
        d["sr_rewarder"], d["sr_generator"] = self.rewarder.state_dict(), self.generator.state_dict()
        d["sr_rewarder_optimizer"], d["sr_max_reward"] = self.rewarder_optimizer.state_dict(), float(self.max_reward)
        return d

    def _sr_load(self, ck):
        if "sr_rewarder" in ck:
            self.rewarder.load_state_dict(ck["sr_rewarder"]); self.generator.load_state_dict(ck["sr_generator"])
            self.rewarder_optimizer.load_state_dict(ck["sr_rewarder_optimizer"])
            self.max_reward.fill_(float(ck["sr_max_reward"]))       # in place: a captured step graph keeps reading this buffer
        return ck

    def get_save_dict(self):
        return self._sr_save(super().get_save_dict())

    def load_model(self, load_path):
        return self._sr_load(super().load_model(load_path))


@ALGORITHMS.register("srflexmatch")
class SRFlexMatch(SRConsistencyBase):
    """semilearn/algorithms/srflexmatch/srflexmatch.py:15-246."""

    graph_safe = True          # hook state = persistent device tables updated in place

    masks_need_weak_logits = False        # FlexMatch thresholds on (max prob, argmax) only

    def _init_thresholds(self, args):
        self.init(T=args.T, p_cutoff=args.p_cutoff, hard_label=args.hard_label, thresh_warmup=args.thresh_warmup)

    def init(self, T, p_cutoff, hard_label=True, thresh_warmup=True):
        self.T, self.p_cutoff, self.use_hard_label, self.thresh_warmup = T, p_cutoff, hard_label, thresh_warmup

    def set_hooks(self):
        self.register_hook(PseudoLabelingHook(), "PseudoLabelingHook")
        self.register_hook(FlexMatchThresholdingHook(ulb_dest_len=self.args.ulb_dest_len, num_classes=self.num_classes,
                                                     thresh_warmup=self.args.thresh_warmup, device=self.device), "MaskingHook")
        super().set_hooks()

    def _masks(self, mp, mi, idx_ulb, P, nu, weak_logits=None):
        # order dependent (mutates selected_label / classwise_acc): pass 0 first, then the K loop passes
        hook = self.hooks_dict["MaskingHook"]
        return hook.masking_passes(self, mp, mi, idx_ulb, P)        # one launch, the passes in order inside it

    def train_step(self, x_lb, y_lb, idx_ulb, x_ulb_w, x_ulb_s):
        with self._step_scope():
            try:
                return self._train_step(x_lb, y_lb, idx_ulb, x_ulb_w, x_ulb_s)
            finally:
                ops.gemm_small_max_grid(ops.GEMM_SMALL_ALONE)

    def get_save_dict(self):
        d = super().get_save_dict()
        d["classwise_acc"] = self.hooks_dict["MaskingHook"].classwise_acc.cpu()
        d["selected_label"] = self.hooks_dict["MaskingHook"].selected_label.cpu()
        return d

    def load_model(self, load_path):
        ck = super().load_model(load_path)
        h = self.hooks_dict["MaskingHook"]
        h.classwise_acc = ck["classwise_acc"].to(self.device)
        h.selected_label = ck["selected_label"].to(self.device)
        return ck

    @staticmethod
    def get_argument():
        return [SSL_Argument("--hard_label", str2bool, True), SSL_Argument("--T", float, 0.5),
                SSL_Argument("--p_cutoff", float, 0.95), SSL_Argument("--thresh_warmup", str2bool, True),
                SSL_Argument("--start_timing", int, 20000), SSL_Argument("--feature_dim", int, 384),
                SSL_Argument("--sr_lr", float, 0.0005), SSL_Argument("--N_k", int, 10),
                SSL_Argument("--sr_ema", str2bool, True), SSL_Argument("--sr_ema_m", float, 0.999)]


@ALGORITHMS.register("srfixmatch")
class SRFixMatch(SRConsistencyBase):
    """semilearn/algorithms/srfixmatch/fixmatch.py:13-225: FixMatch + SemiReward.  Same step as SRFlexMatch with the
    stateless FixedThresholdingHook (masking.py:42-57) -> the masks of all passes come from ONE launch."""

    masks_need_weak_logits = False        # a fixed threshold on the max probability
    graph_safe = True                     # stateless hook

    def _init_thresholds(self, args):
        self.init(T=args.T, p_cutoff=args.p_cutoff, hard_label=args.hard_label)

    def init(self, T, p_cutoff, hard_label=True):
        self.T, self.p_cutoff, self.use_hard_label = T, p_cutoff, hard_label

    def set_hooks(self):
        self.register_hook(PseudoLabelingHook(), "PseudoLabelingHook")
        self.register_hook(FixedThresholdingHook(), "MaskingHook")
        super().set_hooks()

    def _masks(self, mp, mi, idx_ulb, P, nu, weak_logits=None):
        m = torch.empty_like(mp)
        ops.fixed_mask(mp, float(self.p_cutoff), m, P * nu)
        return [m[k * nu:(k + 1) * nu] for k in range(P)]

    def train_step(self, x_lb, y_lb, x_ulb_w, x_ulb_s):
        with self._step_scope():
            try:
                return self._train_step(x_lb, y_lb, None, x_ulb_w, x_ulb_s)
            finally:
                ops.gemm_small_max_grid(ops.GEMM_SMALL_ALONE)

    @staticmethod
    def get_argument():
        return [SSL_Argument("--hard_label", str2bool, True), SSL_Argument("--T", float, 0.5),
                SSL_Argument("--p_cutoff", float, 0.95), SSL_Argument("--start_timing", int, 20000),
                SSL_Argument("--feature_dim", int, 384), SSL_Argument("--sr_lr", float, 0.0005),
                SSL_Argument("--N_k", int, 10), SSL_Argument("--sr_ema", str2bool, True),
                SSL_Argument("--sr_ema_m", float, 0.999)]
