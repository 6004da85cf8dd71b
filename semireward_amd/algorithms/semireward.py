"""SemiReward Rewarder / Generator on libsrhip (reference: semilearn/algorithms/semireward/semireward.py).

Same class names and constructor arguments as the reference; parameters live in one flat fp32 block in the
reference's ``named_parameters()`` order (``state_dict`` round-trips to the reference's keys).  There is no
autograd: ``Rewarder.score`` is the fused scoring forward for G independent groups, ``Rewarder.update`` is
forward + hand-written backward + Adam in a handful of launches.
"""
import numpy as np
import torch

from .. import ops


def label_dim(x, default_dim=100):
    """semireward.py:147-148."""
    return int(max(default_dim, x))


def _default_init(shapes, seed):
    """nn.Linear / nn.Embedding / nn.LayerNorm default-style init (kaiming-uniform(a=sqrt 5) == U(+-1/sqrt(fan_in)))."""
    g = torch.Generator().manual_seed(seed)
    out = []
    for name, shape in shapes:
        if "norm" in name:
            t = torch.ones(shape) if name.endswith("weight") else torch.zeros(shape)
        elif "embedding" in name:
            t = torch.randn(shape, generator=g)
        else:
            fan_in = shape[1] if len(shape) == 2 else None
            if fan_in is None:      # bias: fan_in of the preceding weight
                fan_in = out[-1].shape[1]
            b = 1.0 / np.sqrt(fan_in)
            t = (torch.rand(shape, generator=g) * 2 - 1) * b
        out.append(t)
    return out


class _FlatModule:
    def __init__(self, shapes, device, seed):
        self.names_shapes = shapes
        self.device = torch.device(device)
        ts = _default_init(shapes, seed)
        self.flat = torch.cat([t.reshape(-1) for t in ts]).to(self.device, torch.float32)
        self.training = True

    def _views(self, buf):
        out, o = {}, 0
        for n, s in self.names_shapes:
            k = int(np.prod(s))
            out[n] = buf[o:o + k].view(s)
            o += k
        return out

    def state_dict(self):
        return {k: v.detach().clone() for k, v in self._views(self.flat).items()}

    def load_state_dict(self, sd):
        for k, v in self._views(self.flat).items():
            v.copy_(torch.as_tensor(sd[k]).to(self.device, torch.float32).reshape(v.shape))

    def named_parameters(self):
        return list(self._views(self.flat).items())

    def parameters(self):
        return [v for _, v in self.named_parameters()]

    def train(self, mode=True):
        self.training = mode
        return self

    def eval(self):
        return self.train(False)


class Generator(_FlatModule):
    """Fake-label generator, F->256->128->64->1 with ReLU everywhere (semireward.py:6-24)."""

    def __init__(self, feature_dim=384, device="cuda", seed=1):
        dims = [feature_dim, 256, 128, 64, 1]
        shapes = []
        for li, i in enumerate((0, 2, 4, 6)):
            shapes += [("fc_layers.%d.weight" % i, (dims[li + 1], dims[li])), ("fc_layers.%d.bias" % i, (dims[li + 1],))]
        super().__init__(shapes, device, seed)
        self.feature_dim = feature_dim
        assert self.flat.numel() == ops.generator_param_count(feature_dim)
        self.flat_t = torch.empty(ops.generator_t_floats(feature_dim), dtype=torch.float32, device=self.device)
        self.prepare()

    def prepare(self):
        """refresh the transposed weight copies the forward kernel streams (after any parameter change)"""
        ops.generator_prepare(self.flat, self.flat_t, self.feature_dim)

    def load_state_dict(self, sd):
        super().load_state_dict(sd)
        self.prepare()

    def __call__(self, x):
        """Returns the [B,1] fp32 output, like the reference module."""
        return self.forward_with_labels(x)[0].view(-1, 1)

    def forward_with_labels(self, x):
        B = x.shape[0]
        out = torch.empty(B, dtype=torch.float32, device=self.device)
        lab = torch.empty(B, dtype=torch.int64, device=self.device)
        ops.generator_fwd(self.flat, self.flat_t, x.contiguous(), out, lab, B, self.feature_dim)
        return out, lab


class Rewarder(_FlatModule):
    """Pseudo-label rewarder (semireward.py:27-72)."""

    def __init__(self, label_dim, label_embedding_dim=128, feature_dim=384, device="cuda", seed=2):
        assert label_embedding_dim == 128, "the reference hard-wires 128 (feature_fc / cross_attention_fc widths)"
        F, L = feature_dim, label_dim
        shapes = [("feature_fc.weight", (128, F)), ("feature_fc.bias", (128,)),
                  ("feature_norm.weight", (128,)), ("feature_norm.bias", (128,)),
                  ("label_embedding.weight", (L, 128)), ("label_norm.weight", (128,)), ("label_norm.bias", (128,)),
                  ("cross_attention_fc.weight", (1, 128)), ("cross_attention_fc.bias", (1,)),
                  ("mlp_fc1.weight", (256, 128)), ("mlp_fc1.bias", (256,)), ("mlp_fc2.weight", (128, 256)), ("mlp_fc2.bias", (128,)),
                  ("ffn_fc1.weight", (64, 128)), ("ffn_fc1.bias", (64,)), ("ffn_fc2.weight", (1, 64)), ("ffn_fc2.bias", (1,))]
        super().__init__(shapes, device, seed)
        self.feature_dim, self.label_dim = F, L
        assert self.flat.numel() == ops.rewarder_param_count(F, L)
        self.grad = torch.zeros_like(self.flat)
        self._ws = {}
        self.flat_t = torch.empty(ops.rewarder_t_floats(F), dtype=torch.float32, device=self.device)
        self.prepare()

    def prepare(self):
        """refresh the transposed weight copies the forward kernels stream (after any parameter change)"""
        ops.rewarder_prepare(self.flat, self.flat_t, self.feature_dim, self.label_dim)

    def load_state_dict(self, sd):
        super().load_state_dict(sd)
        self.prepare()

    def _workspace(self, G, B):
        key = (G, B)
        if key not in self._ws:
            self._ws[key] = torch.empty(ops.rewarder_ws_floats(G, B), dtype=torch.float32, device=self.device)
        return self._ws[key]

    def score(self, features, label_indices, groups=1, save_for_bwd=False, max_reward=None):
        """reward [groups*B]; every group of B consecutive rows has its own batch-softmax (semireward.py:60-62).  max_reward (0-d device
        tensor): also max_reward <- max(max_reward, reward.mean()) (srflexmatch.py:166-170), inside the launch when it is a single row tile."""
        R = features.shape[0]
        B = R // groups
        reward = torch.empty(R, dtype=torch.float32, device=self.device)
        fused_max = max_reward is not None and groups == 1 and B <= 8
        ops.rewarder_fwd(self.flat, self.flat_t, features.contiguous(), label_indices.contiguous(), reward, self._workspace(groups, B),
                         groups, B, self.feature_dim, self.label_dim, save_for_bwd, max_reward=max_reward if fused_max else None)
        if max_reward is not None and not fused_max:
            rm = reward.mean()
            max_reward.copy_(torch.where(rm > max_reward, rm, max_reward))     # srflexmatch.py:170 (keeps the old maximum for a NaN mean, like fmaxf in the fused launch)
        return reward

    def score_in_place(self, feat_buffer, first_row, group_rows, rows_per_group, label_indices, groups):
        """score() on feature rows read where they are: group g = rows first_row + g * group_rows .. + rows_per_group of the contiguous
        [*, F] buffer (the weak rows of pass g + 1 inside the step's feature buffer) -- no gathered copy."""
        reward = torch.empty(groups * rows_per_group, dtype=torch.float32, device=self.device)
        ops.rewarder_fwd(self.flat, self.flat_t, feat_buffer, label_indices.contiguous(), reward, self._workspace(groups, rows_per_group),
                         groups, rows_per_group, self.feature_dim, self.label_dim, False, feats_first_row=first_row, group_rows=group_rows)
        return reward

    def __call__(self, features, label_indices):
        return self.score(features, label_indices).view(-1, 1)

    def backward_mse(self, features, label_indices, target, losses=None):
        """grad <- d[MSE(r,1) + MSE(r,target)]/dtheta for the forward just run with save_for_bwd=True."""
        B = features.shape[0]
        ops.rewarder_bwd(self.flat, features.contiguous(), label_indices.contiguous(), target.contiguous(), self._workspace(1, B),
                         self.grad, losses, B, self.feature_dim, self.label_dim)


class FlatAdam:
    """torch.optim.Adam(params, lr) on a flat block (srflexmatch.py:54-55)."""

    def __init__(self, module, lr):
        self.module, self.lr, self.steps = module, lr, 0
        self.m = torch.zeros_like(module.flat)
        self.v = torch.zeros_like(module.flat)

    def zero_grad(self):
        pass            # rewarder_bwd overwrites the whole gradient block

    def step(self):
        self.steps += 1
        sc = getattr(self, "step_scalars", None)            # core/stepgraph.py: the step's bias corrections in device memory
        ops.adam_flat(self.module.flat, self.module.grad, self.m, self.v, self.module.flat.numel(), self.lr, self.steps,
                      dyn=sc.adam_ptr if sc is not None else None)
        self.module.prepare()

    def state_dict(self):
        return dict(m=self.m.cpu(), v=self.v.cpu(), steps=self.steps)

    def load_state_dict(self, sd):
        self.m.copy_(sd["m"]); self.v.copy_(sd["v"]); self.steps = int(sd["steps"])


def cosine_target(gen_labels, ref_labels, num_classes=0):
    """(cosine_similarity_n(one_hot(gen), one_hot(ref)) + 1)/2 == 1.0 / 0.5 (semireward.py:130-139, srflexmatch.py:180-182).  With
    num_classes > 0 a label outside [0, num_classes) -- where F.one_hot raises -- sets the label-error flag (ops.check_label_errors)."""
    B = gen_labels.numel()
    t = torch.empty(B, dtype=torch.float32, device=gen_labels.device)
    ops.sr_target(gen_labels.contiguous(), ref_labels.contiguous(), t, B, num_classes)
    return t
