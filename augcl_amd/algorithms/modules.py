"""Networks of the hot path (drop-in for src/algorithms/modules.py:9-276).

Module / parameter names follow the reference (`encoder.shared_cnn.layers.<i>`,
`encoder.projection.projection.{0,1}`, `Q1.trunk.{0,2,4}`, `mlp.{0,2,4}`) so state dicts
interchange, but the conv stack does not run layer by layer: `SharedCNN.forward` hands the whole
stack (x/255 -> conv3x3 s2 -> [ReLU -> conv3x3 s1]*) to `encoder_ops.conv_stack`, which runs the
hand-written HIP encoder kernels (forward, backward-data, backward-weights) as ONE autograd node.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import encoder_ops, head_ops


def gaussian_logprob(noise, log_std):
    """modules.py:19-22"""
    residual = (-0.5 * noise.pow(2) - log_std).sum(-1, keepdim=True)
    return residual - 0.5 * np.log(2 * np.pi) * noise.size(-1)


def squash(mu, pi, log_pi):
    """modules.py:25-32 (tanh squashing with the log-det correction)"""
    mu = torch.tanh(mu)
    if pi is not None:
        pi = torch.tanh(pi)
    if log_pi is not None:
        log_pi = log_pi - torch.log(F.relu(1 - pi.pow(2)) + 1e-6).sum(-1, keepdim=True)
    return mu, pi, log_pi


def weight_init(m):
    """modules.py:52-66: orthogonal Linear, delta-orthogonal Conv2d (centre tap only), zero bias."""
    if isinstance(m, nn.Linear):
        nn.init.orthogonal_(m.weight.data)
        if m.bias is not None:
            m.bias.data.fill_(0.0)
    elif isinstance(m, nn.Conv2d):
        assert m.weight.size(2) == m.weight.size(3)
        m.weight.data.fill_(0.0)
        if m.bias is not None:
            m.bias.data.fill_(0.0)
        mid = m.weight.size(2) // 2
        nn.init.orthogonal_(m.weight.data[:, :, mid, mid], nn.init.calculate_gain("relu"))


class CenterCrop(nn.Module):
    """modules.py:69-82 (placeholder at index 0 of SharedCNN.layers; 100 -> 84 crop otherwise identity)"""

    def __init__(self, size):
        super().__init__()
        assert size in {84, 100}, f"unexpected size: {size}"
        self.size = size

    def forward(self, x):
        assert x.ndim == 4, "input must be a 4D tensor"
        if x.size(2) == self.size and x.size(3) == self.size:
            return x
        assert x.size(3) == 100, f"unexpected size: {x.size(3)}"
        p = (100 - self.size) // 2
        return x[:, :, p:-p, p:-p].contiguous()


class NormalizeImg(nn.Module):
    """modules.py:85-90; fused into the first conv kernel, kept for index parity."""

    def forward(self, x):
        return x / 255.0


class Flatten(nn.Module):
    def forward(self, x):
        return x.view(x.size(0), -1)


def _conv_out(h, stride):
    return (h - 3) // stride + 1


class SharedCNN(nn.Module):
    """modules.py:145-165.  layers = [CenterCrop, NormalizeImg, Conv(s2), (ReLU, Conv)*]."""

    def __init__(self, obs_shape, num_layers=11, num_filters=32):
        super().__init__()
        assert len(obs_shape) == 3
        self.num_layers, self.num_filters = num_layers, num_filters
        layers = [CenterCrop(size=84), NormalizeImg(), nn.Conv2d(obs_shape[0], num_filters, 3, stride=2)]
        for _ in range(1, num_layers):
            layers += [nn.ReLU(), nn.Conv2d(num_filters, num_filters, 3, stride=1)]
        self.layers = nn.Sequential(*layers)
        hw = _conv_out(84, 2)
        for _ in range(1, num_layers):
            hw = _conv_out(hw, 1)
        self.out_shape = torch.Size((num_filters, hw, hw))
        self.apply(weight_init)

    precision = None  # None = encoder_ops' process default; SAC.set_precision sets it per agent

    def convs(self):
        return [m for m in self.layers if isinstance(m, nn.Conv2d)]

    def forward(self, x):
        x = self.layers[0](x)
        cs = self.convs()
        return encoder_ops.conv_stack(x, [c.weight for c in cs], [c.bias for c in cs], first_stride=2,
                                      input_scale=255.0, relu_first=False, precision=self.precision)

    def forward_with(self, other, x):
        """No-grad forward of this CNN and of `other` (same architecture, other weights) over the same x in one launch
        per layer.  Returns (self(x), other(x)) or None when the fused path does not apply."""
        if torch.is_grad_enabled() or self.num_layers != other.num_layers or self.precision != other.precision:
            return None
        x = self.layers[0](x)
        a, b = self.convs(), other.convs()
        return encoder_ops.conv_stack_dual(x, [c.weight for c in a], [c.bias for c in a], [c.weight for c in b],
                                           [c.bias for c in b], first_stride=2, input_scale=255.0, precision=self.precision)


class HeadCNN(nn.Module):
    """modules.py:168-181: optional extra (ReLU, Conv) pairs, then Flatten."""

    def __init__(self, in_shape, num_layers=0, num_filters=32):
        super().__init__()
        layers = []
        for _ in range(num_layers):
            layers += [nn.ReLU(), nn.Conv2d(num_filters, num_filters, 3, stride=1)]
        layers.append(Flatten())
        self.layers = nn.Sequential(*layers)
        hw = in_shape[1] - 2 * num_layers
        self.out_shape = torch.Size((num_filters * hw * hw,))
        self.apply(weight_init)

    precision = None

    def forward(self, x):
        cs = [m for m in self.layers if isinstance(m, nn.Conv2d)]
        if cs:
            x = encoder_ops.conv_stack(x, [c.weight for c in cs], [c.bias for c in cs], first_stride=1,
                                       input_scale=1.0, relu_first=True, precision=self.precision)
        return x.reshape(x.size(0), -1)


class RLProjection(nn.Module):
    """modules.py:101-111: Linear -> LayerNorm -> Tanh"""

    def __init__(self, in_shape, out_dim):
        super().__init__()
        self.out_dim = out_dim
        self.projection = nn.Sequential(nn.Linear(in_shape[0], out_dim), nn.LayerNorm(out_dim), nn.Tanh())
        self.apply(weight_init)

    def forward(self, x, tail=None):
        """tail (not in the reference's signature): the actions the critic will concatenate to this projection's output;
        returns (features, [features | tail]) when the fused launch can build that row as well, (features, None) if not."""
        lin, ln = self.projection[0], self.projection[1]
        if tail is not None:
            out = None
            if head_ops._ok(x) and ln.weight.is_cuda and self.out_dim <= 128:
                out = head_ops.proj_ln_tanh(x, lin, ln, tail=tail)
            return out if out is not None else (self.forward(x), None)
        if head_ops._ok(x) and ln.weight.is_cuda and self.out_dim <= 256:
            out = head_ops.proj_ln_tanh(x, lin, ln)  # the whole projection: split-K MFMA GEMM + LN + tanh, 2 launches
            if out is not None:
                return out
            # shapes the fused GEMM does not cover: Linear (hipBLASLt), then LayerNorm + tanh as one launch
            return head_ops.ln_tanh(lin(x), ln.weight, ln.bias, ln.eps)
        return self.projection(x)


class Encoder(nn.Module):
    """modules.py:184-197: shared CNN -> head -> (detach) -> projection"""

    def __init__(self, shared_cnn, head_cnn, projection):
        super().__init__()
        self.shared_cnn, self.head_cnn, self.projection = shared_cnn, head_cnn, projection
        self.out_dim = projection.out_dim
        self.cuts = None  # data parallel: a list that collects the CNN features of gradient-carrying passes (SAC._cuts)

    def features(self, x):
        return self.head_cnn(self.shared_cnn(x))

    def forward(self, x, detach=False, features=None, tail=None):
        """features: a precomputed head_cnn(shared_cnn(x)) may be passed to avoid recomputing the CNN
        when two encoders share it and see the same input (actor + critic in update_actor_and_alpha).
        tail: see RLProjection.forward -- then the result is the pair (features, [features | tail] or None)."""
        x = self.features(x) if features is None else features
        if detach:
            x = x.detach()
        elif self.cuts is not None and x.requires_grad:
            self.cuts.append(x)  # the cut between conv stack and heads (flat.FlatAdam.backward_begin)
        return self.projection(x) if tail is None else self.projection(x, tail=tail)


class Actor(nn.Module):
    """modules.py:200-244"""

    def __init__(self, encoder, action_shape, hidden_dim, log_std_min, log_std_max):
        super().__init__()
        self.encoder = encoder
        self.log_std_min, self.log_std_max = log_std_min, log_std_max
        self.mlp = nn.Sequential(nn.Linear(self.encoder.out_dim, hidden_dim), nn.ReLU(),
                                 nn.Linear(hidden_dim, hidden_dim), nn.ReLU(),
                                 nn.Linear(hidden_dim, 2 * action_shape[0]))
        self.mlp.apply(weight_init)

    def forward(self, x, compute_pi=True, compute_log_pi=True, detach=False, features=None):
        x = self.encoder(x, detach, features=features)
        # the three Linears with bias + ReLU in the GEMM epilogues and fused gate / bias-gradient launches backward
        raw = head_ops.mlp3(x, self.mlp) if (head_ops._ok(x) and x.dim() == 2) else self.mlp(x)
        if head_ops._ok(raw) and (compute_pi or not compute_log_pi):
            # one launch forward / one backward for everything below (same operation order inside)
            noise = torch.randn_like(raw[:, :raw.shape[1] // 2]) if compute_pi else None
            return head_ops.gauss_head(raw, noise, self.log_std_min, self.log_std_max, compute_log_pi)
        mu, log_std = raw.chunk(2, dim=-1)
        log_std = torch.tanh(log_std)
        log_std = self.log_std_min + 0.5 * (self.log_std_max - self.log_std_min) * (log_std + 1)
        if compute_pi:
            std = log_std.exp()
            noise = torch.randn_like(mu)
            pi = mu + noise * std
        else:
            pi = None
        log_pi = gaussian_logprob(noise, log_std) if compute_log_pi else None
        mu, pi, log_pi = squash(mu, pi, log_pi)
        return mu, pi, log_pi, log_std


class QFunction(nn.Module):
    """modules.py:247-261"""

    def __init__(self, obs_dim, action_dim, hidden_dim):
        super().__init__()
        self.trunk = nn.Sequential(nn.Linear(obs_dim + action_dim, hidden_dim), nn.ReLU(),
                                   nn.Linear(hidden_dim, hidden_dim), nn.ReLU(), nn.Linear(hidden_dim, 1))
        self.apply(weight_init)

    def forward(self, obs, action):
        assert obs.size(0) == action.size(0)
        return self.trunk(torch.cat([obs, action], dim=1))


class Critic(nn.Module):
    """modules.py:264-276"""

    def __init__(self, encoder, action_shape, hidden_dim):
        super().__init__()
        self.encoder = encoder
        self.Q1 = QFunction(self.encoder.out_dim, action_shape[0], hidden_dim)
        self.Q2 = QFunction(self.encoder.out_dim, action_shape[0], hidden_dim)

    def _twin(self, h, action, freeze=False):
        if action.size(0) != h.size(0):  # fewer action rows than observations: the rows repeat (SVEA's [obs; obs_aug])
            assert h.size(0) % action.size(0) == 0
            action = action.repeat(h.size(0) // action.size(0), 1)
        if head_ops._ok(h, action):  # both trunks as one batched network (head_ops.twin_q); nn.Sequential otherwise
            return head_ops.twin_q(h, action, self.Q1, self.Q2, freeze=freeze)
        return self.Q1(h, action), self.Q2(h, action)

    def forward(self, x, action, detach=False, features=None, freeze_q=False):
        """freeze_q (not in the reference's signature): the Q trunks' weights enter as constants -- the actor step only
        needs the gradient with respect to the action.  action may have B rows for a k*B-row x: the rows repeat."""
        if head_ops._ok(action) and isinstance(self.encoder.projection, RLProjection):
            # the projection's last launch also writes the Q trunks' input rows [features | action]: no concat launches
            h, xcat = self.encoder(x, detach, features=features, tail=action)
            if xcat is not None:
                return head_ops.twin_q(xcat, None, self.Q1, self.Q2, freeze=freeze_q)
            return self._twin(h, action, freeze_q)
        return self._twin(self.encoder(x, detach, features=features), action, freeze_q)

    def forward_Q(self, x_encoded, action):
        return self._twin(x_encoded, action)
