"""Flat parameter groups: every optimiser group of an agent (critic, actor-only parameters)
lives in ONE contiguous f32 buffer with a same-shaped gradient buffer.  nn.Parameters are views
into it, so (a) the optimiser step is one fused HIP pass (augcl_adam_f32), (b) the target-network
EMA is one pass per tau segment (augcl_ema_f32) and (c) the data-parallel gradient exchange is a
single all-reduce of one bucket, with shared tensors (the CNN used by actor AND critic,
src/algorithms/sac.py:17-32) appearing exactly once.
"""
import ctypes

import torch

from . import _lib


def critic_order(named_params):
    """Order of a Critic's parameters inside its flat buffer: the encoder first (nn.Module order), then the twin Q
    trunks INTERLEAVED -- Q1.trunk.0.weight, Q2.trunk.0.weight, Q1.trunk.0.bias, Q2.trunk.0.bias, ... -- so that every
    (Q1.x, Q2.x) pair sits at a fixed distance and head_ops.twin_q can address both as one (2, ...) strided view with
    no copy.  The "Q" tensors stay one contiguous run (soft update with critic_tau, flat.FlatGroup.segment)."""
    named = list(named_params)
    rest = [(n, p) for n, p in named if not (n.startswith("Q1.") or n.startswith("Q2."))]
    q1 = [(n, p) for n, p in named if n.startswith("Q1.")]
    q2 = {n[3:]: (n, p) for n, p in named if n.startswith("Q2.")}
    twins = []
    for n, p in q1:
        twins.append((n, p))
        if n[3:] in q2:
            twins.append(q2.pop(n[3:]))
    return rest + twins + list(q2.values())


class FlatGroup:
    """Owns flat param / grad buffers for an ordered list of (name, nn.Parameter)."""

    def __init__(self, named_params, align=4):
        named_params = list(named_params)
        assert named_params, "empty parameter group"
        dev = named_params[0][1].device
        self.names, self.slices, self.shapes = [], {}, {}
        off = 0
        for n, p in named_params:
            assert p.dtype == torch.float32 and p.device == dev
            k = p.numel()
            self.names.append(n)
            self.slices[n] = (off, off + k)
            self.shapes[n] = tuple(p.shape)
            off = (off + k + align - 1) // align * align  # keep every tensor 16-byte aligned
        self.numel = off
        self.cut_index = 0  # see mark_cut()
        self.prepped = False  # a gather launch has advanced the optimiser's device-side step for the coming Adam launch
        self.param = torch.zeros(off, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(off, dtype=torch.float32, device=dev)
        self.params = []
        for n, p in named_params:
            a, b = self.slices[n]
            self.param[a:b].copy_(p.data.reshape(-1))
            p.data = self.param[a:b].view(self.shapes[n])
            p.grad = None
            self.params.append(p)

    def mark_cut(self, below):
        """Declare the leading run of tensors for which below(name) holds as the part of the network UNDER a cut (the
        conv stack under the CNN's output features): FlatAdam.backward_begin can then finish and all-reduce everything
        above the cut before the backward pass below it has run."""
        k = 0
        while k < len(self.names) and below(self.names[k]):
            k += 1
        assert not any(below(n) for n in self.names[k:]), "tensors under the cut must lead the group"
        self.cut_index = k

    def rebind(self):
        """Re-point parameters at the flat buffers (after something replaced .data/.grad)."""
        for n, p in zip(self.names, self.params):
            a, b = self.slices[n]
            if p.data.data_ptr() != self.param[a:b].data_ptr():
                self.param[a:b].copy_(p.data.reshape(-1))
                p.data = self.param[a:b].view(self.shapes[n])

    def gather_grads(self, step_prep=None):
        """Copy the parameters' autograd gradients into the flat bucket (one launch).  Returns the names of the
        parameters that have NO gradient (torch.optim.Adam skips those: no moment decay, no step)."""
        return self.scatter_into_bucket([p.grad for p in self.params], step_prep=step_prep)

    def scatter_into_bucket(self, grads, names=None, step_prep=None, zero_missing=False):
        """grads[i] -> the bucket slot of names[i] (default: all of the group's tensors, in order).  On the GPU one
        augcl_gather_f32 launch (the pointer table rides in the kernel arguments); step_prep = (device step state, lr,
        beta1, beta2) lets that launch also advance the optimiser's device-side step count -- `self.prepped` then tells
        FlatAdam that its Adam launch needs no preparation launch of its own."""
        g = self
        dsts, srcs, missing = [], [], []
        names = g.names if names is None else names
        if all(t is None for t in grads):
            if zero_missing and names:  # the caller all-reduces this slice of the bucket regardless: keep it well defined
                a0, b0 = g.slices[names[0]][0], g.slices[names[-1]][1]
                g.grad[a0:b0].zero_()
            return list(names)  # nothing to copy; the caller skips these tensors
        base = g.grad.data_ptr()
        on_gpu = g.grad.is_cuda
        for n, t in zip(names, grads):
            a, b = g.slices[n]
            if t is None:
                missing.append(n)
                g.grad[a:b].zero_()  # keeps the data-parallel all-reduce of the whole bucket well defined
                continue
            p = t.data_ptr()
            if p == base + 4 * a:
                continue  # already in the bucket
            if on_gpu and (not t.is_cuda or t.dtype != torch.float32):
                on_gpu = False
            elif not t.is_contiguous():
                t = t.contiguous()
                p = t.data_ptr()
            dsts.append((a, b, p))
            srcs.append(t)  # (also keeps a .contiguous() copy alive until the launch is queued)
        if not srcs:
            return missing
        if on_gpu:
            k = len(srcs)
            ptrs = (ctypes.c_void_p * k)(*[p for _, _, p in dsts])
            offs = (ctypes.c_int64 * k)(*[a for a, _, _ in dsts])
            lens = (ctypes.c_int64 * k)(*[b - a for a, b, _ in dsts])
            prep = step_prep if (step_prep is not None and not missing) else None
            _lib.call("augcl_gather_f32", ctypes.addressof(ptrs), ctypes.addressof(offs), ctypes.addressof(lens), k, base,
                      None if prep is None else prep[0], 0.0 if prep is None else prep[1], 0.0 if prep is None else prep[2],
                      0.0 if prep is None else prep[3], _lib.current_stream(g.grad.device))
            if prep is not None:
                g.prepped = True
        else:
            torch._foreach_copy_([g.grad[a:b].view(t.shape) for (a, b, _), t in zip(dsts, srcs)], srcs)
        return missing

    def segment(self, prefix):
        """[begin, end) of the contiguous run of tensors whose name starts with prefix."""
        idx = [i for i, n in enumerate(self.names) if n.startswith(prefix)]
        assert idx and idx == list(range(idx[0], idx[-1] + 1)), f"{prefix}: not contiguous"
        a = self.slices[self.names[idx[0]]][0]
        b = self.slices[self.names[idx[-1]]][1] if idx[-1] == len(self.names) - 1 else self.slices[self.names[idx[-1] + 1]][0]
        return a, (self.numel if idx[-1] == len(self.names) - 1 else b)


class FlatAdam:
    """torch.optim.Adam(params, lr, betas=(beta1, 0.999)) over a FlatGroup, one HIP launch per step.

    Gradients: zero_grad() drops the parameters' .grad (set_to_none), so autograd hands every parameter a fresh
    gradient tensor without an accumulate kernel; step() gathers them into the flat gradient bucket with one
    launch (augcl_gather_f32; the bucket is what the data-parallel all-reduce and the fused Adam kernel consume).
    Pre-assigned .grad views would cost one `grad += new` launch per parameter per backward (~40 per step)."""

    def __init__(self, group, lr, betas=(0.9, 0.999), eps=1e-8, reducer=None):
        self.group, self.lr, self.betas, self.eps = group, float(lr), (float(betas[0]), float(betas[1])), float(eps)
        self.exp_avg = torch.zeros_like(group.param)
        self.exp_avg_sq = torch.zeros_like(group.param)
        self._step_count = 0
        self._tensor_steps = None  # per-tensor step counts, only once some parameter has missed a gradient
        self.step_state = None  # device-side {int64 step, 2 floats}: set by make_capturable()
        self.reducer = reducer  # data-parallel gradient exchange (dist.GradReducer) or None
        self._unit = {}  # the backward pass's seed gradient d(loss)/d(loss) = 1, kept instead of one fill launch per step

    @property
    def step_count(self):
        if self.step_state is not None:
            return int(self.step_state[0].item())
        return self._step_count

    @step_count.setter
    def step_count(self, v):
        self._step_count = int(v)
        self._tensor_steps = None
        if self.step_state is not None:
            self.step_state[0] = int(v)

    def make_capturable(self):
        """Keep the step count on the device (augcl_adam_f32_devstep) so that step() can be captured in a
        hipGraph and replayed: no host scalar is baked into the launch."""
        if self.step_state is None:
            st = torch.zeros(2, dtype=torch.int64, device=self.group.param.device)
            st[0] = self._step_count
            self.step_state = st

    def zero_grad(self, set_to_none=True):
        for p in self.group.params:
            p.grad = None

    def _step_prep(self):
        """What a gather launch needs to advance the device-side step on this optimiser's behalf (None: host-side steps,
        or per-tensor step counts are in use)."""
        if self.step_state is None or self._tensor_steps is not None:
            return None
        return (self.step_state.data_ptr(), self.lr, self.betas[0], self.betas[1])

    def backward_step(self, loss, cut=None):
        """zero_grad(); loss.backward(); step() of the reference in one go: gradients are taken ONLY with respect to
        this group's parameters (torch.autograd.grad prunes every other weight-gradient branch, e.g. the critic's
        weights inside the actor loss, sac.py:119-133) and land in the flat bucket by one multi-tensor copy."""
        self.backward_finish(self.backward_begin(loss, cut))

    def backward_begin(self, loss, cut=None):
        """The backward half of backward_step: gradients into the bucket, the data-parallel exchange STARTED.
        cut = the activations under which the group's leading `cut_index` tensors sit (the CNN's output features): with
        several ranks the backward pass then runs in two pieces -- everything above the cut first, whose slice of the
        bucket (the projection and the Q / policy heads: 97 % of the bytes) is all-reduced on the side while the conv
        stack's backward runs; the small conv slice follows.  Returns what backward_finish() needs."""
        g, r = self.group, self.reducer
        key = (tuple(loss.shape), loss.dtype, loss.device)
        if key not in self._unit:
            self._unit[key] = torch.ones_like(loss, requires_grad=False)
        one = self._unit[key]
        cut = [t for t in (cut or []) if t.requires_grad]
        overlap = (r is not None and r.world > 1 and cut and 0 < g.cut_index < len(g.params)
                   and not (g.param.is_cuda and torch.cuda.is_current_stream_capturing()))
        if not overlap:
            grads = torch.autograd.grad(loss, g.params, grad_outputs=one, allow_unused=True)
            return g.scatter_into_bucket(grads, step_prep=self._step_prep()), []
        k = g.cut_index
        split = g.slices[g.names[k]][0]
        grads = torch.autograd.grad(loss, g.params[k:] + cut, grad_outputs=one, allow_unused=True)
        # (the device-side step is advanced by the LAST scatter, and only if nothing was missing above the cut either: a
        # partial step must not find the count already moved)
        missing = g.scatter_into_bucket(grads[:len(g.params) - k], g.names[k:], zero_missing=True)
        pending = [r.allreduce_sum_async(g.grad[split:])]
        d_cut = grads[len(g.params) - k:]
        live = [(t, d) for t, d in zip(cut, d_cut) if d is not None]
        if live:
            below = torch.autograd.grad([t for t, _ in live], g.params[:k], grad_outputs=[d for _, d in live], allow_unused=True)
        else:
            below = [None] * k
        missing += g.scatter_into_bucket(below, g.names[:k], step_prep=None if missing else self._step_prep(), zero_missing=True)
        pending.append(r.allreduce_sum_async(g.grad[:split]))
        return missing, pending

    def backward_finish(self, begun):
        missing, pending = begun
        for h in pending:
            if h is not None:
                h.wait()
        self._apply(missing, reduced=bool(pending))

    def step(self):
        self._apply(self.group.gather_grads(self._step_prep()))

    def tensor_step(self, name):
        """Adam step count of one parameter (torch.optim.Adam keeps it per parameter)."""
        return self.step_count if self._tensor_steps is None else self._tensor_steps[name]

    def _apply_partial(self, missing, scale):
        """Some parameters got no gradient: like torch.optim.Adam, leave those alone (moments, step count, value) and
        step the others -- one launch per run of neighbouring tensors that share a step count."""
        g = self.group
        if len(missing) == len(g.names):
            return  # nothing received a gradient (AugCL's orphan weak optimiser before the hand-off, augcl.py:13-14): no-op
        if self.step_state is not None:
            if missing:
                raise _lib.AugclError("FlatAdam: parameters without a gradient cannot be skipped inside a captured hipGraph "
                                      f"(no gradient for {missing[:3]}...)")
            if len(set(self._tensor_steps.values())) != 1:
                # tensors that lag behind (partial eager steps before make_capturable) need their own bias correction; the
                # captured launch has ONE device-side count for the whole group
                raise _lib.AugclError("FlatAdam: per-tensor step counts differ "
                                      f"({sorted(set(self._tensor_steps.values()))}); a captured hipGraph steps every tensor "
                                      "with one count -- bring the group to a common count eagerly first")
            self._tensor_steps = None  # equal per-tensor counts left over from a resume, nothing missing now: the plain path
            return self._apply((), reduced=True, scale=scale)
        if self._tensor_steps is None:
            self._tensor_steps = {n: self._step_count for n in g.names}
        skip = set(missing)
        runs = []  # (begin, end, step)
        for i, n in enumerate(g.names):
            if n in skip:
                continue
            self._tensor_steps[n] += 1
            a = g.slices[n][0]
            b = g.slices[g.names[i + 1]][0] if i + 1 < len(g.names) else g.numel  # include the alignment padding
            if runs and runs[-1][1] == a and runs[-1][2] == self._tensor_steps[n]:
                runs[-1] = (runs[-1][0], b, runs[-1][2])
            else:
                runs.append((a, b, self._tensor_steps[n]))
        self._step_count = max(self._tensor_steps.values())
        for a, b, k in runs:
            self._launch_adam(a, b, k, scale)
        if len(set(self._tensor_steps.values())) == 1:
            self._tensor_steps = None  # every tensor is at the same count again: back to one launch per step

    def _apply(self, missing=(), reduced=False, scale=None):
        g = self.group
        if len(missing) == len(g.names):
            return  # no tensor of the group received a gradient (the same on every rank): nothing to exchange or step
        if scale is None:
            scale = 1.0
            if self.reducer is not None:  # reduced: backward_begin already exchanged the bucket (in two overlapped pieces)
                scale = 1.0 / self.reducer.world if reduced else self.reducer.allreduce_sum_(g.grad)  # 1/world_size
        if missing or self._tensor_steps is not None:
            return self._apply_partial(list(missing), scale)
        if self.step_state is not None:
            advance, g.prepped = (0 if g.prepped else 1), False
            _lib.call("augcl_adam_f32_devstep", g.param.data_ptr(), g.grad.data_ptr(), self.exp_avg.data_ptr(),
                      self.exp_avg_sq.data_ptr(), g.numel, self.lr, self.betas[0], self.betas[1], self.eps,
                      self.step_state.data_ptr(), advance, scale, _lib.current_stream(g.param.device))
            return
        self._step_count += 1
        self._launch_adam(0, g.numel, self._step_count, scale)

    def _launch_adam(self, a, b, step, scale):
        """One fused Adam launch over elements [a, b) of the flat buffers (the only device-specific line of the class)."""
        g = self.group
        _lib.call("augcl_adam_f32", g.param.data_ptr() + 4 * a, g.grad.data_ptr() + 4 * a, self.exp_avg.data_ptr() + 4 * a,
                  self.exp_avg_sq.data_ptr() + 4 * a, b - a, self.lr, self.betas[0], self.betas[1], self.eps, step, scale,
                  _lib.current_stream(g.param.device))

    def state_dict(self):
        return {"step": self.step_count, "exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq, "lr": self.lr,
                "betas": self.betas, "eps": self.eps}

    def load_state_dict(self, sd):
        self.step_count = int(sd["step"])
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])


class ScalarAdam:
    """torch.optim.Adam([log_alpha], lr, betas) for SAC's float64 temperature scalar (src/algorithms/sac.py:54-56) with
    its state on the device: `fused_step` computes the temperature loss, its gradient and the Adam step in ONE launch
    (augcl_alpha_step_f64; the ATen chain is ~25 launches on a single scalar) and can be captured in a hipGraph as is.
    state = float64 [exp_avg, exp_avg_sq, step, grad]."""

    def __init__(self, param, lr, betas=(0.9, 0.999), eps=1e-8):
        assert param.dtype == torch.float64 and param.numel() == 1
        self.param, self.lr, self.betas, self.eps = param, float(lr), (float(betas[0]), float(betas[1])), float(eps)
        self.state = torch.zeros(4, dtype=torch.float64, device=param.device)
        self._out = torch.zeros(2, dtype=torch.float32, device=param.device)

    def fused_step(self, log_pi, target_entropy, reducer=None):
        """Returns (alpha_loss, alpha) as 0-dim float32 views (valid until the next call)."""
        lp = log_pi.detach().contiguous()
        args = (self.param.data_ptr(), lp.data_ptr(), lp.shape[0], float(target_entropy), self.lr, self.betas[0],
                self.betas[1], self.eps, self.state.data_ptr(), self._out.data_ptr())
        st = _lib.current_stream(self.param.device)
        if reducer is not None and reducer.world > 1:  # every rank's gradient of ITS shard's mean, averaged, then the step
            _lib.call("augcl_alpha_step_f64", *args, 1, st)
            reducer.allreduce_mean_(self.state[3:4])
            _lib.call("augcl_alpha_step_f64", *args, 2, st)
        else:
            _lib.call("augcl_alpha_step_f64", *args, 3, st)
        return self._out[0], self._out[1]

    def zero_grad(self, set_to_none=True):
        self.param.grad = None

    def step(self):
        """Plain step from param.grad (the non-fused path: the same arithmetic through the same kernel)."""
        self.state[3] = self.param.grad.detach().to(torch.float64).reshape(())
        _lib.call("augcl_alpha_step_f64", self.param.data_ptr(), None, 1, 0.0, self.lr, self.betas[0], self.betas[1], self.eps,
                  self.state.data_ptr(), None, 2, _lib.current_stream(self.param.device))

    def state_dict(self):
        """torch.optim.Adam.state_dict() layout."""
        st = {}
        if float(self.state[2]) > 0:
            st[0] = {"step": self.state[2].detach().cpu().float().reshape(()), "exp_avg": self.state[0].detach().cpu().reshape(()),
                     "exp_avg_sq": self.state[1].detach().cpu().reshape(())}
        return {"state": st, "param_groups": [{"lr": self.lr, "betas": self.betas, "eps": self.eps, "weight_decay": 0,
                                               "amsgrad": False, "params": [0]}]}

    def load_state_dict(self, sd):
        self.state.zero_()
        for st in sd.get("state", {}).values():
            self.state[0] = float(torch.as_tensor(st["exp_avg"]))
            self.state[1] = float(torch.as_tensor(st["exp_avg_sq"]))
            self.state[2] = float(torch.as_tensor(st["step"]))


def ema2_(target_flat, src_flat, seg0, seg1):
    """Two (begin, end, tau) segments of the same flat buffers in one launch (SAC.soft_update_critic_target)."""
    (a0, b0, t0), (a1, b1, t1) = seg0, seg1
    n0, n1 = max(0, b0 - a0), max(0, b1 - a1)
    if n0 == 0 and n1 == 0:
        return
    tp, sp = target_flat.data_ptr(), src_flat.data_ptr()
    _lib.call("augcl_ema2_f32", tp + 4 * a0, sp + 4 * a0, n0, float(t0), tp + 4 * a1, sp + 4 * a1, n1, float(t1),
              _lib.current_stream(target_flat.device))


def ema_(target_flat, src_flat, begin, end, tau):
    """target[begin:end] <- tau*src + (1-tau)*target (utils.soft_update_params on a flat segment)."""
    if end <= begin:
        return
    _lib.call("augcl_ema_f32", target_flat.data_ptr() + 4 * begin, src_flat.data_ptr() + 4 * begin, end - begin,
              float(tau), _lib.current_stream(target_flat.device))
