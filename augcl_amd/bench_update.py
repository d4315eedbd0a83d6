"""SAC grad-steps/s leg of bench.py: full SVEA update() calls (config 2: walker_walk, A=6, random_shift +
random_conv, B=256 per GPU, f32) on a pre-staged synthetic replay batch -- critic every step, actor+alpha
and target EMA every 2nd -- timed like the headline metric (barrier + synchronize, max over ranks)."""
import time

import torch


class _SyntheticReplay:
    """Device-resident ring of uint8->f32 batches in the 5-tuple layout of ReplayBuffer.sample()
    (src/utils.py:272-280)."""

    def __init__(self, dev, B, A, ring=4, seed=0):
        g = torch.Generator(device=dev)
        g.manual_seed(seed)
        self.items = []
        for _ in range(ring):
            obs = torch.randint(0, 256, (B, 9, 84, 84), device=dev, generator=g, dtype=torch.uint8).float()
            nobs = torch.randint(0, 256, (B, 9, 84, 84), device=dev, generator=g, dtype=torch.uint8).float()
            act = torch.rand((B, A), device=dev, generator=g) * 2 - 1
            rew = torch.rand((B, 1), device=dev, generator=g) * 4
            self.items.append((obs, act, rew, nobs, torch.ones((B, 1), device=dev)))
        self.i = 0

    def sample(self, n=None, idxs=None):
        self.i += 1
        return self.items[self.i % len(self.items)]


def _ring_replay(dev, B, A, transitions=3000, seed=0):
    """The product's ReplayBuffer (augcl_amd/replay.py: uint8 frame ring in HBM, frames stored once) filled with `transitions`
    steps of synthetic 3-frame-stacked episodes -- update() then samples it as the training loop does (SVEA: sample + shift +
    conv in one pass over the ring, next_obs gathered into the graph's static input)."""
    import numpy as np
    from .replay import LazyFrames, ReplayBuffer
    rs = np.random.RandomState(seed)
    buf = ReplayBuffer((9, 84, 84), (A,), capacity=transitions + 64, batch_size=B, device=dev)
    frames = [rs.randint(0, 256, (3, 84, 84)).astype(np.uint8) for _ in range(3)]
    obs = LazyFrames(list(frames))
    for t in range(transitions):
        frames = frames[1:] + [rs.randint(0, 256, (3, 84, 84)).astype(np.uint8)]
        nxt = LazyFrames(list(frames))
        done = (t % 500) == 499
        buf.add(obs, rs.uniform(-1, 1, A).astype(np.float32), float(rs.uniform(0, 4)), nxt, done)
        if done:
            frames = [rs.randint(0, 256, (3, 84, 84)).astype(np.uint8) for _ in range(3)]
            obs = LazyFrames(list(frames))
        else:
            obs = nxt
    return buf


class _ReferenceLikeLogger:
    """What the reference's Logger.log does with a value on the hot path (src/logger.py:138-141): a tensor is turned into a host
    float with .item() -- one device sync per logged scalar -- then accumulated per key (its AverageMeter).  The agent here hands
    over host floats (one packed copy per update()), so the .item() branch is not taken; it is kept so that the timing includes
    it if an algorithm ever logs a tensor again."""

    def __init__(self):
        self.sums, self.counts, self.item_calls = {}, {}, 0

    def log(self, key, value, step, n=1):
        if type(value) is torch.Tensor:
            value = value.item()
            self.item_calls += 1
        self.sums[key] = self.sums.get(key, 0.0) + value * n
        self.counts[key] = self.counts.get(key, 0) + n


def _densify_conv_weights(agent, seed=1234):
    """Mid-training-like conv weights: the reference initialises every conv filter delta-orthogonal (src/algorithms/modules.py:
    52-66: only the centre tap of each 3x3 filter is non-zero, biases 0), which is what a FRESH agent times; a trained network's
    filters are dense.  He-scaled normal weights (std = sqrt(2 / fan_in)) and small biases, written in place into every conv stack
    of the agent (actor / critic share one; the target has its own copy; AugCL's weak critic too)."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    done = set()
    for m in agent._cnn_modules():
        for mod in m.modules():
            if isinstance(mod, torch.nn.Conv2d) and id(mod.weight) not in done:
                done.add(id(mod.weight))
                fan_in = mod.weight.shape[1] * 9
                with torch.no_grad():
                    mod.weight.copy_((torch.randn(mod.weight.shape, generator=g) * (2.0 / fan_in) ** 0.5).to(mod.weight.device))
                    mod.bias.copy_((torch.randn(mod.bias.shape, generator=g) * 0.05).to(mod.bias.device))
    return len(done)


class _PowerSampler:
    """Shader clock / board power while a leg runs (a background thread polling `rocm-smi` as a child process: reads sysfs, does
    not touch the GPU queues).  Best effort: None where the tool or the fields are missing."""

    def __init__(self, period=0.25):
        import threading
        self.period, self.rows, self._stop = period, [], threading.Event()
        self._t = threading.Thread(target=self._loop, daemon=True)

    def _loop(self):
        import json
        import subprocess
        while not self._stop.is_set():
            try:
                out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=5).stdout
                card = next(iter(json.loads(out).values()))
                sclk = next((v for k, v in card.items() if k.lower().startswith("sclk clock speed")), None)
                power = next((v for k, v in card.items() if "power" in k.lower() and "(w)" in k.lower()), None)
                mhz = float(str(sclk).strip("()").lower().replace("mhz", "")) if sclk is not None else None
                self.rows.append((mhz, float(power) if power is not None else None))
            except Exception:  # noqa: BLE001 -- a sampler must never take the bench down
                pass
            self._stop.wait(self.period)

    def __enter__(self):
        self._t.start()
        return self

    def __exit__(self, *a):
        self._stop.set()
        self._t.join(timeout=10)

    def summary(self):
        mhz = sorted(r[0] for r in self.rows if r[0])
        w = sorted(r[1] for r in self.rows if r[1])
        med = lambda v: v[len(v) // 2] if v else None  # noqa: E731
        return {"samples": len(self.rows), "sclk_MHz_median": med(mhz), "sclk_MHz_min": mhz[0] if mhz else None,
                "power_W_median": med(w), "power_W_max": w[-1] if w else None}


def run(dev, dist, world, rank, steps=20, warmup=4, B=256, A=6, algo="svea", data_aug="conv", precision="f32", graph=False,
        min_s=0.5, replay="synthetic", weights="init", logger=False, power=False):
    """weights: "init" = a fresh agent (the reference's delta-orthogonal conv filters), "dense" = He-scaled dense filters
    (_densify_conv_weights); logger: time update(buf, L, step) with a reference-like Logger instead of L = None; power: sample
    sclk / board power during the timed region.  algo "augcl" = the strong phase after the curriculum hand-off
    (src/algorithms/augcl.py:16-27: a non_naive_rad agent's critic aliased as the weak critic)."""
    from . import augmentations
    from . import dist as apdist
    from . import encoder_ops
    from . import utils as au
    from .algorithms import make_agent

    encoder_ops.set_precision(precision)
    hp = dict(discount=0.99, critic_tau=0.01, encoder_tau=0.05, actor_update_freq=2, critic_target_update_freq=2,
              num_shared_layers=11, num_head_layers=0, num_filters=32, projection_dim=100, hidden_dim=1024,
              actor_log_std_min=-10.0, actor_log_std_max=2.0, init_temperature=0.1, actor_lr=1e-3, actor_beta=0.9,
              critic_lr=1e-3, critic_beta=0.9, alpha_lr=1e-4, alpha_beta=0.5, svea_alpha=0.5, svea_beta=0.5, drq_k=1,
              drq_m=1)
    if dist is not None and apdist.default_reducer() is None:
        apdist.set_default_reducer(apdist.GradReducer())
    # replicas are made identical by the parameter broadcast at construction; everything random AFTER that differs per
    # rank (actor noise from torch's generator; augmentation parameters are keyed by the global sample index)
    apdist.seed_all(0, rank)
    if "overlay" in data_aug:
        augmentations.set_overlay_source(augmentations.SyntheticOverlaySource(seed=7))
    agent = make_agent((9, 84, 84), (A,), au.dotdict(algorithm=algo, data_aug=data_aug, aug_params=None, **hp))
    if algo == "augcl":
        prev = make_agent((9, 84, 84), (A,), au.dotdict(algorithm="non_naive_rad", data_aug="shift", aug_params=None, **hp))
        agent.load_pretrained_agent(prev)
    if weights == "dense":
        _densify_conv_weights(agent)
    elif weights != "init":
        raise ValueError("weights must be 'init' or 'dense'")
    if graph:  # replay the learning part of update() as captured hipGraphs (SAC.enable_hip_graphs)
        agent.enable_hip_graphs()
        warmup = max(warmup, 8)  # 3 eager updates, then one capture per (actor step?, target step?) variant
    buf = _SyntheticReplay(dev, B, A, seed=100 + rank) if replay == "synthetic" else _ring_replay(dev, B, A, seed=100 + rank)
    L = _ReferenceLikeLogger() if logger else None  # (default: no logging inside the timed region)
    for s in range(warmup):
        agent.update(buf, L, s)
    torch.cuda.synchronize()
    walls, step0 = [], warmup
    sampler = _PowerSampler() if (power and rank == 0) else None
    if sampler is not None:
        sampler.__enter__()
    while True:  # blocks of exactly `steps` updates between barrier + synchronize, until min_s seconds are timed
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for s in range(step0, step0 + steps):
            agent.update(buf, L, s)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([wall], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            wall = float(t.item())
        walls.append(wall)
        step0 += steps
        if sum(walls) >= min_s or len(walls) >= 200:
            break
    if sampler is not None:
        sampler.__exit__()
    wall = sorted(walls)[len(walls) // 2]
    encoder_ops.set_precision("f32")
    flops = 432e9 * (B / 256)  # ALGORITHMIC: ~1.69 GFLOP per sample per average step of the reference's update
    flops_exec = executed_flops_per_update(B, A)  # what the build actually runs
    return {"grad_steps_per_s": steps / wall, "ms_per_update": wall / steps * 1e3, "samples_per_s": world * B * steps / wall,
            "ms_per_update_min": min(walls) / steps * 1e3, "ms_per_update_max": max(walls) / steps * 1e3,
            "repeats": len(walls), "timed_s": sum(walls),
            "algo": f"{algo}+{data_aug}", "batch_per_gpu": B, "steps": steps,
            "dtype": "f32" if precision.startswith("f32") else "bf16 conv stack (f32 accumulate, f32 heads/optimizer)",
            "approx_TFLOP/s_per_gpu": flops * steps / wall / 1e12, "algorithmic_TFLOP/s": flops * steps / wall / 1e12,
            "executed_TFLOP/s": flops_exec["total"] * steps / wall / 1e12, "executed_GFLOP_per_update": flops_exec["total"] / 1e9,
            "executed_conv32_GFLOP_per_update": flops_exec["conv32"] / 1e9, "encoder_backend": "hip",
            "hip_graph": bool(graph), "world": world, "conv_weights": weights, "with_logger": bool(logger),
            "logger_item_calls": (L.item_calls if L is not None else None),
            "clock_power": sampler.summary() if sampler is not None else None}


def executed_flops_per_update(B, A, n_layers=11, hidden=1024, proj=100):
    """Multiply-add flops (x2) one average SVEA update() of THIS build executes at batch B: the reference's count (SURVEY 8d:
    8 encoder-forward equivalents per critic step + 2 per actor step) minus what the build does not run -- the actor step
    reuses ONE CNN pass for the actor and the critic (the reference runs two), and nothing differentiates with respect to
    the pixels (no first-layer backward-data).  Returns {"total", "conv32" (the 32 -> 32 layers only), "first", "heads"}."""
    hw = [41 - 2 * i for i in range(n_layers)]                      # output size of layer i
    first = 2.0 * 81 * 32 * hw[0] * hw[0]                           # 9 -> 32, 3x3, stride 2: 84 -> 41
    c32 = sum(2.0 * 288 * 32 * h * h for h in hw[1:])               # ten 32 -> 32 layers, 39 .. 21
    feat = 32 * hw[-1] * hw[-1]
    head_fwd = 2.0 * (feat * proj + (proj + A) * hidden * 2 + hidden * hidden * 2 + hidden * 2)  # projection + twin Q
    actor_fwd = 2.0 * (feat * proj + proj * hidden + hidden * hidden + hidden * 2 * A)
    # critic step: 2 no-grad passes over next_obs (actor's + target's CNN), forward over 2B (obs, obs_aug), backward over
    # 2B = backward-data + backward-weights of the 32 -> 32 layers, backward-weights only of the first layer
    conv32 = B * c32 * (2 + 2 + 2 * 2)
    firstl = B * first * (2 + 2 + 2 * 1)
    heads = B * (2 * (head_fwd + actor_fwd) + 2 * 3 * head_fwd)
    # actor + alpha step every second update: ONE CNN pass (no gradient), actor MLP fwd + bwd, critic heads fwd + bwd (dx only)
    conv32 += 0.5 * B * c32
    firstl += 0.5 * B * first
    heads += 0.5 * B * (3 * actor_fwd + 2 * head_fwd)
    return {"total": conv32 + firstl + heads, "conv32": conv32, "first": firstl, "heads": heads}


def action_latency(dev, A=6, calls=300):
    """Batch-1 select_action (src/algorithms/sac.py:73-78: one host frame stack in, one action out) in microseconds per
    call, eager and replayed from the captured hipGraph (SAC.enable_action_graphs)."""
    import numpy as np

    from . import utils as au
    from .algorithms import make_agent
    hp = dict(discount=0.99, critic_tau=0.01, encoder_tau=0.05, actor_update_freq=2, critic_target_update_freq=2,
              num_shared_layers=11, num_head_layers=0, num_filters=32, projection_dim=100, hidden_dim=1024,
              actor_log_std_min=-10.0, actor_log_std_max=2.0, init_temperature=0.1, actor_lr=1e-3, actor_beta=0.9,
              critic_lr=1e-3, critic_beta=0.9, alpha_lr=1e-4, alpha_beta=0.5, svea_alpha=0.5, svea_beta=0.5, drq_k=1, drq_m=1)
    agent = make_agent((9, 84, 84), (A,), au.dotdict(algorithm="sac", data_aug="identity", aug_params=None, **hp))
    obs = np.random.RandomState(0).randint(0, 256, (9, 84, 84)).astype(np.uint8)
    out = {}
    for mode in ("eager", "graph"):
        if mode == "graph":
            agent.enable_action_graphs()
        for _ in range(20):
            agent.select_action(obs)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(calls):
            agent.select_action(obs)
        out[mode] = (time.perf_counter() - t0) / calls * 1e6
    return out
