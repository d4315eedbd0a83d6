#!/usr/bin/env python3
"""bench.py -- headline benchmark of the hot path (BASELINE.json: "augmented obs/sec + SAC
grad-steps/sec, walker_walk 84x84x9 batch=256").

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the config-2 augmentation (SVEA's strong augmentation `random_conv`,
f32 -> f32 drop-in, src/augmentations.py:265-274) over one synthetic batch of B=256 stacked
84x84x9 observations per GPU, inputs already resident in HBM.  The batch ring is larger than the
256 MiB Infinity Cache so every step streams from HBM (as freshly sampled replay batches do).
`value` = observations/s over all ranks (weak scaling: every rank owns its own B=256 shard; the
augmentation path has no collective).  Extra keys report the other operators, the fused u8
front end and SAC grad-steps/s, each measured the same way.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s is the measured copy ceiling
OBS_ELEMS = 9 * 84 * 84
BYTES_PER_OBS = {  # algorithmic bytes per observation
    "random_conv_f32": 2 * OBS_ELEMS * 4 + 324,
    "random_shift_f32": 2 * OBS_ELEMS * 4,
    "random_shift_u8_f32": OBS_ELEMS * 5,
    "random_overlay_f32": 2 * OBS_ELEMS * 4 + 3 * 84 * 84 * 4,
    "color_jitter_f32": 2 * OBS_ELEMS * 4,
    "shift_conv_u8_f32": OBS_ELEMS * 9,  # u8 in, two f32 outputs
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--batch", type=int, default=256, help="observations per GPU per step")
    ap.add_argument("--ring", type=int, default=8, help="distinct batches cycled through (defeats the 256 MiB L3)")
    ap.add_argument("--no-extras", action="store_true", help="headline metric only")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU-baseline work (s)")
    return ap.parse_args()


def timed(fn, steps, warmup, dist, torch):
    """warmup, then time exactly `steps` calls: barrier + synchronize on both sides, wall clock;
    HIP events on the launch stream give the device-side average per launch."""
    t_ramp = time.perf_counter()  # untimed clock ramp: a cold GPU needs tens of ms of load to reach its
    while time.perf_counter() - t_ramp < 0.25:  # steady clock; W short warmup launches alone are ~1 ms
        for i in range(50):
            fn(i)
        torch.cuda.synchronize()
    for i in range(warmup):
        fn(i)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for i in range(steps):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    dev_ms = e0.elapsed_time(e1) / steps
    if dist is not None:
        t = torch.tensor([wall], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())
    return wall, dev_ms


def cpu_baseline(target_s):
    """The oracle (C restatement, OpenMP over frames) of the SAME op on this box's host cores."""
    import numpy as np

    import inputs as gi
    from oracle import aug_oracle as orc
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = orc.set_num_threads(min(avail, 16))  # a 1-GPU box's CPU share is 16 cores
    B = 256
    x = gi.obs_u8(0, B).astype(np.float32)
    w = gi.conv_weights(1, B)
    orc.random_conv_c(x[:32], w[:32])  # warm (thread pool, page faults)
    reps, t0 = 0, time.perf_counter()
    while True:
        orc.random_conv_c(x, w)
        reps += 1
        dt = time.perf_counter() - t0
        if dt >= target_s or reps >= 1000:
            break
    B = B * reps
    model = ""
    try:
        model = [ln.split(":", 1)[1].strip() for ln in open("/proc/cpuinfo") if ln.startswith("model name")][0]
    except Exception:
        pass
    return {"value": B / dt, "unit": "obs/s", "cores": cores, "kind": "port",
            "sample": f"oracle/aug_oracle.c random_conv f32 (same op, same B=256 batches), {B} obs of 9x84x84 in "
                      f"{dt:.1f} s, OpenMP {cores} threads on {model}"}


def graph_collectives_ok(dist, dev, torch):
    """Multi-rank pre-flight for the hipGraph leg: can this build capture an RCCL all-reduce and replay it?  Every
    rank captures (recording only -- nothing can block), the ranks agree on the outcome EAGERLY, and only then is the
    graph replayed, so a rank whose capture failed never leaves the others waiting inside a collective."""
    world = dist.get_world_size()
    t = torch.ones(4096, device=dev)
    dist.all_reduce(t)  # communicator up, outside any capture
    torch.cuda.synchronize()
    g, ok = torch.cuda.CUDAGraph(), 1
    try:
        with torch.cuda.graph(g):
            dist.all_reduce(t)
    except Exception as e:  # noqa: BLE001
        print(f"[bench] rank {dist.get_rank()}: all-reduce capture failed ({e!r}); update() will run eagerly", file=sys.stderr)
        ok = 0
    flag = torch.tensor([ok], device=dev, dtype=torch.int32)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if int(flag.item()) == 0:
        return False
    g.replay()
    torch.cuda.synchronize()
    return abs(float(t[0].item()) - float(world) * float(world)) < 1e-3


def main():
    args = parse()
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path is the only path (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or os.environ.get("AUGCL_FORCE_DIST") == "1":  # the override rehearses the RCCL path on one GPU
        import torch.distributed as dist
        # RCCL prints a version banner on STDOUT when its communicator comes up; stdout carries the one JSON line,
        # so point fd 1 at stderr until the first collective has run
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group("nccl", device_id=dev)
            dist.barrier()
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved, 1)
            os.close(saved)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    from augcl_amd import _lib, augmentations as aug
    _lib.load()
    B, R = args.batch, args.ring
    g = torch.Generator(device=dev)
    g.manual_seed(1000 + rank)
    # every input ring must exceed the 256 MiB Infinity Cache on its own: the outputs are written with non-temporal
    # stores and no longer evict the inputs (an 8-deep uint8 ring = 130 MB stayed cache-resident and read 35 % "faster")
    R8 = max(R, 40)
    ring_u8 = [torch.randint(0, 256, (B, 9, 84, 84), dtype=torch.uint8, device=dev, generator=g) for _ in range(R8)]
    ring_f32 = [t.float() for t in ring_u8[:R]]
    outs = [torch.empty((B, 9, 84, 84), dtype=torch.float32, device=dev) for _ in range(R)]
    outs2 = [torch.empty((B, 9, 84, 84), dtype=torch.float32, device=dev) for _ in range(2)]
    weights = torch.randn((B, 81), device=dev, generator=g)
    oy = torch.randint(0, 9, (B,), dtype=torch.int32, device=dev, generator=g)
    ox = torch.randint(0, 9, (B,), dtype=torch.int32, device=dev, generator=g)
    overlays = [torch.rand((B, 3, 84, 84), device=dev, generator=g) for _ in range(16)]  # 16 x 21.7 MB: not cache-resident
    stream = torch.cuda.current_stream(dev).cuda_stream
    call = _lib.call
    p = _lib.ptr

    def conv_step(i):
        k = i % R
        call("augcl_random_conv_f32", p(ring_f32[k]), p(outs[k]), B, 3, 84, 84, p(weights), 0, 0, 0, stream)

    wall, dev_ms = timed(conv_step, args.steps, args.warmup, dist, torch)
    value = world * B * args.steps / wall
    ach = BYTES_PER_OBS["random_conv_f32"] * B / (dev_ms * 1e-3) / 1e9
    res = {
        "metric": "augmented obs/sec + SAC grad-steps/sec, walker_walk 84x84x9 batch=256",
        "value": value, "unit": "obs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": wall / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "walker_walk SVEA strong augmentation: random_conv f32->f32 drop-in on (256,9,84,84) "
                               "per GPU, explicit per-sample 3x3x3x3 weights, ring of %d batches (> L3)" % R,
                   "batch_per_gpu": B, "obs_shape": [9, 84, 84], "parallelism": f"dp{world} (batch-sharded, no collective)"},
        "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                     "traffic": None, "kernel": "k_random_conv<84,84,12>", "bytes_per_launch": BYTES_PER_OBS["random_conv_f32"] * B,
                     "avg_launch_ms": dev_ms},
    }
    prof = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(prof):
        try:
            t = json.load(open(prof)).get("k_random_conv", {})
            if t.get("batch") == B:  # measured offline with rocprofv3 --pmc (profiles/, tools/summarize_profiles.py)
                res["roofline"]["traffic"] = t.get("hbm_bytes_per_launch")
                res["roofline"]["traffic_source"] = "profiles/pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)"
        except Exception:
            pass

    if not args.no_extras:
        extras = {}
        n2 = max(20, args.steps // 4)

        def rate(name, fn):
            w_, d_ = timed(fn, n2, 10, dist, torch)
            a_ = BYTES_PER_OBS[name] * B / (d_ * 1e-3) / 1e9
            extras[name] = {"obs_per_s": world * B * n2 / w_, "GB/s": a_, "frac": a_ / HBM_PEAK_GBS, "avg_launch_ms": d_}

        rate("random_shift_f32", lambda i: call("augcl_random_shift_f32", p(ring_f32[i % R]), p(outs[i % R]), B, 9, 84, 84, 4,
                                                p(oy), p(ox), 0, 0, 0, stream))
        rate("random_shift_u8_f32", lambda i: call("augcl_random_shift_u8_f32", p(ring_u8[i % R8]), p(outs[i % R]), B, 9, 84,
                                                   84, 4, p(oy), p(ox), 0, 0, 0, stream))
        rate("random_overlay_f32", lambda i: call("augcl_random_overlay_f32", p(ring_f32[i % R]), p(overlays[i % 16]), p(outs[i % R]),
                                                  B, 3, 84, 84, 0.5, stream))
        rate("shift_conv_u8_f32", lambda i: call("augcl_shift_conv_u8_f32", p(ring_u8[i % R8]), p(outs[i % R]),
                                                 p(outs2[i % 2]), B, 3, 84, 84, 4, p(oy), p(ox), p(weights), 0, 0, 0, stream))
        order = (_lib.ctypes.c_int32 * 4)(1, 2, 3, 0)
        order_p = _lib.ctypes.cast(order, _lib.ctypes.c_void_p)
        rate("color_jitter_f32", lambda i: call("augcl_color_jitter_f32", p(outs[i % R]), B, 3, 84, 84, None, 0, order_p,
                                                1.2, 0.8, 0.1, 1.0, 255.0, stream))
        # device-resident replay sampling: gather obs + next_obs (f32) of B transitions from a
        # 20k-frame uint8 ring; algorithmic bytes per sampled transition = 2 x 63504 x (1 + 4)
        from augcl_amd.replay import ReplayBuffer
        import numpy as np
        rb = ReplayBuffer((9, 84, 84), (6,), capacity=20000, batch_size=B, device=dev)
        rb.frames.random_(0, 256, generator=g)
        rb.frame_seq = rb.n_slots
        rb.obs_seq[:] = (np.arange(20000)[:, None] + np.arange(3)[None, :]) % rb.n_slots
        rb.next_seq[:] = (rb.obs_seq + 1) % rb.n_slots
        rb._sync_tables()
        rb.idx, rb.full = 0, True
        BYTES_PER_OBS["replay_sample_f32"] = 2 * OBS_ELEMS * 5
        rate("replay_sample_f32", lambda i: rb.sample())
        res["extras"] = extras
        try:
            from augcl_amd import bench_update
            n_upd = max(10, args.steps // 20)
            # update() replayed from captured hipGraphs (SAC.enable_hip_graphs); the eager figure beside it.
            # With several ranks the graphs contain the gradient all-reduce: checked by a pre-flight first.
            use_graph = dist is None or graph_collectives_ok(dist, dev, torch)
            def leg(precision):
                """Both execution modes of update(); the faster one is the reported figure, the other sits beside it
                (the f32 path is GPU-bound, so replaying graphs buys it nothing; the bf16 mode is launch-bound)."""
                runs = [bench_update.run(dev, dist, world, rank, steps=n_upd, precision=precision, graph=g)
                        for g in ((True, False) if use_graph else (False,))]
                best = min(runs, key=lambda r: r["ms_per_update"])
                for r in runs:
                    if r is not best:
                        best["hip_graph_ms_per_update" if r["hip_graph"] else "eager_ms_per_update"] = r["ms_per_update"]
                return best

            res["sac"] = leg("f32")  # the parity path
            res["sac"]["bf16_mode"] = leg("bf16")
        except ImportError:
            res["sac"] = None

    if rank == 0 and not args.no_cpu_baseline and world == 1:
        res["cpu_baseline"] = cpu_baseline(args.cpu_seconds)
    elif rank == 0:
        res["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(res))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
