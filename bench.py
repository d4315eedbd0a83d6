#!/usr/bin/env python3
"""bench.py -- headline benchmark of the hot path (BASELINE.json: "augmented obs/sec + SAC
grad-steps/sec, walker_walk 84x84x9 batch=256").

  python bench.py --gpus N --steps K --warmup W
      N > 1 without a launcher: bench.py starts N ranks of itself (one process per GPU, RANK / LOCAL_RANK /
      WORLD_SIZE / MASTER_* set, rendezvous on 127.0.0.1) BEFORE anything touches the GPU, relays rank 0's
      JSON line and exits with the ranks' status.
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (the driver's form)

One "step" = one pass of the config-2 augmentation (SVEA's strong augmentation `random_conv`,
f32 -> f32 drop-in, src/augmentations.py:265-274) over one synthetic batch of B=256 stacked
84x84x9 observations per GPU, inputs already resident in HBM.  The batch ring is larger than the
256 MiB Infinity Cache so every step streams from HBM (as freshly sampled replay batches do).
After W warm-up steps a block of EXACTLY K steps is timed between barrier + synchronize on both
sides (max over ranks); the block is repeated until >= 0.5 s have been timed in total and the MEDIAN block
is reported (`repeats`, `timed_s`, min / max beside it) -- K = 20 alone would be a 0.6 ms window.
`value` = observations/s over all ranks (weak scaling: every rank owns its own B=256 shard; the
augmentation path has no collective).  Extra keys report the other operators, the fused u8
front end and SAC grad-steps/s, each measured the same way.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s is the measured copy ceiling
MFMA_F32_PEAK_TFLOPS = 157.3   # dense f32 MFMA (MI355X_MICROARCH.md)
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA
MIN_TIMED_S = 0.5
OBS_ELEMS = 9 * 84 * 84
BYTES_PER_OBS = {  # algorithmic bytes per observation
    "random_conv_f32": 2 * OBS_ELEMS * 4 + 324,
    "random_shift_f32": 2 * OBS_ELEMS * 4,
    "random_shift_u8_f32": OBS_ELEMS * 5,
    "random_overlay_f32": 2 * OBS_ELEMS * 4 + 3 * 84 * 84 * 4,
    "color_jitter_f32": 2 * OBS_ELEMS * 4,
    "shift_conv_u8_f32": OBS_ELEMS * 9,  # u8 in, two f32 outputs
    "replay_sample_f32": 2 * OBS_ELEMS * 5,  # obs + next_obs: u8 ring in, f32 out
    "replay_shift_conv_u8_f32": OBS_ELEMS * 14,  # whole SVEA sample: obs u8 ring -> [shift; shift+conv] f32, next_obs u8 -> f32
}


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--batch", type=int, default=256, help="observations per GPU per step")
    ap.add_argument("--ring", type=int, default=8, help="distinct batches cycled through (defeats the 256 MiB L3)")
    ap.add_argument("--no-extras", action="store_true", help="headline metric only")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU-baseline work (s)")
    ap.add_argument("--no-sac", action="store_true", help="skip the update() legs (profiling the augmentation kernels)")
    ap.add_argument("--dp-graph", action="store_true",
                    help="with --gpus N > 1 also try update() replayed from hipGraphs that contain the RCCL all-reduce "
                         "(pre-flighted; default: the eager multi-rank path only)")
    ap.add_argument("--rehearse-one-gpu", action="store_true",
                    help="NOT a measurement: run the N ranks of --gpus N on GPU 0 with the gloo backend, to exercise the "
                         "multi-rank code path (launcher, sharding, overlapped gradient exchange) on a one-GPU box")
    ap.add_argument("--min-timed-s", type=float, default=MIN_TIMED_S, help="repeat the K-step block until this much is timed")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------ self-launch
def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def child_env(base, rank, world, port):
    """Environment of rank `rank` of `world` on this node (what torchrun would set)."""
    env = dict(base)
    env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL between processes needs it on this pool
    return env


def pick_json_line(text):
    """The ONE result line of rank 0: the last stdout line that parses as a JSON object."""
    for ln in reversed(text.splitlines()):
        ln = ln.strip()
        if ln.startswith("{") and ln.endswith("}"):
            try:
                json.loads(ln)
                return ln
            except ValueError:
                continue
    return None


def launch_ranks(world, argv, cmd=None, env=None, timeout=None):
    """Start `world` ranks of this script as CHILD processes (never an exec: the parent has not touched the GPU and
    never will), relay rank 0's JSON line on stdout and everything else on stderr; returns the exit status."""
    cmd = cmd or [sys.executable, os.path.abspath(__file__)]
    base = dict(os.environ if env is None else env)
    port = free_port()
    procs = []
    for r in range(world):
        procs.append(subprocess.Popen(cmd + list(argv), env=child_env(base, r, world, port), stdout=subprocess.PIPE,
                                      stderr=None if r == 0 else subprocess.STDOUT, text=True))
    outs, status = [""] * world, 0
    import threading

    def drain(r):  # read each child's pipe to EOF as it runs: a full pipe must never stall a rank
        outs[r] = procs[r].stdout.read()

    readers = [threading.Thread(target=drain, args=(r,), daemon=True) for r in range(world)]
    for th in readers:
        th.start()
    deadline = None if timeout is None else time.time() + timeout
    try:
        pending = set(range(world))
        while pending:
            for r in sorted(pending):
                rc = procs[r].poll()
                if rc is None:
                    continue
                pending.discard(r)
                if rc != 0 and status == 0:
                    status = rc
            if status != 0 or (deadline is not None and time.time() > deadline):
                break
            if pending:
                time.sleep(0.05)
    finally:
        for p in procs:  # a failed / timed-out rank takes the others down: exactly the PIDs started here
            if p.poll() is None:
                p.kill()
                status = status or 1
        for p, th in zip(procs, readers):
            p.wait()
            th.join(timeout=10)
    for r in range(1, world):
        if outs[r]:
            sys.stderr.write("".join(f"[rank {r}] {ln}\n" for ln in outs[r].splitlines()))
    line = pick_json_line(outs[0] or "")
    rest = [ln for ln in (outs[0] or "").splitlines() if ln.strip() and ln.strip() != line]
    if rest:
        sys.stderr.write("".join(f"[rank 0] {ln}\n" for ln in rest))
    if line is None:
        sys.stderr.write("[bench] rank 0 produced no JSON line\n")
        return status or 1
    sys.stdout.write(line + "\n")
    sys.stdout.flush()
    return status


# ------------------------------------------------------------------------------------------------ timing
def timed(fn, steps, warmup, dist, torch, min_s=MIN_TIMED_S, max_repeats=2000):
    """warm-up, then blocks of exactly `steps` calls, each between barrier + synchronize on both sides (wall clock,
    MAX over ranks); HIP events on the launch stream give the device-side average per launch.  Blocks repeat until
    `min_s` seconds have been timed; returns the median block and the spread."""
    t_ramp = time.perf_counter()  # untimed clock ramp: a cold GPU needs tens of ms of load to reach its
    while time.perf_counter() - t_ramp < 0.25:  # steady clock; W short warmup launches alone are ~1 ms
        for i in range(50):
            fn(i)
        torch.cuda.synchronize()
    for i in range(warmup):
        fn(i)
    torch.cuda.synchronize()
    walls, devs = [], []
    it = 0
    while True:
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for i in range(steps):
            fn(it + i)
        e1.record()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        it += steps
        if dist is not None:  # max over ranks; every rank then takes the same stop decision
            t = torch.tensor([wall], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            wall = float(t.item())
        walls.append(wall)
        devs.append(e0.elapsed_time(e1) / steps)
        if sum(walls) >= min_s or len(walls) >= max_repeats:
            break
    order = sorted(range(len(walls)), key=lambda k: walls[k])
    med = order[len(order) // 2]
    # dev_ms: the AVERAGE launch duration over everything timed (what a rocprofv3 --stats average of the same command
    # shows), not the median block: on a noisy box the two differ by a few per cent
    return {"wall": walls[med], "dev_ms": sum(devs) / len(devs), "dev_ms_median": sorted(devs)[len(devs) // 2],
            "repeats": len(walls), "timed_s": sum(walls),
            "wall_min": walls[order[0]], "wall_max": walls[order[-1]]}


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
    graph replayed, so a rank whose capture failed never leaves the others waiting inside a collective.  Only called
    with world > 1: a one-rank all-reduce captures as an EMPTY graph and proves nothing."""
    world = dist.get_world_size()
    assert world > 1
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


def load_profile_json(name):
    """A counter summary under profiles/ -- only if it was measured on THIS source tree: every summary carries the hash of
    csrc/ + include/ (augcl_amd._lib.source_hash) of the binary that produced it.  Otherwise {} and the caller reports null
    with "stale_profile": true (round 2 quoted counters of a slightly older binary, silently)."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", name)))
    except Exception:  # noqa: BLE001
        return {}
    from augcl_amd import _lib
    if d.get("source_hash") != _lib.source_hash():
        return {"stale_profile": True, "profile_source_hash": d.get("source_hash")}
    return d


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher around us: become one.  Nothing below this line has run, torch is not even imported here.
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path is the only path (no CPU fallback)")
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.rehearse_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    rccl_ranks = 1
    if world > 1:
        import torch.distributed as dist
        # RCCL prints a version banner on STDOUT when its communicator comes up; stdout carries the one JSON line,
        # so point fd 1 at stderr until the first collective has run
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            if args.rehearse_one_gpu:
                dist.init_process_group("gloo")
            else:
                dist.init_process_group("nccl", device_id=dev)
            probe = torch.ones(1, device=dev)
            dist.all_reduce(probe)  # SUM of ones = the number of ranks RCCL actually connected
            rccl_ranks = int(probe.item())
            dist.barrier()
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved, 1)
            os.close(saved)

    from augcl_amd import _lib, augmentations as aug
    _lib.load()
    B, R = args.batch, args.ring
    aug.set_sample_base(rank * B)  # in-kernel Philox parameters are keyed by the GLOBAL sample index: rank r owns [rB, (r+1)B)
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

    t = timed(conv_step, args.steps, args.warmup, dist, torch, min_s=args.min_timed_s)
    wall, dev_ms = t["wall"], t["dev_ms"]
    value = world * B * args.steps / wall
    ach = BYTES_PER_OBS["random_conv_f32"] * B / (dev_ms * 1e-3) / 1e9
    res = {
        "metric": "augmented obs/sec + SAC grad-steps/sec, walker_walk 84x84x9 batch=256",
        "value": value, "unit": "obs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": wall / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "repeats": t["repeats"], "timed_s": t["timed_s"], "value_min": world * B * args.steps / t["wall_max"],
        "value_max": world * B * args.steps / t["wall_min"], "rccl_ranks": rccl_ranks,
        **({"rehearsal": "ranks share GPU 0 over gloo: exercises the multi-rank path, measures nothing"} if args.rehearse_one_gpu else {}),
        "config": {"workload": "walker_walk SVEA strong augmentation: random_conv f32->f32 drop-in on (256,9,84,84) "
                               "per GPU, explicit per-sample 3x3x3x3 weights, ring of %d batches (> L3)" % R,
                   "batch_per_gpu": B, "obs_shape": [9, 84, 84], "parallelism": f"dp{world} (batch-sharded, no collective)"},
        "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                     "traffic": None, "kernel": "k_conv<84,84,12,256,float,false> (random_conv)",
                     "bytes_per_launch": BYTES_PER_OBS["random_conv_f32"] * B, "avg_launch_ms": dev_ms},
    }
    if dist is not None:  # every rank's own rate (its shard / the common block time is `value / world`; this is its device time)
        mine = torch.tensor([B / (dev_ms * 1e-3)], dtype=torch.float64, device=dev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        res["per_rank_obs_per_s_device"] = [float(v.item()) for v in allr]
    prof = load_profile_json("pmc_traffic.json")
    tr = prof.get("k_random_conv", {})
    if tr.get("batch") == B:  # measured offline with rocprofv3 --pmc (profiles/, tools/summarize_profiles.py) on THIS tree
        res["roofline"]["traffic"] = tr.get("hbm_bytes_per_launch")
        res["roofline"]["traffic_source"] = "profiles/pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)"
    elif prof.get("stale_profile"):
        res["roofline"]["stale_profile"] = True  # counters on file belong to another source tree: not quoted

    if not args.no_extras:
        extras = {}
        n2 = max(20, args.steps // 4)

        def rate(name, fn):
            t_ = timed(fn, n2, 10, dist, torch, min_s=min(0.25, args.min_timed_s))
            a_ = BYTES_PER_OBS[name] * B / (t_["dev_ms"] * 1e-3) / 1e9
            extras[name] = {"obs_per_s": world * B * n2 / t_["wall"], "GB/s": a_, "frac": a_ / HBM_PEAK_GBS,
                            "avg_launch_ms": t_["dev_ms"], "repeats": t_["repeats"]}

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
        rate("replay_sample_f32", lambda i: rb.sample())
        if hasattr(rb, "sample_shift_conv"):  # gather -> random_shift -> random_conv in one pass over the uint8 ring
            rate("replay_shift_conv_u8_f32", lambda i: rb.sample_shift_conv())
        # BASELINE config 4's batch (1024 per launch): the same kernels once the launch is long enough for ramp / drain and
        # workgroup-count quantisation (color_jitter: 3 frames per CU on 2 slots at B=256) not to matter
        B4 = 1024
        x4 = [torch.randint(0, 256, (B4, 9, 84, 84), dtype=torch.uint8, device=dev, generator=g).float() for _ in range(3)]
        y4 = [torch.empty_like(x4[0]) for _ in range(3)]
        w4 = torch.randn((B4, 81), device=dev, generator=g)

        def rate4(name, key, fn):
            t_ = timed(fn, max(10, n2 // 4), 5, dist, torch, min_s=min(0.25, args.min_timed_s))
            a_ = BYTES_PER_OBS[key] * B4 / (t_["dev_ms"] * 1e-3) / 1e9
            extras[name] = {"obs_per_s": world * B4 * max(10, n2 // 4) / t_["wall"], "GB/s": a_, "frac": a_ / HBM_PEAK_GBS,
                            "avg_launch_ms": t_["dev_ms"], "batch": B4}

        rate4("random_conv_f32_B1024", "random_conv_f32", lambda i: call("augcl_random_conv_f32", p(x4[i % 3]), p(y4[i % 3]), B4, 3, 84,
                                                                           84, p(w4), 0, 0, 0, stream))
        rate4("color_jitter_f32_B1024", "color_jitter_f32", lambda i: call("augcl_color_jitter_f32", p(y4[i % 3]), B4, 3, 84, 84, None, 0,
                                                                             order_p, 1.2, 0.8, 0.1, 1.0, 255.0, stream))
        del x4, y4, w4
        res["extras"] = extras

    if not args.no_extras and not args.no_sac:
        from augcl_amd import bench_update
        n_upd = max(10, args.steps // 20)
        # update() replayed from captured hipGraphs (SAC.enable_hip_graphs); the eager figure beside it.
        # With several ranks the graphs contain the gradient all-reduce: checked by a pre-flight first.
        use_graph = dist is None or (args.dp_graph and not args.rehearse_one_gpu and graph_collectives_ok(dist, dev, torch))
        busy = load_profile_json("r04_mfma_busy.json")
        hbm16 = load_profile_json("r04_bf16_conv_hbm.json")
        hbm32 = load_profile_json("r04_f32_conv_hbm.json")

        def leg(precision):
            """Both execution modes of update(); the faster one is the reported figure, the other sits beside it
            (the f32 path is GPU-bound, so replaying graphs buys it nothing; the bf16 mode is launch-bound)."""
            runs = [bench_update.run(dev, dist, world, rank, steps=n_upd, precision=precision, graph=gr)
                    for gr in ((True, False) if use_graph else (False,))]
            best = min(runs, key=lambda r: r["ms_per_update"])
            for r in runs:
                if r is not best:
                    best["hip_graph_ms_per_update" if r["hip_graph"] else "eager_ms_per_update"] = r["ms_per_update"]
            # The f32 parity path runs its 32 -> 32 layers as THREE f16 MFMAs per product group (csrc/encoder_x2.hip: every f32
            # operand an f16 pair under a per-tensor power-of-two scale).  Two roofs are quoted: the f16 matrix pipe -- achieved =
            # the MFMA flops those layers ISSUE (3 x their executed f32-equivalent flops) per second, whole update in the
            # denominator -- and HBM: the conv kernels' algorithmic bytes over their rocprofv3 durations, time-weighted (from the
            # profile of the same tree, null when the one on file is stale).  The bound is the roof the path sits closer to.
            if precision == "f32":
                issued = 3.0 * best["executed_conv32_GFLOP_per_update"] * 1e9 / (best["ms_per_update"] * 1e-3) / 1e12
                h = hbm32 if not hbm32.get("stale_profile") else {}
                hb = h.get("conv_GBps_time_weighted")
                mf = {"achieved": issued, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": issued / MFMA_BF16_PEAK_TFLOPS,
                      "note": "f16 MFMA flops issued (3 per f32 product), whole update() in the denominator"}
                if hb and hb / HBM_PEAK_GBS > mf["frac"]:
                    best["roofline"] = {"bound": "hbm (conv kernels of the f32 path: f32 NCHW activations in and out)", "achieved": hb,
                                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hb / HBM_PEAK_GBS,
                                        "source": "profiles/r04_f32_conv_hbm.json (algorithmic bytes of every conv launch / its rocprofv3 "
                                                  "duration, time-weighted)", "mfma_f16": mf}
                else:
                    best["roofline"] = {"bound": "mfma_f16 (f32-accurate products: 3 f16 MFMAs per product group)", **mf}
                best["roofline"].update({"f32_equivalent_executed_TFLOP/s": best["executed_TFLOP/s"],
                                         "f32_equivalent_algorithmic_TFLOP/s": best["algorithmic_TFLOP/s"],
                                         "native_f32_mfma_peak_TFLOP/s": MFMA_F32_PEAK_TFLOPS})
            else:
                h = hbm16 if not hbm16.get("stale_profile") else {}
                best["roofline"] = {"bound": "hbm (conv kernels; heads and optimiser stay f32)", "achieved": h.get("conv_GBps_time_weighted"),
                                    "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                    "frac": (h.get("conv_GBps_time_weighted") / HBM_PEAK_GBS) if h.get("conv_GBps_time_weighted") else None,
                                    "source": "profiles/r04_bf16_conv_hbm.json (algorithmic bytes of every conv launch / its rocprofv3 "
                                              "duration, time-weighted)" if h else None}
            b_ = busy.get(precision) if not busy.get("stale_profile") else None
            best["roofline"]["mfma_busy"] = b_
            best["roofline"]["mfma_busy_source"] = ("profiles/r04_mfma_busy.json (rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES per conv "
                                                    "kernel / launch duration x 2.4 GHz x 1024 SIMDs, time-weighted)") if b_ else None
            if busy.get("stale_profile") or hbm16.get("stale_profile") or hbm32.get("stale_profile"):
                best["roofline"]["stale_profile"] = True
            return best

        try:
            res["sac"] = leg("f32")  # the parity path
            res["sac"]["bf16_mode"] = leg("bf16")
            # the parity path's two other arithmetic forms on the same box, same harness (short legs): range-free bf16 triples
            # (6 MFMAs per product group) and the native f32 MFMA pipe -- what the f16-pair kernels are measured against
            if use_graph:
                res["sac"]["other_f32_forms_ms_per_update"] = {
                    p_: bench_update.run(dev, dist, world, rank, steps=max(10, n_upd // 2), precision=p_, graph=True, min_s=0.2)["ms_per_update"]
                    for p_ in ("f32_x3", "f32_mfma")}
                # ... and the same update() fed by the product's uint8 replay ring instead of pre-staged float batches (SVEA's
                # sample + shift + conv as one pass over the ring; next_obs gathered straight into the graph's static input)
                res["sac"]["from_uint8_replay_ring_ms_per_update"] = bench_update.run(
                    dev, dist, world, rank, steps=max(10, n_upd // 2), precision="f32", graph=True, min_s=0.2, replay="ring")["ms_per_update"]
                # the same update() on DENSE conv filters (He-scaled; a fresh agent's are delta-orthogonal: 8 of 9 taps exactly zero,
                # src/algorithms/modules.py:52-66 -- dense operands cost the matrix pipe more power) with the shader clock / board power
                # sampled during both, and with a reference-like Logger in the loop (src/logger.py:138-141) instead of L = None
                init_p = bench_update.run(dev, dist, world, rank, steps=n_upd, precision="f32", graph=True, min_s=2.0, power=True)
                dense = bench_update.run(dev, dist, world, rank, steps=n_upd, precision="f32", graph=True, min_s=2.0, power=True,
                                         weights="dense")
                res["sac"]["delta_init_vs_dense_weights"] = {
                    "delta_init": {"ms_per_update": init_p["ms_per_update"], **(init_p["clock_power"] or {})},
                    "dense": {"ms_per_update": dense["ms_per_update"], **(dense["clock_power"] or {})}}
                lg = bench_update.run(dev, dist, world, rank, steps=n_upd, precision="f32", graph=True, min_s=0.5, logger=True)
                res["sac"]["with_logger_ms_per_update"] = lg["ms_per_update"]
                res["sac"]["with_logger_item_calls"] = lg["logger_item_calls"]
                if dist is None:
                    # what each data-parallel rank of config 4 (AugCL conv-overlay, global batch 1024) will run: update() at the
                    # per-rank batches of 1 / 2 / 4 / 8 ranks on ONE GPU -- no collective here; DESIGN.md section 6 turns these
                    # into the scaling ceiling the first 8-GPU run is to be checked against
                    res["sac"]["augcl_conv_overlay_ms_per_update_by_batch"] = {
                        str(b_): bench_update.run(dev, None, 1, 0, steps=max(10, n_upd // 4), B=b_, algo="augcl", data_aug="conv-overlay",
                                                  precision="f32", graph=True, min_s=0.3)["ms_per_update"]
                        for b_ in (1024, 512, 256, 128)}
                    # critic passes of at most one image per CU run the gradient-carrying forward through the storing layer chains
                    # (augcl_enc_stack_forward_chained): RAD at the config's batch, both arms on this box
                    from augcl_amd import encoder_hip
                    ab = {}
                    for arm, on in (("per_layer_kernels", False), ("storing_chains", True)):
                        encoder_hip._CHAINS_GRAD = on
                        try:
                            ab[arm] = bench_update.run(dev, None, 1, 0, steps=max(10, n_upd // 2), B=256, algo="rad", data_aug="shift",
                                                       precision="f32", graph=True, min_s=0.5)["ms_per_update"]
                        finally:
                            encoder_hip._CHAINS_GRAD = True
                    res["sac"]["rad_shift_B256_ms_per_update"] = ab
            res["sac"]["select_action_us"] = bench_update.action_latency(dev)  # batch-1 acting, eager vs captured graph
        except Exception as e:  # noqa: BLE001 -- the headline above is measured: report the failure, do not lose the line
            import traceback
            traceback.print_exc()
            res.setdefault("sac", {})["error"] = repr(e)

    if rank == 0 and not args.no_cpu_baseline and world == 1:
        res["cpu_baseline"] = cpu_baseline(args.cpu_seconds)
    elif rank == 0:
        res["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(res))
        sys.stdout.flush()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
