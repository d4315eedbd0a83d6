#!/usr/bin/env python3
"""Build the tracked summaries under profiles/ from the small per-run files the on-box tools leave in gpurun_out/
(tools/prof_summary.py -> *.csv / *.json of `rocprofv3 --kernel-trace --stats`; tools/pmc_summary.py -> *.json of the
`rocprofv3 --pmc ...` passes, one counter group per pass as MI355X_MICROARCH.md prescribes).

usage: tools/summarize_profiles.py ROUND   (reads gpurun_out/{ROUND}_bench.csv, _fetch.json, _write.json,
       _update_f32.csv, _update_bf16.csv, _mfma_f32.json, _mfma_bf16.json where ROUND is e.g. r2)"""
import csv, json, os, shutil, sys

rnd = sys.argv[1]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = lambda n: os.path.join(ROOT, "gpurun_out", f"{rnd}_{n}")  # noqa: E731
tag = "r%02d" % int(rnd[1:])
out = os.environ.get("AUGCL_PROFILES_OUT", os.path.join(ROOT, "profiles"))
# hash of the source tree whose binary produced these counters (written on the box by tools/profile_round.sh)
SRC_HASH = open(G("source_hash.txt")).read().strip() if os.path.exists(G("source_hash.txt")) else None
sys.path.insert(0, ROOT)
try:
    from augcl_amd import _lib
    if SRC_HASH and SRC_HASH != _lib.source_hash():
        print("NOTE: the profiled tree (%s) is not the tree checked out here (%s): bench.py will report these counters as stale"
              % (SRC_HASH, _lib.source_hash()))
except Exception as e:  # noqa: BLE001
    print("source hash check skipped:", e)
CLK_HZ, SIMDS = 2.4e9, 256 * 4  # MI355X engine clock (peak), 256 CUs x 4 SIMDs

for src, dst in (("bench.csv", f"{tag}_bench_kernel_stats.csv"), ("update_f32.csv", f"{tag}_update_kernel_stats.csv"),
                 ("update_bf16.csv", f"{tag}_update_bf16_kernel_stats.csv"), ("update_f32_mfma.csv", f"{tag}_update_f32_mfma_kernel_stats.csv"),
                 ("update_f32_x3.csv", f"{tag}_update_f32_x3_kernel_stats.csv")):
    if os.path.exists(G(src)):
        shutil.copy(G(src), os.path.join(out, dst))

# ---- HBM traffic per launch of every augmentation / replay kernel (B = 256)
if os.path.exists(G("fetch.json")) and os.path.exists(G("write.json")):
    def smallest_grid(d):
        """{kernel [grid N]: counters} -> {kernel: counters of its SMALLEST launch grid} (the B = 256 launches; the bench
        also runs some kernels at B = 1024)"""
        best = {}
        for k, v in d.items():
            name, _, g = k.partition(" [grid ")
            grid = int(g.rstrip("]")) if g.rstrip("]").isdigit() else 0
            # (k_replay_sample: its LARGEST grid is the full obs + next_obs sample; the half grid is the next_obs-only call)
            if name not in best or ((grid > best[name][0]) if "k_replay_sample" in name else (grid < best[name][0])):
                best[name] = (grid, v)
        return {n: v for n, (_, v) in best.items()}

    fetch, write = smallest_grid(json.load(open(G("fetch.json")))), smallest_grid(json.load(open(G("write.json"))))
    OBS = 9 * 84 * 84
    algo = {  # algorithmic bytes per launch at B = 256
        "k_conv<84, 84, 12, 256, float, false, false": 256 * (2 * OBS * 4 + 324),
        "k_conv<84, 84, 12, 256, unsigned char, true, false": 256 * OBS * 9,
        "k_conv<84, 84, 12, 256, unsigned char, true, true": 256 * OBS * 9,
        "k_random_shift<84, 84, 12, float, float>": 256 * 2 * OBS * 4,
        "k_random_shift<84, 84, 12, unsigned char, float>": 256 * OBS * 5,
        "k_random_overlay": 256 * (2 * OBS * 4 + 3 * 84 * 84 * 4),
        "k_color_jitter_reg": 256 * 2 * OBS * 4,
        "k_replay_sample<float>": 256 * 2 * OBS * 5,
    }
    kernels = {}
    for k in sorted(set(fetch) | set(write)):
        f_kb = fetch.get(k, {}).get("FETCH_SIZE", {}).get("avg")
        w_kb = write.get(k, {}).get("WRITE_SIZE", {}).get("avg")
        ent = {"FETCH_SIZE_KiB_raw": f_kb, "WRITE_SIZE_KiB_raw": w_kb}
        if f_kb is not None and w_kb is not None:
            # MI355X_MICROARCH.md (HBM): units are KiB; on gfx950 FETCH_SIZE reads exactly 1/2 of a wide coalesced stream
            # (16 B/lane) -> doubled; WRITE_SIZE is exact for 16-B/lane streaming stores.  (uint8 inputs are read 4 B per
            # lane -- an access width the guide calls uncalibrated: their figures are indicative.)
            ent["hbm_bytes_per_launch"] = int(2 * f_kb * 1024 + w_kb * 1024)
            a = [v for n, v in algo.items() if n in k]
            if a:
                ent["algorithmic_bytes_per_launch"] = a[0]
                ent["traffic_over_algorithmic"] = round(ent["hbm_bytes_per_launch"] / a[0], 3)
        kernels[k] = ent
    res = {"round": tag, "source_hash": SRC_HASH, "batch": 256, "note": "FETCH_SIZE doubled (gfx950 correction), separate --pmc passes",
           "kernels": kernels}
    key = [k for k in kernels if "k_conv<84, 84, 12, 256, float, false" in k]
    if key:
        res["k_random_conv"] = {"batch": 256, **kernels[key[0]]}
    json.dump(res, open(os.path.join(out, "pmc_traffic.json"), "w"), indent=1)
    print("pmc_traffic.json:", {k[:60]: v.get("traffic_over_algorithmic") for k, v in kernels.items()})

# ---- MFMA-busy of the encoder kernels inside update()
busy = {"round": tag, "source_hash": SRC_HASH, "formula": "whole_launch = SQ_VALU_MFMA_BUSY_CYCLES / (launch duration x 2.4 GHz x 1024 SIMDs); "
        "while_cu_busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_BUSY_CU_CYCLES).  SQ_VALU_MFMA_BUSY_CYCLES sums every SIMD's busy "
        "cycles (checked on k_proj_partial: 16 x 63 x 4 x 7 x 16 MFMAs x 64 cycles = the counter to 0.3 %); durations from the "
        "rocprofv3 --kernel-trace --stats run of the same command; f32 / bf16 = time-weighted over the kernels listed."}
for prec in ("f32", "bf16"):
    if not (os.path.exists(G(f"mfma_{prec}.json")) and os.path.exists(G(f"update_{prec}.csv"))):
        continue
    pm = {}
    for k, v in json.load(open(G(f"mfma_{prec}.json"))).items():  # merge the launch grids of one kernel (count-weighted)
        name = k.partition(" [grid ")[0]
        for c, x in v.items():
            e = pm.setdefault(name, {}).setdefault(c, {"avg": 0.0, "n": 0})
            e["avg"] = (e["avg"] * e["n"] + x["avg"] * x["n"]) / (e["n"] + x["n"])
            e["n"] += x["n"]
    tot = {}
    for r in csv.reader(l for l in open(G(f"update_{prec}.csv")) if not l.startswith("#")):
        if r and r[0] != "name":  # name, grid, calls, total_us, ...: one row per launch grid -> merged per kernel
            t = tot.setdefault(r[0], [0.0, 0])
            t[0] += float(r[3])
            t[1] += int(r[2])
    dur = {n: (t[0] / t[1], t[1]) for n, t in tot.items()}  # avg_us, calls
    per, num, den = {}, 0.0, 0.0
    for k, v in pm.items():
        mf = v.get("SQ_VALU_MFMA_BUSY_CYCLES", {}).get("avg", 0.0)
        if mf <= 0:
            continue
        match = [d for n, d in dur.items() if n.split("(")[0].strip()[:60] == k[:60]]
        if not match:
            continue
        avg_us, calls = match[0]
        cap = avg_us * 1e-6 * CLK_HZ * SIMDS
        per[k] = {"mfma_busy_cycles": mf, "avg_us": avg_us, "calls_in_trace": calls, "whole_launch": round(mf / cap, 3),
                  "while_cu_busy": round(mf / (4.0 * v["SQ_BUSY_CU_CYCLES"]["avg"]), 3)}
        num += mf * calls
        den += cap * calls
    busy[prec] = round(num / den, 3) if den else None
    busy[f"{prec}_kernels"] = per
if "f32" in busy or "bf16" in busy:
    json.dump(busy, open(os.path.join(out, f"{tag}_mfma_busy.json"), "w"), indent=1)
    print("mfma_busy:", busy.get("f32"), busy.get("bf16"))

# ---- the conv encoder's kernels inside update(): algorithmic bytes per launch (launch mix of one average SVEA update at
# B = 256: two no-grad passes as ONE dual launch per layer over 2 x 256 images, the critic's forward / backward over 2B = 512,
# the actor step's forward over 256 every second update) against the PMC traffic, and -- bf16 mode -- the time-weighted HBM rate
HS = [41 - 2 * i for i in range(11)]  # output size of layer i (input of layer i + 1)


def conv_mix(elem, gate_bits=False, chains=False):
    """{kernel-name fragment: (launches per update, algorithmic bytes per update)} for activations of `elem` bytes;
    gate_bits: backward-data reads one 32-bit word per pixel (the forward pass's gate bits) instead of the layer input;
    chains (round 4, f32 path): the no-grad passes (dual forward over 2 x 256 images, the actor step's pass over 256 every second
    update) run as two chain launches each (layers 1-4: 41 -> 33, layers 5-10: 33 -> 21; csrc/encoder_chain.hip) whose
    algorithmic bytes are the chain's input and output only"""
    px = lambda n: 32 * n * n  # noqa: E731
    fwd = bwd = wg = 0.0
    nograd = 0.0 if chains else (512 + 0.5 * 256)
    for i in range(1, 11):
        hi, ho = HS[i - 1], HS[i]
        o_el = 4 if (i == 10) else elem  # the last layer always writes f32
        fwd += (512 + nograd) * (px(hi) * elem + px(ho) * o_el)
        gate = hi * hi * 4 if gate_bits else px(hi) * elem
        bwd += 512 * (px(ho) * elem + gate + px(hi) * elem)    # dZ in, gate in, dX out
        wg += 512 * (px(hi) * elem + px(ho) * elem)
    first = (256 + 512 + 0.5 * 256) * (9 * 84 * 84 * 4) + (512 + 512 + 0.5 * 256) * px(41) * elem  # dual launch reads x once
    wfirst = 512 * (9 * 84 * 84 * 4 + px(41) * elem)
    mix = {"fwd": (10 if chains else 25, fwd), "bwd": (10, bwd), "wgrad": (10, wg), "first": (2.5, first), "wfirst": (1, wfirst)}
    if chains and elem == 4:
        mix["chain"] = (3, (512 + 0.5 * 256) * 4 * (px(41) + 2 * px(33) + px(21)))
    if chains and elem == 2:
        # bf16 mode (round 4): ONE chain launch per pass (a whole image per work item).  No-grad passes read a_0 (bf16) and write
        # the last layer (f32); the gradient-carrying pass also stores every intermediate layer once (bf16); backward-data is one
        # launch as well: it reads the top gradient and every layer's gate (= that layer's bf16 input), writes every gradient map
        inter = sum(px(HS[i]) * 2 for i in range(1, 10))
        mix["fwd"] = (0, 0.0)
        mix["chain"] = (2.5, (512 + 0.5 * 256) * (px(41) * 2 + px(21) * 4) + 512 * (px(41) * 2 + inter + px(21) * 4))
        mix["bwd"] = (1, 512 * (px(21) * 2 + 2 * sum(px(HS[i]) * 2 for i in range(0, 10))))
    return mix


KERNEL_ROLE = [("k_bwd32_chain_bf16", "bwd"), ("k_conv32_chain", "chain"), ("k_conv32_x2<0", "fwd"), ("k_conv32_x2<1", "bwd"), ("k_wgrad32_x2", "wgrad"), ("k_conv32_x3<0", "fwd"), ("k_conv32_x3<1", "bwd"), ("k_wgrad32_x3", "wgrad"), ("k_conv32_bf16<0", "fwd"),
               ("k_conv32_bf16<1", "bwd"), ("k_wgrad32_bf16", "wgrad"), ("k_conv_first9", "first"), ("k_wgrad_first", "wfirst"),
               ("k_conv32<0", "fwd"), ("k_conv32<1", "bwd"), ("k_wgrad32<", "wgrad")]


def role_of(name):
    for frag, role in KERNEL_ROLE:
        if frag in name:
            return role
    return None


def kernel_times(csv_path):
    """{kernel name: (calls, total_us)} merged over launch grids"""
    tot = {}
    for r in csv.reader(l for l in open(csv_path) if not l.startswith("#")):
        if r and r[0] != "name":
            t = tot.setdefault(r[0], [0, 0.0])
            t[0] += int(r[2])
            t[1] += float(r[3])
    return tot


upd = {"round": tag, "source_hash": SRC_HASH, "note": "conv encoder kernels inside SVEA update() B=256; algorithmic bytes = the f32 / bf16 "
       "activations each launch must read and write (launch mix of an average update); hbm = 2 x FETCH_SIZE + WRITE_SIZE (KiB, separate "
       "--pmc passes, gfx950 correction)"}
for prec, elem in (("f32", 4), ("bf16", 2)):
    if not os.path.exists(G(f"update_{prec}.csv")):
        continue
    times = kernel_times(G(f"update_{prec}.csv"))
    mix = conv_mix(elem, gate_bits=(prec == "f32"), chains=any("k_conv32_chain" in n for n in times))
    fetch = json.load(open(G(f"ufetch_{prec}.json"))) if os.path.exists(G(f"ufetch_{prec}.json")) else {}
    write = json.load(open(G(f"uwrite_{prec}.json"))) if os.path.exists(G(f"uwrite_{prec}.json")) else {}

    def pmc_avg(d, name, counter):
        num = den = 0.0
        for k, v in d.items():
            if k.partition(" [grid ")[0][:60] == name.split("(")[0].strip()[:60] and counter in v:
                num += v[counter]["avg"] * v[counter]["n"]
                den += v[counter]["n"]
        return num / den if den else None

    # Kernels are merged BY ROLE before the launch mix is applied: several instantiations share a role (k_conv32_x2<0,0>, the
    # no-grad / dual forward, and k_conv32_x2<0,1>, the gradient-carrying forward that also writes gate bits, are both "fwd"),
    # and round 3's per-kernel loop added the whole role's launches and bytes once per instantiation (conv_us_per_update came
    # out larger than the update itself).  The time per update is MEASURED: every conv kernel's total time in the trace over
    # the number of update() calls in it (= launches of the first layer's backward-weights kernel, one per update).
    by_role = {}
    for name, (calls, total_us) in times.items():
        role = role_of(name)
        if role is None or "augcl::" not in name:
            continue
        by_role.setdefault(role, []).append((name, calls, total_us))
    n_upd = sum(c for _, c, _ in by_role.get("wfirst", [])) or None
    rows, tot_b, tot_t = {}, 0.0, 0.0
    for role, ks in by_role.items():
        launches, bytes_upd = mix[role]
        if not launches:  # a role the launch mix says is gone but the trace still shows (fallback launches): time counted, no bytes
            launches, bytes_upd = sum(c for _, c, _ in ks) / max(n_upd or 1, 1), 0.0
        calls = sum(c for _, c, _ in ks)
        total_us = sum(t for _, _, t in ks)
        us_per_update = total_us / n_upd if n_upd else total_us / calls * launches
        ent = {"kernels": {n.split("(")[0][:70]: {"calls": c, "avg_us": round(t / c, 2)} for n, c, t in ks},
               "launches_per_update_assumed": launches, "launches_per_update_in_trace": round(calls / n_upd, 2) if n_upd else None,
               "us_per_update": round(us_per_update, 1), "algorithmic_bytes_per_update": int(bytes_upd),
               "algorithmic_GBps": round(bytes_upd / us_per_update / 1e3, 1)}
        hb = 0.0
        for n, c, _ in ks:  # PMC traffic: per-launch averages of each instantiation, weighted by its share of the role's launches
            f_kb, w_kb = pmc_avg(fetch, n, "FETCH_SIZE"), pmc_avg(write, n, "WRITE_SIZE")
            if f_kb is None or w_kb is None:
                hb = None
                break
            hb += (2 * f_kb * 1024 + w_kb * 1024) * c / calls * launches
            ent["kernels"][n.split("(")[0][:70]]["hbm_bytes_per_launch"] = int(2 * f_kb * 1024 + w_kb * 1024)
        if hb is not None:
            ent["hbm_bytes_per_update"] = int(hb)
            ent["traffic_over_algorithmic"] = round(hb / bytes_upd, 3)
        rows[role] = ent
        tot_b += bytes_upd
        tot_t += us_per_update
    upd[prec] = {"updates_in_trace": n_upd, "roles": rows, "conv_us_per_update": round(tot_t, 1),
                 "conv_algorithmic_MB_per_update": round(tot_b / 1e6, 1),
                 "conv_GBps_time_weighted": round(tot_b / tot_t / 1e3, 1) if tot_t else None}
    meas = G(f"update_{prec}_ms.txt")  # the same run's update() time, when the profiling script left it
    if os.path.exists(meas):
        ms = float(open(meas).read().split()[0])
        upd[prec]["update_ms_same_run"] = ms
        assert tot_t <= ms * 1e3 * 1.02, f"{prec}: conv kernels {tot_t:.0f} us per update exceed the update itself ({ms} ms)"
if "f32" in upd or "bf16" in upd:
    json.dump(upd, open(os.path.join(out, f"{tag}_update_conv_traffic.json"), "w"), indent=1)
    for p_ in ("f32", "bf16"):
        if p_ in upd:
            json.dump({"round": tag, "source_hash": SRC_HASH, "conv_GBps_time_weighted": upd[p_]["conv_GBps_time_weighted"],
                       "conv_us_per_update": upd[p_]["conv_us_per_update"], "frac_of_8TBps": round(upd[p_]["conv_GBps_time_weighted"] / 8000.0, 3),
                       "source": f"profiles/{tag}_update_conv_traffic.json"}, open(os.path.join(out, f"{tag}_{p_}_conv_hbm.json"), "w"), indent=1)
    print("update conv traffic:", {p: upd[p]["conv_GBps_time_weighted"] for p in ("f32", "bf16") if p in upd})

# ---- the launch sequence of graph-replayed updates: launches per update() with and without the actor step
if os.path.exists(G("seq_f32_sequence.txt")):
    rows = [l.rstrip("\n") for l in open(G("seq_f32_sequence.txt"))]
    body = [l for l in rows if not l.startswith("#")]
    marks = [i for i, l in enumerate(body) if "k_conv<84" in l or "k_replay_shift_conv" in l]  # the eager front end starts an update
    counts = [b - a for a, b in zip(marks[:-1], marks[1:])]
    with open(os.path.join(out, f"{tag}_update_launch_sequence.txt"), "w") as o:
        o.write("# kernel launches of consecutive SVEA update() calls, f32, hipGraph replay (tools/prof_sequence.py); an update starts at\n"
                "# the front end's k_conv<...> launch; the actor / temperature step runs every second update\n"
                f"# launches per update in this window: {counts}  (average over an actor + a critic-only update: "
                f"{sum(counts[-2:]) / 2 if len(counts) >= 2 else 'n/a'})\n")
        o.write("\n".join(rows) + "\n")
    print("launches per update:", counts)

