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
out = os.path.join(ROOT, "profiles")
CLK_HZ, SIMDS = 2.4e9, 256 * 4  # MI355X engine clock (peak), 256 CUs x 4 SIMDs

for src, dst in (("bench.csv", f"{tag}_bench_kernel_stats.csv"), ("update_f32.csv", f"{tag}_update_kernel_stats.csv"),
                 ("update_bf16.csv", f"{tag}_update_bf16_kernel_stats.csv")):
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
    res = {"round": tag, "batch": 256, "note": "FETCH_SIZE doubled (gfx950 correction), separate --pmc passes", "kernels": kernels}
    key = [k for k in kernels if "k_conv<84, 84, 12, 256, float, false" in k]
    if key:
        res["k_random_conv"] = {"batch": 256, **kernels[key[0]]}
    json.dump(res, open(os.path.join(out, "pmc_traffic.json"), "w"), indent=1)
    print("pmc_traffic.json:", {k[:60]: v.get("traffic_over_algorithmic") for k, v in kernels.items()})

# ---- MFMA-busy of the encoder kernels inside update()
busy = {"round": tag, "formula": "whole_launch = SQ_VALU_MFMA_BUSY_CYCLES / (launch duration x 2.4 GHz x 1024 SIMDs); "
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

