#!/usr/bin/env python3
"""Turn rocprofv3 output dirs (gpurun_out/...) into the small summaries committed under profiles/.
usage: tools/summarize_profiles.py ROUND stats_dir pmc_fetch_dir pmc_write_dir [update_stats_dir]"""
import collections, csv, glob, json, os, sys

rnd, stats_dir, fetch_dir, write_dir = sys.argv[1:5]
upd_dir = sys.argv[5] if len(sys.argv) > 5 else None
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(ROOT, "profiles")
os.makedirs(out, exist_ok=True)


def stats(d, dst, top=30):
    f = glob.glob(os.path.join(d, "*", "*_kernel_stats.csv"))[0]
    rows = list(csv.DictReader(open(f)))
    with open(dst, "w") as o:
        o.write("# rocprofv3 --kernel-trace --stats (top kernels)\nname,calls,total_us,avg_us,min_us,max_us,pct\n")
        for r in rows[:top]:
            o.write('"%s",%s,%.1f,%.2f,%.2f,%.2f,%s\n' % (r["Name"][:110], r["Calls"], float(r["TotalDurationNs"]) / 1e3,
                                                     float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, r["Percentage"]))
    return rows


def pmc(d, counter):
    f = glob.glob(os.path.join(d, "*", "*_counter_collection.csv"))[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter and "augcl::" in r["Kernel_Name"]:
            agg[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}


rows = stats(stats_dir, os.path.join(out, f"{rnd}_bench_kernel_stats.csv"))
if upd_dir:
    stats(upd_dir, os.path.join(out, f"{rnd}_update_kernel_stats.csv"), top=40)
fetch, write = pmc(fetch_dir, "FETCH_SIZE"), pmc(write_dir, "WRITE_SIZE")
traffic = {}
for k in sorted(set(fetch) | set(write)):
    f_kb, w_kb = fetch.get(k), write.get(k)
    # MI355X_MICROARCH.md (HBM): units are KiB; on gfx950 FETCH_SIZE reads exactly 1/2 of a wide coalesced stream
    # (16 B/lane) -> doubled; WRITE_SIZE is exact for 16-B/lane streaming stores.
    ent = {"FETCH_SIZE_KiB_raw": f_kb, "WRITE_SIZE_KiB_raw": w_kb}
    if f_kb is not None and w_kb is not None:
        ent["hbm_bytes_per_launch"] = int(2 * f_kb * 1024 + w_kb * 1024)
    traffic[k] = ent
key = [k for k in traffic if "k_conv<84, 84, 12, 256, float, false" in k]
res = {"round": rnd, "batch": 256, "note": "FETCH_SIZE doubled (gfx950 correction), separate --pmc passes", "kernels": traffic}
if key:
    res["k_random_conv"] = {"batch": 256, **traffic[key[0]]}
json.dump(res, open(os.path.join(out, "pmc_traffic.json"), "w"), indent=1)
print(json.dumps(res.get("k_random_conv"), indent=1))
