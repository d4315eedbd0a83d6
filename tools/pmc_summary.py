#!/usr/bin/env python3
"""Run ON the GPU box after `rocprofv3 --pmc C1 [C2 ..] --kernel-trace --output-format csv -d DIR -- ...`: average every
counter per kernel name (augcl:: kernels only unless --all) into DIR.json and delete the raw CSVs.
usage: tools/pmc_summary.py DIR [--all]"""
import collections, csv, glob, json, os, shutil, sys
d = sys.argv[1].rstrip("/")
every = "--all" in sys.argv
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(d, "*", "*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if every or "augcl::" in k:  # one entry per (kernel, launch grid): the same kernel runs at several batch sizes
            agg[k.split("(")[0].strip() + " [grid %s]" % r.get("Grid_Size", "?")][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: {c: {"avg": sum(v) / len(v), "n": len(v)} for c, v in cs.items()} for k, cs in agg.items()}
json.dump(out, open(d + ".json", "w"), indent=1)
shutil.rmtree(d)
print(d + ".json", len(out), "kernels")
