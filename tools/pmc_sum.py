#!/usr/bin/env python3
"""Average rocprofv3 --pmc counters per kernel.  usage: tools/pmc_sum.py DIR [DIR...]"""
import collections, csv, glob, sys
for d in sys.argv[1:]:
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"].split("(")[0][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in agg.items():
        print(d, k, {c: round(sum(v) / len(v)) for c, v in sorted(cs.items())})
