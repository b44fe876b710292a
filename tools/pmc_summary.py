#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc csv outputs: mean counter value per kernel (per dispatch)."""
import csv, glob, sys, collections, re
out = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(d + "/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
            out[k][r["Counter_Name"]].append((f + ":" + r["Dispatch_Id"], float(r["Counter_Value"])))   # one pass = one directory
for k, cs in out.items():
    print(k)
    for c, vals in sorted(cs.items()):
        per = collections.defaultdict(float)
        for did, v in vals:
            per[did] += v
        xs = list(per.values())
        print(f"   {c:40s} mean/dispatch {sum(xs)/len(xs):.6g}  (n={len(xs)})")
