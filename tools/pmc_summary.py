#!/usr/bin/env python
"""Summarise rocprofv3 --pmc counter_collection CSVs per kernel (mean per dispatch).
usage: pmc_summary.py <dir> <substring> [<substring> ...]"""
import csv, glob, re, sys, collections
csv.field_size_limit(10**9)
root, pats = sys.argv[1], sys.argv[2:]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        hit = [p for p in pats if p in name]
        if not hit:
            continue
        m = re.search(r"(\w+_kernel<[^>]*>)", name)
        k = m.group(1) if m else hit[0]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        agg[k]["_dur_us"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, cs in agg.items():
    print(k)
    for c, v in sorted(cs.items()):
        print(f"   {c:28s} n={len(v):4d} mean={sum(v)/len(v):16.1f}")
