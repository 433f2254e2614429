#!/usr/bin/env python
"""Per-kernel-instance AND per-grid summary of rocprofv3 --pmc counter_collection CSVs (mean per dispatch): the persistent
1x1 kernel runs the K = 256 and the K = 1024 layers under the same name, so dispatches are keyed by (kernel, LDS bytes,
grid) and listed in dispatch order.  usage: pmc_by_shape.py <dir> <substring>"""
import csv, glob, re, sys, collections
csv.field_size_limit(10**9)
root, pat = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(root + "/**/*counter_collection.csv", recursive=True)):
    rows = [r for r in csv.DictReader(open(f)) if pat in r["Kernel_Name"]]
    # dispatches in order; the bench runs shape after shape, pass after pass: key by the order of first appearance of a run
    for r in rows:
        m = re.search(r"(\w+_kernel<[^>]*>)", r["Kernel_Name"])
        key = (m.group(1) if m else pat, r.get("LDS_Block_Size", "?"), r.get("Grid_Size", "?"), r.get("Workgroup_Size", "?"))
        agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        agg[key]["_dur_us"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        agg[key]["_first_dispatch"].append(float(r["Dispatch_Id"]))
for key, cs in sorted(agg.items(), key=lambda kv: min(kv[1]["_first_dispatch"])):
    print(key, "first dispatch", int(min(cs["_first_dispatch"])))
    for c, v in sorted(cs.items()):
        if c != "_first_dispatch":
            print(f"   {c:40s} n={len(v):4d} mean={sum(v)/len(v):16.1f}")
