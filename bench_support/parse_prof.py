#!/usr/bin/env python3
"""Summarise rocprofv3 csv output of bench_support/profile*.sh: per-kernel averages."""
import collections, csv, glob, os, sys
d = sys.argv[1]
kern = sys.argv[2] if len(sys.argv) > 2 else "match_kernel"
for f in sorted(glob.glob(os.path.join(d, "*", "*_counter_collection.csv"))):
    acc = collections.defaultdict(list)
    meta = None
    for r in csv.DictReader(open(f)):
        if kern in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta = r
    for k, v in sorted(acc.items()):
        print("%-8s %-24s launches=%d avg=%.6g" % (os.path.basename(os.path.dirname(f)), k, len(v), sum(v) / len(v)))
    if meta and "sq" == os.path.basename(os.path.dirname(f)):
        print("         VGPR_Count=%s SGPR=%s LDS=%s scratch=%s grid=%s" % (meta["VGPR_Count"], meta["SGPR_Count"], meta["LDS_Block_Size"], meta["Scratch_Size"], meta["Grid_Size"]))
for f in glob.glob(os.path.join(d, "*", "*_kernel_stats.csv")):
    for r in csv.DictReader(open(f)):
        if float(r["Percentage"]) > 0.5:
            print("stats    %-60s calls=%s avg_ms=%.3f pct=%s" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e6, r["Percentage"]))
