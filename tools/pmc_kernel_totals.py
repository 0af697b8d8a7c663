#!/usr/bin/env python3
"""Per-kernel totals of a rocprofv3 --pmc pass (counter_collection CSV): counters per wavefront.
    tools/pmc_kernel_totals.py <dir> [substring of kernel name]"""
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
want = sys.argv[2] if len(sys.argv) > 2 else ""
tot = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    if want not in r["Kernel_Name"]:
        continue
    d = tot.setdefault(r["Kernel_Name"][:90], collections.OrderedDict())
    d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
for k, d in tot.items():
    w = d.get("SQ_WAVES", 1.0) or 1.0
    print(k)
    print("   ", "  ".join(f"{c}/wave={v / w:.0f}" for c, v in d.items()))
