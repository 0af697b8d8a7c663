#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection CSV of bench.py per pipeline kernel and solver iteration."""
import csv, re, sys, glob, collections
path = sys.argv[1]
show = set(int(v) for v in sys.argv[2].split(",")) if len(sys.argv) > 2 else {1, 2, 5, 10, 20, 30, 40, 50}
f = glob.glob(path + "/**/*counter_collection.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
disp = collections.OrderedDict()
for r in rows:
    d = disp.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"], "grid": int(r["Grid_Size"])})
    d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
it = 0
for k in sorted(disp):
    d = disp[k]
    m = re.search(r"pipe_(\w+?)_kernel", d["name"])
    if not m:
        continue
    name = m.group(1)
    if name == "init": it = 0
    if name == "backward": it += 1
    if it not in show:
        continue
    waves = d.get("SQ_WAVES", d["grid"] / 64) or 1
    cs = [c for c in d if c not in ("name", "grid", "SQ_WAVES")]
    print(f"it {it:2d} {name:16s} waves {int(waves):6d} " + " ".join(f"{c[3:] if c.startswith('SQ_') else c}/w={d[c] / waves:10.1f}" for c in cs))
