#!/usr/bin/env python3
"""Shader clock seen by each pipeline kernel: GRBM_GUI_ACTIVE (cycles the GPU was busy during the dispatch) over
the dispatch's duration.  Needs a rocprofv3 --pmc GRBM_GUI_ACTIVE run with --kernel-trace in the same directory:
tools/kernel_clock.py <dir>"""
import csv, glob, re, sys, collections
d = sys.argv[1]
cc = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
rows = list(csv.DictReader(open(cc)))
it = 0
for r in rows:
    m = re.search(r"pipe_(\w+?)_kernel", r["Kernel_Name"])
    if not m or r["Counter_Name"] != "GRBM_GUI_ACTIVE":
        continue
    name = m.group(1)
    if name == "init": it = 0
    if name == "backward": it += 1
    dur = int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) if "End_Timestamp" in r else 0
    if name == "backward" and (it <= 3 or it % 10 == 0):
        cyc = float(r["Counter_Value"])
        print(f"it {it:2d} {name:10s} grid {r['Grid_Size']:>8s} cycles {cyc:12.0f} dur {dur / 1e3:9.1f} us  -> {cyc / dur if dur else 0:.2f} GHz")
