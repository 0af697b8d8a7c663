#!/usr/bin/env python3
"""Timeline of one stream's kernels late in a solve, from a rocprofv3 --kernel-trace CSV:
tools/tail_timeline.py <dir> [first_backward_index] [count]"""
import csv, glob, re, sys, collections
path = sys.argv[1]
first = int(sys.argv[2]) if len(sys.argv) > 2 else 30
cnt = int(sys.argv[3]) if len(sys.argv) > 3 else 3
f = glob.glob(path + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "pipe_" in r["Kernel_Name"]]
byq = collections.defaultdict(list)
for r in rows:
    byq[r["Queue_Id"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.search(r"pipe_(\w+?)_kernel", r["Kernel_Name"]).group(1),
                               int(r["Grid_Size_X"]) if "Grid_Size_X" in r else int(r.get("Grid_Size", 0))))
for q, ks in sorted(byq.items()):
    ks.sort()
    # the last solve on this queue: after the last init
    li = max(i for i, k in enumerate(ks) if k[2] == "init")
    ks = ks[li:]
    nb = 0
    t_prev = None
    print(f"queue {q}: {len(ks)} launches in the last solve, span {(ks[-1][1] - ks[0][0]) / 1e6:.2f} ms")
    for s, e, name, grid in ks:
        if name == "backward":
            nb += 1
        if first <= nb < first + cnt:
            gap = (s - t_prev) / 1e3 if t_prev is not None else 0.0
            print(f"   it {nb:2d} {name:16s} grid {grid:8d}  gap {gap:7.1f} us  dur {(e - s) / 1e3:8.1f} us")
        t_prev = e
