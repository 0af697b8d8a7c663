#!/usr/bin/env python3
"""Where the wavefronts' cycles go, per pipeline kernel: summary of a rocprofv3 --pmc pass of bench.py with
    SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES
(the cycle counters count quad-cycles; WAIT_ANY = parked on a wait count, WAIT_INST_ANY = ready but not issued,
ACTIVE_INST_ANY = issuing; the three add up to about WAVE_CYCLES).

    tools/sq_summary.py <dir> [iterations, e.g. 1,10,30]
"""
import collections, csv, glob, re, sys

path = sys.argv[1]
show = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 5, 10, 20, 30, 50]
f = glob.glob(path + "/**/*counter_collection.csv", recursive=True)[0]
disp = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    d = disp.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"]})
    d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])


def line(label, d):
    wc = d.get("SQ_WAVE_CYCLES", 0.0) or 1.0
    waves = d.get("SQ_WAVES", 0.0) or 1.0
    g = lambda k: d.get(k, 0.0)
    return (f"{label:28s} waves {int(waves):8d}  cycles/wave {4 * wc / waves:11.0f}  VALU insts/wave {g('SQ_INSTS_VALU') / waves:9.0f}  "
            f"issuing {100 * g('SQ_ACTIVE_INST_ANY') / wc:5.1f} % (VALU {100 * g('SQ_ACTIVE_INST_VALU') / wc:5.1f} %)  "
            f"ready, not issued {100 * g('SQ_WAIT_INST_ANY') / wc:5.1f} %  waiting {100 * g('SQ_WAIT_ANY') / wc:5.1f} %")


tot = collections.OrderedDict()
per_it = []
it = 0
for k in sorted(disp):
    d = disp[k]
    m = re.search(r"pipe_(\w+?)_kernel", d["name"])
    if not m:
        continue
    name = m.group(1)
    lapack_long = name == "backward" and re.search(r"true, false>", d["name"]) is not None
    if name == "backward" and re.search(r"true, (true|false)>", d["name"]):
        name = "backward_lapack_long" if lapack_long else "backward_lapack"
    if name == "init":
        it = 0
    if name in ("backward", "backward_lapack"):
        it += 1
    t = tot.setdefault(name, collections.Counter())
    for c, v in d.items():
        if c != "name":
            t[c] += v
    if name in ("backward", "backward_lapack_long") and it in show:
        per_it.append((it, d))
print("all launches of the run, per kernel")
for name, t in tot.items():
    print("  " + line(name, t))
print("backward kernel by solver iteration (first solve of the run first)")
for it, d in per_it:
    print("  " + line(f"backward, iteration {it}", d))
