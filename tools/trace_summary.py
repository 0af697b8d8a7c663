#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV of bench.py: time per pipeline kernel and per solver iteration."""
import csv, re, sys, glob, collections
path = sys.argv[1]
files = glob.glob(path + "/**/*kernel_trace.csv", recursive=True)
rows = list(csv.DictReader(open(files[0])))
pipe = [r for r in rows if "pipe_" in r["Kernel_Name"]]
tot = collections.defaultdict(float); calls = collections.Counter()
it = 0
lines = []
for r in pipe:
    name = re.search(r"pipe_(\w+?)_kernel", r["Kernel_Name"]).group(1)
    # ILQR_FLAG_LAPACK_EIGVALS: the launches that drop long regularisation loops and the launch that runs them
    if name == "backward" and re.search(r"true, true>", r["Kernel_Name"]): name = "backward_lapack"
    elif name == "backward" and re.search(r"true, false>", r["Kernel_Name"]): name = "backward_lapack_long"
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    tot[name] += dur; calls[name] += 1
    if name == "init": it = 0
    if name in ("backward", "backward_lapack"): it += 1
    lines.append((it, name, int(r["Grid_Size_X"]) // 64, dur, r.get("VGPR_Count"), r.get("Accum_VGPR_Count"), r.get("Scratch_Size"), r.get("LDS_Block_Size")))
span = (int(pipe[-1]["End_Timestamp"]) - int(pipe[0]["Start_Timestamp"])) / 1e6
print(f"pipeline span {span:.2f} ms over {len(pipe)} launches; sum of kernel time {sum(tot.values()):.2f} ms")
for k, v in sorted(tot.items(), key=lambda kv: -kv[1]):
    print(f"  {k:20s} calls {calls[k]:4d}  total {v:9.3f} ms  avg {v / calls[k]:8.3f} ms")
show = int(sys.argv[2]) if len(sys.argv) > 2 else 12
for (i, name, waves, dur, vg, ag, sc, lds) in lines:
    if i <= show or i % 10 == 0:
        print(f"it {i:2d} {name:20s} waves {waves:6d} {dur:9.3f} ms  vgpr {vg} agpr {ag} scratch {sc} lds {lds}")
