#!/usr/bin/env python3
"""Instruction budget of one backward stage of the headline kernel, from the gfx950 assembly of a build with named
comments at the phase boundaries (-DILQR_PHASE_MARKERS):

    python tools/stage_budget.py [kernel-name-pattern] > profiles/rNN_stage_budget.txt

Counts the instructions between consecutive markers inside pipe_backward_kernel<CarModel, false, false, false>, by
class (VALU fp64 fma / mul / add, other VALU, SALU, LDS, global memory, waits).  Static counts of the straight-line
text: the group scan's body runs once per pass of its `while (pend)` loop (1.2 passes per stage on the headline
batch), the stuck scan's body once per 64 rounds of a trajectory still failing after 8.
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PAT = sys.argv[1] if len(sys.argv) > 1 else "pipe_backward_kernelINS_8CarModelELb0ELb0ELb0E"
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=on", "-S", "--cuda-device-only", "-DILQR_PHASE_MARKERS", *os.environ.get("ILQR_BUDGET_DEFINES", "").split(),
         "-mllvm", "-disable-machine-licm"]


def classify(op):
    if op.startswith(("v_fma_f64", "v_fmac_f64")):
        return "fma64"
    if op.startswith("v_mul_f64"):
        return "mul64"
    if op.startswith("v_add_f64"):
        return "add64"
    if op.startswith(("v_max_f64", "v_min_f64", "v_cmp", "v_cndmask", "v_rcp_f64", "v_div", "v_sqrt", "v_rsq", "v_trig", "v_fract",
                      "v_ldexp", "v_frexp")):
        return "valu_other64"
    if op.startswith("v_accvgpr"):
        return "agpr_move"
    if op.startswith("v_"):
        return "valu_misc"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    out = tempfile.NamedTemporaryFile(suffix=".s", delete=False).name
    subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), *FLAGS, "-o", out,
                           os.path.join(ROOT, "ilqr_amd", "csrc", "lane_plants.hip")], stderr=subprocess.DEVNULL)
    inside, mark = False, None
    order, counts = [], {}
    for line in open(out):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            inside = PAT in m.group(1)
            mark = "prologue"
            continue
        if not inside:
            continue
        if line.startswith(".Lfunc_end"):
            inside = False
            continue
        t = line.strip()
        mm = re.match(r"^; ILQR_MARK (\w+)", t)
        if mm:
            mark = mm.group(1)
            continue
        if not t or t.startswith((";", ".")) or t.endswith(":"):
            continue
        op = t.split()[0]
        if mark not in counts:
            counts[mark] = collections.Counter()
            order.append(mark)
        counts[mark][classify(op)] += 1
    os.unlink(out)
    cols = ["fma64", "mul64", "add64", "valu_other64", "valu_misc", "agpr_move", "lds", "vmem", "salu", "wait", "other"]
    print("instructions after each marker up to the next one (static text of %s)" % PAT)
    print("%-14s %6s  " % ("phase", "total") + " ".join("%12s" % c for c in cols))
    for k in order:
        c = counts[k]
        print("%-14s %6d  " % (k, sum(c.values())) + " ".join("%12d" % c[x] for x in cols))


if __name__ == "__main__":
    main()
