#!/usr/bin/env python3
"""Register, scratch and LDS use of every kernel of the library, from the compiler's own resource report
(hipcc -Rpass-analysis=kernel-resource-usage on the two translation units, the flags of ilqr_amd/build.py).

    python tools/kernel_resources.py [> profiles/rNN_kernel_resources.txt]
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ilqr_amd import build as recipe  # noqa: E402


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout
    return out.splitlines()


def main():
    rows = []
    for src, extra in recipe.UNITS:
        cmd = ["/opt/rocm/bin/hipcc", *recipe.HIPCC_FLAGS, *extra, "-Rpass-analysis=kernel-resource-usage", "-c", "-o",
               "/dev/null", os.path.join(recipe.CSRC, src)]
        err = subprocess.run(cmd, capture_output=True, text=True).stderr
        cur = None
        for line in err.splitlines():
            m = re.search(r"remark: Function Name: (\S+)", line)
            if m:
                cur = dict(name=m.group(1), unit=src)
                rows.append(cur)
                continue
            m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\S+) \[-Rpass", line)
            if m and cur is not None:
                cur[m.group(1).strip()] = m.group(2)
    names = demangle([r["name"] for r in rows])
    print("%-6s %-5s %-5s %-5s %-8s %-7s %-4s  %s" % ("VGPRs", "AGPRs", "SGPRs", "spill", "scratchB", "LDS B", "occ", "kernel"))
    for r, n in zip(rows, names):
        n = re.sub(r"\(anonymous namespace\)::", "", n)
        n = re.sub(r"ilqr::", "", n)
        n = re.sub(r"QuadraticRK4Model<4, 1,.*?>\s*>", "AcrobotModel>", n)
        n = re.sub(r"QuadraticRK4Model<12, 4,.*?>\s*>", "QuadEAModel>", n)
        n = re.sub(r"QuadraticRK4Model<4, 1,[^()]*", "AcrobotModel...", n)
        n = re.sub(r"QuadraticRK4Model<12, 4,[^()]*", "QuadEAModel...", n)
        print("%-6s %-5s %-5s %-5s %-8s %-7s %-4s  %s" % (r.get("VGPRs"), r.get("AGPRs"), r.get("TotalSGPRs"),
              r.get("VGPRs Spill", "0") + "/" + r.get("SGPRs Spill", "0"), r.get("ScratchSize"), r.get("LDS Size"),
              r.get("Occupancy"), n[:150]))


if __name__ == "__main__":
    main()
