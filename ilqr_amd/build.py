"""Build recipe of the solver library (hipcc, gfx950): two translation units, one shared library.

ilqr_hip.hip      the C ABI, host code, rollout / evaluation kernels, the acrobot and the 12-state plants
lane_plants.hip   the pipeline for the car and for user-defined plants (one trajectory per lane),
                  compiled with `-mllvm -disable-machine-licm` (csrc/pipeline_host.hpp says why)
"""
from __future__ import annotations

import os
import re
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
# -ffp-contract=on: a*b+c fuses only where one source expression says so, so every kernel that inlines a
# plant function rounds it the same way (costs and rollouts are compared bit for bit between kernels)
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=on", "-fPIC"]
UNITS = (("ilqr_hip.hip", ()), ("lane_plants.hip", ("-mllvm", "-disable-machine-licm")))


def sources():
    out = [os.path.join(dp, f) for dp, _, fs in os.walk(CSRC) for f in fs if f.endswith((".hip", ".hpp", ".h"))]
    out.append(os.path.join(os.path.dirname(_HERE), "include", "ilqr_hip.h"))
    return out


def up_to_date(target, deps):
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(s) <= t for s in deps)


class DeviceCallLeft(RuntimeError):
    pass


def is_kernel_symbol(mangled: str) -> bool:
    """The library's kernels are function templates or functions named <something>_kernel directly inside namespace
    ilqr or the anonymous namespace of ilqr_hip.hip: _ZN4ilqr<len><name>_kernel... / _ZN<anon>...<len><name>_kernel....
    A device helper that merely has `kernel` somewhere in its name or in a template argument does not match."""
    m = re.match(r"_ZN(?:4ilqr|12_GLOBAL__N_1)(\d+)", mangled)
    if not m:
        return False
    n = int(m.group(1))
    name = mangled[m.end():m.end() + n]
    return len(name) == n and name.endswith("_kernel")


def check_everything_inlined(remarks: str, unit: str, may_be_empty: bool = False):
    """Every device function a kernel uses must have been inlined into it.  A kernel's scratch size is then what
    its own body spills; with a callee left out of line the size travels from callee to caller through assembler
    symbols, and builds were seen in which that came out as 0 for a callee with a 676-byte frame (its stack accesses
    then leave the wave's scratch window: a memory aperture violation on the GPU).  The compiler's resource report
    names every function it emitted code for: anything that is not a kernel fails the build."""
    names = re.findall(r"remark: Function Name: (\S+)", remarks)
    if not names and may_be_empty:
        return
    if not names:
        raise DeviceCallLeft(f"{unit}: no resource report from the compiler (is -Rpass-analysis still passed?)")
    left = sorted({n for n in names if not is_kernel_symbol(n)})
    dyn = re.findall(r"Dynamic Stack: True", remarks)
    if left or dyn:
        raise DeviceCallLeft(f"{unit}: device functions left out of line (mark them __forceinline__): {left[:8]}"
                             + (" ; a kernel has a dynamic stack" if dyn else ""))


def build_library(out_so, defines=(), include_dirs=(), hipcc=None, no_kernels_in=()):
    """Compile both translation units with the given -D / -I options and link them into out_so.
    no_kernels_in: units that hold no kernel in this configuration (no resource report is expected from them)."""
    hipcc = hipcc or os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    out_dir = os.path.dirname(out_so)
    os.makedirs(out_dir, exist_ok=True)
    tag = f"{os.path.basename(out_so)}.{os.getpid()}"
    objs, procs, logs = [], [], []
    common = [*HIPCC_FLAGS, *[f"-D{d}" for d in defines], *[f"-I{d}" for d in include_dirs]]
    for src, extra in UNITS:
        obj = os.path.join(out_dir, f".{tag}.{os.path.splitext(src)[0]}.o")
        objs.append(obj)
        # each compiler writes its remarks to a file of its own: through pipes drained one after the other the second
        # compiler would block on a full pipe until the first has finished
        log = open(obj + ".log", "w+")
        logs.append(log)
        procs.append((src, log, subprocess.Popen([hipcc, *common, *extra, "-Rpass-analysis=kernel-resource-usage", "-c",
                                                  "-o", obj, os.path.join(CSRC, src)], stderr=log, text=True)))
    try:
        for src, log, p in procs:
            p.wait()
            log.seek(0)
            err = log.read()
            rest = "\n".join(line for line in err.splitlines() if "-Rpass-analysis=kernel-resource-usage" not in line
                             and not re.match(r"^\s*(\d+ \||\||\^)|^In file included from", line))
            if rest.strip():
                print(rest)
            if p.returncode != 0:
                raise subprocess.CalledProcessError(p.returncode, p.args)
            check_everything_inlined(err, src, may_be_empty=src in no_kernels_in)
        tmp = os.path.join(out_dir, f".{tag}.so")
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-fPIC", "-shared", "-o", tmp, *objs])
        os.replace(tmp, out_so)
    finally:
        for _, _, p in procs:
            if p.poll() is None:
                p.kill()
        for log in logs:
            log.close()
        for o in objs:
            for f in (o, o + ".log"):
                if os.path.exists(f):
                    os.remove(f)
    return out_so
