#!/usr/bin/env python3
"""Do two builds contain the same instructions for a kernel?  Compares the gfx950 assembly of the kernels whose
mangled name contains a pattern between two source trees (labels normalised, directives and comments dropped).

    tools/kernel_isa_diff.py <tree_a> <tree_b> [pattern] [source]   (default: pipe_backward_team_kernel, ilqr_hip.hip)

Used to check that a change elsewhere in the translation unit left a kernel that is not to be touched as it was.
"""
import difflib, os, re, subprocess, sys, tempfile

FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=on", "-S", "--cuda-device-only"]


def assembly(tree, source):
    out = tempfile.NamedTemporaryFile(suffix=".s", delete=False).name
    subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), *FLAGS, "-o", out,
                           os.path.join("ilqr_amd", "csrc", source)], cwd=tree, stderr=subprocess.DEVNULL)
    funcs, name, buf = {}, None, []
    for line in open(out):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            name, buf = m.group(1), []
            funcs[name] = buf
        elif name is not None:
            if line.startswith(".Lfunc_end"):
                name = None
            elif not line.strip().startswith((";", ".")):
                buf.append(re.sub(r"\.LBB\d+_", ".LBB_", line))
    os.unlink(out)
    return funcs


def main():
    a_tree, b_tree = sys.argv[1], sys.argv[2]
    pat = sys.argv[3] if len(sys.argv) > 3 else "pipe_backward_team_kernel"
    src = sys.argv[4] if len(sys.argv) > 4 else "ilqr_hip.hip"
    a, b = assembly(a_tree, src), assembly(b_tree, src)
    # a kernel that gained a trailing `false` template argument since tree_a is the same kernel
    for k in list(b):
        k0 = k.replace("Lb0EEEvNS_4PipeE", "EEvNS_4PipeE")
        if k0 != k and k0 not in b:
            b[k0] = b[k]
    rc = 0
    for k in sorted(a):
        if pat not in k:
            continue
        if k not in b:
            print(k[:70], "missing in", b_tree); rc = 1; continue
        d = [l for l in difflib.unified_diff(a[k], b[k], lineterm="", n=0) if l[:1] in "+-" and l[:3] not in ("+++", "---")]
        from collections import Counter
        plus, minus = Counter(l[1:] for l in d if l[0] == "+"), Counter(l[1:] for l in d if l[0] == "-")
        d = ["+" + l for l in (plus - minus).elements()] + ["-" + l for l in (minus - plus).elements()]
        only_args = all(re.match(r"^[+-]\ts_load_dword\w* s(\d+|\[\d+:\d+\]), s\[\d+:\d+\], 0x[0-9a-f]+\s*$", l) for l in d)
        print(k[:70], len(a[k]), "instructions:", "identical" if not d else
              f"{len(d) // 2} differ" + (" (kernel-argument offsets only)" if only_args else ""))
        rc |= bool(d and not only_args)
    sys.exit(rc)


if __name__ == "__main__":
    main()
