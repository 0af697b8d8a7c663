#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc passes over `bench.py --steps 1 --warmup 0` (one with FETCH_SIZE, one with
WRITE_SIZE, --output-format csv) into profiles/traffic_latest.json, which bench.py reports as
roofline.traffic.

    python tools/collect_traffic.py <fetch_dir> <write_dir> <out.json> [solves]

Bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: both counters are in KiB, and on gfx950 FETCH_SIZE counts
half the bytes of 16-byte-per-lane streaming reads, which is how every solver kernel reads its records.
The bench process runs `solves` whole-batch solves in this mode (default 2: the timed step and the per-phase
timing step); per-solve figures are the totals divided by that number.
"""
import collections
import csv
import glob
import json
import re
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def per_kernel(path, counter):
    f = glob.glob(path + "/**/*counter_collection.csv", recursive=True)[0]
    tot = collections.defaultdict(float)
    launches = collections.Counter()
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        m = re.search(r"pipe_(\w+?)_kernel", r["Kernel_Name"])
        if not m:
            continue
        tot[m.group(1)] += float(r["Counter_Value"])
        launches[m.group(1)] += 1
    return tot, launches


def main():
    fetch, launches = per_kernel(sys.argv[1], "FETCH_SIZE")
    write, _ = per_kernel(sys.argv[2], "WRITE_SIZE")
    solves = int(sys.argv[4]) if len(sys.argv) > 4 else 2
    kib = {k: (2.0 * fetch.get(k, 0.0) + write.get(k, 0.0)) for k in set(fetch) | set(write)}
    b = {k: v * 1024.0 / solves for k, v in kib.items()}
    ls = sum(b.get(k, 0.0) for k in ("forward1", "backtrack_group", "forward_level", "commit"))
    out = {
        "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), bytes = (2*FETCH_SIZE + WRITE_SIZE) KiB",
        "solves_in_profile": solves,
        "library_fingerprint": __import__("bench").library_fingerprint(),
        "per_kernel_hbm_bytes_per_solve": b,
        "per_kernel_fetch_kib_raw": dict(fetch), "per_kernel_write_kib_raw": dict(write),
        "backward_hbm_bytes_per_solve": b.get("backward", 0.0),
        "line_search_hbm_bytes_per_solve": ls,
        "solve_hbm_bytes": sum(b.values()),
    }
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    print(json.dumps({k: out[k] for k in ("backward_hbm_bytes_per_solve", "line_search_hbm_bytes_per_solve", "solve_hbm_bytes")}))


if __name__ == "__main__":
    main()
