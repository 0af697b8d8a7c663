#!/bin/bash
# A/B runs of bench.py over several builds of the library: tools/ab_bench.sh <parts> lib1.so lib2.so ...
parts=$1; shift
for lib in "$@"; do
  ILQR_HIP_LIB=$PWD/$lib ILQR_HIP_PARTS=$parts timeout -k 10 300 python bench.py --steps 3 --cpu-sample 0 --check 0 --faithful-batch 0 > gpurun_out/ab.json 2> gpurun_out/ab.err || { echo "$lib FAILED"; tail -3 gpurun_out/ab.err; exit 1; }
  python - "$lib" "$parts" <<'PY'
import json, sys
d = json.load(open("gpurun_out/ab.json"))
print(sys.argv[1], "parts", sys.argv[2], "%.2f ms" % d["ms_per_step"], {k: round(v, 1) for k, v in d["roofline"]["phase_ms"].items()}, d["exit_counts"]["linalg"])
PY
done
