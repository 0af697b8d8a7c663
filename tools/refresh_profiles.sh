#!/bin/bash
# Regenerates the files under profiles/ on the GPU box: tools/refresh_profiles.sh <tag>   (e.g. r01)
# Run from the repository root through gpurun; every GPU step has its own time limit.
# Afterwards, here: copy gpurun_out/prof_<tag>/* to profiles/<tag>_*, and hbm_traffic.json ALSO to
# profiles/traffic_latest.json (bench.py reports traffic only from a file whose library_fingerprint matches its library).
set -e
tag=${1:-r01}
R=${GRAFT_REPO_ROOT:-$PWD}
out=$R/gpurun_out/prof_$tag
rm -rf $out; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
B="python3 $R/bench.py --steps 1 --warmup 0 --cpu-sample 0 --check 0 --faithful-batch 0"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -- $B > $out/kt.log 2>&1
ILQR_HIP_PARTS=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt1 -- $B > $out/kt1.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- $B > $out/fetch.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- $B > $out/write.log 2>&1
cd $R
python tools/collect_traffic.py $out/fetch $out/write profiles/traffic_latest.json 2
cp profiles/traffic_latest.json $out/hbm_traffic.json
cp $(ls $out/kt/*/*kernel_stats.csv | head -1) $out/kernel_stats.csv
cp $(ls $out/kt1/*/*kernel_stats.csv | head -1) $out/kernel_stats_single_stream.csv
python tools/trace_summary.py $out/kt 3 > $out/kernel_trace_summary.txt
python tools/trace_summary.py $out/kt1 3 > $out/kernel_trace_summary_single_stream.txt
timeout -k 10 600 python bench.py --steps 5 > $out/bench_line.json 2> $out/bench.err
tail -c 1500 $out/bench_line.json
# the same batch under ILQR_FLAG_LAPACK_EIGVALS (three backward launches per iteration and part)
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/ktf -- python3 $R/tools/faithful_probe.py 1048576 0 1 > $out/ktf.log 2>&1
cd $R
cp $(ls $out/ktf/*/*kernel_stats.csv | head -1) $out/kernel_stats_lapack_eigvals.csv
python tools/trace_summary.py $out/ktf 3 > $out/kernel_trace_summary_lapack_eigvals.txt || true
# where the wavefronts' cycles go (single stream; last, so that trouble with a counter name costs nothing else)
cd /tmp
ILQR_HIP_PARTS=1 timeout -s KILL 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES --output-format csv -d $out/sq -- $B > $out/sq.log 2>&1
cd $R
python tools/sq_summary.py $out/sq > $out/sq_counters.txt
cat $out/sq_counters.txt
