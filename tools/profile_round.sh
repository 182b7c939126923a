#!/bin/bash
# Collect the rocprofv3 evidence for one round on the GPU box (run through gpurun):
#   1. kernel-trace + stats of the bench command  -> gpurun_out/prof_<tag>/stats
#   2. PMC passes (own runs, kernel-trace only) for the frame-gather kernel:
#      FETCH_SIZE and WRITE_SIZE cannot share a pass (TCC slots), see MI355X_MICROARCH.md
set -u
TAG="${1:-r01}"
REPO="$(pwd)"
OUT="$REPO/gpurun_out/prof_$TAG"
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
# MIOpen's find phase (first use of every conv shape) runs seconds-long naive
# reference kernels; an un-profiled run first leaves its results in the user
# find-db so that the profiled run below shows the steady state.
timeout 600 python "$REPO/bench.py" --steps 2 --warmup 2 --no-cpu-baseline > "$OUT/bench_prewarm.log" 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o bench -- \
  python "$REPO/bench.py" --steps 5 --warmup 2 --no-cpu-baseline > "$OUT/bench_under_rocprof.log" 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  PROBE_SIZE=131072 PROBE_ITERS=3 PROBE_NT="${PROBE_NT:-0}" timeout 400 rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/pmc_$C" -o probe -- \
    python "$REPO/tools/gather_probe.py" > "$OUT/probe_$C.log" 2>&1
done
cd "$REPO"
find "$OUT" -type f | head -40 > "$OUT/files.txt"
python tools/summarize_prof.py "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
# keep gpurun_out small (64 MiB cap): the raw traces are not needed once summarised
find "$OUT" -name "*kernel_trace.csv" -delete
find "$OUT" -name "*counter_collection.csv" -size +2M -delete
find "$OUT" -name "*.db" -delete
du -sh "$OUT"
