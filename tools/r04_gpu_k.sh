#!/bin/bash
# round 4, session K: kernel trace of the bench -> where is the GPU idle inside a step
set -u
R="$(pwd)"; OUT=gpurun_out/r04k; mkdir -p $OUT
export TMPDIR=/tmp
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d "$R/$OUT/trace" -o bench -- python "$R/bench.py" --steps 6 --warmup 3 --no-cpu-baseline --profile-steps 0 > "$R/$OUT/bench_traced.json" 2> "$R/$OUT/bench_traced.err"); echo "trace rc=$?"
python tools/gap_report.py $OUT/trace 3 > $OUT/gap_report.txt 2>&1; head -120 $OUT/gap_report.txt
find $OUT -name "*kernel_trace.csv" -size +40M -delete
