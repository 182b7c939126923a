#!/bin/bash
set -u
OUT=gpurun_out/r04i; mkdir -p $OUT
export PYTHONPATH=.
timeout 300 python tools/conv_col_probe.py 5120 640 > $OUT/conv_col_probe.jsonl 2> $OUT/conv_col_probe.err; echo "probe rc=$?"; cat $OUT/conv_col_probe.jsonl; tail -3 $OUT/conv_col_probe.err
