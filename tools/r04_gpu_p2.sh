#!/bin/bash
set -u
OUT=gpurun_out/r04p; mkdir -p $OUT
export MIRL_TEST_ARTIFACTS=$OUT
timeout 900 python -m pytest tests/test_conv_mid_gpu.py tests/test_network_ab_gpu.py tests/test_e2e_gpu.py tests/test_conv3_gpu.py tests/test_conv_col_gpu.py tests/test_conv_wrw_gpu.py -m gpu -q --timeout 600 > $OUT/pytest3.log 2>&1; echo "rc=$?"; grep -E "passed|failed|^FAILED" $OUT/pytest3.log | head -30
