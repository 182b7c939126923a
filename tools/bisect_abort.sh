#!/bin/bash
# which predecessor makes test_fused_gpu's first conv backward abort?  every process gets a cold MIOpen cache
i=0
for pre in "tests/test_conv_in_gpu.py tests/test_conv_mid_gpu.py" "tests/test_e2e_gpu.py" "tests/test_fast_acting_gpu.py"; do
  i=$((i+1))
  export MIOPEN_USER_DB_PATH=/tmp/midb$i MIOPEN_CUSTOM_CACHE_DIR=/tmp/micache$i
  mkdir -p $MIOPEN_USER_DB_PATH $MIOPEN_CUSTOM_CACHE_DIR
  echo "== $pre + test_fused"
  timeout 600 python -m pytest $pre tests/test_fused_gpu.py -x -q --timeout 300 -k "not test_nothing" 2>&1 | grep -v "^  File" | tail -6
done
