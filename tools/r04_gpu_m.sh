#!/bin/bash
# round 4, session M: input layer's weight gradient on the bf16 pipe (k_conv1_u8_wrw_b3): tests, timing, A/B bench
set -u
OUT=gpurun_out/r04m; mkdir -p $OUT
export MIRL_TEST_ARTIFACTS=$OUT PYTHONPATH=.
timeout 900 python -m pytest tests/test_conv_in_gpu.py tests/test_abi.py -m gpu -q --timeout 600 -x > $OUT/pytest.log 2>&1; echo "rc=$?"; grep -E "passed|failed|^E  |^FAILED" $OUT/pytest.log | head -20
python - <<'PY'
import torch, ctypes as C
from rltime_amd._lib import lib, check
n, h, w = 40960, 84, 84
x = torch.randint(0, 256, (n, 4, h, w), dtype=torch.uint8, device="cuda")
dy = torch.randn(n, 32, 20, 20, device="cuda").contiguous(memory_format=torch.channels_last)
y = torch.randn(n, 32, 20, 20, device="cuda").clamp(min=0).contiguous(memory_format=torch.channels_last)
need = C.c_int64(); check(lib.mirl_conv1_u8_wrw_scratch_floats(C.byref(need)))
scratch = torch.empty(need.value, device="cuda"); dw = torch.empty(32, 4, 8, 8, device="cuda"); db = torch.empty(32, device="cuda")
p = lambda t: C.c_void_p(t.data_ptr())
so, sc, sh, sw = dw.stride()
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for pipe in (1, 0):
    check(lib.mirl_conv1_wrw_bf16_set(pipe))
    f = lambda: check(lib.mirl_conv1_u8_wrw_masked(n, h, w, p(x), p(dy), p(y), 1/255., p(scratch), p(dw), so, sc, sh, sw, p(db), st))
    for _ in range(3): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): f()
    b.record(); torch.cuda.synchronize()
    print("wrw masked pipe", pipe, "ms", round(a.elapsed_time(b) / 10, 3))
check(lib.mirl_conv1_wrw_bf16_set(-1))
PY
