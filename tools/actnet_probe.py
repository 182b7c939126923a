"""Per-kernel timing of the acting network's own kernels (csrc/actnet.hip) at one rank's acting batch: HIP events around
`iters` back-to-back launches.  python tools/actnet_probe.py [E] [need_q]"""
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rltime_amd._lib import lib, check  # noqa: E402

E = int(sys.argv[1]) if len(sys.argv) > 1 else 32
need_q = (sys.argv[2] != "0") if len(sys.argv) > 2 else True
N, H, D, A, F = 32, 512, 64, 6, 3136
HID = 1024 if need_q else 512
NO = A + (1 if need_q else 0)
dev = "cuda"
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(None)   # noqa: E731
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)                # noqa: E731
torch.manual_seed(0)
y1 = torch.rand(E, 20, 20, 32, device=dev)
w2, b2 = torch.randn(64, 512, device=dev) * 0.05, torch.randn(64, device=dev) * 0.1
y2 = torch.empty(E, 9, 9, 64, device=dev)
w3, b3 = torch.randn(64, 576, device=dev) * 0.05, torch.randn(64, device=dev) * 0.1
xh = torch.randn(E, F + H, device=dev) * 0.3
wcat = torch.randn(4 * H, F + H, device=dev) * 0.02
bias = torch.randn(4 * H, device=dev) * 0.1
c_in, h, c = (torch.randn(E, H, device=dev) * 0.5 for _ in range(3))
freq = (torch.arange(1, D + 1, device=dev, dtype=torch.float32) * np.pi).contiguous()
wq, bq = torch.randn(H, D, device=dev) * 0.125, torch.randn(H, device=dev) * 0.1
wfc, bfc = torch.randn(HID, H, device=dev) * 0.044, torch.randn(HID, device=dev) * 0.1
wout, bout = torch.randn(NO, HID, device=dev) * 0.03, torch.randn(NO, device=dev) * 0.1
parts, pitch = C.c_int32(), C.c_int32()
check(lib.mirl_act_head_parts(HID, NO, C.byref(parts), C.byref(pitch)))
part = torch.zeros(parts.value * E * N * pitch.value, device=dev)
step = torch.tensor([5], dtype=torch.int64, device=dev)
acts = torch.empty(E, dtype=torch.int32, device=dev)
q = torch.empty(E, A, device=dev)
gates = torch.empty(E, 4 * H, device=dev)
xq = torch.randn(E * N, H, device=dev) * 0.3

calls = {
    "conv2": lambda: check(lib.mirl_act_conv_fwd(2, E, 20, 20, p(y1), p(w2), p(b2), p(y2), 9 * 9 * 64, st())),
    "conv3": lambda: check(lib.mirl_act_conv_fwd(3, E, 9, 9, p(y2), p(w3), p(b3), p(xh), F + H, st())),
    "embed": lambda: check(lib.mirl_act_embed(E, N, H, D, p(h), p(freq), None, 99, p(step), p(wq), p(bq), p(xq), None, st())),
    "head_hidden": lambda: check(lib.mirl_act_head_hidden(E * N, H, HID, NO, p(xq), p(wfc), p(bfc), p(wout), p(part), st())),
    "head_select": lambda: check(lib.mirl_act_head_select(E, N, A, parts.value, pitch.value, p(part), p(bout), 1 if need_q else 0, None, None, 0.0,
                                                          99, p(step), p(acts), p(q), st())),
    "lib_lstm_gemm+cell": lambda: (torch.addmm(bias, xh, wcat.t(), out=gates),
                                   check(lib.mirl_lstm_cell_fwd(E, H, p(gates), p(c_in), None, None, None, p(h), p(c), st()))),
}
if lib.mirl_act_lstm_supported(E, H, F + H):
    need = C.c_int64()
    check(lib.mirl_act_lstm_workspace_bytes(E, H, F + H, C.byref(need)))
    ws = torch.zeros((need.value + 3) // 4, dtype=torch.int32, device=dev)
    calls["lstm"] = lambda: check(lib.mirl_act_lstm_fwd(E, H, F + H, p(xh), F + H, p(wcat), p(bias), p(c_in), p(h), p(c), p(ws), st()))
only = os.environ.get("PROBE_ONLY")
res = {"E": E, "need_q": need_q, "dbg": os.environ.get("MIRL_ACT_DBG", "0")}
iters = 300
for name, fn in calls.items():
    if only and name not in only.split(","):
        continue
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    res[name] = round(a.elapsed_time(b) / iters * 1e3, 2)
print(json.dumps(res))
