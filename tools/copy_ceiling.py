#!/usr/bin/env python3
"""Device-copy ceiling on this box: the frame gather's byte count (1.76 GB read
+ 1.76 GB written) through a plain and a non-temporal 16 B/lane copy kernel."""
import ctypes as C
import json
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1:
    import torch
    from rltime_amd._lib import lib, check
    n = 122 * 512 * 28224
    a = torch.randint(0, 256, (n,), dtype=torch.uint8, device="cuda")
    b = torch.empty_like(a)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    f = lambda: check(lib.mirl_copy_bytes(C.c_void_p(b.data_ptr()), C.c_void_p(a.data_ptr()), n, st))  # noqa: E731
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        f()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    assert torch.equal(a, b)
    print(json.dumps({"nt": os.environ.get("MIRL_COPY_NT", "0"), "ms": ms, "GBps_rw": 2 * n / ms / 1e6}))
else:
    for nt in ("0", "1"):
        subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, MIRL_COPY_NT=nt))
