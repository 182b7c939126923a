"""Plain device-copy rate by variant and buffer size (mirl_copy_bytes_ex): what this box's HBM path gives a 16 B / lane
copy, next to the guide's 6.29 TB/s "float4 copy" figure.  usage: python tools/copy_probe.py"""
import ctypes as C
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rltime_amd._lib import lib, check

NAMES = {0: "grid-stride, cached", 1: "4 x 16 B per lane, non-temporal loads + stores (512 lanes)", 2: "8 per lane, nt loads + nt stores",
         3: "hipMemcpyAsync D2D", 4: "4 per lane, cached loads + nt stores", 5: "4 per lane, cached loads + stores",
         6: "2 per lane, nt + nt", 7: "8 per lane, cached + cached"}
for mb in (256, 1024, 3363):
    n = mb * (1 << 20)
    a = torch.empty(n, dtype=torch.uint8, device="cuda").random_(0, 256)
    b = torch.empty_like(a)
    st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)          # noqa: E731
    rec = {"MiB": mb}
    for v in sorted(NAMES):
        f = lambda: check(lib.mirl_copy_bytes_ex(C.c_void_p(b.data_ptr()), C.c_void_p(a.data_ptr()), n, v, st()))   # noqa: E731
        for _ in range(3):
            f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            f()
        e1.record()
        torch.cuda.synchronize()
        rec[NAMES[v]] = round(2.0 * n / (e0.elapsed_time(e1) / 10) / 1e6, 1)
        assert torch.equal(a[:4096], b[:4096])
    print(json.dumps(rec), flush=True)
    del a, b
    torch.cuda.empty_cache()
