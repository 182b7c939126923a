"""Launch-shape sweep of the CNN input conversion (csrc/convert.hip) at the
config-D block size: tiles per workgroup x non-temporal loads/stores.  JSON lines."""
import ctypes as C
import json
import sys

import torch

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rltime_amd._lib import lib, check  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 62464
x = torch.randint(0, 256, (N, 4, 84, 84), dtype=torch.uint8, device="cuda")
out = torch.empty((N, 84, 84, 4), dtype=torch.float32, device="cuda")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
bytes_ = N * 4 * 7056 * 5.0
ONLY_DEFAULT = len(sys.argv) > 2 and sys.argv[2] == "default"
for per_wg in ((1,) if ONLY_DEFAULT else (1, 2, 4, 7)):
    for flags in ((1,) if ONLY_DEFAULT else (0, 1, 2, 3)):
        times = []
        for it in range(6):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            check(lib.mirl_frames_to_f32_nhwc_ex(N, 4, 7056, C.c_void_p(x.data_ptr()), 1.0 / 255.0,
                                                 C.c_void_p(out.data_ptr()), per_wg, flags, st))
            b.record()
            torch.cuda.synchronize()
            times.append(a.elapsed_time(b))
        best = min(times[1:])
        print(json.dumps({"frames": N, "tiles_per_wg": per_wg, "nt_loads": not (flags & 1), "nt_stores": not (flags & 2),
                          "ms": round(best, 4), "GBps": round(bytes_ / best / 1e6, 1), "frac_of_8TBps": round(bytes_ / best / 1e6 / 8000, 4)}))
