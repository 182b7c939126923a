#!/usr/bin/env python3
"""Times the fused target / loss kernels at the BASELINE shape (M = 80*512 rows,
N = N' = 32 quantiles, A = 6 actions) and prints achieved algorithmic GB/s."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rltime_amd.training import qops  # noqa: E402

M, N, A = 40960, 32, 6
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(0)
z, zt, zs = (torch.randn(M, N, A, device=dev, generator=g) for _ in range(3))
taus = torch.rand(M, N, device=dev, generator=g)
act = torch.randint(0, A, (M,), device=dev, generator=g)
ret = torch.randn(M, device=dev, generator=g)
ns = torch.full((M,), 2.0, device=dev)
mk = torch.ones(M, device=dev)
w = torch.rand(M, device=dev, generator=g)
y = qops.q_target_iqn(zt, zs, ret, ns, mk, 0.99, None)


def timed(fn, iters=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


t_target = timed(lambda: qops.q_target_iqn(zt, zs, ret, ns, mk, 0.99, None))
zz = z.clone().requires_grad_(True)
t_loss = timed(lambda: qops.iqn_loss(zz, taus, act, y, w, 1.0, 80, "mean", None))
bytes_target = M * (2 * N * A * 4 + N * 4 + 12)
bytes_loss = M * (2 * N * A * 4 + 2 * N * 4 + 20)
print(json.dumps({"q_target_iqn_us": t_target, "q_target_iqn_GBps": bytes_target / t_target / 1e3,
                  "iqn_loss_fwd_bwd_us(incl. torch sum)": t_loss, "iqn_loss_GBps": bytes_loss / t_loss / 1e3}))
