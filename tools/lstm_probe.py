#!/usr/bin/env python3
"""Time one LSTM sweep (T steps, B sequences, H units) two ways: persistent kernel
(csrc/lstm_seq.hip) and rocBLAS GEMM + cell kernel per step.
Prints one JSON line per variant: microseconds per time step (HIP events, 10 sweeps)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rltime_amd.models.torch import lstm_seq   # noqa: E402


def main():
    shapes = [(80, 512, 512), (40, 512, 512), (80, 64, 512), (80, 256, 512)]
    if len(sys.argv) > 1:
        shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
    for T, B, H in shapes:
        g = torch.Generator(device="cuda").manual_seed(0)
        gx = torch.randn(T, B, 4 * H, device="cuda", generator=g)
        w = (torch.randn(4 * H, H, device="cuda", generator=g) / H ** 0.5).contiguous()
        h0, c0 = torch.randn(B, H, device="cuda", generator=g), torch.randn(B, H, device="cuda", generator=g)
        keep = (torch.rand(T, B, device="cuda", generator=g) > 0.01).float()
        lstm_seq._BWD_PERSISTENT_MAX_B = 1 << 20
        for name, pers in (("persistent", True), ("gemm+cell", False)):
            for need_grad in (False, True):
                lstm_seq._PERSISTENT = pers
                if pers and not lstm_seq.persistent_supported(T, B, H):
                    continue
                gates = gx.clone()
                for _ in range(3):
                    gates.copy_(gx)
                    lstm_seq._forward_sweep(gates, w, h0, c0, keep, need_grad)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                reps, total = 10, 0.0
                for _ in range(reps):
                    gates.copy_(gx)
                    e0.record()
                    lstm_seq._forward_sweep(gates, w, h0, c0, keep, need_grad)
                    e1.record()
                    torch.cuda.synchronize()
                    total += e0.elapsed_time(e1)
                ms = total / reps
                if need_grad and pers and lstm_seq.lib.mirl_lstm_seq_bwd_supported(T, B, H):
                    out, hm, cm, c_all, _, _ = lstm_seq._forward_sweep(gates, w, h0, c0, keep, True)
                    d_out = torch.randn_like(out)
                    saved = gates.clone()
                    tot = 0.0
                    for _ in range(reps + 2):
                        dg = saved.clone()
                        e0.record()
                        lstm_seq._backward_sweep(dg, c_all, cm, d_out, keep, w)
                        e1.record()
                        torch.cuda.synchronize()
                        tot += e0.elapsed_time(e1) if _ >= 2 else 0.0
                    print(json.dumps({"variant": name + " BACKWARD", "T": T, "B": B, "H": H, "ms_per_sweep": round(tot / reps, 4),
                                      "us_per_step": round(tot / reps * 1e3 / T, 2)}), flush=True)
                print(json.dumps({"variant": name, "need_grad": need_grad, "T": T, "B": B, "H": H,
                                  "ms_per_sweep": round(ms, 4), "us_per_step": round(ms * 1e3 / T, 2),
                                  "f32_mfma_floor_us_per_step": round(2.0 * B * H * 4 * H / 157.3e12 * 1e6, 2)}), flush=True)


if __name__ == "__main__":
    main()
