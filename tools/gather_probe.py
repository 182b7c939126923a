#!/usr/bin/env python3
"""Micro-probe of the replay gather / ingest kernels at the BASELINE shapes
(B=512, T=80, burn-in 40, n=2, 84x84x4 u8 + 2x512 f32 LSTM state).  Prints
algorithmic GB/s per variant; used while tuning, bench.py is the judged number."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(nt, size, E, B, T, P, n, iters):
    os.environ["MIRL_GATHER_NT"] = str(nt)
    from rltime_amd.history import PrioritizedReplayHistoryBuffer
    H = 512
    buf = PrioritizedReplayHistoryBuffer(
        size=size, train_frequency=4, nstep_target=n, nstep_train=T, prefix_steps=P,
        alpha=0.9, beta=0.6, gamma=0.997, device_rng=True, keep_policy_outputs=False)
    ex = {"x": np.zeros((4, 84, 84), np.uint8),
          "layer1_state": {"hx": np.zeros(H, np.float32), "cx": np.zeros(H, np.float32),
                           "initials": np.float32(0)}}
    buf.configure(ex, num_envs=E)
    dev = buf.device
    g = torch.Generator(device=dev).manual_seed(0)
    frames = torch.randint(0, 256, (E, 4, 84, 84), dtype=torch.uint8, device=dev, generator=g)
    state = torch.randn(E, 2 * H, device=dev, generator=g)
    init = torch.zeros(E, device=dev)
    act = torch.zeros(E, dtype=torch.int32, device=dev)
    rew = torch.ones(E, device=dev)
    done = torch.zeros(E, dtype=torch.uint8, device=dev)
    steps = size // E + 8
    torch.cuda.synchronize()
    t0 = time.time()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        buf.update_batch(frames, act, rew, done, state=state, initials=init)
    e1.record()
    torch.cuda.synchronize()
    ing_ms = e0.elapsed_time(e1) / steps
    ing_bytes = E * (4 * 84 * 84 + 2 * H * 4 + 16) * 2
    out = {"nt": nt, "ingest_us_per_vector_step": ing_ms * 1e3,
           "ingest_GBps_rw": ing_bytes / ing_ms / 1e6, "fill_wall_s": time.time() - t0}
    # whole get_train_data
    for _ in range(3):
        batch = buf.get_train_data(B, 0.5)
    torch.cuda.synchronize()
    buf.profile(True)
    e0.record()
    for _ in range(iters):
        batch = buf.get_train_data(B, 0.5)
    e1.record()
    torch.cuda.synchronize()
    kn, kms = buf.profile(False)
    out["frames_kernel_ms"] = kms / max(kn, 1)
    out["frames_kernel_GBps"] = 2 * (T + P + n) * B * 4 * 84 * 84 / (kms / max(kn, 1)) / 1e6
    ms = e0.elapsed_time(e1) / iters
    L = T + P
    frame_bytes = 2 * (L + n) * B * 4 * 84 * 84
    state_bytes = 2 * (L + n) * B * 2 * H * 4
    out.update({"get_train_data_ms": ms, "frames_GB": frame_bytes / 1e9,
                "algo_GBps_frames_only": frame_bytes / ms / 1e6,
                "algo_GBps_frames_plus_state": (frame_bytes + state_bytes) / ms / 1e6})
    # update_losses
    idx = batch["extra_data"]["loss_indices"][P:].reshape(-1, 2)
    losses = torch.rand(idx.shape[0], device=dev)
    for _ in range(3):
        buf.update_losses(idx, losses)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        buf.update_losses(idx, losses)
    e1.record()
    torch.cuda.synchronize()
    out["update_losses_us"] = e0.elapsed_time(e1) / iters * 1e3
    # plain device copy of the same byte count for the measured HBM ceiling
    a = torch.empty(frame_bytes // 2, dtype=torch.uint8, device=dev)
    b = torch.empty_like(a)
    from rltime_amd._lib import lib, check
    import ctypes as C
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for _ in range(3):
        check(lib.mirl_copy_bytes(C.c_void_p(b.data_ptr()), C.c_void_p(a.data_ptr()), a.numel(), st))
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        check(lib.mirl_copy_bytes(C.c_void_p(b.data_ptr()), C.c_void_p(a.data_ptr()), a.numel(), st))
    e1.record()
    torch.cuda.synchronize()
    out["plain_copy_GBps_rw"] = frame_bytes / (e0.elapsed_time(e1) / iters) / 1e6
    e0.record()
    for _ in range(iters):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    out["torch_copy_GBps_rw"] = frame_bytes / (e0.elapsed_time(e1) / iters) / 1e6
    buf.close()
    del a, b
    torch.cuda.empty_cache()
    return out


def sweep():
    size = int(os.environ.get("PROBE_SIZE", 262144))
    combos = [dict(V=0, O=0), dict(V=1, O=0), dict(V=1, O=1)]
    for c in combos:
        os.environ["MIRL_GATHER_VARIANT"] = str(c["V"])
        os.environ["MIRL_GATHER_ORDER"] = str(c["O"])
        r = run(1, size, 256, 512, 80, 40, 2, 10)
        print(json.dumps({"combo": c, "get_train_data_ms": r["get_train_data_ms"],
                          "frames_kernel_ms": r["frames_kernel_ms"],
                          "frames_kernel_GBps": r["frames_kernel_GBps"]}), flush=True)


if __name__ == "__main__":
    if os.environ.get("PROBE_SWEEP"):
        sweep()
        sys.exit(0)
    size = int(os.environ.get("PROBE_SIZE", 262144))
    iters = int(os.environ.get("PROBE_ITERS", 10))
    nts = [int(os.environ["PROBE_NT"])] if "PROBE_NT" in os.environ else [0, 1]
    for nt in nts:
        print(json.dumps(run(nt, size, 256, 512, 80, 40, 2, iters)), flush=True)
