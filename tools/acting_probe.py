"""Acting-only probe at the config-D shapes (E=256 envs, IQN-LSTM policy, HIP-graph
replay, device ingest): N vector steps, event-timed; run under
`rocprofv3 --kernel-trace --stats` to see which kernels an acting step is made of."""
import json
import sys
import time

import torch

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


class A:
    config, scaling, mbatch, nstep_train, burn_in, nstep_target, envs, replay_size = "iqn_lstm", "weak", None, None, None, None, None, 60000
    train_arg, frame_dedup, no_acting, overlap_acting, no_policy_outputs = [], False, False, "off", False


for arg in sys.argv[1:]:
    if arg.startswith("--envs="):
        A.envs = int(arg.split("=")[1])                 # 32 = one rank's share of the 8-GPU job
    if arg == "--no-policy-outputs":
        A.no_policy_outputs = True

cfg = bench.build_config(A, 0, 1, "strong")
trainer = bench.build_trainer(cfg, torch.device("cuda", 0), use_graph="--eager" not in sys.argv, data_parallel=None)
actors, hist = trainer.actors, trainer.history_buffer
steps = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 200
E = actors.get_env_count()
for _ in range(3):
    hist.update(actors.get_samples(E * 10))
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0 = time.perf_counter()
a.record()
s = actors.get_samples(E * steps)
mid = torch.cuda.Event(enable_timing=True)
mid.record()
hist.update(s)
b.record()
torch.cuda.synchronize()
print(json.dumps({"envs": E, "vector_steps": steps, "acting_us_per_step": a.elapsed_time(mid) / steps * 1e3,
                  "ingest_us_per_step": mid.elapsed_time(b) / steps * 1e3,
                  "host_us_per_step": (time.perf_counter() - t0) / steps * 1e6}))
hist.close()
