#!/usr/bin/env python3
"""Benchmark of the rltime Q-learning hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W          # N=1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[3], the configuration the metric is quoted on):
recurrent IQN (conv -> LSTM512 -> FC512, dueling, 32 quantiles, double-Q,
rnn_bootstrap) with prioritized sequence replay, B=512 sequences x T=80 train
steps + 40 burn-in steps, n=2, observations (4,84,84) uint8, synthetic data,
replay of 1M transitions per GPU pre-filled before timing.

One *step* = one pass of THE LOOP body of the reference
(rltime/training/multi_step_trainer.py:245-375) over one batch:
  acting for the transitions the train quota asks for (train_frequency=4 ->
  10 240 per step = 40 vector steps of 256 envs: policy forward replayed from a
  HIP graph, epsilon-greedy, synthetic env step, device-resident ingest)
  -> stratified sum-tree sampling -> sequence gather -> burn-in -> IQN
  double-Q targets -> forward/backward -> grad all-reduce (N>1) -> clip + Adam
  -> update_losses.  (--no-acting feeds pre-generated actor output instead.)
Nothing is skipped inside the timed region.  Weak scaling: every rank owns a
replay shard (its envs) and trains B=512 local sequences; gradients are
all-reduced, importance weights globalised (rltime_amd/parallel.py).

Prints ONE JSON line (rank 0).  `roofline` is for the dominant HIP kernel of the
path, the frame gather: algorithmic bytes = 2*(L+n)*B*F per launch (read once,
write once) over the mean launch duration measured with
HIP events on the launch stream inside the timed region.  `cpu_baseline` is the
oracle (the reference's algorithm class restated, oracle/) timed on this box's
host cores on a bounded sample and scaled linearly (stated in `sample`).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--mbatch", type=int, default=512)
    ap.add_argument("--nstep-train", type=int, default=80)
    ap.add_argument("--burn-in", type=int, default=40)
    ap.add_argument("--nstep-target", type=int, default=2)
    ap.add_argument("--envs", type=int, default=256, help="envs per GPU")
    ap.add_argument("--replay-size", type=int, default=1000000, help="transitions per GPU")
    ap.add_argument("--no-acting", action="store_true", help="feed pre-generated actor output instead of running the device actor's policy forward inside the step")
    ap.add_argument("--no-acting-graph", action="store_true", help="run the acting forward eagerly instead of replaying it from a HIP graph")
    ap.add_argument("--amp", default="none", choices=["none", "bf16"], help="autocast dtype of the network (none = fp32, the parity precision)")
    ap.add_argument("--channels-last", action="store_true", help="NHWC conv stack (experiment)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
    ap.add_argument("--pmc-traffic", default=os.path.join(ROOT, "profiles", "gather_traffic.json"))
    return ap.parse_args()


def build_trainer(args, rank, world, device):
    from rltime_amd.general.config import load_config
    from rltime_amd.general.loggers import NullLogger
    from rltime_amd.general.type_registry import get_registered_type
    from rltime_amd.general.utils import deep_dictionary_update
    from rltime_amd.train import create_actors
    config = load_config("synthetic_atari_iqn_lstm.json")
    deep_dictionary_update(config, {
        "acting": {"actor_envs": args.envs, "env_base": rank * args.envs, "total_envs": world * args.envs},
        "training": {"args": {
            "mbatch_size": args.mbatch, "nstep_train": args.nstep_train,
            "burn_in_timesteps": args.burn_in, "nstep_target": args.nstep_target,
            "warmup_steps": 0, "total_steps": 10 ** 12, "log_freq": 10 ** 12,
            "history_mode": {"args": {"size": args.replay_size, "device_rng": True,
                                      "keep_policy_outputs": False}}}}})
    if args.channels_last:
        config["model"]["args"]["layer_configs"][0]["args"]["channels_last"] = True
    actors = create_actors(config, device, device_acting=True, use_graph=not args.no_acting_graph)
    cls = get_registered_type("trainers", config["training"]["type"])
    trainer = cls(logger=NullLogger(), actors=actors, model_config=config["model"],
                  policy_args=config.get("policy_args", {}))
    trainer.setup(**config["training"]["args"])
    return trainer, config


class SyntheticFeeder:
    """Stands in for Actor.get_samples during the bench: the same DeviceSamples a
    device-resident actor emits (frames, LSTM state, actions, rewards, dones),
    pre-generated in HBM, without the policy forward."""

    def __init__(self, trainer, envs, env_base, device, seed):
        from rltime_amd.acting.acting_interface import DeviceSamples
        self.cls = DeviceSamples
        self.envs, self.env_base = envs, env_base
        g = torch.Generator(device=device).manual_seed(seed)
        H = 512
        self.pool = []
        for _ in range(8):
            u = torch.rand(2, envs, device=device, generator=g)
            self.pool.append(dict(
                frames=torch.randint(0, 256, (envs, 4, 84, 84), dtype=torch.uint8, device=device, generator=g),
                state=torch.randn(envs, 2 * H, device=device, generator=g) * 0.3,
                initials=(u[1] < 0.002).float(),
                actions=torch.randint(0, 6, (envs,), dtype=torch.int32, device=device, generator=g),
                rewards=torch.bucketize(u[0], torch.tensor([0.1, 0.9, 1.0], device=device)).clamp(max=2).float() - 1.0,
                dones=(u[1] < 0.002).to(torch.uint8)))
        self.example = {"x": np.zeros((4, 84, 84), np.uint8), "layer0_state": {},
                        "layer1_state": {"hx": np.zeros(H, np.float32), "cx": np.zeros(H, np.float32),
                                         "initials": np.float32(0)},
                        "layer2_state": {}}
        self.t = 0
        self.count = envs

    def get_env_count(self):
        return self.envs

    def update_state(self, progress, policy_state=None):
        pass

    def get_samples(self, min_samples):
        iters = (max(1, min_samples) + self.envs - 1) // self.envs
        out = self.cls(self.example, self.envs, self.env_base)
        for _ in range(iters):
            self.t += 1
            out.append(**self.pool[self.t % len(self.pool)])
        return out


def cpu_baseline(args, seconds):
    """The oracle — the reference's algorithm class (per-transition records,
    np.stack batch assembly, torch-CPU fwd/bwd with the reference's 1 thread) —
    on a bounded sample of the same workload, scaled linearly to B x T."""
    from oracle import replay as orc
    from oracle import qmath
    from rltime_amd.general.config import load_config
    from rltime_amd.policies.iqn import IQNPolicy
    from rltime_amd.spaces import Box, Discrete
    torch.set_num_threads(1)          # reference: models/torch/torch_model.py:25
    T, P, n = args.nstep_train, args.burn_in, args.nstep_target
    Bs, E, H, A = 4, 8, 512, 6
    config = load_config("synthetic_atari_iqn_lstm.json")
    mk = lambda: IQNPolicy.create(model_config=config["model"], observation_space=Box(0, 255, (4, 84, 84), np.uint8),  # noqa: E731
                                  action_space=Discrete(A), cuda=False, **config["policy_args"])
    policy, target = mk(), mk()
    opt = torch.optim.Adam(policy.parameters(), eps=1e-5)
    buf = orc.OraclePrioritizedReplay(
        size=E * 400, train_frequency=4, nstep_target=n, nstep_train=T, prefix_steps=P,
        alpha=0.9, beta=0.6, max_weight_factor=0.9, discount_function=orc.make_discount(0.99))
    rng = np.random.RandomState(0)
    frame_pool = [rng.randint(0, 256, (4, 84, 84)).astype(np.uint8) for _ in range(64)]
    t0 = time.time()
    fed = 0
    for s in range(300):
        samples = []
        for e in range(E):
            samples.append({
                "policy_output": {"actions": int(rng.randint(A))},
                "next_state": {"x": frame_pool[(s * E + e) % 64].copy(), "layer0_state": {},
                               "layer1_state": {"hx": rng.randn(H).astype(np.float32), "cx": rng.randn(H).astype(np.float32),
                                                "initials": np.float32(rng.rand() < 0.002)},
                               "layer2_state": {}},
                "reward": float(rng.choice([-1.0, 0.0, 1.0], p=[.1, .8, .1])), "done": bool(rng.rand() < 0.002),
                "info": {}, "env_id": e})
        buf.update(samples)
        fed += E
    ingest_rate = fed / (time.time() - t0)
    buf.train_quota = 0

    def flat(x):
        return x.reshape((x.shape[0] * x.shape[1],) + x.shape[2:])

    def tt(tree):
        from rltime_amd.models.torch.utils import make_tensor
        return make_tensor(tree, "cpu")

    steps, t_steps = 0, 0.0
    while t_steps < seconds and steps < 3:
        t1 = time.time()
        batch = buf.get_train_data(Bs, 0.5)
        from oracle.replay import tree_map
        # burn-in (multi_step_trainer.py:90-131)
        for pol, key in ((policy, "states"), (target, "target_states")):
            st = tt(tree_map(batch[key], lambda x: flat(x[:P])))
            with torch.no_grad():
                pol.predict(st, P)
            hx, cx = pol.model.layers[1].last_state
            keep = 1 - torch.from_numpy(batch[key]["layer1_state"]["initials"][P]).unsqueeze(-1)
            batch[key]["layer1_state"]["hx"][P] = (hx * keep).numpy()
            batch[key]["layer1_state"]["cx"][P] = (cx * keep).numpy()
        data = tree_map(batch, lambda x: flat(x[P:]))
        f32 = lambda a: torch.from_numpy(np.asarray(a).astype(np.float32))  # noqa: E731
        with torch.no_grad():
            z_t = target.predict(tt(data["target_states"]), T)[0]
            z_s = policy.predict(tt(data["target_states"]), T)[0]
            y = qmath.nstep_target(qmath.iqn_bootstrap(z_t, z_s), f32(data["returns"]),
                                   f32(data["target_masks"]), f32(data["nsteps"]), 0.99, None)
        opt.zero_grad()
        z, taus = policy.predict(tt(data["states"]), T)
        loss, rep = qmath.iqn_loss(z, taus, torch.from_numpy(data["policy_outputs"]["actions"]), y,
                                   f32(data["extra_data"]["importance_weights"]), 1.0, T, "mean", None)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(policy.parameters(), 40.0)
        opt.step()
        buf.update_losses(data["extra_data"]["loss_indices"], rep.numpy())
        t_steps += time.time() - t1
        steps += 1
    # acting share (actor.py:108-147): policy forward on a 32-env vector step, 1 thread
    EA = 32
    act_state = {"x": rng.randint(0, 256, (EA, 4, 84, 84)).astype(np.uint8), "layer0_state": {},
                 "layer1_state": {"hx": np.zeros((EA, H), np.float32), "cx": np.zeros((EA, H), np.float32),
                                  "initials": np.zeros(EA, np.float32)}, "layer2_state": {}}
    policy.actor_predict(act_state, 1)
    t2 = time.time()
    for _ in range(5):
        policy.actor_predict(act_state, 1)
    act_rate = 5 * EA / (time.time() - t2)
    acted_per_step = args.mbatch * T / 4                        # train_frequency=4
    per_step_full = (t_steps / steps) * (args.mbatch / Bs)
    per_step_full += acted_per_step / ingest_rate + acted_per_step / act_rate
    return {
        "value": args.mbatch * T / per_step_full, "unit": "transitions/s", "cores": 1, "kind": "port",
        "learner_steps_per_sec": 1.0 / per_step_full,
        "sample": "oracle (reference algorithm restated): %d learner steps at B=%d (x%d to B=%d), T=%d, burn-in %d, "
                  "n=%d, torch-CPU fp32 1 thread; + acting %.0f and ingest %.0f transitions/s for the step's %d acted "
                  "transitions; scaled linearly in B" % (steps, Bs, args.mbatch // Bs, args.mbatch, T, P, n, act_rate,
                                                         ingest_rate, acted_per_step)}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node == --gpus"
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    import torch.distributed as dist
    force_dist = bool(os.environ.get("BENCH_FORCE_DIST"))      # exercise the RCCL path on 1 GPU (debug)
    if world > 1 or force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", device_id=device, rank=rank, world_size=world)

    torch.manual_seed(1234 + rank)
    np.random.seed(1234 + rank)
    trainer, config = build_trainer(args, rank, world, device)
    if world > 1 or force_dist:
        from rltime_amd.parallel import DataParallel
        trainer.data_parallel = DataParallel()
        # identical initial weights on every rank
        for p in trainer.policy.parameters():
            dist.broadcast(p.data, 0)
        trainer.sync_target()
    hist = trainer.history_buffer
    feeder = SyntheticFeeder(trainer, args.envs, rank * args.envs, device, seed=99 + rank)
    real_actors = trainer.actors
    trainer.actors = feeder            # pre-fill with pre-generated actor output (no policy forward)

    # ---- pre-fill the replay shard (untimed) ---------------------------------
    t0 = time.time()
    per_call = 64 * args.envs
    fed = 0
    while fed < args.replay_size + args.envs:
        hist.update(feeder.get_samples(per_call))
        fed += per_call
    torch.cuda.synchronize()
    fill_s = time.time() - t0
    hist_stats = hist.stats()
    # the quota accrued during the fill is not training debt of the timed region:
    # start from the steady-state balance so every step feeds exactly its share
    hist.train_quota = 0
    if not args.no_acting:
        trainer.actors = real_actors    # timed steps run the real device actor: policy forward (HIP-graph
        #                                 replay), epsilon-greedy, synthetic env step, DeviceSamples ingest
    amp = torch.autocast("cuda", dtype=torch.bfloat16) if args.amp == "bf16" else None

    def one_step():
        ok = False
        while not ok:
            if amp is not None:
                with amp:
                    ok = trainer.loop_iteration()
            else:
                ok = trainer.loop_iteration()

    for _ in range(args.warmup):
        one_step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    hist.profile(True)
    steps_before = trainer.steps
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    launches, gather_ms = hist.profile(False)
    acted = trainer.steps - steps_before
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        T, P, n, B = args.nstep_train, args.burn_in, args.nstep_target, args.mbatch
        L = T + P
        F = 4 * 84 * 84
        rows = L + n if n < L else 2 * L
        algo_bytes = 2.0 * rows * B * F
        avg_ms = gather_ms / max(launches, 1)
        achieved = algo_bytes / (avg_ms * 1e-3) / 1e9 if launches else None
        traffic = None
        if os.path.isfile(args.pmc_traffic):
            try:
                traffic = json.load(open(args.pmc_traffic)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "sampled transitions/sec (= learner steps/sec x B x T), IQN-LSTM B=512 T=80 84x84x4",
            "value": world * B * T * args.steps / dt,
            "unit": "transitions/s",
            "learner_steps_per_sec": args.steps / dt,
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if args.amp == "none" else "bf16(network autocast)+f32(hot path)",
            "data": "synthetic",
            "config": {
                "workload": "BASELINE configs[3] atari_iqn_lstm: recurrent IQN, prioritized sequence replay",
                "mbatch_per_gpu": B, "nstep_train": T, "burn_in": P, "nstep_target": n,
                "frame": "(4,84,84) u8", "lstm_state": "2x512 f32 per transition",
                "replay_transitions_per_gpu": hist_stats["total_items"],
                "active_sequences_per_gpu": hist_stats["active_sequences"],
                "envs_per_gpu": args.envs, "acted_transitions_per_step_per_gpu": acted / args.steps,
                "acting_policy_forward_in_step": not args.no_acting,
                "acting_forward_hip_graph": (not args.no_acting) and (not args.no_acting_graph),
                "parallelism": "dp%d (replay sharded by env, grad all-reduce)" % world,
                "replay_fill_seconds": round(fill_s, 2)},
            "roofline": {
                "kernel": "k_gather_rows (frames)", "bound": "hbm",
                "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": (achieved / HBM_PEAK_GBPS) if achieved else None,
                "algorithmic_bytes_per_launch": algo_bytes, "avg_launch_ms": avg_ms, "launches": launches,
                "traffic": traffic},
        }
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(args, args.cpu_seconds)
                out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
            except Exception as e:            # the baseline must never sink the GPU number
                out["cpu_baseline"] = {"error": repr(e)}
        print(json.dumps(out), flush=True)
    trainer.actors = real_actors
    hist.close()
    if world > 1 or force_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
