#!/usr/bin/env python3
"""Benchmark of the rltime Q-learning hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W          # any N: starts its own N ranks when not under torchrun
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
    python bench.py --config {iqn_lstm,rainbow_iqn,dqn_uniform} [--scaling {both,strong,weak}]
    python bench.py --gpus N --dry-launch                  # print the launch command only

Workloads (`--config`, default = the one BASELINE.json's
metric is quoted on):
  iqn_lstm     BASELINE configs[3]: recurrent IQN (conv -> LSTM512 -> FC512, dueling,
               32 quantiles, double-Q, rnn_bootstrap), prioritized sequence replay,
               B=512 sequences x T=80 train steps + 40 burn-in steps, n=2, 256 envs
  rainbow_iqn  BASELINE configs[2]: IQN (dueling, 3-step), prioritized replay with
               the 2^20-leaf sum tree, B=512, T=1, 32 envs
  dqn_uniform  BASELINE configs[1]: DQN + uniform replay, B=256, T=1, n=1, 32 envs
all on (4,84,84) uint8 synthetic frames with a replay of 1M transitions per GPU
pre-filled before timing, fp32 (the reference's precision).

One *step* = one pass of THE LOOP body of the reference
(rltime/training/multi_step_trainer.py:245-375) over one batch:
  acting for the transitions the train quota asks for (policy forward replayed from
  a HIP graph, epsilon-greedy, synthetic env step, device-resident ingest)
  -> sampling (stratified sum-tree / uniform) -> gather -> burn-in -> targets
  -> forward/backward -> grad all-reduce (N>1) -> clip + Adam -> update_losses.
Nothing is skipped inside the timed region.

Multi-GPU (`--scaling`): every rank owns a replay shard (its envs).  The headline
at N>1 is STRONG scaling: the configured batch (global B=512),
envs (256) and replay size (1M) are whole-job values split over the ranks, so
learner steps/s is the same job at every N.  The default `both` then runs the WEAK
mode too (every rank trains the configured batch on its own full-size shard) and
reports it as the `weak` sub-record of the same line.  Gradients are all-reduced
(RCCL; the `rccl` record says how many ranks the collective saw and what it cost
per step), importance weights globalised (rltime_amd/parallel.py).

Prints ONE JSON line (rank 0).  `value` comes from the wall time of exactly K steps
between barriers (max over ranks); `step_ms` holds median / p10 / p90 of the K
per-step device times (HIP events).  `roofline` is the dominant HIP kernel of the
path, the frame gather: algorithmic bytes per launch over the mean launch duration
measured with HIP events on the launch stream inside the timed region.
`roofline_all` lists every librltime_hip kernel (events around every launch, in a
separate short pass after the timed region so that ~1500 event records per step do
not perturb `value`).  `cpu_baseline` is the oracle (the reference's algorithm class
restated, oracle/) timed on this box's host cores on a bounded sample.
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
BF16_MFMA_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA peak (MI355X_MICROARCH.md; AMD's 5 PF headline is 2:1 sparse)
F32_MFMA_PEAK_TFLOPS = 157.3   # dense f32-input MFMA peak (MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 / 32x32x2_f32)

CONFIGS = {
    "iqn_lstm": dict(
        file="synthetic_atari_iqn_lstm.json", envs=256,
        workload="BASELINE configs[3] atari_iqn_lstm: recurrent IQN, prioritized sequence replay",
        metric="sampled transitions/sec (= learner steps/sec x B x T), IQN-LSTM B=512 T=80 84x84x4"),
    "rainbow_iqn": dict(
        file="synthetic_atari_rainbow_iqn.json", envs=32, train_args={"graph_learner_step": True},
        workload="BASELINE configs[2] Rainbow-style IQN: dueling, 3-step, prioritized replay (2^20-leaf sum tree)",
        metric="sampled transitions/sec (= learner steps/sec x B), Rainbow-IQN B=512 T=1 84x84x4"),
    "dqn_uniform": dict(
        file="synthetic_atari_dqn.json", envs=32, train_args={"graph_learner_step": True},
        workload="BASELINE configs[1] DQN + uniform replay, 1M buffer",
        metric="sampled transitions/sec (= learner steps/sec x B), DQN uniform replay B=256 T=1 84x84x4"),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="iqn_lstm", choices=sorted(CONFIGS))
    ap.add_argument("--scaling", default="both", choices=["both", "strong", "weak"],
                    help="N>1: both = strong-scaling headline + a `weak` sub-record in the same line")
    ap.add_argument("--dry-launch", action="store_true", help="print the N-rank launch command as JSON and exit")
    ap.add_argument("--master-port", type=int, default=None)
    ap.add_argument("--share-gpu", action="store_true", help="launcher check on a box with fewer GPUs than ranks: ranks share "
                    "the visible GPUs (rank %% device_count) and talk over gloo (RCCL refuses two ranks per device); not a scaling number")
    ap.add_argument("--mbatch", type=int, default=None, help="override mbatch_size (whole job under --scaling strong)")
    ap.add_argument("--nstep-train", type=int, default=None)
    ap.add_argument("--burn-in", type=int, default=None)
    ap.add_argument("--nstep-target", type=int, default=None)
    ap.add_argument("--envs", type=int, default=None, help="envs (per GPU when weak, whole job when strong)")
    ap.add_argument("--replay-size", type=int, default=1000000, help="transitions (per GPU when weak, whole job when strong)")
    ap.add_argument("--no-acting", action="store_true", help="feed pre-generated actor output instead of running the device actor's policy forward inside the step")
    ap.add_argument("--overlap-acting", default="off", choices=["auto", "on", "off"],
                    help="run the acting + ingest of iteration k+1 on a second HIP stream against iteration k's training "
                         "(MultiStepTrainer overlap_acting: the actor's weights are one learner step old, like the reference's async "
                         "actors).  off (default): the HEADLINE is the same algorithm at every N — the synchronous actor of "
                         "rltime/acting/actor.py:108-147; at N > 1 the overlapped schedule is measured as the `overlapped_acting` "
                         "sub-record when a rank trains at most 8192 rows per step (where it pays).  auto: on for such ranks "
                         "(the round-3 headline at N = 8); on: always")
    ap.add_argument("--no-policy-outputs", action="store_true", help="replay without the actor's q-values (keep_policy_outputs=False: "
                    "the acting head then skips the dueling value stream).  Default: stored with every transition like the "
                    "reference (acting_interface.py:58-90, policies/torch/dqn.py:143-148), although training never reads them")
    ap.add_argument("--no-acting-graph", action="store_true", help="run the acting forward eagerly instead of replaying it from a HIP graph")
    ap.add_argument("--amp", default="none", choices=["none", "bf16"], help="autocast dtype of the network (none = fp32, the parity precision)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=30.0)
    ap.add_argument("--no-cpu-linearity-check", dest="cpu_linearity_check", action="store_false",
                    help="cpu_baseline: skip the one extra 1-thread learner step at B=64 (~20 s) that checks the linear-in-B "
                         "extrapolation from B=16 the headline cpu_baseline figure uses (on by default)")
    ap.add_argument("--cpu-linearity-check", dest="cpu_linearity_check", action="store_true", help=argparse.SUPPRESS)
    ap.set_defaults(cpu_linearity_check=True)
    ap.add_argument("--no-other-configs", action="store_true", help="default N=1 run: skip the `other_configs` sub-records "
                    "(BASELINE configs[1] DQN + uniform replay B=256 and configs[2] Rainbow-IQN B=512, each with its own "
                    "gather roofline and CPU baseline)")
    ap.add_argument("--other-steps", type=int, default=200, help="timed learner steps of each `other_configs` run (20 warm-up)")
    ap.add_argument("--profile-steps", type=int, default=3, help="extra steps after the timed region with per-kernel HIP events (roofline_all); 0 = skip")
    ap.add_argument("--miopen-find", action="store_true", help="torch.backends.cudnn.benchmark = True: MIOpen benchmarks its solvers "
                    "per conv shape instead of taking the immediate-mode pick (experiment)")
    ap.add_argument("--frame-dedup", action="store_true", help="SURVEY 8(f)2: stack-consistent synthetic frames and de-duplicated "
                    "storage (one 84x84 plane per transition in the ring, stacks rebuilt by the gather)")
    ap.add_argument("--train-arg", action="append", default=[], help="KEY=JSON extra training.args override (experiments)")
    ap.add_argument("--pmc-traffic", default=os.path.join(ROOT, "profiles", "gather_traffic.json"))
    return ap.parse_args()


def build_config(args, rank, world, scaling, overlap=None):
    from rltime_amd.general.config import load_config
    from rltime_amd.general.utils import deep_dictionary_update
    from rltime_amd.parallel import shard_config
    spec = CONFIGS[args.config]
    config = load_config(spec["file"])
    targs = {"warmup_steps": 0, "total_steps": 10 ** 12, "log_freq": 10 ** 12,
             "history_mode": {"args": {"size": args.replay_size, "device_rng": True,
                                      "keep_policy_outputs": not getattr(args, "no_policy_outputs", False)}}}
    for key, val in (("mbatch_size", args.mbatch), ("nstep_train", args.nstep_train),
                     ("burn_in_timesteps", args.burn_in), ("nstep_target", args.nstep_target)):
        if val is not None:
            targs[key] = val
    targs.update(spec.get("train_args", {}))        # T = 1 configs: the learner step from a captured HIP graph
    for kv in args.train_arg:
        k, v = kv.split("=", 1)
        targs[k] = json.loads(v)
    if args.frame_dedup:
        targs["history_mode"]["args"]["frame_stack_dedup"] = True
        config.setdefault("env_args", {})["frame_stack"] = True
    deep_dictionary_update(config, {"acting": {"actor_envs": args.envs or spec["envs"]}, "training": {"args": targs}})
    config = shard_config(config, rank, world, scaling)
    ta = config["training"]["args"]
    if "overlap_acting" not in targs and not args.no_acting:
        rows = int(ta.get("mbatch_size") or 0) * int(ta.get("nstep_train") or 1)
        pays = 0 < rows <= 8192 and ta.get("nstep_train", 1) > 1
        mode = args.overlap_acting if overlap is None else overlap
        ta["overlap_acting"] = mode == "on" or (mode == "auto" and pays)
    return config


def build_trainer(config, device, use_graph, data_parallel):
    from rltime_amd.general.loggers import NullLogger
    from rltime_amd.general.type_registry import get_registered_type
    from rltime_amd.train import create_actors
    actors = create_actors(config, device, device_acting=True, use_graph=use_graph)
    cls = get_registered_type("trainers", config["training"]["type"])
    trainer = cls(logger=NullLogger(), actors=actors, model_config=config["model"],
                  policy_args=config.get("policy_args", {}))
    trainer.data_parallel = data_parallel
    trainer.setup(**config["training"]["args"])
    return trainer


class SyntheticFeeder:
    """Stands in for Actor.get_samples during the replay pre-fill (and for the
    --no-acting variant): the same DeviceSamples a device-resident actor emits,
    pre-generated in HBM with the layout of one real acting step, without the
    policy forward."""

    def __init__(self, probe, device, seed, stacked_env=None):
        from rltime_amd.acting.acting_interface import DeviceSamples
        self.cls = DeviceSamples
        self.stacked_env = stacked_env        # frame de-dup: frames must follow the stack-shift contract
        self.envs, self.env_base, self.example = probe.num_envs, probe.env_base, probe.example_state
        step = probe.vector_steps[0]
        g = torch.Generator(device=device).manual_seed(seed)
        E = self.envs
        self.pool = []
        for _ in range(8):
            u = torch.rand(2, E, device=device, generator=g)
            fields = dict(
                frames=torch.randint(0, 256, tuple(step["frames"].shape), dtype=torch.uint8, device=device, generator=g),
                actions=torch.randint(0, 6, (E,), dtype=torch.int32, device=device, generator=g),
                rewards=torch.bucketize(u[0], torch.tensor([0.1, 0.9, 1.0], device=device)).clamp(max=2).float() - 1.0,
                dones=(u[1] < 0.002).to(torch.uint8))
            if step.get("state") is not None:
                fields["state"] = torch.randn(tuple(step["state"].shape), device=device, generator=g) * 0.3
            if step.get("initials") is not None:
                fields["initials"] = (u[1] < 0.002).float()
            if step.get("extra") is not None:
                fields["extra"] = torch.randn(tuple(step["extra"].shape), device=device, generator=g)
            if step.get("policy") is not None:          # the actor's q-values, stored with every transition (keep_policy_outputs)
                fields["policy"] = torch.randn(tuple(step["policy"].shape), device=device, generator=g)
            self.pool.append(fields)
        self.t = 0

    def get_env_count(self):
        return self.envs

    def update_state(self, progress, policy_state=None):
        pass

    def get_samples(self, min_samples):
        iters = (max(1, min_samples) + self.envs - 1) // self.envs
        out = self.cls(self.example, self.envs, self.env_base)
        for _ in range(iters):
            self.t += 1
            fields = self.pool[self.t % len(self.pool)]
            if self.stacked_env is not None:
                obs, _, dones, _ = self.stacked_env.step_device(None)
                fields = dict(fields, frames=obs, dones=dones.to(torch.uint8))
                if "initials" in fields:
                    fields["initials"] = dones.float()
            out.append(**fields)
        return out


def seed_priorities(hist, device, chunk=1 << 18):
    """after the fill, priorities = |N(0,1)| + 1e-6 written
    through update_losses for every live transition (T = 1: one leaf each)."""
    first, count = hist.env_meta()
    g = torch.Generator(device=device).manual_seed(5)
    for e in range(len(first)):
        offs = torch.arange(int(first[e]), int(count[e]), device=device, dtype=torch.int64)
        for at in range(0, offs.numel(), chunk):
            o = offs[at:at + chunk]
            idx = torch.stack([torch.full_like(o, e + hist._env_base), o], dim=1).contiguous()
            hist.update_losses(idx, torch.randn(o.numel(), device=device, generator=g).abs() + 1e-6)


class CpuPath:
    """The oracle — the reference's algorithm class (per-transition records,
    np.stack batch assembly, torch-CPU fwd/bwd) — set up for the config-D workload
    (T=80, burn-in 40, n=2, nature CNN + LSTM512 + IQN head).  `learner_step(B)` is
    one pass of multi_step_trainer.py:278-353 over a batch of B sequences."""

    T, P, n, E, H, A = 80, 40, 2, 8, 512, 6

    def __init__(self, fill_steps=400, size_per_env=500):
        from oracle import replay as orc
        from rltime_amd.general.config import load_config
        from rltime_amd.policies.iqn import IQNPolicy
        from rltime_amd.spaces import Box, Discrete
        T, P, n, E, H, A = self.T, self.P, self.n, self.E, self.H, self.A
        config = load_config("synthetic_atari_iqn_lstm.json")
        torch.set_num_threads(1)
        mk = lambda: IQNPolicy.create(model_config=config["model"], observation_space=Box(0, 255, (4, 84, 84), np.uint8),  # noqa: E731
                                      action_space=Discrete(A), cuda=False, **config["policy_args"])
        self.policy, self.target = mk(), mk()
        self.opt = torch.optim.Adam(self.policy.parameters(), eps=1e-5)
        self.buf = orc.OraclePrioritizedReplay(
            size=E * size_per_env, train_frequency=4, nstep_target=n, nstep_train=T, prefix_steps=P,
            alpha=0.9, beta=0.6, max_weight_factor=0.9, discount_function=orc.make_discount(0.99))
        self.rng = rng = np.random.RandomState(0)
        frame_pool = [rng.randint(0, 256, (4, 84, 84)).astype(np.uint8) for _ in range(64)]
        t0 = time.time()
        fed = 0
        for s in range(fill_steps):
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
            self.buf.update(samples)
            fed += E
        self.ingest_rate = fed / (time.time() - t0)

    def learner_step(self, Bs):
        from oracle import qmath
        from oracle.replay import tree_map
        from rltime_amd.models.torch.utils import make_tensor
        T, P = self.T, self.P
        buf, policy, target, opt = self.buf, self.policy, self.target, self.opt

        def flat(x):
            return x.reshape((x.shape[0] * x.shape[1],) + x.shape[2:])

        def tt(tree):
            return make_tensor(tree, "cpu")

        f32 = lambda a: torch.from_numpy(np.asarray(a).astype(np.float32))  # noqa: E731
        buf.train_quota = 0
        batch = buf.get_train_data(Bs, 0.5)
        for pol, key in ((policy, "states"), (target, "target_states")):      # multi_step_trainer.py:90-131
            st = tt(tree_map(batch[key], lambda x: flat(x[:P])))
            with torch.no_grad():
                pol.predict(st, P)
            hx, cx = pol.model.layers[1].last_state
            keep = 1 - torch.from_numpy(batch[key]["layer1_state"]["initials"][P]).unsqueeze(-1)
            batch[key]["layer1_state"]["hx"][P] = (hx * keep).numpy()
            batch[key]["layer1_state"]["cx"][P] = (cx * keep).numpy()
        data = tree_map(batch, lambda x: flat(x[P:]))
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

    def acting_rate(self, EA=32, reps=5):
        """actor.py:108-147: policy forward on a 32-env vector step, 1 thread."""
        H = self.H
        act_state = {"x": self.rng.randint(0, 256, (EA, 4, 84, 84)).astype(np.uint8), "layer0_state": {},
                     "layer1_state": {"hx": np.zeros((EA, H), np.float32), "cx": np.zeros((EA, H), np.float32),
                                      "initials": np.zeros(EA, np.float32)}, "layer2_state": {}}
        self.policy.actor_predict(act_state, 1)
        t2 = time.time()
        for _ in range(reps):
            self.policy.actor_predict(act_state, 1)
        return reps * EA / (time.time() - t2)


class CpuPathT1:
    """The oracle's T = 1 path for BASELINE configs[1] / [2] — per-transition records, uniform (replay_history.py:93-140)
    or stratified sum-tree sampling (prioritized_replay_history.py:281-356), np.stack batch assembly, torch-CPU
    forward / backward of the nature CNN + FC512 (+ dueling IQN head), Adam, update_losses — at the FULL batch size:
    one learner step here is seconds, so nothing is extrapolated."""

    E, A = 8, 6

    def __init__(self, name, fill_steps=600):
        from oracle import replay as orc
        from rltime_amd.general.config import load_config
        from rltime_amd.general.type_registry import get_registered_type
        from rltime_amd.spaces import Box, Discrete
        spec = CONFIGS[name]
        config = load_config(spec["file"])
        ta = config["training"]["args"]
        self.iqn = config["training"]["type"] == "iqn"
        self.B, self.n = ta["mbatch_size"], ta.get("nstep_target") or 1
        self.double_q, self.gamma = bool(ta.get("double_q")), ta["gamma"]
        self.per = ta["history_mode"]["type"] == "prioritized_replay"
        torch.set_num_threads(1)
        pol_cls = get_registered_type("trainers", config["training"]["type"]).create_policy
        mk = lambda: pol_cls(model_config=config["model"], observation_space=Box(0, 255, (4, 84, 84), np.uint8),  # noqa: E731
                             action_space=Discrete(self.A), cuda=False, **config.get("policy_args", {}))
        self.policy, self.target = mk(), mk()
        self.opt = torch.optim.Adam(self.policy.parameters(), eps=ta.get("adam_epsilon", 1e-8))
        self.clip = ta.get("clip_grad")
        hist = dict(size=self.E * fill_steps, train_frequency=8, nstep_target=self.n, nstep_train=1, prefix_steps=0,
                    discount_function=orc.make_discount(self.gamma))
        if self.per:
            hargs = ta["history_mode"]["args"]
            self.buf = orc.OraclePrioritizedReplay(alpha=hargs["alpha"], beta=hargs["beta"], beta_anneal=hargs.get("beta_anneal", False), **hist)
        else:
            self.buf = orc.OracleReplay(**hist)
        rng = self.rng = np.random.RandomState(0)
        frame_pool = [rng.randint(0, 256, (4, 84, 84)).astype(np.uint8) for _ in range(64)]
        t0 = time.time()
        for s in range(fill_steps):
            self.buf.update([{
                "policy_output": {"actions": int(rng.randint(self.A))},
                "next_state": {"x": frame_pool[(s * self.E + e) % 64].copy(), "layer0_state": {}, "layer1_state": {}},
                "reward": float(rng.choice([-1.0, 0.0, 1.0], p=[.1, .8, .1])), "done": bool(rng.rand() < 0.002),
                "info": {}, "env_id": e} for e in range(self.E)])
        self.ingest_rate = self.E * fill_steps / (time.time() - t0)

    def learner_step(self):
        """-> seconds spent in (get_train_data, targets + forward / backward + Adam, update_losses)."""
        from oracle import qmath
        from oracle.replay import tree_map
        from rltime_amd.models.torch.utils import make_tensor
        buf, policy, target = self.buf, self.policy, self.target
        f32 = lambda a: torch.from_numpy(np.asarray(a).astype(np.float32))  # noqa: E731
        flat = lambda x: x.reshape((x.shape[0] * x.shape[1],) + x.shape[2:])  # noqa: E731
        buf.train_quota = 0
        t0 = time.time()
        batch = buf.get_train_data(self.B, 0.5)
        data = tree_map(batch, flat)
        t1 = time.time()
        tt = lambda tree: make_tensor(tree, "cpu")  # noqa: E731
        with torch.no_grad():
            if self.iqn:
                z_t = target.predict(tt(data["target_states"]), 1)[0]
                z_s = (policy if self.double_q else target).predict(tt(data["target_states"]), 1)[0]
                boot = qmath.iqn_bootstrap(z_t, z_s)
            else:
                q_t = target.predict(tt(data["target_states"]), 1)
                boot = qmath.dqn_bootstrap(q_t, policy.predict(tt(data["target_states"]), 1) if self.double_q else q_t)
            y = qmath.nstep_target(boot, f32(data["returns"]), f32(data["target_masks"]), f32(data["nsteps"]), self.gamma, None)
        self.opt.zero_grad()
        actions = torch.from_numpy(data["policy_outputs"]["actions"])
        w = f32(data["extra_data"]["importance_weights"]) if "importance_weights" in data.get("extra_data", {}) else None
        if self.iqn:
            z, taus = policy.predict(tt(data["states"]), 1)
            loss, rep = qmath.iqn_loss(z, taus, actions, y, w, 1.0, 1, "mean", None)
        else:
            loss, rep = qmath.dqn_loss(policy.predict(tt(data["states"]), 1), actions, y, w, 1.0, "huber", 1, "mean", None)
        loss.backward()
        if self.clip:
            torch.nn.utils.clip_grad_norm_(policy.parameters(), self.clip)
        self.opt.step()
        t2 = time.time()
        if self.per:
            buf.update_losses(data["extra_data"]["loss_indices"], rep.detach().numpy())
        return t1 - t0, t2 - t1, time.time() - t2

    def acting_rate(self, EA=32, reps=5):
        act_state = {"x": self.rng.randint(0, 256, (EA, 4, 84, 84)).astype(np.uint8), "layer0_state": {}, "layer1_state": {}}
        self.policy.actor_predict(act_state, 1)
        t0 = time.time()
        for _ in range(reps):
            self.policy.actor_predict(act_state, 1)
        return reps * EA / (time.time() - t0)


def cpu_baseline_t1(name, seconds=12.0):
    """CpuPathT1 at the config's full batch, 1 torch thread (what the reference runs with): median of up to 5 steps."""
    cpu = CpuPathT1(name)
    cpu.learner_step()                                    # untimed warm-up
    runs, spent = [], 0.0
    while len(runs) < 5 and (spent < seconds or len(runs) < 2):
        parts = cpu.learner_step()
        runs.append(parts)
        spent += sum(parts)
    med = [float(np.median([r[i] for r in runs])) for i in range(3)]
    step_s = sum(med)
    act_rate = cpu.acting_rate()
    acted = cpu.B / 8.0                                   # train_frequency = 8 trained samples per acted one
    extra = acted / cpu.ingest_rate + acted / act_rate
    return {"value": cpu.B / (step_s + extra), "unit": "transitions/s", "cores": 1, "kind": "port",
            "learner_steps_per_sec": 1.0 / (step_s + extra), "host_nproc": os.cpu_count() or 1,
            "ms_get_train_data": med[0] * 1e3, "ms_targets_forward_backward_adam": med[1] * 1e3, "ms_update_losses": med[2] * 1e3,
            "runs": len(runs),
            "sample": "oracle (reference algorithm restated) at the FULL batch B=%d, T=1, n=%d, %s, torch-CPU fp32 with 1 thread like the "
                      "reference's torch.set_num_threads(1): median of %d learner steps (%.1f s of CPU work); + acting %.0f and ingest "
                      "%.0f transitions/s for the step's %d acted transitions.  Sanity anchor (BASELINE.md section 2, the unmodified "
                      "reference in the survey container): get_train_data 13.6 ms uniform B=256 / 44.2 ms + 7.0 ms update_losses PER B=512"
                      % (cpu.B, cpu.n, "sum-tree PER" if cpu.per else "uniform replay", len(runs), spent, act_rate, cpu.ingest_rate, acted)}


def cpu_baseline(args, seconds):
    """CpuPath on bounded samples, scaled linearly in B to B=512.  Headline leg: 1
    torch thread, what the reference runs with (models/torch/torch_model.py:25 calls
    torch.set_num_threads(1)); second leg: more host cores."""
    nproc = os.cpu_count() or 1
    B_full, T = 512, CpuPath.T
    # B = 64 needs 64 active sequences in the oracle's buffer: 8 envs x 750 steps at gap 40 behind a 122-step window
    cpu = CpuPath(fill_steps=750, size_per_env=800) if getattr(args, "cpu_linearity_check", False) else CpuPath()
    learner_step, ingest_rate = cpu.learner_step, cpu.ingest_rate

    def leg(threads, plan, budget):
        """plan: [(B, runs)] -> seconds per B=512 learner step from the largest B that ran
        (the MEDIAN of its runs), {B: (min, median, runs)}, seconds spent."""
        torch.set_num_threads(threads)
        per_b, spent = {}, 0.0
        for Bs, reps in plan:
            times = []
            for _ in range(reps):
                if spent > budget and times:
                    break
                t1 = time.time()
                learner_step(Bs)
                times.append(time.time() - t1)
                spent += times[-1]
            if times:
                per_b[Bs] = (min(times), float(np.median(times)), len(times))
            if spent > budget:
                break
        Bmax = max(per_b)
        return per_b[Bmax][1] * (B_full / Bmax), per_b, spent

    learner_step(2)                                           # untimed warm-up (allocator, MKL)
    one_s, one_b, one_spent = leg(1, [(4, 3), (16, 3)], seconds * 0.7)
    # "all cores": torch intra-op threads capped at 32 — the LSTMCell time loop is a chain of
    # small GEMMs; with one thread per logical core of a 256-core host the same step measured
    # 33x SLOWER than 1 thread (profiles/README.md, round 2).  B=4 first; B=32 only if the
    # threads actually help, so a pathological setting cannot eat minutes of the budget.
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = nproc
    many = max(2, min(usable, 32))
    all_s, all_b, all_spent = leg(many, [(4, 3)], seconds * 0.1)
    if all_b[4][1] < one_b[4][1]:
        s2, b2, sp2 = leg(many, [(32, 3)], seconds * 0.4)
        all_s, all_spent = s2, all_spent + sp2
        all_b.update(b2)
    torch.set_num_threads(1)
    linearity = None
    if getattr(args, "cpu_linearity_check", False):
        t1 = time.time()
        learner_step(64)
        t64 = time.time() - t1
        linearity = {"B": 64, "seconds": t64, "B16_median_x4": one_b[16][1] * 4 if 16 in one_b else None,
                     "ratio_to_linear_extrapolation": (t64 / (one_b[16][1] * 4)) if 16 in one_b else None}
    act_rate = cpu.acting_rate()
    acted_per_step = B_full * T / 4                        # train_frequency=4
    extra = acted_per_step / ingest_rate + acted_per_step / act_rate
    P, n = CpuPath.P, CpuPath.n
    fmt = lambda d: ", ".join("B=%d: median %.2f / min %.2f s/step (%d run%s)" % (b, md, mn, k, "" if k == 1 else "s")  # noqa: E731
                              for b, (mn, md, k) in sorted(d.items()))
    return {
        "value": B_full * T / (one_s + extra), "unit": "transitions/s", "cores": 1, "kind": "port",
        "learner_steps_per_sec": 1.0 / (one_s + extra),
        "host_nproc": nproc,
        "runs": {str(b): {"min_s": mn, "median_s": md, "runs": k} for b, (mn, md, k) in sorted(one_b.items())},
        "linearity_check": linearity,
        "all_cores": {"value": B_full * T / (all_s + extra), "unit": "transitions/s", "cores": many, "usable_cores": usable,
                      "learner_steps_per_sec": 1.0 / (all_s + extra),
                      "sample": "same oracle path with torch.set_num_threads(%d): %s, scaled x%d to B=512; acting/ingest shares as in the 1-thread leg"
                                % (many, fmt(all_b), B_full // max(all_b))},
        "sample": "oracle (reference algorithm restated, config D: T=80, burn-in 40, n=2, torch-CPU fp32, 1 thread like the "
                  "reference's torch.set_num_threads(1)): %s; the largest B's MEDIAN scaled linearly x%d to B=512 (%.1f s of CPU work); "
                  "+ acting %.0f and ingest %.0f transitions/s for the step's %d acted transitions; host has %d logical cores; "
                  "oracle / reference time ratio at identical inputs: BASELINE.md (tools/ref_vs_oracle_cpu.py)"
                  % (fmt(one_b), B_full // max(one_b), one_spent + all_spent, act_rate, ingest_rate, acted_per_step, nproc)}


def launch_argv(argv, n, port):
    """The command the driver would use for N ranks on one node (one process per
    GPU over RCCL), built around THIS script and its own arguments."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def self_launch(args, argv):
    """`python bench.py --gpus N` with no torchrun environment: start the N ranks
    ourselves.  Rank 0's JSON line passes through on stdout."""
    import subprocess
    cmd = launch_argv([a for a in argv if a != "--dry-launch"], args.gpus, args.master_port or free_port())
    if args.dry_launch:
        print(json.dumps({"launch": cmd}))
        return 0
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on these hosts (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "4")
    # stdout carries exactly ONE line: rank 0's JSON record.  Everything else the ranks or the launcher print there
    # (RCCL / gloo banners, warnings) goes to stderr so that a parser of the last stdout line cannot trip over it.
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, bufsize=1)
    record = None
    for line in proc.stdout:
        text = line.strip()
        is_record = False
        if text.startswith("{") and text.endswith("}"):
            try:
                is_record = "metric" in json.loads(text)
            except ValueError:
                is_record = False
        if is_record:
            record = text
        else:
            sys.stderr.write(line)
    rc = proc.wait()
    if record is not None:
        print(record, flush=True)
    return rc


def copy_peak(device, stream_ptr_fn):
    """Measured HBM copy rate of this box (read + write bytes of one mirl_copy_bytes_ex pass over the
    frame gather's byte count; the better of the cached grid-stride and the non-temporal variant),
    next to the spec peak."""
    import ctypes as C
    from rltime_amd._lib import lib, check
    n = 122 * 512 * 28224
    a = torch.empty(n, dtype=torch.uint8, device=device).random_(0, 256)
    b = torch.empty_like(a)
    best = None
    names = {0: "cached, grid-stride", 1: "non-temporal, one 4-vector chunk per lane", 4: "4 vectors per lane, cached loads + non-temporal stores",
             6: "2 vectors per lane, non-temporal loads + stores"}
    for nt in (0, 1, 4, 6):
        f = lambda: check(lib.mirl_copy_bytes_ex(C.c_void_p(b.data_ptr()), C.c_void_p(a.data_ptr()), n, nt, stream_ptr_fn()))  # noqa: E731
        for _ in range(3):
            f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            f()
        e1.record()
        torch.cuda.synchronize()
        rate = 2.0 * n / (e0.elapsed_time(e1) / 10) / 1e6
        if best is None or rate > best[0]:
            best = (rate, names[nt])
    del a, b
    return best


# Which pipe a kernel's matrix work runs on (its `flop` is the f32-product work 2 M N K the call site states):
#   bf16x6  six exact-split bf16 MFMAs per f32 product block (csrc/gemm3.hip, conv3.hip)  -> 2.5 PF / 6
#   bf16x3  uint8 pixel x three-way split weight (csrc/conv_in.hip forward)                 -> 2.5 PF / 3
#   f32     v_mfma_f32_16x16x4_f32 (input layer's weight gradient, layer 2's data gradient, LSTM sweeps)
PIPE_OF = {"k_gemm3_nt": "bf16x6", "k_gemm3_nt_mid": "bf16x6", "k_gemm3_nt_head": "bf16x6", "k_gemm3_ps": "bf16x6", "k_gemm3_nn": "bf16x6", "k_gemm3_tn": "bf16x6", "k_gemm3_nt_mul": "bf16x6", "k_conv3_fwd": "bf16x6",
           "k_conv1_u8_fwd": "bf16x3", "k_conv1_u8_wrw": "f32", "k_conv1_u8_wrw_b3": "bf16x3", "k_conv2_bwd_data": "f32", "k_conv2_bwd_data_b3": "bf16x6", "k_conv3_bwd_data_b3": "bf16x6", "k_conv_wrw_b3": "bf16x6", "k_gemm3_nn_qp": "bf16x6",
           "k_lstm_seq_fwd": "f32", "k_lstm_seq_bwd": "f32", "k_lstm_step_fwd": "f32"}
# launch / dependency-latency bound by construction (one workgroup of bookkeeping, one tree level per barrier, 256-row
# acting batches, one LSTM step per launch): microseconds per call is the figure, an HBM fraction would be noise
LATENCY_KERNELS = {"k_ingest_fused", "k_ingest_scalars", "k_plan_apply", "k_actor_pre", "k_actor_head", "k_lstm_cell_fwd", "k_lstm_cell_bwd",
                   "k_tree_fix", "k_tree_fix(ingest)", "k_per_sample", "k_per_sample_global", "k_uniform_sample", "k_loss_stamp",
                   "k_loss_write", "k_recalc_flagged", "k_recalc_flagged_wave", "k_gather_scalars", "k_conv1_pack_w", "k_conv2_pack_w", "k_conv2_pack_w3b", "k_conv3_pack_w3b",
                   "k_colsum_partials", "k_episode_track", "k_acting_td", "k_dedup_depth", "k_gemm3_reduce", "k_conv1_wrw_reduce", "k_conv_wrw_reduce"}


def pipe_peak_tflops(pipe):
    if pipe == "bf16x6":
        return BF16_MFMA_PEAK_TFLOPS / 6
    if pipe == "bf16x3":
        return BF16_MFMA_PEAK_TFLOPS / 3 if os.environ.get("MIRL_CONV1_BF16", "1") != "0" else F32_MFMA_PEAK_TFLOPS
    return F32_MFMA_PEAK_TFLOPS


def kernel_table(table, profile_steps):
    """mirl_profile_* rows -> (roofline_all kernel entries, sum of the kernels' roofline times in ms per step).
    Every kernel is priced by ITS bound: max(algorithmic HBM bytes / 8 TB/s, f32-product flop / its pipe's dense peak);
    launch- / latency-bound kernels report microseconds per call only and count with a roofline time of zero."""
    kernels, ideal_ms = [], 0.0
    for row in sorted(table, key=lambda r: -r["total_ms"]):
        if not row["calls"]:
            continue
        name = row["name"]
        avg_us = row["total_ms"] / row["calls"] * 1e3
        by, fl = row["algorithmic_bytes"] / row["calls"], row.get("flop", 0.0) / row["calls"]
        entry = {"kernel": name, "launches_per_step": round(row["calls"] / profile_steps, 2),
                 "avg_us": round(avg_us, 2), "ms_per_step": round(row["total_ms"] / profile_steps, 4)}
        if name in LATENCY_KERNELS or (by < (1 << 20) and fl <= 0):
            entry["bound"] = "latency"
            kernels.append(entry)
            continue
        hbm_us = by / (HBM_PEAK_GBPS * 1e9) * 1e6
        pipe = PIPE_OF.get(name)
        mfma_us = fl / (pipe_peak_tflops(pipe) * 1e12) * 1e6 if (pipe and fl > 0) else 0.0
        floor_us = max(hbm_us, mfma_us)
        entry.update({"bound": "mfma" if mfma_us > hbm_us else "hbm", "roofline_us": round(floor_us, 2),
                      "frac_of_roofline": round(floor_us / avg_us, 4) if avg_us > 0 else None,
                      "algorithmic_bytes_per_launch": by if by > 0 else None,
                      "achieved_GBps": round(by / (avg_us * 1e-6) / 1e9, 1) if by > 0 else None,
                      "frac_of_hbm_peak": round(by / (avg_us * 1e-6) / 1e9 / HBM_PEAK_GBPS, 4) if by > 0 else None})
        if pipe and fl > 0:
            tf = fl / (avg_us * 1e-6) / 1e12
            entry.update({"flop_per_launch": fl, "achieved_TFLOPs": round(tf, 1), "pipe": pipe,
                          "peak_TFLOPs": round(pipe_peak_tflops(pipe), 1), "frac_of_pipe_peak": round(tf / pipe_peak_tflops(pipe), 4),
                          "x_f32_mfma_peak": round(tf / F32_MFMA_PEAK_TFLOPS, 3)})
            if name in ("k_lstm_seq_fwd", "k_lstm_seq_bwd"):
                entry["note"] = "MFMA + per-step exchange latency: roofline_us is the f32-MFMA time of the recurrent products alone"
        ideal_ms += floor_us * row["calls"] / profile_steps * 1e-3
        kernels.append(entry)
    return kernels, ideal_ms


def kernel_family(name):
    """Device kernels that are not this library's, by what they are (the named list of the step's unpriced time)."""
    n = name
    if n.startswith("mirl::") or "mirl::" in n.split("(")[0]:
        return None
    if "copyBuffer" in n or "Memcpy" in n or "memcpy" in n.lower():
        return "runtime copies (hipMemcpyAsync D2D / H2D: plan uploads, staging)"
    if "fillBuffer" in n or "Memset" in n or "memset" in n.lower():
        return "runtime fills (hipMemsetAsync)"
    if n.startswith("Cijk_") or "gemm" in n.lower() and "igemm" not in n.lower():
        return "hipBLASLt / rocBLAS GEMMs (acting batch, per-step LSTM backward products, small layers)"
    if "igemm" in n or "SubTensorOp" in n or "miopen" in n.lower() or "naive_conv" in n:
        return "MIOpen convolutions (+ their fills)"
    if "multi_tensor" in n or "FusedAdam" in n or "adam" in n.lower():
        return "optimizer (multi-tensor Adam, foreach scale / norm)"
    if "reduce_kernel" in n or "Norm" in n:
        return "PyTorch reductions (grad-norm, sums)"
    if "elementwise" in n or "vectorized" in n or "CatArray" in n or "distribution" in n or "index" in n.lower():
        return "PyTorch elementwise / copy / cat / rand kernels"
    return "other"


def library_roofline(one_step):
    """One extra step under the torch profiler: the LIBRARY contractions of the step — hipBLASLt / rocBLAS GEMMs and MIOpen
    convolutions called through aten — with their flop (aten's own count for mm / addmm / conv2d; 2 x MACs per requested
    gradient for aten::convolution_backward from its operand shapes and output mask), priced at the dense f32 MFMA peak
    (they compute f32 products on the f32 pipe), and the device time of every kernel that is not librltime_hip.
    -> dict or None when the profiler is unavailable."""
    try:
        from torch.profiler import profile, ProfilerActivity
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_flops=True, record_shapes=True) as prof:
            one_step()
            torch.cuda.synchronize()
        flop = 0.0
        for ev in prof.events():
            if ev.flops:
                flop += float(ev.flops)
            elif ev.name == "aten::convolution_backward" and len(ev.input_shapes) >= 3:
                go, _, w = ev.input_shapes[:3]
                mask = (getattr(ev, "concrete_inputs", None) or [None])[-1]
                wanted = sum(1 for m in mask[:2] if m) if isinstance(mask, (list, tuple)) and len(mask) >= 2 else 2
                if len(go) == 4 and len(w) == 4:
                    flop += 2.0 * go[0] * go[1] * go[2] * go[3] * w[1] * w[2] * w[3] * wanted
        dev_ms, lib_ms = 0.0, 0.0
        for ev in prof.key_averages():
            if "CPU" not in str(getattr(ev, "device_type", "CPU")):
                continue                  # device-side rows repeat the time their launching op already carries
            t = getattr(ev, "self_device_time_total", getattr(ev, "self_cuda_time_total", 0)) / 1e3
            dev_ms += t
            if "mm" in ev.key or "convolution" in ev.key or "conv2d" in ev.key:
                lib_ms += t
        # every DEVICE kernel of the step that is not librltime_hip's, by family: the part of the step no roofline prices
        families = {}
        for ev in prof.key_averages():
            if "CPU" in str(getattr(ev, "device_type", "CPU")):
                continue
            fam = kernel_family(ev.key)
            if fam is None:
                continue
            t = getattr(ev, "device_time_total", getattr(ev, "cuda_time_total", 0)) / 1e3
            row = families.setdefault(fam, {"ms": 0.0, "launches": 0})
            row["ms"] += t
            row["launches"] += int(ev.count)
        return {"library_flop_per_step": flop, "library_contraction_ms": lib_ms, "aten_device_ms": dev_ms,
                "library_roofline_ms": flop / (F32_MFMA_PEAK_TFLOPS * 1e12) * 1e3,
                "other_kernels_by_family": {k: {"ms": round(v["ms"], 3), "launches": v["launches"]} for k, v in sorted(families.items(), key=lambda kv: -kv[1]["ms"])}}
    except Exception as e:            # evidence, not the measurement: never sink the line
        print("library roofline pass failed: %r" % (e,), file=sys.stderr)
        return None


def run_mode(args, scaling, rank, world, device, dp, want_tables, overlap=None):
    """Build the trainer for one scaling mode, pre-fill its replay shard, run W
    warm-up + K timed steps between barriers.  Returns the raw measurements."""
    import gc
    import torch.distributed as dist
    torch.manual_seed(1234 + rank)
    np.random.seed(1234 + rank)
    config = build_config(args, rank, world, scaling, overlap)
    targs = config["training"]["args"]
    trainer = build_trainer(config, device, use_graph=not args.no_acting_graph, data_parallel=dp)
    hist = trainer.history_buffer
    real_actors = trainer.actors
    envs = real_actors.get_env_count()
    size = targs["history_mode"]["args"]["size"]

    # ---- pre-fill the replay shard (untimed) ---------------------------------
    t0 = time.time()
    sink, real_actors._sink = getattr(real_actors, "_sink", None), None       # the probe step comes back as tensors
    probe = real_actors.get_samples(envs)                    # one real acting step: learns the transition layout
    hist.update(probe)
    real_actors._sink = sink
    feeder = SyntheticFeeder(probe, device, seed=99 + rank,
                             stacked_env=real_actors._vec_env if args.frame_dedup else None)
    trainer.actors = feeder            # pre-generated actor output (no policy forward) for the fill
    per_call = 64 * envs
    fed = envs
    while fed < size + envs:
        hist.update(feeder.get_samples(per_call))
        fed += per_call
    per = targs["history_mode"]["type"] == "prioritized_replay"
    if per and targs["nstep_train"] == 1:
        seed_priorities(hist, device)
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
    if dp is not None:
        dp.timing = []
        dp.buckets_overlapped = 0
    if world > 1:
        dist.barrier()
    hist.profile(True)
    steps_before = trainer.steps
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        one_step()
        marks[i + 1].record()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    launches, gather_ms = hist.profile(False)
    acted = trainer.steps - steps_before
    step_ms = np.array([marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps)])
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    rccl = None
    if dp is not None:
        pairs, dp.timing = dp.timing, None
        ar_ms = [a.elapsed_time(b) for a, b in pairs]
        rccl = {"backend": dist.get_backend(), "ranks_seen": dp.ranks_seen(device), "world_size": world,
                "allreduce_calls_per_step": len(ar_ms) / max(args.steps, 1),
                "allreduce_ms_per_step": float(np.sum(ar_ms)) / max(args.steps, 1) if ar_ms else None,
                "allreduce_ms_median": float(np.median(ar_ms)) if ar_ms else None,
                "bucket_bytes": dp.bucket_bytes,
                "buckets": [b["hi"] - b["lo"] for b in dp._buckets],
                "buckets_issued_from_backward_hooks_per_step": dp.buckets_overlapped / max(args.steps, 1),
                "source": "HIP events on rank 0's stream around the part of the gradient all-reduce that is still EXPOSED after "
                          "the backward pass (buckets head -> LSTM -> conv go out asynchronously from autograd's post-accumulate "
                          "hooks, rltime_amd/parallel.py); ranks_seen = all-reduced sum of one int per rank over the same group"}

    # ---- per-kernel pass (untimed): HIP events around every librltime_hip launch
    table, prof_step_ms = [], None
    gstep = getattr(trainer, "_gstep", None)
    graph_step = bool(gstep is not None and gstep.get("graph") is not None)
    if graph_step and want_tables and args.profile_steps > 0:
        # a replayed graph carries no per-launch events: the per-kernel passes issue the SAME step launch by launch
        trainer._graph_capture, trainer._gstep = False, None
    if want_tables and args.profile_steps > 0:
        from rltime_amd import _lib
        _lib.check(_lib.lib.mirl_profile_reset())
        _lib.check(_lib.lib.mirl_profile_set(2))
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.profile_steps):
            one_step()
        torch.cuda.synchronize()
        prof_step_ms = (time.perf_counter() - t1) / args.profile_steps * 1e3
        _lib.check(_lib.lib.mirl_profile_set(0))
        table = _lib.profile_table()

    lib_roof = library_roofline(one_step) if (want_tables and args.profile_steps > 0) else None

    if want_tables and os.environ.get("BENCH_GEMM_SHAPES"):
        # one extra step under the torch profiler: every library GEMM call with its operand shapes
        from torch.profiler import profile, ProfilerActivity
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
            one_step()
            torch.cuda.synchronize()
        rows = []
        for ev in prof.key_averages(group_by_input_shape=True):
            if ev.key in ("aten::mm", "aten::addmm", "aten::_addmm_activation", "aten::bmm", "aten::addmm_"):
                rows.append({"op": ev.key, "shapes": str(ev.input_shapes), "calls": ev.count,
                             "device_ms": round(getattr(ev, "device_time_total", getattr(ev, "cuda_time_total", 0)) / 1e3, 3)})
        rows.sort(key=lambda r: -r["device_ms"])
        with open(os.environ["BENCH_GEMM_SHAPES"], "w") as f:
            for r in rows:
                f.write(json.dumps(r) + "\n")

    T, P, B = targs["nstep_train"], targs.get("burn_in_timesteps", 0), targs["mbatch_size"]
    n = targs.get("nstep_target") or targs["nstep_train"]
    res = dict(scaling=scaling, dt=dt, step_ms=step_ms, launches=launches, gather_ms=gather_ms, acted=acted,
               graph_step=graph_step,
               table=table, prof_step_ms=prof_step_ms, lib_roof=lib_roof, T=T, P=P, n=n, B=B, rows=hist._rows, envs=envs, per=per,
               hist_stats=hist_stats, fill_s=fill_s, rccl=rccl, overlap=bool(targs.get("overlap_acting")) and not args.no_acting)
    trainer.actors = real_actors
    hist.close()
    if hasattr(real_actors, "_graphed"):
        real_actors._graphed = None
    del trainer, hist, real_actors, feeder, probe
    gc.collect()
    torch.cuda.empty_cache()
    return res


def summary(res, world, steps):
    """Throughput figures of one run_mode result."""
    B, T, dt = res["B"], res["T"], res["dt"]
    sm = res["step_ms"]
    return {"value": world * B * T * steps / dt, "unit": "transitions/s", "learner_steps_per_sec": steps / dt,
            "ms_per_step": dt / steps * 1e3,
            "step_ms": {"median": float(np.median(sm)), "p10": float(np.percentile(sm, 10)),
                        "p90": float(np.percentile(sm, 90)), "source": "HIP events at step boundaries, rank 0"},
            "mbatch_per_gpu": B, "global_mbatch": B * world, "envs_per_gpu": res["envs"],
            "replay_transitions_per_gpu": res["hist_stats"]["total_items"],
            "acted_transitions_per_step_per_gpu": res["acted"] / steps}


def gather_roofline(res, frame_dedup=False):
    """The frame gather of one run_mode result: algorithmic bytes per launch (every gathered row read once and
    written once = 2 x state rows x B x F) over the mean launch duration (HIP events on the launch stream, timed region)."""
    F = 4 * 84 * 84
    rows, B = res["rows"], res["B"]
    algo = 2.0 * rows * B * F
    if frame_dedup:            # every stack written once, every distinct plane of a window read once
        algo = rows * B * F + B * (rows + 3) * (F / 4.0)
    launches, ms = res["launches"], res["gather_ms"]
    avg_ms = ms / max(launches, 1)
    achieved = algo / (avg_ms * 1e-3) / 1e9 if launches else None
    return algo, avg_ms, achieved


def other_config_record(args, name, rank, world, device):
    """BASELINE configs[1] / [2] as a sub-record of the default line: the same run_mode protocol (1M-transition replay
    pre-filled, warm-up, K timed steps between synchronisations), its gather roofline and its CPU baseline."""
    import argparse as ap
    sub = ap.Namespace(**vars(args))
    sub.config, sub.steps, sub.warmup = name, args.other_steps, 20
    sub.mbatch = sub.nstep_train = sub.burn_in = sub.nstep_target = sub.envs = None
    sub.train_arg, sub.frame_dedup, sub.replay_size, sub.no_acting = [], False, 1000000, False
    res = run_mode(sub, "strong", rank, world, device, None, want_tables=False)
    head = summary(res, world, sub.steps)
    algo, avg_ms, achieved = gather_roofline(res)
    rec = {"metric": CONFIGS[name]["metric"], "workload": CONFIGS[name]["workload"],
           "value": head["value"], "unit": "transitions/s", "learner_steps_per_sec": head["learner_steps_per_sec"],
           "ms_per_step": head["ms_per_step"], "step_ms": head["step_ms"], "steps": sub.steps, "warmup": sub.warmup,
           "mbatch": res["B"], "nstep_train": res["T"], "nstep_target": res["n"], "envs": res["envs"],
           "replay_transitions": res["hist_stats"]["total_items"], "tree_capacity": res["hist_stats"]["tree_capacity"] if res["per"] else None,
           "acted_transitions_per_step": res["acted"] / sub.steps, "learner_step_hip_graph": res["graph_step"],
           "roofline": {"kernel": "k_gather_rows (frames)", "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                        "frac": (achieved / HBM_PEAK_GBPS) if achieved else None, "algorithmic_bytes_per_launch": algo,
                        "avg_launch_ms": avg_ms, "launches": res["launches"], "traffic": None}}
    if not args.no_cpu_baseline:
        try:
            rec["cpu_baseline"] = cpu_baseline_t1(name)
            rec["speedup_vs_cpu_baseline"] = rec["value"] / rec["cpu_baseline"]["value"]
        except Exception as e:
            rec["cpu_baseline"] = {"error": repr(e)}
    return rec


def file_date(path):
    """Collection date of an evidence file: its own "collected" field, else its modification time."""
    try:
        d = json.load(open(path))
        if isinstance(d, dict) and d.get("collected"):
            return d["collected"]
    except Exception:
        pass
    try:
        return time.strftime("%Y-%m-%d", time.gmtime(os.path.getmtime(path)))
    except OSError:
        return None


def main():
    args = parse()
    launched = "WORLD_SIZE" in os.environ
    force = bool(os.environ.get("BENCH_FORCE_DIST"))
    if not launched and (args.gpus > 1 or force or args.dry_launch):
        sys.exit(self_launch(args, sys.argv[1:]))
    from rltime_amd import parallel
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks; start it as `python bench.py --gpus N` "
                 "(self-launching) or `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`" % (args.gpus, world))
    if args.share_gpu:
        local = local % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    import torch.distributed as dist
    rank, world, _, dp = parallel.init_from_env(backend="gloo" if args.share_gpu else None, device=device)
    if args.miopen_find:
        torch.backends.cudnn.benchmark = True
    spec = CONFIGS[args.config]

    # N = 1: both modes are the same job.  N > 1: the headline is STRONG scaling (SURVEY 8d
    # config 5: the configured batch / envs / replay are whole-job values split over the ranks,
    # so learner steps/s is comparable across N); the weak run (every rank keeps the full
    # configured batch on its own full-size shard) rides along as the `weak` sub-record.
    modes = [args.scaling] if args.scaling != "both" else (["strong", "weak"] if world > 1 else ["strong"])
    runs = [run_mode(args, m, rank, world, device, dp, want_tables=(i == 0)) for i, m in enumerate(modes)]
    res = runs[0]
    # N > 1: the headline keeps the synchronous actor (same algorithm as N = 1); the overlapped schedule (actor weights
    # one learner step stale) is measured next to it where a rank's batch is small enough for it to pay
    overlapped = None
    if world > 1 and args.overlap_acting == "off" and not args.no_acting and modes[0] == "strong" \
            and 0 < res["B"] * res["T"] <= 8192 and res["T"] > 1:
        overlapped = run_mode(args, "strong", rank, world, device, dp, want_tables=False, overlap="on")
    measured_peak = None
    if rank == 0:
        try:
            import ctypes as C
            measured_peak = copy_peak(device, lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream))
        except Exception as e:
            measured_peak = None
            print("copy peak pass failed: %r" % (e,), file=sys.stderr)

    if rank == 0:
        T, P, n, B = res["T"], res["P"], res["n"], res["B"]
        per, hist_stats, envs = res["per"], res["hist_stats"], res["envs"]
        table, launches, gather_ms = res["table"], res["launches"], res["gather_ms"]
        algo_bytes, avg_ms, achieved = gather_roofline(res, args.frame_dedup)
        traffic, traffic_src = None, None
        if args.config == "iqn_lstm" and world == 1 and os.path.isfile(args.pmc_traffic):
            try:
                traffic = json.load(open(args.pmc_traffic)).get("hbm_bytes_per_launch")
                traffic_src = "profiles/gather_traffic.json: rocprofv3 PMC passes (FETCH_SIZE x2 + WRITE_SIZE, separate runs) over " \
                              "tools/gather_probe.py with the same 1M-transition replay and B/T/P/n, collected by " \
                              "`tools/gpu_round.sh <tag> pmc` on %s — NOT measured in this run" % file_date(args.pmc_traffic)
            except Exception:
                traffic = None
        kernels, ideal_ours_ms = kernel_table(table, args.profile_steps)
        head = summary(res, world, args.steps)
        mode_text = {"strong": "strong scaling: the configured batch (global B=%d), envs and replay size are whole-job values "
                               "split evenly over the ranks" % (B * world),
                     "weak": "weak scaling: every rank keeps the configured batch, envs and replay size"}
        out = {
            "metric": spec["metric"],
            "value": head["value"], "unit": "transitions/s",
            "learner_steps_per_sec": head["learner_steps_per_sec"],
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": head["ms_per_step"], "step_ms": head["step_ms"],
            "higher_is_better": True, "scaling": res["scaling"], "vs_baseline": None,
            "dtype": "f32" if args.amp == "none" else "bf16(network autocast)+f32(hot path)",
            "dtype_note": "f32 operands, f32 accumulation and f32 results everywhere (sum tree: the reference's f32 / f64 scalar kinds; "
                          "n-step returns f64 -> f32).  The WIDE f32 products (head / LSTM-projection GEMMs, conv layers 2-3, the conv "
                          "backward) run on the bf16 matrix pipe as split-bf16: every f32 operand is split exactly into three bf16 "
                          "parts and SIX of the nine part products are accumulated in f32, smallest first; the three dropped ones "
                          "(mid.lo, lo.mid, lo.lo) are each below 2^-23 |a||b|, the size of one f32 rounding — results stay within the "
                          "library f32 kernels' own distance from float64 (tests/test_gemm3_gpu.py); non-finite inputs become NaN "
                          "(inf - inf in the split) instead of propagating as inf.  MIRL_GEMM3=0 MIRL_CONV1_BF16=0 MIRL_CONV3=0 selects "
                          "the f32 MFMA pipe only",
            "data": "synthetic",
            "config": {
                "workload": spec["workload"] + " [%s]" % mode_text[res["scaling"]],
                "mbatch_per_gpu": B, "global_mbatch": B * world, "nstep_train": T, "burn_in": P, "nstep_target": n,
                "frame": "(4,84,84) u8" + (" stack-consistent, stored de-duplicated (one 84x84 plane per transition)" if args.frame_dedup else ""),
                "replay_transitions_per_gpu": hist_stats["total_items"],
                "active_sequences_per_gpu": hist_stats["active_sequences"],
                "tree_capacity": hist_stats["tree_capacity"] if per else None,
                "envs_per_gpu": envs, "acted_transitions_per_step_per_gpu": res["acted"] / args.steps,
                "acting_policy_forward_in_step": not args.no_acting,
                "acting_forward_hip_graph": (not args.no_acting) and (not args.no_acting_graph),
                "acting_overlapped_on_second_stream": res["overlap"],
                "learner_step_hip_graph": res["graph_step"],
                "parallelism": "dp%d (replay sharded by env, grad all-reduce)" % world + (" — ranks SHARE GPUs over gloo (--share-gpu): launcher check, not a scaling number" if args.share_gpu else ""),
                "replay_fill_seconds": round(res["fill_s"], 2),
                # switches that differ from what a reference json config would select on its own
                "keep_policy_outputs": not args.no_policy_outputs,
                "device_rng": "Philox4x32-10 on the device for replay sampling, epsilon-greedy and acting-time quantile fractions "
                              "(the reference draws them with random / np.random / torch.rand on the host: same distributions, other streams; "
                              "sum-tree indices are bit-exact under the reference's RNG in tests/)",
                "selection_advantage_only": "double-Q action selection from the advantage stream alone (arg-max unchanged: V and "
                                            "mean_a A are constant over actions, policies/dqn.py predict_selection)",
                "f32_products_on": ("bf16 matrix pipe for the wide GEMMs, conv layers 2-3 forward and the conv stack's backward (6 exact-split "
                                    "bf16 MFMAs per f32 product block: csrc/gemm3.hip, conv3.hip, conv_mid.hip, conv_wrw.hip) and the input "
                                    "layer's forward and weight gradient (uint8 pixels are exact bf16, the f32 operand split three ways: 3 "
                                    "MFMAs per block, csrc/conv_in.hip); f32 accumulation, results within the library f32 kernels' own "
                                    "distance from float64 (tests/test_gemm3_gpu.py, test_conv*_gpu.py, DESIGN 3.6-3.8); LSTM sweeps and "
                                    "the small / acting-batch products on the f32 pipe (own kernels, hipBLASLt, MIOpen)"
                                    if (os.environ.get("MIRL_GEMM3", "1") != "0" or os.environ.get("MIRL_CONV1_BF16", "1") != "0"
                                        or os.environ.get("MIRL_CONV3", "1") != "0")
                                    else "f32 MFMA pipe only (MIRL_GEMM3=0 MIRL_CONV1_BF16=0 MIRL_CONV3=0)")},
            "roofline": {
                "kernel": "k_gather_rows_dedup (frames)" if args.frame_dedup else "k_gather_rows (frames)", "bound": "hbm",
                "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": (achieved / HBM_PEAK_GBPS) if achieved else None,
                "measured_copy_peak_GBps": measured_peak[0] if measured_peak else None,
                "frac_of_measured_copy_peak": (achieved / measured_peak[0]) if (achieved and measured_peak) else None,
                "measured_copy_peak_how": "read + written bytes of a plain 16 B/lane device copy (mirl_copy_bytes_ex, the best of "
                                          "four variants: %s) over the gather's 3.5 GB on this box, after the timed region; the "
                                          "gather itself is such a copy with indexed rows; `peak` is the HBM3E spec figure"
                                          % (measured_peak[1] if measured_peak else "-"),
                "algorithmic_bytes_per_launch": algo_bytes, "avg_launch_ms": avg_ms, "launches": launches,
                "traffic": traffic, "traffic_source": traffic_src},
            "roofline_all": {
                "how": "%d extra steps after the timed region with a HIP event pair around every librltime_hip launch "
                       "(mirl_profile_*).  Every kernel is priced by its own bound: roofline_us = max(algorithmic HBM bytes / %.0f GB/s, "
                       "f32-product flop / the dense peak of the pipe it runs on) and frac_of_roofline = roofline_us / avg_us; pipes: "
                       "bf16x6 = dense bf16 MFMA peak / 6 exact-split part products (k_gemm3_*, k_conv3_fwd), bf16x3 = / 3 (input "
                       "layer's forward: uint8 pixels are exact bf16), f32 = v_mfma_f32 peak (input layer's weight gradient, layer 2's "
                       "data gradient, the LSTM sweeps' recurrent products).  Launch- / latency-bound kernels (bookkeeping, tree, "
                       "sampling, 256-row acting batches, per-step LSTM cells) report microseconds per call only"
                       % (args.profile_steps, HBM_PEAK_GBPS),
                "ms_per_step_with_events": res["prof_step_ms"], "kernels": kernels} if kernels else None,
        }
        if kernels:
            # how good is the WHOLE step: the sum of every kernel's roofline time (ours from the table above, the library
            # contractions from one torch-profiler step at the f32 MFMA peak) over the measured step time
            lr = res["lib_roof"]
            lib_ms = lr["library_roofline_ms"] if lr else None
            total = ideal_ours_ms + (lib_ms or 0.0)
            out["roofline_step"] = {
                "roofline_ms": round(total, 3), "ms_per_step": head["ms_per_step"], "frac": round(total / head["ms_per_step"], 4),
                "librltime_hip_roofline_ms": round(ideal_ours_ms, 3),
                "librltime_hip_measured_ms": round(sum(k["ms_per_step"] for k in kernels), 3),
                "library_contractions_roofline_ms": round(lib_ms, 3) if lib_ms is not None else None,
                "library_contractions_measured_ms": round(lr["library_contraction_ms"], 3) if lr else None,
                "library_flop_per_step": lr["library_flop_per_step"] if lr else None,
                "all_aten_kernels_measured_ms": round(lr["aten_device_ms"], 3) if lr else None,
                # the step's time that carries no roofline at all, by kernel family (one step under torch.profiler; the acting
                # rollout's kernels run from a captured graph and appear under their own names)
                "unpriced_by_family": lr.get("other_kernels_by_family") if lr else None,
                "how": "sum over the step's kernels of max(algorithmic bytes / 8 TB/s, flop / pipe peak): librltime_hip kernels from "
                       "roofline_all (latency-bound ones count 0), hipBLASLt / MIOpen contractions from one extra step under "
                       "torch.profiler(with_flops) priced at the dense f32 MFMA peak (157.3 TFLOP/s); elementwise / copy kernels of "
                       "PyTorch and the runtime count 0 (pure overhead); divided by the timed ms_per_step"}

        if res["rccl"] is not None:
            out["rccl"] = res["rccl"]
        for other in runs[1:]:
            sub = summary(other, world, args.steps)
            sub["workload"] = mode_text[other["scaling"]]
            if other["rccl"] is not None:
                sub["rccl"] = other["rccl"]
            out[other["scaling"]] = sub
        if overlapped is not None:
            sub = summary(overlapped, world, args.steps)
            sub["workload"] = mode_text["strong"] + "; acting + ingest of iteration k+1 on a second HIP stream against iteration k's " \
                              "training (actor weights one learner step stale, like the reference's async actors) — NOT the headline"
            if overlapped["rccl"] is not None:
                sub["rccl"] = overlapped["rccl"]
            out["overlapped_acting"] = sub
        if res["graph_step"] and isinstance(out.get("roofline_all"), dict):
            out["roofline_all"]["graphed_step"] = "the timed steps replay the learner step from a captured HIP graph (no per-launch events " \
                                                  "inside a replay); the per-kernel figures here come from the same step issued launch by launch"
        if args.config == "iqn_lstm":
            out["config"]["lstm_state"] = "2x512 f32 per transition"
        if world == 1 and not args.no_cpu_baseline and args.config == "iqn_lstm":
            try:
                out["cpu_baseline"] = cpu_baseline(args, args.cpu_seconds)
                out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
            except Exception as e:            # the baseline must never sink the GPU number
                out["cpu_baseline"] = {"error": repr(e)}
        elif world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline_t1(args.config)
                out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
            except Exception as e:
                out["cpu_baseline"] = {"error": repr(e)}
        plain = all(v is None for v in (args.mbatch, args.nstep_train, args.burn_in, args.nstep_target, args.envs)) \
            and not (args.train_arg or args.frame_dedup or args.no_acting or args.amp != "none" or args.replay_size != 1000000)
        if world == 1 and dp is None and args.config == "iqn_lstm" and plain and not args.no_other_configs:
            # BASELINE configs[1] and [2] ride along in the driver's line (same protocol, own roofline and CPU baseline)
            out["other_configs"] = {}
            for name in ("dqn_uniform", "rainbow_iqn"):
                try:
                    out["other_configs"][name] = other_config_record(args, name, rank, world, device)
                except Exception as e:        # never sink the headline
                    out["other_configs"][name] = {"error": repr(e)}
        out["evidence_dates"] = {
            "profiles/gather_traffic.json": file_date(args.pmc_traffic),
            "gemm3 PMC (profiles/r06_gemm3_pmc_round5_kernels_nt_head_tn.json)": file_date(os.path.join(ROOT, "profiles", "r06_gemm3_pmc_round5_kernels_nt_head_tn.json")),
            "conv PMC (profiles/r06_conv_mfma_util_pmc.json, r06_conv3_pmc.json)": file_date(os.path.join(ROOT, "profiles", "r06_conv3_pmc.json")),
            "rocprofv3 --kernel-trace --stats (profiles/r06_rocprofv3_kernel_stats_summary.txt)": file_date(os.path.join(ROOT, "profiles", "r06_rocprofv3_kernel_stats_summary.txt")),
            "this line": time.strftime("%Y-%m-%d", time.gmtime())}
        print(json.dumps(out), flush=True)
    if dp is not None:
        if world > 1:
            dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
