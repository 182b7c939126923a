"""GPU: the N>1 path of THE LOOP, executed for real — two ranks, each a full
trainer process (device actor, its own replay shard, sampling, gather, burn-in,
IQN targets, forward/backward, gradient all-reduce, clip + Adam, priority
update).  Transport (`_join`): on a box with at least as many GPUs as ranks every
rank takes its OWN device and the group is "nccl" = RCCL over xGMI, with the gloo
side group bench.py / rltime_amd.train use for the host-side lock-step flag — the
bucketed async ReduceOp.AVG from autograd's hooks, the (R, 3) weight exchange and
mirl_replay_sample_global then run over real RCCL with the same assertions.  On a
one-GPU box both ranks share cuda:0 over gloo (RCCL refuses two ranks on one
device): rltime_amd/parallel.py only uses all_reduce / broadcast, and under gloo
the device tensors are staged through the host inside DataParallel._all_reduce.
MIRL_TEST_BACKEND=gloo|nccl forces one of the two.

Checked:
  (i)  the parameters are bit-identical across the ranks after every learner step
       (different initial seeds per rank: the broadcast and the gradient bucket
       must make them so);
  (ii) the globalised importance weights equal what ONE reference tree over the
       union of both shards' leaves gives (oracle.sumtree.SumTree built from the
       leaves read back with mirl_replay_tree_nodes, weights by
       prioritized_replay_history.py:327,353-354), for every step;
  (iii) the torchrun entry of rltime_amd.train runs at world 1 with the
       collectives forced.
"""
import copy
import hashlib
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CNN = {"type": "cnn", "args": {"layers": [{"filters": 8, "kernel": 4, "stride": 2},
                                          {"filters": 8, "kernel": 3, "stride": 1}]}}
CONFIG = {
    "acting": {"actor_envs": 8, "exploration": {"type": "epsilon_greedy", "args": {
        "eps_start": 1.0, "eps_final": 0.05, "exploration_fraction": 0.5}}},
    "env": "synthetic-atari", "env_args": {"frame_shape": [2, 20, 20], "n_actions": 4, "done_prob": 0.02},
    "model": {"type": "sequential", "args": {"layer_configs": [
        CNN, {"type": "lstm", "args": {"num_units": 16}}, {"type": "fc", "args": {"fc_size": 32}}]}},
    "policy_args": {"dueling": True, "embedding_dim": 8, "num_sampling_quantiles": 4},
    "training": {"type": "iqn", "args": {
        "clip_rewards": False, "vf_scale_epsilon": 1e-3, "gamma": 0.99, "mbatch_size": 8, "nstep_train": 8,
        "burn_in_timesteps": 4, "nstep_target": 2, "lr": 1e-3, "double_q": True, "rnn_bootstrap": True,
        "clip_grad": 10.0, "target_update_freq": 96, "total_steps": 10 ** 9, "log_freq": 10 ** 9,
        "warmup_steps": 0,
        "history_mode": {"type": "prioritized_replay", "args": {
            "size": 1200, "train_frequency": 4, "alpha": 0.9, "beta": 0.6, "max_weight_factor": 0.9}}}},
}
STEPS = 12


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _join(rank, world, port):
    """Process group of one test rank -> (DataParallel, device index, backend name)."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")    # dmabuf IPC only on these hosts (RCCL needs it)
    sys.path.insert(0, ROOT)
    import datetime
    import torch
    import torch.distributed as dist
    from rltime_amd.parallel import DataParallel
    backend = os.environ.get("MIRL_TEST_BACKEND") or ("nccl" if torch.cuda.device_count() >= world else "gloo")
    limit = datetime.timedelta(seconds=120)
    if backend == "nccl":
        assert torch.cuda.device_count() >= world, "RCCL needs one device per rank"
        torch.cuda.set_device(rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, timeout=limit, device_id=torch.device("cuda", rank))
        return DataParallel(host_group=dist.new_group(backend="gloo", timeout=limit)), rank, backend
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=limit)
    return DataParallel(), 0, backend


def _rank_main(rank, world, port, out_dir, global_sampling=False):
    import faulthandler
    import random
    faulthandler.dump_traceback_later(240, exit=True)        # a stuck rank reports where, instead of hanging the suite
    dp, dev, backend = _join(rank, world, port)
    import torch
    import torch.distributed as dist
    from rltime_amd.general.loggers import NullLogger
    from rltime_amd.general.type_registry import get_registered_type
    from rltime_amd.parallel import shard_config
    from rltime_amd.train import create_actors
    cfg = shard_config(copy.deepcopy(CONFIG), rank, world, "strong")
    assert cfg["acting"]["actor_envs"] == 8 // world and cfg["training"]["args"]["mbatch_size"] == 8 // world
    if global_sampling:
        # one tree over the union of the shards (mirl_replay_sample_global): padded batches
        cfg["training"]["args"]["global_sampling"] = True
        cfg["training"]["args"]["history_mode"]["args"]["device_rng"] = True
    random.seed(10 + rank); np.random.seed(20 + rank); torch.manual_seed(30 + rank)   # noqa: E702
    actors = create_actors(cfg, "cuda", device_acting=True)
    trainer = get_registered_type("trainers", "iqn")(
        logger=NullLogger(), actors=actors, model_config=cfg["model"], policy_args=cfg["policy_args"])
    trainer.data_parallel = dp
    trainer.setup(**cfg["training"]["args"])
    hist = trainer.history_buffer
    T = cfg["training"]["args"]["nstep_train"]
    rec = {"hash": [], "w": [], "slots": [], "leaf_v": [], "leaf_k": [], "active": [], "beta": [], "target_hash": []}
    inner = trainer._weights

    def tap(extra):
        w = inner(extra)
        if global_sampling:
            assert hist.last_sample.get("global") and w.reshape(T, -1).shape[1] == hist.last_sample["rows"] >= 4   # padded rows
            rec["w"].append(np.zeros(1)); rec["slots"].append(np.zeros(1, np.int64))     # noqa: E702
            rec["leaf_v"].append(np.zeros(1)); rec["leaf_k"].append(np.zeros(1, np.uint8))   # noqa: E702
            rec["active"].append(0); rec["beta"].append(hist.last_beta)                  # noqa: E702
            return w
        v, k, _ = hist.tree_nodes()
        cap = len(v) // 2
        rec["w"].append(w.reshape(T, -1)[0].double().cpu().numpy())
        rec["slots"].append(hist.last_sample["slot"].cpu().numpy().astype(np.int64))
        rec["leaf_v"].append(v[cap:].copy())
        rec["leaf_k"].append(k[cap:].copy())
        rec["active"].append(hist.stats()["active_sequences"])
        rec["beta"].append(hist.last_beta)
        return w
    trainer._weights = tap

    def digest(policy):
        flat = torch.cat([p.detach().reshape(-1) for p in policy.parameters()])
        return hashlib.sha1(flat.cpu().numpy().tobytes()).hexdigest()

    rec["hash"].append(digest(trainer.policy))            # after the broadcast, before any step
    rec["target_hash"].append(digest(trainer.target_policy))
    done, guard = 0, 0
    while done < STEPS:
        guard += 1
        assert guard < 400, "no learner step was reached"
        if trainer.loop_iteration():
            done += 1
            rec["hash"].append(digest(trainer.policy))
            rec["target_hash"].append(digest(trainer.target_policy))
    torch.cuda.synchronize()
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank),
             hash=np.array(rec["hash"]), target_hash=np.array(rec["target_hash"]),
             w=np.stack(rec["w"]), slots=np.stack(rec["slots"]), leaf_v=np.stack(rec["leaf_v"]),
             leaf_k=np.stack(rec["leaf_k"]), active=np.array(rec["active"]), beta=np.array(rec["beta"]),
             steps=trainer.steps, backend=backend, overlapped=dp.buckets_overlapped)
    hist.close()
    dist.destroy_process_group()
    faulthandler.cancel_dump_traceback_later()


def _kinded(v, k):
    """(value, kind) as stored on the device -> the scalar object the reference's
    list-of-scalars tree would hold (csrc/np_emul.h: 0 weak python float, 1 f32, 2 f64)."""
    if k == 1:
        return np.float32(v)
    if k == 2:
        return np.float64(v)
    return float(v)


@pytest.mark.parametrize("sampling", ["per_shard", "global"])
def test_two_ranks_full_loop_on_one_gpu(tmp_path, sampling):
    import torch.multiprocessing as mp
    from oracle.sumtree import SumTree
    world = 2
    mp.spawn(_rank_main, args=(world, _free_port(), str(tmp_path), sampling == "global"), nprocs=world, join=True)
    r = [np.load(tmp_path / ("rank%d.npz" % i)) for i in range(world)]
    # (i) replicas stay bit-identical (initial broadcast included), and they moved
    assert list(r[0]["hash"]) == list(r[1]["hash"])
    assert list(r[0]["target_hash"]) == list(r[1]["target_hash"])
    assert len(set(r[0]["hash"])) == STEPS + 1
    assert len(set(r[0]["target_hash"])) > 1                      # the target sync fired on both
    assert int(r[0]["steps"]) == int(r[1]["steps"])
    if sampling == "global":
        return          # the sampler itself is checked in test_global_sampling_is_one_tree_over_the_union_of_shards
    # (ii) importance weights == one reference tree over the union of the shards
    assert r[0]["w"].shape[0] == STEPS
    worst = 0.0
    for s in range(STEPS):
        n0 = r[0]["leaf_v"][s].shape[0]
        leaves = [_kinded(v, k) for i in range(world) for v, k in zip(r[i]["leaf_v"][s], r[i]["leaf_k"][s])]
        cap = 1
        while cap < len(leaves):
            cap *= 2
        tree = SumTree(cap)
        for j, x in enumerate(leaves):
            if x != 0:
                tree.set_leaf(j, x)
        p_sum = tree.total()
        active = int(r[0]["active"][s] + r[1]["active"][s])
        beta = float(r[0]["beta"][s])
        assert beta == float(r[1]["beta"][s])
        raw = [np.array([((tree.leaf(int(sl) + i * n0) / p_sum) * active) ** (-beta) for sl in r[i]["slots"][s]],
                        dtype=np.float64) for i in range(world)]
        top = max(x.max() for x in raw)
        for i in range(world):
            want = raw[i] / top
            np.testing.assert_allclose(r[i]["w"][s], want, rtol=5e-6, atol=0)
            worst = max(worst, float(np.max(np.abs(r[i]["w"][s] - want) / want)))
        assert max(r[0]["w"][s].max(), r[1]["w"][s].max()) == pytest.approx(1.0, rel=1e-6)
    print("max rel deviation of globalised weights vs union-tree oracle: %.2e" % worst)


def test_one_rank_over_rccl_with_the_collectives_forced(tmp_path, monkeypatch):
    """What a one-GPU box CAN run of the RCCL path: the same full trainer rank at world size 1 over `nccl` with the
    collectives forced on (BENCH_FORCE_DIST=1) — process-group init with a device id, the gloo side group, parameter
    broadcast, the three gradient buckets issued asynchronously as ReduceOp.AVG from autograd's post-accumulate hooks and
    waited for on the stream, the (R, 3) importance-weight exchange.  Twelve learner steps must move the weights every step
    and every bucket must have gone out from a hook."""
    import torch.multiprocessing as mp
    monkeypatch.setenv("BENCH_FORCE_DIST", "1")
    monkeypatch.setenv("MIRL_TEST_BACKEND", "nccl")
    mp.spawn(_rank_main, args=(1, _free_port(), str(tmp_path), False), nprocs=1, join=True)
    r = np.load(tmp_path / "rank0.npz")
    assert str(r["backend"]) == "nccl"
    assert len(set(r["hash"])) == STEPS + 1 and len(set(r["target_hash"])) > 1
    assert int(r["overlapped"]) >= 3 * STEPS                      # head, recurrent and conv bucket of every step, from hooks
    assert np.all(np.isfinite(r["w"])) and r["w"].max() == pytest.approx(1.0, rel=1e-6)


def test_train_entry_under_torchrun_world1_forced_collectives(tmp_path):
    """`python -m torch.distributed.run ... -m rltime_amd.train` (the product entry,
    not bench.py): process-group init, config sharding, parameter broadcast,
    bucketed gradient all-reduce and the importance-weight exchange all execute
    (BENCH_FORCE_DIST=1 keeps them on at world 1; gloo so that it also runs next
    to another process on the same GPU)."""
    env = dict(os.environ, BENCH_FORCE_DIST="1", PYTHONPATH=ROOT)
    upd = {"acting": {"actor_envs": 8},
           "env_args": {"frame_shape": [2, 20, 20], "n_actions": 4},
           "model": CONFIG["model"], "policy_args": CONFIG["policy_args"],
           "training": {"args": dict(CONFIG["training"]["args"], total_steps=600, log_freq=200, warmup_steps=100,
                                     lr_anneal=False)}}
    import json
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           "-m", "rltime_amd.train", "synthetic_atari_iqn_lstm.json", "--backend", "gloo", "--seed", "3",
           "--log-dir", str(tmp_path), "--log-name", "run", "--conf-update", json.dumps(upd)]
    p = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    rows = [json.loads(line) for line in open(tmp_path / "run" / "train.json")]
    assert rows and rows[-1]["this_interval"]["steps_trained"] > 0


# --------------------------------------------------------------------------
# exact global sampling over two shards == ONE tree over their union
# --------------------------------------------------------------------------
GS = dict(size=600, train_frequency=0, nstep_target=2, nstep_train=8, prefix_steps=4,
          alpha=0.9, beta=0.6, beta_anneal=True, max_weight_factor=0.9)
GS_B, GS_DRAWS = 8, 6


def _global_rank(rank, world, port, out_dir, imbalance=1.0):
    import faulthandler
    faulthandler.dump_traceback_later(240, exit=True)
    dp, dev, backend = _join(rank, world, port)
    import torch
    import torch.distributed as dist
    from rltime_amd.history import PrioritizedReplayHistoryBuffer
    from tests.golden.streams import StreamSpec, vector_steps, as_reference_samples
    E = 5
    spec = StreamSpec(seed=80 + rank, num_envs=E, frame_shape=(1, 8, 8), lstm_units=4, n_actions=4,
                      done_prob=0.05, env_base=rank * E)
    buf = PrioritizedReplayHistoryBuffer(**GS, gamma=0.99, device_rng=True, num_envs=E, env_base=rank * E)
    buf.enable_global_sampling(dp)
    rec = {k: [] for k in ("stratum", "slot", "weight", "leaf", "seed", "call", "beta", "active", "kept", "dropped", "rows", "mass")}
    step_no, calls = 0, 0
    feeds = [70, 12, 0, 25, 9, 40]
    for draw in range(GS_DRAWS):
        for st in vector_steps(spec, feeds[draw], start_step=step_no):
            buf.update(as_reference_samples(spec, st))
        step_no += feeds[draw]
        progress = draw / float(GS_DRAWS)
        v, k, _ = buf.tree_nodes()                        # the tree the draw will see
        batch = buf.get_train_data(GS_B, train_progress=progress)
        calls += 1
        assert batch is not None
        last = buf.last_sample
        rows = last["slot"].shape[0]
        # rows come from the exchanged shard totals: the strata bound of THIS shard, rounded up to the quantum
        share = float(last["shard"][rank, 0] / last["shard"][:, 0].sum())
        assert batch["states"]["x"].shape[1] == rows == buf.global_rows_for(GS_B, share)[0]
        assert rows >= int(np.ceil(GS_B * world * share)) + 2 and rows % 4 == 0
        pad_to = rows + (-rows) % 64
        rec["rows"].append(rows)
        rec["mass"].append(float(last["shard"][rank, 0]))
        cap = len(v) // 2
        fill = lambda a, v: np.concatenate([a, np.full(pad_to - rows, v, a.dtype)])   # noqa: E731  (rows vary per draw)
        rec["stratum"].append(fill(last["stratum"].cpu().numpy(), -1))
        rec["slot"].append(fill(last["slot"].cpu().numpy(), -1))
        rec["weight"].append(fill(last["weight"].double().cpu().numpy(), 0.0))
        rec["leaf"].append(v[cap:].copy())
        rec["seed"].append(buf._seed)
        rec["call"].append(calls)
        rec["beta"].append(buf.last_beta)
        rec["active"].append(buf.stats()["active_sequences"])
        rec["kept"].append(float(last["stats"][2].item()))
        rec["dropped"].append(float(last["stats"][3].item()))
        # padding rows: weight 0, no loss index
        pad = last["slot"].cpu().numpy() < 0
        w = batch["extra_data"]["importance_weights"][0].cpu().numpy()
        li = batch["extra_data"]["loss_indices"][GS["prefix_steps"]:].cpu().numpy()
        assert np.all(w[pad] == 0) and np.all(li[:, pad] == -1) and np.all(li[:, ~pad, 0] >= rank * E)
        P = GS["prefix_steps"]
        idx = batch["extra_data"]["loss_indices"][P:].reshape(-1, 2)
        g = torch.Generator(device="cuda").manual_seed(draw * 10 + rank)
        buf.update_losses(idx, torch.randn(idx.shape[0], device="cuda", generator=g) * 0.7 * (imbalance if rank == 1 else 1.0))
    assert buf.check_dropped_strata() == 0
    np.savez(os.path.join(out_dir, "gs_rank%d.npz" % rank), **{k: np.array(v) for k, v in rec.items()})
    buf.close()
    dist.destroy_process_group()
    faulthandler.cancel_dump_traceback_later()


@pytest.mark.parametrize("imbalance", [1.0, 40.0])
def test_global_sampling_is_one_tree_over_the_union_of_shards(tmp_path, imbalance):
    """imbalance 40: rank 1 reports 40x larger TD errors, so after a few draws its shard holds
    well over 3x the priority mass of rank 0 and owns most strata — no stratum may be lost
    (every rank sizes its padded batch from the exchanged shard totals).
    `enable_global_sampling` (mirl_replay_sample_global): with the same Philox
    stream on both ranks, the strata of the GLOBAL priority mass go to the shard whose
    cumulative range contains them.  Checked against ONE reference tree
    (oracle.sumtree.SumTree, float64 leaves) over the concatenation of both shards'
    leaves with host-recomputed Philox uniforms: every stratum is drawn exactly once,
    by the right rank, at the leaf the union tree's descent reaches (a stratum whose
    mass sits within 1e-6 relative of a leaf boundary may land on the neighbour: the
    shards sum their float32 leaves in float32 like the reference does, the union tree
    in float64), and the importance weights are the union tree's."""
    import torch.multiprocessing as mp
    from oracle.sumtree import SumTree
    from tests.test_replay_gpu import _philox_u53
    world = 2
    mp.spawn(_global_rank, args=(world, _free_port(), str(tmp_path), imbalance), nprocs=world, join=True)
    r = [np.load(tmp_path / ("gs_rank%d.npz" % i)) for i in range(world)]
    Bg = GS_B * world
    exact = near = 0
    if imbalance > 1:
        ratio = r[1]["mass"][-1] / r[0]["mass"][-1]
        assert ratio >= 3.0, ratio                                   # the scenario really is imbalanced
        assert r[1]["rows"][-1] > r[0]["rows"][-1] and r[1]["rows"][-1] > GS_B * 1.25 + 4   # beyond the old fixed padding
        print("priority-mass ratio rank1/rank0 at the last draw: %.1f, rows %d vs %d" % (ratio, r[1]["rows"][-1], r[0]["rows"][-1]))
    for s in range(GS_DRAWS):
        assert r[0]["seed"][s] == r[1]["seed"][s] and r[0]["call"][s] == r[1]["call"][s]
        n0 = r[0]["leaf"][s].shape[0]
        leaves = np.concatenate([r[0]["leaf"][s], r[1]["leaf"][s]]).astype(np.float64)
        tree = SumTree(2 * n0)
        for j, x in enumerate(leaves):
            if x != 0:
                tree.set_leaf(j, np.float64(x))
        Pg = float(tree.total())
        Ng = int(r[0]["active"][s] + r[1]["active"][s])
        beta = float(r[0]["beta"][s])
        seg = Pg / Bg
        got = {}
        for i in range(world):
            assert r[i]["dropped"][s] == 0
            for st, sl, w in zip(r[i]["stratum"][s], r[i]["slot"][s], r[i]["weight"][s]):
                if st >= 0:
                    assert st not in got
                    got[int(st)] = (i, int(sl), float(w))
            assert int(r[i]["kept"][s]) == int((r[i]["stratum"][s] >= 0).sum())
        assert sorted(got) == list(range(Bg))                       # every stratum exactly once
        raws = {}
        for st in range(Bg):
            mass = (_philox_u53(int(r[0]["seed"][s]), int(r[0]["call"][s]), st) + st) * seg
            want = tree.descend(mass)
            rank_got, slot_got, _ = got[st]
            leaf_got = rank_got * n0 + slot_got
            if leaf_got == want:
                exact += 1
            else:
                # a boundary case: the mass must sit within 1e-6 (relative) of the edge between the two leaves
                prefix = float(np.sum(leaves[:max(leaf_got, want)]))
                assert abs(leaf_got - want) == 1 or leaves[min(leaf_got, want) + 1:max(leaf_got, want)].sum() == 0
                assert abs(mass - prefix) <= 1e-6 * Pg, (s, st, mass, prefix)
                near += 1
            raws[st] = ((leaves[leaf_got] / Pg) * Ng) ** (-beta)
        top = max(raws.values())
        for st in range(Bg):
            want_w = raws[st] / top * (float(r[got[st][0]]["rows"][s]) / float(GS_B))
            assert abs(got[st][2] - want_w) <= 5e-6 * want_w, (s, st)
    assert exact >= GS_DRAWS * Bg - 2
    print("global sampling: %d strata identical to the union tree, %d at a leaf boundary" % (exact, near))
