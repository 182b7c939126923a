"""CPU, world_size=2 over gloo: the N>1 path of the hot loop — env-sharded
replay configuration, flat-bucket gradient all-reduce, and the 3-doubles
exchange that turns shard-local importance weights into the globally
normalised ones (rltime_amd/parallel.py).  The RCCL path is the same code with
backend "nccl"."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from rltime_amd.parallel import DataParallel
    dp = DataParallel()
    # 1. parameter broadcast + gradient bucket (the .grad tensors are views of it)
    torch.manual_seed(rank)                                   # ranks start from different weights
    net = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.ReLU(), torch.nn.Linear(7, 3))
    dp.broadcast_parameters(net)
    start = [p.detach().clone() for p in net.parameters()]
    flat = dp.attach(net)
    g = torch.Generator().manual_seed(100 + rank)
    x = torch.randn(11, 5, generator=g)
    for _ in range(2):                                        # second pass: zero_grad keeps the views alive
        dp.zero_grad()
        net(x).pow(2).mean().backward()
    assert all(p.grad.untyped_storage().data_ptr() == flat.untyped_storage().data_ptr() for p in net.parameters())
    local = [p.grad.clone() for p in net.parameters()]
    dp.all_reduce_gradients(net)
    reduced = [p.grad.clone() for p in net.parameters()]
    # 3. lock-step guard: ready only when every rank is
    guard = (dp.all_ready(True), dp.all_ready(rank == 0), dp.all_ready(False))
    # an NHWC conv weight's .grad view has the parameter's own element order (the optimizer kernels walk both as one flat range)
    conv = torch.nn.Conv2d(3, 4, 2).to(memory_format=torch.channels_last)
    dp2 = DataParallel()
    flat2 = dp2.attach(conv)
    assert conv.weight.grad.stride() == conv.weight.stride() != conv.weight.contiguous().stride()
    conv(torch.randn(2, 3, 5, 5, generator=g)).sum().backward()
    assert conv.weight.grad.untyped_storage().data_ptr() == flat2.untyped_storage().data_ptr() and float(flat2.abs().sum()) > 0
    # 2. importance weights: every rank holds a shard of priorities, samples some
    beta = 0.6
    rs = np.random.RandomState(7)
    prios = [np.abs(rs.randn(40)) + 0.05, np.abs(rs.randn(25)) + 0.05]      # shard trees (leaf priorities)
    picks = [rs.randint(0, 40, 6), rs.randint(0, 25, 6)]
    p_l, pk = prios[rank], picks[rank]
    raw_local = (p_l[pk] / p_l.sum() * len(p_l)) ** (-beta)
    w_local = torch.from_numpy(raw_local / raw_local.max())
    w = dp.globalize_weights(w_local, torch.tensor(p_l.sum()), len(p_l), torch.tensor(raw_local.max()), beta)
    # what one tree over the union of the shards gives for the same sampled items
    P_g, N_g = prios[0].sum() + prios[1].sum(), 65
    raw_all = np.concatenate([(prios[r][picks[r]] / P_g * N_g) ** (-beta) for r in range(2)])
    want = (prios[rank][pk] / P_g * N_g) ** (-beta) / raw_all.max()
    out[rank] = dict(local=local, reduced=reduced, w=w.numpy(), want=want, start=start, guard=guard)
    dist.destroy_process_group()


def test_two_rank_exchange():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    r0, r1 = out[0], out[1]
    for a, b in zip(r0["start"], r1["start"]):
        assert torch.equal(a, b)                              # rank 0's weights everywhere
    assert r0["guard"] == r1["guard"] == (True, False, False)
    for a, b, m0, m1 in zip(r0["local"], r1["local"], r0["reduced"], r1["reduced"]):
        assert torch.allclose(m0, (a + b) / 2, atol=1e-7)
        assert torch.equal(m0, m1)
    for r in (r0, r1):
        np.testing.assert_allclose(r["w"], r["want"], rtol=1e-12)
    assert max(r0["w"].max(), r1["w"].max()) == 1.0


def test_shard_config():
    from rltime_amd.general.config import load_config
    from rltime_amd.parallel import shard_config
    cfg = load_config("synthetic_atari_iqn_lstm.json")
    shards = [shard_config(cfg, r, 8) for r in range(8)]          # strong: whole-job values split
    assert [s["acting"]["actor_envs"] for s in shards] == [32] * 8
    assert [s["acting"]["env_base"] for s in shards] == [32 * r for r in range(8)]
    assert [s["acting"]["total_envs"] for s in shards] == [256] * 8
    assert shards[3]["training"]["args"]["history_mode"]["args"]["size"] == 125000
    assert shards[3]["training"]["args"]["mbatch_size"] == 64       # global B=512
    assert cfg["acting"]["actor_envs"] == 256       # input untouched
    # strong: step-denominated settings count whole-job acted steps -> 1/R per rank (rounded up)
    from rltime_amd.parallel import STEP_FIELDS
    full = cfg["training"]["args"]
    for key in STEP_FIELDS:
        if full.get(key):
            assert shards[5]["training"]["args"][key] == -(-full[key] // 8), key
            assert shard_config(cfg, 1, 4, "weak")["training"]["args"][key] == full[key], key
    assert any(full.get(k) for k in ("total_steps", "target_update_freq"))
    weak = [shard_config(cfg, r, 4, "weak") for r in range(4)]     # weak: every rank keeps the configured job
    assert [s["acting"]["actor_envs"] for s in weak] == [256] * 4
    assert [s["acting"]["env_base"] for s in weak] == [0, 256, 512, 768]
    assert weak[1]["acting"]["total_envs"] == 1024
    assert weak[2]["training"]["args"]["mbatch_size"] == 512
    assert weak[2]["training"]["args"]["history_mode"]["args"]["size"] == 1000000


class _Rec(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.cell = torch.nn.Linear(6, 6)

    @staticmethod
    def is_recurrent():
        return True

    def forward(self, x):
        return torch.tanh(self.cell(x))


class _Front(torch.nn.Linear):
    @staticmethod
    def is_recurrent():
        return False


class _ToyPolicy(torch.nn.Module):
    """The shape DataParallel's automatic buckets look for: policy.model.layers = [front, recurrent, last] + heads."""

    def __init__(self):
        super().__init__()
        self.model = torch.nn.Module()
        self.model.layers = torch.nn.ModuleList([_Front(5, 6), _Rec(), _Front(6, 4)])
        self.out_layer = torch.nn.Linear(4, 3)
        self.unused = torch.nn.Linear(2, 2)            # never takes part in the loss: its bucket cannot complete by hooks

    def forward(self, x, use_head=True):
        for layer in self.model.layers:
            x = layer(x)
        return self.out_layer(x) if use_head else x


def _bucket_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from rltime_amd.parallel import DataParallel
    dp = DataParallel()
    torch.manual_seed(3)
    net = _ToyPolicy()
    dp.broadcast_parameters(net)
    flat = dp.attach(net)
    spans = [(b["lo"], b["hi"], b["count"]) for b in dp._buckets]
    names = {id(p): n for n, p in net.named_parameters()}
    order = [names[id(p)] for p in dp._params]
    g = torch.Generator().manual_seed(50 + rank)
    res = []
    for step, use_head in enumerate((True, True, False)):       # third pass: the head gets no gradient at all
        x = torch.randn(9, 5, generator=g)
        dp.zero_grad()
        net(x, use_head).pow(2).mean().backward()
        issued_by_hooks = [b["done"] for b in dp._buckets]
        dp.all_reduce_gradients(net)
        res.append(dict(issued=issued_by_hooks, reduced=flat.clone()))
        # the same gradients without any exchange, for the expected mean
        ref = _ToyPolicy()
        ref.load_state_dict(net.state_dict())
        ref(x, use_head).pow(2).mean().backward()
        local = torch.cat([(dict(ref.named_parameters())[n].grad if dict(ref.named_parameters())[n].grad is not None
                            else torch.zeros_like(dict(ref.named_parameters())[n])).reshape(-1) for n in order])
        res[-1]["local"] = local
    out[rank] = dict(spans=spans, order=order, res=res, overlapped=dp.buckets_overlapped)
    dist.destroy_process_group()


def test_bucketed_gradient_all_reduce_from_backward_hooks():
    """Three buckets in backward order (heads + last layer, recurrent layer, front), each reduced exactly once — by
    the post-accumulate hook of its last gradient when every parameter of it (and of the buckets before it) got one,
    by all_reduce_gradients() otherwise — and the result is the mean of the ranks' gradients, bit-identical on both."""
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_bucket_worker, args=(world, port, out), nprocs=world, join=True)
    r0, r1 = out[0], out[1]
    assert r0["order"] == r1["order"]
    order = r0["order"]
    first_front = order.index("model.layers.0.weight")
    first_rec = order.index("model.layers.1.cell.weight")
    assert order.index("out_layer.weight") < first_rec < first_front          # head -> recurrent -> front
    assert order.index("model.layers.2.weight") < first_rec                    # the model's last layer rides with the heads
    assert len(r0["spans"]) == 3 and r0["spans"][0][0] == 0 and r0["spans"][-1][1] == len(r0["res"][0]["reduced"])
    for step in range(3):
        a, b = r0["res"][step], r1["res"][step]
        assert torch.equal(a["reduced"], b["reduced"])
        assert torch.allclose(a["reduced"], (a["local"] + b["local"]) / 2, atol=1e-7)
        # `unused` sits in the head bucket: that bucket never completes by hooks, so nothing may go out early
        # (buckets are issued strictly in order) and all_reduce_gradients() reduces all three
        assert a["issued"] == [False, False, False]
    assert r0["overlapped"] == 0


def _bucket_worker_all_used(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from rltime_amd.parallel import DataParallel
    dp = DataParallel()
    torch.manual_seed(4)
    net = _ToyPolicy()
    del net.unused
    dp.broadcast_parameters(net)
    flat = dp.attach(net)
    x = torch.randn(9, 5, generator=torch.Generator().manual_seed(60 + rank))
    dp.zero_grad()
    net(x).pow(2).mean().backward()
    issued = [b["done"] for b in dp._buckets]
    dp.all_reduce_gradients(net)
    names = {id(p): n for n, p in net.named_parameters()}
    ref = _ToyPolicy()
    del ref.unused
    ref.load_state_dict(net.state_dict())
    ref(x).pow(2).mean().backward()
    grads = dict(ref.named_parameters())
    local = torch.cat([grads[names[id(p)]].grad.reshape(-1) for p in dp._params])
    reduced = flat.clone()
    # a second backward before zero_grad() must not slip through un-reduced, and a second attach() must replace the
    # hooks of the first (not add a second set that would fire every bucket early)
    try:
        net(x).pow(2).mean().backward()
        twice = "no error"
    except RuntimeError as e:
        twice = str(e)
    versions = [p._version for p in net.parameters()]
    dp.broadcast_parameters(net)                             # must move the version counters (caches key on them)
    moved = all(p._version > v for p, v in zip(net.parameters(), versions))
    flat2 = dp.attach(net)
    dp.zero_grad()
    net(x).pow(2).mean().backward()
    issued2 = [b["done"] for b in dp._buckets]
    dp.all_reduce_gradients(net)
    out[rank] = dict(issued=issued, reduced=reduced, overlapped=dp.buckets_overlapped, local=local, twice=twice, moved=moved,
                     hooks=len(dp._hooks), params=len(dp._params), issued2=issued2, reduced2=flat2.clone())
    dist.destroy_process_group()


def test_every_bucket_goes_out_from_the_hooks_when_all_parameters_get_gradients():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_bucket_worker_all_used, args=(world, port, out), nprocs=world, join=True)
    assert out[0]["issued"] == out[1]["issued"] == [True, True, True]
    assert torch.equal(out[0]["reduced"], out[1]["reduced"])
    assert torch.allclose(out[0]["reduced"], (out[0]["local"] + out[1]["local"]) / 2, atol=1e-7)
    for r in (out[0], out[1]):
        assert "already reduced" in r["twice"]
        assert r["moved"] and r["hooks"] == r["params"]
        assert r["issued2"] == [True, True, True] and r["overlapped"] == 6
    assert torch.equal(out[0]["reduced2"], out[1]["reduced2"])
    assert torch.allclose(out[0]["reduced2"], out[0]["reduced"], atol=1e-7)      # same weights, same inputs
