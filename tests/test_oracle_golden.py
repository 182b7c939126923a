"""CPU: the oracle restatement reproduces every golden vector taken from the
unmodified reference (tests/golden/generate.py), bit for bit and dtype for
dtype.  This is what 'parity pinned' rests on."""
import os
import random

import numpy as np
import pytest
import torch

from oracle import replay as orc
from oracle import qmath
from oracle.sumtree import SumTree
from tests import scenario
from tests.golden.streams import scalar_kind

GOLDEN = scenario.GOLDEN


def _make(cfg, gamma):
    cls = orc.OraclePrioritizedReplay if cfg["mode"] == "per" else orc.OracleReplay
    return cls(**cfg["hist"], discount_function=orc.make_discount(gamma))


def _per_state(buf):
    if not isinstance(buf, orc.OraclePrioritizedReplay):
        return None
    cap = buf.tree.capacity
    leaves = buf.tree.nodes[cap:]
    n = len(buf.slot_record)
    slot_env = np.full(n, -1, dtype=np.int64)
    slot_base = np.full(n, -1, dtype=np.int64)
    for s, rec in enumerate(buf.slot_record):
        if rec is not None:
            slot_env[s], slot_base[s] = rec["env_id"], rec["env_buffer_offset"]
    return {
        "leaf_val": np.array([float(v) for v in leaves]),
        "leaf_kind": np.array([scalar_kind(v) for v in leaves], dtype=np.uint8),
        "leaf_exact": True,
        "free_slots": np.array(list(buf.free_slots), dtype=np.int64),
        "slot_env": slot_env, "slot_base": slot_base,
        "env_first": np.array([buf.env_first_offset[e] for e in sorted(buf.env_first_offset)]),
    }


@pytest.mark.parametrize("name", scenario.SCENARIOS)
def test_replay_scenarios(name):
    scenario.run(name, _make, exact_dtypes=True, per_state=_per_state)


def test_tree_cases():
    d = np.load(os.path.join(GOLDEN, "tree_cases.npz"))
    mk = {0: float, 1: np.float32, 2: np.float64}
    for case in ("f32_regime", "weak_regime", "mixed_regime", "f32_large"):
        cap = int(d[case + ".capacity"])
        tree = SumTree(cap)
        for i, (v, k) in enumerate(zip(d[case + ".leaf_val"], d[case + ".leaf_kind"])):
            tree.set_leaf(i, mk[int(k)](v))
        assert [float(v) for v in tree.nodes] == list(d[case + ".node_val"])
        assert [scalar_kind(v) for v in tree.nodes[1:]] == list(d[case + ".node_kind"][1:])
        for B in (8, 32):
            us = d["%s.B%d.uniforms" % (case, B)]
            seg = tree.total() / B
            got = [tree.descend(float(u) * seg + i * seg) for i, u in enumerate(us)]
            assert got == list(d["%s.B%d.index" % (case, B)])


def test_qmath_cases():
    d = np.load(os.path.join(GOLDEN, "qmath_cases.npz"))
    t = lambda k: torch.from_numpy(d[k])                           # noqa: E731
    f32 = lambda k: torch.from_numpy(d[k].astype(np.float32))      # noqa: E731
    for eps in (1e-3, 1e-2):
        x = t("vf.eps%g.x" % eps)
        assert torch.equal(qmath.vf_scale(x, eps), t("vf.eps%g.scale" % eps))
        assert torch.equal(qmath.vf_unscale(x, eps), t("vf.eps%g.unscale" % eps))
    for vf_eps, vtag in ((None, "none"), (1e-3, "1e-3")):
        for dq in (0, 1):
            tag = "tg.vf%s.dq%d" % (vtag, dq)
            y = qmath.nstep_target(
                qmath.dqn_bootstrap(t("tg.q_target"), t("tg.q_online") if dq else t("tg.q_target")),
                f32("tg.returns"), f32("tg.masks"), f32("tg.nsteps"), 0.97, vf_eps)
            assert torch.equal(y, t(tag + ".dqn"))
            y = qmath.nstep_target(
                qmath.iqn_bootstrap(t("tg.z_target"), t(tag + ".z_select")),
                f32("tg.returns"), f32("tg.masks"), f32("tg.nsteps"), 0.97, vf_eps)
            assert torch.equal(y, t(tag + ".iqn"))
    T = int(d["ls.timesteps"])
    for bm, tm in [("mean", None), ("sum", None), ("mean", "mean"), ("sum", "mean"), ("mean", "sum")]:
        for use_w in (0, 1):
            for kappa in (1.0, 0.5):
                tag = "ls.%s.%s.w%d.k%g" % (bm, tm, use_w, kappa)
                w = f32("ls.weights") if use_w else None
                for mode in ("huber", "mse"):
                    q = t("ls.q").clone().requires_grad_(True)
                    loss, rep = qmath.dqn_loss(q, t("ls.actions"), t("ls.y_dqn"), w, kappa, mode, T, bm, tm)
                    loss.backward()
                    assert torch.equal(loss.detach(), t(tag + ".dqn_%s.loss" % mode))
                    assert torch.equal(q.grad, t(tag + ".dqn_%s.grad" % mode))
                    assert torch.equal(rep, t(tag + ".dqn_%s.report" % mode))
                z = t("ls.z").clone().requires_grad_(True)
                loss, rep = qmath.iqn_loss(z, t("ls.taus"), t("ls.actions"), t("ls.y_iqn"), w, kappa, T, bm, tm)
                loss.backward()
                assert torch.equal(loss.detach(), t(tag + ".iqn.loss"))
                assert torch.equal(z.grad, t(tag + ".iqn.grad"))
                assert torch.equal(rep, t(tag + ".iqn.report"))


def test_network64_is_pinned_to_the_reference_network():
    """oracle/network64.py (the float64 anchor of tests/test_network_ab_gpu.py) against the reference's own
    SequentialModel + IQNPolicy run in float32 on the CPU (rltime/models/torch/modules/{cnn,lstm,fc}.py,
    policies/torch/{iqn,dqn}.py; fixture written by tests/golden/generate.py run_network64_pin from the imported
    reference): outputs within 1e-5, gradients of a fixed linear functional within 1e-4 (relative to the largest
    entry) — float32 rounding of the reference itself, nothing else."""
    import io
    import os
    import numpy as np
    import torch
    from oracle.network64 import Net64
    fx = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "network64_pin.npz"))
    sd = torch.load(io.BytesIO(fx["state_dict"].tobytes()), map_location="cpu", weights_only=True)
    T, B = fx["x"].shape[:2]
    N = fx["z"].shape[1]
    net = Net64(sd, [2, 1], N, "cpu")
    z64 = net.predict(torch.from_numpy(fx["x"]), torch.from_numpy(fx["hx"])[0], torch.from_numpy(fx["cx"])[0],
                      torch.from_numpy(fx["initials"]), torch.from_numpy(fx["taus"]), T)
    z = torch.from_numpy(fx["z"])
    assert z64.shape == z.shape == (T * B, N, z.shape[2])
    assert float((z64.detach().float() - z).abs().max() / z.abs().max()) <= 1e-5
    names = list(net.params())
    grads = torch.autograd.grad((z64 * torch.from_numpy(fx["weight"]).double()).sum(), [net.p[k] for k in names])
    checked = 0
    for k, g in zip(names, grads):
        want = torch.from_numpy(fx["grad." + k])
        assert float((g.float() - want).abs().max() / (want.abs().max() + 1e-12)) <= 1e-4, k
        checked += 1
    assert checked == sum(1 for k in fx.files if k.startswith("grad.")) >= 12
