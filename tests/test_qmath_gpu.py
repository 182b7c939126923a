"""GPU parity of the fused target / loss kernels (through the C-ABI) against
the reference's golden vectors and against oracle/qmath.py on random shapes.
Tolerance: 1e-4 absolute + 1e-4 relative fp32 (BASELINE.json north_star)."""
import os

import numpy as np
import pytest
import torch

from tests import scenario

pytestmark = pytest.mark.gpu
TOL = dict(rtol=1e-4, atol=1e-4)


def _g(d, k, dtype=None):
    a = d[k]
    if dtype is not None:
        a = a.astype(dtype)
    return torch.from_numpy(a).cuda()


def test_targets_golden():
    from rltime_amd.training import qops
    d = np.load(os.path.join(scenario.GOLDEN, "qmath_cases.npz"))
    ret, ns, mk = (_g(d, k, np.float32) for k in ("tg.returns", "tg.nsteps", "tg.masks"))
    for vf_eps, vtag in ((None, "none"), (1e-3, "1e-3")):
        for dq in (0, 1):
            tag = "tg.vf%s.dq%d" % (vtag, dq)
            y = qops.q_target_dqn(_g(d, "tg.q_target"), _g(d, "tg.q_online") if dq else _g(d, "tg.q_target"),
                                  ret, ns, mk, 0.97, vf_eps)
            np.testing.assert_allclose(y.cpu().numpy(), d[tag + ".dqn"], **TOL)
            y = qops.q_target_iqn(_g(d, "tg.z_target"), _g(d, tag + ".z_select"), ret, ns, mk, 0.97, vf_eps)
            np.testing.assert_allclose(y.cpu().numpy(), d[tag + ".iqn"], **TOL)


def test_losses_golden():
    from rltime_amd.training import qops
    d = np.load(os.path.join(scenario.GOLDEN, "qmath_cases.npz"))
    T = int(d["ls.timesteps"])
    act = _g(d, "ls.actions")
    for bm, tm in [("mean", None), ("sum", None), ("mean", "mean"), ("sum", "mean"), ("mean", "sum")]:
        for use_w in (0, 1):
            for kappa in (1.0, 0.5):
                tag = "ls.%s.%s.w%d.k%g" % (bm, tm, use_w, kappa)
                w = _g(d, "ls.weights", np.float32) if use_w else None
                for mode in ("huber", "mse"):
                    q = _g(d, "ls.q").requires_grad_(True)
                    loss, rep = qops.dqn_loss(q, act, _g(d, "ls.y_dqn"), w, kappa, mode, T, bm, tm)
                    loss.backward()
                    np.testing.assert_allclose(loss.item(), d[tag + ".dqn_%s.loss" % mode], **TOL)
                    np.testing.assert_allclose(q.grad.cpu().numpy(), d[tag + ".dqn_%s.grad" % mode], **TOL)
                    np.testing.assert_allclose(rep.cpu().numpy(), d[tag + ".dqn_%s.report" % mode], **TOL)
                z = _g(d, "ls.z").requires_grad_(True)
                loss, rep = qops.iqn_loss(z, _g(d, "ls.taus"), act, _g(d, "ls.y_iqn"), w, kappa, T, bm, tm)
                loss.backward()
                np.testing.assert_allclose(loss.item(), d[tag + ".iqn.loss"], **TOL)
                np.testing.assert_allclose(z.grad.cpu().numpy(), d[tag + ".iqn.grad"], **TOL)
                np.testing.assert_allclose(rep.cpu().numpy(), d[tag + ".iqn.report"], **TOL)


@pytest.mark.parametrize("M,N,Nt,A", [(1, 1, 1, 2), (7, 8, 8, 3), (64, 32, 32, 6), (257, 32, 16, 18),
                                       (40, 64, 64, 4), (33, 70, 5, 9), (4096, 32, 32, 6)])
def test_iqn_random_shapes_vs_oracle(M, N, Nt, A):
    from oracle import qmath
    from rltime_amd.training import qops
    g = torch.Generator().manual_seed(M * 131 + N)
    z = torch.randn(M, N, A, generator=g) * 2
    zt = torch.randn(M, Nt, A, generator=g) * 2
    zs = torch.randn(M, N, A, generator=g) * 2
    taus = torch.rand(M * N, generator=g)
    act = torch.randint(0, A, (M,), generator=g)
    ret = torch.randn(M, generator=g)
    ns = torch.randint(1, 4, (M,), generator=g).float()
    mk = (torch.rand(M, generator=g) > 0.2).float()
    w = torch.rand(M, generator=g) + 0.1
    for vf_eps in (None, 1e-3):
        want = qmath.nstep_target(qmath.iqn_bootstrap(zt, zs), ret, mk, ns, 0.99, vf_eps)
        got = qops.q_target_iqn(zt.cuda(), zs.cuda(), ret.cuda(), ns.cuda(), mk.cuda(), 0.99, vf_eps)
        np.testing.assert_allclose(got.cpu().numpy(), want.numpy(), **TOL)
    y = torch.randn(M, Nt, generator=g) * 2
    T = 1
    z1 = z.clone().requires_grad_(True)
    l1, r1 = qmath.iqn_loss(z1, taus, act, y, w, 1.0, T, "mean", None)
    l1.backward()
    z2 = z.clone().cuda().requires_grad_(True)
    l2, r2 = qops.iqn_loss(z2, taus.cuda(), act.cuda(), y.cuda(), w.cuda(), 1.0, T, "mean", None)
    l2.backward()
    np.testing.assert_allclose(l2.item(), l1.item(), **TOL)
    np.testing.assert_allclose(r2.cpu().numpy(), r1.numpy(), **TOL)
    np.testing.assert_allclose(z2.grad.cpu().numpy(), z1.grad.numpy(), rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("M,A", [(1, 2), (100, 6), (5000, 18)])
def test_dqn_random_shapes_vs_oracle(M, A):
    from oracle import qmath
    from rltime_amd.training import qops
    g = torch.Generator().manual_seed(M)
    q = torch.randn(M, A, generator=g) * 2
    qt = torch.randn(M, A, generator=g) * 2
    act = torch.randint(0, A, (M,), generator=g)
    ret = torch.randn(M, generator=g)
    ns = torch.randint(1, 6, (M,), generator=g).float()
    mk = (torch.rand(M, generator=g) > 0.2).float()
    for dq in (False, True):
        want = qmath.nstep_target(qmath.dqn_bootstrap(qt, q if dq else qt), ret, mk, ns, 0.99, 1e-2)
        got = qops.q_target_dqn(qt.cuda(), (q if dq else qt).cuda(), ret.cuda(), ns.cuda(), mk.cuda(), 0.99, 1e-2)
        np.testing.assert_allclose(got.cpu().numpy(), want.numpy(), **TOL)
    y = torch.randn(M, generator=g)
    q1 = q.clone().requires_grad_(True)
    l1, r1 = qmath.dqn_loss(q1, act, y, None, 1.0, "huber", 1, "mean", None)
    l1.backward()
    q2 = q.clone().cuda().requires_grad_(True)
    l2, r2 = qops.dqn_loss(q2, act.cuda(), y.cuda(), None, 1.0, "huber", 1, "mean", None)
    l2.backward()
    np.testing.assert_allclose(l2.item(), l1.item(), **TOL)
    np.testing.assert_allclose(r2.cpu().numpy(), r1.numpy(), **TOL)
    np.testing.assert_allclose(q2.grad.cpu().numpy(), q1.grad.numpy(), rtol=1e-4, atol=1e-7)


@pytest.mark.parametrize("algo", ["dqn", "iqn"])
@pytest.mark.parametrize("vf_eps", [None, 1e-3])
def test_reference_target_hooks_compose_to_the_fused_targets(algo, vf_eps):
    """The reference's TorchTrainer contract (torch_trainer.py:91-147): a subclass
    supplies `_get_bootstrap_target_value`, the base class composes
    h(ret + gamma^n h^-1(v) mask).  rltime_amd's DQN / IQN override calc_target_values
    with one fused kernel; the hook path must give the same targets (1e-4)."""
    from rltime_amd.models.torch.utils import make_tensor
    from rltime_amd.training.dqn import DQN
    from rltime_amd.training.iqn import IQN
    from rltime_amd.training.torch_trainer import TorchTrainer
    g = torch.Generator().manual_seed(7)
    M, N, A = 300, 8, 5

    class Stub:
        def __init__(self, out):
            self.out = out

        def predict(self, x, timesteps):
            return (self.out, None) if algo == "iqn" else self.out

        def make_tensor(self, x, non_blocking=False):
            return make_tensor(x, "cuda", non_blocking)

    shape = (M, N, A) if algo == "iqn" else (M, A)
    tr = (IQN if algo == "iqn" else DQN).__new__(IQN if algo == "iqn" else DQN)
    tr.policy, tr.target_policy = Stub(torch.randn(shape, generator=g).cuda()), Stub(torch.randn(shape, generator=g).cuda())
    tr.gamma, tr.vf_scale_epsilon, tr.double_q = 0.97, vf_eps, True
    returns = torch.randn(M, generator=g).numpy().astype(np.float64)
    nsteps = torch.randint(1, 4, (M,), generator=g).numpy()
    masks = (torch.rand(M, generator=g) > 0.2).numpy().astype(np.int64)
    states = {"x": torch.zeros(M, 1).cuda()}
    fused = tr.calc_target_values(returns, states, masks, nsteps, 1)
    hooks = TorchTrainer.calc_target_values(tr, returns, states, masks, nsteps, 1)
    assert fused.shape == hooks.shape
    np.testing.assert_allclose(fused.cpu().numpy(), hooks.cpu().numpy(), rtol=1e-4, atol=1e-4)
