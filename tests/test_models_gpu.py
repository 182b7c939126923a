"""GPU: the restated policies against the unmodified reference's outputs for the
same weights and inputs (tests/golden/model_cases.npz, the fixture
tests/test_models_cpu.py checks on the CPU), through the device paths: fused
frame conversion, fused LSTM sequence op, fused IQN head.  IQN quantile fractions
are replayed from the fixture (the reference drew them on the CPU generator).
Tolerance 1e-4 (north-star bar; CPU-fp32 reference vs GPU-fp32 kernels)."""
import os

import numpy as np
import pytest
import torch

from tests import scenario
from tests.test_models_cpu import CASES, T, B, _policy

pytestmark = pytest.mark.gpu
TOL = dict(rtol=1e-4, atol=1e-5)


def _cuda(pol, state):
    from rltime_amd.general.utils import deep_apply
    pol = pol.cuda()
    state = deep_apply(state, lambda x: torch.as_tensor(np.asarray(x)).cuda())
    return pol, state


@pytest.mark.parametrize("name", sorted(CASES))
def test_policy_matches_reference_on_device(name):
    d = np.load(os.path.join(scenario.GOLDEN, "model_cases.npz"))
    pol, state, rec = _policy(d, name)
    pol, state = _cuda(pol, state)
    if name + ".taus" in d.files:
        pol.tau_source = lambda count: torch.from_numpy(d[name + ".taus"])
    pred = pol.predict(state, T if rec else 1)
    if isinstance(pred, tuple):
        assert np.array_equal(pred[1].cpu().numpy(), d[name + ".taus"])
        pred = pred[0]
    np.testing.assert_allclose(pred.detach().cpu().numpy(), d[name + ".pred"], **TOL)
    if rec:
        np.testing.assert_allclose(pol.model.layers[1].last_state[0].cpu().numpy(), d[name + ".last_hx"], **TOL)


@pytest.mark.parametrize("full_forward", [True, False])
def test_burn_in_golden_on_device(full_forward):
    """multi_step_trainer.py:90-131 on the GPU, in both modes: the reference's exact
    call (actor_predict: whole head) and the default head-less pass that stops after
    the recurrent layer — the burned-in state only depends on the layers up to it."""
    from rltime_amd.training.iqn import IQN
    from rltime_amd.general.value_log import ValueLog
    d = np.load(os.path.join(scenario.GOLDEN, "model_cases.npz"))
    name = "iqn_lstm"
    pol, state, _ = _policy(d, name)
    pol, state = _cuda(pol, state)
    tr = IQN.__new__(IQN)
    tr.policy = tr.target_policy = pol
    tr.value_log = ValueLog()
    tr.burn_in_full_forward = full_forward
    tr._gpu_spans = []
    shaped = {"states": {
        "x": state["x"].view(T, B, 2, 12, 12), "layer0_state": {}, "layer2_state": {},
        "layer1_state": {k: v.view((T, B) + tuple(v.shape[1:])) for k, v in state["layer1_state"].items()}},
        "returns": torch.arange(T * B, dtype=torch.float32, device="cuda").view(T, B)}
    res = tr._burn_in(shaped, 2, do_target_states=False)
    np.testing.assert_allclose(res["states"]["layer1_state"]["hx"].cpu().numpy(), d[name + ".burn.hx"], **TOL)
    np.testing.assert_allclose(res["states"]["layer1_state"]["cx"].cpu().numpy(), d[name + ".burn.cx"], **TOL)
    assert np.array_equal(res["returns"].cpu().numpy(), d[name + ".burn.returns"])
