"""GPU: true resume (the reference checkpoints weights only,
policy_trainer.py:170-185).  Three fresh processes run the tiny recurrent-IQN /
prioritized-replay config through `rltime_amd.train.train`:

  A  uninterrupted, 2H acted steps
  B  the same run stopped after H acted steps, writing a full checkpoint
     (weights, Adam state, counters, RNG streams, replay shard + trees, actor state)
  C  a new process resumed from B's directory, running to 2H

C's per-learner-step loss and grad-norm series must continue B's exactly as A
does.  Bit-identity is demanded whenever the kernels are run-to-run deterministic
(checked by running A twice); otherwise 1e-5 closeness."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
H = 480


def _run(tmp, name, **kw):
    out = os.path.join(tmp, name + ".json")
    cmd = [sys.executable, os.path.join(ROOT, "tests", "resume_driver.py"), "--log-dir", tmp, "--name", name,
           "--total", str(2 * H), "--out", out]
    for k, v in kw.items():
        if v is not None:
            cmd += ["--" + k, str(v)]
    p = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-3000:]
    return json.load(open(out))


def test_resumed_run_continues_the_uninterrupted_series(tmp_path):
    tmp = str(tmp_path)
    a = _run(tmp, "a", full=0)
    a2 = _run(tmp, "a2", full=0)
    b = _run(tmp, "b", stop=H, full=1)
    assert os.path.isfile(os.path.join(tmp, "b", "resume", "train_state_rank0.pt"))
    assert os.path.isfile(os.path.join(tmp, "b", "resume", "replay_rank0.snap"))
    c = _run(tmp, "c", resume=os.path.join(tmp, "b"), full=0)
    nb = len(b["qloss"])
    assert 10 < nb < len(a["qloss"])
    assert nb + len(c["qloss"]) == len(a["qloss"]), (nb, len(c["qloss"]), len(a["qloss"]))
    assert b["steps_at"] + c["steps_at"] == a["steps_at"]           # same acted-step schedule
    assert c["final_steps"] == a["final_steps"]
    deterministic = a["qloss"] == a2["qloss"] and a["grad_norm"] == a2["grad_norm"]
    assert b["qloss"] == a["qloss"][:nb] or not deterministic
    for key in ("qloss", "grad_norm"):
        joined = b[key] + c[key]
        if deterministic:
            assert joined == a[key], key                              # bit-identical continuation
        else:
            np.testing.assert_allclose(joined, a[key], rtol=1e-5, atol=1e-7, err_msg=key)
    if deterministic:
        assert c["param_sum"] == a["param_sum"]
    print("kernels deterministic run-to-run: %s; resumed %d + %d learner steps" % (deterministic, nb, len(c["qloss"])))
