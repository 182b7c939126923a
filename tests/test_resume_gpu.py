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


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _run(tmp, name, ranks=1, **kw):
    out = os.path.join(tmp, name + ".json")
    cmd = [sys.executable]
    if ranks > 1:
        cmd += ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks),
                "--master-addr", "127.0.0.1", "--master-port", str(_free_port())]
    cmd += [os.path.join(ROOT, "tests", "resume_driver.py"), "--log-dir", tmp, "--name", name,
            "--total", str(2 * H), "--out", out]
    for k, v in kw.items():
        if v is not None:
            cmd += ["--" + k, str(v)]
    p = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-3000:]
    if ranks > 1:
        return [json.load(open(out.replace(".json", "_rank%d.json" % r))) for r in range(ranks)]
    return json.load(open(out))


@pytest.mark.parametrize("overlap", [0, 1])
def test_resumed_run_continues_the_uninterrupted_series(tmp_path, overlap):
    """overlap=1: acting of iteration k+1 on a second stream (overlap_acting=True); the
    checkpoint then also carries the pipeline state and the actors' weight copy is
    refreshed from the restored weights."""
    tmp = str(tmp_path)
    a = _run(tmp, "a", full=0, overlap=overlap)
    a2 = _run(tmp, "a2", full=0, overlap=overlap)
    b = _run(tmp, "b", stop=H, full=1, overlap=overlap)
    assert os.path.isfile(os.path.join(tmp, "b", "resume", "train_state_rank0.pt"))
    assert os.path.isfile(os.path.join(tmp, "b", "resume", "replay_rank0.snap"))
    c = _run(tmp, "c", resume=os.path.join(tmp, "b"), full=0, overlap=overlap)
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


def test_resume_with_graphed_learner_step_keeps_the_device_learning_rate(tmp_path):
    """graph_learner_step + lr_anneal: the captured Adam update reads the learning rate from a device word that set_lr
    refills every step.  A checkpoint load (torch.load(map_location="cpu") + optimizer.load_state_dict) must leave that
    word on the device — otherwise set_lr fills a CPU copy and every replay keeps the value it was captured with."""
    tmp = str(tmp_path)
    a = _run(tmp, "a", full=0, graph=1)
    a2 = _run(tmp, "a2", full=0, graph=1)
    b = _run(tmp, "b", stop=H, full=1, graph=1)
    c = _run(tmp, "c", resume=os.path.join(tmp, "b"), full=0, graph=1)
    assert a["graph_replayed"] and c["graph_replayed"]
    nb = len(b["qloss"])
    assert 10 < nb < len(a["qloss"]) and nb + len(c["qloss"]) == len(a["qloss"])
    # the word the update reads follows the schedule on the device, before and after the resume
    for run in (a, c):
        assert all(on_dev for _, on_dev in run["lr_word"])
        np.testing.assert_allclose([w for w, _ in run["lr_word"]], run["lr"], rtol=1e-6)
    assert c["lr"][0] < 0.75e-3 and c["lr"][-1] < c["lr"][0]          # annealing went on after the resume
    np.testing.assert_allclose(b["lr"] + c["lr"], a["lr"], rtol=1e-12)
    deterministic = a["qloss"] == a2["qloss"] and a["grad_norm"] == a2["grad_norm"]
    for key in ("qloss", "grad_norm"):
        joined = b[key] + c[key]
        if deterministic:
            assert joined == a[key], key
        else:
            np.testing.assert_allclose(joined, a[key], rtol=1e-5, atol=1e-7, err_msg=key)
    if deterministic:
        assert c["param_sum"] == a["param_sum"]


def test_two_rank_job_checkpoints_and_resumes(tmp_path):
    """A 2-rank job (torch.distributed.run, both ranks on this GPU over gloo): every rank
    writes ITS replay shard / optimizer / RNG files into the run directory rank 0 created
    (a rank that does not log still knows the directory), and a 2-rank --resume continues
    the uninterrupted series on both ranks."""
    tmp = str(tmp_path)
    a = _run(tmp, "a", ranks=2, full=0)
    b = _run(tmp, "b", ranks=2, stop=H, full=1)              # strong scaling: step fields are whole-job values (H -> H/2 per rank)
    for r in range(2):
        assert os.path.isfile(os.path.join(tmp, "b", "resume", "train_state_rank%d.pt" % r))
        assert os.path.isfile(os.path.join(tmp, "b", "resume", "replay_rank%d.snap" % r))
    c = _run(tmp, "c", ranks=2, resume=os.path.join(tmp, "b"), full=0)
    for r in range(2):
        nb = len(b[r]["qloss"])
        assert 5 < nb < len(a[r]["qloss"])
        assert nb + len(c[r]["qloss"]) == len(a[r]["qloss"])
        assert c[r]["final_steps"] == a[r]["final_steps"]
        for key in ("qloss", "grad_norm"):
            np.testing.assert_allclose(b[r][key] + c[r][key], a[r][key], rtol=1e-5, atol=1e-7, err_msg="%s rank %d" % (key, r))
    assert a[0]["param_sum"] == a[1]["param_sum"] and c[0]["param_sum"] == c[1]["param_sum"]   # replicas stay identical
