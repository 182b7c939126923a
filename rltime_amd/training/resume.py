"""True resume.

The reference checkpoints the policy weights only (`_get_train_state` returns
{}, rltime/training/policy_trainer.py:170-185): a restarted run begins with an
empty replay, a fresh optimizer and reseeded RNGs.  Here a *full* checkpoint is
everything THE LOOP's next iteration depends on, taken at an iteration boundary:

  * online + target network weights, Adam moments and step counts;
  * the acted / trained / learner-step counters, learning rate, dynamic
    gradient-clip average, actor-update bookkeeping;
  * the RNG streams the path consumes: Python `random` (prioritized sampling,
    prioritized_replay_history.py:238), NumPy's global stream (uniform sampling
    replay_history.py:118, epsilon pick epsilon_greedy.py:67), torch CPU and
    device generators (IQN taus iqn.py:76, device exploration);
  * the replay shard — rings, priority trees, free list, quota, sampling
    counter — through mirl_replay_save (include/mirl.h), one file per rank;
  * the actor: recurrent carry, last input state, synthetic-env generator.

A run restored from it continues with the loss series the uninterrupted run
produces (tests/test_resume_gpu.py).  Files, per rank r, under <dir>/resume/:
train_state_rank{r}.pt (torch.save) and replay_rank{r}.snap.
"""
import os
import random

import numpy as np
import torch

from rltime_amd.general.utils import deep_apply


def _rank(trainer):
    dp = getattr(trainer, "data_parallel", None)
    return dp.rank if dp is not None else 0


def _dir(base):
    return os.path.join(base, "resume")


def _to_cpu(tree):
    return deep_apply(tree, lambda x: x.detach().cpu() if isinstance(x, torch.Tensor) else x)


def collect(trainer):
    """Everything except the replay shard, as CPU tensors / plain Python."""
    hist = trainer.history_buffer
    state = {
        "policy": _to_cpu(trainer.policy.state_dict()),
        "target": _to_cpu(trainer.target_policy.state_dict()) if trainer.target_policy is not trainer.policy else None,
        "optimizer": _to_cpu(trainer.optimizer.state_dict()),
        "counters": {
            "steps": trainer.steps, "acted": trainer.clock.acted, "trained": trainer.clock.trained,
            "learner_steps": trainer.clock.learner_steps, "lr": trainer.lr,
            "actors_last_update_steps": trainer._actors_last_update_steps,
            "grad_norm_ma": None if getattr(trainer, "_grad_norm_moving_average", None) is None
            else trainer._grad_norm_moving_average.detach().cpu()},
        "rng": {"python": random.getstate(), "numpy": np.random.get_state(), "torch": torch.get_rng_state(),
                "cuda": torch.cuda.get_rng_state() if torch.cuda.is_available() else None},
        "history": {"seed": getattr(hist, "_seed", None), "last_beta": getattr(hist, "last_beta", None),
                    "created": getattr(hist, "_h", None) is not None,
                    "example_state": getattr(hist, "_example_state", None),
                    "num_envs": getattr(hist, "_num_envs", None), "env_base": getattr(hist, "_env_base", None),
                    "policy_f32": getattr(hist, "_policy_f32", 0)},
        "actors": trainer.actors.get_state() if hasattr(trainer.actors, "get_state") else None,
        # overlapped acting (multi_step_trainer._loop_iteration_overlapped): whether the next
        # iteration's feed is already in the replay, and its not yet applied target-sync / log flags
        "overlap": None if getattr(trainer, "_ov", None) is None
        else {"fed": trainer._ov["fed"], "deferred": trainer._ov["deferred"]},
    }
    return state


def save(trainer, base_dir):
    d = _dir(base_dir)
    os.makedirs(d, exist_ok=True)
    r = _rank(trainer)
    torch.cuda.synchronize()
    hist = trainer.history_buffer
    if getattr(hist, "_h", None) is not None:
        tmp = os.path.join(d, "replay_rank%d.snap.tmp" % r)
        hist.save(tmp)
        os.replace(tmp, os.path.join(d, "replay_rank%d.snap" % r))
    tmp = os.path.join(d, "train_state_rank%d.pt.tmp" % r)
    torch.save(collect(trainer), tmp)
    os.replace(tmp, os.path.join(d, "train_state_rank%d.pt" % r))


def available(base_dir, rank=0):
    return os.path.isfile(os.path.join(_dir(base_dir), "train_state_rank%d.pt" % rank))


def load(trainer, base_dir):
    """Restore into a trainer whose policies, optimizer and (empty) history buffer
    were just created by the normal start-up path."""
    d = _dir(base_dir)
    r = _rank(trainer)
    state = torch.load(os.path.join(d, "train_state_rank%d.pt" % r), map_location="cpu", weights_only=False)
    trainer.policy.load_state_dict(state["policy"])
    if state["target"] is not None and trainer.target_policy is not trainer.policy:
        trainer.target_policy.load_state_dict(state["target"])
    # load_state_dict replaces param_groups wholesale, and map_location="cpu" turns a device learning rate (graphed
    # learner step: train_init keeps it in a 0-dim device tensor that the captured update reads) into a CPU tensor that
    # set_lr would then fill without the device ever seeing it: keep the ORIGINAL device words and fill them
    device_lrs = [g["lr"] if (torch.is_tensor(g["lr"]) and g["lr"].is_cuda) else None for g in trainer.optimizer.param_groups]
    trainer.optimizer.load_state_dict(state["optimizer"])
    for g, word in zip(trainer.optimizer.param_groups, device_lrs):
        if word is not None:
            word.fill_(float(g["lr"]))
            g["lr"] = word
    c = state["counters"]
    trainer.steps = c["steps"]
    trainer.clock.acted, trainer.clock.trained, trainer.clock.learner_steps = c["acted"], c["trained"], c["learner_steps"]
    trainer.lr = c["lr"]
    trainer.set_lr(trainer.lr)
    trainer._actors_last_update_steps = c["actors_last_update_steps"]
    if c["grad_norm_ma"] is not None:
        trainer._grad_norm_moving_average = c["grad_norm_ma"].to(trainer.policy.device())
    h = state["history"]
    hist = trainer.history_buffer
    if h["created"]:
        hist.load(os.path.join(d, "replay_rank%d.snap" % r), example_state=h["example_state"],
                  num_envs=h["num_envs"], env_base=h["env_base"], policy_f32=h["policy_f32"])
    if h["seed"] is not None:
        hist._seed = h["seed"]
    if h["last_beta"] is not None:
        hist.last_beta = h["last_beta"]
    if state["actors"] is not None and hasattr(trainer.actors, "set_state"):
        # includes the training progress the actors were last TOLD about (their epsilon
        # schedule runs on it; update_actors refreshes it only every
        # actor_update_frequency_steps, multi_step_trainer.py:369-373)
        trainer.actors.set_state(state["actors"])
    ov = getattr(trainer, "_ov", None)
    if ov is not None:
        # the actors' weight copy was snapshotted from the freshly initialised policy before
        # this load; at a checkpoint (end of an iteration) it equals the online weights
        ov["actor_policy"].copy_from(trainer.policy)
        ov["weights_ready"].record()
        if state.get("overlap") is not None:
            ov["fed"], ov["deferred"] = state["overlap"]["fed"], state["overlap"]["deferred"]
    # RNG streams last: nothing above may consume them afterwards
    rng = state["rng"]
    random.setstate(rng["python"])
    np.random.set_state(rng["numpy"])
    torch.set_rng_state(rng["torch"])
    if rng["cuda"] is not None and torch.cuda.is_available():
        torch.cuda.set_rng_state(rng["cuda"])
    return state
