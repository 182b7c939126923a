"""Subprocess body of tests/test_resume_gpu.py: one training run of the tiny
recurrent-IQN / prioritized-replay config through rltime_amd.train.train, the
per-learner-step loss / grad-norm series dumped as JSON."""
import argparse
import json
import os
import random
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CNN = {"type": "cnn", "args": {"channels_last": True, "layers": [{"filters": 8, "kernel": 4, "stride": 2},
                                                                {"filters": 8, "kernel": 3, "stride": 1}]}}


def config(total, stop, full, overlap=False, graph=False):
    cfg = _config(total, stop, full, overlap)
    if graph:
        # the T = 1 learner step replayed from a captured HIP graph (training/torch_trainer.py): feed-forward net, fixed
        # clip, the learning rate a device word that lr_anneal refills every step — and that a resume must keep on the device
        cfg["model"]["args"]["layer_configs"] = [CNN, {"type": "fc", "args": {"fc_size": 32}}]
        t = cfg["training"]["args"]
        t.update(nstep_train=1, burn_in_timesteps=0, rnn_bootstrap=False, clip_grad_dynamic_alpha=None, graph_learner_step=True)
    return cfg


def _config(total, stop, full, overlap=False):
    return {
        "acting": {"actor_envs": 8, "exploration": {"type": "epsilon_greedy", "args": {
            "eps_start": 1.0, "eps_final": 0.05, "exploration_fraction": 0.5}}},
        "env": "synthetic-atari", "env_args": {"frame_shape": [4, 20, 20], "n_actions": 4, "done_prob": 0.02},
        "model": {"type": "sequential", "args": {"layer_configs": [
            CNN, {"type": "lstm", "args": {"num_units": 16}}, {"type": "fc", "args": {"fc_size": 32}}]}},
        "policy_args": {"dueling": True, "embedding_dim": 8, "num_sampling_quantiles": 4},
        "training": {"type": "iqn", "args": {
            "clip_rewards": False, "vf_scale_epsilon": 1e-3, "gamma": 0.99, "mbatch_size": 8, "nstep_train": 8,
            "burn_in_timesteps": 4, "nstep_target": 2, "lr": 1e-3, "lr_anneal": True, "double_q": True,
            "rnn_bootstrap": True, "clip_grad": 10.0, "clip_grad_dynamic_alpha": 0.9, "target_update_freq": 160,
            "total_steps": total, "early_stop_steps": stop, "log_freq": total // 2, "warmup_steps": 0,
            "full_checkpoints": full, "overlap_acting": bool(overlap),
            "history_mode": {"type": "prioritized_replay", "args": {
                "size": 600, "train_frequency": 4, "alpha": 0.9, "beta": 0.6, "beta_anneal": True}}}},
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log-dir", required=True)
    ap.add_argument("--name", required=True)
    ap.add_argument("--total", type=int, required=True)
    ap.add_argument("--stop", type=int, default=None)
    ap.add_argument("--full", type=int, default=1)
    ap.add_argument("--resume", default=None)
    ap.add_argument("--out", required=True)
    ap.add_argument("--overlap", type=int, default=0)
    ap.add_argument("--graph", type=int, default=0)
    a = ap.parse_args()
    torch.backends.cudnn.deterministic = True          # MIOpen: no atomically-accumulating solvers
    random.seed(1); np.random.seed(2); torch.manual_seed(3)   # noqa: E702
    from rltime_amd.general.loggers import DirectoryLogger
    from rltime_amd.train import train
    series = {"qloss": [], "grad_norm": [], "steps_at": [], "lr": [], "lr_word": []}

    def hook(trainer):
        orig = trainer.value_log.log

        def tap(key, value, *args, **kw):
            if key in ("qloss", "grad_norm") and kw.get("group") == "train":
                series[key].append(float(value.item() if hasattr(value, "item") else value))
                if key == "qloss":
                    series["steps_at"].append(trainer.steps)
            if key == "lr" and kw.get("group") == "train":
                # what the trainer set, and the word the (captured) update actually reads
                word = trainer.optimizer.param_groups[0]["lr"]
                series["lr"].append(float(value))
                series["lr_word"].append([float(word), bool(torch.is_tensor(word) and word.is_cuda)])
            return orig(key, value, *args, **kw)
        trainer.value_log.log = tap

    cfg = config(a.total, a.stop, bool(a.full), a.overlap, bool(a.graph))
    dp, rank = None, 0
    if "WORLD_SIZE" in os.environ:
        # one rank of a torch.distributed.run launch: both ranks share GPU 0, so gloo (RCCL refuses
        # two ranks per device); the same entry path as rltime_amd.train.train_from_config
        from rltime_amd import parallel
        from rltime_amd.train import make_logger
        torch.cuda.set_device(0)
        rank, world, _, dp = parallel.init_from_env(backend="gloo")
        cfg = parallel.shard_config(cfg, rank, world, "strong")
        random.seed(1 + rank); np.random.seed(2 + rank); torch.manual_seed(3 + rank)   # noqa: E702
        logger = make_logger(rank, dp, a.log_dir, a.name)
        assert logger.path == os.path.join(a.log_dir, a.name)
    else:
        logger = DirectoryLogger(os.path.join(a.log_dir, a.name), echo=False)
    trainer = train(cfg, logger, resume=a.resume, on_trainer=hook, data_parallel=dp)
    series["final_steps"] = trainer.steps
    gstep = getattr(trainer, "_gstep", None)
    series["graph_replayed"] = bool(gstep is not None and gstep.get("graph") is not None)
    series["param_sum"] = float(sum(p.double().sum().item() for p in trainer.policy.parameters()))
    json.dump(series, open(a.out if dp is None else a.out.replace(".json", "_rank%d.json" % rank), "w"))
    if dp is not None:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
