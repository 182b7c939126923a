"""Host driver: json config -> actors -> trainer (reference rltime/train.py:26-153).

    python -m rltime_amd.train synthetic_atari_iqn_lstm.json --num-envs 64 \
        --conf-update '{"training": {"args": {"total_steps": 200000}}}'

One process per GPU (the reference has no multi-GPU path):

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        -m rltime_amd.train synthetic_atari_iqn_lstm.json [--scaling strong|weak]

Every rank reads RANK / LOCAL_RANK / WORLD_SIZE, takes its shard of the config
(rltime_amd.parallel.shard_config: envs and replay split by env id; "strong" =
the configured mbatch_size is the global batch), joins the RCCL process group and
trains data-parallel; rank 0 logs.
"""
import argparse
import json
import logging

from rltime_amd.general.config import load_config, validate_config
from rltime_amd.general.loggers import DirectoryLogger, NullLogger
from rltime_amd.general.type_registry import get_registered_type
from rltime_amd.general.utils import deep_dictionary_update


def make_vec_env(env, env_args, num_envs, device, seed=0):
    if isinstance(env, str) and env.startswith("CartPole-"):
        # the CPU plumbing config (BASELINE configs[0]): the build's own cart-pole, v0 = 200-step episodes, v1 = 500
        from rltime_amd.acting.cartpole_env import CartPoleVecEnv
        limit = (env_args or {}).get("max_episode_steps", 500 if env.endswith("v1") else 200)
        return CartPoleVecEnv(num_envs, max_episode_steps=limit, seed=seed)
    if env != "synthetic-atari":
        raise ValueError(
            "rltime_amd ships the synthetic vector env ('synthetic-atari') and its own CartPole ('CartPole-v0/-v1'): "
            "real emulators are CPU code outside the scope of this backend (DESIGN.md)")
    from rltime_amd.acting.synthetic_env import SyntheticAtariVecEnv
    args = dict(env_args or {})
    args["frame_shape"] = tuple(args.get("frame_shape", (4, 84, 84)))
    return SyntheticAtariVecEnv(num_envs, device=device, seed=seed, **args)


def create_actors(config, device="cuda", device_acting=True, use_graph=False):
    """acting/create.py:4-27 for the local synchronous actor."""
    from rltime_amd.acting.actor import Actor
    acting = config.get("acting", {})
    n = acting.get("actor_envs", 1)
    env = make_vec_env(config.get("env"), config.get("env_args"), n, device)
    if device_acting:
        # the device-resident actor needs a GPU policy and an env that steps on the device; a host env / CPU policy
        # (policy_args.cuda false: the cartpole configs) takes the reference's per-env dict path (actor.py:97-149)
        import torch
        cuda = config.get("policy_args", {}).get("cuda", "auto")
        on_gpu = torch.cuda.is_available() if cuda == "auto" else bool(cuda)
        device_acting = on_gpu and hasattr(env, "step_device")
    return Actor(env, exploration_config=acting.get("exploration"), device=device_acting, use_graph=use_graph,
                 base_env_id=acting.get("env_base", 0), total_env_ids=acting.get("total_envs"))


def train(config, logger=None, device="cuda", device_acting=True, data_parallel=None, use_graph=False,
          resume=None, on_trainer=None):
    """train.py:26-63.  `data_parallel`: rltime_amd.parallel.DataParallel when this
    process is one rank of a multi-GPU job (config already sharded)."""
    logger = logger or NullLogger(echo=True)
    logger.log_config(config)
    actors = create_actors(config, device, device_acting, use_graph=use_graph)
    training = config["training"]
    trainer_cls = get_registered_type("trainers", training["type"])
    trainer = trainer_cls(logger=logger, actors=actors, model_config=config["model"],
                          policy_args=config.get("policy_args", {}))
    trainer.data_parallel = data_parallel
    if resume:
        trainer.resume_from = resume
    if on_trainer is not None:
        on_trainer(trainer)
    try:
        trainer.train(**training["args"])
    finally:
        actors.close()
    return trainer


def make_logger(rank, dp, log_dir, log_name=None):
    """Rank 0 writes the log rows; EVERY rank knows the run directory, because full
    checkpoints are one file set per rank (training/resume.py).  create_new()
    uniquifies the name, so rank 0's choice is broadcast."""
    if not log_dir:
        return NullLogger(echo=(rank == 0))
    logger, path = None, None
    if rank == 0:
        logger = DirectoryLogger.create_new(log_dir, log_name)
        path = logger.path
    if dp is not None:
        path = dp.broadcast_object(path)
    return logger if rank == 0 else NullLogger(echo=False, path=path)


def train_from_config(path, num_envs=None, env=None, conf_update=None, log_dir=None, log_name=None,
                      scaling="strong", backend=None, resume=None, seed=None):
    """train.py:66-111, plus the one-process-per-GPU launch."""
    import torch
    from rltime_amd import parallel
    config = load_config(path)
    validate_config(config)
    if env is not None:
        config["env"] = env
    if num_envs is not None:
        config.setdefault("acting", {})["actor_envs"] = num_envs
    if conf_update:
        deep_dictionary_update(config, conf_update)
    rank, world, local, dp = parallel.init_from_env(backend=backend)
    device = "cuda"
    if torch.cuda.is_available():
        torch.cuda.set_device(local % max(torch.cuda.device_count(), 1))
        device = torch.device("cuda", torch.cuda.current_device())
    if dp is not None:
        config = parallel.shard_config(config, rank, world, scaling)
    if seed is not None:
        import random
        import numpy as np
        random.seed(seed + rank); np.random.seed(seed + rank); torch.manual_seed(seed + rank)
    if seed is not None and config["training"]["args"].get("history_mode", {}).get("type") in ("replay", "prioritized_replay"):
        hm = config["training"]["args"]["history_mode"].setdefault("args", {})
        hm.setdefault("seed", seed)          # device-RNG sampling stream (the shard mixes its env_base in)
    logger = make_logger(rank, dp, log_dir, log_name)
    try:
        return train(config, logger, device=device, data_parallel=dp, resume=resume)
    finally:
        if dp is not None:
            import torch.distributed as dist
            if dist.is_initialized():
                dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("config")
    ap.add_argument("--num-envs", type=int)
    ap.add_argument("--env")
    ap.add_argument("--log-dir")
    ap.add_argument("--log-name")
    ap.add_argument("--conf-update", type=json.loads)
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"],
                    help="multi-GPU: strong = mbatch_size / envs / replay size are whole-job values split over "
                         "the ranks; weak = every rank keeps the configured values")
    ap.add_argument("--backend", default=None, help="torch.distributed backend (default nccl = RCCL)")
    ap.add_argument("--resume", default=None, help="directory written by a previous run's checkpoints (true resume)")
    ap.add_argument("--seed", type=int, default=None)
    a = ap.parse_args()
    logging.basicConfig(level=logging.INFO)
    train_from_config(a.config, a.num_envs, a.env, a.conf_update, a.log_dir, a.log_name,
                      scaling=a.scaling, backend=a.backend, resume=a.resume, seed=a.seed)


if __name__ == "__main__":
    main()
