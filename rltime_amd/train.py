"""Host driver: json config -> actors -> trainer (reference rltime/train.py:26-153).

    python -m rltime_amd.train synthetic_atari_iqn_lstm.json --num-envs 64 \
        --conf-update '{"training": {"args": {"total_steps": 200000}}}'
"""
import argparse
import json
import logging

from rltime_amd.general.config import load_config, validate_config
from rltime_amd.general.loggers import DirectoryLogger, NullLogger
from rltime_amd.general.type_registry import get_registered_type
from rltime_amd.general.utils import deep_dictionary_update


def make_vec_env(env, env_args, num_envs, device, seed=0):
    if env != "synthetic-atari":
        raise ValueError(
            "rltime_amd ships only the synthetic vector env ('synthetic-atari'): real "
            "emulators are CPU code outside the scope of this backend (DESIGN.md)")
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
    return Actor(env, exploration_config=acting.get("exploration"), device=device_acting, use_graph=use_graph,
                 base_env_id=acting.get("env_base", 0), total_env_ids=acting.get("total_envs"))


def train(config, logger=None, device="cuda", device_acting=True):
    """train.py:26-63."""
    logger = logger or NullLogger(echo=True)
    logger.log_config(config)
    actors = create_actors(config, device, device_acting)
    training = config["training"]
    trainer_cls = get_registered_type("trainers", training["type"])
    trainer = trainer_cls(logger=logger, actors=actors, model_config=config["model"],
                          policy_args=config.get("policy_args", {}))
    try:
        trainer.train(**training["args"])
    finally:
        actors.close()
    return trainer


def train_from_config(path, num_envs=None, env=None, conf_update=None, log_dir=None, log_name=None):
    """train.py:66-111."""
    config = load_config(path)
    validate_config(config)
    if env is not None:
        config["env"] = env
    if num_envs is not None:
        config.setdefault("acting", {})["actor_envs"] = num_envs
    if conf_update:
        deep_dictionary_update(config, conf_update)
    logger = DirectoryLogger.create_new(log_dir, log_name) if log_dir else NullLogger(echo=True)
    return train(config, logger)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("config")
    ap.add_argument("--num-envs", type=int)
    ap.add_argument("--env")
    ap.add_argument("--log-dir")
    ap.add_argument("--log-name")
    ap.add_argument("--conf-update", type=json.loads)
    a = ap.parse_args()
    logging.basicConfig(level=logging.INFO)
    train_from_config(a.config, a.num_envs, a.env, a.conf_update, a.log_dir, a.log_name)


if __name__ == "__main__":
    main()
