"""Policy interface (reference rltime/policies/policy.py:4-71)."""
from rltime_amd.general.type_registry import get_registered_type


class Policy:
    def actor_predict(self, state, timesteps=1):
        raise NotImplementedError

    def get_creator(self, cuda=False):
        raise NotImplementedError

    def get_state(self):
        raise NotImplementedError

    def load_state(self, state):
        raise NotImplementedError

    def is_recurrent(self):
        raise NotImplementedError

    def make_input_state(self, inp, initials):
        raise NotImplementedError

    def _create_model_from_config(self, config, observation_space):
        """policy.py:56-61: {"type", "args"} through the "models" registry."""
        cls = get_registered_type("models", config.get("type"))
        return cls(observation_space=observation_space, **config.get("args"))

    def get_state_store(self, is_async_storage):
        return None
