"""TorchPolicy (reference rltime/policies/torch/torch_policy.py:34-111)."""
import io

import torch
import torch.nn as nn

from .policy import Policy
from rltime_amd.models.torch.utils import make_tensor


class TorchPolicy(nn.Module, Policy):
    def __init__(self, model_config, observation_space):
        super().__init__()
        self.model = self._create_model_from_config(model_config, observation_space)

    @classmethod
    def create(cls, *args, cuda="auto", **kwargs):
        """torch_policy.py:43-59: cuda = "auto" | bool | device string."""
        policy = cls(*args, **kwargs)
        if cuda == "auto":
            cuda = torch.cuda.is_available()
        if cuda:
            policy = policy.to(torch.device("cuda" if cuda is True else cuda))
        return policy

    def device(self):
        return next(self.parameters()).device

    def is_cuda(self):
        return self.device().type == "cuda"

    def copy_from(self, source, factor=1.0):
        """torch_policy.py:61-68: parameters only (buffers are constants)."""
        with torch.no_grad():
            src = list(source.parameters())
            dst = list(self.parameters())
            if factor == 1.0:
                torch._foreach_copy_(dst, src)
            else:
                for s, d in zip(src, dst):
                    d.copy_(s * factor + d * (1.0 - factor))

    def get_grad_norm(self):
        """torch_policy.py:70-78: global L2 norm — one fused reduction and one
        host sync instead of one `.item()` per parameter."""
        grads = [p.grad for p in self.parameters() if p.grad is not None]
        norms = torch._foreach_norm(grads, 2)
        return torch.linalg.vector_norm(torch.stack(norms), 2).item()

    def is_recurrent(self):
        return self.model.is_recurrent()

    def make_input_state(self, inp, initials):
        return self.model.make_input_state(inp, initials)

    def make_tensor(self, x, non_blocking=False):
        return make_tensor(x, self.device(), non_blocking)

    def get_creator(self, cuda=False):
        """torch_policy.py:93-97: a callable that builds a CPU copy of this policy
        (what the reference ships to actor processes; `cuda` is unsupported there too)."""
        f = io.BytesIO()
        torch.save(self, f)
        data = f.getvalue()
        return lambda: torch.load(io.BytesIO(data), map_location="cpu", weights_only=False)

    def get_state(self):
        f = io.BytesIO()
        torch.save(self.state_dict(), f)
        return f.getvalue()

    def load_state(self, state):
        self.load_state_dict(torch.load(io.BytesIO(state), map_location=self.device()))

    def get_state_store(self, is_async_storage):
        """The reference returns a StateStore that stacks on the host and
        uploads (general/backend.py:63-153).  Here the replay itself is the
        device-resident store, so there is nothing to hand out."""
        return None
