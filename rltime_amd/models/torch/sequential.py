"""SequentialModel (reference rltime/models/torch/{torch_model,sequential}.py):
a list of modules run in order, with an optional extra-input vector concatenated
at one layer, per-layer pre-processors (how IQN injects its quantile layer) and
`make_input_state` to attach each module's state to an observation batch."""
import numpy as np
import torch
import torch.nn as nn

from rltime_amd.general.type_registry import get_registered_type
from rltime_amd.spaces import is_tuple_space
from .utils import make_tensor


class SequentialModel(nn.Module):
    def __init__(self, observation_space, layer_configs, extra_input_layer=None):
        super().__init__()
        # torch_model.py:27-48: Box, or Tuple(main Box, 1-D extra Boxes...)
        if is_tuple_space(observation_space):
            spaces = observation_space.spaces
            assert all(len(s.shape) == 1 for s in spaces[1:]), \
                "only 1D box spaces are supported for the non-main observation"
            self.main_input_shape = tuple(spaces[0].shape)
            self.extra_input_shape = (int(np.sum([s.shape[0] for s in spaces[1:]])),)
        else:
            self.main_input_shape = tuple(observation_space.shape)
            self.extra_input_shape = None

        self.layers = nn.ModuleList()
        self.layer_input_shapes = []
        if self.extra_input_shape is not None:          # sequential.py:37-46
            self.extra_input_layer = extra_input_layer if extra_input_layer is not None \
                else self._auto_extra_layer(layer_configs)
            if self.extra_input_layer < 0:
                self.extra_input_layer += len(layer_configs)
            assert self.extra_input_layer < len(layer_configs)
        else:
            self.extra_input_layer = None

        shape = self.main_input_shape
        for i, cfg in enumerate(layer_configs):
            cls = get_registered_type("modules", cfg["type"])
            if i == self.extra_input_layer:
                shape = (int(np.prod(shape)) + int(np.prod(self.extra_input_shape)),)
            layer = cls(inp_shape=shape, **cfg.get("args", {}))
            self.layers.append(layer)
            self.layer_input_shapes.append(shape)
            shape = layer.out_shape
        assert len(shape) == 1
        self.out_size = shape[0]
        self.layer_pre_processors = {}

    @staticmethod
    def _auto_extra_layer(layer_configs):
        """sequential.py:70-82: first recurrent layer, else the last layer."""
        for i, cfg in enumerate(layer_configs):
            if get_registered_type("modules", cfg["type"]).is_recurrent():
                return i
        return len(layer_configs) - 1

    def _index(self, i):
        if i < 0:
            i += len(self.layers)
        assert 0 <= i < len(self.layers)
        return i

    def set_layer_preprocessor(self, layer_index, preprocessor):
        i = self._index(layer_index)
        assert i not in self.layer_pre_processors, "at most 1 pre-processor per layer"
        self.layer_pre_processors[i] = preprocessor
        return self.layer_input_shapes[i]

    def get_layer_in_shape(self, layer_index):
        return self.layer_input_shapes[self._index(layer_index)]

    def get_layer_out_shape(self, layer_index):
        return self.layers[self._index(layer_index)].out_shape

    def device(self):
        return next(self.parameters()).device

    def is_cuda(self):
        return self.device().type == "cuda"

    def is_recurrent(self):
        return any(layer.is_recurrent() for layer in self.layers)

    def make_input_state(self, x, initials):
        """sequential.py:128-146."""
        state = {"x": x}
        for i, layer in enumerate(self.layers):
            state["layer%d_state" % i] = layer.get_state(initials)
        return state

    def last_recurrent_layer(self):
        idx = [i for i, layer in enumerate(self.layers) if layer.is_recurrent()]
        return idx[-1] if idx else None

    def forward(self, inp, timesteps, stop_after=None, skip_last=False):
        """sequential.py:167-210 -> {"output", "layer_inputs", + pre-processor extras}.
        stop_after=i ends the pass after layer i (used by the burn-in, which only
        needs the recurrent state and not the head).  skip_last=True runs
        everything up to and including the last layer's pre-processor but not the
        layer itself: "output" is then that layer's input (the policy head fuses
        the last layer with its own parallel branch, DQNPolicy.predict)."""
        inp = make_tensor(inp, self.device())
        x = inp["x"]
        # optional: output of layer 0 computed earlier for exactly these rows by
        # THIS model (trainer-level sharing of the conv stack between the training
        # pass and the double-Q selection pass, multi_step_trainer.py)
        shared = inp.get("x_features")
        first_out = shared.get(id(self)) if isinstance(shared, dict) else None
        # optional: the main observation already converted for layer 0 (float32 *
        # scale, NHWC) by CNN.prepare_input — one conversion of the gathered block
        # serves every pass of a learner step
        prepared = inp.get("x_prepared")
        # optional: the input projection of recurrent layer 1 computed earlier for
        # exactly these rows by THIS model (same sharing, one layer further)
        shared_proj = inp.get("x_projected")
        projected = shared_proj.get(id(self)) if isinstance(shared_proj, dict) else None
        extra = None
        if isinstance(x, (tuple, list)):
            x, extra = x[0], torch.cat([v.reshape(v.shape[0], -1) for v in x[1:]], dim=-1)
        assert (extra is None) == (self.extra_input_layer is None)
        result = {"layer_inputs": []}
        for i, layer in enumerate(self.layers):
            if i == self.extra_input_layer:
                if x.shape[0] != extra.shape[0]:     # multi-sample batch (sequential.py:155-159)
                    assert x.shape[0] % extra.shape[0] == 0
                    extra = extra.repeat_interleave(x.shape[0] // extra.shape[0], dim=0)
                x = torch.cat([x.reshape(x.shape[0], -1), extra], dim=-1)
            if i in self.layer_pre_processors:
                x, more = self.layer_pre_processors[i](x)
                result.update(more)
            result["layer_inputs"].append(x)
            if skip_last and i == len(self.layers) - 1:
                break
            if i == 0 and first_out is not None:
                x = first_out
            elif i == 1 and projected is not None and first_out is not None and i != self.extra_input_layer \
                    and i not in self.layer_pre_processors and hasattr(layer, "project_input"):
                x = layer(x, timesteps=timesteps, projected=projected, **inp.get("layer%d_state" % i, {}))
            elif i == 0 and prepared is not None and self.extra_input_layer != 0 and 0 not in self.layer_pre_processors:
                x = layer(prepared, timesteps=timesteps, prepared=True, **inp.get("layer0_state", {}))
            else:
                x = layer(x, timesteps=timesteps, **inp.get("layer%d_state" % i, {}))
            if stop_after is not None and i == stop_after:
                break
        result["output"] = x
        return result
