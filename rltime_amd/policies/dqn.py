"""DQNPolicy with dueling support (reference rltime/policies/torch/dqn.py:9-148)."""
import numpy as np
import torch
import torch.nn.functional as F

from .torch_policy import TorchPolicy
from rltime_amd.models.torch.fused import linear_relu, dueling_tail
from rltime_amd.models.torch.utils import linear


class DQNPolicy(TorchPolicy):
    def __init__(self, action_space, dueling=False, dueling_value_layer_hidden_size=None, **kwargs):
        super().__init__(**kwargs)
        self.num_actions = action_space.n
        self.out_layer = linear(self.model.out_size, action_space.n * self._outputs_per_action())
        if dueling:
            # dqn.py:50-66: value branch in parallel to the LAST model layer
            inner = int(np.prod(self.model.get_layer_in_shape(-1)))
            hidden = dueling_value_layer_hidden_size or int(np.prod(self.model.get_layer_out_shape(-1)))
            self.value_hidden_layer = linear(inner, hidden)
            self.value_layer = linear(hidden, self._outputs_per_action())
        else:
            self.value_layer = None

    def _outputs_per_action(self):
        return 1

    def _shape_action_outputs(self, output):
        return output, 1

    def _process_dueling(self, action_outputs, state_layer):
        """dqn.py:74-87: V + A - mean_a A."""
        state_layer = state_layer.reshape(state_layer.shape[0], -1)
        v = self.value_layer(linear_relu(state_layer, self.value_hidden_layer.weight,
                                         self.value_hidden_layer.bias))
        v, action_dim = self._shape_action_outputs(v)
        return v + action_outputs - action_outputs.mean(action_dim, keepdim=True)

    def _predict_postprocess(self, output, model_output):
        if self.value_layer is not None:
            output = self._process_dueling(output, model_output["layer_inputs"][-1])
        return output

    def _fused_tail_layer(self):
        """The model's last layer when it can be fused with the dueling value branch:
        a single Linear+ReLU block (no batch-norm) on the GPU in fp32."""
        if self.value_layer is None or not getattr(self, "fuse_dueling_tail", True):
            return None
        last = self.model.layers[-1]
        blocks = getattr(last, "layers", None)
        if not getattr(last, "fuse_relu", False) or blocks is None or len(blocks) != 1 or len(blocks[0]) != 1:
            return None
        fc = blocks[0][0]
        if not (fc.weight.is_cuda and fc.weight.dtype == torch.float32) or torch.is_autocast_enabled():
            return None
        if self.value_hidden_layer.in_features != fc.in_features:
            return None
        return fc

    def predict(self, x, timesteps):
        """dqn.py:101-112."""
        fc = self._fused_tail_layer()
        if fc is not None:
            # last FC layer and the dueling value-hidden layer read the same input:
            # one GEMM for both (models/torch/fused.py), then dqn.py:74-87
            res = self.model(x, timesteps, skip_last=True)
            inner = res["output"].reshape(-1, fc.in_features)
            adv, val = dueling_tail(inner, fc, self.out_layer, self.value_hidden_layer, self.value_layer)
            adv, action_dim = self._shape_action_outputs(adv)
            val, _ = self._shape_action_outputs(val)
            output = val + adv - adv.mean(action_dim, keepdim=True)
            return self._tail_postprocess(output, res)
        res = self.model(x, timesteps)
        output, _ = self._shape_action_outputs(self.out_layer(res["output"]))
        return self._predict_postprocess(output, res)

    def _tail_postprocess(self, output, model_output):
        return output

    def predict_selection(self, x, timesteps):
        """Scores whose arg-max over actions equals predict()'s, for the double-Q ACTION SELECTION only (training/torch/
        dqn.py:52-71, iqn.py:36-45).  With a dueling head, Q(a) = V + A(a) - mean_a A (dqn.py:74-87): V and the mean
        are the same for every action of a row — and stay so under IQN's mean over quantile samples — so
        argmax_a Q(a) = argmax_a A(a) and the value stream (a second 512-wide hidden layer over every quantile row)
        need not be evaluated for this pass.  Same forward otherwise (same quantile-fraction draw); results can differ
        from predict()'s arg-max only where two actions tie to the last float32 bit."""
        if self.value_layer is None:
            return self.predict(x, timesteps)
        fc = self._fused_tail_layer()
        if fc is not None:
            res = self.model(x, timesteps, skip_last=True)
            inner = res["output"].reshape(-1, fc.in_features)
            from rltime_amd.models.torch import gemm3
            if not torch.is_grad_enabled() and gemm3.head_supported(inner, fc.weight, fc.bias, self.out_layer.weight):
                # hidden layer + advantage outputs in one launch; the hidden activation never reaches HBM
                _, adv = gemm3.linear_relu_head(inner, fc.weight, fc.bias, self.out_layer.weight, self.out_layer.bias, False)
            else:
                adv = F.linear(linear_relu(inner, fc.weight, fc.bias), self.out_layer.weight, self.out_layer.bias)
        else:
            res = self.model(x, timesteps)
            adv = self.out_layer(res["output"])
        adv, _ = self._shape_action_outputs(adv)
        return self._tail_postprocess(adv, res)

    def _samples_per_state(self):
        return 1

    def actor_head_raw(self, inp, timesteps):
        """The network outputs the acting head starts from, WITHOUT the dueling
        combine / quantile mean / argmax (those run fused in csrc/acting.hip for the
        device-resident actor): (adv (rows, A), val (rows, q) or None, samples per state)."""
        with torch.no_grad():
            fc = self._fused_tail_layer()
            if fc is not None:
                res = self.model(inp, timesteps, skip_last=True)
                inner = res["output"].reshape(-1, fc.in_features)
                adv, val = dueling_tail(inner, fc, self.out_layer, self.value_hidden_layer, self.value_layer)
                return adv, val, self._samples_per_state()
            res = self.model(inp, timesteps)
            adv = self.out_layer(res["output"])
            val = None
            if self.value_layer is not None:
                state_layer = res["layer_inputs"][-1]
                state_layer = state_layer.reshape(state_layer.shape[0], -1)
                val = self.value_layer(linear_relu(state_layer, self.value_hidden_layer.weight,
                                                   self.value_hidden_layer.bias))
            return adv, val, self._samples_per_state()

    def _actor_predict_postprocess(self, pred):
        return pred

    def actor_predict(self, inp, timesteps, for_eval=False, as_numpy=True):
        """dqn.py:132-148.  as_numpy=False keeps actions / qvalues on the device
        (no host sync) for the device-resident ingest path."""
        with torch.no_grad():
            qvalues = self._actor_predict_postprocess(self.predict(inp, timesteps))
        assert qvalues.dim() == 2
        if not as_numpy:
            return {"actions": qvalues.argmax(dim=1), "qvalues": qvalues}
        q = qvalues.cpu().numpy()
        return {"actions": np.argmax(q, axis=1), "qvalues": q}
