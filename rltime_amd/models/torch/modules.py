"""Model modules: CNN, LSTM, FC (reference rltime/models/torch/modules/
{base,cnn,lstm,fc}.py).  Same constructor arguments and forward contracts; the
dense contractions stay PyTorch-ROCm (MIOpen / rocBLAS / hipBLASLt -> MFMA).

MI355X-first differences:
  * LSTM keeps its running acting state on the device (the reference
    round-trips it through numpy every acting step, lstm.py:145-147);
  * the LSTM sequence forward hoists the input projection of all timesteps into
    one GEMM and keeps only the recurrent GEMM in the time loop.
"""
import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .fused import conv_bias_relu, conv_u8_bias_relu, conv_u8_supported, linear_relu
from .fused import frames_to_f32_nhwc as _frames_to_f32_nhwc
from .utils import conv2d, conv_out_size, init_weight, linear


def _lead_strides(lead, inner):
    out, acc = [], inner
    for n in reversed(lead):
        out.append(acc)
        acc *= n
    return tuple(reversed(out))


class BaseModule(nn.Module):
    """modules/base.py:4-13."""

    def get_state(self, initials):
        return {}

    @staticmethod
    def is_recurrent():
        return False


class CNN(BaseModule):
    """modules/cnn.py:9-53: conv+ReLU stack on channel-first input; uint8 input
    is converted and scaled (1/255) on the device."""

    def __init__(self, inp_shape, layers, scale=1.0 / 255.0, channels_last=False, direct_u8=None):
        super().__init__()
        self.scale = scale
        self.channels_last = channels_last
        # input layer straight from uint8 frames (csrc/conv_in.hip) when its shape is covered
        self.direct_u8 = (os.environ.get("MIRL_CONV1_DIRECT", "1") != "0") if direct_u8 is None else bool(direct_u8)
        self.layers = nn.ModuleList()
        ch = inp_shape[0]
        h, w = inp_shape[1:]
        for spec in layers:
            f, k, s = spec["filters"], spec["kernel"], spec["stride"]
            self.layers.append(conv2d(ch, f, k, s))
            h, w = conv_out_size(h, k, s), conv_out_size(w, k, s)
            ch = f
        self.out_shape = (ch, h, w)
        if channels_last:
            self.to(memory_format=torch.channels_last)

    def prepare_input(self, x):
        """The input conversion of forward() on its own (uint8 NCHW [..., C, H, W] ->
        float32 * scale in NHWC memory, csrc/convert.hip), for callers that feed
        several passes from one gathered block and want to convert it ONCE
        (MultiStepTrainer._prepare_frames).  Returns None when the fused path does
        not apply; the result is handed back through forward(prepared=True)."""
        if not (self.channels_last and self.scale and isinstance(x, torch.Tensor) and x.is_cuda
                and x.dtype == torch.uint8 and x.dim() >= 4 and x.is_contiguous()
                and not torch.is_autocast_enabled()):
            return None
        if self._takes_u8(x.reshape((-1,) + tuple(x.shape[-3:]))):
            return None                       # forward() reads the uint8 rows themselves: nothing to prepare
        lead = tuple(x.shape[:-3])
        flat = _frames_to_f32_nhwc(x.reshape((-1,) + tuple(x.shape[-3:])), self.scale)
        # logical [..., C, H, W] view of the NHWC buffer (each frame is channels_last)
        c, h, w = x.shape[-3:]
        return flat.as_strided(lead + (c, h, w), _lead_strides(lead, c * h * w) + (1, w * c, c))

    def _takes_u8(self, x):
        return bool(self.direct_u8 and self.channels_last and self.scale and len(self.layers)
                    and conv_u8_supported(x, self.layers[0]))

    def forward(self, x, prepared=False, **kwargs):
        layers = self.layers
        if not prepared and self._takes_u8(x):
            x = conv_u8_bias_relu(x, layers[0], self.scale)          # conversion + conv + bias + ReLU, one kernel
            layers = layers[1:]
        elif prepared:
            # already float32 * scale; NHWC memory unless an index op re-laid it out
            if self.channels_last and not x.is_contiguous(memory_format=torch.channels_last):
                x = x.contiguous(memory_format=torch.channels_last)
        elif self.channels_last and self.scale and x.is_cuda and x.dtype == torch.uint8 \
                and x.dim() == 4 and x.is_contiguous() and not torch.is_autocast_enabled():
            x = _frames_to_f32_nhwc(x, self.scale)       # one fused HIP pass (csrc/convert.hip)
        elif self.channels_last:
            x = x.contiguous(memory_format=torch.channels_last)
            if self.scale:
                x = x * self.scale if x.dtype == torch.uint8 else x.float() * self.scale
        elif self.scale:
            # uint8 * python float promotes to float32 in ONE pass (same values as
            # x.float() * scale, cnn.py:44-45)
            x = x * self.scale if x.dtype == torch.uint8 else x.float() * self.scale
        for layer in layers:
            x = conv_bias_relu(x, layer) if self.channels_last else F.relu(layer(x))
        return x


class FC(BaseModule):
    """modules/fc.py:7-41."""

    def __init__(self, inp_shape, fc_size, fc_count=1, batch_norm=False, activation="relu"):
        super().__init__()
        self.flat_size = int(np.prod(inp_shape))
        self.layers = nn.ModuleList()
        sz = self.flat_size
        for _ in range(fc_count):
            block = nn.ModuleList([linear(sz, fc_size)])
            sz = fc_size
            if batch_norm:
                block.append(nn.BatchNorm1d(sz))
            self.layers.append(block)
        self.out_shape = (sz,)
        self.activation = getattr(F, activation)
        self.fuse_relu = activation == "relu"

    def forward(self, x, **kwargs):
        x = x.reshape(-1, self.flat_size)
        for block in self.layers:
            if self.fuse_relu and len(block) == 1:
                x = linear_relu(x, block[0].weight, block[0].bias)   # ReLU in the GEMM epilogue on the GPU
                continue
            for sub in block:
                x = sub(x)
            x = self.activation(x)
        return x


class LSTM(BaseModule):
    """modules/lstm.py:8-161: LSTMCell time loop with per-step state reset on
    `initials`, stored state taken from timestep 0 only, multi-sample (IQN before
    the LSTM) repeat/merge modes."""

    def __init__(self, inp_shape, num_units, multi_sample_merge_mode="inner"):
        super().__init__()
        self.inp_size = int(np.prod(inp_shape))
        self.num_units = num_units
        assert multi_sample_merge_mode in ("outer", "inner")
        self.multi_sample_merge_mode = multi_sample_merge_mode
        self.lstm_cell = nn.LSTMCell(input_size=self.inp_size, hidden_size=num_units)
        self.out_shape = (num_units,)
        self.last_state = None
        self.fused = True          # use the fused HIP sequence op on the GPU
        # NHWC conv output consumed in memory order with permuted W_ih columns instead of a transposing copy (_flat_input)
        self.nhwc_input = os.environ.get("MIRL_LSTM_NHWC_INPUT", "1") != "0"
        self._w_ih_nhwc = None     # (key, permuted W_ih) of the last no-grad pass
        init_weight(self.lstm_cell.weight_hh)
        init_weight(self.lstm_cell.weight_ih)

    def _flat_input(self, x):
        """The layer's input rows and the input weights to multiply them with.  The reference flattens the conv stack's
        (C, H, W) output (lstm.py:60-66: x.view(-1, inp_size)); for an NHWC (channels_last) activation that is a
        transposing copy of the whole block (62 464 x 3136 floats = 0.8 GB per pass at config D).  The rows are taken in
        MEMORY order instead — a view — and the columns of W_ih are permuted to match (a 26 MB copy that autograd undoes
        for the weight gradient): the same products, summed in another column order."""
        w = self.lstm_cell.weight_ih
        if (x.dim() == 4 and x.is_cuda and self.nhwc_input and not x.is_contiguous()
                and x.is_contiguous(memory_format=torch.channels_last) and x.shape[1] * x.shape[2] * x.shape[3] == self.inp_size):
            n, c, h, wd = x.shape
            rows = x.permute(0, 2, 3, 1).reshape(n, h * wd * c)
            if torch.is_grad_enabled() and w.requires_grad:
                # the pass the weight gradient flows through: autograd undoes the permutation (once per learner step)
                return rows, w.view(-1, c, h, wd).permute(0, 2, 3, 1).reshape(-1, h * wd * c)
            # no-grad passes (target net, double-Q selection, burn-in, acting): the permuted copy is a pure function of
            # the weights — kept per (storage address, version counter), rebuilt after an optimizer step / weight copy
            key = (w.data_ptr(), w._version, c, h, wd)
            from . import gemm3
            if self._w_ih_nhwc is None or self._w_ih_nhwc[0] != key or gemm3.REFRESH_ALWAYS:
                with torch.no_grad():
                    src = w.detach().view(-1, c, h, wd).permute(0, 2, 3, 1)
                    if self._w_ih_nhwc is not None and self._w_ih_nhwc[1].shape == (w.shape[0], h * wd * c):
                        self._w_ih_nhwc[1].view(-1, h, wd, c).copy_(src)          # in place: captured graphs keep reading this buffer
                        self._w_ih_nhwc = (key, self._w_ih_nhwc[1])
                    else:
                        self._w_ih_nhwc = (key, src.reshape(-1, h * wd * c).contiguous())
            return rows, self._w_ih_nhwc[1]
        return x.reshape(-1, self.inp_size), w

    def project_input(self, x):
        """x W_ih^T + b_ih + b_hh for every row of x: the part of the sequence forward
        that does not depend on the recurrent state.  A caller that runs this layer
        over overlapping row ranges of one block (the online net's training pass and
        its double-Q selection pass, MultiStepTrainer._share_online_features) computes
        it once and hands each pass its rows through forward(projected=...)."""
        cell = self.lstm_cell
        from .gemm3 import linear as linear3
        rows, w_ih = self._flat_input(x)
        return linear3(rows, w_ih, cell.bias_ih + cell.bias_hh)

    def forward(self, x, hx, cx, initials, timesteps, projected=None):
        if projected is not None and projected.is_cuda and self.fused and projected.shape[0] == hx.shape[0]:
            from .lstm_seq import lstm_sequence_from_projection
            batch = projected.shape[0] // timesteps
            h0 = hx.reshape(timesteps, batch, self.num_units)[0]
            c0 = cx.reshape(timesteps, batch, self.num_units)[0]
            out, h_last, c_last = lstm_sequence_from_projection(
                projected.reshape(timesteps, batch, -1), self.lstm_cell.weight_hh, h0, c0,
                (1 - initials).reshape(timesteps, batch))
            self.last_state = (h_last.detach(), c_last.detach())
            return out.reshape(timesteps * batch, self.num_units)
        x, w_ih = self._flat_input(x)
        assert hx.shape[1] == self.num_units and cx.shape[1] == self.num_units
        assert x.shape[0] % hx.shape[0] == 0
        multi = x.shape[0] // hx.shape[0]
        batch = x.shape[0] // timesteps
        # lstm.py:67-70: only the state stored with timestep 0 is consumed
        hx = hx.reshape(timesteps, batch // multi, self.num_units)[0]
        cx = cx.reshape(timesteps, batch // multi, self.num_units)[0]
        assert initials.shape == ((batch * timesteps) // multi,)
        if multi > 1:
            initials = initials.repeat_interleave(multi, dim=0)
        cell = self.lstm_cell
        if x.is_cuda and multi == 1 and self.fused:
            # MI355X path: one GEMM + one fused HIP kernel per step (lstm_seq.py)
            from .lstm_seq import lstm_sequence
            out, h_last, c_last = lstm_sequence(x, w_ih, cell.weight_hh, cell.bias_ih + cell.bias_hh,
                                                hx, cx, (1 - initials).reshape(timesteps, batch))
            self.last_state = (h_last.detach(), c_last.detach())
            return out.reshape(timesteps * batch, self.num_units)
        keep = (1 - initials).reshape(timesteps, batch, 1)
        # one GEMM for the input projection of every timestep
        gx = F.linear(x, w_ih, cell.bias_ih).reshape(timesteps, batch, -1)
        out = []
        inner = self.multi_sample_merge_mode == "inner"
        for t in range(timesteps):
            if multi > 1 and (t == 0 or inner):
                hx = hx.repeat_interleave(multi, dim=0)
                cx = cx.repeat_interleave(multi, dim=0)
            hx = hx * keep[t]                     # lstm.py:95-97
            cx = cx * keep[t]
            gates = gx[t] + F.linear(hx, cell.weight_hh, cell.bias_hh)
            i, f, g, o = gates.chunk(4, dim=1)    # torch.nn.LSTMCell gate order
            cx = torch.sigmoid(f) * cx + torch.sigmoid(i) * torch.tanh(g)
            hx = torch.sigmoid(o) * torch.tanh(cx)
            out.append(hx)
            if multi > 1 and (t == timesteps - 1 or inner):
                hx = hx.reshape(-1, multi, hx.shape[-1]).mean(1)
                cx = cx.reshape(-1, multi, cx.shape[-1]).mean(1)
        self.last_state = (hx.detach(), cx.detach())      # lstm.py:120
        return torch.cat(out)

    @staticmethod
    def is_recurrent():
        return True

    def get_state(self, initials):
        """lstm.py:131-161: the state to store with the NEXT input — the last
        output state, zeroed where a new episode starts, plus `initials`.
        Stays on the model's device."""
        dev = self.lstm_cell.weight_hh.device
        if not isinstance(initials, torch.Tensor):
            initials = torch.as_tensor(np.asarray(initials, dtype=np.float32))
        initials = initials.to(dev, torch.float32)
        n = initials.shape[0]
        if self.last_state is None:
            assert bool(torch.all(initials != 0)), "first call must be all-initial states"
            z = torch.zeros((n, self.num_units), device=dev)
            self.last_state = (z, z.clone())
        mask = (1 - initials).unsqueeze(-1)
        hx, cx = self.last_state
        self.last_state = (hx * mask, cx * mask)
        assert self.last_state[0].shape[0] == n
        return {"hx": self.last_state[0], "cx": self.last_state[1], "initials": initials}


def get_types():
    return {"cnn": CNN, "fc": FC, "lstm": LSTM}
