"""Oracle: the recurrent dueling-IQN network and one learner evaluation in float64
(TEST INFRASTRUCTURE ONLY — nothing under rltime_amd/ imports this).

Plain-torch float64 restatement, from a float32 state_dict, of

  * rltime/models/torch/modules/cnn.py:43-50       uint8 -> float * scale, conv + ReLU stack
  * rltime/models/torch/modules/lstm.py:60-116     LSTMCell time loop, state reset on `initials`,
                                                   stored state consumed from timestep 0 only
  * rltime/policies/torch/iqn.py:67-106            cos(pi i tau) embedding -> Linear + ReLU ->
                                                   product with the interleaved repeat of the features,
                                                   injected before the LAST model layer
  * rltime/models/torch/modules/fc.py:33-35        Linear + ReLU
  * rltime/policies/torch/dqn.py:50-112            advantage outputs; dueling value branch off the
                                                   INPUT of the last model layer; V + A - mean_a A
  * rltime/training/multi_step_trainer.py:90-131   burn-in: prefix forward, the state stored at row P
                                                   replaced by the fresh one (zeroed where initials[P])
  * rltime/training/torch/iqn.py:15-52,54-129 and torch_trainer.py:101-147 through oracle.qmath

It is the independent anchor of tests/test_network_ab_gpu.py: the HIP path (products on the bf16 matrix
pipe, persistent LSTM sweeps) and the library path are both measured against THIS evaluation of the same
weights on the same gathered batch.  PINNED: tests/golden/generate.py run_network64_pin runs the reference's own
SequentialModel + IQNPolicy (float32, CPU) and Net64 on the same seeded weights, frames, stored state, initials and
quantile fractions and asserts outputs within 1e-5 and gradients within 1e-4 (measured 1.4e-6); the reference's
outputs travel as tests/golden/network64_pin.npz and tests/test_oracle_golden.py repeats the check without it.
"""
import math

import torch
import torch.nn.functional as F

from . import qmath


def _d(t):
    return t.detach().double()


def _conv64(x, w, b, stride):
    """Valid-padding conv2d as im2col + matmul (no library convolution involved: float64 convolutions are not
    something MIOpen offers, and the anchor should not share a kernel with either path under test)."""
    n, _, h, wd = x.shape
    f, _, k, _ = w.shape
    cols = F.unfold(x, k, stride=stride)                       # (n, c*k*k, positions)
    y = w.reshape(f, -1) @ cols + b.view(1, f, 1)
    return y.reshape(n, f, (h - k) // stride + 1, (wd - k) // stride + 1)


class Net64:
    """state_dict of rltime_amd.policies.iqn.IQNPolicy (same parameter names as the reference's
    policies/torch/iqn.py + models/torch/sequential.py) -> float64 leaves with requires_grad."""

    def __init__(self, state_dict, conv_strides, n_quantiles, device):
        self.p = {k: _d(v).to(device).requires_grad_(v.dtype.is_floating_point and "embedding_range" not in k)
                  for k, v in state_dict.items()}
        self.strides = list(conv_strides)
        self.N = n_quantiles
        self.dev = device

    def params(self):
        return {k: v for k, v in self.p.items() if v.requires_grad}

    def features(self, frames_u8, scale=1.0 / 255.0):
        """cnn.py:43-50 on (rows, C, H, W) uint8 -> (rows, C'*H'*W') in the reference's (C, H, W) flatten."""
        x = frames_u8.to(self.dev).double() * scale
        i = 0
        while "model.layers.0.layers.%d.weight" % i in self.p:
            x = F.relu(_conv64(x, self.p["model.layers.0.layers.%d.weight" % i],
                               self.p["model.layers.0.layers.%d.bias" % i], self.strides[i]))
            i += 1
        return x.reshape(x.shape[0], -1)

    def lstm(self, feats, h0, c0, initials, T):
        """lstm.py:60-116: feats (T*B, F) time-major, h0 / c0 (B, H) = the state stored with timestep 0,
        initials (T, B).  -> (out (T*B, H), (h_T, c_T))."""
        p = self.p
        w_ih, w_hh = p["model.layers.1.lstm_cell.weight_ih"], p["model.layers.1.lstm_cell.weight_hh"]
        b_ih, b_hh = p["model.layers.1.lstm_cell.bias_ih"], p["model.layers.1.lstm_cell.bias_hh"]
        B = feats.shape[0] // T
        x = feats.reshape(T, B, -1)
        keep = (1.0 - initials.to(self.dev).double()).reshape(T, B, 1)
        h, c = h0.to(self.dev).double(), c0.to(self.dev).double()
        out = []
        for t in range(T):
            h, c = h * keep[t], c * keep[t]                      # lstm.py:95-97
            g = x[t] @ w_ih.t() + b_ih + h @ w_hh.t() + b_hh     # torch.nn.LSTMCell
            i, f, gg, o = g.chunk(4, dim=1)
            c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
            h = torch.sigmoid(o) * torch.tanh(c)
            out.append(h)
        return torch.cat(out), (h, c)

    def head(self, x, taus):
        """iqn.py:67-106 + fc.py:33-35 + dqn.py:50-112: x (M, H), taus (M*N,) -> Z (M, N, A)."""
        p, N = self.p, self.N
        rng = p["embedding_range"].double()
        phi = torch.cos(taus.to(self.dev).double().unsqueeze(1) * rng.unsqueeze(0) * math.pi)   # iqn.py:78-81
        emb = F.relu(phi @ p["quantile_layer.weight"].t() + p["quantile_layer.bias"])
        inner = (x.unsqueeze(1) * emb.reshape(x.shape[0], N, -1)).reshape(x.shape[0] * N, -1)    # iqn.py:84,102
        pre_h = inner @ p["model.layers.2.layers.0.0.weight"].t() + p["model.layers.2.layers.0.0.bias"]
        h = F.relu(pre_h)
        adv = h @ p["out_layer.weight"].t() + p["out_layer.bias"]
        pre_v = inner @ p["value_hidden_layer.weight"].t() + p["value_hidden_layer.bias"]
        hv = F.relu(pre_v)                                                                       # dqn.py:50-66
        # ReLU units whose pre-activation is within float32 rounding of zero: a float32 evaluation may take the other
        # branch there, which switches that unit's whole gradient row (tests/test_network_ab_gpu.py conditions its
        # per-parameter gradient bar on this count)
        with torch.no_grad():
            self.relu_near_zero = sum(int((t.abs() <= 4e-7 * t.abs().max()).sum()) for t in (pre_h, pre_v))
        val = hv @ p["value_layer.weight"].t() + p["value_layer.bias"]
        z = val + adv - adv.mean(1, keepdim=True)                                                # dqn.py:74-87
        return z.reshape(x.shape[0], N, -1)

    def predict(self, frames, h0, c0, initials, taus, T):
        """frames (T, B, C, H, W) uint8 -> Z (T*B, N, A)."""
        feats = self.features(frames.reshape((-1,) + tuple(frames.shape[2:])))
        out, _ = self.lstm(feats, h0, c0, initials, T)
        return self.head(out, taus)

    def burn_in_state(self, frames, h0, c0, initials, P):
        """multi_step_trainer.py:90-131 (head-less: only the recurrent state is consumed): the state to store
        at row P = the LSTM state after the P prefix steps, zeroed where initials[P] is set (lstm.py:142-153)."""
        with torch.no_grad():
            feats = self.features(frames[:P].reshape((-1,) + tuple(frames.shape[2:])))
            _, (h, c) = self.lstm(feats, h0, c0, initials[:P], P)
            keep = (1.0 - initials[P].to(self.dev).double()).unsqueeze(-1)
            return h * keep, c * keep


def learner_eval(online, target, batch, taus, gamma, P, kappa=1.0, double_q=True):
    """One learner evaluation (multi_step_trainer.py:278-353 up to the gradients) in float64.

    batch: the (L, b, ...) time-major tensors of History._make_train_batch (history.py:203-286):
      x, tx (L, b, C, H, W) uint8 states / target_states frames; hx, cx, thx, tcx (L, b, H); initials,
      tinitials (L, b); returns, nsteps, masks, actions, weights (L, b).
    taus: [target pass, selection pass, training pass], each (T*b*N,), in the reference's draw order
      (training/torch/iqn.py:18,32,70).
    -> dict(targets (T*b, N'), loss, report (T*b,), grads {name: tensor})."""
    L = batch["x"].shape[0]
    T = L - P
    h, c = online.burn_in_state(batch["x"], batch["hx"][0], batch["cx"][0], batch["initials"], P)
    th, tc = target.burn_in_state(batch["tx"], batch["thx"][0], batch["tcx"][0], batch["tinitials"], P)

    def flat(t):
        t = t[P:]
        return t.reshape((t.shape[0] * t.shape[1],) + tuple(t.shape[2:])).to(online.dev).double()

    with torch.no_grad():
        z_t = target.predict(batch["tx"][P:], th, tc, batch["tinitials"][P:], taus[0], T)
        # the selection pass reads the SAME target_states tree, i.e. the state the TARGET net's burn-in stored at
        # row P (multi_step_trainer.py:117-125 writes it into target_states; iqn.py:32 then feeds that tree to the
        # online net)
        sel = online if double_q else target
        z_s = sel.predict(batch["tx"][P:], th, tc, batch["tinitials"][P:], taus[1], T)
        y = qmath.nstep_target(qmath.iqn_bootstrap(z_t, z_s), flat(batch["returns"]), flat(batch["masks"]),
                               flat(batch["nsteps"]), gamma, None)
    z = online.predict(batch["x"][P:], h, c, batch["initials"][P:], taus[2], T)
    loss, report = qmath.iqn_loss(z, taus[2].to(online.dev).double(), flat(batch["actions"]).long(), y,
                                  flat(batch["weights"]), kappa, T, "mean", None)
    names = list(online.params())
    grads = torch.autograd.grad(loss, [online.p[k] for k in names])
    return {"targets": y, "loss": loss.detach(), "report": report, "grads": dict(zip(names, grads)),
            "relu_near_zero": online.relu_near_zero}       # of the training pass (the last head evaluated on `online`)
