"""The device-resident actor's vector step for the CNN -> [LSTM ->] FC (DQN / IQN, dueling or not)
policies, with the launch count cut to what the network itself needs.

Reference order of one vector step (rltime/acting/actor.py:108-147): policy forward on
the last input state -> epsilon-greedy -> env.step -> make_input_state (recurrent state
reset on `done`) -> split into per-env sample dicts -> History.update.  The generic
device path (actor.GraphedStep) replays that as ~48 kernels + 6 for the ingest; most of
them are 2-5 us PyTorch glue (mask multiplies, cat, clones of graph outputs, bias adds,
weight concatenations, rand / randint, fills).  Here one step is

    env.step                                   (on the device; the synthetic env: no launch)
    mirl_actor_pre            1 launch         reset masks, [h|c] pack, initials, reward clip,
                                               episode statistics, RNG step
    mirl_replay_ingest        1 copy + 1 launch  frames + state + scalars + plan + tree, straight
                                               from the static buffers (no clones)
    mirl_conv1_u8_fwd         2 launches       input layer from the env's uint8 frames
    network                   6 launches       csrc/actnet.hip: conv 2, conv 3 (bias + ReLU in the epilogue, layer 3
                                               writes the LSTM product's input rows), [features | h] x [W_ih | W_hh]^T
                                               with the cell in the same launch, quantile embedding x features, hidden
                                               layers -> output shares, head with in-kernel Philox draws.  (Shapes those kernels do not cover keep the round-3 graph
                                               of library calls, ~15 launches: MIRL_ACT_FUSED=0 forces it.)

Everything that only depends on the weights (b_ih + b_hh, [W_ih | W_hh], the joint
[last FC | dueling value-hidden] weights) is rebuilt once per get_samples call
(`refresh`) into static buffers the graph reads, instead of inside every step.

A whole get_samples call as ONE graph launch (`rollout`): when the env can step into static
buffers without host state (env.step_into: the synthetic env's one-kernel step) and the replay
can plan its host bookkeeping ahead (History.plan_ingest: everything History.update decides on
the host is data-independent), the `iters` vector steps of a call — env step, pre-step kernel,
ingest, input layer, network, head — are captured into a single HIP graph.  At 32 envs per rank
(one rank of the 8-GPU job) the per-step host work (~320 us of launches against ~100 us of
kernels) was what bounded the acting; the rollout graph leaves one plan upload and one graph
launch per learner step.
"""
import os
import ctypes as C

import numpy as np
import torch

from rltime_amd._lib import lib, check
from rltime_amd.general.utils import deep_apply, quiet_gc
from rltime_amd.models.torch.fused import conv_bias_relu, conv_u8_supported, cos_embed
from rltime_amd.models.torch import gemm3


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(None)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class IngestedSamples:
    """What get_samples returns when the steps went straight into the replay: only the
    count and the episode statistics are left for the trainer."""
    ingested = True

    def __init__(self, count, tracker):
        self.count = count
        self.episode_tracker = tracker

    def __len__(self):
        return self.count

    def __bool__(self):
        return self.count > 0

    def process(self, trainer):
        if self.episode_tracker is not None:
            for reward, length in self.episode_tracker.drain():
                trainer._log_episode(reward, length)
            trainer.episodes.device_tracker = self.episode_tracker


class FastActingStep:
    @staticmethod
    def _parts(pol):
        """-> (cnn, lstm or None, the last layer's Linear, dueling) for CNN -> [LSTM ->] one Linear+ReLU policies, else None."""
        from rltime_amd.models.torch.modules import CNN, FC, LSTM
        model = pol.model
        layers = list(model.layers)
        if len(layers) == 3:
            cnn, lstm, fc = layers
            if not isinstance(lstm, LSTM) or not lstm.fused or lstm.lstm_cell.bias_ih is None:
                return None
        elif len(layers) == 2:
            (cnn, fc), lstm = layers, None
        else:
            return None
        if not (isinstance(cnn, CNN) and isinstance(fc, FC)) or model.extra_input_layer is not None:
            return None
        dueling = getattr(pol, "value_layer", None) is not None
        if dueling:
            lin = pol._fused_tail_layer()
        else:
            blocks = getattr(fc, "layers", None)
            lin = blocks[0][0] if (getattr(fc, "fuse_relu", False) and blocks is not None and len(blocks) == 1 and len(blocks[0]) == 1) else None
            if lin is not None and not (lin.weight.is_cuda and lin.weight.dtype == torch.float32 and lin.bias is not None):
                lin = None
        if lin is None or pol.out_layer.in_features != lin.out_features:
            return None
        return cnn, lstm, lin, dueling

    @staticmethod
    def supports(actor):
        """CNN (NHWC, input layer straight from uint8) -> [LSTM ->] one Linear+ReLU, plain or dueling head, IQN quantile
        layer (if any) injected before the last layer, epsilon-greedy or greedy."""
        pol = actor._policy
        try:
            if not pol.is_cuda():
                return False
            parts = FastActingStep._parts(pol)
            if parts is None:
                return False
            cnn = parts[0]
            if not (cnn.channels_last and cnn.direct_u8 and cnn.scale and len(cnn.layers) >= 1):
                return False
            pre = pol.model.layer_pre_processors
            iqn = hasattr(pol, "num_sampling_quantiles")
            if (iqn and set(pre) != {len(pol.model.layers) - 1}) or (not iqn and pre):
                return False
            expl = actor._exploration
            if expl is not None and not hasattr(expl, "_device_exponents"):
                return False
            space = actor._vec_env.observation_space
            probe = torch.empty((1,) + tuple(space.shape), dtype=torch.uint8, device=pol.device())
            return bool(conv_u8_supported(probe, cnn.layers[0])) and hasattr(actor._vec_env, "step_device")
        except Exception:
            return False

    def __init__(self, actor, obs0, need_q=True):
        """need_q=False: nobody reads the step's q-values (a replay that does not keep policy outputs): the dueling
        head's VALUE stream is then skipped — argmax_a mean_N (V + A - mean_a A) = argmax_a mean_N A (dqn.py:74-87), so the
        chosen actions are the same while the head's widest GEMM is half as wide."""
        self.actor = actor
        self.need_q = bool(need_q)
        pol = self.pol = actor._policy
        self.cnn, self.lstm, self.fc, self.dueling = self._parts(pol)
        self.fc_layer = pol.model.layers[-1]
        self.iqn = hasattr(pol, "num_sampling_quantiles")
        self.N = pol.num_sampling_quantiles if self.iqn else 1
        dev = self.dev = pol.device()
        E = self.E = actor._num_envs
        # without a recurrent layer (H = 0) the head reads the conv features themselves
        H = self.H = self.lstm.num_units if self.lstm is not None else 0
        F = self.F = self.lstm.inp_size if self.lstm is not None else self.fc.in_features
        W = self.W = H if self.lstm is not None else F            # width of the head's input rows
        A = self.A = actor._action_space.n
        f32 = dict(dtype=torch.float32, device=dev)
        c1 = self.cnn.layers[0]
        _, hh, ww = obs0.shape[1:]
        k, s = c1.kernel_size[0], c1.stride[0]
        self.y1 = torch.empty((E, c1.out_channels, (hh - k) // s + 1, (ww - k) // s + 1), memory_format=torch.channels_last, **f32)
        need = C.c_int64()
        check(lib.mirl_conv1_u8_wpk_floats(C.byref(need)))
        self.wpk = torch.empty(need.value, **f32)
        self.xh = torch.zeros((E, F + H), **f32)                 # [conv features (NCHW order) | masked h]
        self.c_in = torch.zeros((E, H), **f32)
        self.h = torch.zeros((E, H), **f32)                      # raw carry (outputs of the last cell)
        self.c = torch.zeros((E, H), **f32)
        self.state_pack = torch.zeros((E, 2 * H), **f32)
        self.initials = torch.ones(E, **f32)
        self.rewards = torch.zeros(E, **f32)
        self.dones = torch.ones(E, dtype=torch.uint8, device=dev)
        self.gates = torch.empty((E, 4 * H), **f32)
        self.actions = torch.zeros(E, dtype=torch.int32, device=dev)
        self.qvalues = torch.zeros((E, A), **f32)
        self.eps = torch.zeros((), dtype=torch.float64, device=dev)
        self.rng_step = torch.zeros(1, dtype=torch.int64, device=dev)
        self.step_no = 0
        self.rng_seed = (int(torch.initial_seed()) ^ (int(actor._base_env_id) << 32) ^ 0xAC7) & 0x7FFFFFFFFFFFFFFF
        if self.lstm is not None:
            self.bias_sum = torch.empty(4 * H, **f32)
            self.wcat = torch.empty((4 * H, F + H), **f32)
        h1, hv = self.fc.out_features, (pol.value_hidden_layer.out_features if self.dueling else 0)
        self.h1 = h1
        self.fc_w = torch.empty((h1 + hv, self.fc.in_features), **f32)
        self.fc_b = torch.empty(h1 + hv, **f32)
        # advantage and value outputs as ONE GEMM over the joint hidden activation: block-diagonal weights
        self.na, self.nq = pol.out_layer.out_features, (pol.value_layer.out_features if self.dueling else 0)
        assert self.na == A and self.nq == (1 if self.dueling else 0)
        self.out_w = torch.zeros((self.na + self.nq, h1 + hv), **f32)
        self.out_b = torch.zeros(self.na + self.nq, **f32)
        self.adv_w = torch.zeros((self.na, h1), **f32)           # advantage stream alone (need_q=False)
        self.freq = (pol.embedding_range * np.pi).contiguous() if self.iqn else None
        # the network's own kernels (csrc/actnet.hip), piece by piece where the shape is covered
        # MIRL_ACT_FUSED: 0 = library calls only, 1 (default) = own kernels where they measured faster (conv layers and the
        # quantile product at any batch; the LSTM step and the head's hidden layers at the acting batch of one rank of a
        # multi-GPU job: <= 64 envs, <= 2048 quantile rows; profiles/r05_actnet_probe.jsonl), 2 = everything at any size
        mode = os.environ.get("MIRL_ACT_FUSED", "1")
        fused = mode != "0"
        small = mode == "2" or E <= 64
        convs = list(self.cnn.layers)
        self.f_conv = False
        # (without a recurrent layer the features go to the head in the reference's (C, H, W) order: the shared conv path +
        # one reordering copy, models/torch/fused.conv_bias_relu — the same kernels under no_grad)
        if fused and len(convs) == 3 and self.lstm is not None:          # any batch: 16-pixel tiles below ~100 frames, LDS-resident weights above
            c2, c3 = convs[1], convs[2]
            h1o, w1o = self.y1.shape[2], self.y1.shape[3]
            k2, s2 = c2.kernel_size[0], c2.stride[0]
            h2o, w2o = (h1o - k2) // s2 + 1, (w1o - k2) // s2 + 1
            ok2 = lib.mirl_act_conv_supported(2, c2.in_channels, c2.out_channels, k2, s2, h1o, w1o) and c2.kernel_size[0] == c2.kernel_size[1]
            k3, s3 = c3.kernel_size[0], c3.stride[0]
            ok3 = lib.mirl_act_conv_supported(3, c3.in_channels, c3.out_channels, k3, s3, h2o, w2o) and c3.kernel_size[0] == c3.kernel_size[1]
            h3o, w3o = (h2o - k3) // s3 + 1, (w2o - k3) // s3 + 1
            if ok2 and ok3 and c2.bias is not None and c3.bias is not None and h3o * w3o * 64 == F and tuple(c2.padding) == (0, 0) == tuple(c3.padding):
                self.f_conv = True
                self.y2 = torch.empty((E, h2o, w2o, 64), **f32)
                self.w2p = torch.empty((64, k2 * k2 * c2.in_channels), **f32)
                self.w3p = torch.empty((64, k3 * k3 * c3.in_channels), **f32)
                self.conv_dims = (h1o, w1o, h2o, w2o, h3o, w3o)
        self.f_lstm = fused and self.lstm is not None and bool(lib.mirl_act_lstm_supported(E, H, F + H))
        D = int(self.freq.shape[0]) if self.iqn else 0
        self.f_head = fused and (mode == "2" or E * self.N <= 2048) and self.fc.in_features == W and self.dueling \
            and bool(lib.mirl_act_head_supported(E, self.N, W, D, h1 + hv, self.na + self.nq))
        # the quantile product as one launch at any batch (cos features + embedding product + ReLU + feature multiply)
        self.f_embed = fused and self.iqn and W % 16 == 0 and D % 16 == 0 and 0 < D <= 64 and E * self.N <= (1 << 24)
        self.xq = torch.empty((E * self.N, W), **f32) if self.f_embed else None      # quantile product rows
        if self.f_head:
            parts, pitch = C.c_int32(), C.c_int32()
            check(lib.mirl_act_head_parts(h1 + hv, self.na + self.nq, C.byref(parts), C.byref(pitch)))
            self.part = torch.zeros(parts.value * E * self.N * pitch.value, **f32)
        if self.f_lstm:
            need = C.c_int64()
            check(lib.mirl_act_lstm_workspace_bytes(E, H, F + H, C.byref(need)))
            self.lstm_ws = torch.zeros((need.value + 3) // 4, dtype=torch.int32, device=dev)   # zero once: the kernel restores it
        self.in_kernel_taus = self.iqn and getattr(pol, "tau_source", None) is None
        expl = actor._exploration
        self.expo = expl._device_exponents(actor._env_ids, dev) if expl is not None else None
        self.eps_min = float(expl.eps_min) if expl is not None else 0.0
        self.last_obs = obs0
        # a frame-stack env produces its observations by the shift contract itself: de-duplicated
        # storage can take the newest plane without re-verifying it (MIRL_DEDUP_VERIFY=1: whole stacks)
        self.trusted_stack = bool(getattr(actor._vec_env, "frame_stack", False)) and os.environ.get("MIRL_DEDUP_VERIFY", "0") != "1"
        self.tracker = None
        self.graph = None
        # env steps that write static buffers with a fixed launch (capturable); the per-step fast path uses them too, so
        # that the rollout graph and the per-step path see the same env stream
        env = actor._vec_env
        self.env_into = bool(getattr(env, "supports_step_into", lambda: False)())
        self.env_pre = self.env_into and hasattr(env, "step_into_args") and os.environ.get("MIRL_ACT_ENV_PRE", "1") != "0"
        if self.env_into:
            self.obs_buf = torch.empty_like(obs0)
            self.env_rewards = torch.zeros(E, **f32)
            self.env_dones = torch.zeros(E, dtype=torch.uint8, device=dev)
        self._rollouts = {}              # (iters, keep_policy, clip, sink id) -> [calls so far, CUDAGraph or None]
        self.rollout_graphs = os.environ.get("MIRL_ROLLOUT_GRAPH", "1") != "0"
        self.rollout_eager_calls = int(os.environ.get("MIRL_ROLLOUT_EAGER_CALLS", "0"))
        assert self.lstm is None or self.lstm.lstm_cell.weight_ih.shape == (4 * H, F)
        # the reference's first input state: every env starts an episode (actor.py:78-89)
        self.refresh()
        self._pre(torch.zeros(E, **f32), torch.ones(E, dtype=torch.uint8, device=dev), track=False)
        if self.lstm is not None:
            self.example = {"x": obs0[0].cpu().numpy(), "layer0_state": {},
                            "layer1_state": {"hx": np.zeros(H, np.float32), "cx": np.zeros(H, np.float32), "initials": np.float32(1.0)},
                            "layer2_state": {}}
        else:
            self.example = {"x": obs0[0].cpu().numpy(), "layer0_state": {}, "layer1_state": {}}
        self._capture()

    # -- weight-only quantities, once per get_samples call -------------------------------
    def refresh(self):
        pol, F = self.pol, self.F
        with torch.no_grad():
            if self.lstm is not None:
                cell = self.lstm.lstm_cell
                torch.add(cell.bias_ih, cell.bias_hh, out=self.bias_sum)
                if self.f_conv:
                    # layer 3 writes its NHWC rows straight into xh: W_ih's columns follow (same products, modules.LSTM._flat_input)
                    c2, c3 = self.cnn.layers[1], self.cnn.layers[2]
                    h3o, w3o = self.conv_dims[4], self.conv_dims[5]
                    self.wcat[:, :F].view(-1, h3o, w3o, 64).copy_(cell.weight_ih.view(-1, 64, h3o, w3o).permute(0, 2, 3, 1))
                    self.w2p.view(64, c2.kernel_size[0], c2.kernel_size[1], c2.in_channels).copy_(c2.weight.permute(0, 2, 3, 1))
                    self.w3p.view(64, c3.kernel_size[0], c3.kernel_size[1], c3.in_channels).copy_(c3.weight.permute(0, 2, 3, 1))
                else:
                    self.wcat[:, :F].copy_(cell.weight_ih)
                self.wcat[:, F:].copy_(cell.weight_hh)
            na = self.na
            # one multi-tensor copy for the head's buffers instead of a launch each (the T = 1 configs refresh every learner step)
            dst = [self.fc_w[:self.h1], self.fc_b[:self.h1], self.out_w[:na, :self.h1], self.adv_w, self.out_b[:na]]
            src = [self.fc.weight, self.fc.bias, pol.out_layer.weight, pol.out_layer.weight, pol.out_layer.bias]
            if self.dueling:
                dst += [self.fc_w[self.h1:], self.fc_b[self.h1:], self.out_w[na:, self.h1:], self.out_b[na:]]
                src += [pol.value_hidden_layer.weight, pol.value_hidden_layer.bias, pol.value_layer.weight, pol.value_layer.bias]
            torch._foreach_copy_(dst, [t.detach() for t in src])

    def set_eps(self, eps):
        self.eps.fill_(eps)

    # -- pieces ------------------------------------------------------------------------------
    def _pre_args(self, tr, row, clip):
        rec = self.H > 0
        return (self.H, self.A, _p(self.actions), _p(self.h) if rec else None, _p(self.c) if rec else None,
                C.c_void_p(self.xh.data_ptr() + 4 * self.F) if rec else None, self.F + self.H,
                _p(self.c_in) if rec else None, _p(self.state_pack) if rec else None, _p(self.initials),
                _p(self.rewards), _p(self.dones), 1 if clip else 0,
                _p(tr.ep_reward) if tr is not None else None, _p(tr.ep_len) if tr is not None else None,
                _p(tr.out_reward[row]) if tr is not None else None, _p(tr.out_len[row]) if tr is not None else None,
                _p(tr.action_counts) if tr is not None else None, _p(self.rng_step), 0xFFFFFFFFFFFFFFFF, _stream())

    def _pre(self, rewards, dones_u8, track=True, clip=False, row=None):
        """row: the episode tracker's ring row (None: reserve the next one).  The step counter the in-kernel draws are
        keyed with advances ON THE DEVICE (MIRL_STEP_ADVANCE); step_no mirrors it on the host."""
        tr = self.tracker if track else None
        if tr is not None and row is None:
            row = tr.begin_step()
        self.step_no += 1
        a = self._pre_args(tr, row, clip)
        check(lib.mirl_actor_pre(self.E, a[0], a[1], _p(rewards), _p(dones_u8), *a[2:]), "mirl_actor_pre")

    def env_step_pre(self, clip=False, row=None):
        """env.step and the pre-step as ONE launch (csrc/acting.hip k_synth_env_step<true>): the workgroup that draws an
        env's reward / done applies them to that env's recurrent carry, stored state and episode statistics."""
        env = self.actor._vec_env
        tr = self.tracker
        if tr is not None and row is None:
            row = tr.begin_step()
        self.step_no += 1
        check(lib.mirl_synth_env_step_pre(*env.step_into_args(self.obs_buf, self.env_rewards, self.env_dones), *self._pre_args(tr, row, clip)),
              "mirl_synth_env_step_pre")
        env.advance_host()
        return self.obs_buf

    def _conv1(self, obs, packed=False):
        """packed=True: self.wpk still holds the current weights (packed by the call's re-selection)."""
        c1 = self.cnn.layers[0]
        so, sc, sh, sw = c1.weight.stride()
        check(lib.mirl_conv1_u8_fwd_ex(obs.shape[0], obs.shape[2], obs.shape[3], _p(obs), _p(c1.weight), so, sc, sh, sw, _p(c1.bias),
                                       float(self.cnn.scale), _p(self.wpk), _p(self.y1), 8 if packed else 0, _stream()),
              "mirl_conv1_u8_fwd")

    def _body(self):
        """conv 2.. -> LSTM step -> head; reads y1 / xh tail / c_in, writes h, c, actions, qvalues."""
        pol, E, H, F, N = self.pol, self.E, self.H, self.F, self.N
        if self.f_conv:
            c2, c3 = self.cnn.layers[1], self.cnn.layers[2]
            h1o, w1o, h2o, w2o, h3o, w3o = self.conv_dims
            check(lib.mirl_act_conv_fwd(2, E, h1o, w1o, _p(self.y1), _p(self.w2p), _p(c2.bias), _p(self.y2), h2o * w2o * 64, _stream()),
                  "mirl_act_conv_fwd")
            check(lib.mirl_act_conv_fwd(3, E, h2o, w2o, _p(self.y2), _p(self.w3p), _p(c3.bias), _p(self.xh), F + H, _stream()),
                  "mirl_act_conv_fwd")
        else:
            x = self.y1
            for layer in self.cnn.layers[1:]:
                x = conv_bias_relu(x, layer)
            ch, hh, ww = x.shape[1:]
            torch.as_strided(self.xh, (E, ch, hh, ww), (F + H, hh * ww, ww, 1)).copy_(x)   # NHWC -> the reference's (C, H, W) flatten
        if self.f_lstm:
            check(lib.mirl_act_lstm_fwd(E, H, F + H, _p(self.xh), F + H, _p(self.wcat), _p(self.bias_sum), _p(self.c_in), _p(self.h),
                                        _p(self.c), _p(self.lstm_ws), _stream()), "mirl_act_lstm_fwd")
        elif self.lstm is not None:
            torch.addmm(self.bias_sum, self.xh, self.wcat.t(), out=self.gates)
            check(lib.mirl_lstm_cell_fwd(E, H, _p(self.gates), _p(self.c_in), None, None, None, _p(self.h), _p(self.c), _stream()),
                  "mirl_lstm_cell_fwd")
        greedy = self.expo is None
        eps_p, expo_p = (None, None) if greedy else (_p(self.eps), _p(self.expo))
        feat = self.h if self.lstm is not None else self.xh       # (E, W) rows
        H = self.W                                                # from here on: the head's input width
        if self.f_embed:
            # quantile fractions -> cos features -> embedding product + ReLU -> x features: one launch at any batch
            taus = None
            if not self.in_kernel_taus:
                taus = pol._draw_taus(E * N).contiguous()        # a test's tau_source replaces the draw
                self._taus_keep = taus
            check(lib.mirl_act_embed(E, N, H, int(self.freq.shape[0]), _p(feat), _p(self.freq), _p(taus), self.rng_seed, _p(self.rng_step),
                                     _p(pol.quantile_layer.weight), _p(pol.quantile_layer.bias), _p(self.xq), None, _stream()), "mirl_act_embed")
            feat = self.xq
        elif self.iqn:
            if self.in_kernel_taus:
                phi = torch.empty((E * N, self.freq.shape[0]), dtype=torch.float32, device=self.dev)
                check(lib.mirl_cos_embed_rng(E * N, self.freq.shape[0], self.rng_seed, _p(self.rng_step), _p(self.freq), _p(phi),
                                             None, _stream()), "mirl_cos_embed_rng")
            else:                                            # a test's tau_source replaces the draw
                phi = cos_embed(pol._draw_taus(E * N), self.freq)
            emb = torch._addmm_activation(pol.quantile_layer.bias, phi, pol.quantile_layer.weight.t(), use_gelu=False)
            if H % 4 == 0 and 256 % (H // 4) == 0:
                check(lib.mirl_iqn_mul_fwd(E, N, H, _p(feat), _p(emb), _p(emb), _stream()), "mirl_iqn_mul_fwd")   # in place
            else:
                emb = (emb.view(E, N, H) * feat.view(E, 1, H)).view(E * N, H)
            feat = emb
        use_val = self.need_q and self.dueling
        if self.f_head:
            hid = self.fc_w.shape[0] if self.need_q else self.h1
            no = self.na + (self.nq if self.need_q else 0)
            wout = self.out_w if self.need_q else self.adv_w
            parts, pitch = C.c_int32(), C.c_int32()
            check(lib.mirl_act_head_parts(hid, no, C.byref(parts), C.byref(pitch)))
            check(lib.mirl_act_head_hidden(E * N, H, hid, no, _p(feat), _p(self.fc_w), _p(self.fc_b), _p(wout), _p(self.part), _stream()),
                  "mirl_act_head_hidden")
            check(lib.mirl_act_head_select(
                E, N, self.A, parts.value, pitch.value, _p(self.part), _p(self.out_b), 1 if self.need_q else 0,
                eps_p, expo_p, self.eps_min, self.rng_seed, _p(self.rng_step), _p(self.actions), _p(self.qvalues), _stream()),
                "mirl_act_head_select")
            return
        if use_val:
            # the joint [last FC | value-hidden] layer: at 256 envs (8 192 quantile rows) the split-bf16 kernel's 256 x 128
            # tile fills the chip (csrc/gemm3.hip k_gemm3_mid: 55 us against the library's 73); smaller batches stay on
            # the library through linear_fwd's own gates
            both = gemm3.linear_fwd(feat, self.fc_w, self.fc_b, relu=True)
            outs = torch.addmm(self.out_b, both, self.out_w.t())          # (rows, A + 1): [advantages | value]
            pitch, val = self.na + self.nq, C.c_void_p(outs.data_ptr() + 4 * self.na)
        else:
            hidden = torch._addmm_activation(self.fc_b[:self.h1], feat, self.fc_w[:self.h1].t(), use_gelu=False)
            outs = torch.addmm(self.out_b[:self.na], hidden, self.adv_w.t())   # (rows, A): the advantage stream
            pitch, val = self.na, None
        check(lib.mirl_actor_head_rng(
            E, N, self.A, _p(outs), pitch, val, pitch, None if greedy else _p(self.eps), None if greedy else _p(self.expo),
            self.eps_min, self.rng_seed, None if greedy else _p(self.rng_step), _p(self.actions), _p(self.qvalues), None, _stream()),
            "mirl_actor_head_rng")

    def _capture(self):
        keep = (self.h.clone(), self.c.clone())
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side), torch.no_grad():          # warm-up outside capture (MIOpen find, hipBLASLt workspaces)
            self._conv1(self.last_obs)
            for _ in range(3):
                self._body()
        cur.wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with quiet_gc(), torch.no_grad(), torch.cuda.graph(graph, capture_error_mode="thread_local"):
            self._body()
        self.graph = graph
        self.h.copy_(keep[0])
        self.c.copy_(keep[1])
        self.reselect()
        self.selected_with = None         # the caller (actor._fast_steps) refreshes the weight buffers and selects again

    def set_need_q(self, need_q):
        """Switch between the full dueling head and the advantage stream alone; the step graphs are re-captured."""
        if bool(need_q) != self.need_q:
            self.need_q = bool(need_q)
            self._rollouts.clear()
            self._capture()

    # -- the acting loop's entry points ---------------------------------------------------------
    def reselect(self):
        """Forward + exploration on the CURRENT input state with the current weights (the first
        action after a learner update, actor.py:108-122); idempotent on the recurrent carry."""
        self._conv1(self.last_obs)
        self.graph.replay()

    def step(self, obs, rewards, dones, sink=None, keep_policy=False, clip=False, pre_done=False):
        """One vector step AFTER env.step(self.actions) returned (obs, rewards, dones).  With a
        `sink` (device replay) the transition is ingested straight from the static buffers;
        without one the caller gets clones of them (DeviceSamples fields).  pre_done: the pre-step already ran
        with the env step (env_step_pre)."""
        if not pre_done:
            dones_u8 = dones.view(torch.uint8) if dones.dtype == torch.bool else dones.to(torch.uint8)
            self._pre(rewards if rewards.dtype == torch.float32 else rewards.float(), dones_u8, clip=clip)
        fields = None
        if sink is not None:
            rec = self.H > 0
            sink.update_batch(obs, self.actions, self.rewards, self.dones, state=self.state_pack if rec else None,
                              initials=self.initials if rec else None,
                              policy=self.qvalues if keep_policy else None, transient=True,
                              # acting-time priority init reads whole stored stacks through the verifying ingest
                              newest_plane_only=(bool(getattr(sink, "_dedup", False)) and self.trusted_stack
                                                 and not getattr(sink, "_acting_priority_init", False)))
        else:
            if self.env_into and obs is self.obs_buf:
                obs = obs.clone()                 # the caller keeps it; the static block is rewritten by the next env step
            fields = dict(frames=obs, actions=self.actions.clone(),
                          policy=self.qvalues.clone(), rewards=self.rewards.clone(), dones=self.dones.clone(), episode_stats=None)
            if self.H > 0:
                fields.update(state=self.state_pack.clone(), initials=self.initials.clone())
        self.last_obs = obs
        self._conv1(obs, packed=True)
        self.graph.replay()
        return fields

    def env_step(self):
        """One env step into the static buffers (same kernel eagerly or captured)."""
        self.actor._vec_env.step_into(self.obs_buf, self.env_rewards, self.env_dones)
        return self.obs_buf, self.env_rewards, self.env_dones

    # -- a whole get_samples call as one graph launch --------------------------------------------------
    @staticmethod
    def _sink_key(sink):
        h = getattr(sink, "_h", None)
        return ("replay", int(getattr(h, "value", None) or id(sink)))

    def forget_rollouts(self):
        """Drop every captured rollout (the sink was reconfigured or destroyed: its device structures are gone)."""
        self._rollouts.clear()

    def can_rollout(self, iters, sink):
        # (a single step too: for the T = 1 configs, whose learner step is one graph launch, the per-step path's handful of
        # host launches per acting call is what the GPU would wait for)
        return (self.rollout_graphs and self.env_into and sink is not None and iters >= 1 and self.tracker is not None
                and iters <= self.tracker.ROWS and getattr(sink, "supports_planned_ingest", lambda: False)())

    def _rollout_body(self, iters, sink, keep_policy, clip):
        for k in range(iters):
            if self.env_pre:
                obs = self.env_step_pre(clip=clip, row=k)
            else:
                obs, rewards, dones = self.env_step()
                self._pre(rewards, dones, clip=clip, row=k)
            rec = self.H > 0
            sink.ingest_planned(k, obs, self.actions, self.rewards, self.dones, state=self.state_pack if rec else None,
                                initials=self.initials if rec else None, policy=self.qvalues if keep_policy else None)
            self._conv1(obs, packed=True)
            self._body()

    def rollout(self, iters, sink, keep_policy=False, clip=False):
        """`iters` vector steps after reselect(): the first call of a shape captures them into ONE HIP graph, every call
        is a plan upload + a graph launch.  Identical results to the same launches issued one by one
        (tests/test_fast_acting_gpu.py)."""
        env = self.actor._vec_env
        parity = env.clock_parity() if hasattr(env, "clock_parity") else 0
        # a captured env step reads a fixed clock word; the sink is keyed by its replay HANDLE (the Dev struct and plan
        # buffer baked into the graph belong to it — id() of a destroyed sink can be recycled)
        key = (iters, bool(keep_policy), bool(clip), self._sink_key(sink), parity)
        state = self._rollouts.setdefault(key, [0, None])
        from rltime_amd._lib import MirlError, MIRL_ERR_ARG
        try:
            sink.plan_ingest(iters, self.E)
        except MirlError as e:
            # mirl_replay_ingest_plan is transactional: MIRL_ERR_ARG is returned BEFORE any bookkeeping (book, rings and
            # plan buffer untouched), so acting can go on per step.  Anything else (a failure after the book moved: the
            # shard is marked broken in the library) must not be papered over.
            if getattr(e, "code", None) != MIRL_ERR_ARG:
                raise
            import logging
            logging.getLogger().warning("rollout plan refused (%s); acting per step", e)
            self.rollout_graphs = False
            return False
        self.tracker.begin_rollout(iters)
        state[0] += 1
        # capture at the FIRST call of a shape: every kernel of the body has run before (the single-step graph's warm-ups,
        # reselect, the constructor's pre-step), so nothing is left to initialise inside the capture.
        # MIRL_ROLLOUT_EAGER_CALLS=n runs the first n calls eagerly instead (the same launches: debugging, A/B test)
        captured_now = False
        if state[1] is None and state[0] > self.rollout_eager_calls:
            captured_now = True
            keep_step = self.step_no
            graph = torch.cuda.CUDAGraph()
            with quiet_gc(), torch.no_grad(), torch.cuda.graph(graph, capture_error_mode="thread_local"):
                self._rollout_body(iters, sink, keep_policy, clip)
            self.step_no = keep_step                 # capture enqueues nothing: the counters have not moved
            state[1] = graph
        if state[1] is not None:
            state[1].replay()
            self.step_no += iters
            if not captured_now and hasattr(env, "skip_host"):
                env.skip_host(iters)                 # (a capture has already walked the host-side parity through its steps)
        else:
            with torch.no_grad():
                self._rollout_body(iters, sink, keep_policy, clip)
        self.tracker.end_rollout(iters)
        self.last_obs = self.obs_buf
        return True

    # -- resume --------------------------------------------------------------------------------------
    def get_state(self):
        keys = ("h", "c", "xh", "c_in", "state_pack", "initials", "actions", "qvalues", "rewards", "dones")
        st = {k: getattr(self, k).detach().clone().cpu() for k in keys}
        st["xh"] = self.xh[:, self.F:].detach().clone().cpu()
        st.update(step_no=self.step_no, last_obs=self.last_obs.detach().clone().cpu())
        return st

    def set_state(self, st):
        for k in ("h", "c", "c_in", "state_pack", "initials", "actions", "qvalues", "rewards", "dones"):
            getattr(self, k).copy_(st[k].to(self.dev))
        self.xh[:, self.F:].copy_(st["xh"].to(self.dev))
        self.step_no = st["step_no"]
        self.rng_step.fill_(self.step_no)
        self.last_obs = st["last_obs"].to(self.dev)
        self.selected_with = None


def example_input_state(policy, obs, dones):
    """One transition's next_state pytree as host arrays (the device replay's layout probe)."""
    return deep_apply(policy.make_input_state(obs, dones), lambda x: x[0].cpu().numpy())
