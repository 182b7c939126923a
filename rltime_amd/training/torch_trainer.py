"""TorchTrainer (reference rltime/training/torch/torch_trainer.py:6-199):
optimizer, gradient-norm clipping, the n-step bootstrap target tail."""
import os

import torch

from .multi_step_trainer import MultiStepTrainer
from rltime_amd.models.torch.utils import set_lr


class TorchTrainer(MultiStepTrainer):
    def _train(self, clip_grad=None, clip_grad_dynamic_alpha=None, adam_epsilon=1e-8,
               vf_scale_epsilon=None, apply_initial_lr=False, burn_in_full_forward=False,
               share_online_cnn=True, share_online_projection=True, selection_advantage_only=True,
               graph_learner_step=False, **kwargs):
        """torch_trainer.py:9-44.  apply_initial_lr=False mirrors the reference,
        whose train_init ignores `lr` (Adam starts at 1e-3, SURVEY A-14)."""
        self.clip_grad = float(clip_grad) if clip_grad is not None else None
        self.clip_grad_dynamic_alpha = clip_grad_dynamic_alpha
        self._grad_norm_moving_average = None
        self.adam_epsilon = adam_epsilon
        assert vf_scale_epsilon is None or vf_scale_epsilon > 0
        assert (not vf_scale_epsilon) or (not self.clip_rewards), \
            "Value function rescaling only makes sense with clip_rewards=False"
        self.vf_scale_epsilon = vf_scale_epsilon
        self._apply_initial_lr = apply_initial_lr
        self.burn_in_full_forward = burn_in_full_forward
        self.share_online_cnn = share_online_cnn
        self.share_online_projection = share_online_projection
        # double-Q action selection from the dueling head's advantage stream alone (policies/dqn.py predict_selection):
        # the same arg-max, half of the selection pass's widest GEMM
        self.selection_advantage_only = selection_advantage_only
        # graph_learner_step (not in the reference): targets -> forward / backward -> clip -> Adam of one learner step as ONE
        # captured HIP graph (see _learner_step_graphed).  For the launch-bound T = 1 configs, where the ~150 launches of a
        # step cost more host time than GPU time; needs static batch buffers and a capturable optimizer, so it is decided
        # here, before train_init.  MIRL_GRAPH_STEP=0 switches it off.
        # ("no-capture": the same set-up — static batches, capturable Adam — with every step issued eagerly: the A/B of the tests)
        self.graph_learner_step = bool(graph_learner_step) and os.environ.get("MIRL_GRAPH_STEP", "1") != "0"
        self._graph_capture = graph_learner_step != "no-capture"
        self._gstep = None
        super()._train(**kwargs)

    def train_init(self, lr):
        """torch_trainer.py:80-83: Adam over the policy's parameters.  On a GPU it is `ClipAdam` — the same optimizer
        (state, state_dict, param_groups) whose update takes the gradient clip with it in two launches
        (csrc/optim.hip, train_batch below); MIRL_CLIP_ADAM=0 keeps torch.optim.Adam's own kernels."""
        kw = {"lr": lr} if self._apply_initial_lr else {}
        graphed = getattr(self, "graph_learner_step", False) and self.policy.is_cuda()
        if graphed:
            # the step counters and the learning rate live on the device: the update can be captured and replayed
            dev = next(self.policy.parameters()).device
            kw = {"lr": torch.tensor(float(kw.get("lr", 1e-3)), dtype=torch.float32, device=dev)}
        if self.policy.is_cuda() and os.environ.get("MIRL_CLIP_ADAM", "1") != "0":
            from rltime_amd.models.torch.optim import ClipAdam
            self.optimizer = ClipAdam(self.policy.parameters(), eps=self.adam_epsilon, **kw)
            return
        if graphed:
            # (torch's own `fused=True` form was tried here: under capture with a device learning rate its replays left the
            # eager trajectory at the first replayed step — tools/graph_step_deviation.py — so the fallback is the for-each form)
            kw["capturable"] = True
        self.optimizer = torch.optim.Adam(self.policy.parameters(), eps=self.adam_epsilon, **kw)

    def set_lr(self, lr):
        set_lr(self.optimizer, lr)

    def _compute_grads(self, states, targets, policy_outputs, extra_data, timesteps):
        raise NotImplementedError

    # -- the reference's target hooks (torch_trainer.py:91-147).  DQN / IQN override
    # calc_target_values with one fused kernel after their forwards; a subclass that
    # only supplies `_get_bootstrap_target_value` (the reference's plugin contract)
    # gets the generic composition below, same arithmetic in plain torch ops.
    def _get_bootstrap_target_value(self, target_states, timesteps):
        raise NotImplementedError

    def _discount_bootstrap_target_value(self, target_values, nsteps):
        return (self.gamma ** nsteps) * target_values

    def _vf_scale(self, x):
        """torch_trainer.py:46-52."""
        eps = self.vf_scale_epsilon
        if eps is None:
            return x
        return torch.sign(x) * (torch.sqrt(torch.abs(x) + 1) - 1) + eps * x

    def _vf_unscale(self, y):
        """torch_trainer.py:54-78 (float64 inside, float32 out)."""
        eps = self.vf_scale_epsilon
        if eps is None:
            return y
        y64 = y.double()
        a = torch.abs(y64)
        x = a / eps - (1 / (2. * eps ** 2)) * torch.sqrt(4 * eps * a + (2. * eps + 1) ** 2) + (2. * eps + 1) / (2. * eps ** 2)
        return (x * torch.sign(y64)).float()

    def calc_target_values(self, returns, target_states, target_masks, nsteps, timesteps):
        """torch_trainer.py:101-147."""
        with torch.no_grad():
            target_states, returns, target_masks, nsteps = self.target_policy.make_tensor(
                (target_states, returns, target_masks, nsteps), non_blocking=True)
            v = self._vf_unscale(self._get_bootstrap_target_value(target_states, timesteps))
            assert returns.shape == target_masks.shape == nsteps.shape == (v.shape[0],)
            assert v.dim() in (1, 2)
            if v.dim() == 2:
                returns, target_masks, nsteps = (t.unsqueeze(-1) for t in (returns, target_masks, nsteps))
            return self._vf_scale(returns + self._discount_bootstrap_target_value(v, nsteps) * target_masks)

    def _get_grad_norm_clip_value(self, norm):
        """torch_trainer.py:153-175: fixed clip or clip_grad x EMA(norm); the EMA
        lives on the device."""
        if not self.clip_grad:
            return None
        if self.clip_grad_dynamic_alpha is None:
            return self.clip_grad
        a = self.clip_grad_dynamic_alpha
        self._grad_norm_moving_average = norm.detach().clone() if self._grad_norm_moving_average is None \
            else self._grad_norm_moving_average * a + norm.detach() * (1 - a)
        self.value_log.log("grad_norm_ma", self._grad_norm_moving_average, group="train")
        return self._grad_norm_moving_average * self.clip_grad

    def train_batch(self, *args, **kwargs):
        """torch_trainer.py:177-199.  The norm is one fused device reduction and
        the clip a device-side scale: no host synchronisation per step (the
        reference does ~2 x #parameters `.item()` calls, torch_policy.py:70-78)."""
        dp = getattr(self, "data_parallel", None)
        if dp is not None and dp._flat is not None:
            dp.zero_grad()                  # .grad tensors are views of the all-reduce bucket
        else:
            self.optimizer.zero_grad(set_to_none=True)
        self._compute_grads(*args, **kwargs)
        self._reduce_gradients()
        opt = self.optimizer
        if hasattr(opt, "step_clipped") and self.clip_grad_dynamic_alpha is None and opt.fused_step_ok():
            # norm -> clip -> Adam as two launches (the clipped gradients are left in .grad like clip_grad_norm_ does)
            if self.policy.is_cuda() and not torch.cuda.is_current_stream_capturing():
                self._check_sweeps()
            norms = opt.step_clipped(self.clip_grad if self.clip_grad else None)
            self.value_log.log("grad_norm", norms[0], group="train")
            if self.clip_grad:
                self.value_log.log("grad_norm_clipped", norms[1], group="train")
            return
        if hasattr(opt, "why_not_fused") and os.environ.get("MIRL_CLIP_ADAM_WHY") == "1":
            print("ClipAdam not used:", "dynamic clip" if self.clip_grad_dynamic_alpha is not None else opt.why_not_fused(), flush=True)
        params = [p for p in self.policy.parameters() if p.grad is not None]
        grads = [p.grad for p in params]
        norm = torch.linalg.vector_norm(torch.stack(torch._foreach_norm(grads, 2)), 2)
        self.value_log.log("grad_norm", norm, group="train")
        clip = self._get_grad_norm_clip_value(norm)
        if clip is not None:
            coef = torch.clamp(clip / (norm + 1e-6), max=1.0)     # torch.nn.utils.clip_grad_norm_
            torch._foreach_mul_(grads, coef)
            self.value_log.log("grad_norm_clipped", norm * coef, group="train")
        if self.policy.is_cuda() and not torch.cuda.is_current_stream_capturing():
            self._check_sweeps()
        self.optimizer.step()

    def _check_sweeps(self):
        """A persistent LSTM sweep that gave up (csrc/lstm_seq.hip: bounded spin, non-finite state) sets a pinned host
        word.  Reading it only means something at a point the host is SYNCHRONISED with the sweeps it covers — the host
        enqueues about a step ahead of the GPU.  Default: wait for the event recorded after the PREVIOUS step's backward
        (the GPU still holds a whole step of queued work, so it never idles), then read: a failed step k raises before
        step k+1's optimizer is enqueued and before anything of it is logged or checkpointed (policy_trainer.py checks
        again behind its own synchronisation).  At the point of that raise step k's OWN update has already been applied
        (the fused clip + Adam is enqueued right after this look, and a graphed step contains it): the weights and Adam
        moments in memory are then invalid and the run must be restarted from its last checkpoint — which is intact,
        because log rows and both kinds of checkpoint sit behind a synchronised check_status() (policy_trainer.py).
        MIRL_STRICT_SWEEP_CHECK=1 waits for THIS step's backward instead — no invalid gradient can reach the optimizer
        at all, at the price of a drained launch queue per step."""
        from rltime_amd.models.torch import lstm_seq
        if os.environ.get("MIRL_STRICT_SWEEP_CHECK", "0") == "1":
            torch.cuda.current_stream().synchronize()
            lstm_seq.check_status()
            return
        prev = getattr(self, "_backward_done", None)
        if prev is not None:
            prev.synchronize()
            lstm_seq.check_status()
        self._backward_done = torch.cuda.Event()
        self._backward_done.record()

    def _reduce_gradients(self):
        """Data-parallel hook: all-reduce the gradients across ranks (RCCL) when
        a process group is attached (rltime_amd.parallel); no-op on 1 GPU."""
        dp = getattr(self, "data_parallel", None)
        if dp is not None:
            dp.all_reduce_gradients(self.policy)

    # -- one learner step as ONE captured HIP graph -------------------------------------------------------------------
    # The T = 1 configs (DQN + uniform replay B = 256, Rainbow-IQN B = 512) run ~150 kernels of 5-25 us per learner step:
    # 2.16 ms per step of which 1.0 ms is GPU work (profiles/r04_bench_dqn_uniform.json) — the host's launch rate is the
    # bound.  Everything between the gathered batch and the new weights is data-independent control flow, so it is captured
    # once and replayed: frame conversion -> target net + selection forwards -> target kernel -> online forward -> loss kernel
    # -> backward -> gradient norm / clip -> Adam.  Outside the graph, per step: acting (its own rollout graph), sampling and
    # gather (host bookkeeping; they write the SAME buffers every step: History.static_batches), update_losses (its epoch
    # stamp is a launch argument), the learning-rate fill, the logged scalars (one stack), the host counters.
    # What a replay cannot do is run Python, so: (i) every version-keyed operand cache rebuilds inside the capture
    # (gemm3.REFRESH_ALWAYS) and therefore with every replay; (ii) the parameters' version counters are bumped by hand after
    # a replay (the actor's "weights unchanged" stamp reads them); (iii) the persistent-LSTM status check runs after the
    # replay instead of inside train_batch.  The first three steps of a batch shape run eagerly (lazy library state, Adam's
    # state tensors), the fourth is captured and replayed.  Same arithmetic, same order: bit-identical to the same set-up
    # issued eagerly (tests/test_graph_step_gpu.py: every loss, gradient norm and final weight of a whole run).  Against the
    # eager step with torch.optim.Adam the trajectory differs by float32 rounding only (the learning rate is a float32 device
    # word here, a double there; csrc/optim.hip evaluates the bias corrections in float64 like the host-side Adam does).
    def _graph_step_ok(self, train_data, burn_in_timesteps, epochs, minibatches):
        if not getattr(self, "graph_learner_step", False) or not self.policy.is_cuda():
            return False
        dp = getattr(self, "data_parallel", None)
        # (burn-in included since round 6: the prefix passes, the recurrent state substitution and the row drop are
        # tensor ops on the static batch buffers; the persistent LSTM sweeps capture like any other kernel)
        return (epochs * minibatches == 1 and (dp is None or not dp.active) and self._ov is None
                and self.clip_grad_dynamic_alpha is None and getattr(self.history_buffer, "static_batches", False))

    @staticmethod
    def _leaves(tree, out=None):
        out = [] if out is None else out
        if isinstance(tree, dict):
            for k in sorted(tree):
                TorchTrainer._leaves(tree[k], out)
        elif isinstance(tree, (list, tuple)):
            for v in tree:
                TorchTrainer._leaves(v, out)
        elif isinstance(tree, torch.Tensor):
            out.append(tree)
        return out

    def _graph_step_body(self, train_data, nstep_target, rnn_steps_train, rnn_bootstrap, burn_in_timesteps=0):
        from .multi_step_trainer import _flat
        from rltime_amd.general.utils import deep_apply
        self._prepare_frames(train_data)
        if burn_in_timesteps:
            train_data = self._burn_in(train_data, burn_in_timesteps, do_target_states=rnn_bootstrap)
        self._share_online_features(train_data, nstep_target)
        flat = deep_apply(train_data, _flat)
        flat["targets"] = self.calc_target_values(
            flat["returns"], flat["target_states"], flat["target_masks"], nsteps=flat["nsteps"],
            timesteps=1 if not rnn_bootstrap else rnn_steps_train)
        self.train_batch(flat["states"], flat["targets"], flat["policy_outputs"], flat["extra_data"], rnn_steps_train)

    def _learner_step_graphed(self, train_data, nstep_train, nstep_target, rnn_steps_train, rnn_bootstrap, burn_in_timesteps=0):
        from rltime_amd.general.utils import quiet_gc
        from rltime_amd.models.torch import gemm3, lstm_seq
        st = self._gstep
        sig = tuple((t.data_ptr(), tuple(t.shape), t.dtype) for t in self._leaves(train_data))
        if st is None or st["sig"] != sig:
            st = self._gstep = {"sig": sig, "eager": 0, "graph": None, "logs": [], "losses": None, "data": None}
        batch_size = (train_data["returns"].shape[0] - burn_in_timesteps) * train_data["returns"].shape[1]
        self.get_train_indexes(batch_size, batch_size, nstep_train)          # the reference's np.random.shuffle is consumed
        if st["graph"] is None and (st["eager"] < 3 or not self._graph_capture):
            st["eager"] += 1
            self._graph_step_body(dict(train_data), nstep_target, rnn_steps_train, rnn_bootstrap, burn_in_timesteps)
        else:
            if st["graph"] is None:
                for g in self.optimizer.param_groups:
                    # a host learning rate would be baked into the captured update as a launch argument
                    if not (torch.is_tensor(g["lr"]) and g["lr"].is_cuda):
                        raise RuntimeError("graph_learner_step: the optimizer's learning rate must be a device tensor "
                                           "(train_init creates it; a checkpoint load must keep it: training/resume.py)")
                logs, real_log = st["logs"], self.value_log.log

                def tap(key, value, *a, **k):
                    if isinstance(value, torch.Tensor) and value.is_cuda:
                        logs.append((key, value, a, k))
                    else:
                        real_log(key, value, *a, **k)
                self.value_log.log = tap
                self._defer_losses = []
                gemm3.REFRESH_ALWAYS = True
                graph = torch.cuda.CUDAGraph()
                # the eager steps' cached blocks go back to the driver first: the capture's private pool holds a whole
                # step's intermediates for good (tens of GB at B = 512, T = 80) and must not sit NEXT to a cache of the same size
                torch.cuda.empty_cache()
                try:
                    with quiet_gc(), torch.cuda.graph(graph, capture_error_mode="thread_local"):
                        self._graph_step_body(dict(train_data), nstep_target, rnn_steps_train, rnn_bootstrap, burn_in_timesteps)
                finally:
                    gemm3.REFRESH_ALWAYS = False
                    self.value_log.log = real_log
                    st["losses"], self._defer_losses = self._defer_losses, None
                st["graph"] = graph
            st["graph"].replay()
            # a replay moved the data, not the version counters
            torch.autograd.graph.increment_version(list(self.policy.parameters()))
            if st["losses"]:
                for idx, losses in st["losses"]:
                    self._pre_update_losses()
                    self.history_buffer.update_losses(idx, losses)
            if st["logs"]:
                vals = torch.stack([v.detach().reshape(()).float() for _, v, _, _ in st["logs"]])      # one launch for all scalars
                for i, (key, _, a, k) in enumerate(st["logs"]):
                    self.value_log.log(key, vals[i], *a, **k)
            self._check_sweeps()
        self.value_log.log("batch_size", batch_size, group="train")
        self._update_steps_trained(batch_size)
        self.ts_learner_steps += 1
