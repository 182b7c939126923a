"""Data-parallel glue for one process per GPU (torch.distributed; backend
"nccl" = RCCL over xGMI on ROCm, "gloo" in the CPU tests and in the 2-ranks-on-
one-GPU test).

The replay shards by environment: rank r owns envs [r*E/R, (r+1)*E/R), their
rings, their priority tree and free list, ingests only those transitions,
samples its share of the batch locally and gathers locally — no frame ever
crosses xGMI (the reference has no multi-GPU path at all).
Exchanged per learner step:

  1. gradients: ONE flat fp32 buffer that the parameters' .grad tensors are views
     of (no pack / unpack copies), all-reduce(AVG) between backward and the
     clip / Adam step (the hook TorchTrainer._reduce_gradients calls).  8.1 M
     parameters = 32.5 MB.  With overlap (default at world > 1) the buffer is cut into
     buckets in backward order — head, recurrent layer, conv stack — and each bucket's
     all-reduce is issued on a side stream as soon as autograd has finished its last
     gradient, so only the conv bucket (0.3 MB) is exposed after the backward;
  2. 3 doubles per rank — (sum of priorities, active sequences, local max raw
     weight) — exchanged (all-reduce of a zero-padded (R, 3) buffer, so the same
     code serves RCCL and gloo) to turn shard-local importance weights into the
     weights ONE tree over the union of the shards would have produced,
     including the batch-max normalisation
     (prioritized_replay_history.py:327,353-354);
  3. one host int per loop iteration over a gloo side group — "my shard could
     form a batch" — so that ranks only enter the collectives together
     (lock-step guard; no device synchronisation involved).
Target-network sync, learning-rate and epsilon schedules are replicated
deterministically (same acted-step counters on every rank), no communication.
"""
import copy
import os

import torch
import torch.distributed as dist


STEP_FIELDS = ("total_steps", "early_stop_steps", "warmup_steps", "target_update_freq", "log_freq",
               "actor_update_frequency_steps")
# defaults of the step-denominated trainer arguments (reference signatures: policy_trainer.py:284-286 log_freq=10000,
# multi_step_trainer.py:152-156 actor_update_frequency_steps=1000; warmup_steps / target_update_freq default to 0 = off)
STEP_DEFAULTS = {"log_freq": 10000, "actor_update_frequency_steps": 1000}


def shard_config(config, rank, world, scaling="strong"):
    """Per-rank view of a whole-job config.  Envs and replay capacity are always
    split evenly (rank r owns envs [r*E/R, (r+1)*E/R)).  scaling="strong": the
    configured mbatch_size is the GLOBAL batch and every rank trains B/R
    sequences; "weak": every rank keeps the
    configured mbatch_size, envs and replay size (the job grows with R).

    Step-denominated settings (total_steps, early_stop_steps, warmup_steps,
    target_update_freq, log_freq, actor_update_frequency_steps) are compared against
    a rank's OWN acted-step counter.  strong: the configured values count whole-job
    acted steps, every rank acts 1/R of them, so they are divided by R (rounded up) —
    an R-GPU strong run acts `total_steps` transitions in total, syncs the target
    network every `target_update_freq` GLOBAL steps and anneals eps / lr / beta over
    the same global horizon as the 1-GPU run.  weak: they stay per-rank values (the
    job acts R x total_steps transitions)."""
    assert scaling in ("strong", "weak")
    cfg = copy.deepcopy(config)
    acting = cfg.setdefault("acting", {})
    envs = acting.get("actor_envs", 1)
    targs = cfg["training"]["args"]
    hm = targs.setdefault("history_mode", {})
    hm.setdefault("args", {})
    if scaling == "strong":
        assert envs % world == 0, "actor_envs must divide by the number of ranks"
        per = envs // world
        acting["total_envs"] = envs
        if "size" in hm["args"]:
            hm["args"]["size"] = hm["args"]["size"] // world
        mb = targs.get("mbatch_size")
        if mb:
            assert mb % world == 0, "mbatch_size must divide by the number of ranks"
            targs["mbatch_size"] = mb // world
        # the trainers' own defaults are whole-job values too: a config that omits a key must behave like one
        # that spells the default out (training/policy_trainer.py train(), multi_step_trainer.py _train())
        for key, default in STEP_DEFAULTS.items():
            targs.setdefault(key, default)
        for key in STEP_FIELDS:
            if targs.get(key):
                targs[key] = max(1, -(-int(targs[key]) // world))
        if rank == 0:
            import logging
            logging.getLogger(__name__).info("strong scaling over %d ranks: per-rank %s", world,
                                             {k: targs.get(k) for k in STEP_FIELDS + ("mbatch_size",)})
    else:
        per = envs
        acting["total_envs"] = envs * world
    acting["actor_envs"] = per
    acting["env_base"] = rank * per
    return cfg


class DataParallel:
    def __init__(self, group=None, host_group=None, force=None):
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self._flat = None
        self._params = None
        # run the collectives even at world 1 (exercises the RCCL path on one GPU)
        self.force = bool(os.environ.get("BENCH_FORCE_DIST")) if force is None else force
        self.host_group = host_group
        self._backend = dist.get_backend(group)
        self.timing = None      # a list: (start, end) HIP event pairs around the EXPOSED part of every gradient all-reduce (bench.py)
        # bucketed all-reduce overlapped with the backward pass (MIRL_DP_OVERLAP=0: one blocking collective)
        self.overlap = os.environ.get("MIRL_DP_OVERLAP", "1") != "0"
        self.buckets_overlapped = 0
        self._buckets, self._bucket_of, self._hooks = [], {}, []

    @property
    def active(self):
        return self.world > 1 or self.force

    # -- transport ------------------------------------------------------------------
    def _all_reduce(self, t, op):
        """RCCL reduces device tensors in place.  gloo (CPU tests, two ranks on one
        GPU) goes through a host copy: its device-tensor path stages through pinned
        memory on its own streams, which is not robust with two processes time-slicing
        one GPU."""
        if self._backend != "nccl" and t.is_cuda:
            host = t.cpu()
            dist.all_reduce(host, op=op, group=self.group)
            t.copy_(host)
        else:
            dist.all_reduce(t, op=op, group=self.group)

    # -- parameters / gradients ---------------------------------------------------
    def broadcast_parameters(self, module, src=0):
        """Identical initial weights (and buffers) on every rank."""
        if not self.active:
            return
        with torch.no_grad():
            for t in list(module.parameters()) + list(module.buffers()):
                # through the tensor itself, not `.data`: the version counter must move — the weight-derived caches
                # (gemm3.joint_rows / weight_planes, the actor's selection stamp) are keyed on it
                if self._backend != "nccl" and t.is_cuda:
                    host = t.detach().cpu()
                    dist.broadcast(host, src, group=self.group)
                    t.copy_(host)
                else:
                    got = t.detach().clone()
                    dist.broadcast(got, src, group=self.group)
                    t.copy_(got)

    def attach(self, module, buckets="auto"):
        """Make every parameter's .grad a view of one flat buffer.  Autograd then
        accumulates straight into it and the all-reduce needs no copies;
        zero_grad() must zero the buffer (TorchTrainer does) instead of dropping
        the .grad tensors.

        buckets: the buffer is laid out in BACKWARD order (last child module first) and
        cut into contiguous buckets; a bucket's all-reduce is issued from autograd's
        post-accumulate hook of its last gradient, asynchronously on the process group's
        stream, and all_reduce_gradients() only waits for what is still in flight.
        "auto": one bucket per top-level stage of a policy — everything outside
        `module.model` (heads, quantile layer) + the model's last layers, the recurrent
        layer, the conv stack — i.e. head -> LSTM -> conv, the order their gradients
        complete in; modules without a `.model.layers` get one bucket.  None / 1: the
        single blocking all-reduce of rounds 1-3."""
        params = [p for p in module.parameters() if p.requires_grad]
        groups = self._bucket_groups(module, params) if buckets == "auto" else None
        if not groups or buckets in (None, 1) or not self.overlap:
            groups = [params]
        n = sum(p.numel() for p in params)
        flat = torch.zeros(n, dtype=params[0].dtype, device=params[0].device)
        at = 0
        for hook in self._hooks:
            hook.remove()                      # a second attach() must not leave the first one's hooks firing twice
        self._buckets, self._bucket_of, self._hooks = [], {}, []
        for gi, group in enumerate(groups):
            lo = at
            for p in group:
                seg = flat[at:at + p.numel()]
                if p.dim() == 4 and not p.is_contiguous() and p.is_contiguous(memory_format=torch.channels_last):
                    # an NHWC conv weight: the gradient gets the parameter's own element order (autograd's layout contract;
                    # the two-launch optimizer tail walks parameter and gradient as the same flat range, csrc/optim.hip)
                    p.grad = seg.view(p.shape[0], p.shape[2], p.shape[3], p.shape[1]).permute(0, 3, 1, 2)
                else:
                    p.grad = seg.view_as(p)
                at += p.numel()
                self._bucket_of[id(p)] = gi
            self._buckets.append({"lo": lo, "hi": at, "count": len(group), "left": len(group), "work": None, "done": False})
        assert at == n
        self._flat, self._params = flat, [p for g in groups for p in g]
        if len(groups) > 1:
            for p in self._params:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))
        return flat

    @staticmethod
    def _bucket_groups(module, params):
        model = getattr(module, "model", None)
        layers = getattr(model, "layers", None)
        if layers is None or len(layers) < 2:
            return None
        want = {id(p) for p in params}
        in_layer = {}
        for i, layer in enumerate(layers):
            for p in layer.parameters():
                in_layer[id(p)] = i
        rec = [i for i, layer in enumerate(layers) if getattr(layer, "is_recurrent", lambda: False)()]
        first_rec = rec[0] if rec else len(layers) - 1
        head, mid, front = [], [], []
        for p in module.parameters():
            if id(p) not in want:
                continue
            i = in_layer.get(id(p))
            (head if (i is None or i > first_rec) else mid if i == first_rec else front).append(p)
        return [g for g in (head, mid, front) if g]

    def zero_grad(self):
        self._flat.zero_()
        for b in self._buckets:
            b["left"], b["work"], b["done"] = b["count"], None, False

    def _reduce_bucket(self, b, overlap):
        view = self._flat[b["lo"]:b["hi"]]
        if self._backend == "nccl":
            # async: the collective runs on the process group's stream behind an event of the current one;
            # nobody waits here — all_reduce_gradients() does, when the optimizer needs the result
            b["work"] = dist.all_reduce(view, op=dist.ReduceOp.AVG, group=self.group, async_op=overlap)
        else:
            self._all_reduce(view, dist.ReduceOp.SUM)
            view.div_(self.world)
        b["done"] = True
        self.buckets_overlapped += 1 if overlap else 0

    def _on_grad(self, p):
        if not self.active:
            return
        b = self._buckets[self._bucket_of[id(p)]]
        b["left"] -= 1
        if b["done"] or b["left"] < 0:
            # a second backward before zero_grad(): the bucket already went out, these gradients would never be reduced
            raise RuntimeError("DataParallel: a gradient arrived for a bucket that was already reduced "
                               "(one backward per zero_grad(); call zero_grad() before the next backward)")
        # buckets go out strictly in index order (the same order on every rank): a bucket whose gradients complete
        # before an earlier bucket's (a parameter unused this step) waits for all_reduce_gradients()
        while True:
            nxt = next((x for x in self._buckets if not x["done"]), None)
            if nxt is None or nxt["left"] > 0:
                break
            self._reduce_bucket(nxt, overlap=True)

    def all_reduce_gradients(self, module=None):
        if not self.active:
            return
        if self._flat is None:
            raise RuntimeError("DataParallel.attach(module) must run before the first backward")
        p = self._params[-1]            # a set_to_none zero_grad elsewhere would silently detach the views
        assert p.grad is not None and p.grad.untyped_storage().data_ptr() == self._flat.untyped_storage().data_ptr(), \
            "parameter .grad no longer aliases the gradient bucket"
        ev = None
        if self.timing is not None and self._flat.is_cuda:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        for b in self._buckets:                     # whatever the hooks did not issue (in order), then the waits
            if not b["done"]:
                self._reduce_bucket(b, overlap=False)
        for b in self._buckets:
            if b["work"] is not None:
                b["work"].wait()                    # RCCL: the CURRENT STREAM waits for the collective; the host does not
                b["work"] = None
        if ev is not None:
            ev[1].record()
            self.timing.append(ev)

    def ranks_seen(self, device):
        """How many ranks actually take part in a collective on this group: the
        all-reduced sum of a one per rank (bench.py's `rccl.ranks_seen`)."""
        one = torch.ones(1, dtype=torch.int32, device=device)
        self._all_reduce(one, dist.ReduceOp.SUM)
        return int(one.item())

    @property
    def bucket_bytes(self):
        return 0 if self._flat is None else self._flat.numel() * self._flat.element_size()

    # -- importance weights ---------------------------------------------------------
    def exchange_rows(self, mine):
        """mine: 1-D float64 device tensor -> (R, len) tensor holding every rank's
        row (all-reduce of a zero-padded buffer: supported for device tensors by
        both RCCL and gloo, unlike all_gather)."""
        buf = torch.zeros((self.world, mine.numel()), dtype=mine.dtype, device=mine.device)
        buf[self.rank] = mine
        if self.world > 1 or self.force:
            self._all_reduce(buf, dist.ReduceOp.SUM)
        return buf

    def globalize_weights(self, weights, p_sum, n_active, max_raw, beta):
        """weights: shard-normalised importance weights (any shape); p_sum,
        max_raw: 0-dim tensors from mirl_replay_sample's `stats`; n_active: int.
        Returns weights normalised as if all shards were one tree:
            w_global_i = (p_i * N_g / P_g)^-beta / max_j(...)
        using w_local_i * max_raw = (p_i * N_l / P_l)^-beta."""
        if not self.active:
            return weights
        mine = torch.stack([p_sum.double().reshape(()),
                            torch.as_tensor(float(n_active), dtype=torch.float64, device=p_sum.device),
                            max_raw.double().reshape(())])
        allr = self.exchange_rows(mine)                            # (R, 3)
        P_g, N_g = allr[:, 0].sum(), allr[:, 1].sum()
        k = ((N_g * allr[:, 0]) / (allr[:, 1] * P_g)) ** (-beta)   # raw_global = raw_local * k_r
        top = (allr[:, 2] * k).max()
        scale = (allr[self.rank, 2] * k[self.rank] / top)
        return weights * scale.to(weights.dtype)

    # -- host-side helpers -----------------------------------------------------------
    def _host(self):
        return self.host_group if self.host_group is not None else self.group

    def barrier(self):
        if self.world > 1:
            dist.barrier(group=self._host())

    def broadcast_object(self, obj, src=0):
        """A small picklable object from rank `src` to everyone (run directory names)."""
        if self.world == 1:
            return obj
        box = [obj]
        dist.broadcast_object_list(box, src=src, group=self._host())
        return box[0]

    # -- lock-step guard -------------------------------------------------------------
    def all_ready(self, ready):
        """True when EVERY rank reports ready.  A host-side exchange (gloo side
        group, CPU tensor): whether a shard can form a batch is decided by host
        bookkeeping, so no device synchronisation is involved."""
        if not self.active or self.world == 1:
            return bool(ready)
        flag = torch.tensor([1 if ready else 0], dtype=torch.int32)
        group = self.host_group if self.host_group is not None else self.group
        if self.host_group is None and self._backend == "nccl":
            flag = flag.cuda()
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        return bool(flag.item())


def init_from_env(backend=None, device=None):
    """Process-group setup for a torchrun launch (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_* in the environment).  Returns (rank, world, local_rank, DataParallel
    or None).  backend None = "nccl" (RCCL); a gloo side group carries the
    host-side lock-step flag."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    force = bool(os.environ.get("BENCH_FORCE_DIST"))
    if world == 1 and not force:
        return rank, world, local, None
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    backend = backend or "nccl"
    if not dist.is_initialized():
        kw = {}
        if backend == "nccl":
            kw["device_id"] = device if device is not None else torch.device("cuda", local)
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    host_group = dist.new_group(backend="gloo") if (backend == "nccl" and world > 1) else None
    return rank, world, local, DataParallel(host_group=host_group, force=force)
