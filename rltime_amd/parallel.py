"""Data-parallel glue for one process per GPU (torch.distributed, backend
"nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests).

The replay shards by environment: rank r owns envs [r*E/R, (r+1)*E/R), their
rings, their priority tree and free list, ingests only those transitions,
samples B/R... sequences locally and gathers locally — no frame ever crosses
xGMI (the reference has no multi-GPU path at all).  Only
three things are exchanged per learner step:

  1. gradients: one flat fp32 bucket, all-reduce(SUM) / R, between backward and
     the clip/Adam step (the hook TorchTrainer._reduce_gradients calls);
  2. 3 doubles per rank — (sum of priorities, active sequences, local max raw
     weight) — all-gathered to turn shard-local importance weights into the
     weights one global tree would have produced, including the batch-max
     normalisation (prioritized_replay_history.py:327,353-354);
  3. optionally logged scalars.
Everything operates on tensors of whatever device the process group serves."""
import torch
import torch.distributed as dist


def shard_config(config, rank, world):
    """Split acting envs and replay capacity evenly over `world` ranks."""
    import copy
    cfg = copy.deepcopy(config)
    envs = cfg["acting"]["actor_envs"]
    assert envs % world == 0, "actor_envs must divide by the number of ranks"
    cfg["acting"]["actor_envs"] = envs // world
    cfg["acting"]["env_base"] = rank * (envs // world)
    hm = cfg["training"]["args"]["history_mode"]
    hm.setdefault("args", {})
    hm["args"]["size"] = hm["args"]["size"] // world
    return cfg


class DataParallel:
    def __init__(self, group=None):
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self._flat = None
        import os
        self.force = bool(os.environ.get("BENCH_FORCE_DIST"))   # run the collectives even at world 1

    def all_reduce_gradients(self, module):
        grads = [p.grad for p in module.parameters() if p.grad is not None]
        if not grads or (self.world == 1 and not self.force):
            return
        n = sum(g.numel() for g in grads)
        if self._flat is None or self._flat.numel() != n or self._flat.device != grads[0].device:
            self._flat = torch.empty(n, dtype=grads[0].dtype, device=grads[0].device)
        views = []
        at = 0
        for g in grads:
            views.append(self._flat[at:at + g.numel()].view_as(g))
            at += g.numel()
        torch._foreach_copy_(views, grads)
        dist.all_reduce(self._flat, op=dist.ReduceOp.SUM, group=self.group)
        self._flat.div_(self.world)
        torch._foreach_copy_(grads, views)

    def globalize_weights(self, weights, p_sum, n_active, max_raw, beta):
        """weights: shard-normalised importance weights (any shape); p_sum,
        max_raw: 0-dim tensors from mirl_replay_sample's `stats`; n_active: int.
        Returns weights normalised as if all shards were one tree:
            w_global_i = (p_i * N_g / P_g)^-beta / max_j(...)
        using w_local_i * max_raw = (p_i * N_l / P_l)^-beta."""
        if self.world == 1 and not self.force:
            return weights
        mine = torch.stack([p_sum.double().reshape(()),
                            torch.as_tensor(float(n_active), dtype=torch.float64, device=p_sum.device),
                            max_raw.double().reshape(())])
        allr = [torch.empty_like(mine) for _ in range(self.world)]
        dist.all_gather(allr, mine, group=self.group)
        allr = torch.stack(allr)                                   # (R, 3)
        P_g, N_g = allr[:, 0].sum(), allr[:, 1].sum()
        k = ((N_g * allr[:, 0]) / (allr[:, 1] * P_g)) ** (-beta)   # raw_global = raw_local * k_r
        top = (allr[:, 2] * k).max()
        scale = (allr[self.rank, 2] * k[self.rank] / top)
        return weights * scale.to(weights.dtype)
