"""Adam with the global-norm gradient clip in the same two launches (csrc/optim.hip).

reference: rltime/training/torch_trainer.py:177-199 — clip_grad_norm_ followed by torch.optim.Adam.step().  `ClipAdam`
IS a torch.optim.Adam (same state: step / exp_avg / exp_avg_sq per parameter, same state_dict, same param_groups and
lr handling); `step_clipped(clip)` runs norm -> clip -> update as k_adam_sqsum + k_adam_update instead of PyTorch's ~25-60
small launches, and falls back to those (the caller's own clip + `step()`) for anything the kernels do not take."""
import ctypes as C

import torch


def _lib():
    from rltime_amd import _lib as L
    return L


def _dense_like(a, b):
    """Same element order in memory (the stride of a size-1 dimension says nothing)."""
    return (a.dtype == torch.float32 and a.is_cuda and a.shape == b.shape
            and all(x == y for n, x, y in zip(a.shape, a.stride(), b.stride()) if n > 1))


def _packed(t):
    """numel() elements in numel() consecutive words, in whatever order the strides say."""
    return t.is_contiguous() or (t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last))


class ClipAdam(torch.optim.Adam):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        # capturable: the step counters live on the device (the kernels read and advance them there), so the same object
        # works eagerly and inside a captured graph; lr may be a float or a 0-dim device tensor
        super().__init__(params, lr=lr, betas=betas, eps=eps, capturable=True)
        self._ws = None

    def fused_step_ok(self):
        """One parameter group of dense float32 device tensors whose gradients share their layout."""
        return self.why_not_fused() is None

    def why_not_fused(self):
        if len(self.param_groups) != 1:
            return "more than one parameter group"
        g = self.param_groups[0]
        if g["amsgrad"] or g["weight_decay"] != 0 or g["maximize"] or g.get("differentiable") or g.get("decoupled_weight_decay"):
            return "an Adam variant the kernels do not implement"
        ps = [p for p in g["params"] if p.grad is not None]
        if not ps:
            return "no gradients"
        for i, p in enumerate(ps):
            if not (p.is_cuda and p.dtype == torch.float32 and p.numel() > 0 and _packed(p)):
                return "parameter %d %s %s %s is not a packed float32 device tensor" % (i, tuple(p.shape), p.stride(), p.dtype)
            if not _dense_like(p.grad, p):
                return "gradient %d %s %s does not share its parameter's layout %s" % (i, tuple(p.grad.shape), p.grad.stride(), p.stride())
            st = self.state.get(p)
            if st:       # (a state loaded from another optimizer's checkpoint: the caller then keeps torch's own step)
                if not (_dense_like(st["exp_avg"], p) and _dense_like(st["exp_avg_sq"], p) and torch.is_tensor(st["step"])
                        and st["step"].is_cuda and st["step"].dtype == torch.float32):
                    return "optimizer state %d does not share its parameter's layout / device" % i
        return None

    @torch.no_grad()
    def step_clipped(self, clip):
        """-> float32 tensor [norm, norm * coef] (coef = min(clip / (norm + 1e-6), 1); clip None: no scaling).
        The caller checked fused_step_ok()."""
        L = _lib()
        group = self.param_groups[0]
        params, grads, ms, vs, _mx, steps = [], [], [], [], [], []
        self._init_group(group, params, grads, ms, vs, _mx, steps)
        for p, m, v, s in zip(params, ms, vs, steps):
            assert _dense_like(m, p) and _dense_like(v, p) and s.is_cuda and s.dtype == torch.float32      # fused_step_ok()
        n = len(params)
        arr = lambda ts: (C.c_void_p * n)(*[t.data_ptr() for t in ts])                   # noqa: E731
        numel = (C.c_int64 * n)(*[p.numel() for p in params])
        dev = params[0].device
        if self._ws is None or self._ws[0] != tuple(numel) or self._ws[1].device != dev:
            need = C.c_int64()
            L.check(L.lib.mirl_adam_clip_workspace_bytes(n, numel, C.byref(need)), "mirl_adam_clip_workspace_bytes")
            self._ws = (tuple(numel), torch.empty(need.value, dtype=torch.uint8, device=dev))
        ws = self._ws[1]
        out = torch.empty(2, dtype=torch.float32, device=dev)
        lr = group["lr"]
        lr_dev = C.c_void_p(lr.data_ptr()) if torch.is_tensor(lr) and lr.is_cuda else None
        b1, b2 = group["betas"]
        L.check(L.lib.mirl_adam_clip_step(n, arr(params), arr(grads), arr(ms), arr(vs), arr(steps), numel,
                                          float(lr) if lr_dev is None else 0.0, lr_dev, float(b1), float(b2), float(group["eps"]),
                                          float(clip) if clip is not None else 0.0, C.c_void_p(ws.data_ptr()), ws.numel(),
                                          C.c_void_p(out.data_ptr()), C.c_void_p(torch.cuda.current_stream().cuda_stream)),
                "mirl_adam_clip_step")
        # the kernels wrote through raw pointers: move the version counters like the in-place tensor ops they replace
        # (the caches of derived weights — joint / permuted / packed copies, models/torch/gemm3.py — are keyed by them)
        torch.autograd.graph.increment_version(params + grads + ms + vs + steps)
        return out
