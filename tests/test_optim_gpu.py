"""csrc/optim.hip: global gradient norm -> clip -> Adam in two launches against clip_grad_norm_ + torch.optim.Adam
(rltime/training/torch_trainer.py:177-199)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _params(gen, shapes):
    ps = []
    for s in shapes:
        t = torch.randn(*s, device="cuda", generator=gen) * 0.1
        if len(s) == 4:
            t = t.contiguous(memory_format=torch.channels_last)
        ps.append(torch.nn.Parameter(t))
    return ps


SHAPES = [(32, 4, 8, 8), (32,), (64, 32, 4, 4), (64,), (512, 3136), (512,), (6, 512), (6,), (1, 517), (4099,), (3, 4096 + 5)]


@pytest.mark.parametrize("clip,lr_tensor,many", [(40.0, False, False), (0.05, False, False), (None, False, False), (0.5, True, False), (0.5, False, True)])
def test_clip_and_adam_in_two_launches_follow_torch(clip, lr_tensor, many):
    from rltime_amd.models.torch.optim import ClipAdam
    gen = torch.Generator(device="cuda").manual_seed(5)
    shapes = SHAPES + ([(7,), (33, 3)] * 15 if many else [])                # > 32 tensors: two launches per pass
    mine = _params(gen, shapes)
    ref = [torch.nn.Parameter(p.detach().clone(memory_format=torch.preserve_format)) for p in mine]
    lr = 2.5e-4
    opt = ClipAdam(mine, lr=torch.tensor(lr, device="cuda") if lr_tensor else lr, eps=1.5e-4)
    opt_ref = torch.optim.Adam(ref, lr=lr, eps=1.5e-4)
    for step in range(6):
        for a, b in zip(mine, ref):
            g = torch.randn(a.shape, device="cuda", generator=gen) * (0.3 if step % 2 else 0.003)
            if a.dim() == 4:
                g = g.contiguous(memory_format=torch.channels_last)
            a.grad, b.grad = g.clone(memory_format=torch.preserve_format), g.clone(memory_format=torch.preserve_format)
            if a.shape == (1, 517):                                         # a row of a wider matrix: same element order
                a.grad = torch.cat([a.grad, a.grad], 1)[:, :517]
        if step == 2:                                                       # a parameter that sits a step out keeps its own counter
            mine[1].grad = ref[1].grad = None
        assert opt.fused_step_ok()
        before = [a._version for a in mine]
        norms = opt.step_clipped(clip)
        # version-keyed caches of derived weights see the update (and only the tensors that were updated move)
        assert all((a._version > v) == (a.grad is not None) for a, v in zip(mine, before))
        want_norm = torch.linalg.vector_norm(torch.stack([b.grad.norm() for b in ref if b.grad is not None]))
        if clip is not None:
            coef = torch.clamp(clip / (want_norm + 1e-6), max=1.0)
            for b in ref:
                if b.grad is not None:
                    b.grad.mul_(coef)
        opt_ref.step()
        assert abs(float(norms[0]) - float(want_norm)) <= 2e-6 * float(want_norm)
        if clip is not None:
            assert abs(float(norms[1]) - float(want_norm * coef)) <= 2e-6 * float(want_norm)
        for i, (a, b) in enumerate(zip(mine, ref)):
            # one rounding of the update (|lr| per element at most) on top of the parameter's own
            assert float((a - b).abs().max()) <= 1e-6 * float(b.abs().max()) + 2e-7 * (step + 1), (step, i)
            if b.grad is not None:
                assert float((a.grad - b.grad).abs().max()) <= 1e-6 * float(b.grad.abs().max()) + 1e-12, (step, i)
            sa, sb = opt.state[a], opt_ref.state[b]
            assert float(sa["step"]) == float(sb["step"]) == step + 1 - (1 if (i == 1 and step >= 2) else 0)
            assert float((sa["exp_avg"] - sb["exp_avg"]).abs().max()) <= 2e-6 * float(sb["exp_avg"].abs().max())
            assert float((sa["exp_avg_sq"] - sb["exp_avg_sq"]).abs().max()) <= 2e-6 * float(sb["exp_avg_sq"].abs().max())


def test_clip_adam_is_a_torch_adam_for_checkpoints_and_schedules():
    """state_dict of ClipAdam loads into torch.optim.Adam and back (rltime_amd/training/resume.py stores it); a parameter
    without the kernels' layout contract makes fused_step_ok() refuse and the plain step() still works."""
    from rltime_amd.models.torch.optim import ClipAdam
    gen = torch.Generator(device="cuda").manual_seed(6)
    ps = _params(gen, SHAPES[:6])
    opt = ClipAdam(ps, lr=1e-3)
    for p in ps:
        p.grad = torch.randn(p.shape, device="cuda", generator=gen).contiguous(memory_format=torch.channels_last if p.dim() == 4 else torch.contiguous_format)
    opt.step_clipped(1.0)
    import copy
    sd = copy.deepcopy(opt.state_dict())                      # load_state_dict keeps same-device tensors by reference
    other = torch.optim.Adam([torch.nn.Parameter(p.detach().clone(memory_format=torch.preserve_format)) for p in ps], lr=1e-3)
    other.load_state_dict(sd)
    assert all(float(other.state[q]["step"]) == 1.0 for q in other.param_groups[0]["params"])
    back = ClipAdam([torch.nn.Parameter(p.detach().clone(memory_format=torch.preserve_format)) for p in ps], lr=1e-3)
    back.load_state_dict(copy.deepcopy(other.state_dict()))
    for q in back.param_groups[0]["params"]:
        assert back.state[q]["step"].is_cuda and back.state[q]["step"].dtype == torch.float32
    for q, p in zip(back.param_groups[0]["params"], ps):
        q.grad = p.grad.clone(memory_format=torch.preserve_format)
    assert back.fused_step_ok()
    back.step_clipped(1.0)
    assert all(float(back.state[q]["step"]) == 2.0 for q in back.param_groups[0]["params"])
    ps[0].grad = ps[0].grad.contiguous()                      # NCHW gradient for an NHWC weight: not the kernels' contract
    assert not opt.fused_step_ok()
    opt.step()                                                # torch's own path
    assert float(opt.state[ps[0]]["step"]) == 2.0
