"""GPU: the fused LSTM sequence op (csrc/lstm.hip + lstm_seq.py) against the
plain torch time loop (the reference's LSTMCell recurrence with resets,
modules/lstm.py:83-116): outputs, final state and all gradients, fp32 1e-5."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("T,B,I,H", [(1, 3, 5, 4), (7, 5, 12, 8), (20, 33, 40, 64), (80, 16, 64, 512),
                                     (5, 32, 24, 64), (12, 64, 40, 128), (9, 256, 48, 512),
                                     (80, 512, 32, 512), (6, 48, 20, 256), (3, 80, 8, 128)])
@pytest.mark.parametrize("mode", ["gemm+cell", "persistent"])
def test_fused_lstm_matches_loop(T, B, I, H, mode, monkeypatch):
    """persistent = the one-launch-per-sweep kernel (csrc/lstm_seq.hip), incl. the config-D
    shape T=80, B=512, H=512 with mid-sequence resets (20 % of the initials set), a ragged
    row block (B=48, 80: 3 and 5 sixteen-row tiles) and H = 128 / 256."""
    from rltime_amd.models.torch import lstm_seq
    from rltime_amd.models.torch.modules import LSTM
    monkeypatch.setattr(lstm_seq, "_PERSISTENT", mode == "persistent")
    if mode == "persistent" and not lstm_seq.persistent_supported(T, B, H):
        pytest.skip("shape outside the persistent kernel (B % 16, H in {128, 256, 512}, T >= 2)")
    torch.manual_seed(T * 100 + B)
    a = LSTM((I,), H).cuda()
    b = LSTM((I,), H).cuda()
    b.load_state_dict(a.state_dict())
    a.fused, b.fused = True, False
    g = torch.Generator().manual_seed(1)
    x = torch.randn(T * B, I, generator=g).cuda()
    hx = torch.randn(T * B, H, generator=g).cuda()
    cx = torch.randn(T * B, H, generator=g).cuda()
    initials = (torch.rand(T * B, generator=g) < 0.2).float().cuda()
    up = torch.randn(T * B, H, generator=g).cuda()
    res = []
    for m in (a, b):
        xi = x.clone().requires_grad_(True)
        out = m(xi, hx=hx, cx=cx, initials=initials, timesteps=T)
        (out * up).sum().backward()
        res.append((out.detach(), m.last_state, xi.grad, {k: p.grad for k, p in m.named_parameters()}))
    tol = dict(rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(res[0][0].cpu(), res[1][0].cpu(), **tol)
    np.testing.assert_allclose(res[0][1][0].cpu(), res[1][1][0].cpu(), **tol)
    np.testing.assert_allclose(res[0][1][1].cpu(), res[1][1][1].cpu(), **tol)
    np.testing.assert_allclose(res[0][2].cpu(), res[1][2].cpu(), **tol)
    for k in res[0][3]:
        scale = float(res[1][3][k].abs().max()) + 1e-6
        np.testing.assert_allclose(res[0][3][k].cpu() / scale, res[1][3][k].cpu() / scale, rtol=1e-3, atol=2e-5, err_msg=k)


def test_fused_lstm_no_grad_matches():
    from rltime_amd.models.torch.modules import LSTM
    torch.manual_seed(0)
    a = LSTM((16,), 32).cuda()
    T, B = 11, 9
    x = torch.randn(T * B, 16).cuda()
    hx, cx = torch.randn(T * B, 32).cuda(), torch.randn(T * B, 32).cuda()
    ini = (torch.rand(T * B) < 0.3).float().cuda()
    with torch.no_grad():
        a.fused = True
        o1 = a(x, hx=hx, cx=cx, initials=ini, timesteps=T)
        s1 = a.last_state
        a.fused = False
        o2 = a(x, hx=hx, cx=cx, initials=ini, timesteps=T)
        s2 = a.last_state
    np.testing.assert_allclose(o1.cpu(), o2.cpu(), rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(s1[1].cpu(), s2[1].cpu(), rtol=1e-4, atol=2e-5)


def test_linear_relu_epilogue_matches_autograd():
    """fused.linear_relu (hipBLASLt ReLU epilogue + explicit backward) == relu(linear)."""
    import torch.nn.functional as F
    from rltime_amd.models.torch.fused import linear_relu
    g = torch.Generator().manual_seed(0)
    for rows, fin, fout in [(7, 5, 3), (4096, 64, 512), (20000, 512, 512)]:
        x = torch.randn(rows, fin, generator=g).cuda()
        w = (torch.randn(fout, fin, generator=g) * 0.1).cuda()
        b = torch.randn(fout, generator=g).cuda()
        up = torch.randn(rows, fout, generator=g).cuda()
        outs = []
        for fn in (linear_relu, lambda a, ww, bb: F.relu(F.linear(a, ww, bb))):
            xi, wi, bi = (t.clone().requires_grad_(True) for t in (x, w, b))
            y = fn(xi, wi, bi)
            (y * up).sum().backward()
            outs.append((y.detach(), xi.grad, wi.grad, bi.grad))
        for k, (a, c) in enumerate(zip(outs[0], outs[1])):
            scale = float(c.abs().max()) + 1e-6
            if k == 0 or rows * fin * fout < (1 << 31):
                np.testing.assert_allclose(a.cpu() / scale, c.cpu() / scale, rtol=1e-4, atol=1e-5)
                continue
            # the largest case runs the forward on the split-bf16 GEMM (models/torch/gemm3.py): its pre-activations
            # differ from the library's in the last bit, so a unit sitting within an ulp of zero may get the other
            # ReLU mask — that sample's row of dx / the unit's row of dW / element of db then differ legitimately
            off = ((a - c).abs() / scale) > (1e-5 + 1e-4 * (c.abs() / scale))
            bad = int(off.any(dim=-1).sum()) if off.dim() == 2 else int(off.sum())
            assert bad <= 4, (k, bad)


@pytest.mark.parametrize("shape", [
    (3, 4, 84, 84),        # LDS-tiled kernel, one tile per workgroup (few frames)
    (2400, 4, 84, 84),     # LDS-tiled kernel, one workgroup per frame walking its 7 tiles (last one partial)
    (5, 4, 16, 16),        # a single partial tile
    (9, 4, 36, 36),        # two tiles, second partial
    (2600, 4, 64, 64),     # exactly 4 full tiles per frame
    (5, 4, 7, 9), (2, 1, 20, 20), (7, 3, 6, 6), (6, 2, 20, 20), (1000, 4, 84, 84)])   # generic shapes / sizes
def test_fused_frame_conversion_is_exact(shape):
    """u8 NCHW -> f32 NHWC * (1/255) in one pass == x.float() * scale (cnn.py:44-45), bit for bit."""
    from rltime_amd.models.torch.modules import _frames_to_f32_nhwc
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randint(0, 256, shape, dtype=torch.uint8, generator=g).cuda()
    got = _frames_to_f32_nhwc(x, 1.0 / 255.0)
    want = x.float() * (1.0 / 255.0)
    assert got.shape == want.shape and got.is_contiguous(memory_format=torch.channels_last)
    assert torch.equal(got, want)



def test_prepared_input_feeds_the_same_values():
    """CNN.prepare_input converts a whole time-major (R, B, C, H, W) block once; row
    slices of it, flattened like the trainer does, must drive the model to exactly
    the outputs of the raw uint8 slices (same conv inputs, same kernels)."""
    from rltime_amd.models.torch.sequential import SequentialModel
    from rltime_amd.spaces import Box
    torch.manual_seed(0)
    layers = [{"type": "cnn", "args": {"channels_last": True, "layers": [
        {"filters": 8, "kernel": 4, "stride": 2}, {"filters": 8, "kernel": 3, "stride": 1}]}},
        {"type": "fc", "args": {"fc_size": 16}}]
    model = SequentialModel(Box(0, 255, (4, 20, 20), np.uint8), layers).cuda()
    R, B = 7, 5
    block = torch.randint(0, 256, (R, B, 4, 20, 20), dtype=torch.uint8, device="cuda")
    prep = model.layers[0].prepare_input(block)
    assert prep.shape == block.shape and prep.dtype == torch.float32
    assert torch.equal(prep, block.float() * (1.0 / 255.0))
    flat = lambda x: x.reshape((x.shape[0] * x.shape[1],) + tuple(x.shape[2:]))   # noqa: E731
    for lo, hi in ((0, 5), (2, 7), (3, 4)):
        raw = {"x": flat(block[lo:hi]), "layer0_state": {}, "layer1_state": {}}
        pre = dict(raw, x_prepared=flat(prep[lo:hi]))
        assert pre["x_prepared"].is_contiguous(memory_format=torch.channels_last)
        with torch.no_grad():
            a = model(raw, 1)["output"]
            b = model(pre, 1)["output"]
        assert torch.equal(a, b)


def _sweep_inputs(T, B, H, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    rnd = lambda *s: torch.randn(*s, device="cuda", generator=g)   # noqa: E731
    gx = rnd(T, B, 4 * H)
    w = (rnd(4 * H, H) * (1.5 / H ** 0.5)).contiguous()
    h0, c0 = rnd(B, H) * 0.5, rnd(B, H)
    keep = (torch.rand(T, B, device="cuda", generator=g) > 0.15).float()
    return gx, w, h0, c0, keep


@pytest.mark.parametrize("T,B,H", [(80, 512, 512), (40, 512, 512), (7, 1024, 512), (5, 16, 128), (33, 208, 256)])
@pytest.mark.parametrize("need_grad", [False, True])
def test_persistent_sweep_equals_the_per_step_path(T, B, H, need_grad, monkeypatch):
    """mirl_lstm_seq_fwd against the rocBLAS-GEMM + cell-kernel loop on the same projection:
    outputs, final state and — with need_grad — everything the backward pass reads
    (activated gates, c(t), the masked step inputs).  B = 1024 makes clusters loop over two
    row blocks (512 workgroups would not be co-resident)."""
    from rltime_amd.models.torch import lstm_seq
    gx, w, h0, c0, keep = _sweep_inputs(T, B, H, T + B + H)
    assert lstm_seq.persistent_supported(T, B, H)
    res = []
    for persistent in (True, False):
        monkeypatch.setattr(lstm_seq, "_PERSISTENT", persistent)
        gates = gx.clone()
        out, hm, cm, c_all, h_last, c_last = lstm_seq._forward_sweep(gates, w, h0, c0, keep, need_grad)
        res.append(dict(out=out, h_last=h_last.clone(), c_last=c_last.clone(),
                        gates=gates if need_grad else None, hm=hm if need_grad else None,
                        cm=cm if need_grad else None, c_all=c_all))
    for k, a in res[0].items():
        b = res[1][k]
        assert (a is None) == (b is None), k
        if a is not None:
            np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=2e-5, atol=3e-6, err_msg=k)


@pytest.mark.parametrize("T,B,H", [(40, 512, 512), (24, 256, 512), (24, 192, 256), (30, 64, 512)])
def test_persistent_sweep_under_uneven_load_and_back_to_back(T, B, H):
    """B = 512: one cluster per XCD; B = 256 / 192 / 64: 4 / 3 / 1 clusters, every cluster spread
    over several XCDs (cross-XCD hand-offs).  The inter-workgroup hand-off must not depend on timing or placement: 12 sweeps back to
    back (re-used workspace memory, poisoned with NaN first) while a second stream keeps the
    chip unevenly busy with GEMMs of varying size; every sweep must reproduce the first
    result bit for bit, and no workgroup may have timed out."""
    import ctypes as C
    from rltime_amd._lib import lib, check
    from rltime_amd.models.torch import lstm_seq
    gx, w, h0, c0, keep = _sweep_inputs(T, B, H, 7)
    side = torch.cuda.Stream()
    a = torch.randn(4096, 4096, device="cuda")
    junk = torch.full((8 << 20,), float("nan"), device="cuda")
    del junk                                      # the allocator hands these NaN bytes to the next workspaces
    first = None
    for rep in range(12):
        with torch.cuda.stream(side):
            for k in range(1 + rep % 4):
                n = 512 * (1 + (rep + k) % 7)
                _ = a[:n, :n] @ a[:n, :n]
        gates = gx.clone()
        out, _, _, _, h_last, c_last = lstm_seq._forward_sweep(gates, w, h0, c0, keep, False)
        got = torch.cat([out.reshape(-1), h_last.reshape(-1), c_last.reshape(-1)]).clone()
        if first is None:
            first = got
            assert torch.isfinite(first).all()
        else:
            assert torch.equal(got, first), "sweep %d differs" % rep
    # and the repeated result is the right one
    from rltime_amd.models.torch import lstm_seq as ls
    keep_flag = ls._PERSISTENT
    ls._PERSISTENT = False
    try:
        gates = gx.clone()
        out, _, _, _, h_last, c_last = ls._forward_sweep(gates, w, h0, c0, keep, False)
    finally:
        ls._PERSISTENT = keep_flag
    want = torch.cat([out.reshape(-1), h_last.reshape(-1), c_last.reshape(-1)])
    np.testing.assert_allclose(first.cpu().numpy(), want.cpu().numpy(), rtol=2e-5, atol=3e-6)
    torch.cuda.synchronize()
    st = C.c_int32(-1)
    check(lib.mirl_lstm_seq_status(C.byref(st)))
    assert st.value == 0


def test_two_sweeps_side_by_side_only_where_both_grids_fit():
    """The independent no-grad passes of a learner step go on two streams (multi_step_trainer.py _side_by_side); their
    persistent sweeps spin on peer workgroups, so the pair must fit the chip together.  B = 64 (narrow form: 128
    workgroups of ~34 KB LDS) does; B = 512 (one 135 KB workgroup per compute unit) does not, whatever
    `overlap_passes` says — and where the pair fits, two concurrent sweeps give the results of two sequential ones."""
    from types import SimpleNamespace
    from rltime_amd.models.torch import lstm_seq
    from rltime_amd.training.multi_step_trainer import MultiStepTrainer
    assert lstm_seq.two_sweeps_fit(64, 512) and lstm_seq.two_sweeps_fit(32, 128)
    assert not lstm_seq.two_sweeps_fit(512, 512)
    layer = SimpleNamespace(num_units=512, fused=True)
    pol = SimpleNamespace(is_cuda=lambda: True, model=SimpleNamespace(layers=[layer]))
    fake = SimpleNamespace(overlap_passes=True, policy=pol, target_policy=pol, _ov=None)
    assert MultiStepTrainer._passes_overlap(fake, 8192, 64)
    assert not MultiStepTrainer._passes_overlap(fake, 8192, 512)          # forced on, still refused
    assert MultiStepTrainer._passes_overlap(fake, 8192, None)             # no sequence pass involved
    T, B, H = 16, 64, 512
    a_in, b_in = _sweep_inputs(T, B, H, 3), _sweep_inputs(T, B, H, 4)
    seq = [lstm_seq._forward_sweep(x[0].clone(), *x[1:], False) for x in (a_in, b_in)]
    side = torch.cuda.Stream()
    for _ in range(6):
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            pa = lstm_seq._forward_sweep(a_in[0].clone(), *a_in[1:], False)
        pb = lstm_seq._forward_sweep(b_in[0].clone(), *b_in[1:], False)
        torch.cuda.current_stream().wait_stream(side)
        for got, want in ((pa, seq[0]), (pb, seq[1])):
            assert torch.equal(got[0], want[0]) and torch.equal(got[4], want[4]) and torch.equal(got[5], want[5])
    torch.cuda.synchronize()
    lstm_seq.check_status()


@pytest.mark.parametrize("T,B", [(80, 512), (9, 256), (5, 48), (33, 1024)])
def test_persistent_backward_sweep_equals_the_per_step_path(T, B, monkeypatch):
    """mirl_lstm_seq_bwd (one launch: cell backward in registers, the recurrent contraction split by
    column owner, 16 x 16 partial blocks summed in a fixed order) against mirl_lstm_cell_bwd + one
    recurrent GEMM per step on the same saved forward state: d loss / d pre-activation of every step.
    B = 1024: clusters loop over row blocks; B = 48: a ragged row block."""
    from rltime_amd.models.torch import lstm_seq
    H = 512
    gx, w, h0, c0, keep = _sweep_inputs(T, B, H, 3 * T + B)
    monkeypatch.setattr(lstm_seq, "_PERSISTENT", False)
    gates = gx.clone()
    out, hm, cm, c_all, _, _ = lstm_seq._forward_sweep(gates, w, h0, c0, keep, True)
    g = torch.Generator(device="cuda").manual_seed(T)
    d_out = torch.randn(T, B, H, device="cuda", generator=g)
    res = []
    monkeypatch.setattr(lstm_seq, "_BWD_PERSISTENT_MAX_B", 1 << 20)       # every batch size through the persistent kernel
    for persistent in (True, False):
        monkeypatch.setattr(lstm_seq, "_PERSISTENT", persistent)
        dg = gates.clone()
        lstm_seq._backward_sweep(dg, c_all, cm, d_out, keep, w)
        res.append(dg)
    scale = float(res[1].abs().max())
    np.testing.assert_allclose(res[0].cpu().numpy() / scale, res[1].cpu().numpy() / scale, rtol=1e-4, atol=2e-6)
    # fixed summation order: a second persistent run is bit-identical
    monkeypatch.setattr(lstm_seq, "_PERSISTENT", True)
    dg = gates.clone()
    lstm_seq._backward_sweep(dg, c_all, cm, d_out, keep, w)
    assert torch.equal(dg, res[0])


def test_nhwc_conv_output_is_consumed_in_memory_order():
    """LSTM._flat_input: an NHWC (channels_last) conv output enters the input projection as a VIEW in memory order with the
    columns of W_ih permuted to match, instead of the transposing copy the reference's x.view(-1, inp_size) implies
    (modules/lstm.py:60-66).  Same outputs, same gradients (incl. d W_ih back in the parameter's own column order) as the
    copying path, through the sequence op and through project_input."""
    from rltime_amd.models.torch.modules import LSTM
    torch.manual_seed(3)
    T, B, C, Hh, Ww, H = 5, 16, 64, 7, 7, 128
    layer = LSTM((C, Hh, Ww), H).cuda()
    x0 = torch.randn(T * B, C, Hh, Ww, device="cuda").contiguous(memory_format=torch.channels_last)
    hx, cx = torch.randn(T * B, H, device="cuda") * 0.3, torch.randn(T * B, H, device="cuda") * 0.3
    ini = (torch.rand(T * B, device="cuda") < 0.2).float()
    res = {}
    for mode in (True, False):
        layer.nhwc_input = mode
        layer.zero_grad()
        x = x0.clone().requires_grad_(True)
        out = layer(x, hx, cx, ini, T)
        proj = layer.project_input(x)
        (out.square().sum() + proj.sum()).backward()
        res[mode] = (out.detach(), proj.detach(), x.grad.clone(), layer.lstm_cell.weight_ih.grad.clone(), layer.lstm_cell.weight_hh.grad.clone())
    rows, w = layer._flat_input(x0)
    layer.nhwc_input = True
    rows2, w2 = layer._flat_input(x0)
    assert rows2.data_ptr() == x0.data_ptr() and rows.data_ptr() != x0.data_ptr()          # a view vs a copy
    assert torch.equal(rows2 @ w2.t(), rows2 @ w2.t()) and torch.allclose(rows @ w.t(), rows2 @ w2.t(), rtol=1e-4, atol=1e-4)
    for a, b in zip(res[True], res[False]):
        assert a.shape == b.shape
        assert float((a - b).abs().max()) <= 2e-5 * max(float(b.abs().max()), 1e-6)
