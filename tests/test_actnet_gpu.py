"""GPU: the acting vector step's own network kernels (csrc/actnet.hip) against float64 restatements of the layers they
replace (reference: rltime/models/torch/modules/cnn.py:43-50, lstm.py:83-116, policies/torch/iqn.py:67-106,
dqn.py:50-87,132-148) — each within a small multiple of the float32 library kernels' own distance from float64."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(None)


def _st():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _bar(got, want64, lib32, floor=2e-6):
    """|got - f64| <= 4 x max|library f32 - f64| (or an absolute floor scaled by the output magnitude)."""
    scale = float(want64.abs().max())
    err = float((got.double() - want64).abs().max())
    ref = float((lib32.double() - want64).abs().max())
    assert err <= max(4 * ref, floor * max(scale, 1.0)), (err, ref, scale)


@pytest.mark.parametrize("layer,frames", [(2, 1), (2, 32), (2, 256), (2, 700), (3, 5), (3, 32), (3, 256), (3, 1100)])
def test_conv_layers(layer, frames):
    from rltime_amd._lib import lib, check
    torch.manual_seed(layer * 100 + frames)
    ci, k, s, hi = (32, 4, 2, 20) if layer == 2 else (64, 3, 1, 9)
    ho = (hi - k) // s + 1
    x = torch.rand(frames, ci, hi, hi, device="cuda") * 2 - 0.5
    w = torch.randn(64, ci, k, k, device="cuda") * 0.05
    b = torch.randn(64, device="cuda") * 0.1
    x_nhwc = x.permute(0, 2, 3, 1).contiguous()
    w_taps = w.permute(0, 2, 3, 1).contiguous().view(64, -1)
    pitch = ho * ho * 64 + (128 if layer == 3 else 0)            # layer 3 writes the head of wider rows
    y = torch.full((frames, pitch), -7.0, device="cuda")
    assert lib.mirl_act_conv_supported(layer, ci, 64, k, s, hi, hi) == 1
    check(lib.mirl_act_conv_fwd(layer, frames, hi, hi, _p(x_nhwc), _p(w_taps), _p(b), _p(y), pitch, _st()), "mirl_act_conv_fwd")
    want = F.relu(F.conv2d(x.double(), w.double(), b.double(), stride=s)).permute(0, 2, 3, 1).reshape(frames, -1)
    lib32 = F.relu(F.conv2d(x, w, b, stride=s)).permute(0, 2, 3, 1).reshape(frames, -1)
    _bar(y[:, :ho * ho * 64], want, lib32)
    assert torch.all(y[:, ho * ho * 64:] == -7.0)                 # nothing written past a frame's rows


def test_conv_shape_gate():
    from rltime_amd._lib import lib
    assert lib.mirl_act_conv_supported(2, 32, 64, 4, 2, 20, 20) == 1 and lib.mirl_act_conv_supported(3, 64, 64, 3, 1, 9, 9) == 1
    for bad in [(2, 64, 64, 4, 2, 20, 20), (2, 32, 32, 4, 2, 20, 20), (3, 64, 64, 3, 2, 9, 9), (1, 4, 32, 8, 4, 84, 84), (3, 64, 64, 3, 1, 2, 9)]:
        assert lib.mirl_act_conv_supported(*bad) == 0, bad


@pytest.mark.parametrize("E,H,Fin", [(16, 64, 3136), (32, 512, 3136), (7, 64, 48), (33, 64, 3136), (64, 512, 3136), (48, 128, 16), (32, 8, 8)])
def test_lstm_step(E, H, Fin):
    from rltime_amd._lib import lib, check
    torch.manual_seed(E + H)
    K = Fin + H
    xh = torch.randn(E, K + 4, device="cuda") * 0.3                # pitch wider than K
    w = torch.randn(4 * H, K, device="cuda") * (1.0 / np.sqrt(K))
    bias = torch.randn(4 * H, device="cuda") * 0.1
    c_in = torch.randn(E, H, device="cuda") * 0.5
    h = torch.empty(E, H, device="cuda")
    c = torch.empty(E, H, device="cuda")
    assert lib.mirl_act_lstm_supported(E, H, K) == 1
    need = C.c_int64()
    check(lib.mirl_act_lstm_workspace_bytes(E, H, K, C.byref(need)))
    ws = torch.zeros((need.value + 3) // 4, dtype=torch.int32, device="cuda")
    for _ in range(3):                                      # the launch restores its arrival counters: repeatable as is
        h.fill_(float("nan"))
        check(lib.mirl_act_lstm_fwd(E, H, K, _p(xh), K + 4, _p(w), _p(bias), _p(c_in), _p(h), _p(c), _p(ws), _st()), "mirl_act_lstm_fwd")

    def cell(dt):
        g = F.linear(xh[:, :K].to(dt), w.to(dt), bias.to(dt))
        i, f, gg, o = g.chunk(4, dim=1)
        cc = torch.sigmoid(f) * c_in.to(dt) + torch.sigmoid(i) * torch.tanh(gg)
        return torch.sigmoid(o) * torch.tanh(cc), cc
    h64, c64 = cell(torch.float64)
    h32, c32 = cell(torch.float32)
    _bar(h, h64, h32)
    _bar(c, c64, c32)
    assert lib.mirl_act_lstm_supported(65, H, K) == 0 and lib.mirl_act_lstm_supported(E, H, K + 8) == 0


def _head_ref(dt, h, taus, freq, wq, bq, wfc, bfc, wout, bout, N, A, has_val):
    x = h.to(dt).repeat_interleave(N, dim=0)
    if freq is not None:
        phi = torch.cos(freq.to(dt) * taus.to(dt).unsqueeze(1))
        x = F.relu(F.linear(phi, wq.to(dt), bq.to(dt))) * x
    hid = F.relu(F.linear(x, wfc.to(dt), bfc.to(dt)))
    out = F.linear(hid, wout.to(dt), bout.to(dt))
    adv = out[:, :A]
    q = adv
    if has_val:
        q = out[:, A:A + 1] + adv - adv.mean(1, keepdim=True)
    return q.reshape(-1, N, A).mean(1)


@pytest.mark.parametrize("E,N,H,D,HID,A,has_val", [
    (32, 32, 512, 64, 1024, 6, True), (32, 32, 512, 64, 512, 6, False), (16, 8, 64, 16, 128, 6, True), (16, 8, 64, 16, 64, 6, False),
    (32, 1, 64, 0, 128, 6, True), (256, 1, 512, 0, 1024, 18, True), (3, 5, 64, 32, 80, 4, False), (256, 32, 512, 64, 512, 6, False),
    (5, 7, 128, 48, 272, 18, True), (64, 32, 512, 64, 1024, 6, True), (16, 4, 1024, 16, 64, 30, True)])
def test_head(E, N, H, D, HID, A, has_val):
    """hidden layers + output shares + selection against the float64 head; quantile fractions handed in."""
    from rltime_amd._lib import lib, check
    torch.manual_seed(E * 7 + N + HID)
    dev = "cuda"
    NO = A + (1 if has_val else 0)
    assert lib.mirl_act_head_supported(E, N, H, D, HID, NO) == 1
    h = torch.randn(E, H, device=dev) * 0.5
    taus = torch.rand(E * N, device=dev) if D else None
    freq = (torch.arange(1, D + 1, device=dev, dtype=torch.float32) * np.pi).contiguous() if D else None
    wq = torch.randn(H, D, device=dev) * (1.0 / np.sqrt(D)) if D else None
    bq = torch.randn(H, device=dev) * 0.1 if D else None
    wfc = torch.randn(HID, H, device=dev) * (1.0 / np.sqrt(H))
    bfc = torch.randn(HID, device=dev) * 0.1
    wout = torch.randn(NO, HID, device=dev) * (1.0 / np.sqrt(HID))
    bout = torch.randn(NO, device=dev) * 0.1
    parts, pitch = C.c_int32(), C.c_int32()
    check(lib.mirl_act_head_parts(HID, NO, C.byref(parts), C.byref(pitch)))
    assert parts.value == (HID + 63) // 64 and pitch.value >= NO and pitch.value % 8 == 0
    part = torch.full((parts.value * E * N * pitch.value,), float("nan"), device=dev)
    step = torch.tensor([5], dtype=torch.int64, device=dev)
    x = h
    if D:
        x = torch.full((E * N, H), float("nan"), device=dev)
        check(lib.mirl_act_embed(E, N, H, D, _p(h), _p(freq), _p(taus), 99, _p(step), _p(wq), _p(bq), _p(x), None, _st()), "mirl_act_embed")
    check(lib.mirl_act_head_hidden(E * N, H, HID, NO, _p(x), _p(wfc), _p(bfc), _p(wout), _p(part), _st()), "mirl_act_head_hidden")
    actions = torch.full((E,), -1, dtype=torch.int32, device=dev)
    q = torch.empty(E, A, device=dev)
    check(lib.mirl_act_head_select(E, N, A, parts.value, pitch.value, _p(part), _p(bout), 1 if has_val else 0, None, None, 0.0, 99,
                                   _p(step), _p(actions), _p(q), _st()), "mirl_act_head_select")
    want = _head_ref(torch.float64, h, taus, freq, wq, bq, wfc, bfc, wout, bout, N, A, has_val)
    lib32 = _head_ref(torch.float32, h, taus, freq, wq, bq, wfc, bfc, wout, bout, N, A, has_val)
    _bar(q, want, lib32, floor=5e-6)
    top2 = want.topk(2, dim=1).values
    clear = (top2[:, 0] - top2[:, 1]) > 1e-4
    assert torch.equal(actions[clear].long(), want.argmax(1)[clear])
    assert int(actions.min()) >= 0 and int(actions.max()) < A


def test_head_draws_its_fractions_like_cos_embed_rng():
    """taus NULL: the fractions are the Philox draws of mirl_cos_embed_rng (same key: seed, step word, row), and the
    epsilon-greedy draws those of mirl_actor_head_rng."""
    from rltime_amd._lib import lib, check
    E, N, H, D, HID, A = 32, 32, 512, 64, 512, 6
    torch.manual_seed(3)
    dev = "cuda"
    h = torch.randn(E, H, device=dev) * 0.5
    freq = (torch.arange(1, D + 1, device=dev, dtype=torch.float32) * np.pi).contiguous()
    wq, bq = torch.randn(H, D, device=dev) * 0.125, torch.randn(H, device=dev) * 0.1
    wfc, bfc = torch.randn(HID, H, device=dev) * 0.044, torch.randn(HID, device=dev) * 0.1
    wout, bout = torch.randn(A, HID, device=dev) * 0.044, torch.randn(A, device=dev) * 0.1
    step = torch.tensor([11], dtype=torch.int64, device=dev)
    parts, pitch = C.c_int32(), C.c_int32()
    check(lib.mirl_act_head_parts(HID, A, C.byref(parts), C.byref(pitch)))
    part = torch.zeros(parts.value * E * N * pitch.value, device=dev)
    tau_out = torch.empty(E * N, device=dev)
    x = torch.empty(E * N, H, device=dev)
    check(lib.mirl_act_embed(E, N, H, D, _p(h), _p(freq), None, 1234, _p(step), _p(wq), _p(bq), _p(x), _p(tau_out), _st()))
    phi = torch.empty((E * N, D), device=dev)
    tau_ref = torch.empty(E * N, device=dev)
    check(lib.mirl_cos_embed_rng(E * N, D, 1234, _p(step), _p(freq), _p(phi), _p(tau_ref), _st()))
    assert torch.equal(tau_out, tau_ref)
    # the same product from the handed-in fractions: identical rows
    x2 = torch.empty_like(x)
    check(lib.mirl_act_embed(E, N, H, D, _p(h), _p(freq), _p(tau_ref), 1234, _p(step), _p(wq), _p(bq), _p(x2), None, _st()))
    assert torch.equal(x, x2)
    check(lib.mirl_act_head_hidden(E * N, H, HID, A, _p(x), _p(wfc), _p(bfc), _p(wout), _p(part), _st()))
    # epsilon-greedy: same draws as the library-path head on the same (seed, step)
    eps = torch.tensor(0.5, dtype=torch.float64, device=dev)
    acts = torch.empty(E, dtype=torch.int32, device=dev)
    q = torch.empty(E, A, device=dev)
    check(lib.mirl_act_head_select(E, N, A, parts.value, pitch.value, _p(part), _p(bout), 0, _p(eps), None, 0.0, 77, _p(step), _p(acts),
                                   _p(q), _st()))
    out = (part.view(parts.value, E * N, pitch.value).sum(0)[:, :A] + bout).contiguous()
    acts2 = torch.empty(E, dtype=torch.int32, device=dev)
    q2 = torch.empty(E, A, device=dev)
    check(lib.mirl_actor_head_rng(E, N, A, _p(out), A, None, 0, _p(eps), None, 0.0, 77, _p(step), _p(acts2), _p(q2), None, _st()))
    explored = acts != q.argmax(1).int()
    assert 0.2 < float(explored.float().mean()) < 0.65            # eps 0.5, one pick in six lands on the greedy action
    assert torch.equal(acts[explored], acts2[explored])
    assert torch.allclose(q, q2, rtol=1e-5, atol=1e-6)


def test_shape_gates_are_host_logic():
    from rltime_amd._lib import lib
    assert lib.mirl_act_head_supported(32, 32, 512, 64, 1024, 7) == 1
    for bad in [(32, 32, 500, 64, 1024, 7), (32, 32, 512, 80, 1024, 7), (32, 32, 512, 64, 1000, 7), (32, 32, 512, 64, 1024, 33),
                (0, 32, 512, 64, 1024, 7), (32, 32, 192, 64, 1024, 7)]:
        assert lib.mirl_act_head_supported(*bad) == 0, bad
    assert lib.mirl_act_lstm_supported(64, 512, 3648) == 1 and lib.mirl_act_lstm_supported(64, 508, 3648) == 0


def test_env_step_and_pre_step_in_one_launch():
    """mirl_synth_env_step_pre == mirl_synth_env_step followed by mirl_actor_pre on its outputs (csrc/acting.hip)."""
    from rltime_amd._lib import lib, check
    from rltime_amd.acting.synthetic_env import SyntheticAtariVecEnv
    E, H, A, Fq = 48, 64, 6, 128
    dev = "cuda"
    outs = []
    for fused in (False, True):
        torch.manual_seed(4)
        env = SyntheticAtariVecEnv(E, frame_shape=(4, 84, 84), n_actions=A, seed=9, done_prob=0.2)
        obs = torch.zeros((E, 4, 84, 84), dtype=torch.uint8, device=dev)
        rew, don = torch.zeros(E, device=dev), torch.zeros(E, dtype=torch.uint8, device=dev)
        h, c = torch.randn(E, H, device=dev), torch.randn(E, H, device=dev)
        actions = torch.randint(0, A, (E,), dtype=torch.int32, device=dev)
        xh = torch.zeros(E, Fq + H, device=dev)
        c_in, pack, init = torch.zeros(E, H, device=dev), torch.zeros(E, 2 * H, device=dev), torch.zeros(E, device=dev)
        r_out, d_out = torch.zeros(E, device=dev), torch.zeros(E, dtype=torch.uint8, device=dev)
        ep_r, ep_l = torch.zeros(E, device=dev), torch.zeros(E, dtype=torch.int32, device=dev)
        o_r, o_l = torch.zeros(E, device=dev), torch.zeros(E, dtype=torch.int32, device=dev)
        counts = torch.zeros(A, dtype=torch.int32, device=dev)
        rng = torch.zeros(1, dtype=torch.int64, device=dev)
        pre = (H, A, _p(actions), _p(h), _p(c), C.c_void_p(xh.data_ptr() + 4 * Fq), Fq + H, _p(c_in), _p(pack), _p(init), _p(r_out), _p(d_out),
               1, _p(ep_r), _p(ep_l), _p(o_r), _p(o_l), _p(counts), _p(rng), 0xFFFFFFFFFFFFFFFF, _st())
        for _ in range(5):
            if fused:
                check(lib.mirl_synth_env_step_pre(*env.step_into_args(obs, rew, don), *pre), "mirl_synth_env_step_pre")
                env.advance_host()
            else:
                env.step_into(obs, rew, don)
                check(lib.mirl_actor_pre(E, pre[0], pre[1], _p(rew), _p(don), *pre[2:]), "mirl_actor_pre")
        torch.cuda.synchronize()
        outs.append([t.clone() for t in (obs, rew, don, xh, c_in, pack, init, r_out, d_out, ep_r, ep_l, o_r, o_l, counts, rng)])
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    assert int(outs[0][-1].item()) == 5 and int(outs[0][2].sum()) > 0
