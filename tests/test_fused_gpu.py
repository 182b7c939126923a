"""GPU: the single-pass network glue (rltime_amd/models/torch/fused.py on
csrc/nnops.hip) against the plain PyTorch expressions it replaces — the ones the
reference spells out (rltime/models/torch/modules/cnn.py:47-49,
rltime/policies/torch/iqn.py:78-102, rltime/policies/torch/dqn.py:74-112) —
forward and backward, fp32, tolerance 1e-4 of the tensor's scale (north-star
bar; identical math, different summation order in the reductions)."""
import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _close(a, b, what, rtol=1e-4):
    scale = float(b.abs().max()) + 1e-12
    err = float((a - b).abs().max()) / scale
    assert err <= rtol, "%s: max deviation %.3e of scale" % (what, err)


@pytest.mark.parametrize("n,cin,hw,cout,k,s,need_x", [
    (64, 4, 84, 32, 8, 4, False), (33, 32, 20, 64, 4, 2, True), (17, 64, 9, 64, 3, 1, True), (5, 32, 20, 16, 4, 2, True)])
def test_conv_bias_relu_matches_relu_conv(n, cin, hw, cout, k, s, need_x):
    from rltime_amd.models.torch.fused import conv_bias_relu
    torch.manual_seed(n)
    conv = nn.Conv2d(cin, cout, k, s).cuda().to(memory_format=torch.channels_last)
    with torch.no_grad():
        conv.bias.uniform_(-0.3, 0.3)
    x = torch.randn(n, cin, hw, hw, device="cuda").contiguous(memory_format=torch.channels_last)
    up = None
    res = []
    for fn in (lambda t: conv_bias_relu(t, conv), lambda t: F.relu(conv(t))):
        conv.zero_grad(set_to_none=True)
        xi = x.clone().requires_grad_(need_x)
        y = fn(xi)
        if up is None:
            up = torch.randn_like(y)
        (y * up).sum().backward()
        res.append((y.detach(), xi.grad, conv.weight.grad.clone(), conv.bias.grad.clone()))
    _close(res[0][0], res[1][0], "y", rtol=1e-6)                   # forward: same conv, same add, same max
    assert res[0][0].is_contiguous(memory_format=torch.channels_last)
    if need_x:
        _close(res[0][1], res[1][1], "dx")
    _close(res[0][2], res[1][2], "dW")
    _close(res[0][3], res[1][3], "db")


@pytest.mark.parametrize("n", [1, 32, 100])
@pytest.mark.parametrize("cin,hw,k,s", [(32, 20, 4, 2), (64, 9, 3, 1)])
def test_no_grad_small_batches_of_conv_2_and_3_run_on_the_acting_kernels(n, cin, hw, k, s):
    """Under no_grad and below the implicit GEMM's work threshold (the generic actor graph of the non-recurrent policies)
    conv layers 2-3 run on csrc/actnet.hip's kernel — the NHWC weight as stored is its tap-major operand — and give the
    library's relu(conv(x)) to f32 rounding; with gradients enabled the stored-activation path is unchanged."""
    from rltime_amd import _lib
    from rltime_amd.models.torch.fused import conv_bias_relu
    torch.manual_seed(n + cin)
    conv = nn.Conv2d(cin, 64, k, s).cuda().to(memory_format=torch.channels_last)
    x = torch.randn(n, cin, hw, hw, device="cuda").contiguous(memory_format=torch.channels_last)
    _lib.check(_lib.lib.mirl_profile_reset())
    _lib.check(_lib.lib.mirl_profile_set(2))
    try:
        with torch.no_grad():
            y = conv_bias_relu(x, conv)
        torch.cuda.synchronize()
        ran = {r["name"]: r["calls"] for r in _lib.profile_table()}
    finally:
        _lib.check(_lib.lib.mirl_profile_set(0))
    assert any(name.startswith("k_act_conv") for name in ran), ran
    want = F.relu(F.conv2d(x.double(), conv.weight.double(), conv.bias.double(), s))
    assert y.shape == want.shape and y.is_contiguous(memory_format=torch.channels_last)
    assert float((y.double() - want).abs().max()) <= 2e-6 * float(want.abs().max())
    y2 = conv_bias_relu(x, conv)                                   # grad mode: MIOpen + the in-place bias / ReLU pass
    assert y2.requires_grad and float((y2.detach() - y).abs().max()) <= 2e-6 * float(want.abs().max())


def test_mask_and_bias_gradient_pass_is_deterministic_and_exact():
    from rltime_amd.models.torch.fused import relu_bwd_bias_rows
    g = torch.Generator(device="cuda").manual_seed(1)
    for rows, c in [(1, 4), (63, 32), (100003, 64), (40961, 512), (9000, 1024)]:
        y = torch.randn(rows, c, device="cuda", generator=g).clamp(min=0)
        dy = torch.randn(rows, c, device="cuda", generator=g)
        a, da = relu_bwd_bias_rows(dy, y, c)
        b, dbb = relu_bwd_bias_rows(dy, y, c)
        assert torch.equal(a, b) and torch.equal(da, dbb)          # no atomics: bit-identical reruns
        want = torch.ops.aten.threshold_backward(dy, y, 0.0)
        assert torch.equal(a, want)
        _close(da.double(), want.double().sum(0), "db rows=%d C=%d" % (rows, c), rtol=2e-6)


def test_cos_embedding_is_bit_exact():
    from rltime_amd.models.torch.fused import cos_embed
    g = torch.Generator(device="cuda").manual_seed(2)
    for rows, d in [(1, 4), (77, 8), (4096 * 32 + 5, 64)]:
        tau = torch.rand(rows, device="cuda", generator=g)
        freq = torch.arange(1, d + 1, dtype=torch.float32, device="cuda") * np.pi
        assert torch.equal(cos_embed(tau, freq), torch.cos(freq * tau.unsqueeze(1)))


@pytest.mark.parametrize("m,n,c,d", [(3, 4, 32, 8), (257, 32, 512, 64), (40, 7, 64, 16), (4100, 32, 512, 64)])
def test_quantile_product_matches_plain_expression(m, n, c, d):
    from rltime_amd.models.torch.fused import quantile_product
    torch.manual_seed(m)
    lin = nn.Linear(d, c).cuda()
    x = torch.randn(m, c, device="cuda")
    phi = torch.randn(m * n, d, device="cuda")
    up = torch.randn(m * n, c, device="cuda")
    res = []
    for fused in (True, False):
        lin.zero_grad(set_to_none=True)
        xi = x.clone().requires_grad_(True)
        if fused:
            out = quantile_product(xi, phi, lin.weight, lin.bias, n)
        else:
            emb = F.relu(F.linear(phi, lin.weight, lin.bias))
            out = (xi.unsqueeze(1) * emb.reshape(m, n, -1)).reshape(m * n, -1)
        (out * up).sum().backward()
        res.append((out.detach(), xi.grad, lin.weight.grad.clone(), lin.bias.grad.clone()))
    if m * n * c * d < (1 << 31):
        _close(res[0][0], res[1][0], "out", rtol=1e-5)
        for i, what in ((1, "dx"), (2, "dWq"), (3, "dbq")):
            _close(res[0][i], res[1][i], what)
        return
    # the largest case runs the embedding GEMM on the split-bf16 kernel with the product in its epilogue
    # (models/torch/gemm3.py): pre-activations differ from the library's in the last bits, so a unit within an ulp
    # of zero may take the other ReLU branch — its gradient rows then differ legitimately (a handful at most)
    _close(res[0][0], res[1][0], "out", rtol=2e-5)
    for i, what in ((1, "dx"), (2, "dWq"), (3, "dbq")):
        a, b = res[0][i], res[1][i]
        off = (a - b).abs() > 1e-4 * float(b.abs().max())
        bad = int(off.any(dim=-1).sum()) if off.dim() == 2 else int(off.sum())
        assert bad <= 8, (what, bad)


@pytest.mark.parametrize("rows,f,h1,hv,a,q", [(9, 16, 32, 32, 4, 1), (5000, 512, 512, 512, 6, 1), (300, 64, 32, 16, 5, 3)])
def test_dueling_tail_matches_separate_layers(rows, f, h1, hv, a, q):
    from rltime_amd.models.torch.fused import dueling_tail
    torch.manual_seed(rows)
    fc, out, vh, vl = (nn.Linear(f, h1).cuda(), nn.Linear(h1, a).cuda(), nn.Linear(f, hv).cuda(), nn.Linear(hv, q).cuda())
    mods = (fc, out, vh, vl)
    x = torch.randn(rows, f, device="cuda")
    ua, uv = torch.randn(rows, a, device="cuda"), torch.randn(rows, q, device="cuda")
    res = []
    for fused in (True, False):
        for mod in mods:
            mod.zero_grad(set_to_none=True)
        xi = x.clone().requires_grad_(True)
        if fused:
            adv, val = dueling_tail(xi, fc, out, vh, vl)
        else:
            adv, val = out(F.relu(fc(xi))), vl(F.relu(vh(xi)))
        ((adv * ua).sum() + (val * uv).sum()).backward()
        res.append([adv.detach(), val.detach(), xi.grad] + [p.grad.clone() for mod in mods for p in mod.parameters()])
    names = ["adv", "val", "dx", "fc.W", "fc.b", "out.W", "out.b", "vh.W", "vh.b", "vl.W", "vl.b"]
    for name, u, v in zip(names, res[0], res[1]):
        _close(u, v, name)


def test_iqn_policy_fused_head_equals_unfused_head():
    """IQNPolicy.predict with the fused tail (one GEMM for the last FC layer and
    the dueling value branch, single-pass quantile product) against the layer-by-
    layer path, same weights and the same tau stream: outputs, taus and every
    parameter gradient."""
    from rltime_amd.policies.iqn import IQNPolicy
    from rltime_amd.spaces import Box, Discrete
    model = {"type": "sequential", "args": {"layer_configs": [
        {"type": "cnn", "args": {"channels_last": True, "layers": [{"filters": 8, "kernel": 4, "stride": 2},
                                                                  {"filters": 16, "kernel": 3, "stride": 1}]}},
        {"type": "lstm", "args": {"num_units": 32}}, {"type": "fc", "args": {"fc_size": 64}}]}}
    torch.manual_seed(0)
    pol = IQNPolicy.create(model_config=model, observation_space=Box(0, 255, (4, 20, 20), np.uint8),
                           action_space=Discrete(5), cuda=True, dueling=True, embedding_dim=16, num_sampling_quantiles=8)
    T, B = 6, 7
    g = torch.Generator(device="cuda").manual_seed(3)
    state = {"x": torch.randint(0, 256, (T * B, 4, 20, 20), dtype=torch.uint8, device="cuda", generator=g),
             "layer0_state": {}, "layer2_state": {},
             "layer1_state": {"hx": torch.randn(T * B, 32, device="cuda", generator=g) * 0.1,
                              "cx": torch.randn(T * B, 32, device="cuda", generator=g) * 0.1,
                              "initials": (torch.rand(T * B, device="cuda", generator=g) < 0.1).float()}}
    up = torch.randn(T * B, 8, 5, device="cuda", generator=g)
    res = []
    for fused in (True, False):
        pol.fuse_dueling_tail = fused
        pol.zero_grad(set_to_none=True)
        torch.manual_seed(11)
        z, taus = pol.predict(state, T)
        (z * up).sum().backward()
        res.append((z.detach(), taus, {k: p.grad.clone() for k, p in pol.named_parameters()}))
    assert torch.equal(res[0][1], res[1][1])
    _close(res[0][0], res[1][0], "z", rtol=1e-5)
    for k in res[0][2]:
        _close(res[0][2][k], res[1][2][k], "grad " + k)
