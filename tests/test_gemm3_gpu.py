"""GPU: the wide layers' f32 GEMMs on the bf16 matrix pipe (csrc/gemm3.hip, mirl_gemm3) against the
float64 product and the library's f32 GEMM on the same inputs — the three products of the reference's
nn.Linear layers (rltime/policies/torch/dqn.py:50-112, iqn.py:82-102, models/torch/modules/lstm.py:60-81):
forward x W^T + b, data gradient g W, weight gradient g^T x.

Small-integer operands are exactly representable in one bf16 part and every partial sum is exact in
f32: indexing / tile mapping / split-K order are checked BIT-exactly.  Real operands: the result must be
an f32 GEMM — no further from the float64 product than twice the library f32 GEMM's own error, and
within 1e-5 of the largest output (the north-star tolerance is 1e-4)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

LAYOUTS = {"nt": 0, "nn": 1, "tn": 2}


def _operands(lay, M, N, K, gen, integer=False, pad_a=0, pad_b=0):
    def make(r, c, pad):
        full = (torch.randint(-8, 9, (r, c + pad), device="cuda", generator=gen).float() if integer
                else torch.randn(r, c + pad, device="cuda", generator=gen))
        return full[:, :c] if pad else full
    if lay == "nt":
        return make(M, K, pad_a), make(N, K, pad_b)
    if lay == "nn":
        return make(M, K, pad_a), make(K, N, pad_b)
    return make(K, M, pad_a), make(K, N, pad_b)


def _product(lay, a, b):
    if lay == "nt":
        return a @ b.t()
    if lay == "nn":
        return a @ b
    return a.t() @ b


SHAPES = [("nt", 256, 256, 16), ("nt", 1000, 300, 64), ("nt", 513, 1024, 512), ("nt", 4096, 2048, 3136),
          ("nn", 256, 256, 16), ("nn", 777, 260, 48), ("nn", 2048, 3136, 2048), ("nn", 5000, 512, 1024),
          ("tn", 256, 256, 128), ("tn", 300, 260, 4112), ("tn", 1024, 512, 40960), ("tn", 2048, 3136, 2064),
          ("nt", 1, 1, 16), ("tn", 4, 4, 128),
          # N <= 64: the NARROW wave tiling (eight waves along M, csrc/gemm3.hip) — the quantile layer's weight gradient
          ("tn", 512, 64, 40960), ("tn", 300, 60, 4112), ("tn", 700, 64, 2064), ("tn", 256, 8, 1040)]
NARROW_SHAPES = SHAPES[-4:]


@pytest.mark.parametrize("lay,M,N,K", SHAPES)
def test_integer_operands_are_bit_exact(lay, M, N, K):
    from rltime_amd.models.torch import gemm3
    gen = torch.Generator(device="cuda").manual_seed(M * 7 + N * 3 + K)
    a, b = _operands(lay, M, N, K, gen, integer=True)
    assert gemm3.supported(LAYOUTS[lay], a, b, min_work=0)
    want = _product(lay, a.double(), b.double())
    assert float(want.abs().max()) < 2 ** 24
    got = gemm3.gemm(LAYOUTS[lay], a, b)
    assert got.shape == (M, N)
    assert torch.equal(got.double(), want)


@pytest.mark.parametrize("lay,M,N,K", SHAPES[:12] + NARROW_SHAPES)
def test_real_operands_are_an_f32_gemm(lay, M, N, K):
    from rltime_amd.models.torch import gemm3
    gen = torch.Generator(device="cuda").manual_seed(M + N + K)
    a, b = _operands(lay, M, N, K, gen)
    a[::7] *= 37.0                     # rows of very different magnitude inside one tile
    b[::5] *= 0.013
    want = _product(lay, a.double(), b.double())
    scale = float(want.abs().max())
    got = gemm3.gemm(LAYOUTS[lay], a, b)
    lib = _product(lay, a, b)
    err3 = float((got.double() - want).abs().max()) / scale
    errl = float((lib.double() - want).abs().max()) / scale
    assert err3 <= 2.0 * errl + 1e-7, (err3, errl)
    assert err3 <= 1e-5


@pytest.mark.parametrize("M,N,K", [(8192, 1024, 512), (6200, 900, 80), (8192 + 77, 1024 + 4, 512), (300, 260, 48)])
def test_mid_tile_for_products_with_few_big_tiles(M, N, K, monkeypatch):
    """csrc/gemm3.hip k_gemm3_mid (256 x 128 tiles): taken by plain NT products whose 256 x 256 tiling would leave most
    compute units idle — the acting batch's joint hidden layer at 256 envs is the case it was written for.  Integer
    operands bit-exact (tile mapping, ragged edges in both directions, bias + ReLU epilogue); real operands an f32 GEMM;
    and the launch table shows which kernel ran, with the switch off the big tile."""
    import ctypes as C
    from rltime_amd import _lib
    from rltime_amd.models.torch import gemm3
    gen = torch.Generator(device="cuda").manual_seed(M + 3 * N + K)
    a, b = _operands("nt", M, N, K, gen, integer=True)
    bias = torch.randint(-4, 5, (N,), device="cuda", generator=gen).float()
    _lib.check(_lib.lib.mirl_profile_reset())
    _lib.check(_lib.lib.mirl_profile_set(2))
    got = gemm3.gemm(gemm3.NT, a, b, bias, relu=True)
    torch.cuda.synchronize()
    _lib.check(_lib.lib.mirl_profile_set(0))
    ran = {r["name"]: r["calls"] for r in _lib.profile_table()}
    assert ran.get("k_gemm3_nt_mid") == 1 and not ran.get("k_gemm3_nt"), ran
    want = torch.relu(a.double() @ b.double().t() + bias.double())
    assert torch.equal(got.double(), want)
    a, b = _operands("nt", M, N, K, gen)
    a[::7] *= 37.0
    b[::5] *= 0.013
    want = a.double() @ b.double().t()
    scale = float(want.abs().max())
    got = gemm3.gemm(gemm3.NT, a, b)
    lib = a @ b.t()
    err3 = float((got.double() - want).abs().max()) / scale
    errl = float((lib.double() - want).abs().max()) / scale
    assert err3 <= 2.0 * errl + 1e-7 and err3 <= 1e-5, (err3, errl)


def test_strided_operands_bias_and_relu():
    from rltime_amd.models.torch import gemm3
    gen = torch.Generator(device="cuda").manual_seed(5)
    # column-slice views: row pitch larger than the row length (the dueling head's two halves)
    a, b = _operands("nt", 1500, 520, 256, gen, pad_a=256, pad_b=64)
    assert a.stride(0) == 512 and b.stride(0) == 320
    bias = torch.randn(520, device="cuda", generator=gen)
    want = torch.relu(a.double() @ b.double().t() + bias.double())
    got = gemm3.gemm(0, a, b, bias=bias, relu=True)
    assert float((got.double() - want).abs().max()) <= 1e-5 * float(want.abs().max())
    assert float(got.min()) >= 0.0 and float((got == 0).float().mean()) > 0.3
    # into a column block of a wider output (ldc > N)
    wide = torch.full((1500, 1040), float("nan"), device="cuda")
    gemm3.gemm(0, a, b, bias=bias, relu=True, out=wide[:, 520:])
    assert torch.equal(wide[:, 520:], got) and bool(torch.isnan(wide[:, :520]).all())
    # k-strided operands with a pitch: weight gradient of a column block
    g, x = _operands("tn", 384, 512, 8192, gen, pad_a=128, pad_b=0)
    want = g.double().t() @ x.double()
    got = gemm3.gemm(2, g, x)
    assert float((got.double() - want).abs().max()) <= 1e-5 * float(want.abs().max())


def test_split_k_reduction_is_deterministic():
    from rltime_amd.models.torch import gemm3
    gen = torch.Generator(device="cuda").manual_seed(9)
    g, x = _operands("tn", 1024, 512, 65536, gen)
    first = gemm3.gemm(2, g, x).clone()
    for _ in range(3):
        assert torch.equal(gemm3.gemm(2, g, x), first)


def test_rejects_what_it_cannot_take():
    from rltime_amd.models.torch import gemm3
    from rltime_amd import _lib
    a = torch.randn(64, 24, device="cuda")                     # K % 16 != 0
    assert not gemm3.supported(0, a, torch.randn(32, 24, device="cuda"), min_work=0)
    assert not gemm3.supported(0, torch.randn(64, 32, device="cuda").double(), torch.randn(32, 32, device="cuda").double(), min_work=0)
    assert not gemm3.supported(0, torch.randn(64, 32, device="cuda")[:, ::2], torch.randn(32, 16, device="cuda"), min_work=0)
    assert _lib.lib.mirl_gemm3_supported(3, 64, 64, 64) == 0
    with pytest.raises(_lib.MirlError):
        gemm3.gemm(0, a, torch.randn(32, 24, device="cuda"))


def test_linear_autograd_matches_the_library(monkeypatch):
    from rltime_amd.models.torch import gemm3
    gen = torch.Generator(device="cuda").manual_seed(11)
    M, K, N = 8192, 3136, 512                                   # M*N*K above the library threshold
    x = torch.randn(M, K, device="cuda", generator=gen)
    w = (torch.randn(N, K, device="cuda", generator=gen) / K ** 0.5)
    b = torch.randn(N, device="cuda", generator=gen)
    go = torch.randn(M, N, device="cuda", generator=gen)
    res = {}
    for mode in ("gemm3", "library"):
        monkeypatch.setenv("MIRL_GEMM3", "1" if mode == "gemm3" else "0")
        xx, ww, bb = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
        y = gemm3.linear(xx, ww, bb)
        y.backward(go)
        res[mode] = (y.detach(), xx.grad, ww.grad, bb.grad)
    want = F.linear(x.double(), w.double(), b.double())
    assert float((res["gemm3"][0].double() - want).abs().max()) <= 1e-5 * float(want.abs().max())
    for got, lib in zip(res["gemm3"], res["library"]):
        assert float((got - lib).abs().max()) <= 2e-5 * float(lib.abs().max())


def test_dueling_tail_and_linear_relu_match_the_library_path(monkeypatch):
    from rltime_amd.models.torch import fused
    gen = torch.Generator(device="cuda").manual_seed(13)
    M, Fd, H, A = 16384, 512, 512, 6
    mk = lambda *s: torch.randn(*s, device="cuda", generator=gen)
    x = mk(M, Fd)
    params = [mk(H, Fd) / Fd ** 0.5, mk(H), mk(A, H) / H ** 0.5, mk(A), mk(H, Fd) / Fd ** 0.5, mk(H), mk(1, H) / H ** 0.5, mk(1)]
    ga, gv = mk(M, A), mk(M, 1)
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("MIRL_GEMM3", mode)
        xx = x.clone().requires_grad_(True)
        pp = [p.clone().requires_grad_(True) for p in params]
        a, v = fused._DuelingTail.apply(xx, *pp)
        torch.autograd.backward([a, v], [ga, gv])
        y = fused.linear_relu(xx.detach().requires_grad_(True), pp[0].detach().requires_grad_(True), pp[1].detach().requires_grad_(True))
        out[mode] = [a.detach(), v.detach(), xx.grad] + [p.grad for p in pp] + [y.detach()]
    names = ["a", "v", "dx", "dw1", "db1", "dwo", "dbo", "dwv", "dbv", "dwq", "dbq", "linear_relu"]
    for name, got, lib in zip(names, out["1"], out["0"]):
        assert got.shape == lib.shape
        fwd = name in ("a", "v", "linear_relu")
        # forward outputs: K = 512 products; gradients sum over 16 384 rows in different orders in the two paths
        tol = (2e-5 if fwd else 1e-4) * max(float(lib.abs().max()), 1e-6)
        off = (got - lib).abs() > tol
        if fwd:
            assert not bool(off.any()), name
        else:
            # a hidden unit whose pre-activation is within an ulp of zero can land on the other side of the ReLU
            # in the two paths (expected: a few of the 16.8 M): the sample's row of dx and the unit's row of
            # dW1 / dWv / element of db then differ legitimately — a handful of rows at most
            bad = int(off.any(dim=-1).sum()) if off.dim() == 2 else int(off.sum())
            assert bad <= 8, (name, bad)


@pytest.mark.parametrize("M,n", [(4096, 32), (3000, 8), (1500, 64), (777, 1)])
def test_quantile_product_in_the_epilogue(M, n):
    """relu(linear(phi)) * x[row // n] as one NT product with the multiply in its epilogue (iqn.py:82-102)."""
    from rltime_amd.models.torch import gemm3
    gen = torch.Generator(device="cuda").manual_seed(M + n)
    K, N = 64, 512
    x = torch.randn(M, N, device="cuda", generator=gen)
    phi = torch.cos(torch.rand(M * n, 1, device="cuda", generator=gen) * torch.arange(1, K + 1, device="cuda") * 3.14159265)
    w = torch.randn(N, K, device="cuda", generator=gen) / K ** 0.5
    b = torch.randn(N, device="cuda", generator=gen) * 0.1
    emb64 = torch.relu(phi.double() @ w.double().t() + b.double())
    want = emb64 * x.double().repeat_interleave(n, dim=0)
    for keep in (True, False):
        out, emb = gemm3.quantile_product(x, phi, w, b, n, keep)
        assert float((out.double() - want).abs().max()) <= 1e-5 * float(want.abs().max())
        if keep:
            assert float((emb.double() - emb64).abs().max()) <= 1e-5 * float(emb64.abs().max())
            assert torch.equal(out, emb * x.repeat_interleave(n, dim=0))       # the product itself is one f32 multiply
        else:
            assert emb is None


def test_quantile_product_autograd_with_and_without_the_epilogue(monkeypatch):
    from rltime_amd.models.torch import fused
    gen = torch.Generator(device="cuda").manual_seed(21)
    M, n, K, N = 4096, 32, 64, 512
    x = torch.randn(M, N, device="cuda", generator=gen)
    phi = torch.cos(torch.rand(M * n, 1, device="cuda", generator=gen) * torch.arange(1, K + 1, device="cuda") * 3.14159265)
    w = torch.randn(N, K, device="cuda", generator=gen) / K ** 0.5
    b = torch.randn(N, device="cuda", generator=gen) * 0.1
    go = torch.randn(M * n, N, device="cuda", generator=gen)
    res = {}
    for mode in (True, False):
        monkeypatch.setattr(fused, "_QP_EPILOGUE", mode)
        xx, ww, bb = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
        out = fused.quantile_product(xx, phi, ww, bb, n)
        out.backward(go)
        with torch.no_grad():
            nograd = fused.quantile_product(xx, phi, ww, bb, n)
        res[mode] = (out.detach(), nograd, xx.grad, ww.grad, bb.grad)
    for k, (got, lib) in enumerate(zip(res[True], res[False])):
        tol = (2e-5 if k < 2 else 1e-4) * float(lib.abs().max())
        off = (got - lib).abs() > tol
        if k < 2:
            assert not bool(off.any()), k
        else:                                                   # ReLU-threshold flips: a handful of rows at most
            bad = int(off.any(dim=-1).sum()) if off.dim() == 2 else int(off.sum())
            assert bad <= 8, (k, bad)


@pytest.mark.parametrize("M,N,K,O", [(256, 256, 16, 1), (1000, 512, 64, 7), (777, 1024, 512, 7), (4099, 300, 128, 8), (33, 68, 32, 3)])
def test_fused_following_layer_matches_two_products(M, N, K, O, monkeypatch):
    """mirl_gemm3_nt_head: hidden = relu(x W^T + b) and out = hidden W2^T + b2 (O <= 8 units) from ONE launch + a
    fixed-order reduction — the dueling head's output layers in the epilogue of the joint hidden layer
    (policies/torch/dqn.py:50-66,101-112).  Integer operands: hidden and out bit-exact; real operands: hidden
    bit-identical to mirl_gemm3's, out within 1e-5 of the float64 product of that same hidden activation; without the
    stored activation (the no-grad passes) the same outputs; ragged row / column tiles."""
    from rltime_amd.models.torch import gemm3
    monkeypatch.setattr(gemm3, "_MIN_WORK", 0)
    gen = torch.Generator(device="cuda").manual_seed(M + N + K + O)
    xi = torch.randint(-4, 5, (M, K), device="cuda", generator=gen).float()
    wi = torch.randint(-4, 5, (N, K), device="cuda", generator=gen).float()
    bi = torch.randint(-8, 9, (N,), device="cuda", generator=gen).float()
    w2i = torch.randint(-2, 3, (O, N), device="cuda", generator=gen).float()
    b2i = torch.randint(-3, 4, (O,), device="cuda", generator=gen).float()
    assert gemm3.head_supported(xi, wi, bi, w2i)
    hid, out = gemm3.linear_relu_head(xi, wi, bi, w2i, b2i, True)
    want_h = torch.relu(xi.double() @ wi.double().t() + bi.double())
    want_o = want_h @ w2i.double().t() + b2i.double()
    assert float(want_o.abs().max()) < 2 ** 24
    assert torch.equal(hid.double(), want_h) and torch.equal(out.double(), want_o)
    none, out2 = gemm3.linear_relu_head(xi, wi, bi, w2i, b2i, False)
    assert none is None and torch.equal(out2, out)
    # real operands
    x = torch.randn(M, K, device="cuda", generator=gen)
    w = torch.randn(N, K, device="cuda", generator=gen) / K ** 0.5
    b = torch.randn(N, device="cuda", generator=gen) * 0.1
    w2 = torch.randn(O, N, device="cuda", generator=gen) / N ** 0.5
    b2 = torch.randn(O, device="cuda", generator=gen) * 0.1
    hid, out = gemm3.linear_relu_head(x, w, b, w2, b2, True)
    assert torch.equal(hid, gemm3.gemm(gemm3.NT, x, w, b, relu=True))
    want = hid.double() @ w2.double().t() + b2.double()
    assert float((out.double() - want).abs().max()) <= 1e-5 * float(want.abs().max())
    _, out2 = gemm3.linear_relu_head(x, w, b, w2, b2, False)
    assert torch.equal(out2, out)
    a, c = gemm3.linear_relu_head(x, w, b, w2, b2, False)[1], gemm3.linear_relu_head(x, w, b, w2, b2, False)[1]
    assert torch.equal(a, c)                                     # fixed reduction order: bit-reproducible


def test_dueling_tail_with_and_without_the_fused_output_layers(monkeypatch):
    """fused._DuelingTail with its two output layers in the epilogue of the joint hidden product (gemm3._HEAD) against
    the same op with two library GEMMs over the stored activation: advantage / value outputs within 2e-5 of scale, every
    gradient bit-identical (the backward reads the same stored activation), and the no-grad form (nothing stored) gives
    the same outputs."""
    from rltime_amd.models.torch import fused, gemm3
    gen = torch.Generator(device="cuda").manual_seed(31)
    M, Fd, H, A = 8192 + 40, 512, 512, 6
    mk = lambda *s: torch.randn(*s, device="cuda", generator=gen)       # noqa: E731
    x = mk(M, Fd)
    params = [mk(H, Fd) / Fd ** 0.5, mk(H), mk(A, H) / H ** 0.5, mk(A), mk(H, Fd) / Fd ** 0.5, mk(H), mk(1, H) / H ** 0.5, mk(1)]
    ga, gv = mk(M, A), mk(M, 1)
    monkeypatch.setattr(gemm3, "_MIN_WORK", 0)
    out = {}
    for head in (True, False):
        monkeypatch.setattr(gemm3, "_HEAD", head)
        xx = x.clone().requires_grad_(True)
        pp = [p.clone().requires_grad_(True) for p in params]
        a, v = fused._DuelingTail.apply(xx, *pp)
        torch.autograd.backward([a, v], [ga, gv])
        with torch.no_grad():
            a0, v0 = fused._DuelingTail.apply(xx, *pp, False)
        out[head] = [a.detach().clone(), v.detach().clone(), a0.clone(), v0.clone(), xx.grad] + [p.grad for p in pp]
    for i, (got, ref) in enumerate(zip(out[True], out[False])):
        if i < 4:
            assert float((got - ref).abs().max()) <= 2e-5 * float(ref.abs().max()), i
        else:
            assert torch.equal(got, ref), i
    assert torch.equal(out[True][0], out[True][2]) and torch.equal(out[True][1], out[True][3])


@pytest.mark.parametrize("M,N,K", [(4096, 1024, 512), (4096 + 32 * 5, 528, 264), (64, 16, 64)])
def test_feature_product_backward_in_the_data_gradient_epilogue(M, N, K, monkeypatch):
    """mirl_gemm3_nn_qp: with d = g @ w (never stored), d_pre = (emb > 0) * d * x[row // 32], dx = group sums of d * emb,
    db = column sums of d_pre — against the float64 evaluation of the same expressions; integer operands bit-exactly
    (tile tails: M not a multiple of 256, K = the product's width not a multiple of 64)."""
    from rltime_amd.models.torch import gemm3
    monkeypatch.setattr(gemm3, "_MIN_WORK", 0)
    gen = torch.Generator(device="cuda").manual_seed(M + N)
    for integer in (True, False):
        if integer:
            g = torch.randint(-3, 4, (M, N), device="cuda", generator=gen).float()
            w = torch.randint(-3, 4, (N, K), device="cuda", generator=gen).float()
            emb = torch.randint(-2, 5, (M, K), device="cuda", generator=gen).float().clamp_(min=0)
            x = torch.randint(-4, 5, (M // 32, K), device="cuda", generator=gen).float()
        else:
            g = torch.randn(M, N, device="cuda", generator=gen)
            w = torch.randn(N, K, device="cuda", generator=gen) / N ** 0.5
            emb = torch.randn(M, K, device="cuda", generator=gen).clamp_(min=0)
            x = torch.randn(M // 32, K, device="cuda", generator=gen)
        assert gemm3.grad_input_qp_supported(g, w, emb, x, 32)
        d_pre, dx, db = gemm3.grad_input_qp(g, w, emb, x)
        d = g.double() @ w.double()
        want_pre = (emb > 0) * d * x.double().repeat_interleave(32, dim=0)
        want_dx = (d * emb.double()).view(M // 32, 32, K).sum(1)
        want_db = want_pre.sum(0)
        for got, want, what in ((d_pre, want_pre, "d_pre"), (dx, want_dx, "dx"), (db, want_db, "db")):
            if integer:
                assert torch.equal(got.double(), want), what
            else:
                assert float((got.double() - want).abs().max()) <= 1e-5 * float(want.abs().max()), what
    again = gemm3.grad_input_qp(g, w, emb, x)
    assert all(torch.equal(a, b) for a, b in zip(again, (d_pre, dx, db)))          # fixed summation orders
    assert not gemm3.grad_input_qp_supported(g, w, emb, x, 16)                      # groups of 32 rows only


def _iqn_head(gen, M, n, D, Fd, H, A):
    mk = lambda *s: torch.randn(*s, device="cuda", generator=gen)       # noqa: E731
    x = mk(M, Fd)
    phi = torch.cos(torch.rand(M * n, 1, device="cuda", generator=gen) * torch.arange(1, D + 1, device="cuda") * 3.14159265)
    qp = [mk(Fd, D) / D ** 0.5, mk(Fd) * 0.1]
    tail = [mk(H, Fd) / Fd ** 0.5, mk(H), mk(A, H) / H ** 0.5, mk(A), mk(H, Fd) / Fd ** 0.5, mk(H), mk(1, H) / H ** 0.5, mk(1)]
    return x, phi, qp, tail, mk(M * n, A), mk(M * n, 1)


@pytest.mark.parametrize("second_consumer", [False, True])
def test_dueling_tail_runs_the_feature_products_backward(monkeypatch, second_consumer):
    """quantile_product -> (a reshape view) -> dueling tail, the IQN head of policies/torch/iqn.py:82-102 + dqn.py:101-112:
    with the hand-over (fused._QPLink) the tail's data-gradient GEMM runs the product's backward (k_gemm3_nn_qp), the
    stand-alone pass k_iqn_mul_bwd does not run, and every gradient equals the separate path's within f32 rounding; with
    a SECOND consumer of the product's output the stand-alone pass runs on that consumer's gradient alone and the sums
    are the same."""
    from rltime_amd import _lib
    from rltime_amd.models.torch import fused, gemm3
    gen = torch.Generator(device="cuda").manual_seed(77)
    M, n, D, Fd, H, A = 256 + 8, 32, 64, 512, 512, 6
    x, phi, qp, tail, ga, gv = _iqn_head(gen, M, n, D, Fd, H, A)
    side = torch.randn(M * n, Fd, device="cuda", generator=gen)
    monkeypatch.setattr(gemm3, "_MIN_WORK", 0)
    out = {}
    for mode in (True, False):
        monkeypatch.setattr(gemm3, "_QP_BWD", mode)
        xx = x.clone().requires_grad_(True)
        pq = [p.clone().requires_grad_(True) for p in qp]
        pt = [p.clone().requires_grad_(True) for p in tail]
        _lib.check(_lib.lib.mirl_profile_reset())
        _lib.check(_lib.lib.mirl_profile_set(2))
        try:
            prod = fused.quantile_product(xx, phi, pq[0], pq[1], n)
            a, v = fused._DuelingTail.apply(prod.reshape(-1, Fd), *pt)
            outs, grads = [a, v], [ga, gv]
            if second_consumer:
                outs.append((prod * side).sum())
                grads.append(torch.ones((), device="cuda"))
            torch.autograd.backward(outs, grads)
            torch.cuda.synchronize()
            ran = {r["name"]: r["calls"] for r in _lib.profile_table()}
        finally:
            _lib.check(_lib.lib.mirl_profile_set(0))
        assert bool(ran.get("k_gemm3_nn_qp")) == mode, ran
        assert bool(ran.get("k_iqn_mul_bwd")) == (second_consumer or not mode), ran
        out[mode] = [xx.grad] + [p.grad for p in pq] + [p.grad for p in pt]
    for i, (got, ref) in enumerate(zip(out[True], out[False])):
        assert float((got - ref).abs().max()) <= 2e-5 * float(ref.abs().max()), i


@pytest.mark.parametrize("second_consumer", [False, True])
def test_feature_product_of_a_width_that_is_not_a_power_of_two(monkeypatch, second_consumer):
    """The IQN head straight on the conv stack's output (no FC layer in front of the product: 3136 features in the
    Rainbow-IQN config; 784 here): the product runs in the embedding GEMM's epilogue and its backward in the dueling
    tail's data-gradient GEMM like the 512-wide one; a second consumer's share goes through plain tensor expressions
    (the stand-alone kernel takes 4 * 2^k columns only).  Against the unfused expressions of iqn.py:82-102."""
    from rltime_amd import _lib
    from rltime_amd.models.torch import fused, gemm3
    gen = torch.Generator(device="cuda").manual_seed(78)
    M, n, D, Fd, H, A = 128 + 8, 32, 64, 784, 512, 6
    x, phi, qp, tail, ga, gv = _iqn_head(gen, M, n, D, Fd, H, A)
    side = torch.randn(M * n, Fd, device="cuda", generator=gen)
    monkeypatch.setattr(gemm3, "_MIN_WORK", 0)
    out = {}
    for mode in ("fused", "plain"):
        xx = x.clone().requires_grad_(True)
        pq = [p.clone().requires_grad_(True) for p in qp]
        pt = [p.clone().requires_grad_(True) for p in tail]
        _lib.check(_lib.lib.mirl_profile_reset())
        _lib.check(_lib.lib.mirl_profile_set(2))
        try:
            if mode == "fused":
                prod = fused.quantile_product(xx, phi, pq[0], pq[1], n)
                a, v = fused._DuelingTail.apply(prod.reshape(-1, Fd), *pt)
            else:
                emb = torch.relu(torch.nn.functional.linear(phi, pq[0], pq[1]))
                prod = (xx.unsqueeze(1) * emb.view(M, n, Fd)).reshape(M * n, Fd)
                h = torch.relu(torch.nn.functional.linear(prod, pt[0], pt[1]))
                hv = torch.relu(torch.nn.functional.linear(prod, pt[4], pt[5]))
                a, v = torch.nn.functional.linear(h, pt[2], pt[3]), torch.nn.functional.linear(hv, pt[6], pt[7])
            outs, grads = [a, v], [ga, gv]
            if second_consumer:
                outs.append((prod * side).sum())
                grads.append(torch.ones((), device="cuda"))
            torch.autograd.backward(outs, grads)
            torch.cuda.synchronize()
            ran = {r["name"]: r["calls"] for r in _lib.profile_table()}
        finally:
            _lib.check(_lib.lib.mirl_profile_set(0))
        if mode == "fused":
            assert ran.get("k_gemm3_nt_mul") and ran.get("k_gemm3_nn_qp") and not ran.get("k_iqn_mul_bwd"), ran
        out[mode] = [a.detach(), v.detach(), xx.grad] + [p.grad for p in pq] + [p.grad for p in pt]
    for i, (got, ref) in enumerate(zip(out["fused"], out["plain"])):
        # a ReLU whose pre-activation rounds to the other side of zero in one of the two paths (expected: about one of the
        # 4.4 M hidden units here) moves ITS unit's row of dW1 / dWv and ITS sample's row of dx by one sample's term, and
        # through that row of dx every other weight gradient by one sample's share of a 4352-sample sum: a handful of rows
        # may leave the bar, which is 1e-4 of scale for the outputs and dx, 1e-3 for the weight gradients
        scale = float(ref.abs().max())
        off = (got - ref).abs() > (1e-4 if i <= 2 else 1e-3) * scale
        bad = int(off.any(dim=-1).sum()) if off.dim() == 2 else int(off.sum())
        assert bad <= (0 if i < 2 else 8), (i, bad)
    with torch.no_grad():
        again = fused.quantile_product(x, phi, qp[0], qp[1], n)
        want = torch.relu(phi.double() @ qp[0].double().t() + qp[1].double()) * x.double().repeat_interleave(n, dim=0)
    assert float((again.double() - want).abs().max()) <= 1e-5 * float(want.abs().max())
