"""Correctness + time of csrc/gemm3.hip against the float64 product and the library f32 GEMM.
usage: python tools/gemm3_probe.py [layout:MxNxK ...]   (layout in nt, nn, tn; also qp:RxNxK:n and head:MxNxK:O)"""
import json
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rltime_amd.models.torch import gemm3

DEFAULT = ["tn:512x64x1310720", "nt:655360x1024x512", "nn:655360x512x1024", "tn:1024x512x655360", "nt:40960x2048x3136",
           "nn:20480x3136x2048", "tn:2048x3136x20480", "nt:1000x300x64", "nn:777x260x48", "tn:300x260x4112"]


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def quantile_product_probe(spec, g):
    """qp:RxNxK:n — out[r] = x[r // n] * relu(phi[r] @ W^T + b) (mirl_gemm3_nt_mul), with and without the stored embedding:
    checked against float64 on a row slice, timed, priced against its HBM bytes (a K = 64 product is a store kernel)."""
    _, dims, n = spec.split(":")
    R, N, K = (int(v) for v in dims.split("x"))
    n = int(n)
    x = torch.randn(R // n, N, device="cuda", generator=g)
    phi = torch.cos(torch.rand(R, K, device="cuda", generator=g) * 3.0)
    w = torch.randn(N, K, device="cuda", generator=g) * 0.125
    b = torch.randn(N, device="cuda", generator=g) * 0.1
    rec = {"layout": "nt_mul", "R": R, "N": N, "K": K, "group": n, "supported": bool(gemm3.quantile_product_supported(x, phi, w, b, n))}
    if rec["supported"]:
        rows = torch.unique(torch.cat([torch.arange(0, 96), torch.randint(0, R, (256,)), torch.arange(R - 96, R)])).cuda()
        ref = x[rows // n].double() * torch.relu(phi[rows].double() @ w.double().t() + b.double())
        scale = float(ref.abs().max())
        for keep in (False, True):
            out, emb = gemm3.quantile_product(x, phi, w, b, n, keep)
            rec["max_err_keep%d" % keep] = float((out[rows].double() - ref).abs().max()) / scale
            if keep:
                e64 = torch.relu(phi[rows].double() @ w.double().t() + b.double())
                rec["max_err_embedding"] = float((emb[rows].double() - e64).abs().max()) / float(e64.abs().max())
            t = timed(lambda: gemm3.quantile_product(x, phi, w, b, n, keep), 10)
            by = 4.0 * (R * K + N * K + R * N * (2 if keep else 1) + (R // n) * N)
            rec["ms_keep%d" % keep] = round(t, 4)
            rec["GBps_keep%d" % keep] = round(by / t / 1e6, 1)
            rec["frac_of_hbm_peak_keep%d" % keep] = round(by / t / 1e6 / 8000.0, 3)
            del out, emb
    print(json.dumps(rec), flush=True)
    torch.cuda.empty_cache()


def head_probe(spec, g):
    """head:MxNxK:O — relu(x @ w^T + b) and the O output units behind it in one launch (mirl_gemm3_nt_head), with and
    without the hidden activation written, timed beside the plain NT product of the same shape."""
    _, dims, o = spec.split(":")
    M, N, K = (int(v) for v in dims.split("x"))
    O = int(o)
    x = torch.randn(M, K, device="cuda", generator=g)
    w = torch.randn(N, K, device="cuda", generator=g) * 0.05
    b = torch.randn(N, device="cuda", generator=g) * 0.1
    w2 = torch.randn(O, N, device="cuda", generator=g) * 0.05
    b2 = torch.randn(O, device="cuda", generator=g)
    rec = {"layout": "nt_head", "M": M, "N": N, "K": K, "O": O, "supported": bool(gemm3.head_supported(x, w, b, w2))}
    if rec["supported"]:
        flop = 2.0 * M * N * K
        rec["ms_plain_nt"] = round(timed(lambda: gemm3.gemm(gemm3.NT, x, w, bias=b, relu=True), 20), 4)
        for keep in (False, True):
            t = timed(lambda: gemm3.linear_relu_head(x, w, b, w2, b2, keep), 20)
            rec["ms_keep%d" % keep] = round(t, 4)
            rec["tflops_keep%d" % keep] = round(flop / t / 1e9, 1)
    print(json.dumps(rec), flush=True)
    torch.cuda.empty_cache()


def main():
    specs = sys.argv[1:] or DEFAULT
    g = torch.Generator(device="cuda").manual_seed(0)
    for spec in specs:
        if spec.startswith("qp:"):
            quantile_product_probe(spec, g)
            continue
        if spec.startswith("head:"):
            head_probe(spec, g)
            continue
        lay, dims = spec.split(":")
        M, N, K = (int(v) for v in dims.split("x"))
        layout = {"nt": gemm3.NT, "nn": gemm3.NN, "tn": gemm3.TN}[lay]
        if layout == gemm3.NT:
            a, b = torch.randn(M, K, device="cuda", generator=g), torch.randn(N, K, device="cuda", generator=g)
            lib = lambda: torch.mm(a, b.t())
        elif layout == gemm3.NN:
            a, b = torch.randn(M, K, device="cuda", generator=g), torch.randn(K, N, device="cuda", generator=g)
            lib = lambda: torch.mm(a, b)
        else:
            a, b = torch.randn(K, M, device="cuda", generator=g), torch.randn(K, N, device="cuda", generator=g)
            lib = lambda: torch.mm(a.t(), b)
        # a few scaled rows / columns so magnitudes differ across the tile
        a[:: 7] *= 37.0
        b[:: 5] *= 0.013
        if os.environ.get("GEMM3_ZERO") == "1":        # power experiment: the same instruction stream on all-zero operands
            a.zero_(); b.zero_()
        elif os.environ.get("GEMM3_ZERO") == "2":      # ... and on operands with one significant bit (lo / mid parts are zero)
            a.copy_(torch.sign(a)); b.copy_(torch.sign(b))
        ok = gemm3.supported(layout, a, b, min_work=0)
        rec = {"layout": lay, "M": M, "N": N, "K": K, "supported": ok}
        if ok:
            out = gemm3.gemm(layout, a, b)
            ref32 = lib()
            # float64 check on a slice of rows (all columns): rows across the whole range incl. the tail
            rows = torch.unique(torch.cat([torch.arange(0, min(M, 64)), torch.randint(0, M, (192,)), torch.arange(max(M - 64, 0), M)])).cuda()
            if layout == gemm3.TN:
                ref64 = a[:, rows].double().t() @ b.double()
            elif layout == gemm3.NN:
                ref64 = a[rows].double() @ b.double()
            else:
                ref64 = a[rows].double() @ b.double().t()
            scale = float(ref64.abs().max()) or 1.0
            rec["max_err_gemm3"] = float((out[rows].double() - ref64).abs().max()) / scale
            rec["max_err_lib_f32"] = float((ref32[rows].double() - ref64).abs().max()) / scale
            rec["max_diff_vs_lib"] = float((out - ref32).abs().max()) / scale
            flop = 2.0 * M * N * K
            reps = 5 if flop > 1e11 else 20
            t3 = timed(lambda: gemm3.gemm(layout, a, b, out=out), reps)
            tl = timed(lib, reps)
            rec.update(ms_gemm3=round(t3, 4), ms_lib_f32=round(tl, 4), tflops_gemm3=round(flop / t3 / 1e9, 1),
                       tflops_lib=round(flop / tl / 1e9, 1), frac_of_bf16x6_peak=round(flop / t3 / 1e9 / 416.7, 3))
        print(json.dumps(rec), flush=True)
        del a, b
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
