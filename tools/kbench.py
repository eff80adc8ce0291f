#!/usr/bin/env python3
"""Per-kernel micro-benchmark at the headline shapes (T=21, 576x576 -> latent 72x72, CFG batch 42).

    python tools/kbench.py [attn|gemm|conv|norm|all] [--iters N] [--quick]

Times every distinct (kernel, shape) of one denoising step with HIP events on the launch stream and
prints achieved TFLOP/s (MFMA kernels) or GB/s (HBM kernels) next to the call count per step, i.e. a
per-shape breakdown of where the step time goes.  Random operands (zero data flatters MFMA clocks).
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stable-virtual-camera_amd"))
import torch  # noqa: E402

from seva import ops  # noqa: E402
from seva._engine import interleave_geglu  # noqa: E402

dev = torch.device("cuda:0")
F16, F32 = torch.float16, torch.float32
T, N = 21, 42
LEVELS = [(72, 320, 5), (36, 640, 10), (18, 1280, 20), (9, 1280, 20)]  # (side, C, heads)


def timeit(fn, iters):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3  # us


def r16(*s):
    return torch.randn(*s, device=dev, dtype=F16)


def r32(*s):
    return torch.randn(*s, device=dev, dtype=F32)


def bench_gemm(iters, quick):
    print("== plain GEMM (fp16 MFMA): name M N K | us | TFLOP/s | GB/s(alg) | calls/step | ms/step")
    tot = 0.0
    for side, C, _ in LEVELS[: 3 if quick else 4]:
        M = N * side * side
        nmvt = {72: 5, 36: 5, 18: 5, 9: 1}[side]
        cases = [
            ("proj_in  f32out", C, C, dict(o32=1), nmvt),
            ("qkv      f16out", 3 * C, C, dict(o16=1), 2 * nmvt),
            ("attn_out +res", C, C, dict(o32=1, res=1, radd=1), 2 * nmvt),
            ("geglu    f16out", 8 * C, C, dict(o16=1, geglu=1), 3 * nmvt),
            ("ff2      +res", C, 4 * C, dict(o32=1, res=1), 2 * nmvt),
            ("ff2_last f16out", C, 4 * C, dict(o16=1, res=1), nmvt),
            ("proj_out +res", C, C, dict(o32=1, res=1), nmvt),
        ]
        for name, n_, k_, fl, calls in cases:
            a = r16(M, k_)
            w = r16(n_, k_) * (k_ ** -0.5)
            bias = r32(n_)
            nout = n_ // 2 if fl.get("geglu") else n_
            if fl.get("geglu"):
                w, bias = interleave_geglu(w, bias)
            o32 = torch.empty(M, nout, device=dev) if fl.get("o32") else None
            o16 = torch.empty(M, nout, device=dev, dtype=F16) if fl.get("o16") else None
            res = r32(M, nout) if fl.get("res") else None
            radd = r32(N, nout) if fl.get("radd") else None
            fn = lambda: ops.gemm(a, w, bias=bias, residual=res, row_add=radd, rows_per_group=side * side,
                                  out_f32=o32, out_f16=o16, geglu=bool(fl.get("geglu")))
            us = timeit(fn, iters)
            flops = 2.0 * M * n_ * k_
            byts = M * k_ * 2 + n_ * k_ * 2 + M * nout * (4 * bool(o32 is not None) + 2 * bool(o16 is not None) + 4 * bool(res is not None))
            tot += us * calls / 1e3
            print(f"ds{72 // side} {name:16s} {M:7d} {n_:6d} {k_:5d} | {us:9.1f} | {flops / us / 1e6:7.1f} | {byts / us / 1e3:7.1f} | {calls:3d} | {us * calls / 1e3:7.2f}")
            del a, w, o32, o16, res
    print(f"   plain+geglu GEMM total per step: {tot:.2f} ms")


def bench_conv(iters, quick):
    print("== conv3x3 implicit GEMM: side cin cout | us | TFLOP/s | calls/step | ms/step")
    cases = [(72, 320, 320, 9), (72, 960, 320, 1), (72, 640, 320, 2), (36, 640, 640, 9), (36, 320, 640, 1),
             (36, 1920, 640, 1), (36, 1280, 640, 1), (36, 960, 640, 1), (18, 1280, 1280, 9), (18, 640, 1280, 1),
             (18, 2560, 1280, 2), (18, 1920, 1280, 1), (9, 1280, 1280, 9), (9, 2560, 1280, 3)]
    tot = 0.0
    for side, cin, cout, calls in cases[: 9 if quick else None]:
        x = r16(N, side, side, cin)
        w = r16(cout, 9 * cin) * ((9 * cin) ** -0.5)
        bias, res = r32(cout), r32(N, side * side, cout)
        out = torch.empty(N, side * side, cout, device=dev)
        us = timeit(lambda: ops.conv3x3(x, w, bias=bias, residual=res, out_f32=out), iters)
        flops = 2.0 * N * side * side * cout * 9 * cin
        tot += us * calls / 1e3
        print(f"{side:3d} {cin:5d} {cout:5d} | {us:9.1f} | {flops / us / 1e6:7.1f} | {calls:2d} | {us * calls / 1e3:7.2f}")
    print(f"   conv total per step (approx.): {tot:.2f} ms")


def bench_attn(iters, quick):
    print("== attention d=64: regime B H L | us | TFLOP/s | calls/step | ms/step")
    tot = 0.0
    for side, C, H in LEVELS:
        hw = side * side
        pre = os.environ.get("KBENCH_ATTN_PRE", "1") != "0"  # engine path: q carries scale*log2(e)
        qkv32 = torch.randn(N * hw, 3 * C, device=dev)
        if pre:
            qkv32[:, :C] *= 0.125 * 1.4426950408889634
        qkv = qkv32.half()
        del qkv32
        out = torch.empty(N * hw, C, device=dev, dtype=F16)
        q, k, v = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]
        c3 = 3 * C
        regs = []
        if side in (72, 36, 18):
            regs.append(("frame", dict(nb0=N, nb1=1, lq=hw, lk=hw, q_strides=(hw * c3, 0, c3), k_strides=(hw * c3, 0, c3), o_strides=(hw * C, 0, C)), 5 if side == 72 else 2))
        if side in (36, 18, 9):
            regs.append(("joint", dict(nb0=2, nb1=1, lq=T * hw, lk=T * hw, q_strides=(T * hw * c3, 0, c3), k_strides=(T * hw * c3, 0, c3), o_strides=(T * hw * C, 0, C)), 3 if side != 9 else 1))
        regs.append(("temporal", dict(nb0=2, nb1=hw, lq=T, lk=T, q_strides=(T * hw * c3, c3, hw * c3), k_strides=(T * hw * c3, c3, hw * c3), o_strides=(T * hw * C, C, hw * C)), {72: 5, 36: 5, 18: 5, 9: 1}[side]))
        for name, kw, calls in regs:
            us = timeit(lambda: ops.attention(q, k, v, out, heads=H, q_prescaled=pre, **kw), iters)
            flops = 4.0 * kw["nb0"] * kw["nb1"] * H * kw["lq"] * kw["lk"] * 64
            byts = N * hw * C * 2 * 4
            tot += us * calls / 1e3
            print(f"ds{72 // side} {name:9s} B={kw['nb0'] * kw['nb1']:6d} H={H:2d} L={kw['lq']:6d} | {us:9.1f} | {flops / us / 1e6:7.1f} TF | {byts / us / 1e3:7.1f} GB/s | {calls} | {us * calls / 1e3:7.2f}")
        del qkv, out
    print(f"   attention total per step: {tot:.2f} ms")


def bench_norm(iters, quick):
    print("== norms (HBM-bound): kind shape | us | GB/s(alg)")
    for side, C, _ in LEVELS[:3]:
        hw = side * side
        x = r32(N, hw, C)
        g, b = r32(C), r32(C)
        o = torch.empty(N, hw, C, device=dev, dtype=F16)
        ws = ops.groupnorm_workspace(N, dev)
        dense, dw, db = r32(N, hw, 6), r32(2 * C, 6), r32(2 * C)
        us = timeit(lambda: ops.groupnorm(x, None, g, b, o, ws, silu=True, dense=dense, dense_w=dw, dense_b=db), iters)
        print(f"groupnorm+silu+mod [{N},{hw},{C}] | {us:8.1f} | {N * hw * C * 10 / us / 1e3:7.1f}")
        us = timeit(lambda: ops.groupnorm(x, None, g, b, o, ws, silu=False), iters)
        print(f"groupnorm plain    [{N},{hw},{C}] | {us:8.1f} | {N * hw * C * 10 / us / 1e3:7.1f}")
        us = timeit(lambda: ops.layernorm(x, g, b, o), iters)
        print(f"layernorm          [{N * hw},{C}] | {us:8.1f} | {N * hw * C * 6 / us / 1e3:7.1f}")
        del x, o


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("what", nargs="?", default="all")
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--quick", action="store_true")
    a = ap.parse_args()
    torch.manual_seed(0)
    for name, fn in (("attn", bench_attn), ("gemm", bench_gemm), ("conv", bench_conv), ("norm", bench_norm)):
        if a.what in ("all", name):
            fn(a.iters, a.quick)
