#!/usr/bin/env python3
"""The window-staged 3x3 conv on e4m3 operands (fp8 mode, BASELINE config 5) against the per-tap fp8 kernel:

    python tools/kconvwin_fp8.py [--iters N] [--rounds R]

1. exactness on integer data (e4m3 holds small integers exactly; power-of-two weight scales) against torch, knob conv_win unset / 1 / 2 / 0;
2. interleaved timing on the fp8 conv shapes of a step (the C >= 640 levels)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stable-virtual-camera_amd"))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from seva import ops  # noqa: E402
from seva._engine import pack_conv3x3  # noqa: E402

dev = torch.device("cuda:0")
U8 = torch.uint8
ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--rounds", type=int, default=3)
args = ap.parse_args()


def ints(shape, lo, hi, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(lo, hi + 1, shape, generator=g).float().to(dev)


def f8(x):
    return x.to(torch.float8_e4m3fn).view(U8)


nbad = 0
for k, (n, ih, iw, cin, cout, stats) in enumerate([(3, 8, 8, 640, 640, False), (42, 9, 9, 128, 1280, False), (2, 36, 36, 128, 640, False), (5, 18, 18, 256, 1280, False),
                                                    (2, 72, 72, 128, 128, True), (1, 33, 31, 128, 256, False), (2, 16, 16, 128, 384, True), (7, 5, 4, 256, 640, False)]):
    x = ints((n, cin, ih, iw), -3, 3, 50 + k)
    w = ints((cout, cin, 3, 3), -2, 2, 60 + k)
    g = torch.Generator().manual_seed(70 + k)
    e = torch.randint(-2, 3, (cout,), generator=g).to(dev)
    wf = w * torch.exp2(e.float())[:, None, None, None]
    b = ints((cout,), -4, 4, 80 + k)
    res = ints((n, ih * iw, cout), -5, 5, 90 + k)
    ref = F.conv2d(x, wf, b, padding=1).permute(0, 2, 3, 1).reshape(n, ih * iw, cout) + res
    x8, w8 = f8(x.permute(0, 2, 3, 1).contiguous()), f8(pack_conv3x3(w).float())
    M = n * ih * iw
    for knob in (-1, 1, 2, 0):
        ops.set_knob("conv_win", knob)
        out = torch.full((n, ih * iw, cout), float("nan"), device=dev)
        st = torch.full(ops.channel_stats_shape(M, cout), float("nan"), device=dev) if stats else None
        ops.conv3x3(x8, w8, w_exp=(e + 127).to(U8), bias=b, residual=res, out_f32=out, ch_stats=st)
        torch.cuda.synchronize()
        ok = torch.equal(out, ref)
        if st is not None:
            nb = ih * iw // 64
            ok = ok and torch.equal(st[:, 0].view(n, nb, cout).sum(1), ref.sum(1))
        if not ok:
            nbad += 1
            print(f"MISMATCH {(n, ih, iw, cin, cout)} knob {knob}: max diff {float((out - ref).abs().nan_to_num(1e9).max())}", flush=True)
print(f"exactness: {nbad} mismatches", flush=True)


def timeit(fn, iters):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


print("== fp8 conv3x3 at batch 42: side cin cout | us per-tap (TFLOP/s) | us window default (TFLOP/s) | 4-wave | 8-wave", flush=True)
tot = {0: 0.0, -1: 0.0, 1: 0.0, 2: 0.0}
for side, cin, cout, calls in [(36, 640, 640, 9), (36, 1920, 640, 1), (36, 1280, 640, 1), (18, 1280, 1280, 9), (18, 640, 1280, 1), (18, 2560, 1280, 2), (18, 1920, 1280, 1),
                               (9, 1280, 1280, 9), (9, 2560, 1280, 3)]:
    n = 42
    x8 = torch.randint(0, 120, (n, side, side, cin), device=dev, dtype=U8)
    w8 = torch.randint(0, 120, (cout, 9 * cin), device=dev, dtype=U8)
    we = torch.full((cout,), 120, device=dev, dtype=U8)
    b = torch.randn(cout, device=dev)
    out = torch.empty(n, side * side, cout, device=dev)
    best = dict.fromkeys(tot, 1e30)

    def call(knob):
        ops.set_knob("conv_win", knob)
        ops.conv3x3(x8, w8, w_exp=we, bias=b, out_f32=out)

    for _ in range(args.rounds):
        for knob in best:
            best[knob] = min(best[knob], timeit(lambda: call(knob), args.iters))
    fl = 2.0 * n * side * side * cout * 9 * cin
    for knob in best:
        tot[knob] += best[knob] * calls / 1e3
    print(f"{side:3d} {cin:5d} {cout:5d} x{calls} | {best[0]:8.1f} ({fl / best[0] / 1e6:6.1f}) | {best[-1]:8.1f} ({fl / best[-1] / 1e6:6.1f}) | {best[1]:8.1f} | {best[2]:8.1f}", flush=True)
print(f"   fp8 convs per step: per-tap {tot[0]:.2f} ms, window {tot[-1]:.2f} ms (4-wave only {tot[1]:.2f}, 8-wave only {tot[2]:.2f})", flush=True)
ops.set_knob("conv_win", -1)
sys.exit(1 if nbad else 0)
