#!/usr/bin/env python3
"""The window-staged 3x3 conv on the VAE's shapes (128 / 256 / 512 channels, 72 .. 576 px images): the 128-column family of
csrc/conv_win.hip, linear tiles at 72 px, 2-D tiles (16 output columns x 8 / 16 rows) from 144 px, plain and with the fused nearest-2x
upsample.

    python tools/kconvwin_vae.py [--iters N] [--rounds R]

1. exactness on integer data against torch (bias, fp32 residual, f16 / fp32 outputs, GroupNorm statistics summed per image and channel),
   window kernel (knob conv_win unset / 1 / 2) and per-tap gather (0);
2. interleaved timing on the decoder's conv shapes at 7 frames per pass."""
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
ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--skip-exact", action="store_true")
args = ap.parse_args()


def ints(shape, lo, hi, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(lo, hi + 1, shape, generator=g).float().to(dev)


def exact_case(n, ih, iw, cin, cout, up, seed):
    x = ints((n, cin, ih, iw), -3, 3, seed)
    w = ints((cout, cin, 3, 3), -2, 2, seed + 1)
    b = ints((cout,), -4, 4, seed + 2)
    s = 2 if up else 1
    oh, ow = s * ih, s * iw
    res = ints((n, oh * ow, cout), -5, 5, seed + 4)
    xin = F.interpolate(x, scale_factor=2, mode="nearest") if up else x
    ref = F.conv2d(xin, w, b, padding=1).permute(0, 2, 3, 1).reshape(n, oh * ow, cout) + res
    xh, wp = x.permute(0, 2, 3, 1).contiguous().half(), pack_conv3x3(w)
    M = n * oh * ow
    bad = []
    for knob in (-1, 1, 2, 0):
        ops.set_knob("conv_win", knob)
        out = torch.full((n, oh * ow, cout), float("nan"), device=dev)
        o16 = torch.full((n, oh * ow, cout), float("nan"), device=dev, dtype=torch.float16)
        narrow = cout <= 32
        st = None if narrow else torch.full(ops.channel_stats_shape(M, cout), float("nan"), device=dev)
        ops.conv3x3(xh, wp, bias=b, residual=res, out_f32=out, out_f16=None if narrow else o16, ch_stats=st, upsample=up)
        torch.cuda.synchronize()
        ok = torch.equal(out, ref) and (narrow or torch.equal(o16, ref.half()))
        if not narrow:
            nb = (oh * ow) // 64  # blocks per image: the consumer sums them per image
            ssum = st[:, 0].view(n, nb, cout).sum(1)
            ok = ok and torch.equal(ssum, ref.sum(1)) and torch.allclose(st[:, 1].view(n, nb, cout).double().sum(1), (ref.double() ** 2).sum(1), rtol=1e-6, atol=0)
        if not ok:
            bad.append((knob, float((out - ref).abs().nan_to_num(1e9).max())))
    return bad


nbad = 0
if not args.skip_exact:
    cases = [(2, 72, 72, 128, 128, False), (1, 144, 144, 64, 256, False), (2, 144, 144, 128, 128, False), (1, 288, 288, 64, 128, False),
             (1, 576, 576, 64, 128, False), (3, 32, 48, 64, 128, False), (2, 16, 16, 64, 384, False), (1, 160, 96, 64, 128, False),
             (2, 72, 72, 64, 128, True), (1, 144, 144, 64, 256, True), (1, 288, 288, 64, 128, True), (3, 24, 40, 64, 128, True), (2, 8, 8, 128, 256, True),
             (2, 72, 72, 128, 4, False), (1, 576, 576, 64, 4, False), (3, 9, 9, 64, 4, False), (2, 36, 36, 128, 32, False), (1, 144, 144, 64, 8, False)]
    for k, c in enumerate(cases):
        bad = exact_case(*c, seed=300 + 7 * k)
        nbad += len(bad)
        print(f"exact {c}: {'OK' if not bad else 'MISMATCH ' + str(bad)}", flush=True)
    print(f"exactness: {len(cases)} cases x knobs (-1, 1, 2, 0), {nbad} mismatches", flush=True)


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


print("== VAE decoder convs at 7 frames per pass: side cin cout up | us per-tap (TFLOP/s) | us window, default dispatch (TFLOP/s) | 4-wave family | 8-wave family", flush=True)
tot = {0: 0.0, -1: 0.0, 1: 0.0, 2: 0.0}
for side, cin, cout, up, calls in [(72, 512, 512, False, 9), (144, 512, 512, False, 6), (288, 512, 256, False, 1), (288, 256, 256, False, 5), (576, 256, 128, False, 1),
                                   (576, 128, 128, False, 5), (72, 512, 512, True, 1), (144, 512, 512, True, 1), (288, 256, 256, True, 1), (576, 128, 4, False, 1), (-72, 640, 4, False, 0)]:
    n = 7 if side > 0 else 42  # side < 0: the UNet's head conv at batch 42 (not part of a decode)
    side = abs(side)
    s = 2 if up else 1
    M = n * (s * side) ** 2
    x = torch.randn(n, side, side, cin, device=dev, dtype=torch.float16)
    w = (torch.randn(cout, 9 * cin, device=dev) * 0.02).half()
    b = torch.randn(cout, device=dev)
    out = torch.empty(n, (s * side) ** 2, cout, device=dev)
    st = torch.empty(ops.channel_stats_shape(M, cout), device=dev) if cout > 32 else None
    best = {0: 1e30, -1: 1e30, 1: 1e30, 2: 1e30}

    def call(knob):
        ops.set_knob("conv_win", knob)
        ops.conv3x3(x, w, bias=b, out_f32=out, ch_stats=st, upsample=up)

    for _ in range(args.rounds):
        for knob in (0, -1, 1, 2):
            best[knob] = min(best[knob], timeit(lambda: call(knob), args.iters))
    fl = 2.0 * M * cout * 9 * cin
    for knob in (0, -1, 1, 2):
        tot[knob] += best[knob] * calls / 1e3
    print(f"{side:4d} {cin:4d} {cout:4d} {'up' if up else '  '} x{calls} | {best[0]:9.1f} ({fl / best[0] / 1e6:6.1f}) | {best[-1]:9.1f} ({fl / best[-1] / 1e6:6.1f}) | {best[1]:9.1f} | {best[2]:9.1f}", flush=True)
    del x, out, st
print(f"   decoder 3x3 convs per 7-frame pass (approx. call counts): per-tap {tot[0]:.2f} ms, window {tot[-1]:.2f} ms (4-wave only {tot[1]:.2f}, 8-wave only {tot[2]:.2f})", flush=True)
ops.set_knob("conv_win", -1)
sys.exit(1 if nbad else 0)
