#!/usr/bin/env python3
"""Window-staged 3x3 conv (csrc/conv_win.hip) against the per-tap gather (csrc/gemm.hip conv mode), one process:

    python tools/kconvwin.py [--iters N] [--rounds R] [--variants 0,1,2]

1. exactness on integer data against torch for every UNet conv shape of a step (bias + row_add / residual, GroupNorm
   statistics where the engine asks for them) and a few ragged ones (M tails, windows that straddle images, 9x9 images);
2. random data: relative difference to the per-tap kernel (the reduction order differs: slab-outer instead of tap-outer);
3. timing of knob conv_win = 0 (per-tap gather), 1 (4-wave window kernel), 2 (8-wave 256-row window kernel), interleaved rounds.
"""
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
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--variants", default="0,1,2")
ap.add_argument("--skip-exact", action="store_true")
args = ap.parse_args()
variants = [int(v) for v in args.variants.split(",")]

# (side, cin, cout, calls per step, statistics emitted)
UP_SHAPES = [(36, 640, 640, 1), (18, 1280, 1280, 1), (9, 1280, 1280, 1)]  # source side, cin, cout, calls per step (fused nearest-2x upsample)
SHAPES = [(72, 320, 320, 9, True), (72, 960, 320, 1, True), (72, 640, 320, 2, True),
          (36, 640, 640, 9, False), (36, 320, 640, 1, False), (36, 1920, 640, 1, False), (36, 1280, 640, 1, False), (36, 960, 640, 1, False),
          (18, 1280, 1280, 9, False), (18, 640, 1280, 1, False), (18, 2560, 1280, 2, False), (18, 1920, 1280, 1, False),
          (9, 1280, 1280, 9, False), (9, 2560, 1280, 3, False)]


def ints(shape, lo, hi, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(lo, hi + 1, shape, generator=g).float().to(dev)


def run(x, w, knob, **kw):
    ops.set_knob("conv_win", knob)
    ops.conv3x3(x, w, **kw)


def exact_case(n, ih, iw, cin, cout, stats, seed):
    x = ints((n, cin, ih, iw), -3, 3, seed)
    w = ints((cout, cin, 3, 3), -2, 2, seed + 1)
    b = ints((cout,), -4, 4, seed + 2)
    emb = ints((n, cout), -2, 2, seed + 3)
    res = ints((n, ih * iw, cout), -5, 5, seed + 4)
    ref = F.conv2d(x, w, b, padding=1).permute(0, 2, 3, 1).reshape(n, ih * iw, cout) + emb[:, None, :] + res
    xh, wp = x.permute(0, 2, 3, 1).contiguous().half(), pack_conv3x3(w)
    M = n * ih * iw
    bad = []
    for knob in variants:
        out = torch.full((n, ih * iw, cout), float("nan"), device=dev)
        st = torch.full(ops.channel_stats_shape(M, cout), float("nan"), device=dev) if stats else None
        run(xh, wp, knob, bias=b, row_add=emb, rows_per_group=ih * iw, residual=res, out_f32=out, ch_stats=st)
        torch.cuda.synchronize()
        ok = torch.equal(out, ref)
        if st is not None:
            nb = st.shape[0]
            rp = torch.zeros(nb * 64, cout, device=dev)
            rp[:M] = ref.reshape(M, cout)
            rp = rp.view(nb, 64, cout)
            ok = ok and torch.equal(st[:, 0], rp.sum(1)) and torch.allclose(st[:, 1].double(), (rp.double() ** 2).sum(1), rtol=1e-5, atol=0)
        if not ok:
            bad.append((knob, float((out - ref).abs().nan_to_num(1e9).max())))
    return bad


nbad = 0
if not args.skip_exact:
    cases = [(42, s, s, ci, co, st) for s, ci, co, _, st in SHAPES if ci <= 960 or s <= 18]
    cases += [(3, 9, 9, 128, 160, False), (5, 7, 11, 64, 320, False), (1, 33, 31, 192, 160, False), (2, 16, 16, 64, 160, True),
              (7, 5, 4, 128, 320, False), (1, 72, 72, 64, 160, True), (4, 24, 40, 128, 480, False), (42, 9, 9, 256, 1280, False)]
    for k, c in enumerate(cases):
        bad = exact_case(*c, seed=100 + 7 * k)
        nbad += len(bad)
        print(f"exact {c}: {'OK' if not bad else 'MISMATCH ' + str(bad)}", flush=True)
    print(f"exactness: {len(cases)} cases x variants {variants}, {nbad} mismatches", flush=True)


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


print("== conv3x3 at the headline shapes (batch 42): side cin cout | us per variant | TFLOP/s | ms/step", flush=True)
tot = {v: 0.0 for v in variants}
for side, cin, cout, calls, stats in SHAPES:
    n = 42
    M = n * side * side
    x = torch.randn(n, side, side, cin, device=dev, dtype=torch.float16)
    w = (torch.randn(cout, 9 * cin, device=dev) * 0.02).half()
    b = torch.randn(cout, device=dev)
    res = torch.randn(n, side * side, cout, device=dev)
    out = torch.empty(n, side * side, cout, device=dev)
    st = torch.empty(ops.channel_stats_shape(M, cout), device=dev) if stats else None
    kw = dict(bias=b, residual=res, out_f32=out, ch_stats=st)
    outs = {}
    for v in variants:
        run(x, w, v, **kw)
        torch.cuda.synchronize()
        outs[v] = out.clone()
    rel = {v: float((outs[v] - outs[variants[0]]).norm() / outs[variants[0]].norm()) for v in variants[1:]}
    best = {v: 1e30 for v in variants}
    for _ in range(args.rounds):
        for v in variants:
            best[v] = min(best[v], timeit(lambda: run(x, w, v, **kw), args.iters))
    fl = 2.0 * M * cout * 9 * cin
    line = f"{side:3d} {cin:5d} {cout:5d} |"
    for v in variants:
        line += f" v{v} {best[v]:8.1f} us {fl / best[v] / 1e6:7.1f} TF |"
        tot[v] += best[v] * calls / 1e3
    line += " rel " + " ".join(f"{rel[v]:.1e}" for v in variants[1:])
    print(line, flush=True)
print("   conv total per step (ms): " + "  ".join(f"v{v} {tot[v]:.2f}" for v in variants), flush=True)
print("== fused nearest-2x upsample + conv3x3: source side cin cout | us per variant | TFLOP/s", flush=True)
totu = {v: 0.0 for v in variants}
for side, cin, cout, calls in UP_SHAPES:
    n = 42
    M = n * 4 * side * side
    x = torch.randn(n, side, side, cin, device=dev, dtype=torch.float16)
    w = (torch.randn(cout, 9 * cin, device=dev) * 0.02).half()
    b = torch.randn(cout, device=dev)
    out = torch.empty(n, 4 * side * side, cout, device=dev)
    st = torch.empty(ops.channel_stats_shape(M, cout), device=dev) if (4 * side * side) % 64 == 0 else None
    kw = dict(bias=b, out_f32=out, ch_stats=st, upsample=True)
    outs = {}
    for v in variants:
        run(x, w, v, **kw)
        torch.cuda.synchronize()
        outs[v] = out.clone()
    rel = {v: float((outs[v] - outs[variants[0]]).norm() / outs[variants[0]].norm()) for v in variants[1:]}
    best = {v: 1e30 for v in variants}
    for _ in range(args.rounds):
        for v in variants:
            best[v] = min(best[v], timeit(lambda: run(x, w, v, **kw), args.iters))
    fl = 2.0 * M * cout * 9 * cin
    line = f"{side:3d}->{2 * side:3d} {cin:5d} {cout:5d} |"
    for v in variants:
        line += f" v{v} {best[v]:8.1f} us {fl / best[v] / 1e6:7.1f} TF |"
        totu[v] += best[v] * calls / 1e3
    line += " rel " + " ".join(f"{rel[v]:.1e}" for v in variants[1:])
    print(line, flush=True)
print("   upsample conv total per step (ms): " + "  ".join(f"v{v} {totu[v]:.2f}" for v in variants), flush=True)
ops.set_knob("conv_win", -1)
sys.exit(1 if nbad else 0)
