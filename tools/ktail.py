#!/usr/bin/env python3
"""How much does tile quantisation cost?  conv 640->640 at 36x36 and 1280->1280 at 18x18 for batch sizes around the
points where the tile count crosses a multiple of the 512 workgroup slots."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stable-virtual-camera_amd"))
import torch
from seva import ops
dev = torch.device("cuda:0")
def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for side, c, ns in [(36, 640, (30, 36, 38, 39, 40, 42, 46, 50, 51)), (18, 1280, (24, 30, 36, 40, 42, 48, 54, 60, 61))]:
    w = torch.randn(c, 9 * c, device=dev, dtype=torch.float16) * (9 * c) ** -0.5
    b = torch.randn(c, device=dev)
    for n in ns:
        x = torch.randn(n, side, side, c, device=dev, dtype=torch.float16)
        res = torch.randn(n, side * side, c, device=dev)
        out = torch.empty(n, side * side, c, device=dev)
        us = timeit(lambda: ops.conv3x3(x, w, bias=b, residual=res, out_f32=out))
        M = n * side * side
        tiles = ((M + 127) // 128) * (c // 160)
        print(f"{side}x{side} c{c} n={n:3d} M={M:6d} tiles={tiles:5d} rounds={tiles / 512:5.2f} | {us:7.1f} us | {us / M * 1e3:6.3f} ns/row | {2.0 * M * c * 9 * c / us / 1e6:6.0f} TF", flush=True)
