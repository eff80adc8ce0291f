#!/usr/bin/env python3
"""Does running producer->consumer GEMM pairs slice-by-slice over M keep the intermediate in the Infinity Cache?
ds1 feed-forward: hidden = geglu(x16 @ W1^T) [M x 1280 f16], out = hidden @ W2^T + res [M x 320 f32]."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stable-virtual-camera_amd"))
import torch
from seva import ops
from seva._engine import interleave_geglu
dev = torch.device("cuda:0")
def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for side, C in [(72, 320), (36, 640)]:
    hw = side * side
    M = 42 * hw
    x16 = torch.randn(M, C, device=dev, dtype=torch.float16)
    w1 = torch.randn(8 * C, C, device=dev, dtype=torch.float16) * C ** -0.5
    b1 = torch.randn(8 * C, device=dev)
    w1, b1 = interleave_geglu(w1, b1)
    w2 = torch.randn(C, 4 * C, device=dev, dtype=torch.float16) * (4 * C) ** -0.5
    b2 = torch.randn(C, device=dev)
    res = torch.randn(M, C, device=dev)
    out = torch.empty(M, C, device=dev)
    line = f"ds{72 // side} M={M} C={C} |"
    for frames in (42, 21, 14, 7, 6, 3, 2):
        rows = frames * hw
        hid = torch.empty(rows, 4 * C, device=dev, dtype=torch.float16)  # ONE slice-sized buffer, reused
        def run():
            for r0 in range(0, M, rows):
                r1 = min(r0 + rows, M)
                ops.gemm(x16[r0:r1], w1, bias=b1, out_f16=hid[: r1 - r0], geglu=True)
                ops.gemm(hid[: r1 - r0], w2, bias=b2, residual=res[r0:r1], out_f32=out[r0:r1])
        us = timeit(run)
        line += f" {frames}f: {us:7.1f}us"
        del hid
    print(line, flush=True)
