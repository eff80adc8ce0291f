#!/usr/bin/env python3
"""GEMM ablation timing (default 128x128 kernel): SEVA_GEMM_DBG bits -> which resource bounds each shape."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stable-virtual-camera_amd"))
import torch
from seva import ops
dev = torch.device("cuda:0")
SHAPES = [(13608, 1280, 5120, "o16"), (54432, 1920, 640, "o16"), (13608, 3840, 1280, "o16")]
# 1024 = dbg build with every bit off (its codegen differs from the production instantiation)
MODES = [(0, "prod"), (1024, "dbg build"), (8, "no frag reads"), (1, "no dma"), (1 | 8, "mfma only"), (2 | 8, "dma only"), (2, "no mfma"), (16, "no barrier")]
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for M, N, K, fl in SHAPES:
    a = torch.randn(M, K, device=dev, dtype=torch.float16)
    w = torch.randn(N, K, device=dev, dtype=torch.float16) * K ** -0.5
    o16 = torch.empty(M, N, device=dev, dtype=torch.float16) if fl == "o16" else None
    o32 = torch.empty(M, N, device=dev) if fl != "o16" else None
    res = torch.randn(M, N, device=dev) if fl == "o32res" else None
    line = f"{M}x{N}x{K} {fl:7s}"
    for bits, name in MODES:
        ops.set_knob("gemm_dbg", bits if bits else -1)
        us = timeit(lambda: ops.gemm(a, w, residual=res, out_f32=o32, out_f16=o16))
        line += f" | {name}: {us:6.1f}"
    print(line, flush=True)
