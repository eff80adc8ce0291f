#!/usr/bin/env python3
"""Ablation timing of the fp32-output kernels on 160x160 tiles (knob gemm_bm = 160 + SEVA_GEMM_DBG bits of the ablation instantiation):
what the fp32 residual read and the fp32 output write cost next to the main loop.  Timing only -- ablated results are wrong by design."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stable-virtual-camera_amd"))
import torch
from seva import ops
from seva._engine import pack_conv3x3
dev = torch.device("cuda:0")
MODES = [(0, "prod"), (1024, "dbg build"), (32, "no residual"), (64, "no stores"), (256, "stores in L2"), (96, "neither"), (96 | 1 | 8, "neither, mfma only"),
         (1 | 8, "mfma only"), (2 | 8, "dma only"), (16, "no barrier")]
def timeit(fn, n=8):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
cases = [("gemm ff2 36^2", ("g", 54432, 640, 2560)), ("gemm attn_out 36^2", ("g", 54432, 640, 640)), ("gemm ff2 18^2", ("g", 13608, 1280, 5120)),
         ("conv 36^2 640->640", ("c", 42, 36, 640, 640))]
for name, c in cases:
    if c[0] == "g":
        _, M, N, K = c
        a = torch.randn(M, K, device=dev).half(); w = (torch.randn(N, K, device=dev) * 0.05).half()
        r = torch.randn(M, N, device=dev); b = torch.randn(N, device=dev); o = torch.empty_like(r)
        fn = lambda: ops.gemm(a, w, bias=b, residual=r, out_f32=o)
    else:
        _, n, side, cin, cout = c
        x = torch.randn(n, side, side, cin, device=dev).half(); w = pack_conv3x3(torch.randn(cout, cin, 3, 3, device=dev) * 0.05).half()
        r = torch.randn(n * side * side, cout, device=dev); b = torch.randn(cout, device=dev); o = torch.empty_like(r)
        fn = lambda: ops.conv3x3(x, w, bias=b, residual=r, out_f32=o)
    ops.set_knob("gemm_bm", 160)
    line = f"{name:20s}"
    for bits, label in MODES:
        ops.set_knob("gemm_dbg", bits if bits else -1)
        line += f" | {label}: {timeit(fn):6.1f}"
    ops.set_knob("gemm_dbg", -1); ops.set_knob("gemm_bm", -1)
    print(line, flush=True)
