#!/usr/bin/env python3
"""GEGLU GEMM: 128x128 tiles (default) vs 160x128 tiles (knob gemm_bm = 160) at the ds2 / ds4 / ds8 shapes; checks equality."""
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
for M, C in ((54432, 640), (13608, 1280), (3402, 1280), (217728, 320)):
    a = torch.randn(M, C, device=dev).half(); w = (torch.randn(8 * C, C, device=dev) * 0.05).half(); b = torch.randn(8 * C, device=dev)
    o0 = torch.empty(M, 4 * C, device=dev, dtype=torch.float16); o1 = torch.empty_like(o0)
    line = f"geglu M={M} C={C}:"
    for rnd in range(2):
        for bm, o in ((-1, o0), (160, o1)):
            ops.set_knob("gemm_bm", bm)
            us = timeit(lambda: ops.gemm(a, w, bias=b, out_f16=o, geglu=True))
            line += f" | bm={bm}: {us:7.1f} us {2.0 * M * 8 * C * C / us / 1e6:6.1f} TF"
    ops.set_knob("gemm_bm", -1)
    print(line, "| equal:", bool(torch.equal(o0, o1)), flush=True)
