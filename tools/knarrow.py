#!/usr/bin/env python3
"""The HBM-bound GEMMs of the 72^2 level (N = K = 320; fp32 residual in, fp32 out: 697 MB per launch) under every tile shape the template has."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stable-virtual-camera_amd"))
import torch
from seva import ops
dev = torch.device("cuda:0")
def timeit(fn, iters=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
M, N, K = 217728, 320, 320
a = torch.randn(M, K, device=dev).half(); w = (torch.randn(N, K, device=dev) * 0.05).half()
r = torch.randn(M, N, device=dev); b = torch.randn(N, device=dev)
cases = {"attn_out +res (in place)": lambda o: ops.gemm(a, w, bias=b, residual=o, out_f32=o),
         "attn_out +res (separate)": lambda o: ops.gemm(a, w, bias=b, residual=r, out_f32=o),
         "proj_in f32 out": lambda o: ops.gemm(a, w, bias=b, out_f32=o)}
byt = {"attn_out +res (in place)": M * K * 2 + 2 * M * N * 4, "attn_out +res (separate)": M * K * 2 + 2 * M * N * 4, "proj_in f32 out": M * K * 2 + M * N * 4}
for name, fn in cases.items():
    o = torch.zeros(M, N, device=dev)
    for rnd in range(2):
        line = f"{name:26s}"
        for bm in (64, 128, 160):
            for bn in (128, 160):
                ops.set_knob("gemm_bm", bm); ops.set_knob("gemm_bn", bn)
                us = timeit(lambda: fn(o))
                line += f" | {bm}x{bn}: {us:6.1f} us {byt[name] / us / 1e6:5.2f} TB/s"
        ops.set_knob("gemm_bm", -1); ops.set_knob("gemm_bn", -1)
        us = timeit(lambda: fn(o))
        print(line + f" | default {us:6.1f} us", flush=True)
