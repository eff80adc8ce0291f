#!/usr/bin/env python3
"""The HBM-bound K = 320 fp32-output GEMMs of the 72x72 level (proj_in, attention out-projection + residual, proj_out + residual) under
the tile knobs: which tile shape streams best when the arithmetic is a fifth of the time?  Interleaved rounds, one process."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stable-virtual-camera_amd"))
import torch
from seva import ops
dev = torch.device("cuda:0")
M, C = 42 * 72 * 72, 320
a = torch.randn(M, C, device=dev, dtype=torch.float16)
w = (torch.randn(C, C, device=dev) * 0.05).half()
b = torch.randn(C, device=dev)
res = torch.randn(M, C, device=dev)
radd = torch.randn(42, C, device=dev)
out = torch.empty(M, C, device=dev)
cases = {"proj_in f32out": dict(bias=b, out_f32=out), "attn_out +res +row_add": dict(bias=b, residual=res, row_add=radd, rows_per_group=5184, out_f32=out),
         "proj_out +res": dict(bias=b, residual=res, out_f32=out)}
knobs = [(-1, -1), (160, 160), (128, 160), (64, 160), (128, 128), (64, 128)]
def t(fn, iters=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for name, kw in cases.items():
    best = {k: 1e9 for k in knobs}
    for _ in range(3):
        for bm, bn in knobs:
            ops.set_knob("gemm_bm", bm); ops.set_knob("gemm_bn", bn)
            best[(bm, bn)] = min(best[(bm, bn)], t(lambda: ops.gemm(a, w, **kw)))
    nbytes = M * C * (2 + 4 + (4 if "residual" in kw else 0))
    print(f"{name:24s} " + "  ".join(f"bm{bm} bn{bn}: {v:6.1f} us {nbytes / v / 1e6:5.2f} TB/s" for (bm, bn), v in best.items()), flush=True)
ops.set_knob("gemm_bm", -1); ops.set_knob("gemm_bn", -1)
