#!/usr/bin/env python3
"""Time of seva_ff_fused_f16 at the ds1 shape (217728 rows, C = 320, LayerNorm prologue as the engine calls it); for A/B builds via SEVA_HIP_LIB."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "stable-virtual-camera_amd"))
import torch
from seva import ops
from seva._engine import interleave_geglu
dev = torch.device("cuda:0")
M, C = 217728, 320
g = torch.Generator().manual_seed(0)
a = torch.randn(M, C, generator=g).half().to(dev)
w1 = (torch.randn(8 * C, C, generator=g) * C ** -0.5).half().to(dev)
b1 = (0.1 * torch.randn(8 * C, generator=g)).to(dev)
w2 = (torch.randn(C, 4 * C, generator=g) * (4 * C) ** -0.5).half().to(dev)
b2 = (0.1 * torch.randn(C, generator=g)).to(dev)
res = torch.randn(M, C, generator=g).to(dev)
wi, bi = interleave_geglu(w1, b1)
o = torch.empty((M, C), device=dev)
best = 1e30
for r in range(5):
    ops.ff_fused(a, wi, bi, w2, b2, residual=res, out_f32=o)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.ff_fused(a, wi, bi, w2, b2, residual=res, out_f32=o)
    e1.record()
    torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) * 100)
print(f"ff_fused ds1: {best:.1f} us, checksum {float(o.double().sum()):.6e}")
