#!/usr/bin/env python3
"""3 launches of each of: seva_ff_fused_f16, GEGLU GEMM, FF2 GEMM at the ds1 shape (driver for tools/pmc_kernel.sh)."""
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
hid = torch.empty((M, 4 * C), device=dev, dtype=torch.float16)
o = torch.empty((M, C), device=dev)
for _ in range(3):
    ops.ff_fused(a, wi, bi, w2, b2, residual=res, out_f32=o)
    ops.gemm(a, wi, bias=bi, out_f16=hid, geglu=True)
    ops.gemm(hid, w2, bias=b2, residual=res, out_f32=o)
torch.cuda.synchronize()
