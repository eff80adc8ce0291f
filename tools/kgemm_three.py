#!/usr/bin/env python3
"""Driver for tools/pmc_kernel.sh: the three GEMM-class kernels that carry most of a step, one shape each, five launches each --
fp32-output GEMM on 160x160 tiles (ff2 + residual, 36x36 level), implicit-GEMM 3x3 conv on 160x160 tiles (36x36, 640 -> 640),
GEGLU on 160x128 tiles (36x36 level)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stable-virtual-camera_amd"))
import torch
from seva import ops
from seva._engine import pack_conv3x3
dev = torch.device("cuda:0")
M = 54432
a = torch.randn(M, 2560, device=dev).half(); w = (torch.randn(640, 2560, device=dev) * 0.05).half()
r = torch.randn(M, 640, device=dev); b = torch.randn(640, device=dev); o = torch.empty_like(r)
x = torch.randn(42, 36, 36, 640, device=dev).half(); wc = pack_conv3x3(torch.randn(640, 640, 3, 3, device=dev) * 0.05).half()
a2 = torch.randn(M, 640, device=dev).half(); wg = (torch.randn(5120, 640, device=dev) * 0.05).half(); bg = torch.randn(5120, device=dev)
og = torch.empty(M, 2560, device=dev, dtype=torch.float16)
for _ in range(5):
    ops.gemm(a, w, bias=b, residual=r, out_f32=o)
    ops.conv3x3(x, wc, bias=b, residual=r, out_f32=o)
    ops.gemm(a2, wg, bias=bg, out_f16=og, geglu=True)
torch.cuda.synchronize()
