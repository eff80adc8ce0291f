#!/usr/bin/env python3
"""Driver for tools/pmc_kernel.sh: the window-staged conv (both families) and the per-tap gather on ONE shape each of the 36x36 / 18x18
levels, five launches each, so that their SQ / traffic counters can be read side by side."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stable-virtual-camera_amd"))
import torch
from seva import ops
from seva._engine import pack_conv3x3
dev = torch.device("cuda:0")
x = torch.randn(42, 36, 36, 640, device=dev).half(); wc = pack_conv3x3(torch.randn(640, 640, 3, 3, device=dev) * 0.05).half()
r = torch.randn(42 * 1296, 640, device=dev); b = torch.randn(640, device=dev); o = torch.empty_like(r)
for knob in (0, 1, 2):
    ops.set_knob("conv_win", knob)
    for _ in range(5):
        ops.conv3x3(x, wc, bias=b, residual=r, out_f32=o)
torch.cuda.synchronize()
