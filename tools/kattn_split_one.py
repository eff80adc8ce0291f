#!/usr/bin/env python3
"""Joint attention at 36x36 (B=2 H=10 L=27216, q pre-scaled) with the K/V-split workspace, a few launches (for tools/pmc_kernel.sh):
attn2_kernel<64, true> + attn_combine_kernel."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stable-virtual-camera_amd"))
import torch
from seva import ops
dev = torch.device("cuda:0")
B, H, L = 2, 10, 27216
C = 64 * H
qkv = torch.randn(B * L, 3 * C, device=dev)
qkv[:, :C] *= 0.125 * 1.4426950408889634
qkv = qkv.half()
o = torch.empty(B * L, C, device=dev, dtype=torch.float16)
ws = torch.empty(ops.attention_split_workspace_numel(B, H, L), device=dev)
for _ in range(3):
    ops.attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], o, nb0=B, nb1=1, heads=H, lq=L, lk=L,
                  q_strides=(L * 3 * C, 0, 3 * C), k_strides=(L * 3 * C, 0, 3 * C), o_strides=(L * C, 0, C), q_prescaled=True,
                  split_ws=ws)
torch.cuda.synchronize()
print("done")
