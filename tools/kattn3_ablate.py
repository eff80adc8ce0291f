#!/usr/bin/env python3
"""One attention shape (ds2 joint by default), timed; the library comes from SEVA_HIP_LIB.  Used with the compile-time
ablation builds of attn3_kernel (-DSEVA_ATTN3_ABL=bits; results are wrong by design, timing only)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "stable-virtual-camera_amd"))
import torch
from seva import ops
dev = torch.device("cuda:0")
B, H, L = [int(v) for v in os.environ.get("KATTN_SHAPE", "2,10,27216").split(",")]
C = 64 * H
g = torch.Generator().manual_seed(1)
qkv = (torch.randn(B * L, 3 * C, generator=g)).half().to(dev)
qkv[:, :C] *= 0.125 * 1.4426950408889634
out = torch.empty(B * L, C, device=dev, dtype=torch.float16)
c3 = 3 * C
def run():
    ops.attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], out, nb0=B, nb1=1, heads=H, lq=L, lk=L,
                  q_strides=(L * c3, 0, c3), k_strides=(L * c3, 0, c3), o_strides=(L * C, 0, C), q_prescaled=True)
for _ in range(2):
    run()
torch.cuda.synchronize()
ts = []
for _ in range(5):
    t0 = time.perf_counter(); run(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e6)
ts.sort()
fl = 4.0 * B * H * L * L * 64
print(f"{os.environ.get('SEVA_HIP_LIB', 'default'):40s} attn_two={os.environ.get('SEVA_ATTN_TWO', '-')}: median {ts[2]:9.1f} us = {fl / ts[2] / 1e6:7.1f} TFLOP/s")
