#!/usr/bin/env python3
"""Workgroup-count quantisation of the attention launch: one (batch, heads) configuration, the K/V length fixed, the number of
query rows swept across a multiple of the 512 workgroup slots (256 CUs x 2).  Time per query row shows what the partly filled last
round costs."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "stable-virtual-camera_amd"))
import torch
from seva import ops
dev = torch.device("cuda:0")
B, H, LK = [int(v) for v in os.environ.get("KATTN_SHAPE", "2,10,27216").split(",")]
C = 64 * H
g = torch.Generator().manual_seed(1)
qkv = (torch.randn(B * LK, 3 * C, generator=g)).half().to(dev)
qkv[:, :C] *= 0.125 * 1.4426950408889634
out = torch.empty(B * LK, C, device=dev, dtype=torch.float16)
c3 = 3 * C
for LQ in [int(v) for v in os.environ.get("KATTN_LQ", "24576,26112,26368,27216").split(",")]:
    assert 0 < LQ <= LK, "the query rows are rows of the same buffer"
    def run():
        ops.attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], out, nb0=B, nb1=1, heads=H, lq=LQ, lk=LK,
                      q_strides=(LK * c3, 0, c3), k_strides=(LK * c3, 0, c3), o_strides=(LK * C, 0, C), q_prescaled=True)
    for _ in range(2):
        run()
    torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        t0 = time.perf_counter(); run(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e6)
    ts.sort()
    wgs = B * H * ((LQ + 255) // 256)
    print(f"B={B} H={H} lk={LK} lq={LQ:6d}: {wgs:5d} workgroups = {wgs / 512:5.2f} rounds | median {ts[3]:8.1f} us | {ts[3] * 1e3 / (B * H * LQ):7.3f} ns per query row "
          f"| {4.0 * B * H * LQ * LK * 64 / ts[3] / 1e6:6.1f} TFLOP/s", flush=True)
