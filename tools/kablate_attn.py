#!/usr/bin/env python3
"""Attention ablation timing: SEVA_ATTN_DBG bits (1 no K/V reloads, 2 no softmax, 4 no P*V, 8 no Q*K)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stable-virtual-camera_amd"))
import torch
from seva import ops
dev = torch.device("cuda:0")
def timeit(fn, n=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
MODES = [(0, "full"), (1, "no kv reload"), (2, "no softmax"), (4, "no PV"), (8, "no QK"), (12, "no mfma"), (14, "loads+lds only"), (15, "skeleton"), (3, "mfma only(ish)")]
for B, H, L in [(2, 10, 27216), (42, 5, 5184)]:
    C = 64 * H
    qkv = torch.randn(B * L, 3 * C, device=dev, dtype=torch.float16)
    o = torch.empty(B * L, C, device=dev, dtype=torch.float16)
    line = f"B{B} H{H} L{L}"
    for bits, name in MODES:
        ops.set_knob("attn_dbg", bits if bits else -1)
        us = timeit(lambda: ops.attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], o, nb0=B, nb1=1, heads=H, lq=L, lk=L,
                                          q_strides=(L * 3 * C, 0, 3 * C), k_strides=(L * 3 * C, 0, 3 * C), o_strides=(L * C, 0, C)))
        line += f" | {name}: {us:7.0f}us {4.0 * B * H * L * L * 64 / us / 1e6:5.0f}TF"
    print(line, flush=True)
