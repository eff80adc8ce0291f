#!/usr/bin/env python3
"""Segment durations of the ping-pong attention kernel from in-kernel s_memtime stamps (DIAGNOSTIC build only):

    make -C stable-virtual-camera_amd/csrc variant NAME=stamp EXTRA=-DSEVA_ATTN_STAMP
    SEVA_HIP_LIB=build_ab/libseva_hip_stamp.so python tools/kattn_stamps.py

Workgroup 0, tiles 8..15, per wave: M = PV(t-1) + scores(t) until the scores are written; wait = own DMA pieces + barrier;
V = softmax until the probabilities are written; bar = the closing barrier.  Cycles of the shader clock."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stable-virtual-camera_amd"))
import torch  # noqa: E402

from seva import ops  # noqa: E402

dev = torch.device("cuda:0")
B, H, L = 8, 8, 4096
C = 64 * H
qkv = torch.randn(B * L, 3 * C, device=dev)
qkv[:, :C] *= 0.125 * 1.4426950408889634
qkv = qkv.half()
o = torch.empty(B * L, C, device=dev, dtype=torch.float16)
ws = torch.zeros(ops.attention_split_workspace_numel(B, H, L), device=dev)
ops.set_knob("attn_two", 4)
ops.set_knob("attn_split", 0)
for it in range(30):  # the chip under load before the launch that is read
    ops.attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], o, nb0=B, nb1=1, heads=H, lq=L, lk=L, q_strides=(L * 3 * C, 0, 3 * C),
                  k_strides=(L * 3 * C, 0, 3 * C), o_strides=(L * C, 0, C), q_prescaled=True, split_ws=ws)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for it in range(10):
    ops.attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], o, nb0=B, nb1=1, heads=H, lq=L, lk=L, q_strides=(L * 3 * C, 0, 3 * C),
                  k_strides=(L * 3 * C, 0, 3 * C), o_strides=(L * C, 0, C), q_prescaled=True, split_ws=ws)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 100
print(f"this build: {us:.1f} us per launch = {us / 2 / 64 * 1e3:.0f} ns per tile of a workgroup (512 workgroups = 2 rounds of 256 CUs, 64 tiles)")
st = ws.view(torch.int64)[: 8 * 8 * 8].view(8, 8, 8).cpu()
if int(st.abs().max()) == 0:
    print("no stamps: not the -DSEVA_ATTN_STAMP build")
    sys.exit(1)
print("wave | per tile 8..15: M / wait+bar / V / bar  (cycles)")
for w in range(8):
    cells = []
    for t in range(8):
        s = st[w, t]
        cells.append(f"{int(s[1] - s[0]):5d}/{int(s[2] - s[1]):5d}/{int(s[3] - s[2]):5d}/{int(s[4] - s[3]):5d}")
    print(f"w{w} | " + "  ".join(cells))
t0 = st[:, :, 0]
print("tile period (cycles, wave 0):", [int(t0[0, i + 1] - t0[0, i]) for i in range(7)])
print("group skew (wave 4 M start - wave 0 M start):", [int(t0[4, i] - t0[0, i]) for i in range(8)])
