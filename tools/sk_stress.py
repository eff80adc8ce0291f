#!/usr/bin/env python3
"""Stress of the workgroup-to-workgroup hand-off (split-K conv at the 9x9 level = the default path, and stream-K): many launches
with changing data through one workspace, every result compared bitwise (stream-K) / to 2e-6 (split-K) with the unsplit kernel."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stable-virtual-camera_amd"))
import torch
from seva import ops
from seva._engine import pack_conv3x3
dev = torch.device("cuda:0")
ws = ops.splitk_workspace(0, 0, dev)
ROUNDS = int(os.environ.get("SK_STRESS_ROUNDS", "40"))
def mk(seed, n, side, cin, cout, stride=1):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, side, side, cin, generator=g).half().to(dev)
    w = pack_conv3x3(torch.randn(cout, cin, 3, 3, generator=g) * 0.05).half().to(dev)
    oh = (side - 1) // stride + 1
    M = n * oh * oh
    return x, w, torch.randn(cout, generator=g).to(dev), torch.randn(M, cout, generator=g).to(dev), M, cout, stride
splitk = [mk(s, 42, 9, 1280, 1280) for s in (1, 2, 3)] + [mk(4, 42, 18, 640, 1280, 2), mk(5, 7, 9, 2560, 1280), mk(6, 3, 8, 128, 128)]
streamk = [mk(11, 16, 72, 320, 320), mk(12, 16, 72, 64, 320), mk(13, 42, 18, 1280, 1280), mk(14, 42, 36, 640, 640)]
def run(c, wsp):
    x, w, b, r, M, N, s = c
    o = torch.full((M, N), float("nan"), device=dev)
    ops.conv3x3(x, w, stride=s, bias=b, residual=r, out_f32=o, splitk_ws=wsp)
    return o
ref_split = [run(c, None) for c in splitk]
ops.set_knob("gemm_streamk", 0)
ref_stream = [run(c, None) for c in streamk]
bad = 0
for rnd in range(ROUNDS):
    ops.set_knob("gemm_streamk", 0)
    outs = [run(c, ws) for c in splitk]            # back-to-back launches, no host sync in between
    ops.set_knob("gemm_streamk", 1)
    outs2 = [run(c, ws) for c in streamk]
    torch.cuda.synchronize()
    for i, (o, r) in enumerate(zip(outs, ref_split)):
        e = float((o - r).norm() / r.norm())
        if not (e < 2e-6):
            bad += 1; print(f"round {rnd} split-K case {i}: rel-L2 {e:.3e}")
    for i, (o, r) in enumerate(zip(outs2, ref_stream)):
        if not torch.equal(o, r):
            bad += 1; print(f"round {rnd} stream-K case {i}: max diff {float((o - r).abs().max()):.3e}")
print(f"{ROUNDS} rounds x {len(splitk) + len(streamk)} launches: {bad} mismatches; flags nonzero: {int(ws[:16384].view(torch.int32).abs().sum())}")
