#!/usr/bin/env python3
"""QKV projection of the C = 320 level with its LayerNorm (a) as a separate kernel, (b) in the GEMM's prologue
(seva_gemm_desc.ln_x): interleaved rounds, medians, equality check.  Run on the GPU box from the repo root."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stable-virtual-camera_amd"))
from seva import ops  # noqa: E402

dev = torch.device("cuda:0")
QS = 0.125 * 1.4426950408889634


def bench(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return sorted(ts)[len(ts) // 2]


for rows, c in [(217728, 320), (54432, 320), (217728, 256), (3000, 64)]:
    g = torch.Generator().manual_seed(1)
    x = (torch.randn(rows, c, generator=g) * 2 + 0.3).to(dev)
    gm, bt = (1 + 0.1 * torch.randn(c, generator=g)).to(dev), (0.1 * torch.randn(c, generator=g)).to(dev)
    w = (torch.randn(3 * c, c, generator=g) * c ** -0.5).half().to(dev)
    a16 = torch.empty((rows, c), device=dev, dtype=torch.float16)
    q1 = torch.empty((rows, 3 * c), device=dev, dtype=torch.float16)
    q2 = torch.empty_like(q1)

    def two():
        ops.layernorm(x, gm, bt, a16)
        ops.gemm(a16, w, out_f16=q1, col_scale=QS, col_scale_n=c)

    def fused():
        ops.gemm(None, w, out_f16=q2, col_scale=QS, col_scale_n=c, ln_x=x, ln_gamma=gm, ln_beta=bt)

    t2, tf = [], []
    for _ in range(3):
        t2.append(bench(two))
        tf.append(bench(fused))
    two(); fused(); torch.cuda.synchronize()
    d = (q1.float() - q2.float())
    print(f"rows {rows} C {c}: LayerNorm + QKV {sorted(t2)[1]:.1f} us, fused {sorted(tf)[1]:.1f} us; "
          f"rel-L2 between them {float(d.norm() / q1.float().norm()):.2e}, differing elements {float((d != 0).float().mean()):.2e}")
