#!/usr/bin/env python3
"""Which layer of the engine (if any) gives a half CFG batch different bits than the same rows inside the full batch?
Tiny UNet at the shapes of tests/test_pipeline_gpu.py (T=21, 16x16) and a few others; compares every `out:<prefix>` buffer."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stable-virtual-camera_amd")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import torch
from test_model_gpu import _build
dev = torch.device("cuda:0")
net, _ = _build("tiny", dev)
eng = net.engine()
eng.use_graph = False
for T, hw in [(21, 16), (21, 8), (4, 32), (21, 24)]:
    g = torch.Generator().manual_seed(5)
    n = 2 * T
    x = torch.randn(n, 4, hw, hw, generator=g).to(dev)
    concat = torch.randn(n, 7, hw, hw, generator=g).to(dev)
    t = torch.full((n,), 700, dtype=torch.int64, device=dev)
    y = torch.randn(n, 1, 1024, generator=g).to(dev)
    dense = torch.randn(n, 6, hw * 8, hw * 8, generator=g).to(dev)
    full = eng.forward(x, concat, t, y, dense, T).clone()
    snap = {k[0]: v.clone() for k, v in eng.arena.bufs.items() if k[0].startswith("out:") and k[1][0] == n}
    bad = []
    for h, sl in ((0, slice(0, T)), (1, slice(T, n))):
        half = eng.forward(x[sl], concat[sl], t[sl], y[sl], dense[sl], T).clone()
        for k, v in eng.arena.bufs.items():
            if k[0] in snap and k[1][0] == T:
                ref = snap[k[0]][sl]
                if not torch.equal(v, ref):
                    bad.append((h, k[0], float((v - ref).abs().max())))
        print(f"T={T} hw={hw} half {h}: output equal = {torch.equal(half, full[sl])}")
    seen = set()
    for h, name, d in bad:
        if name not in seen:
            seen.add(name)
            print("   first differing buffers:", name, f"max diff {d:.3e}")
        if len(seen) >= 6:
            break
