#!/usr/bin/env python3
"""Repeatability of the fused feed-forward (LayerNorm prologue + GEGLU + FF2 + residual) on fixed inputs while a second stream of the
same process runs the tiny UNet: per channel count, launches whose output differs from the first one, and where."""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stable-virtual-camera_amd")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import torch
from test_model_gpu import _build
from seva import ops
from seva._engine import interleave_geglu

dev = torch.device("cuda:0")
stop = False


def load():
    s2 = torch.cuda.Stream()
    with torch.cuda.stream(s2):
        net, _ = _build("tiny", dev)
        eng = net.engine(); eng.use_graph = False
        T, hw = 21, 16
        g = torch.Generator().manual_seed(11); n = 2 * T
        a = ((torch.randn(n, 4, hw, hw, generator=g) * 10).to(dev), torch.randn(n, 7, hw, hw, generator=g).to(dev),
             torch.full((n,), 700, dtype=torch.int64, device=dev), torch.randn(n, 1, 1024, generator=g).to(dev),
             torch.randn(n, 6, hw * 8, hw * 8, generator=g).to(dev))
        while not stop:
            for _ in range(10): eng.forward(*a, T)
            s2.synchronize()


th = None
if os.environ.get("LOAD", "1") == "1":
    th = threading.Thread(target=load); th.start(); time.sleep(10)
g = torch.Generator().manual_seed(3)
R = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dev)
M = 10752
for c in [int(v) for v in os.environ.get("CS", "64,128,256,320").split(",")]:
    w1, b1 = interleave_geglu(R(8 * c, c, sc=c ** -0.5), R(8 * c, sc=0.1))
    w1 = w1.half().contiguous(); w2 = R(c, 4 * c, sc=(4 * c) ** -0.5).half(); b2 = R(c, sc=0.1)
    x = R(M, c); gam, bet = R(c), R(c); res = R(M, c)
    for mode in ("ln prologue", "f16 operand"):
        a16 = torch.nn.functional.layer_norm(x, (c,), gam, bet).half()
        ref = None; bad = 0; shown = 0
        for it in range(int(os.environ.get("REPS", "300"))):
            o = torch.empty((M, c), device=dev)
            if mode == "ln prologue": ops.ff_fused(None, w1, b1, w2, b2, residual=res, out_f32=o, ln_x=x, ln_gamma=gam, ln_beta=bet)
            else: ops.ff_fused(a16, w1, b1, w2, b2, residual=res, out_f32=o)
            torch.cuda.synchronize()
            if ref is None: ref = o.clone(); continue
            if not torch.equal(o, ref):
                bad += 1
                if shown < 3:
                    shown += 1
                    d = (o != ref)
                    rows = d.any(1).nonzero().flatten(); cols = d.any(0).nonzero().flatten()
                    print(f"  C={c} {mode} launch {it}: {int(d.sum())} values differ, rows {rows[:20].tolist()} ({rows.numel()}), cols {cols[:20].tolist()} ({cols.numel()}), max |diff| {float((o - ref).abs().max()):.3e}", flush=True)
        print(f"C={c} {mode}: launches differing from the first {bad} of {int(os.environ.get('REPS', '300')) - 1}", flush=True)
stop = True
if th: th.join()
