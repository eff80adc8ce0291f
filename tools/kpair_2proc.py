#!/usr/bin/env python3
"""Under load from a second process: (1) producer kernel -> consumer kernel on one stream, elementwise, checked exactly;
(2) groupnorm alone, (3) ff_fused alone, each repeated on fixed inputs and compared with the first result."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMMON = r'''
import os, sys
ROOT = os.environ["SEVA_ROOT"]
sys.path.insert(0, os.path.join(ROOT, "stable-virtual-camera_amd")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import torch
from seva import ops
dev = torch.device("cuda:0")
'''
LOAD = COMMON + r'''
from test_model_gpu import _build
import time
net, _ = _build("tiny", dev)
eng = net.engine(); eng.use_graph = False
T, hw = 21, 16
g = torch.Generator().manual_seed(5); n = 2 * T
x = (torch.randn(n, 4, hw, hw, generator=g) * 10).to(dev); concat = torch.randn(n, 7, hw, hw, generator=g).to(dev)
t = torch.full((n,), 700, dtype=torch.int64, device=dev); y = torch.randn(n, 1, 1024, generator=g).to(dev)
dense = torch.randn(n, 6, hw * 8, hw * 8, generator=g).to(dev)
t0 = time.time()
while time.time() - t0 < float(os.environ.get("SECS", "45")):
    for _ in range(10): eng.forward(x, concat, t, y, dense, T)
    torch.cuda.synchronize()
print("load generator done", flush=True)
'''
TEST = COMMON + r'''
from seva._engine import interleave_geglu
g = torch.Generator().manual_seed(3)
R = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dev)
res = {}
# (1) producer -> consumer
x = R(42, 4, 72, 72); two = torch.full((42,), 2.0, device=dev)
bad = 0
for i in range(2000):
    y = torch.empty_like(x); z = torch.empty_like(x)
    ops.scale_rows(x, two, y); ops.scale_rows(y, two, z)
    if i % 50 == 49:
        torch.cuda.synchronize()
        bad += int(not torch.equal(z, 4 * x))
res["scale->scale"] = bad
# (2) groupnorm variants
n, hw = 42, 256
for name, c1, c2, mod in [("gn 64", 64, 0, False), ("gn 64 mod", 64, 0, True), ("gn 256+256 (16 px)", 256, 256, True), ("gn 128 (64 px)", 128, 0, False)]:
    hwv = 16 if "16 px" in name else (64 if "64 px" in name else hw)
    C = c1 + c2
    x1 = R(n, hwv, c1, sc=3.0); x2 = R(n, hwv, c2, sc=3.0) if c2 else None
    gam, bet = R(C), R(C)
    dense, dw, db = R(n, hwv, 6), R(2 * C, 6, sc=0.1), R(2 * C, sc=0.1)
    ws = ops.groupnorm_workspace(n, dev)
    ref = None; bad = 0
    for i in range(400):
        o = torch.empty((n, hwv, C), device=dev, dtype=torch.float16)
        kw = dict(dense=dense, dense_w=dw, dense_b=db) if mod else {}
        ops.groupnorm(x1, x2, gam, bet, o, ws, silu=True, **kw)
        torch.cuda.synchronize()
        if ref is None: ref = o.clone()
        else: bad += int(not torch.equal(o, ref))
    res[name] = bad
# (3) fused feed-forward
for M, C in [(10752, 64), (2688, 128)]:
    a = R(M, C).half(); xln = R(M, C, sc=2.0)
    w1 = (R(8 * C, C) * C ** -0.5).half(); b1 = R(8 * C, sc=0.3); w2 = (R(C, 4 * C) * (4 * C) ** -0.5).half(); b2 = R(C, sc=0.3)
    gm, bt = R(C), R(C)
    wi, bi = interleave_geglu(w1, b1)
    r = R(M, C)
    ref = None; bad = 0
    for i in range(400):
        o = torch.empty((M, C), device=dev)
        ops.ff_fused(None, wi, bi, w2, b2, residual=r, out_f32=o, ln_x=xln, ln_gamma=gm, ln_beta=bt)
        torch.cuda.synchronize()
        if ref is None: ref = o.clone()
        else: bad += int(not torch.equal(o, ref))
    res[f"ff_fused {M}x{C}"] = bad
print("non-repeatable / wrong results:", res, flush=True)
'''
e = dict(os.environ, SEVA_ROOT=ROOT)
pb = subprocess.Popen([sys.executable, "-c", LOAD], env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
pa = subprocess.Popen([sys.executable, "-c", TEST], env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
oa, ea = pa.communicate(); ob, eb = pb.communicate()
print(oa.strip() or ea.strip()[-1500:]); print(ob.strip() or eb.strip()[-500:])
