#!/usr/bin/env python3
"""Under load from a second process: does a consumer kernel ever see stale data of its producer on ONE stream?  Data change
every iteration, each pipeline runs once back-to-back and once with a device sync between the stages; results must be equal.
(a) torch-only pipeline, (b) HIP library: bilinear resize -> modulated GroupNorm, (c) conv3x3 -> GroupNorm."""
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
while time.time() - t0 < float(os.environ.get("SECS", "60")):
    for _ in range(10): eng.forward(x, concat, t, y, dense, T)
    torch.cuda.synchronize()
print("load generator done", flush=True)
'''
TEST = COMMON + r"""
g = torch.Generator().manual_seed(3)
R = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dev)
n, h, w, c = 42, 16, 16, 64
hw = h * w
gam, bet = R(c), R(c)
dw, db = R(2 * c, 6, sc=0.1), R(2 * c, sc=0.1)
ws = ops.groupnorm_workspace(n, dev)
x32 = R(n, hw, c, sc=3.0)
dense_buf = torch.empty((n, hw, 6), device=dev)
cnt = {"bilinear output seen by a torch clone": 0, "groupnorm after bilinear": 0, "groupnorm after a torch copy_": 0}
VARIANT = os.environ.get("VARIANT", "fill")
SILU = os.environ.get("SILU", "1") == "1"
for kv in [int(v) for v in os.environ.get("KVARIANTS", "0").split(",")]:
  ops.set_knob("gn_min_iter", 1000 * kv if kv else -1)
  shown = 0
  for k in cnt: cnt[k] = 0
  for it in range(int(os.environ.get("ITERS", "400"))):
      src = R(n, 6, 128, 128)
      pre = R(n, hw, 6)
      outs = []
      for sync in (False, True, True, False):
          if VARIANT == "fill": dense_buf.fill_(float("nan"))
          ops.bilinear_to_nhwc(src, dense_buf, h, w)
          if sync: torch.cuda.synchronize()
          seen = dense_buf.clone()
          if sync: torch.cuda.synchronize()
          o1 = torch.empty((n, hw, c), device=dev, dtype=torch.float16)
          ops.groupnorm(x32, None, gam, bet, o1, ws, silu=SILU, dense=dense_buf, dense_w=dw, dense_b=db)
          torch.cuda.synchronize()
          if VARIANT == "fill": dense_buf.fill_(float("nan"))
          dense_buf.copy_(pre)
          if sync: torch.cuda.synchronize()
          o2 = torch.empty((n, hw, c), device=dev, dtype=torch.float16)
          ops.groupnorm(x32, None, gam, bet, o2, ws, silu=SILU, dense=dense_buf, dense_w=dw, dense_b=db)
          torch.cuda.synchronize()
          outs.append((seen, o1.clone(), o2.clone()))
      ne = lambda a, b: not torch.equal(a.view(torch.int32) if a.dtype == torch.float32 else a.view(torch.int16),
                                        b.view(torch.int32) if b.dtype == torch.float32 else b.view(torch.int16))
      b0, b1, b2 = ne(outs[0][0], outs[1][0]), ne(outs[0][1], outs[1][1]), ne(outs[0][2], outs[1][2])
      cnt['two synchronised launches differ'] = cnt.get('two synchronised launches differ', 0) + int(ne(outs[1][1], outs[2][1])) + int(ne(outs[1][2], outs[2][2]))
      cnt['two back-to-back launches differ'] = cnt.get('two back-to-back launches differ', 0) + int(ne(outs[0][1], outs[3][1])) + int(ne(outs[0][2], outs[3][2]))
      cnt["bilinear output seen by a torch clone"] += int(b0); cnt["groupnorm after bilinear"] += int(b1); cnt["groupnorm after a torch copy_"] += int(b2)
      if (b0 or b1) and shown < 6:
          shown += 1
          if b0:
              d = (outs[0][0].view(torch.int32) != outs[1][0].view(torch.int32)).flatten().nonzero().flatten()
              print(f"it {it}: clone sees {d.numel()} wrong floats at flat offsets {d[:12].tolist()} ... values {outs[0][0].flatten()[d[:6]].tolist()} expected {outs[1][0].flatten()[d[:6]].tolist()}", flush=True)
          if b1:
              d = (outs[0][1] != outs[1][1]) | torch.isnan(outs[0][1])
              px = d.any(-1).nonzero()
              ch = d.any(0).any(0).nonzero().flatten()
              print(f"it {it}: groupnorm output wrong at {px.shape[0]} pixels {px[:6].tolist()}, channels {ch[:16].tolist()} ({ch.numel()}), NaNs {int(torch.isnan(outs[0][1]).sum())}; clone wrong too: {b0}", flush=True)
  print("kernel variant", kv, "silu", SILU, VARIANT, "iterations whose back-to-back result differs from the synchronised one:", cnt, flush=True)
"""
e = dict(os.environ, SEVA_ROOT=ROOT)
pb = subprocess.Popen([sys.executable, "-c", LOAD], env=e)   # children print straight to this process's stdout (line by line)
pa = subprocess.Popen([sys.executable, "-c", TEST], env=e)
pa.wait(); pb.wait()
