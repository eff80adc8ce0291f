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
if os.environ.get("INPROC_LOAD", "0") == "1":
    import threading, time
    from test_model_gpu import _build
    def _load():
        s2 = torch.cuda.Stream()
        with torch.cuda.stream(s2):
            net, _ = _build("tiny", dev)
            eng = net.engine(); eng.use_graph = False
            T, hw_ = 21, 16
            g2 = torch.Generator().manual_seed(5); n2 = 2 * T
            x = (torch.randn(n2, 4, hw_, hw_, generator=g2) * 10).to(dev); concat = torch.randn(n2, 7, hw_, hw_, generator=g2).to(dev)
            t = torch.full((n2,), 700, dtype=torch.int64, device=dev); y = torch.randn(n2, 1, 1024, generator=g2).to(dev)
            dense = torch.randn(n2, 6, hw_ * 8, hw_ * 8, generator=g2).to(dev)
            t0 = time.time()
            while time.time() - t0 < float(os.environ.get("SECS", "60")):
                for _ in range(10): eng.forward(x, concat, t, y, dense, T)
                s2.synchronize()
        print("in-process load generator (second stream) done", flush=True)
    if os.environ.get("NOLOAD", "0") != "1":
        threading.Thread(target=_load, daemon=True).start()
        time.sleep(8)

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
          raw2 = torch.empty((n, hw, c), device=dev, dtype=torch.float16)
          ops.groupnorm(x32, None, gam, bet, o2, ws, silu=SILU, dense=dense_buf, dense_w=dw, dense_b=db, raw_f16=raw2)
          torch.cuda.synchronize()
          cnt['raw copy of x wrong'] = cnt.get('raw copy of x wrong', 0) + int(not torch.equal(raw2, x32.half()))
          outs.append((seen, o1.clone(), o2.clone()))
      ne = lambda a, b: not torch.equal(a.view(torch.int32) if a.dtype == torch.float32 else a.view(torch.int16),
                                        b.view(torch.int32) if b.dtype == torch.float32 else b.view(torch.int16))
      b0, b1, b2 = ne(outs[0][0], outs[1][0]), ne(outs[0][1], outs[1][1]), ne(outs[0][2], outs[1][2])
      cnt['two synchronised launches differ'] = cnt.get('two synchronised launches differ', 0) + int(ne(outs[1][1], outs[2][1])) + int(ne(outs[1][2], outs[2][2]))
      cnt['two back-to-back launches differ'] = cnt.get('two back-to-back launches differ', 0) + int(ne(outs[0][1], outs[3][1])) + int(ne(outs[0][2], outs[3][2]))
      cnt["bilinear output seen by a torch clone"] += int(b0); cnt["groupnorm after bilinear"] += int(b1); cnt["groupnorm after a torch copy_"] += int(b2)
      if (b0 or b1 or b2) and shown < 8:
          shown += 1
          if b0:
              d = (outs[0][0].view(torch.int32) != outs[1][0].view(torch.int32)).flatten().nonzero().flatten()
              print(f"it {it}: clone sees {d.numel()} wrong floats at flat offsets {d[:12].tolist()} ... values {outs[0][0].flatten()[d[:6]].tolist()} expected {outs[1][0].flatten()[d[:6]].tolist()}", flush=True)
          if b1 or b2:
              which = 1 if b1 else 2
              pre_d = outs[1][0] if which == 1 else pre          # the modulation input of that launch
              xg = x32.view(n, hw, 32, 2)
              mean = xg.mean((1, 3), keepdim=True); var = xg.var((1, 3), unbiased=False, keepdim=True)
              ypre = ((xg - mean) * torch.rsqrt(var + 1e-5)).view(n, hw, c) * gam + bet
              mod = pre_d @ dw.t() + db
              ex = torch.nn.functional.silu(ypre) * (1 + mod[..., :c]) + mod[..., c:]
              for leg in range(4):
                  o = outs[leg][which].float()
                  wrong = ((o - ex).abs() > 0.02 + 0.004 * ex.abs()).nonzero()
                  for s_, p_, c_ in wrong[:3].tolist():
                      got = float(o[s_, p_, c_]); ms = float(mod[s_, p_, c_]); mh = float(mod[s_, p_, c + c_]); yp = float(ypre[s_, p_, c_])
                      ys = (got - mh) / (1 + ms)
                      sig = ys / yp if yp else float('nan')
                      import math
                      others = {f"r{(c_ & ~3) + k}": 1 / (1 + math.exp(-float(ypre[s_, p_, (c_ & ~3) + k]))) for k in range(4)}
                      prevs = {f"pix{d:+d}": 1 / (1 + math.exp(-float(ypre[s_, p_ + d, c_]))) for d in (-48, -32, -16, 16, 32, 48) if 0 <= p_ + d < hw}
                      print(f"it {it} launch {which} leg {leg}: (sample {s_}, pixel {p_}, channel {c_}) got {got:.4f} want {float(ex[s_, p_, c_]):.4f}; y before silu {yp:.4f}; implied silu(y) {ys:.4f}, "
                            f"implied sigmoid {sig:.4f} vs true {1 / (1 + math.exp(-yp)):.4f}; sigmoid of the quad's channels {({k: round(v, 4) for k, v in others.items()})}; of the thread's other pixels {({k: round(v, 4) for k, v in prevs.items()})}", flush=True)
          if b1:
              d = (outs[0][1] != outs[1][1]) | torch.isnan(outs[0][1])
              px = d.any(-1).nonzero()
              ch = d.any(0).any(0).nonzero().flatten()
              print(f"it {it}: groupnorm output wrong at {px.shape[0]} pixels {px[:6].tolist()}, channels {ch[:16].tolist()} ({ch.numel()}), NaNs {int(torch.isnan(outs[0][1]).sum())}; clone wrong too: {b0}", flush=True)
  print("kernel variant", kv, "silu", SILU, VARIANT, "iterations whose back-to-back result differs from the synchronised one:", cnt, flush=True)
"""
e = dict(os.environ, SEVA_ROOT=ROOT)
pb = subprocess.Popen([sys.executable, "-c", LOAD if os.environ.get("INPROC_LOAD", "0") != "1" else "print('no second process')"], env=e)   # children print straight to this process's stdout (line by line)
pa = subprocess.Popen([sys.executable, "-c", TEST], env=e)
pa.wait(); pb.wait()
