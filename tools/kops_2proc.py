#!/usr/bin/env python3
"""Which OPERATOR is not repeatable while another process runs on the same card?  Each child loops single operators on fixed inputs."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys
ROOT = os.environ["SEVA_ROOT"]
sys.path.insert(0, os.path.join(ROOT, "stable-virtual-camera_amd"))
import torch
from seva import ops
from seva._engine import pack_conv3x3
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(3)
R = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dev)
n, h, w, c = 42, 16, 16, 64
hw = h * w
x32 = R(n, hw, c, sc=3.0)
gam, bet = R(c), R(c)
dense, dw, db = R(n, hw, 6), R(2 * c, 6, sc=0.1), R(2 * c, sc=0.1)
ws = ops.groupnorm_workspace(n, dev)
x16 = R(n, h, w, c).half()
wc = pack_conv3x3(R(c, c, 3, 3, sc=0.05)).half()
bias, emb, res = R(c), R(n, c), R(n * hw, c)
a16 = R(n * hw, c).half(); wl = R(c, c, sc=0.1).half()
results = {}
def check(name, fn):
    ref = None; bad = 0
    for i in range(int(os.environ.get("REPS", "300"))):
        out = fn()
        torch.cuda.synchronize()
        if ref is None: ref = [o.clone() for o in out]
        elif any(not torch.equal(o, r) for o, r in zip(out, ref)): bad += 1
    results[name] = bad
def f_gn_plain():
    o = torch.empty((n, hw, c), device=dev, dtype=torch.float16); ops.groupnorm(x32, None, gam, bet, o, ws, silu=True); return [o]
def f_gn_mod():
    o = torch.empty((n, hw, c), device=dev, dtype=torch.float16); raw = torch.empty_like(o)
    ops.groupnorm(x32, None, gam, bet, o, ws, silu=True, dense=dense, dense_w=dw, dense_b=db, raw_f16=raw); return [o, raw]
def f_conv():
    o = torch.empty((n * hw, c), device=dev); ops.conv3x3(x16, wc, bias=bias, row_add=emb, rows_per_group=hw, out_f32=o); return [o]
def f_conv_res_stats():
    o = torch.empty((n * hw, c), device=dev); st = torch.empty(ops.channel_stats_shape(n * hw, c), device=dev)
    ops.conv3x3(x16, wc, bias=bias, residual=res, out_f32=o); return [o]
def f_gemm():
    o = torch.empty((n * hw, c), device=dev); ops.gemm(a16, wl, bias=bias, residual=res, out_f32=o); return [o]
def f_ln():
    o = torch.empty((n * hw, c), device=dev, dtype=torch.float16); ops.layernorm(x32.view(n * hw, c), gam, bet, o); return [o]
for name, fn in [("groupnorm+silu", f_gn_plain), ("groupnorm+silu+mod+raw", f_gn_mod), ("conv3x3+row_add", f_conv),
                 ("conv3x3+residual", f_conv_res_stats), ("gemm+residual", f_gemm), ("layernorm", f_ln)]:
    check(name, fn)
print(os.environ.get("TAG"), "non-repeatable launches out of", os.environ.get("REPS", "300"), ":", results, flush=True)
'''
procs = []
for tag in ("procA", "procB"):
    e = dict(os.environ, SEVA_ROOT=ROOT, TAG=tag)
    procs.append(subprocess.Popen([sys.executable, "-c", CHILD], env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
for p in procs:
    o, er = p.communicate()
    print(o.strip() or er.strip()[-800:], flush=True)
