#!/usr/bin/env python3
"""ds1 f16-only K=320 shapes under SEVA_GEMM_BN / SEVA_GEMM_ASTAT (which variant wins where)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stable-virtual-camera_amd"))
import torch
from seva import ops
from seva._engine import interleave_geglu
dev = torch.device("cuda:0")
def timeit(fn, iters=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
M, K = 217728, 320
a = torch.randn(M, K, device=dev, dtype=torch.float16)
for N, geglu in [(960, False), (2560, True)]:
    w = torch.randn(N, K, device=dev, dtype=torch.float16) * K ** -0.5
    b = torch.randn(N, device=dev)
    if geglu: w, b = interleave_geglu(w, b)
    o = torch.empty(M, N // 2 if geglu else N, device=dev, dtype=torch.float16)
    cfgs = [(bn, ast) for bn in ("128", "160") for ast in ("0", "1")]
    res = {c: [] for c in cfgs}
    for rnd in range(4):  # interleaved rounds: clock / thermal drift hits every config alike
        for bn, ast in cfgs:
            ops.set_knob("gemm_bn", int(bn)); ops.set_knob("gemm_astat", int(ast))
            res[(bn, ast)].append(timeit(lambda: ops.gemm(a, w, bias=b, out_f16=o, geglu=geglu, col_scale=0.18 if not geglu else 1.0, col_scale_n=320 if not geglu else 0)))
    line = f"{M}x{N}x{K} {'geglu' if geglu else 'qkv  '}"
    for c in cfgs:
        r = sorted(res[c])
        line += f" | bn{c[0]} astat{c[1]}: min {r[0]:6.1f} med {r[len(r)//2]:6.1f}"
    print(line, flush=True)
