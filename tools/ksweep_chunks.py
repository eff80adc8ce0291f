#!/usr/bin/env python3
"""Sweep SEVA_GEMM_CHUNKS (sibling workgroups per M-tile) over every plain/GEGLU GEMM shape of one step."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stable-virtual-camera_amd"))
import torch
from seva import ops
dev = torch.device("cuda:0")
def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
SH = []
for side, C in [(72, 320), (36, 640), (18, 1280)]:
    M = 42 * side * side
    SH += [(M, C, C, "res"), (M, 3 * C, C, "o16"), (M, 8 * C, C, "geglu"), (M, C, 4 * C, "res")]
CH = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "0,2,3,4,5,8,10,20,999").split(",")]
for M, N, K, kind in SH:
    a = torch.randn(M, K, device=dev, dtype=torch.float16)
    w = torch.randn(N, K, device=dev, dtype=torch.float16) * 0.05
    b = torch.randn(N, device=dev, dtype=torch.float32)
    if kind == "geglu":
        o = torch.empty(M, N // 2, device=dev, dtype=torch.float16)
        fn = lambda: ops.gemm(a, w, bias=b, out_f16=o, geglu=True)
    elif kind == "res":
        r = torch.randn(M, N, device=dev, dtype=torch.float32); o = torch.empty_like(r)
        fn = lambda: ops.gemm(a, w, bias=b, residual=r, out_f32=o)
    else:
        o = torch.empty(M, N, device=dev, dtype=torch.float16)
        fn = lambda: ops.gemm(a, w, bias=b, out_f16=o)
    tn = (N + 127) // 128
    line = f"{M}x{N}x{K} {kind:5s} tm={(M+127)//128} tn={tn} |"
    for c in CH:
        ops.set_knob("gemm_chunks", c)
        us = timeit(fn)
        line += f" c{c}:{us:7.1f}"
    print(line, flush=True)
