#!/usr/bin/env python3
"""A/B a handful of GEMM shapes (incl. GEGLU) on whichever build SEVA_HIP_LIB points at."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stable-virtual-camera_amd"))
import torch
from seva import ops
dev = torch.device("cuda:0")
def timeit(fn, iters=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
SH = [(217728, 2560, 320, "geglu"), (54432, 5120, 640, "geglu"), (13608, 10240, 1280, "geglu"),
      (217728, 320, 320, "res"), (217728, 960, 320, "o16"), (54432, 640, 2560, "res")]
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
    us = timeit(fn)
    print(f"{M}x{N}x{K} {kind:6s} {us:8.1f}us {2.0*M*N*K/us/1e6:6.0f}TF", flush=True)
