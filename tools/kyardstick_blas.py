#!/usr/bin/env python3
"""Yardstick only (nothing of the product path calls a BLAS library): torch.matmul = hipBLASLt / rocBLAS on the GEMM shapes of a
denoising step, f16 in / f16 out, no epilogue.  Says what a vendor-tuned main loop reaches on this chip at these shapes, i.e. how much
of the gap to the 2.5 PFLOP/s peak is the shapes' (short K) and how much is the kernels'.

    python tools/kyardstick_blas.py            # us and TFLOP/s per shape
    rocprofv3 --kernel-trace --stats -d out -- python3 tools/kyardstick_blas.py   # kernel names = the tile configurations chosen
"""
import torch

dev = torch.device("cuda:0")
SHAPES = [("ds1 qkv", 217728, 960, 320), ("ds1 geglu", 217728, 2560, 320), ("ds1 ff2", 217728, 320, 1280),
          ("ds2 qkv", 54432, 1920, 640), ("ds2 geglu", 54432, 5120, 640), ("ds2 ff2", 54432, 640, 2560), ("ds2 attn_out", 54432, 640, 640),
          ("ds4 qkv", 13608, 3840, 1280), ("ds4 geglu", 13608, 10240, 1280), ("ds4 ff2", 13608, 1280, 5120), ("ds4 attn_out", 13608, 1280, 1280),
          ("square 8192", 8192, 8192, 8192)]


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


print("== torch.matmul (vendor BLAS), f16, A [M,K] row-major x W [N,K]^T: name M N K | us | TFLOP/s", flush=True)
for name, M, N, K in SHAPES:
    a = torch.randn(M, K, device=dev, dtype=torch.float16)
    w = torch.randn(N, K, device=dev, dtype=torch.float16) * 0.05
    out = torch.empty(M, N, device=dev, dtype=torch.float16)
    us = min(timeit(lambda: torch.matmul(a, w.t(), out=out)) for _ in range(3))
    print(f"{name:14s} {M:7d} {N:6d} {K:5d} | {us:9.1f} | {2.0 * M * N * K / us / 1e6:7.1f}", flush=True)
