#!/usr/bin/env python3
"""Re-sweep SEVA_GEMM_CHUNKS for the ASTAT / ASYNC shapes (interleaved rounds, median)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stable-virtual-camera_amd"))
import torch
from seva import ops
from seva._engine import interleave_geglu
dev = torch.device("cuda:0")
def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
CASES = [(217728, 960, 320, False), (217728, 2560, 320, True), (54432, 1920, 640, False), (54432, 5120, 640, True),
         (13608, 3840, 1280, False), (13608, 10240, 1280, True)]
CH = [0, 1, 2, 3, 4, 5, 8, 10, 20]
for M, N, K, geglu in CASES:
    a = torch.randn(M, K, device=dev, dtype=torch.float16)
    w = torch.randn(N, K, device=dev, dtype=torch.float16) * K ** -0.5
    b = torch.randn(N, device=dev)
    if geglu: w, b = interleave_geglu(w, b)
    o = torch.empty(M, N // 2 if geglu else N, device=dev, dtype=torch.float16)
    res = {c: [] for c in CH}
    for rnd in range(3):
        for c in CH:
            ops.set_knob("gemm_chunks", c if c else -1)
            res[c].append(timeit(lambda: ops.gemm(a, w, bias=b, out_f16=o, geglu=geglu)))
    tn = (N + (127 if geglu or N % 160 else 159)) // (128 if geglu or N % 160 else 160)
    print(f"{M}x{N}x{K} {'geglu' if geglu else 'plain'} tn={tn} | " + " ".join(f"c{c}:{sorted(v)[1]:6.1f}" for c, v in res.items()), flush=True)
