#!/usr/bin/env python3
"""Square/large-K GEMMs: compare the tile configurations (SEVA_GEMM_CFG) on the shapes the CDNA4 guide quotes."""
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
SH = [(217728, 960, 320), (217728, 2560, 320), (54432, 1920, 640), (54432, 5120, 640)]
if os.environ.get("KBIG_SQUARE"):
    SH = [(4096, 4096, 4096), (8192, 8192, 8192), (13824, 10240, 1280), (13824, 1280, 5120), (54272, 5120, 640)]
CFG = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "0,4").split(",")]
for M, N, K in SH:
    a = (torch.rand(M, K, device=dev) * 2 - 1).to(torch.float16)
    w = (torch.rand(N, K, device=dev) * 2 - 1).to(torch.float16)
    o = torch.empty(M, N, device=dev, dtype=torch.float16)
    line = f"{M}x{N}x{K} f16out |"
    for c in CFG:
        ops.set_knob("gemm_cfg", c)  # needs SEVA_HIP_LIB=build_ab/libseva_hip_exp.so for c > 0
        us = timeit(lambda: ops.gemm(a, w, out_f16=o))
        line += f" cfg{c}: {us:8.1f}us {2.0*M*N*K/us/1e6:6.0f}TF |"
    print(line, flush=True)
