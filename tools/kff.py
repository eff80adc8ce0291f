#!/usr/bin/env python3
"""Fused GEGLU->FF2 kernel (seva_ff_fused_f16) vs the two-kernel path at the ds1 shape of the headline config
(M = 42 x 72 x 72 = 217,728 token rows, C = 320), interleaved rounds in one process on one device."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "stable-virtual-camera_amd"))
import torch
from seva import ops
from seva._engine import interleave_geglu

dev = torch.device("cuda:0")
M, C = int(os.environ.get("KFF_M", 217728)), int(os.environ.get("KFF_C", 320))
g = torch.Generator().manual_seed(0)
a = torch.randn(M, C, generator=g).half().to(dev)
w1 = (torch.randn(8 * C, C, generator=g) * C ** -0.5).half().to(dev)
b1 = (0.1 * torch.randn(8 * C, generator=g)).to(dev)
w2 = (torch.randn(C, 4 * C, generator=g) * (4 * C) ** -0.5).half().to(dev)
b2 = (0.1 * torch.randn(C, generator=g)).to(dev)
res = torch.randn(M, C, generator=g).to(dev)
wi, bi = interleave_geglu(w1, b1)
hid = torch.empty((M, 4 * C), device=dev, dtype=torch.float16)
o_two = torch.empty((M, C), device=dev)
o_f = torch.empty((M, C), device=dev)


def two():
    ops.gemm(a, wi, bias=bi, out_f16=hid, geglu=True)
    ops.gemm(hid, w2, bias=b2, residual=res, out_f32=o_two)


def fused():
    ops.ff_fused(a, wi, bi, w2, b2, residual=res, out_f32=o_f)


def fused4():
    ops.set_knob("ff_variant", 4)
    ops.ff_fused(a, wi, bi, w2, b2, residual=res, out_f32=o_f)
    ops.set_knob("ff_variant", -1)


def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


flop = 2.0 * M * C * 8 * C + 2.0 * M * 4 * C * C
for rnd in range(3):
    t2, tf, t4 = timeit(two), timeit(fused), timeit(fused4)
    print(f"round {rnd}: two kernels {t2:8.1f} us ({flop / t2 / 1e6:6.1f} TFLOP/s) | fused 8-wave {tf:8.1f} us ({flop / tf / 1e6:6.1f} TFLOP/s) "
          f"x{t2 / tf:.2f} | fused 4-wave {t4:8.1f} us x{t2 / t4:.2f}", flush=True)
print("rel-L2 fused vs two-kernel:", float((o_f - o_two).norm() / o_two.norm()))
