#!/usr/bin/env python3
"""fp32-output GEMM / conv: the default tiles (160x160, or 128x160 where GroupNorm statistics are emitted) against the 8-wave
128x320 tile (knob gemm_bn = 320: ONE workgroup per CU whose two halves share the staged A rows).  Outputs must be bitwise equal.
Needs the experimental library: make -C stable-virtual-camera_amd/csrc exp; SEVA_HIP_LIB=build_ab/libseva_hip_exp.so."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stable-virtual-camera_amd"))
import torch
from seva import ops
from seva._engine import pack_conv3x3
dev = torch.device("cuda:0")
def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
cases = [("gemm ff2 ds2", ("g", 54432, 640, 2560)), ("gemm attn_out ds2", ("g", 54432, 640, 640)), ("gemm ff2 ds4", ("g", 13608, 1280, 5120)),
         ("gemm attn_out ds1", ("g", 217728, 320, 320)), ("conv 36 640->640", ("c", 42, 36, 640, 640)), ("conv 36 1280->640", ("c", 42, 36, 1280, 640)),
         ("conv 18 1280->1280", ("c", 42, 18, 1280, 1280)), ("conv 72 320->320", ("c", 42, 72, 320, 320)),
         ("conv 72 320->320 +stats", ("c", 42, 72, 320, 320, True)), ("conv 72 640->320 +stats", ("c", 42, 72, 640, 320, True))]
for name, c in cases:
    if c[0] == "g":
        _, M, N, K = c
        a = torch.randn(M, K, device=dev).half(); w = (torch.randn(N, K, device=dev) * 0.05).half()
        r = torch.randn(M, N, device=dev); b = torch.randn(N, device=dev)
        outs = [torch.empty_like(r), torch.empty_like(r)]
        fn = lambda o: ops.gemm(a, w, bias=b, residual=r, out_f32=o)
        fl = 2.0 * M * N * K
    else:
        _, n, side, cin, cout = c[:5]
        x = torch.randn(n, side, side, cin, device=dev).half(); w = pack_conv3x3(torch.randn(cout, cin, 3, 3, device=dev) * 0.05).half()
        M = n * side * side
        r = torch.randn(M, cout, device=dev); b = torch.randn(cout, device=dev)
        outs = [torch.empty_like(r), torch.empty_like(r)]
        st = [torch.zeros(ops.channel_stats_shape(M, cout), device=dev) for _ in range(2)] if len(c) > 5 else None
        fn = (lambda o: ops.conv3x3(x, w, bias=b, residual=r, out_f32=o, ch_stats=st[0] if o is outs[0] else st[1])) if st else (lambda o: ops.conv3x3(x, w, bias=b, residual=r, out_f32=o))
        fl = 2.0 * M * cout * 9 * cin
    line = f"{name:20s}"
    for rnd in range(2):
        for k, bn in enumerate((-1, 320)):
            ops.set_knob("gemm_bn", bn)
            us = timeit(lambda: fn(outs[k]))
            line += f" | bn={bn}: {us:7.1f} us {fl / us / 1e6:6.1f} TF"
    ops.set_knob("gemm_bn", -1)
    eq = bool(torch.equal(outs[0], outs[1]))
    if c[0] == "c" and len(c) > 5:
        eq = eq and bool(torch.equal(st[0], st[1]))
    print(line, "| equal:", eq, flush=True)
