#!/usr/bin/env python3
"""Launch a fixed list of representative kernels (2 launches each, in order) so that a
`rocprofv3 --pmc ... --kernel-trace` run can be mapped back to shapes by dispatch order.
    rocprofv3 --kernel-trace --pmc <counters> --output-format csv -d out -- python3 tools/kprof.py
`python tools/kprof.py --parse out_dir` prints per-case counter averages.
"""
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stable-virtual-camera_amd"))

CASES = [  # name, kind, params
    ("gemm ds1 qkv 217728x960x320", "gemm", (217728, 960, 320, "o16")),
    ("gemm ds1 attn_out 217728x320x320 +res", "gemm", (217728, 320, 320, "res")),
    ("gemm ds2 qkv 54432x1920x640", "gemm", (54432, 1920, 640, "o16")),
    ("gemm ds4 geglu 13608x10240x1280", "gemm", (13608, 10240, 1280, "geglu")),
    ("conv 36x36 640->640", "conv", (36, 640, 640)),
    ("conv 72x72 320->320", "conv", (72, 320, 320)),
    ("attn ds1 frame B42 H5 L5184", "attn", (42, 5, 5184)),
    ("attn ds2 joint B2 H10 L27216", "attn", (2, 10, 27216)),
    ("gn+silu+mod [42,5184,320]", "gn", (42, 5184, 320)),
    ("layernorm [217728,320]", "ln", (217728, 320)),
]
REPS = 2


def run():
    import torch
    from seva import ops
    from seva._engine import interleave_geglu
    dev = torch.device("cuda:0")
    F16 = torch.float16
    torch.manual_seed(0)
    for name, kind, prm in CASES:
        if kind == "gemm":
            M, N, K, fl = prm
            a = torch.randn(M, K, device=dev, dtype=F16)
            w = torch.randn(N, K, device=dev, dtype=F16) * K ** -0.5
            bias = torch.randn(N, device=dev)
            if fl == "geglu":
                w, bias = interleave_geglu(w, bias)
                o = torch.empty(M, N // 2, device=dev, dtype=F16)
                fn = lambda: ops.gemm(a, w, bias=bias, out_f16=o, geglu=True)
            elif fl == "res":
                r, o = torch.randn(M, N, device=dev), torch.empty(M, N, device=dev)
                fn = lambda: ops.gemm(a, w, bias=bias, residual=r, out_f32=o)
            else:
                o = torch.empty(M, N, device=dev, dtype=F16)
                fn = lambda: ops.gemm(a, w, out_f16=o)
        elif kind == "conv":
            side, cin, cout = prm
            x = torch.randn(42, side, side, cin, device=dev, dtype=F16)
            w = torch.randn(cout, 9 * cin, device=dev, dtype=F16) * (9 * cin) ** -0.5
            bias, r = torch.randn(cout, device=dev), torch.randn(42, side * side, cout, device=dev)
            o = torch.empty(42, side * side, cout, device=dev)
            fn = lambda: ops.conv3x3(x, w, bias=bias, residual=r, out_f32=o)
        elif kind == "attn":
            B, H, L = prm
            C = 64 * H
            qkv = torch.randn(B * L, 3 * C, device=dev, dtype=F16)
            o = torch.empty(B * L, C, device=dev, dtype=F16)
            fn = lambda: ops.attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], o, nb0=B, nb1=1, heads=H, lq=L, lk=L,
                                       q_strides=(L * 3 * C, 0, 3 * C), k_strides=(L * 3 * C, 0, 3 * C), o_strides=(L * C, 0, C))
        elif kind == "gn":
            n, hw, C = prm
            x = torch.randn(n, hw, C, device=dev)
            g, b = torch.randn(C, device=dev), torch.randn(C, device=dev)
            o = torch.empty(n, hw, C, device=dev, dtype=F16)
            ws = ops.groupnorm_workspace(n, dev)
            d, dw, db = torch.randn(n, hw, 6, device=dev), torch.randn(2 * C, 6, device=dev), torch.randn(2 * C, device=dev)
            fn = lambda: ops.groupnorm(x, None, g, b, o, ws, silu=True, dense=d, dense_w=dw, dense_b=db)
        else:
            rows, C = prm
            x = torch.randn(rows, C, device=dev)
            g, b = torch.randn(C, device=dev), torch.randn(C, device=dev)
            o = torch.empty(rows, C, device=dev, dtype=F16)
            fn = lambda: ops.layernorm(x, g, b, o)
        for _ in range(REPS):
            fn()
        torch.cuda.synchronize()


def parse(d):
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    rows = []
    for f in files:
        rows += list(csv.DictReader(open(f)))
    ours = [r for r in rows if any(k in r["Kernel_Name"] for k in ("gemm_kernel", "gemm_ring", "attn_kernel", "gn_", "layernorm"))]
    by = {}
    for r in ours:
        by.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"][:60]})[r["Counter_Name"]] = float(r["Counter_Value"])
    disp = [by[k] for k in sorted(by)]
    idx = 0
    for name, kind, _ in CASES:
        n = REPS * (2 if kind == "gn" else 1)
        grp = disp[idx: idx + n]
        idx += n
        if not grp:
            continue
        keys = sorted(k for k in grp[-1] if k != "name")
        print(f"{name:40s} " + " ".join(f"{k}={sum(g.get(k, 0) for g in grp) / len(grp):.4g}" for k in keys))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--parse":
        parse(sys.argv[2])
    else:
        run()
