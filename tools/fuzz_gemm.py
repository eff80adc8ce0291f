#!/usr/bin/env python3
"""Randomised exactness fuzz of seva_gemm_f16 (integer data => every path must be bit-exact): random M/N/K, output
kinds, residual / row_add / column scale, forced chunk counts, tile widths and heights (64 / 128 / 160 rows), A-in-registers
on/off, stream-K on/off, and -- for fp32 outputs -- the epilogue-emitted GroupNorm statistics (exact on integer data too)."""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stable-virtual-camera_amd"))
import torch
from seva import ops
dev = torch.device("cuda:0")
rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
ncases = int(sys.argv[2]) if len(sys.argv) > 2 else 300
g = torch.Generator().manual_seed(rng.randrange(1 << 30))
def ints(shape, lo, hi):
    return torch.randint(lo, hi + 1, shape, generator=g).float().to(dev)
bad = 0
cnt = {'stats': 0, 'bm160': 0}
skws = ops.splitk_workspace(70000, 2048, dev)  # stream-K workspace (used by the launches the knob sends there: M >= 512 tiles' worth)
for case in range(ncases):
    M = rng.choice([1, 7, 64, 127, 128, 129, 300, 777, 1025, 2049, 4100, 4100, 20000, 66000])
    N = rng.choice([4, 36, 128, 132, 160, 320, 324, 480, 640, 960, 1280, 1924])
    K = 64 * rng.choice([1, 2, 3, 5, 8, 10, 20])
    kind = rng.choice(["f16", "f16", "f32", "f32", "both", "f16res", "f32res", "f32res", "f16radd"])
    knobs = {}
    for k, vals in (("SEVA_GEMM_CHUNKS", [None, "1", "2", "3", "5"]), ("SEVA_GEMM_BN", [None, "128", "160"]),
                    ("SEVA_GEMM_BM", [None, None, "64", "128", "160"]), ("SEVA_GEMM_ASTAT", [None, "0"])):
        v = rng.choice(vals)
        knobs[k] = v
        ops.set_knob(k[5:].lower(), -1 if v is None else int(v))  # knobs are read from the environment only at load
    a, w, bias = ints((M, K), -4, 4), ints((N, K), -3, 3), ints((N,), -5, 5)
    res = ints((M, N), -9, 9) if "res" in kind else None
    rpg = rng.choice([1, 5, 64])
    radd = ints(((M + rpg - 1) // rpg, N), -3, 3) if "radd" in kind else None
    ns = 0
    if kind == "f16" and N > 32 and rng.random() < 0.5:
        ns = rng.randrange(0, N // 4 + 1) * 4
    o16 = torch.full((M, N), float("nan"), device=dev, dtype=torch.float16) if kind != "f32" and kind != "f32res" else None
    o32 = torch.full((M, N), float("nan"), device=dev) if kind in ("f32", "both", "f32res") else None
    stats = None
    if o32 is not None and N >= 128 and N % 4 == 0 and ns == 0 and rng.random() < 0.5:
        stats = torch.full(ops.channel_stats_shape(M, N), float("nan"), device=dev)
    cnt['stats'] += stats is not None
    cnt['bm160'] += knobs.get('SEVA_GEMM_BM') == '160' or (M >= 2048 and N % 160 == 0 and o32 is not None and all(v is None for k, v in knobs.items() if k != 'SEVA_GEMM_ASTAT'))
    for _ in range(2):
        ops.gemm(a.half(), w.half(), bias=bias, residual=res, row_add=radd, rows_per_group=rpg if radd is not None else 0,
                 out_f32=o32, out_f16=o16, col_scale=0.5 if ns else 1.0, col_scale_n=ns, ch_stats=stats,
                 splitk_ws=skws if o32 is not None else None)
    ref = a @ w.T + bias
    if ns: ref[:, :ns] *= 0.5
    if radd is not None: ref = ref + radd.repeat_interleave(rpg, 0)[:M]
    if res is not None: ref = ref + res
    torch.cuda.synchronize()
    ok = (o32 is None or torch.equal(o32, ref)) and (o16 is None or torch.equal(o16.float(), ref.half().float()))
    if stats is not None:  # sums of small integers are exact in fp32 whatever the order; the sums of squares can pass 2^24
        nb = stats.shape[0]
        rp = torch.zeros(nb * 64, N, device=dev); rp[:M] = ref
        rp = rp.view(nb, 64, N)
        ok = ok and torch.equal(stats[:, 0], rp.sum(1)) and torch.allclose(stats[:, 1].double(), (rp.double() ** 2).sum(1), rtol=1e-5, atol=0)
    if not ok:
        bad += 1
        print("MISMATCH", case, M, N, K, kind, ns, stats is not None, knobs, flush=True)
print(f"fuzz: {ncases} cases, {bad} mismatches; with statistics {cnt['stats']}, 160-row tiles {cnt['bm160']}")
sys.exit(1 if bad else 0)
