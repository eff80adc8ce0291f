#!/usr/bin/env python3
"""Randomised exactness fuzz of the implicit-GEMM 3x3 conv (integer data): sizes, stride 1/2, fused nearest-2x upsample,
bottom/right-only padding, residual / row_add, tile shapes; every fourth case aims at the 128-column / 2-D-tile / 32-column families of the
window-staged kernel (the VAE's shapes)."""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stable-virtual-camera_amd"))
import torch
import torch.nn.functional as F
from seva import ops
from seva._engine import pack_conv3x3
dev = torch.device("cuda:0")
rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
ncases = int(sys.argv[2]) if len(sys.argv) > 2 else 150
g = torch.Generator().manual_seed(rng.randrange(1 << 30))
def ints(shape, lo, hi):
    return torch.randint(lo, hi + 1, shape, generator=g).float().to(dev)
bad = 0
cnt = {'stats': 0, 'splitk': 0}
skws = ops.splitk_workspace(42 * 33 * 31 * 4, 320, dev)
for case in range(ncases):
    n = rng.choice([1, 2, 3, 5, 42])
    ih, iw = rng.choice([4, 5, 8, 9, 16, 33]), rng.choice([4, 6, 7, 12, 16, 31])
    cin, cout = 64 * rng.choice([1, 2, 3, 4]), rng.choice([4, 64, 96, 160, 320])
    mode = rng.choice(["s1", "s1", "s2", "up", "s2br"])
    kn = {}
    if mode == "s2br" and (ih % 2 or iw % 2):
        mode = "s2"
    # (the tile knobs switch the window-staged kernel off: half of the cases leave them all unset and pick one of its families instead)
    tile_knobs = rng.random() < 0.5
    ops.set_knob("conv_win", -1 if tile_knobs else rng.choice([-1, 1, 2]))
    for k, vals in (("SEVA_GEMM_BN", [None, "128", "160"] if tile_knobs else [None]), ("SEVA_GEMM_BM", [None, None, "64", "128", "160"] if tile_knobs else [None]),
                    ("SEVA_GEMM_CHUNKS", [None, "1", "2"] if tile_knobs else [None])):
        v = rng.choice(vals)
        kn[k] = v
        ops.set_knob(k[5:].lower(), -1 if v is None else int(v))  # knobs are read from the environment only at load
    if case % 5 == 4:  # every fifth case aims at the split-K = 2 launch: small image, deep even K, no knob
        ih, iw, cin, cout = rng.choice([4, 5, 8, 9]), rng.choice([4, 6, 7, 12]), rng.choice([128, 256]), rng.choice([160, 320])
        mode = rng.choice(["s1", "s1", "s2"])
        for k in kn:
            kn[k] = None
            ops.set_knob(k[5:].lower(), -1)
    if case % 4 == 2:  # the window kernel's other families: 128 / 256 columns, rows wide enough for 2-D tiles, a handful of output channels
        n = rng.choice([1, 2, 3])
        ih, iw = rng.choice([8, 16, 24, 40, 64]), rng.choice([16, 32, 48, 80, 96, 160])
        cin, cout = 64 * rng.choice([1, 2]), rng.choice([128, 256, 384, 4, 8, 32])
        mode = rng.choice(["s1", "s1", "up"])
        if mode == "up":
            ih, iw = max(4, ih // 2), iw // 2
        for k in kn:
            kn[k] = None
            ops.set_knob(k[5:].lower(), -1)
        ops.set_knob("conv_win", rng.choice([-1, -1, 1, 2]))
    x, w, b = ints((n, cin, ih, iw), -3, 3), ints((cout, cin, 3, 3), -2, 2), ints((cout,), -4, 4)
    xi = F.interpolate(x, scale_factor=2, mode="nearest") if mode == "up" else x
    if mode == "s2br":
        ref = F.conv2d(F.pad(xi, (0, 1, 0, 1)), w, b, stride=2)
    else:
        ref = F.conv2d(xi, w, b, stride=2 if mode == "s2" else 1, padding=1)
    oh, ow = ref.shape[-2:]
    res = ints((n, oh * ow, cout), -9, 9) if rng.random() < 0.5 else None
    radd = ints((n, cout), -3, 3) if rng.random() < 0.5 else None
    out = torch.full((n, oh * ow, cout), float("nan"), device=dev)
    stats = None
    if cout >= 128 and rng.random() < 0.5:  # epilogue-emitted GroupNorm statistics (64-row blocks of the [n * oh * ow, cout] output)
        stats = torch.full(ops.channel_stats_shape(n * oh * ow, cout), float("nan"), device=dev)
    use_sk = case % 5 == 4 or rng.random() < 0.5  # split-K = 2 hand-off for the small-image launches that qualify (the others ignore the workspace)
    cnt['stats'] += stats is not None
    cnt['splitk'] += use_sk and mode != 'up' and oh * ow <= 128 and (9 * cin // 64) >= 16 and (9 * cin // 64) % 2 == 0 and cout >= 64 and all(os.environ.get(k) is None for k in ()) and all(v is None for v in kn.values())
    for _ in range(2):
      ops.conv3x3(x.permute(0, 2, 3, 1).contiguous().half(), pack_conv3x3(w), stride=2 if mode in ("s2", "s2br") else 1,
                upsample=mode == "up", pad_br_only=mode == "s2br", bias=b, residual=res, row_add=radd,
                rows_per_group=oh * ow if radd is not None else 0, out_f32=out, ch_stats=stats, splitk_ws=skws if use_sk else None)
    refl = ref.permute(0, 2, 3, 1).reshape(n, oh * ow, cout)
    if radd is not None: refl = refl + radd[:, None, :]
    if res is not None: refl = refl + res
    torch.cuda.synchronize()
    ok = torch.equal(out, refl)
    if stats is not None:  # sums of small integers are exact in fp32 whatever the order; the sums of squares can pass 2^24
        M = n * oh * ow
        nb = stats.shape[0]
        if (oh * ow) % 64 == 0:  # the contract: the blocks of an image add up to the image (2-D tiles: a block is not 64 consecutive rows)
            nbi = oh * ow // 64
            ok = ok and torch.equal(stats[:, 0].view(n, nbi, cout).sum(1), refl.sum(1)) and \
                torch.allclose(stats[:, 1].view(n, nbi, cout).double().sum(1), (refl.double() ** 2).sum(1), rtol=1e-5, atol=0)
        else:
            rp = torch.zeros(nb * 64, cout, device=dev); rp[:M] = refl.reshape(M, cout)
            rp = rp.view(nb, 64, cout)
            ok = ok and torch.equal(stats[:, 0], rp.sum(1)) and torch.allclose(stats[:, 1].double(), (rp.double() ** 2).sum(1), rtol=1e-5, atol=0)
    if not ok:
        bad += 1
        print("MISMATCH", case, n, ih, iw, cin, cout, mode, stats is not None, use_sk, {k: os.environ.get(k) for k in ("SEVA_GEMM_BN", "SEVA_GEMM_BM", "SEVA_GEMM_CHUNKS")}, flush=True)
ops.set_knob("conv_win", -1)
print(f"conv fuzz: {ncases} cases, {bad} mismatches; with statistics {cnt['stats']}, split-K candidates {cnt['splitk']}")
sys.exit(1 if bad else 0)
