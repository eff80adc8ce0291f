"""GPU parity of every C-ABI operator against a plain PyTorch fp32 reference of the same op.

Run on the MI355X box: `python -m pytest tests -m gpu`.  GEMM-type ops are checked twice: exactly
(small-integer data, where fp16 inputs + fp32 accumulation are exact) and on random data against
fp32 math on the fp16-rounded operands (tolerance = accumulation-order noise only).
"""
import math
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from conftest import rel_l2


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from seva import _native
    _native.load()
    return torch.device("cuda:0")


def _ints(shape, lo, hi, dev, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return torch.randint(lo, hi + 1, shape, generator=g).float().to(dev)


def _rand(shape, dev, seed, scale=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dev)


GEMM_SHAPES = [(256, 256, 128), (128, 128, 64), (300, 320, 320), (42, 1280, 320), (1000, 4, 64),
               (777, 960, 640), (4097, 132, 192), (64, 36, 1024), (513, 480, 128), (130, 160, 64)]


@pytest.mark.parametrize("M,N,K", GEMM_SHAPES)
def test_gemm_exact_integers(dev, M, N, K):
    from seva import ops
    a = _ints((M, K), -4, 4, dev, 1)
    w = _ints((N, K), -3, 3, dev, 2)  # asymmetric, non-identity
    bias = _ints((N,), -5, 5, dev, 3)
    res = _ints((M, N), -9, 9, dev, 4)
    rpg = 7
    radd = _ints(((M + rpg - 1) // rpg, N), -3, 3, dev, 5)
    o32 = torch.full((M, N), float("nan"), device=dev)
    o16 = torch.full((M, N), float("nan"), device=dev, dtype=torch.float16)
    ops.gemm(a.half(), w.half(), bias=bias, row_add=radd, rows_per_group=rpg, residual=res,
             out_f32=o32, out_f16=o16)
    ref = a @ w.T + bias + res + radd.repeat_interleave(rpg, 0)[:M]
    torch.cuda.synchronize()
    assert torch.equal(o32, ref), f"max diff {(o32 - ref).abs().max()}"
    assert torch.equal(o16.float(), ref.half().float())
    # no-epilogue variant, f16 output only
    o16b = torch.empty((M, N), device=dev, dtype=torch.float16)
    ops.gemm(a.half(), w.half(), out_f16=o16b)
    assert torch.equal(o16b.float(), (a @ w.T).half().float())


@pytest.mark.parametrize("bn", ["128", "160"])
@pytest.mark.parametrize("M,N,K", [(300, 320, 320), (777, 132, 192), (257, 484, 64)])
def test_gemm_tile_width_knob_exact(dev, M, N, K, bn, knobs):
    """Both tile widths (knob gemm_bn) on widths that leave ragged last tiles either way."""
    from seva import ops
    knobs(gemm_bn=bn)
    a = _ints((M, K), -4, 4, dev, 11)
    w = _ints((N, K), -3, 3, dev, 12)
    bias = _ints((N,), -5, 5, dev, 13)
    res = _ints((M, N), -9, 9, dev, 14)
    o32 = torch.full((M, N), float("nan"), device=dev)
    ops.gemm(a.half(), w.half(), bias=bias, residual=res, out_f32=o32)
    torch.cuda.synchronize()
    assert torch.equal(o32, a @ w.T + bias + res)


@pytest.mark.parametrize("astat", ["0", "1"])
@pytest.mark.parametrize("M,N,K", [(300, 960, 320), (1000, 1920, 640), (515, 640, 192), (257, 2560, 320)])
def test_gemm_f16_only_multi_tile_walk_exact(dev, M, N, K, astat, knobs):
    """f16-only outputs take the ASYNC schedule (next-tile stages issued before the epilogue, counted vmcnt leaves the
    stores in flight) and, for K <= 320, the A-in-registers variant.  One workgroup per M-tile (knob gemm_chunks=1)
    makes it walk every N-tile, so the cross-tile bookkeeping is exercised; exact on integers, with bias, column
    scale and ragged M."""
    from seva import ops
    knobs(gemm_chunks=1, gemm_astat=astat)
    a = _ints((M, K), -4, 4, dev, 61)
    w = _ints((N, K), -3, 3, dev, 62)
    bias = _ints((N,), -5, 5, dev, 63)
    o16 = torch.full((M, N), float("nan"), device=dev, dtype=torch.float16)
    for _ in range(3):  # back-to-back launches: stores of one launch in flight under the next
        ops.gemm(a.half(), w.half(), bias=bias, out_f16=o16, col_scale=0.5, col_scale_n=N // 4 // 4 * 4)
    ref = a @ w.T + bias
    ref[:, : N // 4 // 4 * 4] *= 0.5
    torch.cuda.synchronize()
    assert torch.equal(o16.float(), ref.half().float())


@pytest.mark.parametrize("astat", ["0", "1"])
@pytest.mark.parametrize("M,C,K", [(300, 320, 320), (1000, 640, 640)])
def test_geglu_multi_tile_walk(dev, M, C, K, astat, knobs):
    from seva import ops
    from seva._engine import interleave_geglu
    knobs(gemm_chunks=1, gemm_astat=astat)
    a = _rand((M, K), dev, 71).half()
    w = (_rand((8 * C, K), dev, 72) * K ** -0.5).half()
    b = _rand((8 * C,), dev, 73)
    wi, bi = interleave_geglu(w, b)
    o16 = torch.full((M, 4 * C), float("nan"), device=dev, dtype=torch.float16)
    ops.gemm(a, wi, bias=bi, out_f16=o16, geglu=True)
    y = a.float() @ w.float().T + b
    ref = y[:, : 4 * C] * F.gelu(y[:, 4 * C:])
    assert torch.isfinite(o16).all() and rel_l2(o16, ref) < 1e-3


@pytest.mark.parametrize("M,N,K", [(300, 320, 320), (2049, 1280, 1280)])
def test_gemm_random(dev, M, N, K):
    from seva import ops
    a, w = _rand((M, K), dev, 1).half(), (_rand((N, K), dev, 2) / math.sqrt(K)).half()
    bias = _rand((N,), dev, 3)
    o32 = torch.empty((M, N), device=dev)
    ops.gemm(a, w, bias=bias, out_f32=o32)
    ref = a.float() @ w.float().T + bias
    assert rel_l2(o32, ref) < 2e-6


@pytest.mark.parametrize("M,C", [(200, 64), (1000, 320)])
def test_gemm_geglu(dev, M, C):
    from seva import ops
    from seva._engine import interleave_geglu
    nh = 4 * C
    a = _rand((M, C), dev, 1).half()
    w = (_rand((2 * nh, C), dev, 2) / math.sqrt(C)).half()
    b = _rand((2 * nh,), dev, 3, 0.1)
    wi, bi = interleave_geglu(w, b)
    o16 = torch.empty((M, nh), device=dev, dtype=torch.float16)
    o32 = torch.empty((M, nh), device=dev)
    ops.gemm(a, wi, bias=bi, out_f16=o16, out_f32=o32, geglu=True)
    y = a.float() @ w.float().T + b
    ref = y[:, :nh] * F.gelu(y[:, nh:])
    assert rel_l2(o32, ref) < 3e-6
    assert rel_l2(o16, ref) < 1e-3


@pytest.mark.parametrize("M,C,K", [(1000, 640, 640), (2049, 320, 1280), (1600, 128, 704), (4000, 1280, 1280)])
def test_geglu_tile_heights_bitwise_equal(dev, M, C, K, knobs):
    """GEGLU GEMM on 160 x 128 tiles (the default for K > 320, M >= 1024), 128 x 128 and 64 x 128 tiles: the same bits (a row's
    dot products do not depend on the tile it falls in), M tails against every height, f16 and f32 outputs; and right."""
    from seva import ops
    from seva._engine import interleave_geglu
    a = _rand((M, K), dev, 81).half()
    w = (_rand((8 * C, K), dev, 82) * K ** -0.5).half()
    b = _rand((8 * C,), dev, 83)
    wi, bi = interleave_geglu(w, b)
    outs = []
    for bm in (160, 128, 64, -1):
        knobs(gemm_bm=bm)
        o16 = torch.full((M, 4 * C), float("nan"), device=dev, dtype=torch.float16)
        o32 = torch.full((M, 4 * C), float("nan"), device=dev)
        ops.gemm(a, wi, bias=bi, out_f16=o16, out_f32=o32, geglu=True)
        outs.append((o16, o32))
    for o16, o32 in outs[1:]:
        assert torch.equal(o16, outs[0][0]) and torch.equal(o32, outs[0][1])
    y = a.float() @ w.float().T + b
    ref = y[:, : 4 * C] * F.gelu(y[:, 4 * C:])
    assert rel_l2(outs[0][1], ref) < 3e-6 and rel_l2(outs[0][0], ref) < 1e-3


@pytest.mark.parametrize("kind,shape", [("gemm", (3000, 640, 640)), ("gemm", (2049, 320, 1280)), ("gemm", (5000, 1280, 192)),
                                        ("conv", (9, 18, 18, 128, 320, 1)), ("conv", (8, 36, 36, 64, 640, 2)), ("conv", (3, 40, 24, 192, 160, 1))])
def test_tile_160x160_bitwise_equal(dev, kind, shape, knobs):
    """fp32-output GEMM / conv on 160 x 160 tiles (the default from M >= 2048 when N % 160 == 0 and no statistics are emitted) vs
    128 x 160 and 64 x 160 tiles: the same bits, with bias + row_add + residual and M tails against every height."""
    from seva import ops
    from seva._engine import pack_conv3x3
    if kind == "gemm":
        M, N, K = shape
        a, w = _rand((M, K), dev, 1).half(), _rand((N, K), dev, 2, 0.05).half()
        rpg = 100
        fn = lambda o: ops.gemm(a, w, bias=bias, row_add=radd, rows_per_group=rpg, residual=res, out_f32=o)
    else:
        n, ih, iw, cin, N, stride = shape
        oh, ow = (ih - 1) // stride + 1, (iw - 1) // stride + 1
        M, rpg = n * oh * ow, oh * ow
        x, w = _rand((n, ih, iw, cin), dev, 1).half(), pack_conv3x3(_rand((N, cin, 3, 3), dev, 2, 0.05)).half()
        fn = lambda o: ops.conv3x3(x, w, stride=stride, bias=bias, row_add=radd, rows_per_group=rpg, residual=res, out_f32=o)
    bias, res = _rand((N,), dev, 3), _rand((M, N), dev, 4)
    radd = _rand(((M + rpg - 1) // rpg, N), dev, 5)
    outs = []
    for bm in (160, 128, 64, -1):
        knobs(gemm_bm=bm, conv_win=0)  # the per-tap kernel's tile shapes (the window kernel reduces slab-outer: its own tests below)
        o = torch.full((M, N), float("nan"), device=dev)
        fn(o)
        outs.append(o)
    assert torch.isfinite(outs[0]).all()
    for o in outs[1:]:
        assert torch.equal(o, outs[0])


WIN_CASES = [  # n, ih, iw, cin, cout, statistics, fused nearest-2x upsample
    (42, 18, 18, 128, 320, False, False), (5, 36, 36, 64, 640, False, False), (2, 72, 72, 64, 320, True, False), (3, 9, 9, 128, 160, False, False),
    (5, 7, 11, 64, 320, False, False), (1, 33, 31, 192, 160, False, False), (2, 16, 16, 64, 160, True, False), (7, 5, 4, 128, 320, False, False),
    (4, 24, 40, 128, 480, False, False), (3, 8, 8, 64, 160, True, False),
    (42, 9, 9, 128, 320, False, True), (3, 18, 18, 64, 160, True, True), (2, 36, 36, 64, 320, True, True), (5, 7, 5, 128, 160, False, True),
    (1, 16, 24, 192, 320, True, True),
]


@pytest.mark.parametrize("n,ih,iw,cin,cout,stats,up", WIN_CASES)
def test_conv3x3_window_kernel(dev, n, ih, iw, cin, cout, stats, up, knobs):
    """csrc/conv_win.hip (reference convs: seva/modules/layers.py:101,113): the tile's input window + halo staged in LDS once per
    64-channel slab, nine taps from shifted fragment addresses.  Integer data: bit-exact against torch for both instantiation
    families (two 4-wave workgroups per CU / one 8-wave 256-row tile), with bias + row_add + residual and epilogue-emitted
    GroupNorm statistics; tiles that straddle images (18 x 18, 9 x 9, 5 x 4), per-image tiling (72 x 72), M tails, windows
    cut by the end of the batch.  Random data: the two families agree bit for bit (same reduction order) and differ from the
    per-tap kernel only by the order of the fp32 additions (slab-outer instead of tap-outer).  `up`: the fused nearest-2x upsample
    (reference layers.py:35-46): the window is staged from the SOURCE image, tap (ky, kx) of output pixel (y, x) reads source pixel
    ((y + ky - 1) >> 1, (x + kx - 1) >> 1)."""
    from seva import ops
    from seva._engine import pack_conv3x3
    sc = 2 if up else 1
    M, hw = n * ih * iw * sc * sc, ih * iw * sc * sc
    x = _ints((n, cin, ih, iw), -3, 3, dev, 1)
    w = _ints((cout, cin, 3, 3), -2, 2, dev, 2)
    bias, emb, res = _ints((cout,), -4, 4, dev, 3), _ints((n, cout), -2, 2, dev, 4), _ints((n, hw, cout), -5, 5, dev, 5)
    xin = F.interpolate(x, scale_factor=2, mode="nearest") if up else x
    ref = F.conv2d(xin, w, bias, padding=1).permute(0, 2, 3, 1).reshape(n, hw, cout) + emb[:, None, :] + res
    xh, wp = x.permute(0, 2, 3, 1).contiguous().half(), pack_conv3x3(w)
    for fam in (1, 2):
        knobs(conv_win=fam)
        out = torch.full((n, hw, cout), float("nan"), device=dev)
        st = torch.full(ops.channel_stats_shape(M, cout), float("nan"), device=dev) if stats else None
        ops.conv3x3(xh, wp, upsample=up, bias=bias, row_add=emb, rows_per_group=hw, residual=res, out_f32=out, ch_stats=st)
        assert torch.equal(out, ref), f"family {fam}: max diff {(out - ref).abs().max()}"
        if st is not None:
            assert torch.equal(st.double(), _block_stats(out.view(M, cout), M, cout))
    xr, wr = _rand((n, ih, iw, cin), dev, 6).half(), pack_conv3x3(_rand((cout, cin, 3, 3), dev, 7, 0.05).cpu()).half().to(dev)
    rr = _rand((n, hw, cout), dev, 8)
    outs = []
    for fam in (0, 1, 2):
        knobs(conv_win=fam)
        o = torch.full((n, hw, cout), float("nan"), device=dev)
        ops.conv3x3(xr, wr, upsample=up, bias=bias, residual=rr, out_f32=o)
        outs.append(o)
    assert torch.equal(outs[1], outs[2])
    assert rel_l2(outs[1], outs[0]) < 3e-6


WIN128_CASES = [  # n, ih, iw, cin, cout, fused nearest-2x upsample: the 128-column family (the VAE's channel counts)
    (2, 72, 72, 64, 128, False), (1, 144, 144, 64, 256, False), (3, 32, 48, 64, 128, False), (1, 160, 96, 128, 128, False), (1, 288, 288, 64, 128, False),
    (2, 16, 16, 64, 384, False), (2, 72, 72, 64, 128, True), (1, 144, 144, 64, 128, True), (3, 24, 40, 64, 256, True), (2, 8, 8, 128, 128, True),
]


@pytest.mark.parametrize("n,ih,iw,cin,cout,up", WIN128_CASES)
def test_conv3x3_window_kernel_128_columns_and_2d_tiles(dev, n, ih, iw, cin, cout, up, knobs):
    """The window-staged conv on the VAE decoder's shapes (reference seva/modules/autoencoder.py -> diffusers AutoencoderKL decoder convs):
    N % 128 == 0 instantiations, linear tiles while the window of consecutive pixels fits LDS (72 px rows), 2-D tiles of 16 output
    columns x 8 / 16 rows from 144 px rows (and wherever the linear window is too wide, e.g. 160 x 96), plain and with the fused
    nearest-2x upsample.  Integer data: bit-exact against torch for the default dispatch and both families, fp32 and f16 outputs,
    residual; GroupNorm statistics: the blocks of an image add up to that image's sums (with 2-D tiles a block is 4 tile rows x 16
    pixels, not 64 consecutive rows).  Random data: one frame of a batch is bitwise the frame computed alone (the kernel choice and
    the tiling depend on per-sample dimensions only), and the result differs from the per-tap kernel by summation order only."""
    from seva import ops
    from seva._engine import pack_conv3x3
    sc = 2 if up else 1
    hw = ih * iw * sc * sc
    M = n * hw
    x = _ints((n, cin, ih, iw), -3, 3, dev, 11)
    w = _ints((cout, cin, 3, 3), -2, 2, dev, 12)
    bias, res = _ints((cout,), -4, 4, dev, 13), _ints((n, hw, cout), -5, 5, dev, 15)
    xin = F.interpolate(x, scale_factor=2, mode="nearest") if up else x
    ref = F.conv2d(xin, w, bias, padding=1).permute(0, 2, 3, 1).reshape(n, hw, cout) + res
    xh, wp = x.permute(0, 2, 3, 1).contiguous().half(), pack_conv3x3(w)
    for fam in (-1, 1, 2):
        knobs(conv_win=fam)
        out = torch.full((n, hw, cout), float("nan"), device=dev)
        o16 = torch.full((n, hw, cout), float("nan"), device=dev, dtype=torch.float16)
        st = torch.full(ops.channel_stats_shape(M, cout), float("nan"), device=dev)
        ops.conv3x3(xh, wp, upsample=up, bias=bias, residual=res, out_f32=out, out_f16=o16, ch_stats=st)
        assert torch.equal(out, ref), f"family {fam}: max diff {(out - ref).abs().max()}"
        assert torch.equal(o16, ref.half())
        nb = hw // 64
        assert torch.equal(st[:, 0].view(n, nb, cout).sum(1), ref.sum(1))
        assert torch.allclose(st[:, 1].view(n, nb, cout).double().sum(1), (ref.double() ** 2).sum(1), rtol=1e-6, atol=0)
    xr, wr = _rand((n, ih, iw, cin), dev, 16).half(), pack_conv3x3(_rand((cout, cin, 3, 3), dev, 17, 0.05).cpu()).half().to(dev)
    outs = []
    for fam in (0, -1, 1, 2):
        knobs(conv_win=fam)
        o = torch.full((n, hw, cout), float("nan"), device=dev)
        ops.conv3x3(xr, wr, upsample=up, bias=bias, out_f32=o)
        outs.append(o)
    assert torch.equal(outs[1], outs[2]) and torch.equal(outs[2], outs[3])
    assert rel_l2(outs[1], outs[0]) < 3e-6
    knobs(conv_win=-1)
    one = torch.full((1, hw, cout), float("nan"), device=dev)
    ops.conv3x3(xr[-1:].contiguous(), wr, upsample=up, bias=bias, out_f32=one)
    assert torch.equal(one[0], outs[1][-1])


@pytest.mark.parametrize("n,ih,iw,cin,cout", [(2, 72, 72, 128, 4), (3, 9, 9, 64, 4), (2, 36, 36, 128, 32), (1, 144, 144, 64, 8), (1, 160, 96, 64, 4)])
def test_conv3x3_window_kernel_narrow_outputs(dev, n, ih, iw, cin, cout, knobs):
    """Convs with a handful of output channels (reference model.py:170-174: the UNet's head, 320 -> 4; the VAE's conv_out) are bound
    by reading their input; the 32-column family of the window kernel reads it once instead of nine times.  Integer data: bit-exact
    against torch, linear and 2-D tiles, and equal to the per-tap kernel's result."""
    from seva import ops
    from seva._engine import pack_conv3x3
    hw = ih * iw
    x = _ints((n, cin, ih, iw), -3, 3, dev, 21)
    w = _ints((cout, cin, 3, 3), -2, 2, dev, 22)
    bias, res = _ints((cout,), -4, 4, dev, 23), _ints((n, hw, cout), -5, 5, dev, 25)
    ref = F.conv2d(x, w, bias, padding=1).permute(0, 2, 3, 1).reshape(n, hw, cout) + res
    xh, wp = x.permute(0, 2, 3, 1).contiguous().half(), pack_conv3x3(w)
    for fam in (-1, 0):
        knobs(conv_win=fam)
        out = torch.full((n, hw, cout), float("nan"), device=dev)
        ops.conv3x3(xh, wp, bias=bias, residual=res, out_f32=out)
        assert torch.equal(out, ref), f"conv_win {fam}: max diff {(out - ref).abs().max()}"


CONV_CASES = [  # n, ih, iw, cin, cout, stride, upsample
    (2, 9, 9, 64, 64, 1, False), (3, 16, 12, 128, 96, 1, False), (2, 16, 12, 64, 64, 2, False),
    (2, 9, 7, 64, 128, 2, False), (2, 8, 6, 64, 64, 1, True), (1, 5, 5, 192, 4, 1, False),
    (5, 33, 31, 64, 320, 1, False),
]


@pytest.mark.parametrize("n,ih,iw,cin,cout", [(2, 16, 12, 64, 64), (1, 10, 14, 128, 96)])
def test_conv3x3_stride2_bottom_right_padding(dev, n, ih, iw, cin, cout):
    """diffusers Downsample2D: F.pad(x, (0,1,0,1)) then conv3x3 stride 2 padding 0."""
    from seva import ops
    from seva._engine import pack_conv3x3
    x = _ints((n, cin, ih, iw), -3, 3, dev, 41)
    w = _ints((cout, cin, 3, 3), -2, 2, dev, 42)
    b = _ints((cout,), -4, 4, dev, 43)
    oh, ow = ih // 2, iw // 2
    out = torch.full((n, oh * ow, cout), float("nan"), device=dev)
    ops.conv3x3(x.permute(0, 2, 3, 1).contiguous().half(), pack_conv3x3(w), stride=2, pad_br_only=True, bias=b, out_f32=out)
    ref = F.conv2d(F.pad(x, (0, 1, 0, 1)), w, b, stride=2)
    assert torch.equal(out.view(n, oh, ow, cout).permute(0, 3, 1, 2), ref)


@pytest.mark.parametrize("n,ih,iw,cin,cout,stride,up", CONV_CASES)
def test_conv3x3(dev, n, ih, iw, cin, cout, stride, up):
    from seva import ops
    from seva._engine import pack_conv3x3
    x = _ints((n, cin, ih, iw), -3, 3, dev, 1)
    w = _ints((cout, cin, 3, 3), -2, 2, dev, 2)
    bias = _ints((cout,), -4, 4, dev, 3)
    xin = F.interpolate(x, scale_factor=2, mode="nearest") if up else x
    ref = F.conv2d(xin, w, bias, stride=stride, padding=1)
    oh, ow = ref.shape[-2:]
    temb = _ints((n, cout), -2, 2, dev, 4)
    res = _ints((n, oh * ow, cout), -5, 5, dev, 5)
    ref = ref + temb[:, :, None, None] + res.view(n, oh, ow, cout).permute(0, 3, 1, 2)
    x_nhwc = x.permute(0, 2, 3, 1).contiguous().half()
    out = torch.full((n, oh * ow, cout), float("nan"), device=dev)
    ops.conv3x3(x_nhwc, pack_conv3x3(w), stride=stride, upsample=up, bias=bias, row_add=temb,
                rows_per_group=oh * ow, residual=res, out_f32=out)
    got = out.view(n, oh, ow, cout).permute(0, 3, 1, 2)
    assert torch.equal(got, ref), f"max diff {(got - ref).abs().max()}"


def _attn_ref(q, k, v, scale):
    att = torch.softmax((q.float() @ k.float().transpose(-1, -2)) * scale, -1)
    return att @ v.float()


@pytest.mark.parametrize("no_tr", ["0", "1"])
@pytest.mark.parametrize("B,H,Lq,Lk", [(3, 2, 200, 200), (2, 5, 128, 128), (1, 1, 1701, 1701),
                                       (2, 3, 70, 5), (4, 2, 21, 21), (1, 2, 33, 64)])
def test_attention_fused_qkv(dev, B, H, Lq, Lk, no_tr, knobs):
    """q,k,v as column slices of a [B, L, 3C] buffer (the layout the engine uses)."""
    from seva import ops
    knobs(attn_no_tr=no_tr)
    C = 64 * H
    L = max(Lq, Lk)
    qkv = _rand((B, L, 3 * C), dev, 7).half()
    out = torch.full((B, Lq, C), float("nan"), device=dev, dtype=torch.float16)
    q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
    ops.attention(q, k, v, out, nb0=B, nb1=1, heads=H, lq=Lq, lk=Lk,
                  q_strides=(L * 3 * C, 0, 3 * C), k_strides=(L * 3 * C, 0, 3 * C),
                  o_strides=(Lq * C, 0, C))
    qh = q[:, :Lq].reshape(B, Lq, H, 64).transpose(1, 2)
    kh = k[:, :Lk].reshape(B, Lk, H, 64).transpose(1, 2)
    vh = v[:, :Lk].reshape(B, Lk, H, 64).transpose(1, 2)
    ref = _attn_ref(qh, kh, vh, 0.125).transpose(1, 2).reshape(B, Lq, C)
    err = rel_l2(out, ref)
    assert err < 2e-3, f"rel_l2 {err}"


QK_C = 0.125 * 1.4426950408889634


@pytest.mark.parametrize("B,H,Lq,Lk", [(3, 2, 200, 200), (1, 1, 1701, 1701), (2, 3, 70, 5), (4, 2, 21, 21),
                                       (1, 2, 33, 64), (2, 1, 500, 777), (2, 2, 1024, 900), (1, 3, 777, 1300)])
@pytest.mark.parametrize("spike", [False, True])
@pytest.mark.parametrize("two", ["0", "1", "4"])
def test_attention_prescaled_q(dev, B, H, Lq, Lk, spike, two, knobs):
    """q already multiplied by scale*log2(e) (what the engine's QKV projection emits): the kernel starts
    its score accumulators at -m_run and exponentiates the MFMA output directly.  `spike` plants, late in
    the key sequence, keys that beat the running maximum by far more than the deferred-rescale threshold
    (rescale branch), and shifts all logits of the first tile far below zero (first-tile reference)."""
    from seva import ops
    knobs(attn_two=two)  # 0: attn_kernel everywhere; from Lq >= 512: 1 = attn2_kernel (32x32x16 MFMA), 4 = attn16_kernel (16x16x32)
    C = 64 * H
    g = torch.Generator().manual_seed(31)
    q = torch.randn((B, Lq, H, 64), generator=g)
    k = torch.randn((B, Lk, H, 64), generator=g)
    v = torch.randn((B, Lk, H, 64), generator=g)
    if spike:
        k[:, : min(64, Lk)] -= 3.0 * q[:, :1].mean(1, keepdim=True).sign()  # first tile: strongly negative-ish
        if Lk > 70:
            k[:, Lk - 3] = 6.0 * q[:, Lq // 2]  # a late key that dominates one query row
            k[:, Lk // 2] = 4.0 * q[:, 0]
    qs = (q * QK_C).half()
    q16, k16, v16 = qs.to(dev).view(B, Lq, C), k.half().to(dev).view(B, Lk, C), v.half().to(dev).view(B, Lk, C)
    out = torch.full((B, Lq, C), float("nan"), device=dev, dtype=torch.float16)
    ops.attention(q16, k16, v16, out, nb0=B, nb1=1, heads=H, lq=Lq, lk=Lk, q_strides=(Lq * C, 0, C),
                  k_strides=(Lk * C, 0, C), o_strides=(Lq * C, 0, C), q_prescaled=True)
    # reference on exactly the operands the kernel sees: softmax base 2 of q' . k
    qh = qs.double().transpose(1, 2)
    kh = k.half().double().transpose(1, 2)
    vh = v.half().double().transpose(1, 2)
    att = torch.softmax(qh @ kh.transpose(-1, -2) * math.log(2.0), -1) @ vh
    ref = att.transpose(1, 2).reshape(B, Lq, C)
    assert torch.isfinite(out).all()
    assert rel_l2(out.cpu(), ref) < 2e-3


@pytest.mark.parametrize("B,H,L,Lk,split", [(2, 3, 6804, 6804, -1), (1, 2, 2300, 6150, -1), (2, 2, 2100, 700, 2), (1, 3, 2049, 1300, 3),
                                            (3, 1, 2500, 257, 4), (1, 1, 2048, 129, 2)])
def test_attention_kv_split(dev, B, H, L, Lk, split, knobs):
    """K/V split of long key sequences (seva_attn_desc.split_ws: two workgroups per query block + an fp32 combine): against an
    fp64 softmax reference, and against the unsplit kernel (same per-tile arithmetic, one extra fp32 re-association at the
    combine).  Default rule (split -1): lk >= 6144 -> 2 halves; forced factors 2..4 cover uneven tile counts, a ragged last
    tile in the last split and spikes that trigger the rescale branch in only one of the splits.  Without a workspace, or with
    the knob at 0, nothing is split (bitwise the unsplit result)."""
    from seva import ops
    C = 64 * H
    g = torch.Generator().manual_seed(313)
    q = torch.randn((B, L, H, 64), generator=g)
    k = torch.randn((B, Lk, H, 64), generator=g)
    v = torch.randn((B, Lk, H, 64), generator=g)
    k[:, Lk - 3] = 5.0 * q[:, L // 2]  # a late key (last split) that dominates one query row
    k[:, 5] = 4.0 * q[:, 7]            # an early one (first split)
    qs = (q * QK_C).half()
    q16, k16, v16 = qs.to(dev).view(B, L, C), k.half().to(dev).view(B, Lk, C), v.half().to(dev).view(B, Lk, C)
    ws = torch.empty(ops.attention_split_workspace_numel(B, H, L, 4), device=dev)

    def run(knob, w):
        knobs(attn_split=knob)
        o = torch.full((B, L, C), float("nan"), device=dev, dtype=torch.float16)
        ops.attention(q16, k16, v16, o, nb0=B, nb1=1, heads=H, lq=L, lk=Lk, q_strides=(L * C, 0, C), k_strides=(Lk * C, 0, C),
                      o_strides=(L * C, 0, C), q_prescaled=True, split_ws=w)
        torch.cuda.synchronize()
        return o

    plain = run(0, ws)
    assert torch.equal(plain, run(split, None))  # no workspace -> never split
    got = run(split, ws)
    assert torch.isfinite(got).all()
    qh = qs.double().transpose(1, 2).to(dev)
    kh, vh = k.half().double().transpose(1, 2).to(dev), v.half().double().transpose(1, 2).to(dev)
    ref = torch.empty((B, H, L, 64), dtype=torch.float64, device=dev)
    for s0 in range(0, L, 1024):
        ref[:, :, s0:s0 + 1024] = torch.softmax(qh[:, :, s0:s0 + 1024] @ kh.transpose(-1, -2) * math.log(2.0), -1) @ vh
    ref = ref.transpose(1, 2).reshape(B, L, C)
    e_split, e_plain, e_pair = rel_l2(got.double(), ref), rel_l2(plain.double(), ref), rel_l2(got.float(), plain.float())
    print(f"\nK/V split B={B} H={H} L={L} Lk={Lk} factor {split}: vs fp64 {e_split:.2e} (unsplit {e_plain:.2e}), split vs unsplit {e_pair:.2e}")
    assert e_split < 1e-3 and e_split < 1.2 * e_plain + 1e-5 and e_pair < 6e-4
    assert not torch.equal(got, plain) or Lk < 128  # the split path really ran (a different association somewhere)


@pytest.mark.parametrize("above", [12.0, 13.5, 14.2, 15.0, 15.9, 16.5, 18.0, 40.0])
def test_attention_late_key_around_the_rescale_bound(dev, above):
    """attn2_kernel exponentiates first and looks at the maximum only when a lane's partial row sum reaches 2^14: a late key whose
    score lies `above` log2 units over the first tile's maximum walks that decision through every regime -- well below the bound
    (P = 2^12, no rescale), just below / at / above it (2^13.5 ... 2^15.9: the sum test fires), beyond the f16 range (>= 2^16: the pack
    saturates, the sum test fires) and far beyond (inf).  Every case must agree with fp64."""
    from seva import ops
    B, H, Lq, Lk = 1, 2, 2304, 320
    C = 64 * H
    g = torch.Generator().manual_seed(5)
    q = torch.randn((B, Lq, H, 64), generator=g)
    k = torch.randn((B, Lk, H, 64), generator=g)
    v = torch.randn((B, Lk, H, 64), generator=g)
    qs = (q * QK_C).half()
    # per (row, head) maximum over the first 64-key tile, in the kernel's own log2 units; the planted key (index 200) is the row's
    # own direction scaled so that its score is that maximum + `above`
    s0 = torch.einsum("blhd,bkhd->bhlk", qs.double(), k[:, :64].half().double()).amax(-1)          # [B, H, Lq]
    # one key serves all rows: aim at row 0 of head 0 / 1; other rows see a smaller, harmless score
    for h in range(H):
        d = qs[0, 0, h].double()
        k[0, 200, h] = (d / d.dot(d) * (float(s0[0, h, 0]) + above)).float()
    q16, k16, v16 = qs.to(dev).view(B, Lq, C), k.half().to(dev).view(B, Lk, C), v.half().to(dev).view(B, Lk, C)
    out = torch.full((B, Lq, C), float("nan"), device=dev, dtype=torch.float16)
    ops.attention(q16, k16, v16, out, nb0=B, nb1=1, heads=H, lq=Lq, lk=Lk, q_strides=(Lq * C, 0, C), k_strides=(Lk * C, 0, C),
                  o_strides=(Lq * C, 0, C), q_prescaled=True)
    torch.cuda.synchronize()
    assert torch.isfinite(out).all()
    qd = qs.double().transpose(1, 2)
    kd, vd = k.half().double().transpose(1, 2), v.half().double().transpose(1, 2)
    ref = (torch.softmax(qd @ kd.transpose(-1, -2) * math.log(2.0), -1) @ vd).transpose(1, 2).reshape(B, Lq, C)
    err_all = rel_l2(out.cpu(), ref)
    err_row0 = rel_l2(out[:, :1].cpu(), ref[:, :1])
    print(f"\nlate key {above:4.1f} log2 units above the first tile's maximum: rel-L2 vs fp64 {err_all:.2e}, the targeted row {err_row0:.2e}")
    assert err_all < 1e-3 and err_row0 < 2e-3


@pytest.mark.parametrize("B,H,L,Lk", [(4, 5, 7000, 1100), (1, 10, 13500, 700), (3, 7, 6950, 640)])
def test_attention_kernels_round_identically(dev, B, H, L, Lk, knobs):
    """attn_kernel (32 queries per wave) and attn2_kernel (64 per wave, shared K / V fragments), both on v_mfma_f32_32x32x16, perform
    the same arithmetic in the same order per query row: their outputs are equal bit for bit.  (Both take the first tile's maximum
    as the exponent reference; they differ only in WHEN a later, rare rescale happens -- a score 8 above the reference in
    attn_kernel, a probability near 2^14 in attn2_kernel -- which scores of this spread never reach;
    test_attention_prescaled_q[spike=True] covers those paths against fp64.)  The default long-sequence kernel, attn16_kernel, is
    the same scheme on v_mfma_f32_16x16x32, whose k products are summed in another order: it agrees with the other two to the
    rounding of the f16 output and sits at the same distance from fp64; WHICH kernel a launch gets is a rule on lq alone (a
    per-sample dimension), never on the batch."""
    from seva import ops
    C = 64 * H
    g = torch.Generator().manual_seed(77)
    q = (torch.randn((B, L, C), generator=g) * QK_C).half().to(dev)
    k = torch.randn((B, Lk, C), generator=g).half().to(dev)
    v = torch.randn((B, Lk, C), generator=g).half().to(dev)
    outs = []
    for two in (0, 2, -1):  # attn_kernel everywhere / attn2_kernel / default (attn16_kernel from lq >= 2048)
        knobs(attn_two=two)
        o = torch.full((B, L, C), float("nan"), device=dev, dtype=torch.float16)
        ops.attention(q, k, v, o, nb0=B, nb1=1, heads=H, lq=L, lk=Lk, q_strides=(L * C, 0, C), k_strides=(Lk * C, 0, C),
                      o_strides=(L * C, 0, C), q_prescaled=True)
        torch.cuda.synchronize()
        assert torch.isfinite(o).all()
        outs.append(o)
    assert torch.equal(outs[0], outs[1])
    assert not torch.equal(outs[0], outs[2]), "the default is expected to be the 16x16x32 kernel"
    assert rel_l2(outs[2].float(), outs[0].float()) < 2e-4  # one f16 rounding apart at most, on a few elements
    qh = q[-1:, -300:].cpu().double().view(1, 300, H, 64).transpose(1, 2)
    kh, vh = (t[-1:].cpu().double().view(1, Lk, H, 64).transpose(1, 2) for t in (k, v))
    ref = (torch.softmax(qh @ kh.transpose(-1, -2) * math.log(2.0), -1) @ vh).transpose(1, 2).reshape(1, 300, C)
    e32, e16 = rel_l2(outs[1][-1:, -300:].cpu(), ref), rel_l2(outs[2][-1:, -300:].cpu(), ref)
    assert e32 < 2e-3 and e16 < 2e-3 and e16 < 1.1 * e32 + 1e-5, (e32, e16)


def test_attention_prescaled_temporal(dev):
    from seva import ops
    B, T, S, H = 2, 21, 40, 2
    C = 64 * H
    qkv = _rand((B * T, S, 3 * C), dev, 12)
    qkv[..., :C] *= QK_C
    qkv = qkv.half()
    out = torch.full((B * T, S, C), float("nan"), device=dev, dtype=torch.float16)
    ops.attention(qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:], out, nb0=B, nb1=S, heads=H,
                  lq=T, lk=T, q_strides=(T * S * 3 * C, 3 * C, S * 3 * C),
                  k_strides=(T * S * 3 * C, 3 * C, S * 3 * C), o_strides=(T * S * C, C, S * C), q_prescaled=True)
    x = qkv.view(B, T, S, 3, H, 64).permute(3, 0, 2, 4, 1, 5).double()
    ref = torch.softmax(x[0] @ x[1].transpose(-1, -2) * math.log(2.0), -1) @ x[2]
    ref = ref.permute(0, 3, 1, 2, 4).reshape(B * T, S, C)
    assert rel_l2(out, ref) < 2e-3


@pytest.mark.parametrize("M,N,K,ns", [(300, 960, 320, 320), (130, 480, 64, 160), (257, 384, 128, 128)])
def test_gemm_column_scale_exact(dev, M, N, K, ns):
    """Features < col_scale_n are scaled in fp32 before the f16 rounding; the rest are untouched."""
    from seva import ops
    a = _ints((M, K), -4, 4, dev, 21)
    w = _ints((N, K), -3, 3, dev, 22)
    bias = _ints((N,), -5, 5, dev, 23)
    o32 = torch.full((M, N), float("nan"), device=dev)
    o16 = torch.full((M, N), float("nan"), device=dev, dtype=torch.float16)
    ops.gemm(a.half(), w.half(), bias=bias, out_f32=o32, out_f16=o16, col_scale=0.375, col_scale_n=ns)
    ref = a @ w.T + bias
    ref[:, :ns] *= 0.375
    torch.cuda.synchronize()
    assert torch.equal(o32, ref)
    assert torch.equal(o16.float(), ref.half().float())
    with pytest.raises(Exception):
        ops.gemm(a.half(), w.half(), residual=o32.clone(), out_f32=o32, col_scale=0.5, col_scale_n=ns)



def test_attention_exact_uniform(dev):
    """All-equal scores -> output is the plain mean of V (exact in fp16 for integer V)."""
    from seva import ops
    B, H, L = 1, 1, 64
    q = torch.zeros((B, L, 64), device=dev, dtype=torch.float16)
    k = _rand((B, L, 64), dev, 1).half()
    v = _ints((B, L, 64), -8, 8, dev, 2)
    v[:, 0] = v[:, 0] - v.sum(1)  # column sums exactly zero except row 0 adjusts -> mean = 0
    out = torch.empty((B, L, 64), device=dev, dtype=torch.float16)
    ops.attention(q, k, v.half(), out, nb0=B, nb1=1, heads=H, lq=L, lk=L, q_strides=(L * 64, 0, 64),
                  k_strides=(L * 64, 0, 64), o_strides=(L * 64, 0, 64))
    assert out.float().abs().max() < 1e-3


def test_attention_one_hot_keys(dev):
    """Huge logits make softmax one-hot: out[q] must equal V[perm[q]] -> checks the key<->value
    pairing of the transposed-V operand (asymmetric, every row distinct)."""
    from seva import ops
    L = 128
    perm = torch.randperm(L, generator=torch.Generator().manual_seed(3))
    eye = torch.eye(64)
    k = torch.zeros(L, 64)
    q = torch.zeros(L, 64)
    # give key j a unique +-1 code in 64 dims; query i copies the code of key perm[i] (x40)
    codes = torch.sign(torch.randn(L, 64, generator=torch.Generator().manual_seed(4)))
    k[:] = codes
    q[:] = codes[perm] * 40.0
    v = _ints((L, 64), -20, 20, torch.device("cpu"), 5)
    out = torch.empty((1, L, 64), device=dev, dtype=torch.float16)
    ops.attention(q.half().to(dev)[None], k.half().to(dev)[None], v.half().to(dev)[None], out,
                  nb0=1, nb1=1, heads=1, lq=L, lk=L, q_strides=(L * 64, 0, 64),
                  k_strides=(L * 64, 0, 64), o_strides=(L * 64, 0, 64))
    ref = _attn_ref(q[None, None], k[None, None], v[None, None], 0.125)[0]
    assert rel_l2(out.cpu(), ref) < 2e-3
    assert torch.equal(out[0].float().cpu().round(), v[perm])


@pytest.mark.parametrize("T,S,H", [(21, 50, 2), (4, 36, 1), (24, 9, 5)])
def test_attention_temporal_strided(dev, T, S, H):
    """Temporal regime: tokens = frames, batch = (b, pixel); read in place from [(b t), s, 3C]."""
    from seva import ops
    B, C = 2, 64 * H
    qkv = _rand((B * T, S, 3 * C), dev, 11).half()
    out = torch.full((B * T, S, C), float("nan"), device=dev, dtype=torch.float16)
    ops.attention(qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:], out, nb0=B, nb1=S, heads=H,
                  lq=T, lk=T, q_strides=(T * S * 3 * C, 3 * C, S * 3 * C),
                  k_strides=(T * S * 3 * C, 3 * C, S * 3 * C), o_strides=(T * S * C, C, S * C))
    x = qkv.view(B, T, S, 3, H, 64).permute(3, 0, 2, 4, 1, 5)  # [3, B, S, H, T, 64]
    ref = _attn_ref(x[0], x[1], x[2], 0.125)  # [B,S,H,T,64]
    ref = ref.permute(0, 3, 1, 2, 4).reshape(B * T, S, C)
    assert rel_l2(out, ref) < 2e-3


GN_CASES = [  # n, hw, c1, c2, silu, dense
    (3, 81, 64, 0, False, False), (2, 100, 64, 32, True, True), (4, 324, 320, 0, True, True),
    (2, 50, 1280, 640, True, True), (2, 36, 1280, 1280, True, False), (2, 37, 640, 320, True, True),
    (1, 5184, 320, 0, True, False),
]


def test_groupnorm_raw_f16_second_output(dev):
    """raw_f16 = cat(x1, x2) cast to f16, written by the same pass; the normalised output is unchanged by it."""
    from seva import ops
    n, hw, c1, c2 = 3, 77, 64, 32
    x1, x2 = _rand((n, hw, c1), dev, 51), _rand((n, hw, c2), dev, 52)
    g, b = _rand((c1 + c2,), dev, 53), _rand((c1 + c2,), dev, 54)
    dense, dw, db = _rand((n, hw, 6), dev, 55), _rand((2 * (c1 + c2), 6), dev, 56, 0.1), _rand((2 * (c1 + c2),), dev, 57, 0.1)
    ws = ops.groupnorm_workspace(n, dev)
    for kw in (dict(), dict(dense=dense, dense_w=dw, dense_b=db)):
        o0 = torch.empty((n, hw, c1 + c2), device=dev, dtype=torch.float16)
        o1, raw = torch.empty_like(o0), torch.full_like(o0, float("nan"))
        ops.groupnorm(x1, x2, g, b, o0, ws, silu=True, **kw)
        ops.groupnorm(x1, x2, g, b, o1, ws, silu=True, raw_f16=raw, **kw)
        assert torch.equal(o0, o1)
        assert torch.equal(raw, torch.cat([x1, x2], -1).half())


def test_split_precision_outputs_and_their_consumers(dev):
    """Split-precision operands (seva_groupnorm_desc.split_*, seva_nchw_to_nhwc_f16_split): the [hi | lo] halves are exactly
    hi = f16(v), lo = f16(v - f32(hi)); the hi half is bitwise the plain output; and a GEMM against duplicated weights [W | W]
    reproduces the fp32 product to ~1e-6 where the plain f16 operand gives ~2e-4 (what the engine uses for the stem, the head
    and the 1x1 skip convs: tests/test_f16_floor_cpu.py)."""
    from seva import ops
    n, hw, c1, c2 = 2, 192, 96, 32
    C = c1 + c2
    x1, x2 = _rand((n, hw, c1), dev, 61, 7.0), _rand((n, hw, c2), dev, 62, 7.0)
    g, b = _rand((C,), dev, 63), _rand((C,), dev, 64)
    dense, dw, db = _rand((n, hw, 6), dev, 65), _rand((2 * C, 6), dev, 66, 0.1), _rand((2 * C,), dev, 67, 0.1)
    ws = ops.groupnorm_workspace(n, dev)
    for kw in (dict(), dict(dense=dense, dense_w=dw, dense_b=db)):
        o0, raw0 = torch.empty((n, hw, C), device=dev, dtype=torch.float16), torch.empty((n, hw, C), device=dev, dtype=torch.float16)
        ops.groupnorm(x1, x2, g, b, o0, ws, silu=True, raw_f16=raw0, **kw)
        # raw split (both variants); out split (plain variant only: the head GroupNorm is not modulated)
        o1, raw1 = torch.empty_like(o0), torch.full((n, hw, 2 * C), float("nan"), device=dev, dtype=torch.float16)
        ops.groupnorm(x1, x2, g, b, o1, ws, silu=True, raw_f16=raw1, split_raw=True, **kw)
        x = torch.cat([x1, x2], -1)
        assert torch.equal(o1, o0) and torch.equal(raw1[..., :C], raw0)
        assert torch.equal(raw1[..., C:], (x - raw0.float()).half())
    o2 = torch.full((n, hw, 2 * C), float("nan"), device=dev, dtype=torch.float16)
    ops.groupnorm(x1, x2, g, b, o2, ws, silu=True, split_out=True)
    o0 = torch.empty((n, hw, C), device=dev, dtype=torch.float16)
    ops.groupnorm(x1, x2, g, b, o0, ws, silu=True)
    y = F.silu(F.group_norm(torch.cat([x1, x2], -1).transpose(1, 2), 32, g, b, 1e-5)).transpose(1, 2)
    assert torch.equal(o2[..., :C], o0)
    assert rel_l2(o2[..., :C].float() + o2[..., C:].float(), y) < 2e-6 and rel_l2(o0, y) > 1e-4
    # consumer: GEMM with duplicated weights
    w = _rand((64, C), dev, 68, 0.2).half()
    a_plain = raw0.view(n * hw, C)
    a_split = raw1.view(n * hw, 2 * C)
    out_p, out_s = torch.empty((n * hw, 64), device=dev), torch.empty((n * hw, 64), device=dev)
    ops.gemm(a_plain, w, out_f32=out_p)
    ops.gemm(a_split, torch.cat([w, w], 1).contiguous(), out_f32=out_s)
    ref = torch.cat([x1, x2], -1).view(n * hw, C).double() @ w.double().T
    print(f"\nsplit-precision GEMM operand: plain f16 {rel_l2(out_p.double(), ref):.2e}, [hi | lo] {rel_l2(out_s.double(), ref):.2e}")
    assert rel_l2(out_s.double(), ref) < 5e-6 < 5e-5 < rel_l2(out_p.double(), ref)
    # input cast of the stem
    xa, xb = _rand((n, 4, 7, 5), dev, 71, 30.0), _rand((n, 7, 7, 5), dev, 72)
    sc = _rand((n,), dev, 73).abs() + 0.5
    o = torch.full((n, 35, 64), float("nan"), device=dev, dtype=torch.float16)
    ops.nchw_to_nhwc_f16(xa, xb, o, scale=sc, split=True)
    v = torch.cat([xa * sc[:, None, None, None], xb], 1).permute(0, 2, 3, 1).reshape(n, 35, 11)
    assert torch.equal(o[..., :11], v.half()) and torch.equal(o[..., 11:22], (v - v.half().float()).half())
    assert o[..., 22:].abs().max() == 0


@pytest.mark.parametrize("n,hw,c1,c2,silu,dense", GN_CASES)
def test_groupnorm(dev, n, hw, c1, c2, silu, dense):
    from seva import ops
    C = c1 + c2
    x1 = _rand((n, hw, c1), dev, 1) * 2 + 0.5
    x2 = _rand((n, hw, c2), dev, 2) - 1.0 if c2 else None
    gamma, beta = 1 + 0.1 * _rand((C,), dev, 3), 0.1 * _rand((C,), dev, 4)
    dmap = _rand((n, hw, 6), dev, 5) if dense else None
    dw = _rand((2 * C, 6), dev, 6, 0.3) if dense else None
    db = _rand((2 * C,), dev, 7, 0.1) if dense else None
    out = torch.full((n, hw, C), float("nan"), device=dev, dtype=torch.float16)
    ws = ops.groupnorm_workspace(n, dev)
    eps = 1e-6 if not silu else 1e-5
    ops.groupnorm(x1, x2, gamma, beta, out, ws, eps=eps, silu=silu, dense=dmap, dense_w=dw, dense_b=db)
    x = torch.cat([x1, x2], -1) if c2 else x1
    ref = F.group_norm(x.transpose(1, 2), 32, gamma, beta, eps).transpose(1, 2)
    if silu:
        ref = F.silu(ref)
    if dense:
        d = dmap @ dw.T + db
        ref = ref * (1 + d[..., :C]) + d[..., C:]
    err = rel_l2(out, ref)
    assert err < 6e-4, f"rel_l2 {err}"
    assert (out.float() - ref).abs().max() < 2e-3 * ref.abs().max()


def _block_stats(o32, M, N):
    """fp64 reference of seva_gemm_desc.ch_stats: [ceil(M / 64)][2][N] sums / sums of squares of the fp32 output."""
    nb = (M + 63) // 64
    pad = torch.zeros((nb * 64, N), dtype=torch.float64, device=o32.device)
    pad[:M] = o32.double()
    blk = pad.view(nb, 64, N)
    return torch.stack((blk.sum(1), (blk * blk).sum(1)), 1)


@pytest.mark.parametrize("M,N,K,bn", [(640, 320, 320, "0"), (200, 320, 64, "0"), (1000, 640, 128, "0"), (333, 256, 64, "0"),
                                      (640, 320, 320, "128")])
def test_gemm_channel_stats(dev, M, N, K, bn, knobs):
    """Epilogue-emitted GroupNorm statistics of a GEMM output (bias + row_add + residual): exact on integer data,
    M tails (rows past M contribute nothing), 128- and 160-wide tiles; the output itself is unchanged by the option."""
    from seva import ops
    if bn != "0":
        knobs(gemm_bn=int(bn))
    a, w = _ints((M, K), -3, 3, dev, 1), _ints((N, K), -2, 2, dev, 2)
    bias, res = _ints((N,), -3, 3, dev, 3), _ints((M, N), -5, 5, dev, 4)
    rpg = 48
    radd = _ints(((M + rpg - 1) // rpg, N), -2, 2, dev, 5)
    o_ref = torch.empty((M, N), device=dev)
    ops.gemm(a.half(), w.half(), bias=bias, row_add=radd, rows_per_group=rpg, residual=res, out_f32=o_ref)
    o32 = torch.full((M, N), float("nan"), device=dev)
    st = torch.full(ops.channel_stats_shape(M, N), float("nan"), device=dev)
    ops.gemm(a.half(), w.half(), bias=bias, row_add=radd, rows_per_group=rpg, residual=res, out_f32=o32, ch_stats=st)
    assert torch.equal(o32, o_ref)
    ref = _block_stats(o32, M, N)
    assert torch.equal(st.double(), ref), f"max diff {(st.double() - ref).abs().max()}"  # integers: exact in fp32
    # random data: fp32 sums of 64 values against fp64
    a, w = _rand((M, K), dev, 6), _rand((N, K), dev, 7, 0.2)
    ops.gemm(a.half(), w.half(), bias=bias, out_f32=o32, ch_stats=st)
    ref = _block_stats(o32, M, N)
    assert (st.double() - ref).abs().max() < 1e-5 * ref.abs().max()


@pytest.mark.parametrize("n,ih,iw,cin,cout,stride,up", [(3, 16, 16, 64, 320, 1, False), (2, 16, 32, 128, 256, 2, False),
                                                         (2, 8, 8, 64, 128, 1, True), (5, 8, 8, 64, 160, 1, False)])
def test_conv_channel_stats(dev, n, ih, iw, cin, cout, stride, up):
    from seva import ops
    from seva._engine import pack_conv3x3
    x = _ints((n, ih, iw, cin), -2, 2, dev, 1).half()
    wc = _ints((cout, cin, 3, 3), -1, 1, dev, 2)
    bias = _ints((cout,), -3, 3, dev, 3)
    eh, ew = (2 * ih, 2 * iw) if up else (ih, iw)
    oh, ow = (eh - 1) // stride + 1, (ew - 1) // stride + 1
    M = n * oh * ow
    emb = _ints((n, cout), -2, 2, dev, 4)
    o32 = torch.full((M, cout), float("nan"), device=dev)
    st = torch.full(ops.channel_stats_shape(M, cout), float("nan"), device=dev)
    ops.conv3x3(x, pack_conv3x3(wc).half().to(dev), stride=stride, upsample=up, bias=bias, row_add=emb, rows_per_group=oh * ow,
                out_f32=o32, ch_stats=st)
    xi = x.float().permute(0, 3, 1, 2)
    if up:
        xi = F.interpolate(xi, scale_factor=2, mode="nearest")
    ref_o = F.conv2d(xi, wc, bias, stride=stride, padding=1).permute(0, 2, 3, 1).reshape(M, cout) + emb.repeat_interleave(oh * ow, 0)
    assert torch.equal(o32, ref_o)
    assert torch.equal(st.double(), _block_stats(o32, M, cout))


@pytest.mark.parametrize("n,ih,iw,cin,cout,k2,stats", [(3, 12, 10, 64, 128, 64, False), (2, 16, 16, 128, 320, 192, True),
                                                       (42, 18, 18, 128, 160, 128, False), (2, 40, 36, 64, 64, 128, True),
                                                       (16, 24, 24, 64, 320, 256, False)])
def test_conv3x3_with_folded_second_operand(dev, n, ih, iw, cin, cout, k2, stats):
    """seva_gemm_desc.a2 (MODE 3): out = conv3x3(x, w_conv) + a2 @ w_2^T (+ bias) in ONE accumulation, w = [w_conv | w_2].
    Integer data: bit-exact against torch, including the epilogue-emitted GroupNorm statistics; every tile shape of the mode
    (160 x 160 for big wide launches, 128 x 160, 128 x 128) is reached by the parametrisation."""
    from seva import ops
    from seva._engine import pack_conv3x3
    M = n * ih * iw
    x = _ints((n, ih, iw, cin), -2, 2, dev, 1).half()
    wc = _ints((cout, cin, 3, 3), -1, 1, dev, 2)
    a2 = _ints((M, k2), -2, 2, dev, 3).half()
    w2 = _ints((cout, k2), -1, 1, dev, 4)
    bias = _ints((cout,), -3, 3, dev, 5)
    wcat = torch.cat([pack_conv3x3(wc).half().to(dev), w2.half()], 1).contiguous()
    o = torch.full((M, cout), float("nan"), device=dev)
    st = torch.full(ops.channel_stats_shape(M, cout), float("nan"), device=dev) if stats and (ih * iw) % 64 == 0 else None
    ops.conv3x3(x, wcat, bias=bias, a2=a2, out_f32=o, ch_stats=st)
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), wc, bias, padding=1).permute(0, 2, 3, 1).reshape(M, cout) + a2.float() @ w2.T
    assert torch.equal(o, ref), f"max diff {(o - ref).abs().max()}"
    if st is not None:
        assert torch.equal(st.double(), _block_stats(o, M, cout))


def test_conv_splitk_workspace_too_small_is_an_error(dev):
    """A split-K workspace sized for a smaller batch is refused (it used to fall back silently to the unsplit kernel, which
    made a sample's reduction order depend on the batch it was launched in and on the caller's workspace)."""
    from seva import ops
    from seva._engine import pack_conv3x3
    from seva._native import SevaNativeError
    n, cin, cout = 42, 1280, 1280
    x = _ints((n, 9, 9, cin), -2, 2, dev, 1).half()
    w = pack_conv3x3(_ints((cout, cin, 3, 3), -1, 1, dev, 2)).half().to(dev)
    out = torch.empty((n * 81, cout), device=dev)
    small = ops.splitk_workspace(2 * 81, cout, dev)[: 16384 + 4 * 128 * 160].contiguous()  # 4 tile slots; the launch has 27 x 8 tiles
    with pytest.raises(SevaNativeError, match="splitk_ws too small"):
        ops.conv3x3(x, w, out_f32=out, splitk_ws=small)
    ops.conv3x3(x, w, out_f32=out, splitk_ws=ops.splitk_workspace(n * 81, cout, dev))  # the right size runs
    torch.cuda.synchronize()


@pytest.mark.parametrize("n,ih,iw,cin,cout,stride", [(42, 9, 9, 1280, 1280, 1), (5, 9, 9, 256, 320, 1), (42, 18, 18, 640, 1280, 2),
                                                      (3, 8, 16, 128, 128, 1), (1, 4, 4, 2560, 160, 1)])
def test_conv_splitk_small_images(dev, n, ih, iw, cin, cout, stride):
    """Split-K = 2 convolution over small images (seva_gemm_desc.splitk_ws): exact on integer data with bias, row_add and
    residual; on random data equal to the unsplit kernel up to the one changed association; the flags are left zero (the
    workspace is reusable by the next launch / graph replay) and the error slot stays 0; batch composition does not matter."""
    from seva import ops
    from seva._engine import pack_conv3x3
    oh, ow = (ih - 1) // stride + 1, (iw - 1) // stride + 1
    M = n * oh * ow
    ws = ops.splitk_workspace(M, cout, dev)
    x = _ints((n, ih, iw, cin), -2, 2, dev, 1).half()
    wc = _ints((cout, cin, 3, 3), -1, 1, dev, 2)
    bias, emb, res = _ints((cout,), -3, 3, dev, 3), _ints((n, cout), -2, 2, dev, 4), _ints((M, cout), -4, 4, dev, 5)
    wp = pack_conv3x3(wc).half().to(dev)
    o = torch.full((M, cout), float("nan"), device=dev)
    # a SECOND data set alternates with the first through the ONE workspace (a hand-off that consumed a stale or early partial
    # would go unnoticed if every launch wrote the same partial tiles -- the pattern that once hid exactly that bug)
    x2 = _ints((n, ih, iw, cin), -2, 2, dev, 11).half()
    wc2 = _ints((cout, cin, 3, 3), -1, 1, dev, 12)
    wp2 = pack_conv3x3(wc2).half().to(dev)
    refs = []
    for xx, ww in ((x, wc), (x2, wc2)):
        r = F.conv2d(xx.float().permute(0, 3, 1, 2), ww, bias, stride=stride, padding=1).permute(0, 2, 3, 1).reshape(M, cout)
        refs.append(r + emb.repeat_interleave(oh * ow, 0) + res)
    for it in range(6):
        o.fill_(float("nan"))
        ops.conv3x3((x, x2)[it & 1], (wp, wp2)[it & 1], stride=stride, bias=bias, row_add=emb, rows_per_group=oh * ow,
                    residual=res, out_f32=o, splitk_ws=ws)
        assert torch.equal(o, refs[it & 1]), f"launch {it}: max diff {(o - refs[it & 1]).abs().max()}"
        assert int(ws[:16384].view(torch.int32).abs().sum()) == 0  # flags re-armed, error slot (16383) untouched
    ops.check_handoffs()  # what the product calls at the end of a trajectory: raises if any consumer gave up
    xr, wr = _rand((n, ih, iw, cin), dev, 6).half(), (pack_conv3x3(_rand((cout, cin, 3, 3), dev, 7, 0.05).cpu()).half().to(dev))
    o1, o2 = torch.empty_like(o), torch.empty_like(o)
    ops.conv3x3(xr, wr, stride=stride, bias=bias, residual=res, out_f32=o1)
    ops.conv3x3(xr, wr, stride=stride, bias=bias, residual=res, out_f32=o2, splitk_ws=ws)
    err = rel_l2(o2, o1)
    print(f"\nsplit-K conv {n}x{ih}x{iw} {cin}->{cout} s{stride}: vs unsplit {err:.2e}")
    assert err < 2e-6
    # one sample alone (batch of one) gives bitwise the rows it has inside the batch
    o3 = torch.empty((oh * ow, cout), device=dev)
    ops.conv3x3(xr[n - 1:], wr, stride=stride, bias=bias, residual=res[(n - 1) * oh * ow:], out_f32=o3, splitk_ws=ws)
    assert torch.equal(o3, o2[(n - 1) * oh * ow:])


@pytest.mark.parametrize("n,hw,c1,c2,dense", [(3, 256, 320, 0, True), (2, 1024, 128, 0, False), (2, 64, 640, 320, True),
                                              (5, 5184, 320, 320, False)])
def test_groupnorm_with_producer_statistics(dev, n, hw, c1, c2, dense):
    """GroupNorm fed with the statistics its producers emitted (stats1 / stats2) against the separate statistics pass and
    against torch; batch composition does not change a sample's result (bitwise)."""
    from seva import ops
    C = c1 + c2

    def produce(c, seed, nn=n, first=0):
        # the producer: a GEMM with K = 64 whose fp32 output is the GroupNorm input
        a = _rand((n * hw, 64), dev, seed)[first * hw:(first + nn) * hw].contiguous()
        w = _rand((c, 64), dev, seed + 1, 0.3)
        o = torch.empty((nn * hw, c), device=dev)
        st = torch.empty(ops.channel_stats_shape(nn * hw, c), device=dev)
        ops.gemm(a.half(), w.half(), bias=_rand((c,), dev, seed + 2), out_f32=o, ch_stats=st)
        return o.view(nn, hw, c), st

    x1, s1 = produce(c1, 10)
    x2, s2 = produce(c2, 20) if c2 else (None, None)
    gamma, beta = 1 + 0.1 * _rand((C,), dev, 3), 0.1 * _rand((C,), dev, 4)
    kw = dict(dense=_rand((n, hw, 6), dev, 5), dense_w=_rand((2 * C, 6), dev, 6, 0.3), dense_b=_rand((2 * C,), dev, 7, 0.1)) if dense else {}
    ws = ops.groupnorm_workspace(n, dev)
    o_pass = torch.empty((n, hw, C), device=dev, dtype=torch.float16)
    o_st = torch.full_like(o_pass, float("nan"))
    ops.groupnorm(x1, x2, gamma, beta, o_pass, ws, silu=True, **kw)
    ops.groupnorm(x1, x2, gamma, beta, o_st, ws, silu=True, stats1=s1, stats2=s2, **kw)
    x = torch.cat([x1, x2], -1) if c2 else x1
    ref = F.silu(F.group_norm(x.transpose(1, 2), 32, gamma, beta, 1e-5).transpose(1, 2))
    if dense:
        d = kw["dense"] @ kw["dense_w"].T + kw["dense_b"]
        ref = ref * (1 + d[..., :C]) + d[..., C:]
    e_pass, e_st = rel_l2(o_pass, ref), rel_l2(o_st, ref)
    print(f"\ngroupnorm n={n} hw={hw} c={c1}+{c2}: statistics pass {e_pass:.2e}, producer statistics {e_st:.2e}, "
          f"max |diff| between them {float((o_pass.float() - o_st.float()).abs().max()):.2e}")
    assert e_st < 6e-4 and rel_l2(o_st, o_pass) < 3e-4
    # batch invariance: the last sample alone, produced and normalised as a batch of one
    y1, t1 = produce(c1, 10, 1, n - 1)
    y2, t2 = produce(c2, 20, 1, n - 1) if c2 else (None, None)
    assert torch.equal(y1[0], x1[n - 1])
    o_one = torch.empty((1, hw, C), device=dev, dtype=torch.float16)
    kw1 = dict(kw, dense=kw["dense"][n - 1:]) if dense else {}
    ops.groupnorm(y1, y2, gamma, beta, o_one, ops.groupnorm_workspace(1, dev), silu=True, stats1=t1, stats2=t2, **kw1)
    assert torch.equal(o_one[0], o_st[n - 1])


@pytest.mark.parametrize("rows,c", [(10, 64), (1001, 320), (333, 640), (50, 1280), (7, 128)])
def test_layernorm(dev, rows, c):
    from seva import ops
    x = _rand((rows, c), dev, 1) * 3 + 1
    g, b = 1 + 0.1 * _rand((c,), dev, 2), 0.1 * _rand((c,), dev, 3)
    out = torch.empty((rows, c), device=dev, dtype=torch.float16)
    ops.layernorm(x, g, b, out)
    ref = F.layer_norm(x, (c,), g, b, 1e-5)
    assert rel_l2(out, ref) < 6e-4
    assert (out.float() - ref).abs().max() < 4e-3


def test_layout_and_elementwise(dev):
    from seva import ops
    n, h, w = 3, 7, 5
    x1, x2 = _rand((n, 4, h, w), dev, 1), _rand((n, 7, h, w), dev, 2)
    sc = _rand((n,), dev, 3).abs() + 0.5
    out = torch.full((n, h * w, 64), float("nan"), device=dev, dtype=torch.float16)
    ops.nchw_to_nhwc_f16(x1, x2, out, scale=sc)
    ref = torch.cat([x1 * sc[:, None, None, None], x2], 1).permute(0, 2, 3, 1).reshape(n, h * w, 11)
    assert torch.equal(out[..., :11], ref.half()) and out[..., 11:].abs().max() == 0
    y = _rand((n, h * w, 8), dev, 4)
    o = torch.empty((n, 4, h, w), device=dev)
    ops.nhwc_to_nchw_f32(y, o)
    assert torch.equal(o, y[..., :4].reshape(n, h, w, 4).permute(0, 3, 1, 2))
    a, b = _rand((10, 64), dev, 5), _rand((10, 32), dev, 6)
    cc = torch.empty((10, 96), device=dev, dtype=torch.float16)
    ops.cast_concat_f16(a, b, cc)
    assert torch.equal(cc, torch.cat([a, b], 1).half())
    src = _rand((n, 6, 12, 10), dev, 7)
    for oh, ow in ((6, 5), (3, 3), (12, 10), (1, 1), (24, 20)):
        ob = torch.empty((n, oh * ow, 6), device=dev)
        ops.bilinear_to_nhwc(src, ob, oh, ow)
        refb = F.interpolate(src, size=(oh, ow), mode="bilinear", align_corners=True)
        assert torch.allclose(ob.view(n, oh, ow, 6).permute(0, 3, 1, 2), refb, atol=2e-6), (oh, ow)
    t = torch.tensor([999, 979, 500, 19, 0], device=dev)
    for dim in (320, 64):
        half = dim // 2
        freqs = torch.exp(-math.log(10000) * torch.arange(half, dtype=torch.float32) / half).to(dev)
        te = torch.empty((5, dim), device=dev, dtype=torch.float16)
        ops.timestep_embedding_f16(t, freqs, te)
        args = t[:, None].float().cpu() * freqs.cpu()[None]
        reft = torch.cat([torch.cos(args), torch.sin(args)], -1)
        assert (te.float().cpu() - reft).abs().max() < 1.5e-3
    s = torch.empty((10, 64), device=dev, dtype=torch.float16)
    ops.silu_f16(a, s)
    assert (s.float() - F.silu(a)).abs().max() < 2e-3
    ad = torch.empty_like(a)
    ops.add_f32(a, a * 2, ad)
    assert torch.equal(ad, a + a * 2)


def test_sampler_elementwise(dev):
    from seva import ops
    T, c, h, w = 4, 4, 6, 5
    x = _rand((T, c, h, w), dev, 1) * 10
    rep = _rand((T, c + 1, h, w), dev, 2)
    rep[:, c] = (torch.arange(T, device=dev) % 2).float()[:, None, None]
    o = torch.empty_like(x)
    ops.replace_blend(x, rep, o)
    m = rep[:, c:]
    assert torch.equal(o, x * (1 - m) + rep[:, :c] * m)
    net = _rand((T, c, h, w), dev, 3)
    co, cs = -torch.rand(T, device=dev) * 5, torch.ones(T, device=dev)
    ops.denoiser_combine(net, x, co, cs, o)
    assert torch.allclose(o, net * co[:, None, None, None] + x * cs[:, None, None, None], rtol=1e-6, atol=1e-6)
    eps, ns = _rand((T, c, h, w), dev, 4), torch.rand(T, device=dev)
    ops.add_noise(x, eps, ns, o)
    assert torch.allclose(o, x + eps * ns[:, None, None, None], rtol=1e-6, atol=1e-6)
    den2 = _rand((2 * T, c, h, w), dev, 5)
    scale = torch.tensor([1.2, 2.0, 1.5, 2.0], device=dev)
    sh, dt = torch.full((T,), 3.1, device=dev), torch.full((T,), -1.7, device=dev)
    ops.cfg_euler(x, den2, scale, sh, dt, o)
    u, cnd = den2.chunk(2)
    den = u + scale[:, None, None, None] * (cnd - u)
    ref = x + dt[:, None, None, None] * ((x - den) / sh[:, None, None, None])
    assert torch.allclose(o, ref, rtol=1e-5, atol=1e-5)


def test_errors_are_loud(dev):
    from seva import ops
    from seva._native import SevaNativeError
    a = torch.zeros((8, 100), device=dev, dtype=torch.float16)  # K not a multiple of 64
    w = torch.zeros((8, 100), device=dev, dtype=torch.float16)
    with pytest.raises(SevaNativeError):
        ops.gemm(a, w, out_f32=torch.empty((8, 8), device=dev))
    with pytest.raises(SevaNativeError):
        ops.gemm(a.cpu(), w.cpu(), out_f32=torch.empty((8, 8)))


@pytest.mark.parametrize("slope", [3.0, 0.05, -2.0])
def test_attention_running_max_paths(dev, slope):
    """Scores that grow along the key axis force the online-softmax rescale on every tile
    (slope 3 -> +192 per 64-key tile), exercise the deferred-rescale path (slope 0.05: the
    exponent reference lags by < 2^8 for several tiles before one rescale) and the never-rescale
    path (negative slope: the first tile holds the max)."""
    from seva import ops
    B, H, L = 2, 2, 512
    C = 64 * H
    g = torch.Generator().manual_seed(11)
    q = torch.randn(B, L, C, generator=g) * 0.5
    k = torch.randn(B, L, C, generator=g) * 0.5
    v = torch.randn(B, L, C, generator=g)
    # dimension 0 of every head carries a ramp: q[...,0] = 1, k[...,0] = 8*slope*key  (logit scale 1/8)
    q[:, :, ::64] = 1.0
    k[:, :, ::64] = (torch.arange(L, dtype=torch.float32) * slope * 8.0)[None, :, None]
    qh, kh, vh = q.half().to(dev), k.half().to(dev), v.half().to(dev)
    out = torch.empty(B, L, C, device=dev, dtype=torch.float16)
    ops.attention(qh, kh, vh, out, nb0=B, nb1=1, heads=H, lq=L, lk=L, q_strides=(L * C, 0, C),
                  k_strides=(L * C, 0, C), o_strides=(L * C, 0, C))
    r = lambda t: t.float().view(B, L, H, 64).transpose(1, 2)
    ref = _attn_ref(r(qh), r(kh), r(vh), 0.125).transpose(1, 2).reshape(B, L, C)
    err = rel_l2(out, ref)
    assert torch.isfinite(out.float()).all() and err < 2e-3, f"slope {slope}: rel_l2 {err}"


# ---- conditioning geometry ----------------------------------------------------------------------
def test_plucker_matches_reference_golden(dev):
    """seva.geometry.get_plucker_coordinates (HIP) against vectors produced by the reference itself."""
    from conftest import load_golden
    from seva import geometry as G
    g = load_golden("g8_plucker")
    t = lambda k: torch.as_tensor(g[k]).clone()
    a = G.get_plucker_coordinates(t("a_w2c")[0].to(dev), t("a_w2c").to(dev), None, target_size=[9, 9])
    assert a.device.type == "cuda" and torch.allclose(a.cpu(), t("a_out"), atol=3e-6, rtol=0)
    # host inputs: computed on the GPU, returned on the host; the intrinsics argument is rescaled in place like the reference's
    Kb = t("b_K")
    b = G.get_plucker_coordinates(t("b_w2c")[1], t("b_w2c"), Kb, target_size=[12, 20])
    assert b.device.type == "cpu" and torch.allclose(b, t("b_out"), atol=3e-6, rtol=0)
    assert torch.allclose(Kb[:, 0], t("b_K")[:, 0] * 20) and torch.allclose(Kb[:, 1], t("b_K")[:, 1] * 12)
    c = G.get_plucker_coordinates(t("c_w2c")[0].to(dev), t("c_w2c").to(dev), t("c_K").to(dev), target_size=[8, 6])
    assert torch.allclose(c.cpu(), t("c_out"), atol=3e-6, rtol=0)
    bad = t("c_K") * 50.0
    with pytest.raises(AssertionError):
        G.get_plucker_coordinates(t("c_w2c")[0], t("c_w2c"), bad, target_size=[8, 6])


def test_value_dict_and_cond_assembly(dev):
    from conftest import load_golden
    from oracle import geometry_ref as R
    from seva import conditioning as Cn
    g = load_golden("g8_value_dict")
    for tag in ("a", "b", "c"):
        H, W = (int(v) for v in g[f"{tag}_HW"])
        c2w_in = torch.as_tensor(g[f"{tag}_c2w_in"]).clone()
        vd = Cn.get_value_dict((H, W), [int(i) for i in g[f"{tag}_in_idx"]], c2w_in, torch.as_tensor(g[f"{tag}_K"]).clone(),
                               R.to_hom_pose(c2w_in), 2.0, device=dev)
        assert torch.equal(vd["cond_frames_mask"], torch.as_tensor(g[f"{tag}_mask"]))
        assert torch.allclose(vd["c2w"], torch.as_tensor(g[f"{tag}_c2w"]), atol=1e-6, rtol=0)
        assert torch.allclose(vd["plucker_coordinate"].cpu(), torch.as_tensor(g[f"{tag}_plucker"]), atol=4e-6, rtol=0)
    # channel assembly against the oracle's restatement of do_sample
    T, h, w = 5, 8, 8
    pl = vd_pl = torch.randn(T, 6, h, w)
    mask = torch.tensor([True, False, False, True, False])
    lat, clip = torch.randn(2, 4, h, w), torch.randn(1024)
    c_ref, uc_ref = R.assemble_cond(lat, clip, mask, pl)
    c, uc = Cn.assemble_cond(lat, clip, mask, pl.to(dev))
    for k in c_ref:
        assert torch.equal(c[k].cpu(), c_ref[k]), k
        assert torch.equal(uc[k].cpu(), uc_ref[k]), k


@pytest.mark.parametrize("M,C", [(300, 64), (1000, 128), (515, 256), (2049, 320), (128, 320), (77, 320)])
@pytest.mark.parametrize("with_res", [True, False])
def test_ff_fused_vs_two_kernels_and_fp32(dev, M, C, with_res, knobs):
    """seva_ff_fused_f16 (GEGLU -> FF2 in one kernel, hidden activations in registers) against (a) the two-kernel path it
    replaces -- same f16 rounding of the hidden tensor, so they agree to fp32 accumulation-order noise -- and (b) fp32 torch."""
    from seva import ops
    from seva._engine import interleave_geglu
    g = torch.Generator().manual_seed(91)
    a = torch.randn(M, C, generator=g).half().to(dev)
    w1 = (torch.randn(8 * C, C, generator=g) * C ** -0.5).half().to(dev)
    b1 = (0.3 * torch.randn(8 * C, generator=g)).to(dev)
    w2 = (torch.randn(C, 4 * C, generator=g) * (4 * C) ** -0.5).half().to(dev)
    b2 = (0.3 * torch.randn(C, generator=g)).to(dev)
    res = torch.randn(M, C, generator=g).to(dev) if with_res else None
    wi, bi = interleave_geglu(w1, b1)
    o32 = torch.full((M, C), float("nan"), device=dev)
    o16 = torch.full((M, C), float("nan"), device=dev, dtype=torch.float16)
    ops.ff_fused(a, wi, bi, w2, b2, residual=res, out_f32=o32, out_f16=o16)
    hid = torch.empty((M, 4 * C), device=dev, dtype=torch.float16)
    ops.gemm(a, wi, bias=bi, out_f16=hid, geglu=True)
    two = torch.empty((M, C), device=dev)
    ops.gemm(hid, w2, bias=b2, residual=res, out_f32=two)
    y = a.float() @ w1.float().T + b1
    ref = (y[:, : 4 * C] * F.gelu(y[:, 4 * C:])) @ w2.float().T + b2 + (res if with_res else 0)
    torch.cuda.synchronize()
    assert torch.isfinite(o32).all()
    e_two, e_ref = rel_l2(o32, two), rel_l2(o32, ref)
    assert e_two < 1e-4, e_two  # bias-first accumulation, f16 ties
    assert e_ref < 1e-3, e_ref
    assert torch.equal(o16, o32.half())
    o16b = torch.full((M, C), float("nan"), device=dev, dtype=torch.float16)  # f16-only output (time-mix tail)
    ops.ff_fused(a, wi, bi, w2, b2, residual=res, out_f16=o16b)
    assert torch.equal(o16b, o16)
    # LayerNorm folded into the prologue: vs LayerNorm kernel -> fused kernel (same math, different summation order)
    x = (torch.randn(M, C, generator=g) * 2 + 0.3).to(dev)
    gm, bt = (1 + 0.1 * torch.randn(C, generator=g)).to(dev), (0.1 * torch.randn(C, generator=g)).to(dev)
    a_ln = torch.empty((M, C), device=dev, dtype=torch.float16)
    ops.layernorm(x, gm, bt, a_ln)
    want = torch.empty((M, C), device=dev)
    ops.ff_fused(a_ln, wi, bi, w2, b2, residual=res, out_f32=want)
    got = torch.full((M, C), float("nan"), device=dev)
    ops.ff_fused(None, wi, bi, w2, b2, residual=res, out_f32=got, ln_x=x, ln_gamma=gm, ln_beta=bt)
    e_ln = rel_l2(got, want)
    assert torch.isfinite(got).all() and e_ln < 2e-4, e_ln  # f16 roundings of the normalised row flip at ties only
