"""fp8 (OCP e4m3) operators of BASELINE config 5 through the C-ABI: `seva_gemm_fp8` (plain, GEGLU, conv3x3) on the
block-scaled MFMA, and the e4m3-emitting producers (`seva_layernorm_fp8`, `seva_groupnorm_f16(out_f8=...)`).

GEMM / conv are checked BIT-EXACTLY on integer data: small integers are exact in e4m3, the per-channel weight scale
is a power of two carried by the MFMA's E8M0 block scale, and fp32 accumulation of such products is exact -- any
operand-map, k-permutation or scale-routing mistake shows as a mismatch.  Quantising producers are checked against
torch's own e4m3 cast of an fp32 reference (equal up to one e4m3 ulp at rounding ties of the fp32 arithmetic)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from conftest import rel_l2

U8 = torch.uint8


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from seva import _native
    _native.load()
    return torch.device("cuda:0")


def _ints(shape, lo, hi, dev, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(lo, hi + 1, shape, generator=g).float().to(dev)


def _f8(x):
    from seva import ops
    return ops.to_fp8(x)


def _wq(w_int, dev, seed):
    """integer weights + random power-of-two row scales -> (e4m3 bytes, scale bytes, de-quantised fp32 weights)"""
    g = torch.Generator().manual_seed(seed)
    e = torch.randint(-3, 4, (w_int.shape[0],), generator=g).to(dev)
    return _f8(w_int), (e + 127).to(U8), w_int * torch.exp2(e.float())[:, None]


@pytest.mark.parametrize("chunks", [-1, 1])
@pytest.mark.parametrize("M,N,K", [(300, 640, 640), (1000, 1920, 640), (257, 1280, 256), (2049, 320, 1280),
                                   (64, 640, 128), (515, 160, 384)])
def test_gemm_fp8_exact(dev, M, N, K, chunks, knobs):
    from seva import ops
    knobs(gemm_chunks=chunks)
    a = _ints((M, K), -4, 4, dev, 1)
    w8, wexp, wf = _wq(_ints((N, K), -3, 3, dev, 2), dev, 3)
    bias = _ints((N,), -5, 5, dev, 4)
    res = _ints((M, N), -9, 9, dev, 5)
    rpg = 100
    radd = _ints(((M + rpg - 1) // rpg, N), -3, 3, dev, 6)
    ref = a @ wf.T + bias + res + radd.repeat_interleave(rpg, 0)[:M]
    o32 = torch.full((M, N), float("nan"), device=dev)
    o16 = torch.full((M, N), float("nan"), device=dev, dtype=torch.float16)
    ops.gemm(_f8(a), w8, w_exp=wexp, bias=bias, row_add=radd, rows_per_group=rpg, residual=res, out_f32=o32, out_f16=o16)
    torch.cuda.synchronize()
    assert torch.equal(o32, ref), f"max diff {(o32 - ref).abs().max()}"
    assert torch.equal(o16.float(), ref.half().float())
    # f16-only output: the ASYNC schedule (scale bytes ride in through the LDS slot), with the q-column scale
    o16b = torch.full((M, N), float("nan"), device=dev, dtype=torch.float16)
    csn = N // 4 // 4 * 4
    for _ in range(2):
        ops.gemm(_f8(a), w8, w_exp=wexp, bias=bias, out_f16=o16b, col_scale=0.5, col_scale_n=csn)
    ref2 = a @ wf.T + bias
    ref2[:, :csn] *= 0.5
    assert torch.equal(o16b.float(), ref2.half().float())


@pytest.mark.parametrize("M,C,K", [(300, 640, 640), (1000, 320, 1280), (131, 1280, 128)])
def test_geglu_fp8(dev, M, C, K, knobs):
    """GEGLU epilogue on the fp8 kernel; the e4m3 hidden output equals torch's e4m3 cast of the fp32 output of the same launch."""
    from seva import ops
    from seva._engine import interleave_geglu
    knobs(gemm_chunks=1)
    g = torch.Generator().manual_seed(7)
    a = torch.randn(M, K, generator=g).to(dev)
    w = (torch.randn(8 * C, K, generator=g) * K ** -0.5).to(dev)
    b = torch.randn(8 * C, generator=g).to(dev)
    wi, bi = interleave_geglu(w, b)
    w8, wexp = ops.quantize_weight_fp8(wi)
    a8 = _f8(a)
    o32 = torch.full((M, 4 * C), float("nan"), device=dev)
    o8 = torch.zeros((M, 4 * C), device=dev, dtype=U8)
    ops.gemm(a8, w8, w_exp=wexp, bias=bi, out_f32=o32, out_f8=o8, geglu=True)
    # reference on the operands the kernel sees (de-quantised), original row order
    aq = a8.view(torch.float8_e4m3fn).float()
    wq_i = ops.dequantize_weight_fp8(w8, wexp)
    y = aq @ wq_i.T + bi
    yv = y.view(M, -1, 2, 32)
    ref = (yv[:, :, 0] * F.gelu(yv[:, :, 1])).reshape(M, 4 * C)
    assert torch.isfinite(o32).all() and rel_l2(o32, ref) < 2e-5
    assert torch.equal(o8, _f8(o32))
    o8b = torch.zeros_like(o8)  # e4m3-only output: ASYNC schedule
    ops.gemm(a8, w8, w_exp=wexp, bias=bi, out_f8=o8b, geglu=True)
    assert torch.equal(o8b, o8)


@pytest.mark.parametrize("n,ih,iw,cin,cout,stride", [(2, 9, 7, 128, 64, 1), (1, 12, 10, 256, 96, 2), (3, 8, 8, 640, 160, 1),
                                                     (42, 9, 9, 128, 320, 1)])
def test_conv3x3_fp8_exact(dev, n, ih, iw, cin, cout, stride):
    from seva import ops
    from seva._engine import pack_conv3x3
    x = _ints((n, cin, ih, iw), -3, 3, dev, 1)
    w = _ints((cout, cin, 3, 3), -2, 2, dev, 2)
    g = torch.Generator().manual_seed(3)
    e = torch.randint(-2, 3, (cout,), generator=g).to(dev)
    wf = w * torch.exp2(e.float())[:, None, None, None]
    bias = _ints((cout,), -4, 4, dev, 3)
    ref = F.conv2d(x, wf, bias, stride=stride, padding=1)
    oh, ow = ref.shape[-2:]
    temb = _ints((n, cout), -2, 2, dev, 4)
    res = _ints((n, oh * ow, cout), -5, 5, dev, 5)
    ref = ref + temb[:, :, None, None] + res.view(n, oh, ow, cout).permute(0, 3, 1, 2)
    w8 = _f8(pack_conv3x3(w).float())
    out = torch.full((n, oh * ow, cout), float("nan"), device=dev)
    ops.conv3x3(_f8(x.permute(0, 2, 3, 1).contiguous()), w8, w_exp=(e + 127).to(U8), stride=stride, bias=bias,
                row_add=temb, rows_per_group=oh * ow, residual=res, out_f32=out)
    got = out.view(n, oh, ow, cout).permute(0, 3, 1, 2)
    assert torch.equal(got, ref), f"max diff {(got - ref).abs().max()}"


@pytest.mark.parametrize("n,ih,iw,cin,cout,stats", [(3, 8, 8, 640, 640, False), (42, 9, 9, 128, 1280, False), (2, 36, 36, 128, 640, False),
                                                    (2, 72, 72, 128, 128, True), (1, 33, 31, 128, 256, False), (2, 16, 16, 256, 384, True)])
def test_conv3x3_fp8_window_kernel(dev, n, ih, iw, cin, cout, stats, knobs):
    """The window-staged conv on e4m3 operands (csrc/conv_win.hip, FP8 instantiations of its 128-column family: what the C >= 640 convs of
    a step run in fp8 mode): a window / weight row is 128 e4m3 channels, the two fragment halves of a (tap, slab) form one operand of the
    block-scaled MFMA.  Integer data with power-of-two weight scales: bit-exact against torch for the default dispatch, both families
    and the per-tap kernel; GroupNorm statistics add up per image."""
    from seva import ops
    from seva._engine import pack_conv3x3
    x = _ints((n, cin, ih, iw), -3, 3, dev, 31)
    w = _ints((cout, cin, 3, 3), -2, 2, dev, 32)
    g = torch.Generator().manual_seed(33)
    e = torch.randint(-2, 3, (cout,), generator=g).to(dev)
    bias, res = _ints((cout,), -4, 4, dev, 34), _ints((n, ih * iw, cout), -5, 5, dev, 35)
    ref = F.conv2d(x, w * torch.exp2(e.float())[:, None, None, None], bias, padding=1).permute(0, 2, 3, 1).reshape(n, ih * iw, cout) + res
    x8, w8 = _f8(x.permute(0, 2, 3, 1).contiguous()), _f8(pack_conv3x3(w).float())
    for fam in (-1, 1, 2, 0):
        knobs(conv_win=fam)
        out = torch.full((n, ih * iw, cout), float("nan"), device=dev)
        st = torch.full(ops.channel_stats_shape(n * ih * iw, cout), float("nan"), device=dev) if stats else None
        ops.conv3x3(x8, w8, w_exp=(e + 127).to(U8), bias=bias, residual=res, out_f32=out, ch_stats=st)
        assert torch.equal(out, ref), f"conv_win {fam}: max diff {(out - ref).abs().max()}"
        if st is not None:
            assert torch.equal(st[:, 0].view(n, ih * iw // 64, cout).sum(1), ref.sum(1))


def _close_fp8(got_u8, ref_f32):
    """e4m3 bytes vs an fp32 reference: equal to torch's cast except where fp32 arithmetic differences cross a rounding tie."""
    got = got_u8.view(torch.float8_e4m3fn).float()
    want = _f8(ref_f32).view(torch.float8_e4m3fn).float()
    same = (got == want).float().mean().item()
    ulp = torch.maximum(ref_f32.abs() * 2.0 ** -3, torch.tensor(2.0 ** -9, device=ref_f32.device))
    assert same > 0.995 and bool(((got - ref_f32).abs() <= ulp).all()), same
    return same


@pytest.mark.parametrize("rows,c", [(1000, 640), (70000, 320), (333, 1280)])
def test_layernorm_fp8(dev, rows, c):
    from seva import ops
    g = torch.Generator().manual_seed(5)
    x = (torch.randn(rows, c, generator=g) * 3 + 0.5).to(dev)
    gm, bt = (1 + 0.1 * torch.randn(c, generator=g)).to(dev), (0.1 * torch.randn(c, generator=g)).to(dev)
    out = torch.zeros((rows, c), device=dev, dtype=U8)
    ops.layernorm(x, gm, bt, out)
    ref = F.layer_norm(x, (c,), gm, bt, 1e-5)
    print(f"layernorm fp8 [{rows},{c}]: {_close_fp8(out, ref) * 100:.2f} % identical to torch's e4m3 cast")


def test_groupnorm_fp8_output(dev):
    from seva import ops
    n, hw, c = 3, 100, 256
    g = torch.Generator().manual_seed(6)
    x = (torch.randn(n, hw, c, generator=g) * 2 + 1).to(dev)
    gm, bt = (1 + 0.1 * torch.randn(c, generator=g)).to(dev), (0.1 * torch.randn(c, generator=g)).to(dev)
    ws = ops.groupnorm_workspace(n, dev)
    o16 = torch.zeros((n, hw, c), device=dev, dtype=torch.float16)
    o8 = torch.zeros((n, hw, c), device=dev, dtype=U8)
    ops.groupnorm(x, None, gm, bt, o16, ws, silu=True, out_f8=o8)
    ref = F.silu(F.group_norm(x.permute(0, 2, 1), 32, gm, bt, 1e-5)).permute(0, 2, 1)
    _close_fp8(o8, ref)
    o8b = torch.zeros_like(o8)
    ops.groupnorm(x, None, gm, bt, None, ws, silu=True, out_f8=o8b)  # e4m3 only
    assert torch.equal(o8, o8b) and rel_l2(o16, ref) < 1e-3


def test_fp8_network_accuracy_is_reported(dev):
    """BASELINE config 5 at network level: the 1.3B forward of config 1 in fp8 mode against the REFERENCE golden.  e4m3
    carries 3 mantissa bits, so this mode is far outside the 1e-3 parity tolerance by construction; the number is
    REPORTED (and bounded loosely so that a wiring bug -- O(1) error -- still fails).  f16 stays the parity mode."""
    from conftest import load_golden
    from test_model_gpu import _build
    from seva.model import SGMWrapper
    net, _ = _build("full", dev)
    g = load_golden("g4_full_forward")
    T = int(g["T"])
    c = {k: g[k].to(dev) for k in ("crossattn", "concat", "dense_vector")}
    y16 = SGMWrapper(net)(g["x"].to(dev), g["t"].to(dev), c, num_frames=T).cpu()
    net.set_precision("fp8")
    y8 = SGMWrapper(net)(g["x"].to(dev), g["t"].to(dev), c, num_frames=T).cpu()
    e16, e8 = rel_l2(y16, g["y"]), rel_l2(y8, g["y"])
    nq = sum(1 for k in net.engine().W if k.endswith("8e"))
    print(f"\n1.3B forward (config 1) vs reference: f16 mode rel-L2 {e16:.3e}; fp8 mode rel-L2 {e8:.3e} ({nq} e4m3 weight tensors)")
    assert e16 < 1e-3 and 1e-3 < e8 < 0.2 and nq > 100


def test_fp8_forward_at_the_headline_shape_vs_reference(dev):
    """BASELINE config 5 at the shape the metric is quoted on: ONE 1.3B network call at T=21, 576x576 (latent 72x72, CFG batch
    42) in fp8 mode against the REFERENCE's own output (tests/golden/g9_T21_forward.npz).  Reported: overall and worst-latent
    rel-L2.  Bounded: fp8 is a separate accuracy class (3 mantissa bits per operand), asserted inside [1e-3, 6e-2] overall and
    < 1e-1 per latent -- a wiring error (O(1)) fails, and so does a silent fall-back to f16 (< 1e-3)."""
    import os
    from conftest import GOLD, load_golden
    from test_headline_gpu import FORWARD_SEEDS, HW, _wrapper_inputs
    from test_model_gpu import _build
    from seva.model import SGMWrapper
    if not os.path.exists(os.path.join(GOLD, "g9_T21_forward.npz")):
        pytest.skip("g9_T21_forward.npz not generated")
    g = load_golden("g9_T21_forward")
    T = 21
    net, _ = _build("full", dev)
    net.set_precision("fp8")
    x, t, c = _wrapper_inputs(T, FORWARD_SEEDS[T])
    y = SGMWrapper(net)(x.to(dev), t.to(dev), {k: v.to(dev) for k, v in c.items()}, num_frames=T).cpu()
    ref = g["y"]
    err = rel_l2(y, ref)
    per = [rel_l2(y[i], ref[i]) for i in range(y.shape[0])]
    nq = sum(1 for k in net.engine().W if k.endswith("8e"))
    print(f"\nfp8 mode, 1.3B forward T=21 72x72 (B=42) vs REFERENCE: rel-L2 {err:.3e}; per latent max {max(per):.3e} min {min(per):.3e} "
          f"({nq} e4m3 weight tensors; f16 parity mode: 6.5e-4)")
    assert torch.isfinite(y).all() and 1e-3 < err < 6e-2 and max(per) < 1e-1 and nq > 100


def test_fp8_activation_scale_is_not_the_error_lever(dev):
    """Would a per-row (token) power-of-two activation scale -- carried as the activation operand's E8M0 block scale -- lower the
    fp8 error?  e4m3 is a FLOATING-point format: a power-of-two scale moves the exponent window (normals 2^-6 .. 448), not the
    3-bit mantissa, so it only matters for values outside that window.  Measured here on the tensors the fp8 GEMMs consume
    (LayerNorm output, GroupNorm+SiLU output, GEGLU hidden activations of N(0,1)-scale inputs): the quantisation error with
    the ideal per-row scale is within a few per cent of the unit-scale error, and both sit at the mantissa floor
    (2^-4 / sqrt(3) ~ 3.6e-2 rms per element).  The unit activation scale of seva_gemm_fp8 stays."""
    g = torch.Generator().manual_seed(8)
    x = torch.randn(4096, 1280, generator=g).to(dev)
    tensors = {
        "layernorm": F.layer_norm(x * 3 + 0.5, (1280,)) * (1 + 0.1 * torch.randn(1280, generator=g).to(dev)),
        "groupnorm+silu": F.silu(F.group_norm((x * 2 + 1).view(4, 1024, 1280).permute(0, 2, 1), 32)).permute(0, 2, 1).reshape(4096, 1280),
        "geglu hidden": (x[:, :640] * 1.5) * F.gelu(x[:, 640:] * 1.5),
    }

    def q(v):  # saturating e4m3 round trip
        return v.clamp(-448, 448).to(torch.float8_e4m3fn).float()

    for name, a in tensors.items():
        unit = rel_l2(q(a), a)
        e = torch.ceil(torch.log2(a.abs().amax(1, keepdim=True).clamp_min(1e-30) / 448.0))
        scaled = rel_l2(q(a * torch.exp2(-e)) * torch.exp2(e), a)
        print(f"\ne4m3 quantisation error of {name}: unit scale {unit:.3e}, ideal per-row power-of-two scale {scaled:.3e}")
        assert 1.5e-2 < scaled <= unit * 1.02 and unit < scaled * 1.25


def test_fp8_reduction_padding_320_to_384(dev):
    """C = 320 level in fp8 mode: the reduction length is zero-padded to 384 (one more 128-deep MFMA K-tile instead of 2.5).
    LayerNorm writes its 320 columns into a zero-initialised [rows, 384] e4m3 buffer; GroupNorm writes 320 channels at pixel
    pitch 384; weights carry 64 zero columns / channels per tap.  Results equal the unpadded computation exactly."""
    from seva import ops
    from seva._engine import pack_conv3x3
    rows, c, kp, N = 777, 320, 384, 960
    g = torch.Generator().manual_seed(21)
    x = (torch.randn(rows, c, generator=g) * 2).to(dev)
    gm, bt = (1 + 0.1 * torch.randn(c, generator=g)).to(dev), (0.1 * torch.randn(c, generator=g)).to(dev)
    a8 = torch.zeros((rows, kp), device=dev, dtype=U8)
    ops.layernorm(x, gm, bt, a8)
    assert int(a8[:, c:].max()) == 0  # pad columns untouched
    _close_fp8(a8[:, :c].contiguous(), F.layer_norm(x, (c,), gm, bt, 1e-5))
    w = _ints((N, c), -3, 3, dev, 22)
    wp = torch.cat([w, torch.zeros((N, kp - c), device=dev)], 1)
    w8, wexp, _ = _wq(wp, dev, 23)
    o32 = torch.full((rows, N), float("nan"), device=dev)
    ops.gemm(a8, w8, w_exp=wexp, out_f32=o32)
    e = wexp.float() - 127.0
    ref = a8[:, :c].contiguous().view(torch.float8_e4m3fn).float() @ (w * torch.exp2(e)[:, None]).T
    assert torch.allclose(o32, ref, rtol=1e-6, atol=1e-4)  # (activations are not integers here: fp32 summation order)
    # conv: 320 channels in a 384-channel e4m3 image
    n, ih, iw, cout = 2, 9, 7, 64
    xi = _ints((n, c, ih, iw), -3, 3, dev, 24)
    wc = _ints((cout, c, 3, 3), -2, 2, dev, 25)
    img = torch.zeros((n, ih, iw, kp), device=dev, dtype=U8)
    img[..., :c] = _f8(xi.permute(0, 2, 3, 1).contiguous())
    wc8 = _f8(pack_conv3x3(wc, kp).float())
    out = torch.full((n, ih * iw, cout), float("nan"), device=dev)
    ops.conv3x3(img, wc8, w_exp=torch.full((cout,), 127, device=dev, dtype=U8), out_f32=out)
    assert torch.equal(out.view(n, ih, iw, cout).permute(0, 3, 1, 2), F.conv2d(xi, wc, None, padding=1))
    # GroupNorm with pixel pitch 384
    xg = (torch.randn(n, ih * iw, c, generator=g) + 0.5).to(dev)
    ws = ops.groupnorm_workspace(n, dev)
    o8 = torch.zeros((n, ih * iw, kp), device=dev, dtype=U8)
    ops.groupnorm(xg, None, gm, bt, None, ws, silu=True, out_f8=o8)
    assert int(o8[..., c:].max()) == 0
    _close_fp8(o8[..., :c].contiguous(), F.silu(F.group_norm(xg.permute(0, 2, 1), 32, gm, bt, 1e-5)).permute(0, 2, 1))
