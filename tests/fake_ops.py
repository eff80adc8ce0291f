"""TEST-ONLY torch emulation of the C-ABI operators (same layouts, strides and epilogue contracts
as include/seva_hip.h).  Lets the CPU suite check the *host logic* of `seva._engine` (buffer
wiring, strides, weight packing, residual plumbing) without a GPU.  Never imported by the product.
"""
import math

import torch
import torch.nn.functional as F

F16, F32, U8 = torch.float16, torch.float32, torch.uint8


# host-side e4m3 helpers of the real module (pure torch, no kernels)
def to_fp8(x):
    return x.float().clamp(-448.0, 448.0).to(torch.float8_e4m3fn).view(torch.uint8)


def _from_fp8(x8):
    return x8.view(torch.float8_e4m3fn).float()


def quantize_weight_fp8(w):
    w = w.float()
    amax = w.abs().amax(dim=1).clamp_min(1e-30)
    e = torch.ceil(torch.log2(amax / 448.0)).clamp(-126, 127)
    q = (w * torch.exp2(-e)[:, None]).clamp(-448.0, 448.0).to(torch.float8_e4m3fn)
    return q.view(torch.uint8).contiguous(), (e + 127).to(torch.uint8).contiguous()


def dequantize_weight_fp8(w8, w_exp):
    return _from_fp8(w8) * torch.exp2(w_exp.float() - 127.0)[:, None]


STATS_ROWS = 64


def splitk_workspace(max_rows, max_channels, device):
    return torch.zeros(16, dtype=F32)


def channel_stats_shape(rows, channels):
    return ((rows + STATS_ROWS - 1) // STATS_ROWS, 2, channels)


def _emit_stats(acc, ch_stats):
    """seva_gemm_desc.ch_stats: per 64-row block and channel, sum and sum of squares of the fp32 output."""
    M, N = acc.shape
    nb = (M + STATS_ROWS - 1) // STATS_ROWS
    pad = torch.zeros((nb * STATS_ROWS, N), dtype=F32)
    pad[:M] = acc
    blk = pad.view(nb, STATS_ROWS, N)
    st = ch_stats.view(-1)[: nb * 2 * N].view(nb, 2, N)
    st[:, 0] = blk.sum(1)
    st[:, 1] = (blk * blk).sum(1)


def _epilogue(acc, M, N, bias, row_add, rows_per_group, ld_row_add, residual, out_f32, out_f16, geglu, out_f8=None,
              ch_stats=None):
    if geglu:
        if bias is not None:
            acc = acc + bias
        g = acc.view(M, N // 64, 2, 32)
        acc = (g[:, :, 0] * F.gelu(g[:, :, 1])).reshape(M, N // 2)
    else:
        if bias is not None:
            acc = acc + bias
        if row_add is not None:
            ld = ld_row_add or N
            ng = (M + rows_per_group - 1) // rows_per_group
            ra = torch.as_strided(row_add, (ng, N), (ld, 1), row_add.storage_offset())
            acc = acc + ra.repeat_interleave(rows_per_group, 0)[:M]
        if residual is not None:
            acc = acc + residual.reshape(M, -1)[:, :N]
    if out_f32 is not None:
        out_f32.view(M, -1)[:, : acc.shape[1]].copy_(acc)
    if out_f16 is not None:
        out_f16.view(M, -1)[:, : acc.shape[1]].copy_(acc.half())
    if out_f8 is not None:
        out_f8.view(M, -1)[:, : acc.shape[1]].copy_(to_fp8(acc))
    if ch_stats is not None:
        assert out_f32 is not None and not geglu and N >= 128
        _emit_stats(acc, ch_stats)


def gemm(a, w, *, bias=None, row_add=None, rows_per_group=0, ld_row_add=0, residual=None,
         out_f32=None, out_f16=None, geglu=False, col_scale=1.0, col_scale_n=0, w_exp=None, out_f8=None, ch_stats=None,
         splitk_ws=None, alg_k=0):
    M, N = a.shape[0], w.shape[0]
    if w_exp is not None:  # seva_gemm_fp8
        assert a.dtype == U8 and w.dtype == U8 and a.shape[1] % 128 == 0 and N % 16 == 0
        acc = _from_fp8(a) @ dequantize_weight_fp8(w, w_exp).T
    else:
        assert a.dtype == F16 and w.dtype == F16 and a.shape[1] % 64 == 0 and out_f8 is None
        acc = a.float() @ w.float().T
    if col_scale_n:
        assert residual is None and row_add is None and not geglu
        if bias is not None:
            acc = acc + bias
            bias = None
        acc[:, :col_scale_n] *= col_scale
    _epilogue(acc, M, N, bias, row_add, rows_per_group, ld_row_add, residual, out_f32, out_f16, geglu, out_f8, ch_stats)


FF_FUSED_CHANNELS = (64, 128, 256, 320)


def ff_fused(a, w1, b1, w2, b2, *, residual=None, out_f32=None, out_f16=None, ln_x=None, ln_gamma=None, ln_beta=None,
             ln_eps=1e-5):
    if ln_x is not None:
        c = ln_x.shape[-1]
        a = F.layer_norm(ln_x.reshape(-1, c), (c,), ln_gamma, ln_beta, ln_eps).half()
    M, c = a.shape
    assert a.dtype == F16 and w1.shape == (8 * c, c) and w2.shape == (c, 4 * c) and c in FF_FUSED_CHANNELS
    y = (a.float() @ w1.float().T + b1).view(M, -1, 2, 32)
    h = (y[:, :, 0] * F.gelu(y[:, :, 1])).reshape(M, 4 * c).half().float()  # hidden rounded to f16 once
    acc = h @ w2.float().T + b2
    if residual is not None:
        acc = acc + residual.reshape(M, -1)[:, :c]
    if out_f32 is not None:
        out_f32.view(M, -1)[:, :c].copy_(acc)
    if out_f16 is not None:
        out_f16.view(M, -1)[:, :c].copy_(acc.half())


def conv3x3(x, w, *, stride=1, upsample=False, bias=None, row_add=None, rows_per_group=0,
            ld_row_add=0, residual=None, out_f32=None, out_f16=None, pad_br_only=False, w_exp=None, ch_stats=None,
            splitk_ws=None, a2=None, alg_k=0):
    n, ih, iw, cin = x.shape
    w2 = None
    if a2 is not None:  # folded second operand (seva_gemm_desc.a2): the columns behind the nine taps multiply a2
        assert w_exp is None and not upsample and a2.dtype == F16 and a2.shape[1] % 64 == 0 and w.shape[1] == 9 * cin + a2.shape[1]
        w, w2 = w[:, : 9 * cin], w[:, 9 * cin:]
    if w_exp is not None:  # seva_gemm_fp8, conv mode
        assert x.dtype == U8 and w.dtype == U8 and cin % 128 == 0 and w.shape[1] == 9 * cin and not upsample
        x, w = _from_fp8(x), dequantize_weight_fp8(w, w_exp)
    else:
        assert x.dtype == F16 and cin % 64 == 0 and w.shape[1] == 9 * cin
    xi = x.float().permute(0, 3, 1, 2)
    if upsample:
        xi = F.interpolate(xi, scale_factor=2, mode="nearest")
    wk = w.float().view(-1, 3, 3, cin).permute(0, 3, 1, 2)
    if pad_br_only:
        y = F.conv2d(F.pad(xi, (0, 1, 0, 1)), wk, None, stride=stride, padding=0)
    else:
        y = F.conv2d(xi, wk, None, stride=stride, padding=1)
    N = w.shape[0]
    acc = y.permute(0, 2, 3, 1).reshape(-1, N)
    if w2 is not None:
        acc = acc + a2.float() @ w2.float().T
    _epilogue(acc, acc.shape[0], N, bias, row_add, rows_per_group, ld_row_add, residual, out_f32,
              out_f16, False, ch_stats=ch_stats)


ATTN_SPLIT_MIN_LK = 6144


def attention_split_workspace_numel(batch, heads, lq, nsplit=2):
    return nsplit * batch * heads * lq * 66


def attention(q, k, v, out, *, nb0, nb1, heads, lq, lk, q_strides, k_strides, o_strides, scale=0.125,
              q_prescaled=False, split_ws=None):
    if q_prescaled:  # q carries scale * log2(e): softmax base 2
        scale = math.log(2.0)
    def view(t, st, L):
        return torch.as_strided(t, (nb0, nb1, L, heads, 64), (st[0], st[1], st[2], 64, 1), t.storage_offset())
    qv, kv, vv = view(q, q_strides, lq).float(), view(k, k_strides, lk).float(), view(v, k_strides, lk).float()
    att = torch.einsum("abqhd,abkhd->abhqk", qv, kv) * scale
    o = torch.einsum("abhqk,abkhd->abqhd", torch.softmax(att, -1), vv)
    view(out, o_strides, lq).copy_(o.half())


GN_WORKSPACE_SLABS = 1024  # include/seva_hip.h SEVA_GN_WORKSPACE_SLABS


def groupnorm_workspace(n, device):
    return torch.empty(n * GN_WORKSPACE_SLABS * 32 * 2, dtype=F32, device=device)


def groupnorm(x1, x2, gamma, beta, out_f16, workspace, *, groups=32, eps=1e-5, silu=False,
              dense=None, dense_w=None, dense_b=None, raw_f16=None, out_f8=None, stats1=None, stats2=None,
              split_out=False, split_raw=False):
    def hilo(v):  # [hi | lo] channels of the split-precision outputs (seva_groupnorm_desc.split_*)
        hi = v.half()
        return torch.cat([hi, (v - hi.float()).half()], -1)

    x = torch.cat([x1, x2], -1) if x2 is not None else x1
    if raw_f16 is not None:
        if split_raw:
            raw_f16.view(x.shape[:-1] + (2 * x.shape[-1],)).copy_(hilo(x))
        else:
            raw_f16.view(x.shape).copy_(x.half())
    C = x.shape[-1]
    if stats1 is not None:
        # statistics come from the producers' epilogues (seva_groupnorm_desc.stats1 / stats2): USE them, so that a wrong or
        # stale buffer handed over by the engine shows up as a wrong result
        n, hw = x.shape[0], x.shape[1]
        assert hw % STATS_ROWS == 0 and (x2 is None) == (stats2 is None)
        nb = hw // STATS_ROWS
        parts = [stats1.view(-1)[: n * nb * 2 * x1.shape[-1]].view(n, nb, 2, x1.shape[-1])]
        if x2 is not None:
            parts.append(stats2.view(-1)[: n * nb * 2 * x2.shape[-1]].view(n, nb, 2, x2.shape[-1]))
        st = torch.cat(parts, -1).double().sum(1)  # [n, 2, C]
        cnt = hw * (C // groups)
        mean = st[:, 0].view(n, groups, -1).sum(-1) / cnt
        var = (st[:, 1].view(n, groups, -1).sum(-1) / cnt - mean * mean).clamp_min(0.0)
        rstd = 1.0 / torch.sqrt(var + eps)
        mean_c = mean.float().repeat_interleave(C // groups, 1)[:, None, :]
        rstd_c = rstd.float().repeat_interleave(C // groups, 1)[:, None, :]
        y = (x - mean_c) * rstd_c * gamma + beta
    else:
        y = F.group_norm(x.transpose(1, 2), groups, gamma, beta, eps).transpose(1, 2)
    if silu:
        y = F.silu(y)
    if dense is not None:
        d = dense @ dense_w.T + dense_b
        y = y * (1 + d[..., :C]) + d[..., C:]
    if out_f16 is not None:
        out_f16.copy_(hilo(y) if split_out else y.half())
    if out_f8 is not None:
        out_f8[..., : y.shape[-1]].copy_(to_fp8(y))


def layernorm(x, gamma, beta, out_f16, eps=1e-5):
    c = x.shape[-1]
    y = F.layer_norm(x.reshape(-1, c), (c,), gamma, beta, eps)
    if out_f16.dtype == U8:
        out_f16[:, :c].copy_(to_fp8(y))  # pad columns (K padded to a multiple of 128) untouched
    else:
        out_f16.view(-1, c).copy_(y if out_f16.dtype == F32 else y.half())


def clip_preprocess(x, patches_f16, mean, std, *, out_size=224, patch=14, antialias=True):
    """emulates seva_clip_preprocess_f16 with torch ops (same published algorithm as oracle/clip_ref.py:preprocess)"""
    n, c, H, W = x.shape
    fy, fx = H / out_size, W / out_size
    if antialias and max(fy, fx) > 1.0:
        sy, sx = max((fy - 1.0) / 2.0, 0.001), max((fx - 1.0) / 2.0, 0.001)
        ky, kx = int(max(4.0 * sy, 3)), int(max(4.0 * sx, 3))
        ky, kx = ky + (ky % 2 == 0), kx + (kx % 2 == 0)
        ty, tx = torch.arange(ky, dtype=F32) - ky // 2, torch.arange(kx, dtype=F32) - kx // 2
        gy, gx = torch.exp(-ty * ty / (2 * sy * sy)), torch.exp(-tx * tx / (2 * sx * sx))
        k2 = ((gy / gy.sum())[:, None] * (gx / gx.sum())[None, :])[None, None].repeat(c, 1, 1, 1)
        x = F.conv2d(F.pad(x, (kx // 2, kx // 2, ky // 2, ky // 2), mode="reflect"), k2, groups=c)
    x = F.interpolate(x, size=(out_size, out_size), mode="bicubic", align_corners=True)
    x = ((x + 1.0) / 2.0 - torch.tensor(mean)[None, :, None, None]) / torch.tensor(std)[None, :, None, None]
    g = out_size // patch
    pm = x.view(n, c, g, patch, g, patch).permute(0, 2, 4, 1, 3, 5).reshape(n * g * g, c * patch * patch)
    patches_f16[:, : c * patch * patch].copy_(pm.half())


def attention_small(q, k, v, out, *, batch, heads, L, head_dim, q_strides, k_strides, o_strides, scale):
    def view(t, st):
        return torch.as_strided(t, (batch, L, heads, head_dim), (st[0], st[1], head_dim, 1), t.storage_offset())
    qv, kv, vv = view(q, q_strides).float(), view(k, k_strides).float(), view(v, k_strides).float()
    att = torch.softmax(torch.einsum("bqhd,bkhd->bhqk", qv, kv) * scale, -1)
    view(out, o_strides).copy_(torch.einsum("bhqk,bkhd->bqhd", att, vv).half())


def softmax_rows(x, out_f16, cols, scale):
    out_f16.zero_()
    out_f16[:, :cols] = torch.softmax(x[:, :cols] * scale, -1).half()


def nchw_to_nhwc_f16(x1, x2, out_f16, scale=None, split=False):
    n = x1.shape[0]
    a = x1 if scale is None else x1 * scale.view(-1, 1, 1, 1)
    x = torch.cat([a, x2], 1) if x2 is not None else a
    c = x.shape[1]
    o = out_f16.view(n, -1, out_f16.shape[-1])
    o.zero_()
    v = x.reshape(n, c, -1).transpose(1, 2)
    o[..., :c] = v.half()
    if split:
        o[..., c:2 * c] = (v - v.half().float()).half()


def nhwc_to_nchw_f32(x, out):
    n, c = out.shape[:2]
    out.copy_(x[..., :c].reshape(n, out.shape[2], out.shape[3], c).permute(0, 3, 1, 2))


def cast_concat_f16(x1, x2, out_f16):
    c1 = x1.shape[-1]
    a = x1.reshape(-1, c1)
    x = torch.cat([a, x2.reshape(a.shape[0], -1)], 1) if x2 is not None else a
    out_f16.view(a.shape[0], -1).copy_(x.half())


def bilinear_to_nhwc(src, out, oh, ow):
    y = F.interpolate(src, size=(oh, ow), mode="bilinear", align_corners=True)
    out.copy_(y.permute(0, 2, 3, 1).reshape(out.shape))


def timestep_embedding_f16(t, freqs, out_f16):
    args = t[:, None].float() * freqs[None]
    out_f16.copy_(torch.cat([torch.cos(args), torch.sin(args)], -1).half())


def silu_f16(x, out_f16):
    out_f16.copy_(F.silu(x).half())


def add_f32(a, b, out):
    out.copy_(a + b)


def _rows(v, x):
    return v.view(-1, *([1] * (x.ndim - 1)))


def replace_blend(x, replace, out):
    c = x.shape[1]
    m = replace[:, c:]
    out.copy_(x * (1 - m) + replace[:, :c] * m)


def denoiser_combine(net, x, c_out, c_skip, out):
    out.copy_(net * _rows(c_out, x) + x * _rows(c_skip, x))


def add_noise(x, eps, noise_scale, out):
    out.copy_(x + eps * _rows(noise_scale, x))


def cfg_combine(den2, scale, out):
    u, c = den2.chunk(2)
    out.copy_(u + _rows(scale, u) * (c - u))


def euler_step(x, den, sigma_hat, dt, out):
    out.copy_(x + _rows(dt, x) * ((x - den) / _rows(sigma_hat, x)))


def cfg_euler(x, den2, scale, sigma_hat, dt, out):
    u, c = den2.chunk(2)
    den = u + _rows(scale, u) * (c - u)
    out.copy_(x + _rows(dt, x) * ((x - den) / _rows(sigma_hat, x)))


def to_d(x, den, sigma, out):
    out.copy_((x - den) / _rows(sigma, x))


def scale_rows(x, s, out):
    out.copy_(x * _rows(s, x))


def plucker(kinv, pose_inv, out):
    V, _, h, w = out.shape
    ys = torch.arange(h, dtype=torch.float32) + 0.5
    xs = torch.arange(w, dtype=torch.float32) + 0.5
    Y, X = torch.meshgrid(ys, xs, indexing="ij")
    grid = torch.stack([X, Y, torch.ones_like(X)], -1).view(-1, 3)
    cam = grid[None] @ kinv.transpose(-1, -2)
    world = cam @ pose_inv[:, :, :3].transpose(-1, -2) + pose_inv[:, None, :, 3]
    ctr = pose_inv[:, None, :, 3].expand_as(world)
    ray = F.normalize(world - ctr, dim=-1)
    out.copy_(torch.cat([ray, torch.linalg.cross(ctr, ray, dim=-1)], -1).permute(0, 2, 1).reshape(V, 6, h, w))


def cond_concat(plucker_maps, mask_u8, c_concat, uc_concat):
    V, _, h, w = plucker_maps.shape
    c_concat[:, 1:] = plucker_maps
    uc_concat[:, 1:] = plucker_maps
    c_concat[:, 0] = mask_u8.float()[:, None, None]
    uc_concat[:, 0] = 0
