"""Tensor-level wrappers over the C-ABI operators of libseva_hip.so.

Every function enqueues HIP kernels on PyTorch's current stream through raw device pointers;
PyTorch is used for memory and streams only.  Outputs are caller-provided (graph-capture
friendly: no allocation happens here unless `out=None`).  Layouts follow include/seva_hip.h:
channels-last activations, f16 GEMM operands, f32 residual stream.
"""

from __future__ import annotations

import ctypes as C

import torch

from . import _native as nv
from ._native import AttnDesc, GemmDesc, GroupNormDesc, check, ptr, require_cuda, stream_ptr

F16, F32 = torch.float16, torch.float32
U8 = torch.uint8  # e4m3 bytes travel as uint8 tensors (bit pattern of torch.float8_e4m3fn)


def _lib():
    return nv.load()


def gemm(
    a: torch.Tensor,
    w: torch.Tensor,
    *,
    bias: torch.Tensor | None = None,
    row_add: torch.Tensor | None = None,
    rows_per_group: int = 0,
    ld_row_add: int = 0,
    residual: torch.Tensor | None = None,
    out_f32: torch.Tensor | None = None,
    out_f16: torch.Tensor | None = None,
    geglu: bool = False,
    col_scale: float = 1.0,
    col_scale_n: int = 0,
    w_exp: torch.Tensor | None = None,
    out_f8: torch.Tensor | None = None,
    ch_stats: torch.Tensor | None = None,
    splitk_ws: torch.Tensor | None = None,
    alg_k: int = 0,
) -> None:
    """out = a @ w.T (+bias +row_add[group] +residual); a:[M,K] f16, w:[N,K] f16 (seva_gemm_f16).
    alg_k: accounting only -- the reference-equivalent reduction length when K is padded / split-precision (seva_gemm_desc.alg_K).
    splitk_ws (`splitk_workspace`): only convolutions use it (seva_gemm_desc.splitk_ws); accepted and ignored for plain GEMMs.
    ch_stats (`channel_stats_buffer`): receives per-64-row-block, per-channel sum / sum of squares of out_f32 (GroupNorm
    statistics emitted by the epilogue; `groupnorm(stats1=...)`).
    Output features < col_scale_n are multiplied by col_scale in fp32 (plain epilogue only).
    fp8 mode (seva_gemm_fp8): a, w are uint8 tensors of e4m3 bytes and w_exp [N] uint8 the weights' E8M0 scale bytes;
    out_f8 (GEGLU epilogue only) receives the hidden activations as e4m3."""
    fp8 = w_exp is not None
    require_cuda(a, w)
    assert a.dim() == 2 and w.dim() == 2 and a.dtype == w.dtype == (U8 if fp8 else F16)
    M, K = a.shape
    N = w.shape[0]
    d = GemmDesc()
    d.a, d.w = ptr(a), w.data_ptr()
    d.bias, d.row_add, d.residual = ptr(bias), ptr(row_add), ptr(residual)
    d.out_f32, d.out_f16 = ptr(out_f32), ptr(out_f16)
    d.M, d.N, d.K = M, N, K
    d.alg_K = alg_k
    d.lda = a.stride(0)
    d.ldr = residual.stride(0) if residual is not None else 0
    d.ldo32 = out_f32.stride(0) if out_f32 is not None else 0
    d.ldo16 = out_f16.stride(0) if out_f16 is not None else 0
    d.rows_per_group, d.ld_row_add = rows_per_group, ld_row_add
    d.mode, d.epilogue = 0, 1 if geglu else 0
    d.col_scale, d.col_scale_n = col_scale, col_scale_n
    d.ch_stats = _stats_ptr(ch_stats, M, N, out_f32)
    if splitk_ws is not None and not fp8:
        assert splitk_ws.dtype == F32 and splitk_ws.is_contiguous()
        d.splitk_ws, d.splitk_ws_bytes = splitk_ws.data_ptr(), splitk_ws.numel() * 4
        _register_handoff_ws(splitk_ws)
    if fp8:
        assert w_exp.dtype == U8 and w_exp.numel() == N and (out_f8 is None or out_f8.dtype == U8)
        d.w_exp, d.out_f8 = w_exp.data_ptr(), ptr(out_f8)
        d.ldo8 = out_f8.stride(0) if out_f8 is not None else 0
        check(_lib().seva_gemm_fp8(C.byref(d), stream_ptr(a.device)), "seva_gemm_fp8")
    else:
        assert out_f8 is None
        check(_lib().seva_gemm_f16(C.byref(d), stream_ptr(w.device)), "seva_gemm_f16")



STATS_ROWS = 64  # rows per block of the epilogue-emitted GroupNorm statistics (include/seva_hip.h: seva_gemm_desc.ch_stats)


def channel_stats_shape(rows: int, channels: int) -> tuple[int, int, int]:
    """Shape of the f32 buffer a GEMM / conv fills through `ch_stats`: [row blocks][sum | sum of squares][channel]."""
    return ((rows + STATS_ROWS - 1) // STATS_ROWS, 2, channels)


# Split-K hand-off workspaces that launches have used (data_ptr -> weak reference).  The consumer workgroup of a split tile
# waits for its producer with a BOUNDED spin (csrc/gemm.hip); a give-up leaves a wrong tile and counts itself in int slot
# SPLITK_ERROR_SLOT of the workspace.  `check_handoffs` makes that loud: it is called where a host sync exists anyway (end of a
# sampler trajectory, `prof_collect`, the GPU tests), never inside the step loop.
SPLITK_FLAGS = 16384
SPLITK_ERROR_SLOT = 16383
_handoff_ws: dict = {}


def _register_handoff_ws(ws: torch.Tensor) -> None:
    if ws.data_ptr() not in _handoff_ws:
        import weakref

        _handoff_ws[ws.data_ptr()] = weakref.ref(ws)


def check_handoffs() -> None:
    """Raise SevaNativeError if any split-K consumer gave up waiting for its producer since the last check (the affected
    outputs are wrong).  On error the flag area is zeroed so that a stale flag cannot corrupt the next launch.  Host-syncs."""
    bad = []
    for key, ref in list(_handoff_ws.items()):
        ws = ref()
        if ws is None or ws.data_ptr() != key:
            _handoff_ws.pop(key, None)
            continue
        if torch.cuda.is_current_stream_capturing():
            return
        n = int(ws[SPLITK_ERROR_SLOT:SPLITK_ERROR_SLOT + 1].view(torch.int32).item())
        if n:
            ws[:SPLITK_FLAGS].zero_()
            bad.append(n)
    if bad:
        raise nv.SevaNativeError(
            f"split-K hand-off: {sum(bad)} consumer workgroup(s) gave up waiting for their producer; the outputs of the "
            "affected launches are wrong (flags re-armed; rerun, or set SEVA_CONV_SPLITK=0)")


def splitk_workspace(max_rows: int, max_channels: int, device) -> torch.Tensor:
    """Zeroed workspace for `conv3x3(splitk_ws=...)`: 16384 flags + one fp32 128 x 160 tile per output tile."""
    tiles = max(512, ((max_rows + 127) // 128) * ((max_channels + 127) // 128))
    return torch.zeros(16384 + tiles * 128 * 160, dtype=F32, device=device)


def _stats_ptr(ch_stats, M, N, out_f32):
    if ch_stats is None:
        return None
    assert out_f32 is not None and ch_stats.dtype == F32 and ch_stats.is_contiguous()
    assert ch_stats.numel() >= ((M + STATS_ROWS - 1) // STATS_ROWS) * 2 * N
    return ch_stats.data_ptr()


FF_FUSED_CHANNELS = (64, 128, 256, 320)


def ff_fused(a: torch.Tensor | None, w1: torch.Tensor, b1: torch.Tensor, w2: torch.Tensor, b2: torch.Tensor, *,
             residual: torch.Tensor | None = None, out_f32: torch.Tensor | None = None,
             out_f16: torch.Tensor | None = None, ln_x: torch.Tensor | None = None,
             ln_gamma: torch.Tensor | None = None, ln_beta: torch.Tensor | None = None, ln_eps: float = 1e-5) -> None:
    """out = W2 . geglu(W1 . a + b1) + b2 (+ residual) in one kernel (seva_ff_fused_f16); a: [M, C] f16, w1: [8C, C] f16
    (interleaved GEGLU layout), w2: [C, 4C] f16; C in FF_FUSED_CHANNELS.  With ln_x ([M, C] f32) the A operand is
    LayerNorm(ln_x) * ln_gamma + ln_beta, computed in the kernel's prologue (a = None)."""
    src = ln_x if ln_x is not None else a
    require_cuda(src, w1, w2)
    M, c = src.shape
    assert (ln_x is not None and ln_x.dtype == F32 and ln_gamma is not None and ln_beta is not None) or a.dtype == F16
    assert w1.dtype == F16 and w2.dtype == F16 and w1.shape == (8 * c, c) and w2.shape == (c, 4 * c)
    assert w1.is_contiguous() and w2.is_contiguous() and c in FF_FUSED_CHANNELS
    d = nv.FfDesc()
    d.a, d.w1, d.b1, d.w2, d.b2 = ptr(a) if ln_x is None else None, w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr()
    if ln_x is not None:
        d.ln_x, d.ln_gamma, d.ln_beta, d.ldx, d.ln_eps = ln_x.data_ptr(), ln_gamma.data_ptr(), ln_beta.data_ptr(), ln_x.stride(0), ln_eps
    d.residual, d.out_f32, d.out_f16 = ptr(residual), ptr(out_f32), ptr(out_f16)
    d.M, d.lda, d.C = M, (a.stride(0) if ln_x is None else 0), c
    d.ldr = residual.stride(0) if residual is not None else 0
    d.ldo32 = out_f32.stride(0) if out_f32 is not None else 0
    d.ldo16 = out_f16.stride(0) if out_f16 is not None else 0
    check(_lib().seva_ff_fused_f16(C.byref(d), stream_ptr(src.device)), "seva_ff_fused_f16")


def conv3x3(
    x: torch.Tensor,
    w: torch.Tensor,
    *,
    stride: int = 1,
    upsample: bool = False,
    bias: torch.Tensor | None = None,
    row_add: torch.Tensor | None = None,
    rows_per_group: int = 0,
    ld_row_add: int = 0,
    residual: torch.Tensor | None = None,
    out_f32: torch.Tensor | None = None,
    out_f16: torch.Tensor | None = None,
    pad_br_only: bool = False,
    w_exp: torch.Tensor | None = None,
    ch_stats: torch.Tensor | None = None,
    splitk_ws: torch.Tensor | None = None,
    a2: torch.Tensor | None = None,
    alg_k: int = 0,
) -> None:
    """3x3 pad-1 conv as implicit GEMM; x: [n, ih, iw, cin] f16 NHWC, w: [cout, 9*cin] f16.
    a2 ([M, K2] f16, K2 % 64 == 0): a second operand folded into the reduction behind the nine taps, w: [cout, 9*cin + K2]
    (seva_gemm_desc.a2: the ResBlock's 1x1 skip conv inside its second 3x3 conv).
    pad_br_only: zero padding at the bottom / right edge only (diffusers Downsample2D, pad (0,1,0,1)).
    fp8 mode (w_exp given): x and w are uint8 tensors of e4m3 bytes, cin % 128 == 0, no fused upsample."""
    require_cuda(x, w)
    fp8 = w_exp is not None
    assert x.dtype == w.dtype == (U8 if fp8 else F16) and x.dim() == 4 and x.is_contiguous()
    n, ih, iw, cin = x.shape
    eh, ew = (2 * ih, 2 * iw) if upsample else (ih, iw)
    ps = 1 if pad_br_only else 2
    oh, ow = (eh + ps - 3) // stride + 1, (ew + ps - 3) // stride + 1
    d = GemmDesc()
    d.a, d.w = x.data_ptr(), w.data_ptr()
    d.bias, d.row_add, d.residual = ptr(bias), ptr(row_add), ptr(residual)
    d.out_f32, d.out_f16 = ptr(out_f32), ptr(out_f16)
    d.M, d.N, d.K = n * oh * ow, w.shape[0], 9 * cin
    if a2 is not None:
        assert not fp8 and not upsample and a2.dtype == F16 and a2.dim() == 2 and a2.stride(1) == 1 and a2.shape[0] == n * oh * ow
        d.a2, d.lda2, d.K2 = a2.data_ptr(), a2.stride(0), a2.shape[1]
        d.K = 9 * cin + a2.shape[1]
    assert w.shape[1] == d.K
    d.alg_K = alg_k
    d.lda = cin
    d.ldr = residual.stride(-2) if residual is not None else 0
    d.ldo32 = out_f32.stride(-2) if out_f32 is not None else 0
    d.ldo16 = out_f16.stride(-2) if out_f16 is not None else 0
    d.rows_per_group, d.ld_row_add = rows_per_group, ld_row_add
    d.mode, d.epilogue = 1, 0
    d.n, d.ih, d.iw, d.cin, d.oh, d.ow = n, ih, iw, cin, oh, ow
    d.stride, d.upsample = stride, 1 if upsample else 0
    d.pad_br_only = 1 if pad_br_only else 0
    d.ch_stats = _stats_ptr(ch_stats, n * oh * ow, w.shape[0], out_f32)
    if splitk_ws is not None and not fp8:  # `splitk_workspace`: lets small-image convs run as split-K = 2 (seva_hip.h)
        assert splitk_ws.dtype == F32 and splitk_ws.is_contiguous()
        d.splitk_ws, d.splitk_ws_bytes = splitk_ws.data_ptr(), splitk_ws.numel() * 4
        _register_handoff_ws(splitk_ws)
    if fp8:
        assert w_exp.dtype == U8 and w_exp.numel() == w.shape[0]
        d.w_exp = w_exp.data_ptr()
        check(_lib().seva_gemm_fp8(C.byref(d), stream_ptr(x.device)), "seva_gemm_fp8(conv)")
    else:
        check(_lib().seva_gemm_f16(C.byref(d), stream_ptr(x.device)), "seva_gemm_f16(conv)")


def attention(
    q: torch.Tensor,
    k: torch.Tensor,
    v: torch.Tensor,
    out: torch.Tensor,
    *,
    nb0: int,
    nb1: int,
    heads: int,
    lq: int,
    lk: int,
    q_strides: tuple[int, int, int],
    k_strides: tuple[int, int, int],
    o_strides: tuple[int, int, int],
    scale: float = 0.125,
    q_prescaled: bool = False,
    split_ws: torch.Tensor | None = None,
) -> None:
    """softmax(q k^T * scale) v per (batch, head), head dim 64 (seva_attention_f16).
    q_prescaled: q already holds q * scale * log2(e) (gemm(col_scale=...)); `scale` is then ignored.

    q/k/v/out are f16 tensors (any views); strides are (batch-outer, batch-inner, token) in
    elements relative to the tensors' data pointers; head h sits at element offset 64*h."""
    require_cuda(q, k, v, out)
    d = AttnDesc()
    d.q, d.k, d.v, d.out = q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr()
    d.q_sb0, d.q_sb1, d.q_sl = q_strides
    d.k_sb0, d.k_sb1, d.k_sl = k_strides
    d.o_sb0, d.o_sb1, d.o_sl = o_strides
    d.nb0, d.nb1, d.heads, d.lq, d.lk, d.scale = nb0, nb1, heads, lq, lk, scale
    d.q_prescaled = 1 if q_prescaled else 0
    if split_ws is not None:  # `attention_split_workspace`: lets long key sequences run K/V-split (seva_attn_desc.split_ws)
        assert split_ws.dtype == F32 and split_ws.is_contiguous()
        d.split_ws, d.split_ws_bytes = split_ws.data_ptr(), split_ws.numel() * 4
    check(_lib().seva_attention_f16(C.byref(d), stream_ptr(q.device)), "seva_attention_f16")


ATTN_SPLIT_MIN_LK = 6144  # key length from which seva_attention_f16 splits the K/V range in two (include/seva_hip.h)


def attention_split_workspace_numel(batch: int, heads: int, lq: int, nsplit: int = 2) -> int:
    """fp32 elements of `attention(split_ws=...)`: per split, 64 floats of un-normalised O plus (m, l) per query row."""
    return nsplit * batch * heads * lq * 66


GN_WORKSPACE_SLABS = 1024  # include/seva_hip.h SEVA_GN_WORKSPACE_SLABS


def groupnorm_workspace(n: int, device) -> torch.Tensor:
    return torch.empty(n * GN_WORKSPACE_SLABS * 32 * 2, dtype=F32, device=device)


def groupnorm(
    x1: torch.Tensor,
    x2: torch.Tensor | None,
    gamma: torch.Tensor,
    beta: torch.Tensor,
    out_f16: torch.Tensor,
    workspace: torch.Tensor,
    *,
    groups: int = 32,
    eps: float = 1e-5,
    silu: bool = False,
    dense: torch.Tensor | None = None,
    dense_w: torch.Tensor | None = None,
    dense_b: torch.Tensor | None = None,
    raw_f16: torch.Tensor | None = None,
    out_f8: torch.Tensor | None = None,
    stats1: torch.Tensor | None = None,
    stats2: torch.Tensor | None = None,
    split_out: bool = False,
    split_raw: bool = False,
) -> None:
    """GroupNorm(+SiLU)(+Pluecker modulation) of cat(x1, x2) -> f16; x: [n, hw, c] f32.
    split_out / split_raw: out_f16 / raw_f16 are [n, hw, 2c] and carry [hi | lo] (seva_groupnorm_desc.split_*).
    raw_f16: optional second output, cat(x1, x2) merely cast to f16 (same pass).
    out_f8: optional e4m3 output (uint8 tensor, same layout); out_f16 may then be None.
    stats1 / stats2: the `ch_stats` buffers the kernels that produced x1 / x2 filled (both or none; hw % 64 == 0): the
    statistics pass over the fp32 tensors is skipped."""
    require_cuda(x1, out_f16 if out_f16 is not None else out_f8)
    n, hw, c1 = x1.shape
    c2 = x2.shape[2] if x2 is not None else 0
    d = GroupNormDesc()
    d.x1, d.x2, d.gamma, d.beta = x1.data_ptr(), ptr(x2), gamma.data_ptr(), beta.data_ptr()
    d.dense, d.dense_w, d.dense_b = ptr(dense), ptr(dense_w), ptr(dense_b)
    d.out_f16, d.workspace = ptr(out_f16), workspace.data_ptr()
    d.out_f8 = ptr(out_f8)
    if out_f8 is not None:  # [n, hw, >= c1 + c2] e4m3 bytes; a wider last dim = channel padding of an fp8 conv (pad bytes untouched)
        assert out_f8.dtype == U8 and out_f8.is_contiguous() and out_f8.shape[:2] == (n, hw) and out_f8.shape[2] >= c1 + c2
        d.ld_out_f8 = out_f8.shape[2]
    d.n, d.hw, d.c1, d.c2, d.groups = n, hw, c1, c2, groups
    d.dense_c = dense.shape[-1] if dense is not None else 0
    d.silu, d.eps = 1 if silu else 0, eps
    d.raw_f16 = ptr(raw_f16)
    if stats1 is not None:
        assert hw % STATS_ROWS == 0 and (x2 is None) == (stats2 is None)
        assert stats1.dtype == F32 and stats1.numel() >= (n * hw // STATS_ROWS) * 2 * c1
        assert stats2 is None or (stats2.dtype == F32 and stats2.numel() >= (n * hw // STATS_ROWS) * 2 * c2)
        d.stats1, d.stats2 = stats1.data_ptr(), ptr(stats2)
    else:
        assert stats2 is None
    assert raw_f16 is None or (raw_f16.dtype == F16 and raw_f16.is_contiguous()
                               and raw_f16.numel() == n * hw * (c1 + c2) * (2 if split_raw else 1))
    assert not split_out or (out_f16 is not None and out_f16.is_contiguous() and out_f16.numel() == 2 * n * hw * (c1 + c2))
    d.split_out_f16, d.split_raw_f16 = int(split_out), int(split_raw)
    assert workspace.numel() >= n * GN_WORKSPACE_SLABS * groups * 2
    check(_lib().seva_groupnorm_f16(C.byref(d), stream_ptr(x1.device)), "seva_groupnorm_f16")


def layernorm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, out_f16: torch.Tensor,
              eps: float = 1e-5) -> None:
    """LayerNorm over the last dim; the output dtype selects the kernel: f16, or uint8 = e4m3 bytes (seva_layernorm_fp8)."""
    require_cuda(x, out_f16)
    c = x.shape[-1]
    rows = x.numel() // c
    if out_f16.dtype == F32:
        assert out_f16.data_ptr() != x.data_ptr()
        check(_lib().seva_layernorm_f32(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                        out_f16.data_ptr(), rows, c, eps, stream_ptr(x.device)),
              "seva_layernorm_f32")
        return
    if out_f16.dtype == U8:  # e4m3: the output may be wider than c (K padded to a multiple of 128; pad bytes stay as they are)
        assert out_f16.dim() == 2 and out_f16.stride(1) == 1 and out_f16.shape[0] == rows
        check(_lib().seva_layernorm_fp8(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                        out_f16.data_ptr(), rows, c, eps, out_f16.stride(0), stream_ptr(x.device)),
              "seva_layernorm_fp8")
        return
    assert out_f16.dtype == F16
    check(_lib().seva_layernorm_f16(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                    out_f16.data_ptr(), rows, c, eps, stream_ptr(x.device)),
          "seva_layernorm_f16")


def quantize_weight_fp8(w: torch.Tensor):
    """[N, K] weights -> (e4m3 bytes [N, K] uint8, E8M0 scale bytes [N] uint8): row n holds e4m3(w[n] * 2^-e[n]) with the
    power-of-two scale 2^e[n] chosen so that max|row| lands in (224, 448] (the top binade of e4m3); the scale byte is
    127 + e[n], what the block-scaled MFMA takes (seva_gemm_fp8).  Pack-time host utility (torch's own e4m3 cast)."""
    w = w.float()
    amax = w.abs().amax(dim=1).clamp_min(1e-30)
    e = torch.ceil(torch.log2(amax / 448.0)).clamp(-126, 127)
    q = (w * torch.exp2(-e)[:, None]).clamp(-448.0, 448.0).to(torch.float8_e4m3fn)
    return q.view(torch.uint8).contiguous(), (e + 127).to(torch.uint8).contiguous()


def dequantize_weight_fp8(w8: torch.Tensor, w_exp: torch.Tensor) -> torch.Tensor:
    return w8.view(torch.float8_e4m3fn).float() * torch.exp2(w_exp.float() - 127.0)[:, None]


def to_fp8(x: torch.Tensor) -> torch.Tensor:
    """float tensor -> e4m3 bytes (saturating), uint8 view (host utility for tests / one-off conversions)."""
    return x.float().clamp(-448.0, 448.0).to(torch.float8_e4m3fn).view(torch.uint8)


def clip_preprocess(x: torch.Tensor, patches_f16: torch.Tensor, mean, std, *, out_size: int = 224, patch: int = 14,
                    antialias: bool = True) -> None:
    """kornia-style resize + CLIP normalisation, emitted as the patch matrix of the patch-embedding GEMM
    (seva_clip_preprocess_f16).  x: [n,3,H,W] f32 in [-1,1]; patches_f16: [n*(out/patch)^2, ld >= 3*patch^2] f16."""
    require_cuda(x, patches_f16)
    assert x.dtype == F32 and x.is_contiguous() and x.dim() == 4 and x.shape[1] == 3 and patches_f16.dtype == F16
    n, _, H, W = x.shape
    g = out_size // patch
    assert patches_f16.shape[0] == n * g * g and patches_f16.stride(1) == 1
    m = (C.c_float * 3)(*[float(v) for v in mean])
    sd = (C.c_float * 3)(*[float(v) for v in std])
    check(_lib().seva_clip_preprocess_f16(x.data_ptr(), patches_f16.data_ptr(), n, H, W, out_size, patch,
                                          patches_f16.stride(0), C.cast(m, C.c_void_p), C.cast(sd, C.c_void_p),
                                          1 if antialias else 0, stream_ptr(x.device)), "seva_clip_preprocess_f16")


def attention_small(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, out: torch.Tensor, *, batch: int, heads: int,
                    L: int, head_dim: int, q_strides: tuple[int, int], k_strides: tuple[int, int],
                    o_strides: tuple[int, int], scale: float) -> None:
    """softmax(q k^T * scale) v, short sequences, any even head dim <= 128 (seva_attention_small_f16); strides are
    (batch, token) in elements, head h at column h * head_dim."""
    require_cuda(q, k, v, out)
    assert q.dtype == k.dtype == v.dtype == out.dtype == F16
    check(_lib().seva_attention_small_f16(q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), q_strides[0],
                                          q_strides[1], k_strides[0], k_strides[1], o_strides[0], o_strides[1], batch,
                                          heads, L, head_dim, scale, stream_ptr(q.device)), "seva_attention_small_f16")


def softmax_rows(x: torch.Tensor, out_f16: torch.Tensor, cols: int, scale: float) -> None:
    """x: [rows, >=cols] f32 -> out_f16: [rows, cols_pad] f16 = softmax(x[:, :cols]*scale), zero padded."""
    require_cuda(x, out_f16)
    rows = x.shape[0]
    check(_lib().seva_softmax_rows_f16(x.data_ptr(), x.stride(0), out_f16.data_ptr(), out_f16.stride(0),
                                       rows, cols, out_f16.shape[1], scale, stream_ptr(x.device)),
          "seva_softmax_rows_f16")


def nchw_to_nhwc_f16(x1: torch.Tensor, x2: torch.Tensor | None, out_f16: torch.Tensor,
                     scale: torch.Tensor | None = None, split: bool = False) -> None:
    """split: channels [C, 2C) of the output receive the low parts f16(v - f32(f16(v))) (seva_nchw_to_nhwc_f16_split)."""
    require_cuda(x1, out_f16)
    n, c1 = x1.shape[:2]
    hw = x1.numel() // (n * c1)
    c2 = x2.shape[1] if x2 is not None else 0
    fn = _lib().seva_nchw_to_nhwc_f16_split if split else _lib().seva_nchw_to_nhwc_f16
    check(fn(x1.data_ptr(), c1, ptr(x2), c2, ptr(scale), out_f16.data_ptr(), n, hw, out_f16.shape[-1],
             stream_ptr(x1.device)), "seva_nchw_to_nhwc_f16")


def nhwc_to_nchw_f32(x: torch.Tensor, out: torch.Tensor) -> None:
    """x: [n, hw, ld] f32 (first c channels used) -> out [n, c, h, w] f32."""
    require_cuda(x, out)
    n, c = out.shape[:2]
    hw = out.numel() // (n * c)
    check(_lib().seva_nhwc_to_nchw_f32(x.data_ptr(), x.stride(-2), out.data_ptr(), n, c, hw,
                                       stream_ptr(x.device)), "seva_nhwc_to_nchw_f32")


def cast_concat_f16(x1: torch.Tensor, x2: torch.Tensor | None, out_f16: torch.Tensor) -> None:
    require_cuda(x1, out_f16)
    c1 = x1.shape[-1]
    rows = x1.numel() // c1
    c2 = x2.shape[-1] if x2 is not None else 0
    check(_lib().seva_cast_concat_f16(x1.data_ptr(), c1, ptr(x2), c2, out_f16.data_ptr(), rows,
                                      stream_ptr(x1.device)), "seva_cast_concat_f16")


def bilinear_to_nhwc(src: torch.Tensor, out: torch.Tensor, oh: int, ow: int) -> None:
    require_cuda(src, out)
    n, c, sh, sw = src.shape
    check(_lib().seva_bilinear_to_nhwc_f32(src.data_ptr(), out.data_ptr(), n, c, sh, sw, oh, ow,
                                           stream_ptr(src.device)), "seva_bilinear_to_nhwc_f32")


def timestep_embedding_f16(t: torch.Tensor, freqs: torch.Tensor, out_f16: torch.Tensor) -> None:
    require_cuda(t, out_f16)
    assert t.dtype == torch.int64
    n, dim = out_f16.shape
    check(_lib().seva_timestep_embedding_f16(t.data_ptr(), freqs.data_ptr(), out_f16.data_ptr(),
                                             n, dim, stream_ptr(t.device)),
          "seva_timestep_embedding_f16")


def silu_f16(x: torch.Tensor, out_f16: torch.Tensor) -> None:
    require_cuda(x, out_f16)
    check(_lib().seva_silu_f16(x.data_ptr(), out_f16.data_ptr(), x.numel(), stream_ptr(x.device)),
          "seva_silu_f16")


def add_f32(a: torch.Tensor, b: torch.Tensor, out: torch.Tensor) -> None:
    require_cuda(a, b, out)
    check(_lib().seva_add_f32(a.data_ptr(), b.data_ptr(), out.data_ptr(), a.numel(),
                              stream_ptr(a.device)), "seva_add_f32")


def replace_blend(x: torch.Tensor, replace: torch.Tensor, out: torch.Tensor) -> None:
    require_cuda(x, replace, out)
    n, c = x.shape[:2]
    hw = x.numel() // (n * c)
    check(_lib().seva_replace_blend_f32(x.data_ptr(), replace.data_ptr(), out.data_ptr(), n, c, hw,
                                        stream_ptr(x.device)), "seva_replace_blend_f32")


def denoiser_combine(net: torch.Tensor, x: torch.Tensor, c_out: torch.Tensor, c_skip: torch.Tensor,
                     out: torch.Tensor) -> None:
    require_cuda(net, x, out)
    n = x.shape[0]
    check(_lib().seva_denoiser_combine_f32(net.data_ptr(), x.data_ptr(), c_out.data_ptr(),
                                           c_skip.data_ptr(), out.data_ptr(), n, x.numel() // n,
                                           stream_ptr(x.device)), "seva_denoiser_combine_f32")


def add_noise(x: torch.Tensor, eps: torch.Tensor, noise_scale: torch.Tensor, out: torch.Tensor) -> None:
    require_cuda(x, eps, out)
    n = x.shape[0]
    check(_lib().seva_add_noise_f32(x.data_ptr(), eps.data_ptr(), noise_scale.data_ptr(),
                                    out.data_ptr(), n, x.numel() // n, stream_ptr(x.device)),
          "seva_add_noise_f32")


def cfg_euler(x: torch.Tensor, den2: torch.Tensor, scale: torch.Tensor, sigma_hat: torch.Tensor,
              dt: torch.Tensor, out: torch.Tensor) -> None:
    require_cuda(x, den2, out)
    n = x.shape[0]
    check(_lib().seva_cfg_euler_f32(x.data_ptr(), den2.data_ptr(), scale.data_ptr(),
                                    sigma_hat.data_ptr(), dt.data_ptr(), out.data_ptr(), n,
                                    x.numel() // n, stream_ptr(x.device)), "seva_cfg_euler_f32")


def cfg_combine(den2: torch.Tensor, scale: torch.Tensor, out: torch.Tensor) -> None:
    require_cuda(den2, scale, out)
    n = out.shape[0]
    check(_lib().seva_cfg_combine_f32(den2.data_ptr(), scale.data_ptr(), out.data_ptr(), n,
                                      out.numel() // n, stream_ptr(out.device)), "seva_cfg_combine_f32")


def euler_step(x: torch.Tensor, den: torch.Tensor, sigma_hat: torch.Tensor, dt: torch.Tensor,
               out: torch.Tensor) -> None:
    require_cuda(x, den, out)
    n = x.shape[0]
    check(_lib().seva_euler_step_f32(x.data_ptr(), den.data_ptr(), sigma_hat.data_ptr(),
                                     dt.data_ptr(), out.data_ptr(), n, x.numel() // n,
                                     stream_ptr(x.device)), "seva_euler_step_f32")


def to_d(x: torch.Tensor, den: torch.Tensor, sigma: torch.Tensor, out: torch.Tensor) -> None:
    require_cuda(x, den, out)
    n = x.shape[0]
    check(_lib().seva_to_d_f32(x.data_ptr(), den.data_ptr(), sigma.data_ptr(), out.data_ptr(), n,
                               x.numel() // n, stream_ptr(x.device)), "seva_to_d_f32")


def scale_rows(x: torch.Tensor, s: torch.Tensor, out: torch.Tensor) -> None:
    require_cuda(x, s, out)
    n = x.shape[0]
    check(_lib().seva_scale_rows_f32(x.data_ptr(), s.data_ptr(), out.data_ptr(), n,
                                     x.numel() // n, stream_ptr(x.device)), "seva_scale_rows_f32")


def plucker(kinv: torch.Tensor, pose_inv: torch.Tensor, out: torch.Tensor) -> None:
    """out[v] (6,h,w) = Pluecker map of view v from its inverse intrinsics (3x3, latent-pixel units) and the
    inverse relative pose rows (3x4) (seva_plucker_f32)."""
    require_cuda(kinv, pose_inv, out)
    assert kinv.dtype == F32 and pose_inv.dtype == F32 and out.dtype == F32
    assert kinv.is_contiguous() and pose_inv.is_contiguous() and out.is_contiguous()
    V, six, h, w = out.shape
    assert six == 6 and kinv.shape == (V, 3, 3) and pose_inv.shape == (V, 3, 4)
    check(_lib().seva_plucker_f32(kinv.data_ptr(), pose_inv.data_ptr(), out.data_ptr(), V, h, w,
                                  stream_ptr(out.device)), "seva_plucker_f32")


def cond_concat(plucker_maps: torch.Tensor, mask_u8: torch.Tensor, c_concat: torch.Tensor, uc_concat: torch.Tensor) -> None:
    """c_concat = [mask | plucker], uc_concat = [0 | plucker] (seva_cond_concat_f32)."""
    require_cuda(plucker_maps, mask_u8, c_concat, uc_concat)
    V, six, h, w = plucker_maps.shape
    assert six == 6 and mask_u8.dtype == torch.uint8 and mask_u8.numel() == V
    assert c_concat.shape == (V, 7, h, w) and uc_concat.shape == (V, 7, h, w)
    assert plucker_maps.is_contiguous() and c_concat.is_contiguous() and uc_concat.is_contiguous()
    check(_lib().seva_cond_concat_f32(plucker_maps.data_ptr(), mask_u8.data_ptr(), c_concat.data_ptr(),
                                      uc_concat.data_ptr(), V, h, w, stream_ptr(plucker_maps.device)),
          "seva_cond_concat_f32")


# --- benchmark / debugging knobs (read from SEVA_* once at library load; see include/seva_hip.h) ---------
def set_knob(name: str, value: int) -> None:
    check(_lib().seva_set_knob(name.encode(), int(value)), "seva_set_knob")


def get_knob(name: str) -> int:
    v = C.c_int32()
    check(_lib().seva_get_knob(name.encode(), C.byref(v)), "seva_get_knob")
    return v.value


# --- profiling / graphs ---------------------------------------------------------------------
def prof_enable(on: bool) -> None:
    check(_lib().seva_prof_enable(1 if on else 0))


def prof_collect() -> dict:
    ms = (C.c_double * nv.PROF_CLASSES)()
    n = (C.c_int64 * nv.PROF_CLASSES)()
    work = (C.c_double * nv.PROF_CLASSES)()
    nbytes = (C.c_double * nv.PROF_CLASSES)()
    check(_lib().seva_prof_collect(ms, n, work, nbytes), "seva_prof_collect")
    check_handoffs()  # (prof_collect synchronises anyway)
    return {name: {"ms": ms[i], "launches": n[i], "work": work[i], "bytes": nbytes[i]}
            for i, name in enumerate(nv.PROF_NAMES)}


class Graph:
    """hipGraph captured from PyTorch's current stream through the C-ABI helpers."""

    def __init__(self):
        self._exec = C.c_void_p()

    def capture_begin(self, device=None) -> None:
        check(_lib().seva_graph_begin(stream_ptr(device)), "seva_graph_begin")

    def capture_end(self, device=None) -> None:
        check(_lib().seva_graph_end(stream_ptr(device), C.byref(self._exec)), "seva_graph_end")

    def launch(self, device=None) -> None:
        check(_lib().seva_graph_launch(self._exec, stream_ptr(device)), "seva_graph_launch")

    def __del__(self):
        try:
            if self._exec:
                _lib().seva_graph_destroy(self._exec)
        except Exception:
            pass
