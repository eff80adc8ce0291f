"""CPU oracle for the Seva denoiser network -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A functional, state_dict-driven fp32 restatement of the reference network
(`/root/reference/seva/model.py`, `seva/modules/layers.py`, `seva/modules/transformer.py`).
Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py``
may import this file; the product path (``stable-virtual-camera_amd/seva``) never does.

Parity status: PINNED.  ``tests/test_oracle_golden.py`` checks every function below against
golden vectors produced by importing the reference itself (``oracle/make_goldens.py``,
fixtures under ``tests/golden/``).

The structure of the UNet is recovered from the state_dict *keys* (which sub-modules exist
under ``input_blocks.N.M``), so the oracle shares no architecture code with the product.
Each function cites the reference lines it restates.
"""

from __future__ import annotations

import math
import re

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# ----------------------------------------------------------------------------- layers.py
def timestep_embedding(t: Tensor, dim: int, max_period: int = 10000) -> Tensor:
    """seva/modules/layers.py:11-32 -- cos‖sin sinusoid, `dim//2` frequencies."""
    half = dim // 2
    k = torch.arange(half, dtype=torch.float32)
    freqs = torch.exp(-math.log(max_period) * k / half)
    args = t[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


def group_norm(sd, p: str, x: Tensor, eps: float) -> Tensor:
    """GroupNorm(32, C) -- layers.py:61-63 (fp32), transformer.py:186 (eps 1e-6)."""
    return F.group_norm(x.float(), 32, sd[p + ".weight"], sd[p + ".bias"], eps)


def conv2d(sd, p: str, x: Tensor, stride: int = 1, padding: int = 1) -> Tensor:
    return F.conv2d(x, sd[p + ".weight"], sd[p + ".bias"], stride=stride, padding=padding)


def linear(sd, p: str, x: Tensor, bias: bool = True) -> Tensor:
    return F.linear(x, sd[p + ".weight"], sd[p + ".bias"] if bias else None)


def resblock(sd, p: str, x: Tensor, emb: Tensor, dense: Tensor) -> Tensor:
    """seva/modules/layers.py:120-139."""
    h = F.silu(group_norm(sd, p + ".in_layers.0", x, 1e-5))
    d = F.interpolate(dense, size=h.shape[2:], mode="bilinear", align_corners=True)
    d = conv2d(sd, p + ".dense_emb_layers.0", d, padding=0)
    scale, shift = torch.chunk(d, 2, dim=1)
    h = h * (1 + scale) + shift
    h = conv2d(sd, p + ".in_layers.2", h)
    e = linear(sd, p + ".emb_layers.1", F.silu(emb))
    h = h + e[:, :, None, None]
    h = F.silu(group_norm(sd, p + ".out_layers.0", h, 1e-5))
    h = conv2d(sd, p + ".out_layers.3", h)
    if (p + ".skip_connection.weight") in sd:
        x = conv2d(sd, p + ".skip_connection", x, padding=0)
    return x + h


def upsample(sd, p: str, x: Tensor) -> Tensor:
    """layers.py:35-46 -- nearest x2 then conv3x3."""
    x = F.interpolate(x, scale_factor=2, mode="nearest")
    return conv2d(sd, p + ".conv", x)


def downsample(sd, p: str, x: Tensor) -> Tensor:
    """layers.py:49-58 -- conv3x3 stride 2."""
    return conv2d(sd, p + ".op", x, stride=2)


# ------------------------------------------------------------------------ transformer.py
def attention(sd, p: str, x: Tensor, context: Tensor | None, dim_head: int = 64) -> Tensor:
    """transformer.py:59-74.  SDPA written out: softmax(q k^T / sqrt(d)) v, no mask."""
    ctx = x if context is None else context
    q = F.linear(x, sd[p + ".to_q.weight"])
    k = F.linear(ctx, sd[p + ".to_k.weight"])
    v = F.linear(ctx, sd[p + ".to_v.weight"])
    b, lq, inner = q.shape
    heads = inner // dim_head
    lk = k.shape[1]
    q = q.view(b, lq, heads, dim_head).transpose(1, 2)
    k = k.view(b, lk, heads, dim_head).transpose(1, 2)
    v = v.view(b, lk, heads, dim_head).transpose(1, 2)
    out = torch.empty_like(q)
    scale = 1.0 / math.sqrt(dim_head)
    # chunk over queries to bound the score matrix
    step = max(1, (1 << 24) // max(lk, 1))
    for s in range(0, lq, step):
        att = torch.matmul(q[:, :, s : s + step], k.transpose(-1, -2)) * scale
        att = torch.softmax(att, dim=-1)
        out[:, :, s : s + step] = torch.matmul(att, v)
    out = out.transpose(1, 2).reshape(b, lq, inner)
    return linear(sd, p + ".to_out.0", out)


def feedforward(sd, p: str, x: Tensor) -> Tensor:
    """transformer.py:8-34 -- GEGLU (exact erf GELU) then Linear."""
    a, gate = linear(sd, p + ".net.0.proj", x).chunk(2, dim=-1)
    return linear(sd, p + ".net.2", a * F.gelu(gate))


def layer_norm(sd, p: str, x: Tensor) -> Tensor:
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], 1e-5)


def transformer_block(sd, p: str, x: Tensor, context: Tensor) -> Tensor:
    """transformer.py:106-110."""
    x = attention(sd, p + ".attn1", layer_norm(sd, p + ".norm1", x), None) + x
    x = attention(sd, p + ".attn2", layer_norm(sd, p + ".norm2", x), context) + x
    x = feedforward(sd, p + ".ff", layer_norm(sd, p + ".norm3", x)) + x
    return x


def timemix_block(sd, p: str, x: Tensor, context: Tensor, num_frames: int) -> Tensor:
    """transformer.py:145-155.  x: (b*t, s, c); attention runs over t for every pixel."""
    bt, s, c = x.shape
    b = bt // num_frames
    x = x.view(b, num_frames, s, c).transpose(1, 2).reshape(b * s, num_frames, c)
    x = feedforward(sd, p + ".ff_in", layer_norm(sd, p + ".norm_in", x)) + x
    x = attention(sd, p + ".attn1", layer_norm(sd, p + ".norm1", x), None) + x
    x = attention(sd, p + ".attn2", layer_norm(sd, p + ".norm2", x), context) + x
    x = feedforward(sd, p + ".ff", layer_norm(sd, p + ".norm3", x))  # no residual
    x = x.view(b, s, num_frames, c).transpose(1, 2).reshape(bt, s, c)
    return x


def multiview_transformer(
    sd, p: str, x: Tensor, context: Tensor, num_frames: int, joint: bool
) -> Tensor:
    """transformer.py:215-247.  `joint` == (name in unflatten_names)."""
    n, c, h, w = x.shape
    x_in = x
    time_ctx = context[::num_frames]  # (b,1,1024)
    time_ctx = time_ctx.repeat_interleave(h * w, dim=0)  # "(b n) ..." , n = h*w
    if joint:
        context = context[::num_frames]
    x = group_norm(sd, p + ".norm", x, 1e-6)
    x = x.permute(0, 2, 3, 1).reshape(n, h * w, c)
    x = linear(sd, p + ".proj_in", x)
    depth = 0
    while f"{p}.transformer_blocks.{depth}.norm1.weight" in sd:
        depth += 1
    for i in range(depth):
        if joint:
            x = x.reshape(n // num_frames, num_frames * h * w, -1)
        x = transformer_block(sd, f"{p}.transformer_blocks.{i}", x, context)
        if joint:
            x = x.reshape(n, h * w, -1)
        x_mix = timemix_block(sd, f"{p}.time_mix_blocks.{i}", x, time_ctx, num_frames)
        x = x + x_mix
    x = linear(sd, p + ".proj_out", x)
    x = x.reshape(n, h, w, c).permute(0, 3, 1, 2)
    return x + x_in


# ------------------------------------------------------------------------------ model.py
def _sublayers(sd, prefix: str) -> list[int]:
    pat = re.compile(re.escape(prefix) + r"\.(\d+)\.")
    idx = set()
    for k in sd:
        m = pat.match(k)
        if m:
            idx.add(int(m.group(1)))
    return sorted(idx)


def _run_sequential(sd, prefix, x, emb, ctx, dense, num_frames, tname, joint_names, trace=None):
    """TimestepEmbedSequential dispatch, layers.py:66-83, keyed on which params exist.
    `trace` (dict) receives every sub-layer's output keyed by its state_dict prefix."""
    ds_change = 0
    for j in _sublayers(sd, prefix):
        p = f"{prefix}.{j}"
        if trace is not None and j > 0:
            trace[f"{prefix}.{j - 1}"] = x
        if p + ".in_layers.0.weight" in sd:
            x = resblock(sd, p, x, emb, dense)
        elif p + ".proj_in.weight" in sd:
            x = multiview_transformer(sd, p, x, ctx, num_frames, tname in joint_names)
        elif p + ".op.weight" in sd:
            x = downsample(sd, p, x)
            ds_change = +1
        elif p + ".conv.weight" in sd:
            x = upsample(sd, p, x)
            ds_change = -1
        elif p + ".weight" in sd:
            x = conv2d(sd, p, x)
        else:
            raise KeyError(f"oracle: cannot classify layer {p}")
    if trace is not None:
        trace[p] = x
    return x, ds_change


def seva_forward(
    sd: dict,
    x: Tensor,
    t: Tensor,
    y: Tensor,
    dense_y: Tensor,
    num_frames: int,
    joint_names=("middle_ds8", "output_ds4", "output_ds2"),
    trace: dict | None = None,
) -> Tensor:
    """Seva.forward, seva/model.py:176-216."""
    model_channels = sd["time_embed.0.weight"].shape[1]
    emb = timestep_embedding(t, model_channels)
    emb = linear(sd, "time_embed.2", F.silu(linear(sd, "time_embed.0", emb)))
    hs = []
    h = x
    ds = 1
    for i in _sublayers(sd, "input_blocks"):
        h, dch = _run_sequential(
            sd, f"input_blocks.{i}", h, emb, y, dense_y, num_frames, f"input_ds{ds}", joint_names, trace
        )
        if dch > 0:
            ds *= 2
        hs.append(h)
    h, _ = _run_sequential(
        sd, "middle_block", h, emb, y, dense_y, num_frames, f"middle_ds{ds}", joint_names, trace
    )
    for i in _sublayers(sd, "output_blocks"):
        h = torch.cat([h, hs.pop()], dim=1)
        h, dch = _run_sequential(
            sd, f"output_blocks.{i}", h, emb, y, dense_y, num_frames, f"output_ds{ds}", joint_names, trace
        )
        if dch < 0:
            ds //= 2
    h = F.silu(group_norm(sd, "out.0", h, 1e-5))
    return conv2d(sd, "out.2", h)


def sgm_wrapper_forward(sd: dict, x: Tensor, t: Tensor, c: dict, num_frames: int, **kw) -> Tensor:
    """SGMWrapper.forward, seva/model.py:224-234 (keys carry no `module.` prefix here)."""
    if "concat" in c:
        x = torch.cat((x, c["concat"]), dim=1)
    return seva_forward(sd, x, t, c["crossattn"], c["dense_vector"], num_frames, **kw)
