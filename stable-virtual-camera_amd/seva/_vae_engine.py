"""HIP execution of the SD-2.1 VAE: decoder (reference seva/modules/autoencoder.py:37-48 -> diffusers
AutoencoderKL.decode) and encoder (autoencoder.py:21-35 -> AutoencoderKL.encode(x).latent_dist.mean).  Same kernels and layout rules as the UNet engine: channels-last
fp32 stream, fp16 GEMM operands, 3x3 convs as implicit GEMM with the nearest-2x upsample fused
into the gather.  The single-head d=512 mid-block attention runs as GEMM(QK^T) -> row softmax ->
GEMM(P V^T) with V produced already transposed by swapping the GEMM operand roles; the value
bias is added after P*V (rows of P sum to 1, exact).
"""

from __future__ import annotations

import torch

from . import ops
from ._engine import CIN_PAD, _Arena, pack_conv3x3
from ._native import SevaNativeError, require_cuda

F16, F32 = torch.float16, torch.float32


class _VaeEngineBase:
    PREFIXES: tuple = ()       # state_dict key prefixes this half owns
    CONV_IN = CONV_OUT = ""    # first conv (input channels padded to 64) / last conv (handled by the subclass)

    @staticmethod
    def _resolve_device(weights) -> torch.device:
        params = list(weights.parameters())
        if not params or params[0].device.type != "cuda":
            raise SevaNativeError("AutoEncoder runs only on an AMD GPU (no CPU fallback): call .to('cuda')")
        from . import _native

        _native.load()
        return params[0].device

    def __init__(self, weights):
        self.device = self._resolve_device(weights)
        self.block_out = weights.block_out
        self.out_channels = weights.out_channels
        self.latent = weights.latent_channels
        self.arena = _Arena(self.device)
        import os as _os
        self.gn_fused_stats = int(_os.environ.get("SEVA_GN_FUSED_STATS", "1"))  # seva/_engine.py: same switch
        self._stats: dict = {}
        sd = {k: v.detach().to(self.device) for k, v in weights.state_dict().items() if k.startswith(self.PREFIXES)}
        W = {}

        def conv3(p, cin_pad=None, cout_pad=None):
            w, b = sd[p + ".weight"].float(), sd[p + ".bias"].float()
            if cout_pad and cout_pad > w.shape[0]:
                w = torch.cat([w, w.new_zeros((cout_pad - w.shape[0],) + tuple(w.shape[1:]))], 0)
                b = torch.cat([b, b.new_zeros(cout_pad - b.shape[0])], 0)
            W[p + ".w"], W[p + ".b"] = pack_conv3x3(w, cin_pad), b.contiguous()

        def lin(p, src=None):
            src = src or p
            w = sd[src + ".weight"].float()
            W[p + ".w"] = w.reshape(w.shape[0], -1).to(F16).contiguous()
            W[p + ".b"] = sd[src + ".bias"].float().contiguous()

        def norm(p):
            W[p + ".g"], W[p + ".b"] = sd[p + ".weight"].float().contiguous(), sd[p + ".bias"].float().contiguous()

        for k in list(sd):
            if not k.endswith(".weight"):
                continue
            p = k[: -len(".weight")]
            shp = sd[k].shape
            if len(shp) == 1:
                norm(p)
            elif len(shp) == 2 or shp[-1] == 1:
                if p not in ("post_quant_conv", "quant_conv"):
                    lin(p)
            elif p == self.CONV_IN:
                conv3(p, cin_pad=CIN_PAD)
            elif p == self.CONV_OUT:
                pass  # subclass
            else:
                conv3(p)
        # channel-changing resnets: the 1x1 shortcut conv rides in the K loop of the second 3x3 conv (seva_gemm_desc.a2): one
        # accumulation, and the shortcut result (fp32, up to 576 x 576 x 128 per frame) is neither written nor read back
        self.fold_shortcut = _os.environ.get("SEVA_VAE_FOLD_SHORTCUT", "1") != "0"
        for k in list(W):
            if k.endswith(".conv_shortcut.w") and W[k].shape[1] % 64 == 0:
                p = k[: -len(".conv_shortcut.w")]
                W[p + ".conv2.wf"] = torch.cat([W[p + ".conv2.w"], W[k]], 1).contiguous()
                W[p + ".conv2.bf"] = (W[p + ".conv2.b"] + W[p + ".conv_shortcut.b"]).contiguous()
        self.W = W
        self._pack_ends(sd, conv3)

    def _buf(self, name, shape, dtype, zero=False):
        key = (name, tuple(int(s) for s in shape), dtype)
        fresh = key not in self.arena.bufs
        t = self.arena.get(name, shape, dtype)
        if fresh and zero:
            t.zero_()
        return t

    # GroupNorm statistics from the producing conv / GEMM epilogue (same scheme as seva/_engine.py:_stats_buf)
    def _stats_buf(self, name, rows, hw, c):
        if not self.gn_fused_stats or hw % ops.STATS_ROWS or c < 128 or c % 4:
            return None
        if self.gn_fused_stats < 2 and (hw // 128) * ((c + 159) // 160) < 16:  # per sample, never per batch
            return None
        return self._buf("st:" + name, ops.channel_stats_shape(rows, c), F32)

    def _produced(self, out, st):
        if st is None:
            self._stats.pop(out.data_ptr(), None)
        else:
            self._stats[out.data_ptr()] = st

    def _resnet(self, p, x, n, h, w, cin, cout, f16_out=None):
        """diffusers ResnetBlock2D (no time embedding): GN-SiLU-conv-GN-SiLU-conv + shortcut.
        f16_out: the block's only consumer is a resampling conv (A operand = f16): the second conv's epilogue rounds
        `conv + shortcut` straight into that buffer -- the same rounding the separate cast pass made -- and the fp32 tensor
        (4 B written, 4 B read back per element at up to 576 x 576 x 256) is never formed.  The shortcut conv's f16 input comes
        out of the first GroupNorm's pass over x (`raw_f16`) instead of a cast pass of its own."""
        W, hw = self.W, h * w
        a16 = self._buf("gn16", (n, hw, cin), F16)
        xs16 = self._buf("v_sk16", (n * hw, cin), F16) if cin != cout else None
        ops.groupnorm(x, None, W[p + ".norm1.g"], W[p + ".norm1.b"], a16, self.gn_ws, eps=1e-6, silu=True,
                      stats1=self._stats.get(x.data_ptr()), raw_f16=None if xs16 is None else xs16.view(n, hw, cin))
        mid = self._buf("v_mid", (n, hw, cout), F32)
        st_mid = self._stats_buf("v_mid", n * hw, hw, cout)
        ops.conv3x3(a16.view(n, h, w, cin), W[p + ".conv1.w"], bias=W[p + ".conv1.b"], out_f32=mid, ch_stats=st_mid)
        b16 = self._buf("gn16", (n, hw, cout), F16)
        ops.groupnorm(mid, None, W[p + ".norm2.g"], W[p + ".norm2.b"], b16, self.gn_ws, eps=1e-6, silu=True, stats1=st_mid)
        w2, b2, res, a2 = W[p + ".conv2.w"], W[p + ".conv2.b"], x, None
        if cin != cout:
            if self.fold_shortcut and (p + ".conv2.wf") in W:
                w2, b2, res, a2 = W[p + ".conv2.wf"], W[p + ".conv2.bf"], None, xs16
            else:
                res = self._buf("v_sk32", (n * hw, cout), F32)
                ops.gemm(xs16, W[p + ".conv_shortcut.w"], bias=W[p + ".conv_shortcut.b"], out_f32=res)
        if f16_out is not None:
            ops.conv3x3(b16.view(n, h, w, cout), w2, bias=b2, residual=res, a2=a2, out_f16=f16_out.view(n, hw, cout))
            return f16_out
        out = self._buf("out:" + p, (n, hw, cout), F32)
        st_out = self._stats_buf("out:" + p, n * hw, hw, cout)
        ops.conv3x3(b16.view(n, h, w, cout), w2, bias=b2, residual=res, a2=a2, out_f32=out, ch_stats=st_out)
        self._produced(out, st_out)
        return out

    def _attention(self, p, x, n, h, w, c):
        """Single-head self-attention over h*w tokens of dim c (diffusers Attention in the VAE mid block)."""
        W, hw = self.W, h * w
        if hw % 4:
            raise ValueError(f"VAE attention needs h*w % 4 == 0 (got {h}x{w}); latents are multiples of 8 per side")
        hw_pad = 64 * ((hw + 63) // 64)
        g16 = self._buf("gn16", (n, hw, c), F16)
        ops.groupnorm(x, None, W[p + ".group_norm.g"], W[p + ".group_norm.b"], g16, self.gn_ws, eps=1e-6, silu=False,
                      stats1=self._stats.get(x.data_ptr()))
        q = self._buf("v_q", (n * hw, c), F16)
        k = self._buf("v_k", (n * hw, c), F16)
        ops.gemm(g16.view(n * hw, c), W[p + ".to_q.w"], bias=W[p + ".to_q.b"], out_f16=q)
        ops.gemm(g16.view(n * hw, c), W[p + ".to_k.w"], bias=W[p + ".to_k.b"], out_f16=k)
        att = self._buf("v_att", (n * hw, c), F16)
        vT = self._buf("v_vT", (c, hw_pad), F16, zero=True)       # V^T, zero beyond hw
        sc = self._buf("v_sc", (hw, hw_pad), F32)
        pr = self._buf("v_pr", (hw, hw_pad), F16)
        for i in range(n):
            gi = g16.view(n, hw, c)[i]
            ops.gemm(W[p + ".to_v.w"], gi, out_f16=vT)             # [c, hw] = W_v @ x_i^T (bias folded below)
            ops.gemm(q[i * hw:(i + 1) * hw], k[i * hw:(i + 1) * hw], out_f32=sc)
            ops.softmax_rows(sc, pr, hw, 1.0 / (c**0.5))
            ops.gemm(pr, vT, bias=W[p + ".to_v.b"], out_f16=att[i * hw:(i + 1) * hw])
        out = self._buf("out:" + p, (n, hw, c), F32)
        st_out = self._stats_buf("out:" + p, n * hw, hw, c)
        ops.gemm(att, W[p + ".to_out.0.w"], bias=W[p + ".to_out.0.b"], residual=x.view(n * hw, c),
                 out_f32=out.view(n * hw, c), ch_stats=st_out)
        self._produced(out, st_out)
        return out



class VaeDecoderEngine(_VaeEngineBase):
    PREFIXES = ("decoder.", "post_quant_conv.")
    CONV_IN, CONV_OUT = "decoder.conv_in", "decoder.conv_out"

    def _pack_ends(self, sd, conv3):
        conv3("decoder.conv_out", cout_pad=4)
        # post_quant_conv (1x1, 4->4) as a GEMM over the 64-channel padded latent image
        wq = torch.zeros((self.latent, CIN_PAD), dtype=F16, device=self.device)
        wq[:, : self.latent] = sd["post_quant_conv.weight"].reshape(self.latent, self.latent).to(F16)
        self.W["post_quant_conv.w"], self.W["post_quant_conv.b"] = wq, sd["post_quant_conv.bias"].float().contiguous()

    @torch.no_grad()
    def decode(self, z: torch.Tensor, scale_factor: float) -> torch.Tensor:
        require_cuda(z)
        W = self.W
        z = z.to(F32).contiguous()
        n, cz, h, w = z.shape
        if cz != self.latent:
            raise ValueError(f"expected {self.latent} latent channels, got {cz}")
        self.gn_ws = self._buf("gn_ws", (n * ops.GN_WORKSPACE_SLABS * 32 * 2,), F32)
        self._stats = {}
        inv = torch.full((n,), 1.0 / scale_factor, dtype=F32, device=self.device)
        z16 = self._buf("v_z16", (n, h * w, CIN_PAD), F16)
        ops.nchw_to_nhwc_f16(z, None, z16, scale=inv)                      # z / 0.18215, channels-last, padded
        pq = self._buf("v_pq16", (n * h * w, CIN_PAD), F16, zero=True)      # cols >= 4 stay zero
        ops.gemm(z16.view(n * h * w, CIN_PAD), W["post_quant_conv.w"], bias=W["post_quant_conv.b"], out_f16=pq)
        top = self.block_out[-1]
        x = self._buf("out:conv_in", (n, h * w, top), F32)
        st = self._stats_buf("out:conv_in", n * h * w, h * w, top)
        ops.conv3x3(pq.view(n, h, w, CIN_PAD), W["decoder.conv_in.w"], bias=W["decoder.conv_in.b"], out_f32=x, ch_stats=st)
        self._produced(x, st)
        x = self._resnet("decoder.mid_block.resnets.0", x, n, h, w, top, top)
        x = self._attention("decoder.mid_block.attentions.0", x, n, h, w, top)
        x = self._resnet("decoder.mid_block.resnets.1", x, n, h, w, top, top)
        rev = list(reversed(self.block_out))
        cin = rev[0]
        for i, cout in enumerate(rev):
            up = i != len(rev) - 1
            x16 = self._buf("v_up16", (n, h, w, cout), F16) if up else None
            for j in range(3):
                x = self._resnet(f"decoder.up_blocks.{i}.resnets.{j}", x, n, h, w, cin if j == 0 else cout, cout,
                                 f16_out=x16 if j == 2 else None)
            cin = cout
            if up:
                p = f"decoder.up_blocks.{i}.upsamplers.0.conv"
                h, w = 2 * h, 2 * w
                x = self._buf("out:" + p, (n, h * w, cout), F32)
                st = self._stats_buf("out:" + p, n * h * w, h * w, cout)
                ops.conv3x3(x16, W[p + ".w"], upsample=True, bias=W[p + ".b"], out_f32=x, ch_stats=st)
                self._produced(x, st)
        c = rev[-1]
        g16 = self._buf("gn16", (n, h * w, c), F16)
        ops.groupnorm(x, None, W["decoder.conv_norm_out.g"], W["decoder.conv_norm_out.b"], g16, self.gn_ws,
                      eps=1e-6, silu=True, stats1=self._stats.get(x.data_ptr()))
        o4 = self._buf("v_o4", (n, h * w, 4), F32)
        ops.conv3x3(g16.view(n, h, w, c), W["decoder.conv_out.w"], bias=W["decoder.conv_out.b"], out_f32=o4)
        out = torch.empty((n, self.out_channels, h, w), dtype=F32, device=self.device)
        ops.nhwc_to_nchw_f32(o4, out)
        return out


class VaeEncoderEngine(_VaeEngineBase):
    """x (n,3,H,W) in [-1,1] -> mean latent * scale_factor (n,4,H/8,W/8).  `quant_conv` (1x1) is folded into
    `encoder.conv_out` at pack time in fp64 and only the mean half of the moments is produced."""

    PREFIXES = ("encoder.", "quant_conv.")
    CONV_IN, CONV_OUT = "encoder.conv_in", "encoder.conv_out"

    def _pack_ends(self, sd, conv3):
        L = self.latent
        wo, bo = sd["encoder.conv_out.weight"].double(), sd["encoder.conv_out.bias"].double()
        wq = sd["quant_conv.weight"].double().reshape(2 * L, 2 * L)[:L]  # mean rows only
        w = (wq @ wo.reshape(2 * L, -1)).reshape(L, *wo.shape[1:])
        b = wq @ bo + sd["quant_conv.bias"].double()[:L]
        self.W["enc_out.w"], self.W["enc_out.b"] = pack_conv3x3(w.float()), b.float().contiguous()

    @torch.no_grad()
    def encode(self, x: torch.Tensor, scale_factor: float) -> torch.Tensor:
        require_cuda(x)
        W = self.W
        x = x.to(F32).contiguous()
        n, cx, h, w = x.shape
        nd = len(self.block_out) - 1
        if h % (1 << nd) or w % (1 << nd):
            raise ValueError(f"VAE encode needs H and W divisible by {1 << nd} (got {h}x{w})")
        self.gn_ws = self._buf("gn_ws", (n * ops.GN_WORKSPACE_SLABS * 32 * 2,), F32)
        self._stats = {}
        one = torch.ones((n,), dtype=F32, device=self.device)
        x16 = self._buf("v_x16", (n, h * w, CIN_PAD), F16)
        ops.nchw_to_nhwc_f16(x, None, x16, scale=one)  # channels-last, 3 -> 64 zero-padded channels
        c0 = self.block_out[0]
        cur = self._buf("out:enc_in", (n, h * w, c0), F32)
        st = self._stats_buf("out:enc_in", n * h * w, h * w, c0)
        ops.conv3x3(x16.view(n, h, w, CIN_PAD), W["encoder.conv_in.w"], bias=W["encoder.conv_in.b"], out_f32=cur, ch_stats=st)
        self._produced(cur, st)
        cin = c0
        for i, cout in enumerate(self.block_out):
            d16 = self._buf("v_dn16", (n, h, w, cout), F16) if i != nd else None
            for j in range(2):
                cur = self._resnet(f"encoder.down_blocks.{i}.resnets.{j}", cur, n, h, w, cin if j == 0 else cout, cout,
                                   f16_out=d16 if j == 1 else None)
            cin = cout
            if i != nd:
                p = f"encoder.down_blocks.{i}.downsamplers.0.conv"
                h, w = h // 2, w // 2
                cur = self._buf("out:" + p, (n, h * w, cout), F32)
                st = self._stats_buf("out:" + p, n * h * w, h * w, cout)
                ops.conv3x3(d16, W[p + ".w"], stride=2, pad_br_only=True, bias=W[p + ".b"], out_f32=cur, ch_stats=st)
                self._produced(cur, st)
        top = self.block_out[-1]
        cur = self._resnet("encoder.mid_block.resnets.0", cur, n, h, w, top, top)
        cur = self._attention("encoder.mid_block.attentions.0", cur, n, h, w, top)
        cur = self._resnet("encoder.mid_block.resnets.1", cur, n, h, w, top, top)
        g16 = self._buf("gn16", (n, h * w, top), F16)
        ops.groupnorm(cur, None, W["encoder.conv_norm_out.g"], W["encoder.conv_norm_out.b"], g16, self.gn_ws,
                      eps=1e-6, silu=True, stats1=self._stats.get(cur.data_ptr()))
        o4 = self._buf("v_m4", (n, h * w, self.latent), F32)
        ops.conv3x3(g16.view(n, h, w, top), W["enc_out.w"], bias=W["enc_out.b"], out_f32=o4)
        out = torch.empty((n, self.latent, h, w), dtype=F32, device=self.device)
        ops.nhwc_to_nchw_f32(o4, out)
        ops.scale_rows(out, torch.full((n,), float(scale_factor), dtype=F32, device=self.device), out)
        return out
