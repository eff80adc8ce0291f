"""HIP execution of the CLIP ViT-H-14 image tower (reference seva/modules/conditioner.py:36-39 ->
open_clip `VisionTransformer.forward`).  Same rules as the UNet engine: fp32 residual stream [n*257][1280], f16 GEMM
operands written by the producing LayerNorm / attention / activation kernel, every matrix product on the MFMA GEMM.

  preprocess      seva_clip_preprocess_f16: resize + normalise, emitted directly as the [n*256][640] patch matrix
  patch embed     GEMM with conv1.weight.reshape(1280, 588) (K zero-padded to 640) + positional embedding as `residual`
  ln_pre          seva_layernorm_f32 (its output IS the residual stream)
  block x32       LN -> f16 | QKV GEMM (+bias) | seva_attention_small_f16 (16 heads x d 80, L 257) | out-proj GEMM
                  (+bias +residual) | LN -> f16 | fc1 GEMM with the erf-GELU evaluated by the GEGLU epilogue | fc2 GEMM
  pooled          ln_post on the class tokens -> f16 -> GEMM with proj^T

GELU without a new epilogue: the GEMM's GEGLU epilogue computes value * gelu(gate); fc1's rows are packed as the GATE
rows and the value rows are zero weights with bias 1, so the output is exactly 1 * gelu(fc1(x)).
"""

from __future__ import annotations

import math

import torch

from . import ops
from ._engine import _Arena, interleave_geglu
from ._native import SevaNativeError, require_cuda

F16, F32 = torch.float16, torch.float32


class ClipEngine:
    @staticmethod
    def _resolve_device(weights) -> torch.device:
        params = list(weights.parameters())
        if not params or params[0].device.type != "cuda":
            raise SevaNativeError("CLIPConditioner runs only on an AMD GPU (no CPU fallback): call .to('cuda')")
        from . import _native

        _native.load()
        return params[0].device

    def __init__(self, weights, p, mean, std):
        self.device = self._resolve_device(weights)
        self.p = p
        self.mean, self.std = [float(v) for v in mean], [float(v) for v in std]
        self.arena = _Arena(self.device)
        dev = self.device
        sd = {k: v.detach().to(dev) for k, v in weights.state_dict().items()}
        W = {}
        w = p.width
        kp = 3 * p.patch_size * p.patch_size
        self.kpad = 64 * ((kp + 63) // 64)
        wc = torch.zeros((w, self.kpad), dtype=F16, device=dev)
        wc[:, :kp] = sd["visual.conv1.weight"].reshape(w, kp).to(F16)
        W["patch.w"] = wc
        W["cls"] = sd["visual.class_embedding"].float()
        W["pos"] = sd["visual.positional_embedding"].float().contiguous()
        for n in ("ln_pre", "ln_post"):
            W[n + ".g"], W[n + ".b"] = sd[f"visual.{n}.weight"].float().contiguous(), sd[f"visual.{n}.bias"].float().contiguous()
        W["proj.w"] = sd["visual.proj"].float().T.to(F16).contiguous()  # [embed, width]
        for i in range(p.layers):
            b, o = f"visual.transformer.resblocks.{i}", f"blk{i}"
            for n in ("ln_1", "ln_2"):
                W[f"{o}.{n}.g"], W[f"{o}.{n}.b"] = sd[f"{b}.{n}.weight"].float().contiguous(), sd[f"{b}.{n}.bias"].float().contiguous()
            W[o + ".qkv.w"], W[o + ".qkv.b"] = sd[b + ".attn.in_proj_weight"].to(F16).contiguous(), sd[b + ".attn.in_proj_bias"].float().contiguous()
            W[o + ".out.w"], W[o + ".out.b"] = sd[b + ".attn.out_proj.weight"].to(F16).contiguous(), sd[b + ".attn.out_proj.bias"].float().contiguous()
            fc_w, fc_b = sd[b + ".mlp.c_fc.weight"].float(), sd[b + ".mlp.c_fc.bias"].float()
            # GEGLU packing [value rows ; gate rows]: value = 0 * x + 1, gate = fc1 -> epilogue gives 1 * gelu(fc1(x))
            vg_w = torch.cat([torch.zeros_like(fc_w), fc_w], 0).to(F16)
            vg_b = torch.cat([torch.ones_like(fc_b), fc_b], 0)
            W[o + ".fc1.w"], W[o + ".fc1.b"] = interleave_geglu(vg_w, vg_b)
            W[o + ".fc2.w"], W[o + ".fc2.b"] = sd[b + ".mlp.c_proj.weight"].to(F16).contiguous(), sd[b + ".mlp.c_proj.bias"].float().contiguous()
        self.W = W

    def _buf(self, name, shape, dtype):
        return self.arena.get(name, shape, dtype)

    def _patches(self, x):
        p = self.p
        n = x.shape[0]
        g = p.image_size // p.patch_size
        pm = self._buf("patches", (n * g * g, self.kpad), F16)
        if self.kpad > 3 * p.patch_size ** 2:
            pm[:, 3 * p.patch_size ** 2:].zero_()  # K padding of the patch GEMM
        ops.clip_preprocess(x.to(F32).contiguous(), pm, self.mean, self.std, out_size=p.image_size, patch=p.patch_size)
        return pm, g

    @torch.no_grad()
    def preprocess_image(self, x):
        require_cuda(x)
        p = self.p
        pm, g = self._patches(x)
        P = p.patch_size
        img = pm[:, : 3 * P * P].float().view(x.shape[0], g, g, 3, P, P).permute(0, 3, 1, 4, 2, 5)
        return img.reshape(x.shape[0], 3, p.image_size, p.image_size)

    @torch.no_grad()
    def encode(self, x):
        """x: (n,3,H,W) fp32 in [-1,1] on the GPU -> (n, embed_dim) fp32."""
        require_cuda(x)
        p, W = self.p, self.W
        n, w, heads, d = x.shape[0], p.width, p.heads, p.head_width
        pm, g = self._patches(x)
        L = g * g + 1
        rows = n * L
        tok = self._buf("tok", (n, L, w), F32)
        tok[:, 0] = W["cls"] + W["pos"][0]  # class token rows: 1280 constants per image (index plumbing)
        for i in range(n):  # patch rows of image i = patches @ Wc^T + positional embedding (as the GEMM's residual)
            ops.gemm(pm[i * g * g:(i + 1) * g * g], W["patch.w"], residual=W["pos"][1:], out_f32=tok[i, 1:])
        cur = self._buf("stream_a", (rows, w), F32)
        ops.layernorm(tok.view(rows, w), W["ln_pre.g"], W["ln_pre.b"], cur)
        a16 = self._buf("ln16", (rows, w), F16)
        qkv = self._buf("qkv", (rows, 3 * w), F16)
        att = self._buf("att", (rows, w), F16)
        hid = self._buf("hid", (rows, int(w * p.mlp_ratio)), F16)
        nxt = self._buf("stream_b", (rows, w), F32)
        scale = 1.0 / math.sqrt(d)
        for i in range(p.layers):
            o = f"blk{i}"
            ops.layernorm(cur, W[o + ".ln_1.g"], W[o + ".ln_1.b"], a16)
            ops.gemm(a16, W[o + ".qkv.w"], bias=W[o + ".qkv.b"], out_f16=qkv)
            ops.attention_small(qkv[:, :w], qkv[:, w:2 * w], qkv[:, 2 * w:], att, batch=n, heads=heads, L=L, head_dim=d,
                                q_strides=(L * 3 * w, 3 * w), k_strides=(L * 3 * w, 3 * w), o_strides=(L * w, w), scale=scale)
            ops.gemm(att, W[o + ".out.w"], bias=W[o + ".out.b"], residual=cur, out_f32=nxt)
            ops.layernorm(nxt, W[o + ".ln_2.g"], W[o + ".ln_2.b"], a16)
            ops.gemm(a16, W[o + ".fc1.w"], bias=W[o + ".fc1.b"], out_f16=hid, geglu=True)
            ops.gemm(hid, W[o + ".fc2.w"], bias=W[o + ".fc2.b"], residual=nxt, out_f32=cur)
        cls32 = cur.view(n, L, w)[:, 0].contiguous()  # n class-token rows (index plumbing)
        m = max(n, 1)
        c16 = self._buf("cls16", (m, w), F16)
        ops.layernorm(cls32, W["ln_post.g"], W["ln_post.b"], c16)
        out = torch.empty((n, p.embed_dim), dtype=F32, device=self.device)
        ops.gemm(c16, W["proj.w"], out_f32=out)
        return out
