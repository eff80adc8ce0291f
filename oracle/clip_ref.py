"""TEST INFRASTRUCTURE -- CPU fp32 restatement of the CLIP image conditioner.

**Parity unpinned.**  The reference (seva/modules/conditioner.py:7-39) delegates to two third-party packages that are
absent from /root/reference and from this image: `open_clip` (`create_model_and_transforms("ViT-H-14",
pretrained="laion2b_s32b_b79k")`, `.encode_image`) and `kornia` (`geometry.resize(..., bicubic, align_corners=True,
antialias=True)`, `enhance.normalize`); pyproject.toml lists both unpinned.  The reference holds no fixture for them
and the weights need the network, so this file restates their published algorithms and the HIP path is checked
against it only (self-consistency), exactly like the VAE (oracle/vae_ref.py):

  * open_clip `VisionTransformer.forward` for the ViT-H-14 config (image 224, patch 14, width 1280, 32 layers, 16 heads
    = head width 80, MLP 5120 with exact-erf GELU, LayerNorm eps 1e-5, class token + learned positional embedding,
    `ln_pre`, pooled = `ln_post(x[:, 0]) @ proj`), key names of the `visual.` sub-module of the open_clip state_dict;
  * kornia `resize(antialias=True)`: when down-scaling, Gaussian blur with sigma = max((factor - 1) / 2, 0.001), kernel
    size int(max(4 sigma, 3)) made odd, 'reflect' border; then `F.interpolate(mode="bicubic", align_corners=True)`.

Only tests/ import this module.
"""

from __future__ import annotations

import math

import torch
import torch.nn.functional as F

MEAN = (0.48145466, 0.4578275, 0.40821073)  # conditioner.py:17-19
STD = (0.26862954, 0.26130258, 0.27577711)  # conditioner.py:20-22


def vit_shapes(width=1280, layers=32, patch=14, image=224, mlp=5120, embed=1024) -> dict:
    """{key: shape} of open_clip's `visual.*` state_dict entries for a ViT with these hyper-parameters."""
    g = image // patch
    s = {
        "visual.class_embedding": (width,),
        "visual.positional_embedding": (g * g + 1, width),
        "visual.proj": (width, embed),
        "visual.conv1.weight": (width, 3, patch, patch),
        "visual.ln_pre.weight": (width,), "visual.ln_pre.bias": (width,),
        "visual.ln_post.weight": (width,), "visual.ln_post.bias": (width,),
    }
    for i in range(layers):
        b = f"visual.transformer.resblocks.{i}"
        s.update({
            f"{b}.ln_1.weight": (width,), f"{b}.ln_1.bias": (width,),
            f"{b}.attn.in_proj_weight": (3 * width, width), f"{b}.attn.in_proj_bias": (3 * width,),
            f"{b}.attn.out_proj.weight": (width, width), f"{b}.attn.out_proj.bias": (width,),
            f"{b}.ln_2.weight": (width,), f"{b}.ln_2.bias": (width,),
            f"{b}.mlp.c_fc.weight": (mlp, width), f"{b}.mlp.c_fc.bias": (mlp,),
            f"{b}.mlp.c_proj.weight": (width, mlp), f"{b}.mlp.c_proj.bias": (width,),
        })
    return s


def gaussian_kernel1d(ks: int, sigma: float) -> torch.Tensor:
    x = torch.arange(ks, dtype=torch.float32) - ks // 2
    g = torch.exp(-x * x / (2.0 * sigma * sigma))
    return g / g.sum()


def preprocess(x: torch.Tensor, size: int = 224) -> torch.Tensor:
    """conditioner.py:24-34; x (n,3,H,W) in [-1,1] -> (n,3,size,size) CLIP-normalised."""
    n, c, H, W = x.shape
    fy, fx = H / size, W / size
    if max(fy, fx) > 1.0:
        sy, sx = max((fy - 1.0) / 2.0, 0.001), max((fx - 1.0) / 2.0, 0.001)
        ky, kx = int(max(4.0 * sy, 3)), int(max(4.0 * sx, 3))
        ky, kx = ky + (ky % 2 == 0), kx + (kx % 2 == 0)
        gy, gx = gaussian_kernel1d(ky, sy), gaussian_kernel1d(kx, sx)
        xp = F.pad(x, (kx // 2, kx // 2, ky // 2, ky // 2), mode="reflect")
        k2 = (gy[:, None] * gx[None, :])[None, None].repeat(c, 1, 1, 1)
        x = F.conv2d(xp, k2, groups=c)
    x = F.interpolate(x, size=(size, size), mode="bicubic", align_corners=True)
    x = (x + 1.0) / 2.0
    mean, std = x.new_tensor(MEAN)[None, :, None, None], x.new_tensor(STD)[None, :, None, None]
    return (x - mean) / std


def encode_image(sd: dict, x: torch.Tensor, heads: int, patch: int = 14) -> torch.Tensor:
    """open_clip VisionTransformer.forward (pool 'tok', final LayerNorm then projection); x already pre-processed."""
    w = sd["visual.conv1.weight"]
    width = w.shape[0]
    t = F.conv2d(x, w, None, stride=patch)                      # (n, width, g, g)
    t = t.reshape(t.shape[0], width, -1).permute(0, 2, 1)       # (n, g*g, width)
    cls = sd["visual.class_embedding"].to(t.dtype)[None, None].expand(t.shape[0], 1, width)
    t = torch.cat([cls, t], 1) + sd["visual.positional_embedding"]
    t = F.layer_norm(t, (width,), sd["visual.ln_pre.weight"], sd["visual.ln_pre.bias"], 1e-5)
    L, d = t.shape[1], width // heads
    i = 0
    while f"visual.transformer.resblocks.{i}.ln_1.weight" in sd:
        b = f"visual.transformer.resblocks.{i}"
        y = F.layer_norm(t, (width,), sd[b + ".ln_1.weight"], sd[b + ".ln_1.bias"], 1e-5)
        qkv = y @ sd[b + ".attn.in_proj_weight"].T + sd[b + ".attn.in_proj_bias"]
        q, k, v = (z.reshape(-1, L, heads, d).transpose(1, 2) for z in qkv.chunk(3, -1))
        a = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(d), -1) @ v
        a = a.transpose(1, 2).reshape(-1, L, width)
        t = t + a @ sd[b + ".attn.out_proj.weight"].T + sd[b + ".attn.out_proj.bias"]
        y = F.layer_norm(t, (width,), sd[b + ".ln_2.weight"], sd[b + ".ln_2.bias"], 1e-5)
        y = F.gelu(y @ sd[b + ".mlp.c_fc.weight"].T + sd[b + ".mlp.c_fc.bias"])
        t = t + y @ sd[b + ".mlp.c_proj.weight"].T + sd[b + ".mlp.c_proj.bias"]
        i += 1
    pooled = F.layer_norm(t[:, 0], (width,), sd["visual.ln_post.weight"], sd["visual.ln_post.bias"], 1e-5)
    return pooled @ sd["visual.proj"]


def clip_conditioner(sd: dict, x: torch.Tensor, heads: int = 16, patch: int = 14, size: int = 224) -> torch.Tensor:
    """CLIPConditioner.forward, conditioner.py:36-39."""
    return encode_image(sd, preprocess(x, size), heads, patch)
