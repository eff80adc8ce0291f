"""CPU oracle for the SD-2.1 VAE (decoder and encoder) -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

**Parity unpinned.**  The reference delegates decoding to the third-party `diffusers` package
(`/root/reference/seva/modules/autoencoder.py:2,12-17,38`; `pyproject.toml:22`, version unpinned),
which is not installed here and whose weights need the network.  This file restates the PUBLISHED
topology of `AutoencoderKL` for `stabilityai/stable-diffusion-2-1-base/vae`
(block_out_channels 128/256/512/512, layers_per_block 2, norm_num_groups 32, latent_channels 4,
SiLU, single-head mid-block attention, GroupNorm eps 1e-6) with diffusers' state_dict key names.
No fixture of the real library exists, so tests can only check the HIP decoder against THIS
restatement (self-consistency), not against diffusers.
"""

from __future__ import annotations

import torch
import torch.nn.functional as F

BLOCK_OUT = (128, 256, 512, 512)
LAYERS_PER_BLOCK = 2
SCALE_FACTOR = 0.18215  # reference autoencoder.py:7


def decoder_shapes(block_out=BLOCK_OUT, latent=4, out_ch=3) -> dict[str, tuple]:
    """state_dict keys/shapes of the decoder half (+post_quant_conv) in diffusers naming."""
    s: dict[str, tuple] = {}

    def conv(p, cin, cout, k):
        s[p + ".weight"], s[p + ".bias"] = (cout, cin, k, k), (cout,)

    def norm(p, c):
        s[p + ".weight"], s[p + ".bias"] = (c,), (c,)

    def lin(p, cin, cout):
        s[p + ".weight"], s[p + ".bias"] = (cout, cin), (cout,)

    def resnet(p, cin, cout):
        norm(p + ".norm1", cin); conv(p + ".conv1", cin, cout, 3)
        norm(p + ".norm2", cout); conv(p + ".conv2", cout, cout, 3)
        if cin != cout:
            conv(p + ".conv_shortcut", cin, cout, 1)

    top = block_out[-1]
    conv("post_quant_conv", latent, latent, 1)
    conv("decoder.conv_in", latent, top, 3)
    resnet("decoder.mid_block.resnets.0", top, top)
    a = "decoder.mid_block.attentions.0"
    norm(a + ".group_norm", top)
    for n in ("to_q", "to_k", "to_v"):
        lin(f"{a}.{n}", top, top)
    lin(a + ".to_out.0", top, top)
    resnet("decoder.mid_block.resnets.1", top, top)
    rev = list(reversed(block_out))
    cin = rev[0]
    for i, cout in enumerate(rev):
        for j in range(LAYERS_PER_BLOCK + 1):
            resnet(f"decoder.up_blocks.{i}.resnets.{j}", cin if j == 0 else cout, cout)
        cin = cout
        if i != len(rev) - 1:
            conv(f"decoder.up_blocks.{i}.upsamplers.0.conv", cout, cout, 3)
    norm("decoder.conv_norm_out", rev[-1])
    conv("decoder.conv_out", rev[-1], out_ch, 3)
    return s


def _gn(sd, p, x):
    return F.group_norm(x, 32, sd[p + ".weight"], sd[p + ".bias"], 1e-6)


def _conv(sd, p, x, pad):
    return F.conv2d(x, sd[p + ".weight"], sd[p + ".bias"], padding=pad)


def _resnet(sd, p, x):
    h = _conv(sd, p + ".conv1", F.silu(_gn(sd, p + ".norm1", x)), 1)
    h = _conv(sd, p + ".conv2", F.silu(_gn(sd, p + ".norm2", h)), 1)
    if p + ".conv_shortcut.weight" in sd:
        x = _conv(sd, p + ".conv_shortcut", x, 0)
    return x + h


def _attn(sd, p, x):
    n, c, h, w = x.shape
    t = _gn(sd, p + ".group_norm", x).reshape(n, c, h * w).transpose(1, 2)
    q = F.linear(t, sd[p + ".to_q.weight"], sd[p + ".to_q.bias"])
    k = F.linear(t, sd[p + ".to_k.weight"], sd[p + ".to_k.bias"])
    v = F.linear(t, sd[p + ".to_v.weight"], sd[p + ".to_v.bias"])
    att = torch.softmax(q @ k.transpose(1, 2) / (c**0.5), -1)
    o = F.linear(att @ v, sd[p + ".to_out.0.weight"], sd[p + ".to_out.0.bias"])
    return x + o.transpose(1, 2).reshape(n, c, h, w)


def vae_decode(sd: dict, z: torch.Tensor) -> torch.Tensor:
    """AutoEncoder._decode (reference autoencoder.py:37-38): AutoencoderKL.decode(z / 0.18215).sample."""
    x = _conv(sd, "post_quant_conv", z / SCALE_FACTOR, 0)
    x = _conv(sd, "decoder.conv_in", x, 1)
    x = _resnet(sd, "decoder.mid_block.resnets.0", x)
    x = _attn(sd, "decoder.mid_block.attentions.0", x)
    x = _resnet(sd, "decoder.mid_block.resnets.1", x)
    i = 0
    while f"decoder.up_blocks.{i}.resnets.0.norm1.weight" in sd:
        j = 0
        while f"decoder.up_blocks.{i}.resnets.{j}.norm1.weight" in sd:
            x = _resnet(sd, f"decoder.up_blocks.{i}.resnets.{j}", x)
            j += 1
        up = f"decoder.up_blocks.{i}.upsamplers.0.conv"
        if up + ".weight" in sd:
            x = _conv(sd, up, F.interpolate(x, scale_factor=2.0, mode="nearest"), 1)
        i += 1
    x = F.silu(_gn(sd, "decoder.conv_norm_out", x))
    return _conv(sd, "decoder.conv_out", x, 1)


# ---- encoder (reference autoencoder.py:21-35 -> AutoencoderKL.encode(x).latent_dist.mean * 0.18215) ----
def encoder_shapes(block_out=BLOCK_OUT, latent=4, in_ch=3) -> dict[str, tuple]:
    """state_dict keys/shapes of the encoder half (+quant_conv) in diffusers naming (published SD-2.1 VAE config:
    down blocks of LAYERS_PER_BLOCK resnets, a stride-2 conv after all but the last, mid block with one attention,
    GN+SiLU+conv to 2*latent moments)."""
    s: dict[str, tuple] = {}

    def conv(p, cin, cout, k):
        s[p + ".weight"], s[p + ".bias"] = (cout, cin, k, k), (cout,)

    def norm(p, c):
        s[p + ".weight"], s[p + ".bias"] = (c,), (c,)

    def lin(p, cin, cout):
        s[p + ".weight"], s[p + ".bias"] = (cout, cin), (cout,)

    def resnet(p, cin, cout):
        norm(p + ".norm1", cin); conv(p + ".conv1", cin, cout, 3)
        norm(p + ".norm2", cout); conv(p + ".conv2", cout, cout, 3)
        if cin != cout:
            conv(p + ".conv_shortcut", cin, cout, 1)

    conv("encoder.conv_in", in_ch, block_out[0], 3)
    cin = block_out[0]
    for i, cout in enumerate(block_out):
        for j in range(LAYERS_PER_BLOCK):
            resnet(f"encoder.down_blocks.{i}.resnets.{j}", cin if j == 0 else cout, cout)
        cin = cout
        if i != len(block_out) - 1:
            conv(f"encoder.down_blocks.{i}.downsamplers.0.conv", cout, cout, 3)
    top = block_out[-1]
    resnet("encoder.mid_block.resnets.0", top, top)
    a = "encoder.mid_block.attentions.0"
    norm(a + ".group_norm", top)
    for n in ("to_q", "to_k", "to_v"):
        lin(f"{a}.{n}", top, top)
    lin(a + ".to_out.0", top, top)
    resnet("encoder.mid_block.resnets.1", top, top)
    norm("encoder.conv_norm_out", top)
    conv("encoder.conv_out", top, 2 * latent, 3)
    conv("quant_conv", 2 * latent, 2 * latent, 1)
    return s


def vae_encode(sd: dict, x: torch.Tensor) -> torch.Tensor:
    """AutoEncoder._encode (reference autoencoder.py:21-25): mean of the diagonal Gaussian, times 0.18215.
    Downsample2D pads (0,1,0,1) and convolves with stride 2, padding 0."""
    latent = sd["quant_conv.weight"].shape[0] // 2
    h = _conv(sd, "encoder.conv_in", x, 1)
    i = 0
    while f"encoder.down_blocks.{i}.resnets.0.norm1.weight" in sd:
        j = 0
        while f"encoder.down_blocks.{i}.resnets.{j}.norm1.weight" in sd:
            h = _resnet(sd, f"encoder.down_blocks.{i}.resnets.{j}", h)
            j += 1
        dn = f"encoder.down_blocks.{i}.downsamplers.0.conv"
        if dn + ".weight" in sd:
            h = F.conv2d(F.pad(h, (0, 1, 0, 1)), sd[dn + ".weight"], sd[dn + ".bias"], stride=2)
        i += 1
    h = _resnet(sd, "encoder.mid_block.resnets.0", h)
    h = _attn(sd, "encoder.mid_block.attentions.0", h)
    h = _resnet(sd, "encoder.mid_block.resnets.1", h)
    h = _conv(sd, "encoder.conv_out", F.silu(_gn(sd, "encoder.conv_norm_out", h)), 1)
    moments = _conv(sd, "quant_conv", h, 0)
    return moments[:, :latent] * SCALE_FACTOR
