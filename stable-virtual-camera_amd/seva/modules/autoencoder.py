"""`seva.modules.autoencoder` -- drop-in for the reference wrapper (seva/modules/autoencoder.py).

The reference delegates to `diffusers.AutoencoderKL.from_pretrained("stabilityai/stable-diffusion-2-1-base",
subfolder="vae")` (autoencoder.py:12-17).  Neither diffusers nor the weights are available offline, so this
module owns (a) a parameter holder with diffusers' key names (`encoder.*`, `quant_conv.*`, `post_quant_conv.*`,
`decoder.*`), so a real `diffusion_pytorch_model.safetensors` loads STRICTLY (`load_vae_state_dict`: every key
must match after the deprecated mid-block attention names `query/key/value/proj_attn` -- which diffusers itself remaps at
load time -- are translated to `to_q/to_k/to_v/to_out.0`), and (b) the
encode and decode paths on the HIP kernels (`seva/_vae_engine.py`).  Parity with diffusers is UNPINNED (no fixture can be made here);
tests compare against our own restatement of the published topology.

Kept API: `AutoEncoder(chunk_size=None)`, `.encode(x, chunk_size)`, `.decode(z, chunk_size)`, `.to(device)`,
`scale_factor`, `downsample`.  `encode` returns the mean of the latent distribution times 0.18215,
as the reference does (autoencoder.py:21-25).
"""

from __future__ import annotations

import os
import re
import warnings

import torch
from torch import nn

BLOCK_OUT = (128, 256, 512, 512)
LAYERS_PER_BLOCK = 2


class _Holder(nn.Module):
    def put(self, path: str, module: nn.Module) -> None:
        head, _, rest = path.partition(".")
        if not rest:
            self.add_module(head, module)
            return
        if head not in self._modules:
            self.add_module(head, _Holder())
        self._modules[head].put(rest, module)

    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("parameter holder; compute runs in the HIP engine")


class VaeDecoderWeights(_Holder):
    """Decoder half of the SD-2.1 AutoencoderKL (published config), diffusers key names."""

    def __init__(self, block_out=BLOCK_OUT, latent_channels: int = 4, out_channels: int = 3):
        super().__init__()
        self.block_out, self.latent_channels, self.out_channels = tuple(block_out), latent_channels, out_channels
        top = block_out[-1]

        def resnet(p, cin, cout):
            self.put(p + ".norm1", nn.GroupNorm(32, cin, eps=1e-6))
            self.put(p + ".conv1", nn.Conv2d(cin, cout, 3, padding=1))
            self.put(p + ".norm2", nn.GroupNorm(32, cout, eps=1e-6))
            self.put(p + ".conv2", nn.Conv2d(cout, cout, 3, padding=1))
            if cin != cout:
                self.put(p + ".conv_shortcut", nn.Conv2d(cin, cout, 1))

        self.put("post_quant_conv", nn.Conv2d(latent_channels, latent_channels, 1))
        self.put("decoder.conv_in", nn.Conv2d(latent_channels, top, 3, padding=1))
        resnet("decoder.mid_block.resnets.0", top, top)
        a = "decoder.mid_block.attentions.0"
        self.put(a + ".group_norm", nn.GroupNorm(32, top, eps=1e-6))
        for n in ("to_q", "to_k", "to_v"):
            self.put(f"{a}.{n}", nn.Linear(top, top))
        self.put(a + ".to_out.0", nn.Linear(top, top))
        resnet("decoder.mid_block.resnets.1", top, top)
        rev = list(reversed(block_out))
        cin = rev[0]
        for i, cout in enumerate(rev):
            for j in range(LAYERS_PER_BLOCK + 1):
                resnet(f"decoder.up_blocks.{i}.resnets.{j}", cin if j == 0 else cout, cout)
            cin = cout
            if i != len(rev) - 1:
                self.put(f"decoder.up_blocks.{i}.upsamplers.0.conv", nn.Conv2d(cout, cout, 3, padding=1))
        self.put("decoder.conv_norm_out", nn.GroupNorm(32, rev[-1], eps=1e-6))
        self.put("decoder.conv_out", nn.Conv2d(rev[-1], out_channels, 3, padding=1))


class VaeWeights(VaeDecoderWeights):
    """Encoder + decoder of the SD-2.1 AutoencoderKL (published config), diffusers key names."""

    def __init__(self, block_out=BLOCK_OUT, latent_channels: int = 4, out_channels: int = 3, in_channels: int = 3):
        super().__init__(block_out, latent_channels, out_channels)
        self.in_channels = in_channels

        def resnet(p, cin, cout):
            self.put(p + ".norm1", nn.GroupNorm(32, cin, eps=1e-6))
            self.put(p + ".conv1", nn.Conv2d(cin, cout, 3, padding=1))
            self.put(p + ".norm2", nn.GroupNorm(32, cout, eps=1e-6))
            self.put(p + ".conv2", nn.Conv2d(cout, cout, 3, padding=1))
            if cin != cout:
                self.put(p + ".conv_shortcut", nn.Conv2d(cin, cout, 1))

        self.put("encoder.conv_in", nn.Conv2d(in_channels, block_out[0], 3, padding=1))
        cin = block_out[0]
        for i, cout in enumerate(block_out):
            for j in range(LAYERS_PER_BLOCK):
                resnet(f"encoder.down_blocks.{i}.resnets.{j}", cin if j == 0 else cout, cout)
            cin = cout
            if i != len(block_out) - 1:
                self.put(f"encoder.down_blocks.{i}.downsamplers.0.conv", nn.Conv2d(cout, cout, 3, stride=2, padding=0))
        top = block_out[-1]
        resnet("encoder.mid_block.resnets.0", top, top)
        a = "encoder.mid_block.attentions.0"
        self.put(a + ".group_norm", nn.GroupNorm(32, top, eps=1e-6))
        for n in ("to_q", "to_k", "to_v"):
            self.put(f"{a}.{n}", nn.Linear(top, top))
        self.put(a + ".to_out.0", nn.Linear(top, top))
        resnet("encoder.mid_block.resnets.1", top, top)
        self.put("encoder.conv_norm_out", nn.GroupNorm(32, top, eps=1e-6))
        self.put("encoder.conv_out", nn.Conv2d(top, 2 * latent_channels, 3, padding=1))
        self.put("quant_conv", nn.Conv2d(2 * latent_channels, 2 * latent_channels, 1))


_LEGACY_ATTN = {"query": "to_q", "key": "to_k", "value": "to_v", "proj_attn": "to_out.0"}
_LEGACY_RE = re.compile(r"^((?:encoder|decoder)\.mid_block\.attentions\.0)\.(query|key|value|proj_attn)\.(weight|bias)$")


def remap_legacy_vae_keys(sd: dict) -> dict:
    """Deprecated attention names of older AutoencoderKL checkpoints -> current ones (diffusers does this in
    `_convert_deprecated_attention_blocks` when it loads such a file).  1x1-conv shaped projection weights
    (C, C, 1, 1) become Linear weights (C, C)."""
    out = {}
    for k, v in sd.items():
        m = _LEGACY_RE.match(k)
        if m:
            k = f"{m.group(1)}.{_LEGACY_ATTN[m.group(2)]}.{m.group(3)}"
        if ".mid_block.attentions.0.to_" in k and k.endswith(".weight") and v.ndim == 4:
            v = v.reshape(v.shape[0], v.shape[1])
        if k in out:
            raise RuntimeError(f"VAE checkpoint holds both the deprecated and the current name of {k}")
        out[k] = v
    return out


def load_vae_state_dict(module: nn.Module, sd: dict) -> None:
    """Strict load: a key that is missing or unexpected after the legacy-name translation is an error (a silent
    `strict=False` would leave e.g. the mid-block attention on its random initialisation)."""
    sd = remap_legacy_vae_keys(sd)
    missing, unexpected = module.load_state_dict(sd, strict=False)
    if missing or unexpected:
        raise RuntimeError(
            f"VAE checkpoint does not match the SD-2.1 AutoencoderKL layout: {len(missing)} missing "
            f"(e.g. {list(missing)[:3]}), {len(unexpected)} unexpected (e.g. {list(unexpected)[:3]})")


class AutoEncoder(nn.Module):
    scale_factor: float = 0.18215
    downsample: int = 8

    def __init__(self, chunk_size: int | None = None, *, random_init: bool | None = None):
        """`SEVA_VAE_PATH` = local `diffusion_pytorch_model.safetensors` of stabilityai/stable-diffusion-2-1-base/vae
        (the reference downloads it, autoencoder.py:12-17; there is no network here).  Without it the module refuses to
        run on random weights unless that is asked for explicitly (`random_init=True` or `SEVA_VAE_RANDOM_INIT=1`:
        benchmarks and tests with synthetic weights)."""
        super().__init__()
        self.module = VaeWeights()
        path = os.environ.get("SEVA_VAE_PATH")  # local diffusers VAE safetensors, if the user has one
        if random_init is None:
            random_init = os.environ.get("SEVA_VAE_RANDOM_INIT", "0") == "1"
        self.random_init = False
        if path:
            import safetensors.torch

            load_vae_state_dict(self.module, safetensors.torch.load_file(path))
        elif random_init:
            self.random_init = True
            warnings.warn("seva AutoEncoder: running on RANDOM-INIT VAE weights (no SEVA_VAE_PATH): outputs are "
                          "meaningless images / latents; fine for benchmarks and synthetic-weight tests only.",
                          RuntimeWarning, stacklevel=2)
        else:
            raise RuntimeError(
                "seva AutoEncoder: no VAE weights. Set SEVA_VAE_PATH to the SD-2.1-base VAE safetensors file "
                "(stabilityai/stable-diffusion-2-1-base, subfolder vae), or pass random_init=True / set "
                "SEVA_VAE_RANDOM_INIT=1 to run on random weights on purpose.")
        self.module.eval().requires_grad_(False)
        self.chunk_size = chunk_size
        self._engine = None
        self._enc_engine = None

    def _apply(self, fn, *a, **k):
        self._engine = self._enc_engine = None
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._engine = self._enc_engine = None
        return super().load_state_dict(*a, **k)

    def encoder_engine(self):
        if self._enc_engine is None:
            from .._vae_engine import VaeEncoderEngine

            self._enc_engine = VaeEncoderEngine(self.module)
        return self._enc_engine

    def engine(self):
        if self._engine is None:
            from .._vae_engine import VaeDecoderEngine

            self._engine = VaeDecoderEngine(self.module)
        return self._engine

    def _encode(self, x: torch.Tensor) -> torch.Tensor:
        return self.encoder_engine().encode(x, self.scale_factor)

    ARENA_BYTES_PER_PIXEL = 4.0e9 / (576 * 576)  # decoder arena per frame, measured at 576 x 576 (profiles/r02_kvae.log)

    def _frames_per_pass(self, chunk_size: int | None, pixels: int = 576 * 576, device=None) -> int | None:
        """`chunk_size` is the reference's memory knob (demo.py: AutoEncoder(chunk_size=1), decoding_t=1: 21 sequential
        single-frame passes per window).  Every kernel of the VAE engines is sample-independent and bitwise batch-invariant
        (GroupNorm slab counts depend on the image only; tested), so the frames-per-pass actually executed is a pure
        performance choice: at 576x576 one decode pass costs 9.5 ms for 1 frame but 5.6 ms/frame for 7 -- at about 4 GB of
        arena PER FRAME.  The caller's chunk_size is therefore raised (results identical bit for bit) only
          * to SEVA_VAE_FRAMES_PER_PASS when the user sets it, or
          * up to 7 frames while the arena of that many frames fits in HALF of the card's currently free memory
            (the UNet and CLIP engines share the card; an explicit chunk_size=1 on a nearly full card stays 1)."""
        chunk_size = chunk_size or self.chunk_size
        if chunk_size is None:
            return None
        env = os.environ.get("SEVA_VAE_FRAMES_PER_PASS")
        if env is not None:
            return max(int(chunk_size), int(env))
        want = 7
        try:
            free, _ = torch.cuda.mem_get_info(device)
            fit = int(0.5 * free / (self.ARENA_BYTES_PER_PIXEL * max(int(pixels), 1)))
        except Exception:  # no device query available: honour the caller's value
            fit = 0
        return max(int(chunk_size), min(want, fit))

    def encode(self, x: torch.Tensor, chunk_size: int | None = None) -> torch.Tensor:
        chunk_size = self._frames_per_pass(chunk_size, x.shape[-2] * x.shape[-1], x.device)
        if chunk_size is not None:
            return torch.cat([self._encode(xc) for xc in x.split(chunk_size)], dim=0)
        return self._encode(x)

    def _decode(self, z: torch.Tensor) -> torch.Tensor:
        return self.engine().decode(z, self.scale_factor)

    def decode(self, z: torch.Tensor, chunk_size: int | None = None) -> torch.Tensor:
        chunk_size = self._frames_per_pass(chunk_size, 64 * z.shape[-2] * z.shape[-1], z.device)
        if chunk_size is not None:
            return torch.cat([self._decode(zc) for zc in z.split(chunk_size)], dim=0)
        return self._decode(z)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.decode(self.encode(x))
