"""`seva.modules.conditioner` -- drop-in for the reference's CLIP image conditioner (seva/modules/conditioner.py:7-39).

The reference builds `open_clip.create_model_and_transforms("ViT-H-14", pretrained="laion2b_s32b_b79k")` and resizes
with kornia; neither package, nor the weights, exist offline.  This module owns (a) a parameter holder with open_clip's
key names for the vision tower (`module.visual.*`), so a real open_clip checkpoint loads strictly (`load_open_clip`:
text-tower keys are dropped knowingly, every vision key must match), and (b) the forward on HIP kernels
(`seva/_clip_engine.py`).  Parity with open_clip / kornia is UNPINNED (no fixture can be produced here): tests compare
against our own restatement of the published algorithms (oracle/clip_ref.py).

Kept API: `CLIPConditioner()`, `.forward(x)` with x (n,3,H,W) in [-1,1] -> (n,1024), `.preprocess(x)`, buffers
`mean` / `std`, `.to(device)`.  Called once per window on the input views (reference seva/eval.py:1248).
"""

from __future__ import annotations

import os
import warnings
from dataclasses import dataclass

import torch
from torch import nn

from .autoencoder import _Holder


@dataclass
class ViTParams:  # open_clip model config "ViT-H-14"
    image_size: int = 224
    patch_size: int = 14
    width: int = 1280
    layers: int = 32
    head_width: int = 80
    mlp_ratio: float = 4.0
    embed_dim: int = 1024

    @property
    def heads(self) -> int:
        return self.width // self.head_width


class VisionTowerWeights(_Holder):
    """open_clip `VisionTransformer` parameters under open_clip's names (state_dict prefix `visual.`)."""

    def __init__(self, p: ViTParams):
        super().__init__()
        self.p = p
        w, g = p.width, p.image_size // p.patch_size
        v = _Holder()
        self.add_module("visual", v)
        v.class_embedding = nn.Parameter(torch.zeros(w))
        v.positional_embedding = nn.Parameter(torch.zeros(g * g + 1, w))
        v.proj = nn.Parameter(torch.zeros(w, p.embed_dim))
        v.put("conv1", nn.Conv2d(3, w, p.patch_size, p.patch_size, bias=False))
        v.put("ln_pre", nn.LayerNorm(w))
        v.put("ln_post", nn.LayerNorm(w))
        mlp = int(w * p.mlp_ratio)
        for i in range(p.layers):
            b = f"transformer.resblocks.{i}"
            v.put(b + ".ln_1", nn.LayerNorm(w))
            attn = _Holder()
            attn.in_proj_weight = nn.Parameter(torch.zeros(3 * w, w))
            attn.in_proj_bias = nn.Parameter(torch.zeros(3 * w))
            attn.put("out_proj", nn.Linear(w, w))
            v.put(b + ".attn", attn)
            v.put(b + ".ln_2", nn.LayerNorm(w))
            v.put(b + ".mlp.c_fc", nn.Linear(w, mlp))
            v.put(b + ".mlp.c_proj", nn.Linear(mlp, w))


def load_open_clip(module: nn.Module, sd: dict) -> None:
    """Strict load of an open_clip CLIP state_dict: only `visual.*` is used (the text tower, `logit_scale`, ... are not part of
    `encode_image`); a missing or unexpected vision key is an error."""
    vis = {k: v for k, v in sd.items() if k.startswith("visual.")}
    missing, unexpected = module.load_state_dict(vis, strict=False)
    if missing or unexpected:
        raise RuntimeError(f"open_clip checkpoint does not match ViT-H-14: {len(missing)} missing (e.g. {list(missing)[:3]}), "
                           f"{len(unexpected)} unexpected (e.g. {list(unexpected)[:3]})")


class CLIPConditioner(nn.Module):
    mean: torch.Tensor
    std: torch.Tensor

    def __init__(self, params: ViTParams | None = None, *, random_init: bool | None = None):
        """`SEVA_CLIP_PATH` = local open_clip ViT-H-14 laion2b_s32b_b79k checkpoint (safetensors or torch file).  Without it
        the module refuses to run on random weights unless asked to (`random_init=True` / `SEVA_CLIP_RANDOM_INIT=1`)."""
        super().__init__()
        self.params = params or ViTParams()
        self.module = VisionTowerWeights(self.params)
        path = os.environ.get("SEVA_CLIP_PATH")
        if random_init is None:
            random_init = os.environ.get("SEVA_CLIP_RANDOM_INIT", "0") == "1"
        self.random_init = False
        if path:
            if path.endswith(".safetensors"):
                import safetensors.torch

                sd = safetensors.torch.load_file(path)
            else:
                sd = torch.load(path, map_location="cpu", weights_only=True)
            load_open_clip(self.module, sd)
        elif random_init:
            self.random_init = True
            warnings.warn("seva CLIPConditioner: RANDOM-INIT weights (no SEVA_CLIP_PATH): embeddings are meaningless; "
                          "fine for benchmarks and synthetic-weight tests only.", RuntimeWarning, stacklevel=2)
        else:
            raise RuntimeError("seva CLIPConditioner: no weights. Set SEVA_CLIP_PATH to an open_clip ViT-H-14 "
                               "(laion2b_s32b_b79k) checkpoint, or pass random_init=True / SEVA_CLIP_RANDOM_INIT=1.")
        self.module.eval().requires_grad_(False)
        self.register_buffer("mean", torch.Tensor([0.48145466, 0.4578275, 0.40821073]), persistent=False)
        self.register_buffer("std", torch.Tensor([0.26862954, 0.26130258, 0.27577711]), persistent=False)
        self._engine = None

    def _apply(self, fn, *a, **k):
        self._engine = None
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._engine = None
        return super().load_state_dict(*a, **k)

    def engine(self):
        if self._engine is None:
            from .._clip_engine import ClipEngine

            self._engine = ClipEngine(self.module, self.params, self.mean, self.std)
        return self._engine

    def preprocess(self, x: torch.Tensor) -> torch.Tensor:
        """Resized + normalised image (n,3,224,224) (reference conditioner.py:24-34).  `forward` never materialises it
        (the kernel writes the patch matrix directly); this accessor re-assembles it from that matrix."""
        return self.engine().preprocess_image(x)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.engine().encode(x)
