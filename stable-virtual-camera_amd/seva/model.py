"""`seva.model` -- drop-in operator API of the reference (seva/model.py) on MI355X.

`Seva` owns the parameters under exactly the reference's state_dict keys (1146 tensors for the
default 1.3 B configuration, reference seva/model.py:39-174) but holds no PyTorch compute:
`forward` hands raw device pointers to the hand-written HIP kernels of libseva_hip.so through
`seva._engine.SevaEngine`.  There is no CPU / eager fallback -- calling `forward` with CPU
tensors, or without the built library, raises `SevaNativeError`.

Kept API: `SevaParams` (fields and defaults of model.py:17-36), `Seva(params)`
constructible under `torch.device("meta")`, `.to(dtype/device)`, `load_state_dict(..., assign=True)`,
`Seva.forward(x, t, y, dense_y, num_frames=None)`, `SGMWrapper(module).forward(x, t, c, **kw)`.
"""

from __future__ import annotations

from dataclasses import dataclass, field

import torch
import torch.nn as nn

from ._arch import build_layout


@dataclass
class SevaParams(object):
    in_channels: int = 11
    model_channels: int = 320
    out_channels: int = 4
    num_frames: int = 21
    num_res_blocks: int = 2
    attention_resolutions: list[int] = field(default_factory=lambda: [4, 2, 1])
    channel_mult: list[int] = field(default_factory=lambda: [1, 2, 4, 4])
    num_head_channels: int = 64
    transformer_depth: list[int] = field(default_factory=lambda: [1, 1, 1, 1])
    context_dim: int = 1024
    dense_in_channels: int = 6
    dropout: float = 0.0
    unflatten_names: list[str] = field(
        default_factory=lambda: ["middle_ds8", "output_ds4", "output_ds2"]
    )

    def __post_init__(self):
        assert len(self.channel_mult) == len(self.transformer_depth)


class _Holder(nn.Module):
    """Parameter container: `put("a.0.b", module)` registers nested children so that
    state_dict keys come out as `a.0.b.weight`.  Never called."""

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


def _attention_holder(dim: int, ctx: int) -> _Holder:
    h = _Holder()
    h.put("to_q", nn.Linear(dim, dim, bias=False))
    h.put("to_k", nn.Linear(ctx, dim, bias=False))
    h.put("to_v", nn.Linear(ctx, dim, bias=False))
    h.put("to_out.0", nn.Linear(dim, dim))
    return h


def _ff_holder(dim: int, dim_out: int) -> _Holder:
    h = _Holder()
    h.put("net.0.proj", nn.Linear(dim, dim * 8))
    h.put("net.2", nn.Linear(dim * 4, dim_out))
    return h


def _res_holder(spec, emb_dim: int, dense_c: int) -> _Holder:
    h = _Holder()
    h.put("in_layers.0", nn.GroupNorm(32, spec.cin))
    h.put("in_layers.2", nn.Conv2d(spec.cin, spec.cout, 3, 1, 1))
    h.put("emb_layers.1", nn.Linear(emb_dim, spec.cout))
    h.put("dense_emb_layers.0", nn.Conv2d(dense_c, 2 * spec.cin, 1, 1, 0))
    h.put("out_layers.0", nn.GroupNorm(32, spec.cout))
    h.put("out_layers.3", nn.Conv2d(spec.cout, spec.cout, 3, 1, 1))
    if spec.cin != spec.cout:
        h.put("skip_connection", nn.Conv2d(spec.cin, spec.cout, 1, 1, 0))
    return h


def _mvt_holder(spec, ctx: int) -> _Holder:
    c = spec.channels
    h = _Holder()
    h.put("norm", nn.GroupNorm(32, c, eps=1e-6))
    h.put("proj_in", nn.Linear(c, c))
    for i in range(spec.depth):
        b = f"transformer_blocks.{i}"
        h.put(f"{b}.attn1", _attention_holder(c, c))
        h.put(f"{b}.ff", _ff_holder(c, c))
        h.put(f"{b}.attn2", _attention_holder(c, ctx))
        for n in ("norm1", "norm2", "norm3"):
            h.put(f"{b}.{n}", nn.LayerNorm(c))
    h.put("proj_out", nn.Linear(c, c))
    for i in range(spec.depth):
        b = f"time_mix_blocks.{i}"
        h.put(f"{b}.norm_in", nn.LayerNorm(c))
        h.put(f"{b}.ff_in", _ff_holder(c, c))
        h.put(f"{b}.attn1", _attention_holder(c, c))
        h.put(f"{b}.ff", _ff_holder(c, c))
        h.put(f"{b}.attn2", _attention_holder(c, ctx))
        for n in ("norm1", "norm2", "norm3"):
            h.put(f"{b}.{n}", nn.LayerNorm(c))
    return h


class Seva(nn.Module):
    def __init__(self, params: SevaParams) -> None:
        super().__init__()
        self.params = params
        self.model_channels = params.model_channels
        self.out_channels = params.out_channels
        self.num_head_channels = params.num_head_channels
        lay = build_layout(params)
        self._layout = lay
        emb = lay.time_embed_dim
        root = _Holder()
        root.put("time_embed.0", nn.Linear(params.model_channels, emb))
        root.put("time_embed.2", nn.Linear(emb, emb))
        for spec in lay.all_specs():
            if spec.kind == "conv":
                root.put(spec.prefix, nn.Conv2d(spec.cin, spec.cout, 3, padding=1))
            elif spec.kind == "res":
                root.put(spec.prefix, _res_holder(spec, emb, params.dense_in_channels))
            elif spec.kind == "mvt":
                root.put(spec.prefix, _mvt_holder(spec, params.context_dim))
            elif spec.kind == "down":
                root.put(spec.prefix + ".op", nn.Conv2d(spec.channels, spec.channels, 3, 2, 1))
            elif spec.kind == "up":
                root.put(spec.prefix + ".conv", nn.Conv2d(spec.channels, spec.channels, 3, 1, 1))
        root.put("out.0", nn.GroupNorm(32, lay.final_channels))
        root.put("out.2", nn.Conv2d(params.model_channels, params.out_channels, 3, padding=1))
        # hoist the top-level children so keys carry no extra prefix
        for name, child in list(root._modules.items()):
            self.add_module(name, child)
        self._engine = None

    # any parameter movement / reload invalidates the packed fp16 weights
    def _apply(self, fn, *a, **k):
        self._engine = None
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._engine = None
        return super().load_state_dict(*a, **k)

    def engine(self):
        if self._engine is None:
            from ._engine import SevaEngine

            self._engine = SevaEngine(self, getattr(self, "_precision", None))
        return self._engine

    def set_precision(self, precision: str) -> "Seva":
        """"f16" (default, the parity mode) or "fp8" (BASELINE config 5: e4m3 operands on the fp8 MFMA where the reduction
        length allows; separate accuracy class).  Re-packs the weights on the next forward."""
        self._precision = precision
        self._engine = None
        return self

    def forward(
        self,
        x: torch.Tensor,
        t: torch.Tensor,
        y: torch.Tensor,
        dense_y: torch.Tensor,
        num_frames: int | None = None,
    ) -> torch.Tensor:
        num_frames = num_frames or self.params.num_frames
        return self.engine()(x, None, t, y, dense_y, num_frames)


class SGMWrapper(nn.Module):
    def __init__(self, module: Seva):
        super().__init__()
        self.module = module

    def forward(self, x: torch.Tensor, t: torch.Tensor, c: dict, **kwargs) -> torch.Tensor:
        concat = c.get("concat", None)
        if isinstance(self.module, Seva):
            # the channel concat (reference model.py:227) is folded into the layout kernel
            num_frames = kwargs.pop("num_frames", None) or self.module.params.num_frames
            if kwargs:
                raise TypeError(f"unexpected arguments {sorted(kwargs)}")
            return self.module.engine()(x, concat, t, c["crossattn"], c["dense_vector"], num_frames)
        if concat is not None:
            x = torch.cat((x, concat), dim=1)
        return self.module(x, t=t, y=c["crossattn"], dense_y=c["dense_vector"], **kwargs)
