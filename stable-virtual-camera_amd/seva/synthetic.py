"""Deterministic synthetic weights and inputs for the Seva hot path.

No checkpoints or datasets exist offline, so tests, the golden
generator (``oracle/make_goldens.py``) and ``bench.py`` all draw weights and inputs from
here.  Everything is keyed by *name*, not by construction order, so the same state_dict
can be loaded into the reference ``Seva`` (to make goldens), into the CPU oracle and into
the HIP-backed ``seva.model.Seva``.

The generator is data only -- it performs no part of the hot-path arithmetic.
"""

from __future__ import annotations

import math
import zlib
from typing import Mapping

import torch

# Residual-branch output projections get a smaller gain so that a random-init stack of
# ~60 residual branches keeps O(1) activations (fp16 GEMM operands must not overflow).
_BRANCH_OUT_SUFFIXES = (
    "out_layers.3.weight",
    "proj_out.weight",
    "to_out.0.weight",
    "net.2.weight",
)


def _key_seed(key: str, seed: int) -> int:
    return (zlib.crc32(key.encode("utf-8")) + 7919 * seed) & 0x7FFFFFFF


def synth_tensor(key: str, shape, seed: int = 0) -> torch.Tensor:
    """One fp32 tensor whose values are exactly representable in bf16.

    The reference ships bf16 weights (seva/utils.py:51); keeping synthetic weights
    bf16-representable means the fp32 oracle, the reference and the fp16 HIP path all see
    bit-identical parameters.
    """
    shape = tuple(int(s) for s in shape)
    g = torch.Generator(device="cpu")
    g.manual_seed(_key_seed(key, seed))
    if key.endswith(".weight") and len(shape) >= 2:
        fan_in = 1
        for s in shape[1:]:
            fan_in *= s
        gain = 0.5 if key.endswith(_BRANCH_OUT_SUFFIXES) else 1.0
        w = torch.randn(shape, generator=g, dtype=torch.float32) * (gain / math.sqrt(fan_in))
    elif key.endswith(".weight"):  # 1-D: GroupNorm / LayerNorm scale
        w = 1.0 + 0.1 * torch.randn(shape, generator=g, dtype=torch.float32)
    else:  # biases
        w = 0.05 * torch.randn(shape, generator=g, dtype=torch.float32)
    return w.to(torch.bfloat16).to(torch.float32)


def synth_state_dict(shapes: Mapping[str, tuple], seed: int = 0) -> dict[str, torch.Tensor]:
    """Name-keyed deterministic state_dict for any ``{key: shape}`` mapping."""
    return {k: synth_tensor(k, s, seed) for k, s in shapes.items()}


def orbit_c2w(num_frames: int, radius: float = 2.0, height: float = 0.3) -> torch.Tensor:
    """(T,4,4) camera-to-world poses on a circle looking at the origin (OpenCV axes)."""
    c2ws = []
    for i in range(num_frames):
        a = 2.0 * math.pi * i / max(num_frames, 1) * 0.5  # half orbit
        eye = torch.tensor([radius * math.sin(a), height, -radius * math.cos(a)])
        fwd = -eye / eye.norm()
        up = torch.tensor([0.0, -1.0, 0.0])
        right = torch.linalg.cross(fwd, -up)
        right = right / right.norm()
        down = torch.linalg.cross(fwd, right)
        m = torch.eye(4)
        m[:3, 0], m[:3, 1], m[:3, 2], m[:3, 3] = right, down, fwd, eye
        c2ws.append(m)
    return torch.stack(c2ws)


def default_K(num_frames: int, fov_deg: float = 54.0) -> torch.Tensor:
    """(T,3,3) normalised intrinsics, fx=fy=0.5/tan(fov/2), cx=cy=0.5."""
    f = 0.5 / math.tan(math.radians(fov_deg) / 2.0)
    K = torch.tensor([[f, 0.0, 0.5], [0.0, f, 0.5], [0.0, 0.0, 1.0]])
    return K[None].repeat(num_frames, 1, 1)


def plucker_maps(c2w: torch.Tensor, K: torch.Tensor, h: int, w: int) -> torch.Tensor:
    """(T,6,h,w) Pluecker ray maps (unit direction ‖ moment) relative to camera 0."""
    T = c2w.shape[0]
    rel = torch.linalg.inv(c2w[0])[None] @ c2w  # camera i -> camera 0 frame
    ys, xs = torch.meshgrid(
        (torch.arange(h, dtype=torch.float32) + 0.5) / h,
        (torch.arange(w, dtype=torch.float32) + 0.5) / w,
        indexing="ij",
    )
    out = torch.empty(T, 6, h, w)
    for i in range(T):
        fx, fy, cx, cy = K[i, 0, 0], K[i, 1, 1], K[i, 0, 2], K[i, 1, 2]
        d_cam = torch.stack([(xs - cx) / fx, (ys - cy) / fy, torch.ones_like(xs)], -1)
        d = d_cam @ rel[i, :3, :3].T
        d = d / d.norm(dim=-1, keepdim=True)
        o = rel[i, :3, 3].expand_as(d)
        m = torch.linalg.cross(o, d, dim=-1)
        out[i] = torch.cat([d, m], -1).permute(2, 0, 1)
    return out


def synth_scene(
    num_frames: int,
    latent_hw: tuple[int, int],
    input_indices: tuple[int, ...] = (0,),
    seed: int = 23,
) -> dict:
    """Everything ``do_sample`` (seva/eval.py:1218-1321) would hand to the sampler.

    Returns dict with ``cond``/``uc`` (keys crossattn, replace, concat, dense_vector in the
    shapes of SURVEY §8a/A3), ``noise`` (T,4,h,w) drawn from the CPU generator after
    ``torch.manual_seed(seed)``, ``c2w``, ``K``, ``input_frame_mask``.
    """
    h, w = latent_hw
    T = num_frames
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    mask = torch.zeros(T, dtype=torch.bool)
    mask[list(input_indices)] = True
    n_in = int(mask.sum())
    latents = torch.randn(n_in, 4, h, w, generator=g) * (0.18215 * 5.0)
    clip = torch.randn(1024, generator=g)
    clip = clip / clip.norm()
    c2w = orbit_c2w(T)
    K = default_K(T)
    pluckers = plucker_maps(c2w, K, h, w)
    noise = torch.randn(T, 4, h, w, generator=g)

    c_cross = clip[None, None].repeat(T, 1, 1)
    uc_cross = torch.zeros_like(c_cross)
    c_replace = torch.zeros(T, 5, h, w)
    c_replace[mask] = torch.cat([latents, torch.ones(n_in, 1, h, w)], 1)
    uc_replace = torch.zeros_like(c_replace)
    m = mask.float()[:, None, None, None].expand(T, 1, h, w)
    c_concat = torch.cat([m, pluckers], 1)
    uc_concat = torch.cat([torch.zeros(T, 1, h, w), pluckers], 1)
    cond = {
        "crossattn": c_cross,
        "replace": c_replace,
        "concat": c_concat,
        "dense_vector": pluckers.clone(),
    }
    uc = {
        "crossattn": uc_cross,
        "replace": uc_replace,
        "concat": uc_concat,
        "dense_vector": pluckers.clone(),
    }
    return {
        "cond": cond,
        "uc": uc,
        "noise": noise,
        "c2w": c2w,
        "K": K,
        "input_frame_mask": mask,
    }
