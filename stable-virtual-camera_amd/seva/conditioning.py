"""Step-invariant inputs of the denoising loop, assembled on the GPU.

Host counterpart of the prologue of the reference's `do_sample` / `get_value_dict` (seva/eval.py:1152-1215,
1237-1290) for callers that do not go through the reference's driver (bench.py, tests): camera centring and scale
normalisation (tiny host linear algebra on (T,4,4) matrices, as in the reference), Pluecker maps and the cond / uc
channel assembly on the device (`seva_plucker_f32`, `seva_cond_concat_f32`).  No CPU fallback.
"""

from __future__ import annotations

import torch

from . import ops
from .geometry import get_plucker_coordinates, to_hom_pose


def normalise_cameras(curr_c2ws: torch.Tensor, all_c2ws: torch.Tensor, camera_scale: float = 2.0):
    """Camera centring on the mean of the inlier cameras and rescaling so that camera 0 sits at distance
    `camera_scale` (reference eval.py:1172-1201).  Returns (c2w, w2c), each (T,4,4) fp32 on the host."""
    c2w = to_hom_pose(curr_c2ws.float().cpu()).clone()
    ref = all_c2ws.float().cpu()
    d2med = torch.norm(ref[:, :3, 3] - ref[:, :3, 3].median(0, keepdim=True).values, dim=-1)
    valid = d2med <= torch.clamp(torch.quantile(d2med, 0.97) * 10, max=1e6)
    c2w[:, :3, 3] -= ref[valid, :3, 3].mean(0, keepdim=True)
    w2c = torch.linalg.inv(c2w)
    d0 = torch.norm(c2w[0, :3, 3])
    s = camera_scale if bool(torch.isclose(d0, torch.zeros(1), atol=1e-5).any()) else camera_scale / d0
    w2c[:, :3, 3] *= s
    c2w[:, :3, 3] *= s
    return c2w, w2c


def get_value_dict(image_hw, curr_input_frame_indices, curr_c2ws, curr_Ks, all_c2ws, camera_scale: float = 2.0,
                   device=None, F: int = 8) -> dict:
    """Geometry entries of the reference's value_dict: `cond_frames_mask`, `c2w`, `K`, `plucker_coordinate`
    (the last one on `device`)."""
    H, W = image_hw
    T = curr_c2ws.shape[0]
    mask = torch.zeros(T, dtype=torch.bool)
    mask[list(curr_input_frame_indices)] = True
    c2w, w2c = normalise_cameras(curr_c2ws, all_c2ws, camera_scale)
    dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
    pl = get_plucker_coordinates(w2c[0].to(dev), w2c.to(dev), curr_Ks.float().clone().to(dev), target_size=(H // F, W // F))
    return {"cond_frames_mask": mask, "c2w": c2w, "K": curr_Ks, "plucker_coordinate": pl}


def assemble_cond(latents: torch.Tensor, clip_token: torch.Tensor, input_mask: torch.Tensor, pluckers: torch.Tensor):
    """cond / uc dictionaries of `do_sample` (reference eval.py:1245-1281) on the device of `pluckers`.

    latents (n_in,4,h,w): encoded input views; clip_token (1024,): mean CLIP embedding; input_mask (T,) bool."""
    dev = pluckers.device
    T, _, h, w = pluckers.shape
    mask = input_mask.to(dev)
    c_concat = torch.empty((T, 7, h, w), dtype=torch.float32, device=dev)
    uc_concat = torch.empty_like(c_concat)
    ops.cond_concat(pluckers.contiguous(), mask.to(torch.uint8).contiguous(), c_concat, uc_concat)
    c_replace = torch.zeros((T, 5, h, w), dtype=torch.float32, device=dev)
    idx = torch.nonzero(mask).flatten()
    c_replace[idx, :4] = latents.to(dev, torch.float32)  # scatter of n_in frames (index plumbing)
    c_replace[idx, 4] = 1.0
    c_cross = clip_token.to(dev, torch.float32)[None, None].repeat(T, 1, 1)
    c = {"crossattn": c_cross, "replace": c_replace, "concat": c_concat, "dense_vector": pluckers}
    uc = {"crossattn": torch.zeros_like(c_cross), "replace": torch.zeros_like(c_replace), "concat": uc_concat,
          "dense_vector": pluckers}
    return c, uc
