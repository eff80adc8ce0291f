"""TEST INFRASTRUCTURE -- CPU restatement of the conditioning geometry.

Plain torch-CPU fp32, one function per reference function, each citing the file:line it follows.  Pinned against
tests/golden/g8_plucker.npz and g8_value_dict.npz, which were produced by running the reference itself
(oracle/make_goldens_next.py).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""

from __future__ import annotations

import math

import torch

DEFAULT_FOV_RAD = 0.9424777960769379  # 54 degrees, reference seva/geometry.py:9


def to_hom_pose(pose: torch.Tensor) -> torch.Tensor:
    """(N,3,4) -> (N,4,4) with a [0,0,0,1] row; 4x4 input passes through (geometry.py:49-55)."""
    if pose.shape[-2:] == (3, 4):
        hom = torch.eye(4, dtype=pose.dtype)[None].repeat(pose.shape[0], 1, 1)
        hom[:, :3, :] = pose
        return hom
    return pose


def get_default_intrinsics(fov_rad: float = DEFAULT_FOV_RAD, aspect_ratio: float = 1.0) -> torch.Tensor:
    """(1,3,3) normalised intrinsics: focal 0.5/tan(fov/2) on the long side, principal point 0.5 (geometry.py:58-79)."""
    f = 0.5 / math.tan(0.5 * fov_rad)
    if aspect_ratio >= 1.0:
        fx, fy = f, f * aspect_ratio
    else:
        fy = f
        fx = fy / aspect_ratio
    return torch.tensor([[[fx, 0.0, 0.5], [0.0, fy, 0.5], [0.0, 0.0, 1.0]]], dtype=torch.float32)


def get_plucker_coordinates(extrinsics_src, extrinsics, intrinsics=None, fov_rad=DEFAULT_FOV_RAD, target_size=(72, 72)):
    """(V,6,h,w): unit ray direction || moment (centre x direction) of every latent pixel of every view, expressed in
    the source camera's frame (geometry.py:119-165; pixel grid at +0.5, geometry.py:82-89)."""
    h, w = int(target_size[0]), int(target_size[1])
    V = extrinsics.shape[0]
    if intrinsics is None:
        K = get_default_intrinsics(fov_rad).repeat(V, 1, 1) if V > 1 else get_default_intrinsics(fov_rad)
    else:
        K = intrinsics.clone().float()
        pp = K[:, :2, 2]
        if not (torch.all(pp >= 0) and torch.all(pp <= 1)):
            # pixel units: the reference divides ROW 0 by target_size[0]*8 and ROW 1 by target_size[1]*8 (geometry.py:133)
            K[:, 0] /= h * 8
            K[:, 1] /= w * 8
        pp = K[:, :2, 2]
        if not (torch.all(pp >= 0) and torch.all(pp <= 1)):
            raise AssertionError("Intrinsics should be expressed in resolution-independent normalized image coordinates.")
    K = K.clone()
    if K.shape[0] == 1 and V > 1:
        K = K.repeat(V, 1, 1)
    K[:, 0] *= w  # geometry.py:149-154: row 0 by target w, row 1 by target h
    K[:, 1] *= h
    c2w_src = torch.linalg.inv(extrinsics_src.float())
    rel = extrinsics.float() @ c2w_src[None]  # source-camera coordinates -> camera v (geometry.py:146-148)
    pose = rel[:, :3, :]
    ys = torch.arange(h, dtype=torch.float32) + 0.5
    xs = torch.arange(w, dtype=torch.float32) + 0.5
    Y, X = torch.meshgrid(ys, xs, indexing="ij")
    grid = torch.stack([X, Y, torch.ones_like(X)], -1).view(-1, 3)  # [hw,3]
    cam = grid[None] @ torch.linalg.inv(K).transpose(-1, -2)  # img2cam, geometry.py:92-93
    pose_inv = torch.linalg.inv(to_hom_pose(pose))[:, :3, :4]  # cam2world, geometry.py:96-99
    hom = torch.cat([cam, torch.ones_like(cam[..., :1])], -1)
    world = hom @ pose_inv.transpose(-1, -2)
    zero_h = torch.cat([torch.zeros_like(cam), torch.ones_like(cam[..., :1])], -1)
    centre = zero_h @ pose_inv.transpose(-1, -2)
    ray = world - centre
    ray = torch.nn.functional.normalize(ray, dim=-1)
    pl = torch.cat([ray, torch.linalg.cross(centre, ray, dim=-1)], -1)  # geometry.py:163
    return pl.permute(0, 2, 1).reshape(V, 6, h, w)


def normalise_cameras(curr_c2ws, all_c2ws, camera_scale: float):
    """Camera centring and scale normalisation of get_value_dict (eval.py:1172-1201).  Returns (c2w, w2c), (T,4,4)."""
    c2w = to_hom_pose(curr_c2ws.float()).clone()
    ref = all_c2ws
    d2med = torch.norm(ref[:, :3, 3] - ref[:, :3, 3].median(0, keepdim=True).values, dim=-1)
    valid = d2med <= torch.clamp(torch.quantile(d2med, 0.97) * 10, max=1e6)
    c2w[:, :3, 3] -= ref[valid, :3, 3].mean(0, keepdim=True)
    w2c = torch.linalg.inv(c2w)
    d0 = torch.norm(c2w[0, :3, 3])
    s = camera_scale if bool(torch.isclose(d0, torch.zeros(1), atol=1e-5).any()) else camera_scale / d0
    w2c[:, :3, 3] *= s
    c2w[:, :3, 3] *= s
    return c2w, w2c


def get_value_dict(curr_imgs, curr_input_frame_indices, curr_c2ws, curr_Ks, all_c2ws, camera_scale: float, F: int = 8):
    """The geometry part of eval.py:get_value_dict (1152-1215): mask, normalised c2w, Pluecker maps."""
    H, W, T = curr_imgs.shape[-2], curr_imgs.shape[-1], curr_imgs.shape[0]
    mask = torch.zeros(T, dtype=torch.bool)
    mask[list(curr_input_frame_indices)] = True
    c2w, w2c = normalise_cameras(curr_c2ws, all_c2ws, camera_scale)
    pl = get_plucker_coordinates(w2c[0], w2c, curr_Ks.float().clone(), target_size=(H // F, W // F))
    return {"cond_frames_mask": mask, "c2w": c2w, "K": curr_Ks, "plucker_coordinate": pl}


def assemble_cond(latents, clip_token, input_mask, pluckers):
    """cond / uc dictionaries of do_sample (eval.py:1245-1281).

    latents: (n_in,4,h,w) encoded input views; clip_token: (1024,) mean CLIP embedding of the input views;
    input_mask: (T,) bool; pluckers: (T,6,h,w)."""
    T = input_mask.shape[0]
    lat5 = torch.nn.functional.pad(latents, (0, 0, 0, 0, 0, 1), value=1.0)  # + mask channel of ones
    c_cross = clip_token[None, None].repeat(T, 1, 1)
    c_replace = lat5.new_zeros(T, *lat5.shape[1:])
    c_replace[input_mask] = lat5
    m = input_mask[:, None, None, None].expand(T, 1, *pluckers.shape[-2:]).to(pluckers.dtype)
    c = {"crossattn": c_cross, "replace": c_replace, "concat": torch.cat([m, pluckers], 1), "dense_vector": pluckers}
    uc = {"crossattn": torch.zeros_like(c_cross), "replace": torch.zeros_like(c_replace),
          "concat": torch.cat([torch.zeros_like(m), pluckers], 1), "dense_vector": pluckers}
    return c, uc
