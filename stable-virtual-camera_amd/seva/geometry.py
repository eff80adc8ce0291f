"""`seva.geometry` -- the conditioning geometry of the hot path's inputs on MI355X.

Drop-in for the one function of the reference's `seva/geometry.py` that feeds the denoiser:
`get_plucker_coordinates` (reference geometry.py:119-165), here a HIP kernel (`seva_plucker_f32`) over per-view 3x3 /
3x4 matrices that the host inverts exactly as the reference does.  `get_value_dict` (reference seva/eval.py:1152-1215)
imports it by this name, so the reference's own driver picks it up unchanged.

Every OTHER public name of the reference's geometry module (camera-path presets, scene normalisation, ... -- UI and
planning code outside the path) is re-exported from the reference checkout when `SEVA_REFERENCE_PATH` points at one,
because this module shadows the reference's file of the same name (see seva/__init__.py).
"""

from __future__ import annotations

import importlib.util
import math
import os

import torch

from . import ops
from ._native import SevaNativeError

DEFAULT_FOV_RAD = 0.9424777960769379  # 54 degrees (reference geometry.py:9)


def _reexport_reference() -> None:
    ref = os.environ.get("SEVA_REFERENCE_PATH")
    path = os.path.join(ref, "seva", "geometry.py") if ref else None
    if not path or not os.path.isfile(path):
        return
    spec = importlib.util.spec_from_file_location("seva._reference_geometry", path)
    mod = importlib.util.module_from_spec(spec)
    try:
        spec.loader.exec_module(mod)
    except ImportError as e:  # e.g. the reference's own dependency `roma` is not installed
        import warnings

        warnings.warn(f"seva.geometry: could not re-export the reference's geometry helpers from {path}: {e}")
        return
    for name in dir(mod):
        if not name.startswith("_"):
            globals().setdefault(name, getattr(mod, name))


def to_hom_pose(pose: torch.Tensor) -> torch.Tensor:
    """(N,3,4) -> (N,4,4); 4x4 passes through (reference geometry.py:49-55)."""
    if pose.shape[-2:] == (3, 4):
        hom = torch.eye(4, device=pose.device, dtype=pose.dtype)[None].repeat(pose.shape[0], 1, 1)
        hom[:, :3, :] = pose
        return hom
    return pose


def get_default_intrinsics(fov_rad=DEFAULT_FOV_RAD, aspect_ratio: float = 1.0) -> torch.Tensor:
    """(N,3,3) normalised intrinsics (reference geometry.py:58-79)."""
    if not isinstance(fov_rad, torch.Tensor):
        fov_rad = torch.tensor([fov_rad] if isinstance(fov_rad, (int, float)) else fov_rad)
    if aspect_ratio >= 1.0:
        fx = 0.5 / torch.tan(0.5 * fov_rad)
        fy = fx * aspect_ratio
    else:
        fy = 0.5 / torch.tan(0.5 * fov_rad)
        fx = fy / aspect_ratio
    K = fx.new_zeros((fx.shape[0], 3, 3))
    K[:, 0, 0], K[:, 1, 1], K[:, 2, 2] = fx, fy, 1.0
    K[:, 0, 2], K[:, 1, 2] = 0.5, 0.5
    return K


def _compute_device(*tensors: torch.Tensor) -> torch.device:
    for t in tensors:
        if t is not None and t.is_cuda:
            return t.device
    if not torch.cuda.is_available():
        raise SevaNativeError("get_plucker_coordinates runs on the HIP path only: no AMD GPU is visible")
    return torch.device("cuda", torch.cuda.current_device())


def get_plucker_coordinates(extrinsics_src, extrinsics, intrinsics=None, fov_rad=DEFAULT_FOV_RAD, target_size=[72, 72]):
    """(V,6,h,w) Pluecker maps in the source camera's frame; arguments, the in-place rescaling of `intrinsics`
    and the assertion are those of the reference (geometry.py:119-165).  The result lives on the device of
    `extrinsics` (host inputs are computed on the current GPU and copied back)."""
    h, w = int(target_size[0]), int(target_size[1])
    V = extrinsics.shape[0]
    if intrinsics is None:
        intrinsics = get_default_intrinsics(fov_rad).to(extrinsics.device)
    else:
        pp = intrinsics[:, :2, -1]
        if not (torch.all(pp >= 0) and torch.all(pp <= 1)):
            intrinsics[:, :2] /= intrinsics.new_tensor(target_size).view(1, -1, 1) * 8
        pp = intrinsics[:, :2, -1]
        assert torch.all(pp >= 0) and torch.all(pp <= 1), (
            "Intrinsics should be expressed in resolution-independent normalized image coordinates."
        )
    intrinsics[:, :2] *= extrinsics.new_tensor([w, h]).view(1, -1, 1)  # mutates the argument, like the reference
    # The per-view linear algebra (V + 1 inverses of 4x4 / 3x3 matrices) runs on the HOST whatever device the inputs live on:
    # batched GPU inverses of 21 tiny matrices go through the solver library and cost tens of milliseconds per window
    # (measured: cond assembly 22-66 ms, most of it here); on the CPU they cost microseconds.
    ext_h, src_h = extrinsics.detach().float().cpu(), extrinsics_src.detach().float().cpu()
    c2w_src = torch.linalg.inv(src_h)
    rel = torch.einsum("vnm,vmp->vnp", ext_h, c2w_src[None].repeat(V, 1, 1))
    K = intrinsics.detach().float().cpu()
    if K.shape[0] == 1 and V > 1:
        K = K.repeat(V, 1, 1)
    kinv = torch.linalg.inv(K)
    pose_inv = torch.linalg.inv(to_hom_pose(rel[:, :3, :]))[:, :3, :4]
    dev = _compute_device(extrinsics, intrinsics)
    out = torch.empty((V, 6, h, w), dtype=torch.float32, device=dev)
    ops.plucker(kinv.to(dev).contiguous(), pose_inv.to(dev).contiguous(), out)
    return out if extrinsics.device == dev else out.to(extrinsics.device)


_reexport_reference()
