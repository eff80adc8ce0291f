"""CPU oracle for the Seva sampler -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Restates `/root/reference/seva/sampling.py` (+ `seva/geometry.py:12-40`) as plain functions.
Parity status: PINNED against goldens generated from the reference (tests/golden/*.npz,
`oracle/make_goldens.py`).  Only tests/, smoke() and bench.py's cpu_baseline leg import it.
The per-step noise `eps` is an explicit input: the reference draws it from the device RNG
(sampling.py:359-360), which is not portable between CPU and GPU (SURVEY §7 hard part 3).
"""

from __future__ import annotations

import math

import numpy as np
import torch

Tensor = torch.Tensor


def ddpm_sigmas(
    n: int,
    linear_start: float = 5e-6,
    linear_end: float = 0.012,
    num_timesteps: int = 1000,
    log_snr_shift: float | None = 2.4,
    append_zero: bool = True,
    flip: bool = False,
) -> Tensor:
    """DDPMDiscretization.__call__, sampling.py:57-102 (+make_betas 28-38, steps 41-43)."""
    betas = (
        torch.linspace(linear_start**0.5, linear_end**0.5, num_timesteps, dtype=torch.float64) ** 2
    ).numpy()
    acp = np.cumprod(1.0 - betas, axis=0)
    if n < num_timesteps:
        ts = np.linspace(num_timesteps - 1, 0, n, endpoint=False).astype(int)[::-1]
        acp = acp[ts]
    elif n != num_timesteps:
        raise ValueError(f"Expected n <= {num_timesteps}, but got n = {n}.")
    sig = ((1 - acp) / acp) ** 0.5
    if log_snr_shift is not None:
        sig = sig * np.exp(log_snr_shift)
    sig = torch.tensor(sig[::-1].copy(), dtype=torch.float32)
    if append_zero:
        sig = torch.cat([sig, sig.new_zeros(1)])
    return torch.flip(sig, (0,)) if flip else sig


def sigma_to_idx(table: Tensor, sigma: Tensor) -> Tensor:
    """DiscreteDenoiser.sigma_to_idx, sampling.py:126-128 (nearest table entry)."""
    return (sigma.reshape(1, -1) - table[:, None]).abs().argmin(dim=0).view(sigma.shape)


def denoise(network, table: Tensor, x: Tensor, sigma: Tensor, cond: dict, **kw) -> Tensor:
    """DiscreteDenoiser.__call__ + EpsScaling, sampling.py:133-152, 46-54.

    `network(x_in, idx, cond, **kw)`; `cond` is not mutated (the reference pops "replace").
    """
    sigma = table[sigma_to_idx(table, sigma)]
    s = sigma.view(-1, *([1] * (x.ndim - 1)))
    c_in = 1.0 / (s**2 + 1.0) ** 0.5
    c_out = -s
    c_noise = sigma_to_idx(table, sigma)
    cond = dict(cond)
    if "replace" in cond:
        rep = cond.pop("replace")
        lat, mask = rep[:, : x.shape[1]], rep[:, x.shape[1] :]
        x = x * (1 - mask) + lat * mask
    return network(x * c_in, c_noise, cond, **kw) * c_out + x


def camera_dist(src: Tensor, tgt: Tensor, mode: str) -> Tensor:
    """seva/geometry.py:12-40."""
    if mode == "rotation":
        r = torch.matmul(src[:, None, :3, :3], tgt[None, :, :3, :3].transpose(-1, -2))
        tr = r.diagonal(dim1=-2, dim2=-1).sum(-1)
        return torch.acos(((tr - 1) / 2).clamp(-1, 1)) * (180 / math.pi)
    if mode == "translation":
        return torch.norm(src[:, None, :3, 3] - tgt[None, :, :3, 3], dim=-1)
    raise NotImplementedError(mode)


def multiview_scale(scale, cfg_min: float, c2w: Tensor, K: Tensor, mask: Tensor) -> Tensor:
    """MultiviewScaleRule, sampling.py:160-187.  Returns per-frame scale (T,) or (T,1,1,1)."""
    c2w_in = c2w[mask]
    rot = camera_dist(c2w, c2w_in, "rotation").min(-1).values
    tra = camera_dist(c2w, c2w_in, "translation").min(-1).values
    k_same = ((K[:, None] - K[mask][None]).flatten(-2) == 0).all(-1).any(-1)
    close = (rot < 10.0) & (tra < 1e-5) & k_same
    if isinstance(scale, torch.Tensor):
        scale = scale.clone()
        scale[close] = cfg_min
        return scale
    return torch.where(close, torch.tensor(cfg_min), torch.tensor(float(scale)))


def temporal_scale(scale: float, cfg_min: float, mask: Tensor, num_frames: int) -> Tensor:
    """MultiviewTemporalCFG.__call__ scale ramp, sampling.py:288-297 -> (T,1,1,1)."""
    m = mask.view(-1, num_frames)
    idx = torch.arange(num_frames)
    dist = (idx[None] - idx[:, None]).abs()
    md = (dist[None] + (~m[:, None]) * num_frames).min(-1)[0]
    md = md / md.max(-1, keepdim=True)[0].clamp(min=1)
    s = md * (scale - cfg_min) + cfg_min
    return s.reshape(-1)[:, None, None, None]


def guide(den: Tensor, scale, guider: int, cfg_min: float, c2w, K, mask, num_frames: int) -> Tensor:
    """VanillaCFG / MultiviewCFG / MultiviewTemporalCFG __call__, sampling.py:216-298."""
    x_u, x_c = den.chunk(2)
    if guider == 0:
        s = scale
    elif guider == 1:
        s = multiview_scale(float(scale), cfg_min, c2w, K, mask)
    elif guider == 2:
        s = multiview_scale(temporal_scale(float(scale), cfg_min, mask, num_frames), cfg_min, c2w, K, mask)
    else:
        raise ValueError(guider)
    if isinstance(s, torch.Tensor):
        s = s.view(-1, *([1] * (x_c.ndim - 1)))
    return x_u + s * (x_c - x_u)


def euler_edm_sample(
    network,
    noise: Tensor,
    cond: dict,
    uc: dict,
    num_steps: int,
    scale: float,
    eps_per_step: list[Tensor] | None,
    guider: int = 1,
    cfg_min: float = 1.2,
    c2w: Tensor | None = None,
    K: Tensor | None = None,
    input_frame_mask: Tensor | None = None,
    num_frames: int | None = None,
    s_noise: float = 1.0,
    return_trace: bool = False,
):
    """EulerEDMSampler.__call__ with s_churn=0, sampling.py:325-405.

    `eps_per_step[i]` replaces `torch.randn_like(x)` of step i (None -> zeros, which is what
    the term evaluates to while sqrt(sigma_hat^2 - sigma^2) underflows).
    """
    table = ddpm_sigmas(1000, append_zero=False, flip=True)
    sigmas = ddpm_sigmas(num_steps)
    T = noise.shape[0]
    num_frames = num_frames or T
    x = noise.clone() * torch.sqrt(1.0 + sigmas[0] ** 2.0)
    ones = x.new_ones(T)
    trace = []
    cat_keys = ("vector", "crossattn", "concat", "replace", "dense_vector")
    for i in range(num_steps):
        sigma, nxt = ones * sigmas[i], ones * sigmas[i + 1]
        sigma_hat = sigma * 1.0 + 1e-6
        eps = torch.zeros_like(x) if eps_per_step is None else eps_per_step[i] * s_noise
        x = x + eps * ((sigma_hat**2 - sigma**2).view(-1, 1, 1, 1) ** 0.5)
        c2 = {k: (torch.cat((uc[k], cond[k]), 0) if k in cat_keys else cond[k]) for k in cond}
        den = denoise(network, table, torch.cat([x, x]), torch.cat([sigma_hat] * 2), c2,
                      num_frames=num_frames)
        den = guide(den, scale, guider, cfg_min, c2w, K, input_frame_mask, num_frames)
        d = (x - den) / sigma_hat.view(-1, 1, 1, 1)
        x = x + (nxt - sigma_hat).view(-1, 1, 1, 1) * d
        if return_trace:
            trace.append(x.clone())
    return (x, trace) if return_trace else x
