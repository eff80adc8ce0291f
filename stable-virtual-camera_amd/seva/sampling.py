"""`seva.sampling` -- drop-in operator API of the reference sampler (seva/sampling.py) on MI355X.

Same classes, constructor arguments, attributes and call signatures as the reference
(SURVEY.md §8b): `DDPMDiscretization`, `DiscreteDenoiser`, `VanillaCFG`, `MultiviewCFG`,
`MultiviewTemporalCFG`, `EulerEDMSampler` (+ the helpers they are built from).

Division of labour
  * Everything that touches a latent-sized tensor ((T,4,h,w) and up) -- replace-blend, c_in
    scaling, c_out/c_skip combine, noise injection, CFG combine, Euler update -- is a HIP kernel
    of libseva_hip.so (`seva.ops`); there is no PyTorch/CPU fallback for those.
  * Host logic on O(T) / O(1000) scalars (sigma schedule, nearest-sigma lookup, per-frame CFG
    scale rule) stays in torch/numpy exactly as in the reference; the step-invariant parts are
    hoisted (the CFG scale vector is cached per (c2w, K, mask)), which also removes the per-step
    host syncs listed in SURVEY.md §7 hard part 4.
  * The per-step Gaussian draw keeps the reference's semantics (`torch.randn_like` on the
    sampler state's device, sampling.py:359) but is injectable through `EulerEDMSampler.noise_fn`
    so a CPU oracle and the GPU path can be fed identical noise.
"""

from __future__ import annotations

import numpy as np
import torch
from tqdm import tqdm

from . import _native, _stepgraph, ops
from ._native import SevaNativeError


# ------------------------------------------------------------------------------- helpers
def append_dims(x: torch.Tensor, target_dims: int) -> torch.Tensor:
    """Reference sampling.py:10-17."""
    extra = target_dims - x.ndim
    if extra < 0:
        raise ValueError(f"input has {x.ndim} dims but target_dims is {target_dims}, which is less")
    return x[(...,) + (None,) * extra]


def append_zero(x: torch.Tensor) -> torch.Tensor:
    return torch.cat([x, x.new_zeros([1])])


def to_d(x: torch.Tensor, sigma: torch.Tensor, denoised: torch.Tensor) -> torch.Tensor:
    """(x - denoised) / sigma, reference sampling.py:24-25."""
    _need_gpu(x, denoised)
    x, denoised = _f32c(x), _f32c(denoised)
    out = torch.empty_like(x)
    ops.to_d(x, denoised, _f32c(sigma), out)
    return out


def make_betas(num_timesteps: int, linear_start: float = 1e-4, linear_end: float = 2e-2) -> np.ndarray:
    return (
        torch.linspace(linear_start**0.5, linear_end**0.5, num_timesteps, dtype=torch.float64) ** 2
    ).numpy()


def generate_roughly_equally_spaced_steps(num_substeps: int, max_step: int) -> np.ndarray:
    return np.linspace(max_step - 1, 0, num_substeps, endpoint=False).astype(int)[::-1]


def _need_gpu(*tensors) -> None:
    for t in tensors:
        if isinstance(t, torch.Tensor) and not t.is_cuda:
            raise SevaNativeError(
                "seva.sampling runs its tensor math in HIP kernels: tensors must live on the GPU "
                "(no CPU fallback)."
            )


def _f32c(t: torch.Tensor) -> torch.Tensor:
    return t.to(torch.float32).contiguous()


# ----------------------------------------------------------------------- discretisation
class EpsScaling(object):
    """Reference sampling.py:46-54 (vectors of length batch; host-sized math)."""

    def __call__(self, sigma: torch.Tensor):
        c_skip = torch.ones_like(sigma, device=sigma.device)
        c_out = -sigma
        c_in = 1 / (sigma**2 + 1.0) ** 0.5
        c_noise = sigma.clone()
        return c_skip, c_out, c_in, c_noise


class DDPMDiscretization(object):
    """Reference sampling.py:57-102: sqrt-linear betas -> alpha-bar -> sigma * e^{log_snr_shift}."""

    def __init__(
        self,
        linear_start: float = 5e-06,
        linear_end: float = 0.012,
        num_timesteps: int = 1000,
        log_snr_shift: float | None = 2.4,
    ):
        self.num_timesteps = num_timesteps
        self.log_snr_shift = log_snr_shift
        betas = make_betas(num_timesteps, linear_start=linear_start, linear_end=linear_end)
        self.alphas_cumprod = np.cumprod(1.0 - betas, axis=0)

    def get_sigmas(self, n: int, device: str | torch.device = "cpu") -> torch.Tensor:
        if n < self.num_timesteps:
            acp = self.alphas_cumprod[generate_roughly_equally_spaced_steps(n, self.num_timesteps)]
        elif n == self.num_timesteps:
            acp = self.alphas_cumprod
        else:
            raise ValueError(f"Expected n <= {self.num_timesteps}, but got n = {n}.")
        sigmas = ((1 - acp) / acp) ** 0.5
        if self.log_snr_shift is not None:
            sigmas = sigmas * np.exp(self.log_snr_shift)
        return torch.flip(torch.tensor(sigmas, dtype=torch.float32, device=device), (0,))

    def __call__(self, n: int, do_append_zero: bool = True, flip: bool = False,
                 device: str | torch.device = "cpu") -> torch.Tensor:
        sigmas = self.get_sigmas(n, device=device)
        sigmas = append_zero(sigmas) if do_append_zero else sigmas
        return sigmas if not flip else torch.flip(sigmas, (0,))


class DiscreteDenoiser(object):
    """Reference sampling.py:105-152."""

    sigmas: torch.Tensor

    def __init__(self, discretization: DDPMDiscretization, num_idx: int = 1000,
                 device: str | torch.device = "cpu"):
        self.scaling = EpsScaling()
        self.discretization = discretization
        self.num_idx = num_idx
        self.device = device
        self.register_sigmas()

    def register_sigmas(self):
        self.sigmas = self.discretization(self.num_idx, do_append_zero=False, flip=True, device=self.device)

    def sigma_to_idx(self, sigma: torch.Tensor) -> torch.Tensor:
        dists = sigma - self.sigmas[:, None]
        return dists.abs().argmin(dim=0).view(sigma.shape)

    def idx_to_sigma(self, idx: torch.Tensor | int) -> torch.Tensor:
        return self.sigmas[idx]

    def __call__(self, network, input: torch.Tensor, sigma: torch.Tensor, cond: dict,
                 **additional_model_inputs) -> torch.Tensor:
        _need_gpu(input, sigma)
        sigma = self.idx_to_sigma(self.sigma_to_idx(sigma))  # (B,) nearest table entries
        c_skip, c_out, c_in, c_noise = self.scaling(sigma)
        c_noise = self.sigma_to_idx(c_noise)
        x = _f32c(input)
        if "replace" in cond:
            rep = _f32c(cond.pop("replace"))
            assert rep.shape[1] == x.shape[1] + 1
            blended = torch.empty_like(x)
            ops.replace_blend(x, rep, blended)  # x*(1-mask) + latent*mask
            x = blended
        x_in = torch.empty_like(x)
        ops.scale_rows(x, _f32c(c_in), x_in)
        net = network(x_in, c_noise, cond, **additional_model_inputs)
        out = torch.empty_like(x)
        ops.denoiser_combine(_f32c(net), x, _f32c(c_out), _f32c(c_skip), out)
        return out


# ------------------------------------------------------------------------------ guiders
def get_camera_dist(source_c2ws: torch.Tensor, target_c2ws: torch.Tensor, mode: str = "translation"):
    """Pairwise camera distance (reference seva/geometry.py:12-40); O(N*M) host-sized math."""
    if mode == "rotation":
        rel = torch.matmul(source_c2ws[:, None, :3, :3], target_c2ws[None, :, :3, :3].transpose(-1, -2))
        cos = (rel.diagonal(offset=0, dim1=-2, dim2=-1).sum(-1) - 1) / 2
        return torch.acos(cos.clamp(-1, 1)) * (180 / torch.pi)
    if mode == "translation":
        return torch.norm(source_c2ws[:, None, :3, 3] - target_c2ws[None, :, :3, 3], dim=-1)
    raise NotImplementedError(f"Mode {mode} is not implemented for finding nearest source indices.")


class ConstantScaleRule(object):
    def __call__(self, scale):
        return scale


class MultiviewScaleRule(object):
    """Reference sampling.py:160-187: frames that coincide with an input view get `min_scale`."""

    def __init__(self, min_scale: float = 1.0):
        self.min_scale = min_scale

    def __call__(self, scale, c2w: torch.Tensor, K: torch.Tensor, input_frame_mask: torch.Tensor):
        c2w_input = c2w[input_frame_mask]
        rotation_diff = get_camera_dist(c2w, c2w_input, mode="rotation").min(-1).values
        translation_diff = get_camera_dist(c2w, c2w_input, mode="translation").min(-1).values
        K_diff = ((K[:, None] - K[input_frame_mask][None]).flatten(-2) == 0).all(-1).any(-1)
        close_frame = (rotation_diff < 10.0) & (translation_diff < 1e-5) & K_diff
        if isinstance(scale, torch.Tensor):
            scale = scale.clone()
            scale[close_frame] = self.min_scale
        elif isinstance(scale, float):
            scale = torch.where(close_frame, self.min_scale, scale)
        else:
            raise ValueError(f"Invalid scale type {type(scale)}.")
        return scale


class ConstantScaleSchedule(object):
    def __call__(self, sigma, scale):
        if isinstance(sigma, float):
            return scale
        elif isinstance(sigma, torch.Tensor):
            if len(sigma.shape) == 1 and isinstance(scale, torch.Tensor):
                sigma = append_dims(sigma, scale.ndim)
            return scale * torch.ones_like(sigma)
        else:
            raise ValueError(f"Invalid sigma type {type(sigma)}.")


class ConstantGuidance(object):
    """uncond + scale*(cond - uncond), reference sampling.py:204-213 -- one HIP kernel."""

    def __call__(self, uncond: torch.Tensor, cond: torch.Tensor, scale) -> torch.Tensor:
        den2 = torch.cat([uncond, cond], 0)
        return _cfg_combine(den2, scale)


def _scale_vector(scale, n: int, device) -> torch.Tensor:
    """Any reference-style scale value (float, (n,), (n,1,1,1)) -> contiguous f32 (n,)."""
    if isinstance(scale, torch.Tensor):
        s = scale.to(device=device, dtype=torch.float32).reshape(-1)
        if s.numel() == 1:
            s = s.expand(n)
        if s.numel() != n:
            raise ValueError(f"guidance scale has {s.numel()} entries for {n} frames")
        return s.contiguous()
    return torch.full((n,), float(scale), device=device, dtype=torch.float32)


def _cfg_combine(den2: torch.Tensor, scale) -> torch.Tensor:
    _need_gpu(den2)
    den2 = _f32c(den2)
    n = den2.shape[0] // 2
    out = torch.empty((n,) + tuple(den2.shape[1:]), device=den2.device, dtype=torch.float32)
    ops.cfg_combine(den2, _scale_vector(scale, n, den2.device), out)
    return out


class VanillaCFG(object):
    """Reference sampling.py:216-242."""

    def __init__(self):
        self.scale_rule = ConstantScaleRule()
        self.scale_schedule = ConstantScaleSchedule()
        self.guidance = ConstantGuidance()

    # the per-frame guidance weight; everything below it is step-invariant
    def frame_scale(self, x: torch.Tensor, sigma, scale, **_):
        return self.scale_schedule(sigma, self.scale_rule(scale))

    def __call__(self, x: torch.Tensor, sigma, scale) -> torch.Tensor:
        return _cfg_combine(x, self.frame_scale(x, sigma, scale))

    def prepare_inputs(self, x: torch.Tensor, s: torch.Tensor, c: dict, uc: dict):
        c_out = dict()
        for k in c:
            if k in ["vector", "crossattn", "concat", "replace", "dense_vector"]:
                c_out[k] = torch.cat((uc[k], c[k]), 0)
            else:
                assert c[k] == uc[k]
                c_out[k] = c[k]
        return torch.cat([x] * 2), torch.cat([s] * 2), c_out


class MultiviewCFG(VanillaCFG):
    """Reference sampling.py:245-265."""

    def __init__(self, cfg_min: float = 1.0):
        self.scale_min = cfg_min
        self.scale_rule = MultiviewScaleRule(min_scale=cfg_min)
        self.scale_schedule = ConstantScaleSchedule()
        self.guidance = ConstantGuidance()
        self._rule_cache: tuple | None = None

    def reset_rule_cache(self) -> None:
        """Called by `EulerEDMSampler.prepare_sampling_loop`: a new trajectory never reuses an old rule."""
        self._rule_cache = None

    def _ruled_scale(self, scale, c2w, K, input_frame_mask):
        """scale_rule(...) with its boolean-mask indexing (a host sync) paid once per trajectory.

        The cache entry keeps STRONG references to the very tensor objects it was computed from and matches by
        object identity (`is`), so an address the caching allocator hands out again for another scene can never
        alias an entry, and nothing here reads `_version` (inference tensors -- reference eval.py:1242 creates
        c2w / K / mask under torch.inference_mode() -- do not track one).  `do_sample` passes the same tensor
        objects on every step of a trajectory (eval.py:1285-1312), which is what makes the hit path the common one;
        in-place edits of those tensors between steps are not part of the reference's contract."""
        ent = self._rule_cache
        if ent is not None:
            (s0, a0, b0, m0), val = ent
            same_scale = (s0 is scale) if isinstance(scale, torch.Tensor) else (
                not isinstance(s0, torch.Tensor) and s0 == scale)
            if same_scale and a0 is c2w and b0 is K and m0 is input_frame_mask:
                return val
        val = self.scale_rule(scale, c2w, K, input_frame_mask)
        self._rule_cache = ((scale, c2w, K, input_frame_mask), val)
        return val

    def frame_scale(self, x, sigma, scale, c2w=None, K=None, input_frame_mask=None, **_):
        return self.scale_schedule(sigma, self._ruled_scale(scale, c2w, K, input_frame_mask))

    def __call__(self, x: torch.Tensor, sigma, scale, c2w: torch.Tensor, K: torch.Tensor,  # type: ignore
                 input_frame_mask: torch.Tensor) -> torch.Tensor:
        return _cfg_combine(x, self.frame_scale(x, sigma, scale, c2w, K, input_frame_mask))


class MultiviewTemporalCFG(MultiviewCFG):
    """Reference sampling.py:268-298: scale additionally ramps with index distance to an input."""

    def __init__(self, num_frames: int, cfg_min: float = 1.0):
        super().__init__(cfg_min=cfg_min)
        self.num_frames = num_frames
        idx = torch.arange(num_frames)
        self.distance_matrix = (idx[None] - idx[:, None]).abs()

    def _ruled_scale(self, scale, c2w, K, input_frame_mask, ndim=4):
        """Temporal ramp + close-frame rule: step-invariant, so (like MultiviewCFG) evaluated once per trajectory and
        cached by the identity of the tensors it came from; its boolean indexing is a host sync that must not sit
        inside a captured step."""
        ent = self._rule_cache
        if ent is not None:
            (s0, a0, b0, m0), val = ent
            same_scale = (s0 is scale) if isinstance(scale, torch.Tensor) else (
                not isinstance(s0, torch.Tensor) and s0 == scale)
            if same_scale and a0 is c2w and b0 is K and m0 is input_frame_mask:
                return val
        mask = input_frame_mask.reshape(-1, self.num_frames)
        min_distance = (
            self.distance_matrix[None].to(mask.device) + (~mask[:, None]) * self.num_frames
        ).min(-1)[0]
        min_distance = min_distance / min_distance.max(-1, keepdim=True)[0].clamp(min=1)
        ramp = min_distance * (scale - self.scale_min) + self.scale_min
        ramp = append_dims(ramp.reshape(-1), ndim)
        val = self.scale_rule(ramp, c2w, K, mask.flatten(0, 1))
        self._rule_cache = ((scale, c2w, K, input_frame_mask), val)
        return val

    def frame_scale(self, x, sigma, scale, c2w=None, K=None, input_frame_mask=None, **_):
        return self.scale_schedule(sigma, self._ruled_scale(scale, c2w, K, input_frame_mask, x.ndim))

    def __call__(self, x: torch.Tensor, sigma, scale, c2w: torch.Tensor, K: torch.Tensor,
                 input_frame_mask: torch.Tensor) -> torch.Tensor:
        return _cfg_combine(x, self.frame_scale(x, sigma, scale, c2w, K, input_frame_mask))


# ------------------------------------------------------------------------------ sampler
class EulerEDMSampler(object):
    """Reference sampling.py:301-405 (Euler discretisation of the EDM probability-flow ODE)."""

    def __init__(self, discretization: DDPMDiscretization, guider, num_steps: int | None = None,
                 verbose: bool = False, device: str | torch.device = "cuda", s_churn=0.0, s_tmin=0.0,
                 s_tmax=float("inf"), s_noise=1.0):
        self.num_steps = num_steps
        self.discretization = discretization
        self.guider = guider
        self.verbose = verbose
        self.device = device
        self.s_churn = s_churn
        self.s_tmin = s_tmin
        self.s_tmax = s_tmax
        self.s_noise = s_noise
        self.noise_fn = torch.randn_like  # injectable: fn(x) -> N(0,1) tensor like x
        self._step_graphs = _stepgraph.StepGraphCache()
        # CFG-split (strong scaling of ONE window over two GPUs): (process group of two ranks, half) with
        # half = this rank's position in the group: 0 runs the unconditional half of the CFG batch, 1 the conditional half.
        # Each step the two ranks all-gather their (T,4,h,w) denoised halves (435 KB at 576x576 over xGMI) and then both
        # perform the identical guidance + Euler update, so both hold the full sampler state; the per-step noise must come
        # from identically seeded generators on both ranks (`noise_fn`; seva.pipeline does that).  None = the whole CFG batch
        # on this rank (the reference's single-process semantics, sampling.py:231-242).
        self.cfg_split: tuple | None = None

    def prepare_sampling_loop(self, x: torch.Tensor, cond: dict, uc: dict, num_steps: int | None = None):
        num_steps = num_steps or self.num_steps
        assert num_steps is not None, "num_steps must be specified"
        _need_gpu(x)
        if hasattr(self.guider, "reset_rule_cache"):
            self.guider.reset_rule_cache()
        sigmas = self.discretization(num_steps, device=self.device)
        # x *= sqrt(1 + sigma_0^2), in place on the caller's tensor like the reference (l.331)
        s0 = torch.sqrt(1.0 + sigmas[0] ** 2.0).to(device=x.device, dtype=torch.float32)
        s0 = s0.expand(x.shape[0]).contiguous()
        if x.dtype == torch.float32 and x.is_contiguous():
            ops.scale_rows(x, s0, x)
        else:
            tmp = _f32c(x)
            ops.scale_rows(tmp, s0, tmp)
            x.copy_(tmp)
        num_sigmas = len(sigmas)
        s_in = x.new_ones([x.shape[0]])
        # host copy so the per-step s_tmin/s_tmax test never syncs with the device
        self._sigmas_host = [float(v) for v in sigmas.detach().cpu()]
        return x, s_in, sigmas, num_sigmas, cond, uc

    def get_sigma_gen(self, num_sigmas: int, verbose: bool = True):
        sigma_generator = range(num_sigmas - 1)
        if self.verbose and verbose:
            sigma_generator = tqdm(sigma_generator, total=num_sigmas - 1, desc="Sampling", leave=False)
        return sigma_generator

    def _step_math(self, sigma, next_sigma, x, eps, denoiser, scale, cond, uc, gamma, guider_kwargs):
        """Reference sampling.py:357-368 on f32 contiguous device tensors; sync-free once the guider's rule is cached,
        which is what makes it capturable as one hipGraph (`_stepgraph.StepGraph`)."""
        sigma_hat = sigma * (gamma + 1.0) + 1e-6
        noise_scale = (sigma_hat**2 - sigma**2) ** 0.5 * self.s_noise
        x_noised = torch.empty_like(x)
        ops.add_noise(x, eps, _f32c(noise_scale), x_noised)
        if self.cfg_split is None:
            denoised2 = denoiser(*self.guider.prepare_inputs(x_noised, sigma_hat, cond, uc))
        else:
            denoised2 = self._denoise_cfg_split(denoiser, x_noised, sigma_hat, cond, uc)
        dt = next_sigma - sigma_hat
        out = torch.empty_like(x)
        if hasattr(self.guider, "frame_scale"):
            # CFG combine + to_d + Euler update in one pass
            fs = self.guider.frame_scale(denoised2, sigma_hat, scale, **guider_kwargs)
            ops.cfg_euler(x_noised, _f32c(denoised2), _scale_vector(fs, x.shape[0], x.device),
                          sigma_hat, dt, out)
        else:
            denoised = self.guider(denoised2, sigma_hat, scale, **guider_kwargs)
            ops.euler_step(x_noised, _f32c(denoised), sigma_hat, dt, out)
        return out

    def _denoise_cfg_split(self, denoiser, x_noised, sigma_hat, cond, uc):
        """[uncond ; cond] denoised latents with each half of the CFG batch computed by one rank of `cfg_split`'s
        two-rank group and exchanged by ONE all-gather (rank order of the group = batch order of
        `VanillaCFG.prepare_inputs`, reference sampling.py:231-242).  Every row of the network's output depends only on
        the rows of its own scene, and the kernels' tilings are chosen from per-sample sizes, so the half batch is bitwise
        the corresponding half of the full batch."""
        import torch.distributed as dist

        group, half = self.cfg_split
        xx, ss, cc = self.guider.prepare_inputs(x_noised, sigma_hat, cond, uc)
        n = x_noised.shape[0]
        sl = slice(half * n, (half + 1) * n)
        cc_half = {k: (v[sl] if isinstance(v, torch.Tensor) and v.dim() > 0 and v.shape[0] == 2 * n else v) for k, v in cc.items()}
        mine = _f32c(denoiser(xx[sl], ss[sl], cc_half))
        both = torch.empty((2 * n,) + tuple(mine.shape[1:]), dtype=mine.dtype, device=mine.device)
        if mine.is_cuda and dist.get_backend(group) == "gloo":
            # rehearsal backend only (RCCL is stream-ordered): gloo stages device tensors through the host, and with the half-batch
            # network still in flight its all-gather took 5.5 s per step on a shared card instead of 3 ms (profiles/r03_bench_n2_rehearsal*)
            torch.cuda.synchronize(mine.device)
        dist.all_gather_into_tensor(both, mine, group=group)
        return both

    @staticmethod
    def _flat_key(obj):
        """Flatten dict / tensor / scalar arguments into a tuple compared by identity (tensors, callables) or value."""
        if isinstance(obj, dict):
            out = []
            for k in sorted(obj):
                out.append(k)
                out.extend(EulerEDMSampler._flat_key(obj[k]))
            return tuple(out)
        if isinstance(obj, (list, tuple)):
            out = []
            for v in obj:
                out.extend(EulerEDMSampler._flat_key(v))
            return tuple(out)
        return (obj,)

    def sampler_step(self, sigma: torch.Tensor, next_sigma: torch.Tensor, denoiser, x: torch.Tensor,
                     scale, cond: dict, uc: dict, gamma: float = 0.0, **guider_kwargs) -> torch.Tensor:
        """One Euler-EDM step (reference sampling.py:347-368).  From the second step of a trajectory on, the whole
        step is replayed from one hipGraph (see `_stepgraph`); the first step runs eagerly and warms every cache.
        Stays a callable unit for `GradioTrackedSampler` (reference eval.py:1037-1089)."""
        _need_gpu(x)
        x = _f32c(x)
        sigma = _f32c(sigma)
        next_sigma = _f32c(next_sigma)
        eps = _f32c(self.noise_fn(x))  # drawn eagerly: the generator advances per step exactly as in the reference
        gamma = float(gamma)
        cache = self._step_graphs
        # (CFG-split steps are not captured as one graph: the per-step all-gather runs eagerly between the two ranks;
        #  the network call itself is still replayed from its own hipGraph)
        use_graph = (x.is_cuda and not cache.disabled and _stepgraph.enabled() and self.cfg_split is None
                     and not torch.cuda.is_current_stream_capturing())
        if not use_graph:
            return self._step_math(sigma, next_sigma, x, eps, denoiser, scale, cond, uc, gamma, guider_kwargs)
        key = (tuple(x.shape), gamma, float(self.s_noise), self.guider, denoiser) + self._flat_key(
            (scale, cond, uc, guider_kwargs))
        ent = cache.lookup(key)
        if ent.state == 0:
            # warm-up = the real first step, launched eagerly (also without the network-only graph: it would be
            # captured for this one call only)
            with _native.eager_network():
                out = self._step_math(sigma, next_sigma, x, eps, denoiser, scale, cond, uc, gamma, guider_kwargs)
            ent.state = 1
            return out
        if ent.graph is None:
            try:
                ent.graph = _stepgraph.StepGraph(
                    lambda s, n, xx, ee: self._step_math(s, n, xx, ee, denoiser, scale, cond, uc, gamma, guider_kwargs),
                    (sigma, next_sigma, x, eps))
                cache.captures += 1
            except Exception as e:  # a step that cannot be captured (foreign denoiser with host syncs, ...) runs eagerly
                import warnings

                cache.disabled = True
                cache.reset()
                try:
                    torch.cuda.synchronize(x.device)
                except Exception:
                    pass
                warnings.warn(f"seva: whole-step hipGraph capture failed ({type(e).__name__}: {e}); "
                              "falling back to eager steps", RuntimeWarning)
                return self._step_math(sigma, next_sigma, x, eps, denoiser, scale, cond, uc, gamma, guider_kwargs)
        # the graph's output buffer is overwritten by the next replay: hand the caller its own copy (435 KB)
        return ent.graph(sigma, next_sigma, x, eps).clone()

    def __call__(self, denoiser, x: torch.Tensor, scale, cond: dict, uc: dict | None = None,
                 num_steps: int | None = None, verbose: bool = True, **guider_kwargs) -> torch.Tensor:
        uc = cond if uc is None else uc
        x, s_in, sigmas, num_sigmas, cond, uc = self.prepare_sampling_loop(x, cond, uc, num_steps)
        for i in self.get_sigma_gen(num_sigmas, verbose=verbose):
            gamma = (
                min(self.s_churn / (num_sigmas - 1), 2**0.5 - 1)
                if self.s_tmin <= self._sigmas_host[i] <= self.s_tmax
                else 0.0
            )
            x = self.sampler_step(s_in * sigmas[i], s_in * sigmas[i + 1], denoiser, x, scale, cond, uc,
                                  gamma, **guider_kwargs)
        # once per trajectory, after the last step (never inside the loop): a split-K consumer that gave up waiting for its
        # producer (csrc/gemm.hip) left a wrong tile behind -- make that an error instead of a silently wrong sample
        if x.is_cuda:
            ops.check_handoffs()
        return x
