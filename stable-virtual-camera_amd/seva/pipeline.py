"""Two-pass trajectory driver over the hot path (BASELINE config 4: 168-view orbit on 8 GPUs).

Composes what the reference's `run_one_scene` does in its 2-pass branch (seva/eval.py:1631-1959) out of this
package's parts, one process per GPU:

    planner.two_pass_plan            anchors + pass-1 windows + pass-2 windows            (host, every rank, identical)
    pass 1   windows that read only the input views are sharded round-robin over the ranks; a strategy whose windows
             consume earlier windows' outputs (`gt-nearest`, eval.py:1720-1740) runs serially on rank 0
    exchange ONE all-gather of the anchor latents (RCCL over xGMI; <= 20 x 4 x 72 x 72 fp32 = 1.66 MB per rank slot)
    pass 2   windows read only {input views, anchors} (eval.py:1890-1906): independent units, `shard_windows`
    gather   finished window latents to rank 0 (round by round), reassembled in target order, optionally VAE-decoded

CFG-split (`cfg_split=True`): strong scaling of ONE window over a PAIR of ranks -- rank 2j runs the
unconditional half of every CFG batch, rank 2j+1 the conditional half, one 435 KB all-gather per step
(`EulerEDMSampler.cfg_split`).  Used where whole windows cannot fill the ranks: the serial first pass (pair 0) and the
leftover second-pass round (10 windows on 8 GPUs: 8 whole windows, then the last 2 on pairs (0,1), (2,3)):
168 views on 8 GPUs = 0.5 + 1 + 0.5 window times instead of 1 + 2 (ceiling 5.5x instead of 3.7x).  Bitwise the same trajectory.

Hand-off between the passes: the reference carries the anchors across the pass boundary as decoded RGB and re-encodes
them per window (eval.py:1820-1829, 1246).  Here the default hand-off is the anchor LATENT itself (`handoff="latent"`:
no decode -> encode round trip, 1.66 MB instead of 40 MB on the wire); `handoff="rgb"` reproduces the reference's
round trip through `AutoEncoder.decode` / `.encode` when an `ae` is given.

Randomness under sharding (SURVEY §8e last bullet; reference eval.py:1294-1295, 1450): the initial noise of window i is
the i-th `torch.randn((T,4,h,w))` draw of ONE CPU stream seeded once per scene -- every rank draws the whole sequence
in plan order and keeps its own windows'.  The per-step device noise (`randn_like`, sampling.py:359) comes from a
generator owned by the window (seed = scene seed, window index), so a window's result does not depend on which rank
runs it or on what ran before it: a sharded run equals the sequential run bit for bit (tested on gloo, world size 2).

CLIP token: `clip_token` (one 1024-d token for every window), `clip_fn(window_source_ids) -> (1024,)`, or -- the reference's
own rule (eval.py:1248: mean CLIP embedding of the window's conditioning views) -- `conditioner=` + `input_rgb=` with
`handoff="rgb"` and an `ae`: the anchors are then decoded once after pass 1, every window's token is
`conditioner(rgb of its conditioning frames).mean(0)` and the anchors re-enter pass 2 through `ae.encode`, as in the reference.
"""

from __future__ import annotations

from dataclasses import dataclass, field
from typing import Callable, Sequence

import numpy as np
import torch
import torch.distributed as dist

from . import conditioning, planner
from . import sampling as S
from .distributed import shard_windows

DEPENDENT_FIRST_PASS = ("gt-nearest", "gt-ltr")  # windows consume earlier windows' outputs (eval.py:530-616)


@dataclass
class Window:
    pass_id: int                  # 1 or 2
    index: int                    # position inside its pass
    global_index: int             # position in the scene-wide draw order of the initial noise
    source_ids: list[int]         # trajectory frame ids of the conditioning frames (inputs, earlier outputs, anchors)
    source_slots: list[int]
    target_ids: list[int]         # trajectory frame ids this window generates
    target_slots: list[int]
    slot_frame: list[int] = field(default_factory=list)   # per slot: trajectory frame id (padding repeats a frame)
    slot_is_input: list[bool] = field(default_factory=list)


@dataclass
class TrajectoryPlan:
    T: int
    input_ids: list[int]
    anchor_ids: list[int]
    pass1: list[Window]
    pass2: list[Window]
    pass1_serial: bool


def _windows(pass_id, base, T, chunks, src_pool, tgt_pool, padding_mode="last"):
    _, ins, in_slots, tes, te_slots = chunks
    out = []
    for i, (ci, cs, ti, ts) in enumerate(zip(ins, in_slots, tes, te_slots)):
        cs2, ts2, in_map, te_map = planner.pad_indices(list(cs), list(ts), T=T, padding_mode=padding_mode)
        slot_frame, slot_in = [], []
        for s in range(T):
            if in_map[s] != -1:
                slot_frame.append(src_pool[ci[in_map[s]]])
                slot_in.append(True)
            else:
                slot_frame.append(tgt_pool[ti[te_map[s]]])
                slot_in.append(False)
        out.append(Window(pass_id, i, base + i, [src_pool[j] for j in ci], list(cs), [tgt_pool[j] for j in ti], list(ts),
                          slot_frame, slot_in))
        # a padded slot that repeats an INPUT frame is conditioning too (reference: curr_input_sels after pad_indices)
        out[-1].source_slots = [s for s in range(T) if slot_in[s]]
    return out


def plan_trajectory(c2ws: torch.Tensor, input_ids: Sequence[int], T: int = 21, chunk_strategy: str = "interp",
                    first_pass_strategy: str = "gt-nearest", options: dict | None = None,
                    task: str = "img2trajvid", refine_anchors: bool = True) -> TrajectoryPlan:
    """Anchors + windows of both passes for one trajectory; frame ids index `c2ws` (inputs first, like the reference's
    `input_indices` convention in run_one_scene).

    Defaults are the reference's (eval.py:1656 `chunk_strategy_first_pass` = "gt-nearest"; eval.py:1459 + 1869-1885: the
    second pass targets EVERY non-input frame, so the anchors -- which the `interp` chunker places as the first target of
    the range they open -- are generated again in pass 2 and the trajectory's anchor frames are second-pass samples;
    golden: tests/golden/g10_two_pass_plans.json).  `refine_anchors=False` is this package's cheaper variant: the second
    pass skips the anchors and their final latents are the first-pass samples (one target slot more per range)."""
    n = c2ws.shape[0]
    ins = [int(i) for i in input_ids]
    opts = {"sampler_verbose": False, **(options or {}), "chunk_strategy": chunk_strategy}
    vd = {"T": T, "options": opts}
    n_prior = planner.infer_prior_stats(T, len(ins), n - len(ins), vd)
    T1, T2 = vd["T"] if isinstance(vd["T"], (list, tuple)) else (T, T)
    assert T1 == T2 == T, "different window lengths per pass are not driven by this pipeline"
    anchors = [int(v) for v in planner.infer_prior_inds(c2ws, n_prior, ins, opts)]
    p1 = planner.chunk_input_and_test(T, c2ws[ins], c2ws[anchors], [float(i) for i in ins], [float(a) for a in anchors],
                                      opts, task=task, chunk_strategy=first_pass_strategy,
                                      gt_input_inds=list(range(len(ins))))
    serial = first_pass_strategy in DEPENDENT_FIRST_PASS
    # pass-1 source pool: the inputs, then (dependent strategies) the anchors in the order they are generated
    pool1 = list(ins)
    if serial:
        for ti in p1[3]:
            pool1.extend(anchors[j] for j in ti)
    w1 = _windows(1, 0, T, p1, pool1, anchors, opts.get("t_padding_mode", "last"))
    order = np.argsort(ins + anchors).tolist()
    pool2 = [(ins + anchors)[o] for o in order]
    skip = set(ins) if refine_anchors else set(ins) | set(anchors)
    rest = [i for i in range(n) if i not in skip]
    p2 = planner.chunk_input_and_test(T, c2ws[pool2], c2ws[rest], [float(i) for i in pool2], [float(r) for r in rest],
                                      opts, task=task, chunk_strategy=chunk_strategy,
                                      gt_input_inds=[order.index(i) for i in range(len(ins))])
    w2 = _windows(2, len(w1), T, p2, pool2, rest)
    return TrajectoryPlan(T, ins, anchors, w1, w2, serial)


def _rank_world(group=None):
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def _all_gather_padded(local: torch.Tensor, count: int, max_count: int, group=None) -> list[torch.Tensor]:
    """All-gather of per-rank tensors with different leading sizes: each rank pads to `max_count` rows."""
    rank, world = _rank_world(group)
    if world == 1:
        return [local[:count]]
    buf = torch.zeros((max_count,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    buf[:count] = local[:count]
    out = torch.empty((world * max_count,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, buf.contiguous(), group=group)
    return list(out.view(world, max_count, *local.shape[1:]).unbind(0))


_pair_groups: dict = {}


def _now(device) -> float:
    """Wall clock after the device has drained (per-window times of `run_trajectory(timers=...)`)."""
    import time
    if device.type == "cuda":
        torch.cuda.synchronize(device)
    return time.perf_counter()


def cfg_pair_groups(group=None) -> list:
    """Two-rank process groups (0,1), (2,3), ... of `group` (default: the world), created once per process: every rank
    calls this (new_group is collective) and gets the same list; entry j is the group of ranks 2j and 2j+1."""
    rank, world = _rank_world(group)
    key = (id(group), world)
    if key not in _pair_groups:
        ranks = list(range(world)) if group is None else dist.get_process_group_ranks(group)
        _pair_groups[key] = [dist.new_group([ranks[2 * j], ranks[2 * j + 1]]) for j in range(world // 2)]
    return _pair_groups[key]


def second_pass_schedule(n_windows: int, world: int, cfg_split: bool) -> list[list[tuple]]:
    """Rounds of (window index, ranks) for the second pass.  Whole-window rounds: window r*world + q on rank q.  With
    `cfg_split`, a last round of L <= world/2 leftover windows runs window j on the pair (2j, 2j+1)."""
    rounds, i = [], 0
    while i < n_windows:
        left = n_windows - i
        if cfg_split and world >= 2 and left <= world // 2:
            rounds.append([(i + j, (2 * j, 2 * j + 1)) for j in range(left)])
            i += left
        else:
            k = min(left, world)
            rounds.append([(i + q, (q,)) for q in range(k)])
            i += k
    return rounds


def run_window(win: Window, latents_of: dict, denoise_net: Callable, c2ws, Ks, *, hw, num_steps, cfg, cfg_min, guider,
               camera_scale, noise: torch.Tensor, step_seed: int, clip_token: torch.Tensor, device,
               sampler_hook: Callable | None = None, cfg_split: tuple | None = None) -> torch.Tensor:
    """One window = the reference's get_value_dict + do_sample (eval.py:1152-1321) on this package's parts.
    `latents_of[frame_id]` -> (4,h,w) latent of every conditioning frame.  Returns the (T,4,h,w) sample."""
    T = len(win.slot_frame)
    h, w = hw
    frames = win.slot_frame
    mask = torch.tensor(win.slot_is_input, dtype=torch.bool)
    in_slots = [s for s in range(T) if win.slot_is_input[s]]
    vd = conditioning.get_value_dict((h * 8, w * 8), in_slots, c2ws[frames][:, :3], Ks[frames].clone(), c2ws,
                                     camera_scale, device=device)
    lat = torch.stack([latents_of[frames[s]] for s in in_slots]).to(device)
    cond, uc = conditioning.assemble_cond(lat, clip_token, mask, vd["plucker_coordinate"])
    disc = S.DDPMDiscretization()
    den = S.DiscreteDenoiser(disc, num_idx=1000, device=device)
    g = [S.VanillaCFG(), S.MultiviewCFG(cfg_min), S.MultiviewTemporalCFG(T, cfg_min)][guider]
    sampler = S.EulerEDMSampler(disc, g, num_steps=num_steps, verbose=False, device=device, s_churn=0.0)
    gen = torch.Generator(device=device)
    gen.manual_seed(int(step_seed))
    sampler.noise_fn = lambda x: torch.randn(x.shape, generator=gen, device=x.device, dtype=x.dtype)
    sampler.cfg_split = cfg_split  # (two-rank group, half): both ranks draw the same per-step noise from `step_seed`
    if sampler_hook is not None:
        sampler_hook(sampler)
    kw = {} if guider == 0 else dict(c2w=vd["c2w"].to(device), K=Ks[frames].to(device), input_frame_mask=mask.to(device))
    return sampler(lambda x, s, c: den(denoise_net, x, s, c, num_frames=T), noise.to(device).clone(), scale=cfg,
                   cond=cond, uc=uc, verbose=False, **kw)


def run_trajectory(denoise_net: Callable, input_latents: torch.Tensor, c2ws: torch.Tensor, Ks: torch.Tensor,
                   input_ids: Sequence[int], *, clip_token: torch.Tensor | None = None, clip_fn: Callable | None = None,
                   T: int = 21, num_steps: int = 50, cfg: float = 2.0, cfg_min: float = 1.2, guider: int = 1,
                   camera_scale: float = 2.0, seed: int = 23, chunk_strategy: str = "interp",
                   first_pass_strategy: str = "gt-nearest", refine_anchors: bool = True, device=None, group=None, ae=None,
                   handoff: str = "latent",
                   plan: TrajectoryPlan | None = None, timers: dict | None = None,
                   sampler_hook: Callable | None = None, conditioner: Callable | None = None,
                   input_rgb: torch.Tensor | None = None, cfg_split: bool = False) -> dict:
    """Generate every non-input frame of a trajectory.  `denoise_net(x, t, cond, num_frames=T)` is the network call
    (`SGMWrapper(model)`); `input_latents` (n_in,4,h,w) are the VAE-encoded input views (x 0.18215), frame ids
    `input_ids` index `c2ws` (n,4,4) / `Ks` (n,3,3).  Returns, on rank 0, {"latents": (n,4,h,w) in frame order,
    "frame_ids", "plan", "anchor_latents" (first-pass anchors as pass 2 saw them), ["rgb"]}; other ranks get {"plan"} only."""
    rank, world = _rank_world(group)
    device = torch.device(device) if device is not None else input_latents.device
    h, w = input_latents.shape[-2:]
    plan = plan or plan_trajectory(c2ws, input_ids, T, chunk_strategy, first_pass_strategy, refine_anchors=refine_anchors)
    rgb_of: dict = {}
    if conditioner is not None:
        # the reference's rule (eval.py:1248): token of a window = mean CLIP embedding of its conditioning views' RGB
        assert input_rgb is not None and ae is not None and handoff == "rgb", \
            "conditioner= needs input_rgb=, ae= and handoff='rgb' (anchor RGB comes from the VAE decode between the passes)"
        for i, fid in enumerate(plan.input_ids):
            rgb_of[fid] = input_rgb[i].to(device)

        def tok(ids):
            return conditioner(torch.stack([rgb_of[f] for f in ids])).mean(0)
    else:
        tok = (lambda ids: clip_token) if clip_fn is None else clip_fn
        assert clip_fn is not None or clip_token is not None, "a CLIP token (clip_token / clip_fn / conditioner) is required"

    # initial noise: ONE CPU stream, seeded once per scene, drawn in plan order for EVERY window (eval.py:1294-1295,1450)
    g0 = torch.Generator(device="cpu")
    g0.manual_seed(int(seed))
    n_win = len(plan.pass1) + len(plan.pass2)
    noises = [torch.randn((plan.T, 4, h, w), generator=g0) for _ in range(n_win)]

    def step_seed(win):  # per-window device stream: independent of rank and of execution order
        return (int(seed) * 1000003 + 7919 * (win.global_index + 1)) & 0x7FFFFFFFFFFF

    latents_of = {fid: input_latents[i].to(device) for i, fid in enumerate(plan.input_ids)}
    common = dict(hw=(h, w), num_steps=num_steps, cfg=cfg, cfg_min=cfg_min, guider=guider, camera_scale=camera_scale,
                  device=device, sampler_hook=sampler_hook)

    def mark(name):
        if timers is not None:
            if device.type == "cuda":
                torch.cuda.synchronize(device)
            import time
            timers[name] = time.perf_counter()

    def handoff_latent(z, fids=None):  # (k,4,h,w) generated latents -> what the next window conditions on
        if handoff == "rgb" and ae is not None:
            rgb = ae.decode(z)
            if conditioner is not None and fids is not None:
                for fid, img in zip(fids, rgb):
                    rgb_of[fid] = img
            return ae.encode(rgb)
        return z

    mark("start")
    pairs = cfg_pair_groups(group) if (cfg_split and world >= 2) else []
    # ------------------------------------------------------------------ pass 1
    # serial strategies (and, under cfg_split, every first pass): pair 0 = ranks (0, 1) splits the CFG batch of each window
    split1 = bool(pairs) and (plan.pass1_serial or len(plan.pass1) == 1)
    if split1:
        mine1 = list(range(len(plan.pass1))) if rank < 2 else []
    else:
        mine1 = list(range(len(plan.pass1))) if world == 1 else (
            ([i for i in range(len(plan.pass1))] if rank == 0 else []) if plan.pass1_serial
            else shard_windows(len(plan.pass1), rank, world))
    got_ids, got_lat = [], []
    for i in mine1:
        win = plan.pass1[i]
        t_w = _now(device) if timers is not None else 0.0
        z = run_window(win, latents_of, denoise_net, c2ws, Ks, noise=noises[win.global_index], step_seed=step_seed(win),
                       clip_token=tok(win.source_ids), cfg_split=(pairs[0], rank) if split1 else None, **common)
        if timers is not None:
            timers.setdefault("windows", []).append((1, i, _now(device) - t_w))
        zt = handoff_latent(z[win.target_slots], win.target_ids)
        for fid, lat in zip(win.target_ids, zt):
            latents_of[fid] = lat  # a dependent strategy's next window on this rank may read it
            if not (split1 and rank == 1):  # (rank 1 of the pair holds the same bits; rank 0 is the owner that publishes them)
                got_ids.append(fid)
                got_lat.append(lat)
    mark("pass1")
    # ------------------------------------------------------------------ exchange: one all-gather of the anchors
    counts = [0] * world
    for i, win in enumerate(plan.pass1):
        owner = 0 if (world == 1 or plan.pass1_serial or split1) else i % world
        counts[owner] += len(win.target_ids)
    local = torch.stack(got_lat) if got_lat else torch.zeros((0, 4, h, w), device=device)
    parts = _all_gather_padded(local.to(device=device, dtype=torch.float32), counts[rank], max(max(counts), 1), group)
    owner_ids = [[] for _ in range(world)]
    for i, win in enumerate(plan.pass1):
        owner_ids[0 if (world == 1 or plan.pass1_serial or split1) else i % world].extend(win.target_ids)
    for r in range(world):
        for j, fid in enumerate(owner_ids[r]):
            latents_of[fid] = parts[r][j]
    if conditioner is not None:  # ranks that did not generate an anchor need its RGB for the CLIP token: decode locally (1.66 MB of
        for fid in plan.anchor_ids:  # latents travelled, not 40 MB of RGB)
            if fid not in rgb_of:
                rgb_of[fid] = ae.decode(latents_of[fid][None])[0]
    mark("exchange")
    # ------------------------------------------------------------------ pass 2: independent windows, sharded
    sched = second_pass_schedule(len(plan.pass2), world, bool(pairs))
    outs = {}
    for rnd in sched:
        for i, ranks in rnd:
            if rank not in ranks:
                continue
            win = plan.pass2[i]
            t_w = _now(device) if timers is not None else 0.0
            outs[i] = run_window(win, latents_of, denoise_net, c2ws, Ks, noise=noises[win.global_index],
                                 step_seed=step_seed(win), clip_token=tok(win.source_ids),
                                 cfg_split=(pairs[ranks[0] // 2], ranks.index(rank)) if len(ranks) == 2 else None, **common)
            if timers is not None:
                timers.setdefault("windows", []).append((2, i, _now(device) - t_w))
    mark("pass2")
    # ------------------------------------------------------------------ gather to rank 0, round by round
    collected = {}
    for rnd in sched:
        sender = {ranks[0]: i for i, ranks in rnd}  # a pair's even rank sends (both hold the same bits)
        mine = outs.get(sender[rank]) if rank in sender else None
        if world == 1:
            if mine is not None:
                collected[sender[rank]] = mine
            continue
        send = mine if mine is not None else torch.zeros((plan.T, 4, h, w), device=device)
        bufs = [torch.empty_like(send) for _ in range(world)] if rank == 0 else None
        dist.gather(send.contiguous(), bufs, dst=0, group=group)
        if rank == 0:
            for rr, i in sender.items():
                collected[i] = bufs[rr]
    mark("gather")
    if rank != 0:
        return {"plan": plan}
    n = c2ws.shape[0]
    final = torch.zeros((n, 4, h, w), device=device)
    filled = torch.zeros(n, dtype=torch.bool)
    for fid in plan.input_ids + plan.anchor_ids:  # (anchors: first-pass samples, overwritten below when pass 2 regenerates them)
        final[fid] = latents_of[fid]
        filled[fid] = True
    for i, win in enumerate(plan.pass2):
        z = collected[i]
        for fid, slot in zip(win.target_ids, win.target_slots):
            final[fid] = z[slot]
            filled[fid] = True
    assert bool(filled.all()), f"frames never generated: {torch.nonzero(~filled).flatten().tolist()}"
    # "anchor_latents": what pass 2 conditioned on (first-pass samples after the hand-off), whether or not pass 2 regenerated them
    res = {"latents": final, "frame_ids": list(range(n)), "plan": plan,
           "anchor_latents": {fid: latents_of[fid] for fid in plan.anchor_ids}}
    if ae is not None:
        res["rgb"] = ae.decode(final)
        mark("decode")
    return res
