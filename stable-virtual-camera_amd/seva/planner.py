"""Chunk planner / two-pass scheduler of long trajectories -- host logic.

Decides how a trajectory of M input views and N target views is cut into windows of T frames, which earlier
outputs ("anchors") condition which window, and how short windows are padded.  It is the work distributor of the
multi-GPU path: the windows of the second pass are independent units (`seva.distributed.shard_windows`).

Same function names, arguments, return values and error behaviour as the reference's planner in
`seva/eval.py` (`pad_indices` 44-82, `assemble` 85-96, `infer_prior_stats` 344-421, `infer_prior_inds` 424-452,
`compute_relative_inds` 455-489, `find_nearest_source_inds` 492-500, `chunk_input_and_test` 503-852), re-implemented
on small typed records instead of string tags; pinned by 900+ known-answer layouts generated from the reference
(tests/golden/g8_planner.json).  A window is a list of T slots written like the reference's: "!007" = input view 7,
">012" = target view 12, "NULL" = padding.
"""

from __future__ import annotations

import collections
import math
import re
from typing import List, Literal, Sequence

import numpy as np
import torch

from .sampling import get_camera_dist

NULL = "NULL"


def _inp(i: int) -> str:
    return f"!{int(i):03d}"


def _tst(i: int) -> str:
    return f">{int(i):03d}"


# ---------------------------------------------------------------------------------------------------------------
def pad_indices(input_indices: List[int], test_indices: List[int], T: int,
                padding_mode: Literal["first", "last", "none"] = "last"):
    """Fill the slots of a T-window that neither an input nor a target occupies by repeating the LAST element of
    whichever group reaches further (reference eval.py:44-82).  Returns the (possibly extended, sorted) index lists
    and, per slot, which element of `input` / `test` lands there (-1 = none)."""
    assert padding_mode in ["last", "none"], "`first` padding is not supported yet."
    taken = set(input_indices) | set(test_indices)
    free = [i for i in range(T) if i not in taken] if padding_mode == "last" else []
    in_sel = list(range(len(input_indices)))
    te_sel = list(range(len(test_indices)))

    def extend(idx, sel):
        idx = list(idx) + free
        sel = sel + [sel[-1]] * len(free)
        order = np.argsort(idx)
        return [idx[o] for o in order], [sel[o] for o in order]

    if max(input_indices) > max(test_indices):
        input_indices, in_sel = extend(input_indices, in_sel)
    else:
        test_indices, te_sel = extend(test_indices, te_sel)
    size = T if padding_mode == "last" else len(input_indices) + len(test_indices)
    input_maps = np.full(size, -1)
    test_maps = np.full(size, -1)
    input_maps[input_indices] = in_sel
    test_maps[test_indices] = te_sel
    return input_indices, test_indices, input_maps, test_maps


def assemble(input, test, input_maps, test_maps):
    """Scatter `input` and `test` frames into one T-window according to the maps of `pad_indices`
    (reference eval.py:85-96)."""
    T = len(input_maps)
    out = torch.zeros_like(test[-1:]).repeat_interleave(T, dim=0)
    has_in, has_te = input_maps != -1, test_maps != -1
    out[has_in] = input[input_maps[has_in]]
    out[has_te] = test[test_maps[has_te]]
    assert np.logical_xor(has_in, has_te).all()
    return out


# ---------------------------------------------------------------------------------------------------------------
def infer_prior_stats(T, num_input_frames: int, num_total_frames: int, version_dict: dict) -> int:
    """Number of anchor ("prior") frames the first pass must produce so that the second pass can bound every
    target between two anchors (reference eval.py:344-421).  May rewrite `version_dict["T"]` = [T_pass1, T_pass2]
    in the semi-dense regime, exactly when the reference does."""
    options = version_dict["options"]
    strategy = options.get("chunk_strategy", "nearest")
    two = isinstance(T, (list, tuple))
    T1, T2 = (T[0], T[1]) if two else (T, T)
    semi_dense = num_input_frames >= options.get("num_input_semi_dense", 9)
    ratio = options.get("num_prior_frames_ratio", 1.0)
    floor_prior = options.get("num_prior_frames", 0)

    if strategy.startswith("interp"):
        # the two ends of a second-pass window are anchors; with `gt` the ground-truth inputs also take slots
        # (only subtracted here in the sparse regime, like the reference)
        usable = T2 - 2 - (num_input_frames if ("gt" in strategy and not semi_dense) else 0)
        n = math.ceil(num_total_frames / usable * ratio) + 1
        if n + num_input_frames < T1:
            n = T1 - num_input_frames
        n = max(n, floor_prior)
        if semi_dense:
            T1 = n + num_input_frames
            if "gt" in strategy:
                T2 = T2 + num_input_frames
            version_dict["T"] = [T1, T2]
        return n
    n = max(T1 - num_input_frames, floor_prior)
    if semi_dense:
        version_dict["T"] = [n + num_input_frames, T2]
    return n


def infer_prior_inds(c2ws, num_prior_frames: int, input_frame_indices, options: dict):
    """Which frames of the trajectory become anchors (reference eval.py:424-452): evenly spread over the non-input
    frames for `interp*`, otherwise greedily the frame farthest (in index) from everything chosen so far."""
    n = c2ws.shape[0]
    if options.get("chunk_strategy", "nearest").startswith("interp"):
        free = np.array([i for i in range(n) if i not in input_frame_indices])
        pick = np.ceil(np.linspace(0, free.shape[0] - 1, num_prior_frames, endpoint=True)).astype(int)
        return np.sort(free[pick])
    chosen: list = []
    while len(chosen) < num_prior_frames:
        taken = np.concatenate([np.array(input_frame_indices), np.array(chosen)])
        gap = np.abs(np.arange(n)[None] - taken[:, None]).min(0)
        chosen.append(np.argsort(gap)[-1])
    return np.sort(chosen)


def compute_relative_inds(source_inds, target_inds):
    """Position of each target index on the (fractional) index axis of `source_inds`, extrapolating linearly
    beyond both ends (reference eval.py:455-489)."""
    source_inds = np.asarray(source_inds)
    assert len(source_inds) > 2
    out = []
    for ind in target_inds:
        hit = np.where(source_inds == ind)[0]
        if len(hit):
            rel = int(hit[0])
        elif ind < source_inds[0]:
            rel = -((source_inds[0] - ind) / (source_inds[1] - source_inds[0]))
        elif ind > source_inds[-1]:
            rel = len(source_inds) + (ind - source_inds[-1]) / (source_inds[-1] - source_inds[-2])
        else:
            lo = int(np.where(source_inds < ind)[0][-1])
            hi = int(np.where(source_inds > ind)[0][0])
            rel = lo + (ind - source_inds[lo]) / (source_inds[hi] - source_inds[lo]) * (hi - lo)
        out.append(rel)
    return out


def find_nearest_source_inds(source_c2ws, target_c2ws, nearest_num: int = 1, mode: str = "translation"):
    """(n_target, nearest_num) indices of the closest source cameras (reference eval.py:492-500)."""
    dists = get_camera_dist(source_c2ws, target_c2ws, mode=mode).cpu().numpy()
    return np.argsort(dists, axis=0).T[:, :nearest_num]


# ---------------------------------------------------------------------------------------------------------------
def _most_common(items, k):
    return [v for v, _ in collections.Counter(items).most_common(k)]


def _plan_gt(T, N, test_c2ws, options, strategy, gt):
    """Every window starts with all ground-truth inputs; `gt-ltr` / `gt-nearest` additionally re-use already generated
    targets ("pseudo" inputs, numbered after the gt ones) from the second window on."""
    G = len(gt)
    windows, seen = [], 0
    while seen < N:
        win = [_inp(i) for i in gt]
        room = T - G
        if strategy != "gt" and seen > 0:
            ratio = options.get("pseudo_num_ratio", 0.33)
            left = N - seen
            pseudo = math.ceil(room * ratio) if left >= math.floor(room * ratio) else room - left
            pseudo = min(pseudo, options.get("pseudo_num_max", 10000))
            if "ltr" in strategy:
                picks = list(range(seen - pseudo, seen))
            elif "nearest" in strategy:
                near = np.concatenate(
                    [find_nearest_source_inds(test_c2ws[:seen], test_c2ws[seen:], 1, mode="rotation"),
                     find_nearest_source_inds(test_c2ws[:seen], test_c2ws[seen:], 1, mode="translation")], axis=1)
                # the number of pseudo inputs and the span of upcoming targets they are voted from depend on each
                # other: shrink until stable; the most recent output (seen-1) is always kept
                span = pseudo
                while True:
                    votes = [v for v in near[: room - span].flatten().tolist() if v != seen - 1]
                    picks = np.concatenate([np.sort(_most_common(votes, pseudo - 1)).astype(int), [seen - 1]])
                    if len(picks) >= span:
                        break
                    span = len(picks)
                pseudo = len(picks)
                picks = list(picks)
            else:
                raise NotImplementedError(f"Chunking strategy {strategy} for the first pass is not implemented.")
            win += [_inp(i + G) for i in picks]
            room -= pseudo
        fresh = range(seen, min(seen + room, N))
        win += [_tst(i) for i in fresh]
        seen += len(fresh)
        windows.append(win + [NULL] * (T - len(win)))
    return windows


def _plan_nearest_k(T, N, input_c2ws, test_c2ws, k):
    assert k < T, f"Nearest number of {k} should be less than {T}."
    near = find_nearest_source_inds(input_c2ws, test_c2ws, nearest_num=k, mode="translation")
    step = T - k
    windows = []
    for i in range(0, N, step):
        ins = np.sort(_most_common(near[i:i + step].flatten().tolist(), k))
        win = [_inp(j) for j in ins] + [_tst(j) for j in range(i, min(i + step, N))]
        windows.append(win + [NULL] * (T - len(win)))
    return windows


def _plan_nearest(T, N, input_c2ws, test_c2ws, strategy, gt):
    """Targets are grouped behind their nearest input view and the groups are packed, in input order, into windows."""
    if "gt" not in strategy:
        gt = []
    owner = find_nearest_source_inds(input_c2ws, test_c2ws, nearest_num=1, mode="translation")[:, 0]
    group: dict = {}
    for t, i in enumerate(owner):
        group.setdefault(i, []).append(t)
    base = [_inp(i) for i in gt]
    queue = sorted(group)
    windows, win, seen = [], list(base), 0
    while seen < N:
        i = queue[0]
        cond = i in gt
        lead = [] if cond else [_inp(i)]
        if len(win) == T - len(lead) or not queue:
            if win:
                windows.append(win + [NULL] * (T - len(win)))
                win = list(base)
            if seen >= N:
                break
            continue
        cand = lead + [_tst(t) for t in group[i]]
        room = T - len(win)
        if len(cand) <= room:
            win += cand
            seen += len(group[i])
            queue.pop(0)
        else:
            win += cand[:room]
            used = room - len(lead)
            seen += used
            group[i] = group[i][used:]
        if len(win) == T:
            windows.append(win)
            win = list(base)
    if win and win != base:
        windows.append(win + [NULL] * (T - len(win)))
    return windows


def _plan_interp(T, M, N, input_c2ws, input_ords, test_ords, strategy, task, gt):
    """Second pass of the two-pass scheme: consecutive anchors bound the targets whose trajectory order lies between
    them; a window holds as many [anchor, targets..., anchor] spans as fit and adjacent windows share an anchor."""
    assert input_ords is not None and test_ords is not None, (
        "When using `interp` chunking strategy, ordering of input "
        "and test frames should be provided."
    )
    base_i = 0
    if "img2trajvid" in task:
        # the original input views have no position among the targets: only anchors bound spans
        assert list(range(len(gt))) == gt, "`img2trajvid` task should put `gt_input_inds` in start."
        input_ords = [o for i, o in enumerate(input_ords) if i not in gt]
        M = M - len(gt)
        base_i = len(gt)
    stops = [0] + list(input_ords)   # a virtual stop for targets before the first anchor
    stops[-1] += 0.01                # keeps a target that coincides with the last anchor inside the last span
    lo = np.array(stops, dtype=float)[:, None]
    hi = np.concatenate([lo[1:], np.full((1, 1), np.inf)])
    te = np.array(test_ords, dtype=float)[None]
    inside = np.logical_and(lo <= te, hi > te)  # (M+1, N)
    assert (inside.sum(1) <= T - 2).all(), (
        "More anchor frames need to be sampled during the first pass to ensure "
        f"#target frames during each forward in the second pass will not exceed {T - 2}."
    )
    if lo[1, 0] <= te[0, 0]:
        assert not inside[0].any()
    if lo[-1, 0] >= te[0, -1]:
        assert not inside[-1].any()
    head = [_inp(i) for i in gt] if "gt" in strategy else []
    win = list(head) + [_tst(j) for j in np.nonzero(inside[0])[0]]
    spans = inside[1:]
    windows, i = [], 0
    win.append(_inp(base_i))
    while i < len(spans):
        members = np.nonzero(spans[i])[0]
        if len(members) == 0:
            i += 1
            continue
        closing = i + 1 < M
        if len(members) + closing <= T - len(win):
            win += [_tst(j) for j in members]
            i += 1
            if closing:
                win.append(_inp(i + base_i))
        else:
            windows.append(win + [NULL] * (T - len(win)))
            win = list(head) + [_inp(i + base_i)]
    if len(win) > 1:
        windows.append(win + [NULL] * (T - len(win)))
    return windows


def chunk_input_and_test(T, input_c2ws, test_c2ws, input_ords, test_ords, options, task: str = "img2img",
                         chunk_strategy: str = "gt", gt_input_inds: Sequence[int] = ()):
    """Cut (M inputs, N targets) into windows of T slots (reference eval.py:503-852).

    Returns (chunks, input_inds_per_chunk, input_sels_per_chunk, test_inds_per_chunk, test_sels_per_chunk):
    the windows as tag lists, and per window the raw indices of its inputs / targets and the slots they occupy."""
    M, N = input_c2ws.shape[0], test_c2ws.shape[0]
    gt = list(gt_input_inds)
    if chunk_strategy.startswith("gt"):
        assert len(gt) < T, (
            f"Number of gt input frames {len(gt)} should be "
            f"less than {T} when `gt` chunking strategy is used."
        )
        assert list(range(M)) == gt, "All input_c2ws should be gt when `gt` chunking strategy is used."
        chunks = _plan_gt(T, N, test_c2ws, options, chunk_strategy, gt)
    elif chunk_strategy.startswith("nearest"):
        m = re.match(r"^nearest-(\d+)$", chunk_strategy)
        if m:
            chunks = _plan_nearest_k(T, N, input_c2ws, test_c2ws, int(m.group(1)))
        else:
            chunks = _plan_nearest(T, N, input_c2ws, test_c2ws, chunk_strategy, gt)
    elif chunk_strategy.startswith("interp"):
        chunks = _plan_interp(T, M, N, input_c2ws, input_ords, test_ords, chunk_strategy, task, gt)
    else:
        raise NotImplementedError
    ins, in_slots, tes, te_slots = [], [], [], []
    for win in chunks:
        ins.append([int(s[1:]) for s in win if s.startswith("!")])
        in_slots.append([win.index(s) for s in win if s.startswith("!")])
        tes.append([int(s[1:]) for s in win if s.startswith(">")])
        te_slots.append([win.index(s) for s in win if s.startswith(">")])
    if options.get("sampler_verbose", True):
        print("\nchunks:")
        for win in chunks:
            print(", ".join(win))
    return chunks, ins, in_slots, tes, te_slots


# ---------------------------------------------------------------------------------------------------------------
def two_pass_plan(num_frames_total: int, input_indices: Sequence[int], c2ws, T: int = 21,
                  chunk_strategy: str = "interp", options: dict | None = None, task: str = "img2trajvid",
                  first_pass_strategy: str = "gt-nearest", refine_anchors: bool = True):
    """Convenience for the multi-GPU driver: anchors, first-pass windows (serial) and second-pass windows
    (independent units) of one trajectory, composed from the functions above the way `run_one_scene` does
    (reference eval.py:1653-1885: first pass `chunk_strategy_first_pass` = "gt-nearest" by default; the second pass
    conditions on the argsorted [inputs + anchors] and targets EVERY non-input frame, anchors included).
    `refine_anchors=False`: second pass without the anchors (this package's cheaper variant).
    `c2ws`: (num_frames_total, 4, 4); `input_indices` must come first in order."""
    opts = {"sampler_verbose": False, **(options or {}), "chunk_strategy": chunk_strategy}
    vd = {"T": T, "options": opts}
    n_in = len(input_indices)
    n_prior = infer_prior_stats(T, n_in, num_frames_total - n_in, vd)
    T1, T2 = vd["T"] if isinstance(vd["T"], (list, tuple)) else (T, T)
    anchors = [int(v) for v in infer_prior_inds(c2ws, n_prior, list(input_indices), opts)]
    ins = list(input_indices)
    pass1 = chunk_input_and_test(T1, c2ws[ins], c2ws[anchors], [float(i) for i in ins], [float(a) for a in anchors],
                                 opts, task=task, chunk_strategy=first_pass_strategy, gt_input_inds=list(range(n_in)))
    order = np.argsort(ins + anchors).tolist()
    pool = [(ins + anchors)[o] for o in order]
    rest = [i for i in range(num_frames_total) if i not in ins and (refine_anchors or i not in anchors)]
    pass2 = chunk_input_and_test(T2, c2ws[pool], c2ws[rest], [float(i) for i in pool],
                                 [float(r) for r in rest], opts, task=task, chunk_strategy=chunk_strategy,
                                 gt_input_inds=[order.index(i) for i in range(n_in)])
    return {"anchors": anchors, "pass1": pass1, "pass2": pass2, "T": [T1, T2]}
