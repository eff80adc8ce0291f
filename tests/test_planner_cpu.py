"""Chunk planner: `seva.planner` against known-answer layouts produced by the reference's own
planner (tests/golden/g8_planner.json, oracle/make_goldens_next.py).  Host logic only; no GPU."""
import json
import os

import numpy as np
import pytest
import torch

from seva import planner as P

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def cases():
    with open(os.path.join(GOLD, "g8_planner.json")) as f:
        return json.load(f)


@pytest.fixture(scope="module")
def cams():
    z = np.load(os.path.join(GOLD, "g8_planner_cams.npz"))
    return {k: torch.from_numpy(z[k]) for k in z.files}


def test_pad_indices_known_answers(cases):
    for c in cases["pad_indices"]:
        a, b, m_in, m_te = P.pad_indices(list(c["input_indices"]), list(c["test_indices"]), c["T"], c["padding_mode"])
        assert [list(map(int, a)), list(map(int, b)), m_in.tolist(), m_te.tolist()] == c["out"], c
    with pytest.raises(AssertionError):
        P.pad_indices([0], [1], 4, "first")


def test_assemble_roundtrip():
    inp, tst = torch.arange(2.0)[:, None] + 100, torch.arange(3.0)[:, None]
    a, b, m_in, m_te = P.pad_indices([0, 5], [1, 2, 3], 8, "last")
    out = P.assemble(inp, tst, m_in, m_te)
    assert out[:, 0].tolist() == [100.0, 0.0, 1.0, 2.0, 101.0, 101.0, 101.0, 101.0]


def test_infer_prior_stats_known_answers(cases):
    assert len(cases["infer_prior_stats"]) > 500
    for c in cases["infer_prior_stats"]:
        T = c["T"]
        vd = {"T": list(T) if isinstance(T, list) else T, "options": dict(c["options"])}
        assert P.infer_prior_stats(vd["T"], c["num_input_frames"], c["num_total_frames"], vd) == c["out"], c
        assert vd["T"] == c["T_after"], c


def test_infer_prior_inds_and_relative_inds(cases):
    for c in cases["infer_prior_inds"]:
        out = P.infer_prior_inds(torch.zeros(c["n"], 4, 4), c["num_prior_frames"], c["input_frame_indices"],
                                 {"chunk_strategy": c["chunk_strategy"]})
        assert [int(v) for v in out] == c["out"], c
    for c in cases["compute_relative_inds"]:
        out = P.compute_relative_inds(np.array(c["source"]), np.array(c["target"]))
        assert np.allclose(out, c["out"], rtol=0, atol=1e-12)


def test_chunk_input_and_test_known_answers(cases, cams):
    seen = set()
    for c in cases["chunk_input_and_test"]:
        key = c["cam_key"]
        out = P.chunk_input_and_test(c["T"], cams[f"in_{key}"], cams[f"te_{key}"], list(c["input_ords"]),
                                     list(c["test_ords"]), {"sampler_verbose": False, **c["options"]}, task=c["task"],
                                     chunk_strategy=c["chunk_strategy"], gt_input_inds=list(c["gt_input_inds"]))
        chunks, a, b, cc, d = out
        exp = c["out"]
        assert chunks == exp["chunks"], (c["chunk_strategy"], c["task"], c["T"], c["M"], c["N"], c["options"])
        assert [a, b, cc, d] == [exp["input_inds"], exp["input_sels"], exp["test_inds"], exp["test_sels"]]
        assert all(len(w) == c["T"] for w in chunks)
        seen.add(c["chunk_strategy"])
    assert seen == {"gt", "gt-ltr", "gt-nearest", "nearest", "nearest-gt", "nearest-2", "interp", "interp-gt"}


def test_chunk_error_behaviour(cams):
    k = next(k for k in cams if k.startswith("in_"))
    c_in, c_te = cams[k], cams["te_" + k[3:]]
    opts = {"sampler_verbose": False}
    with pytest.raises(NotImplementedError):
        P.chunk_input_and_test(8, c_in, c_te, None, None, opts, chunk_strategy="bogus")
    with pytest.raises(AssertionError):  # interp needs orders
        P.chunk_input_and_test(8, c_in, c_te, None, None, opts, chunk_strategy="interp")
    with pytest.raises(AssertionError):  # gt: every input must be a gt input
        P.chunk_input_and_test(8, c_in, c_te, None, None, opts, chunk_strategy="gt", gt_input_inds=[])


def test_two_pass_plan_of_168_views(cases, cams):
    """1 input + 167 targets, T=21, interp -> 20 anchors, 1 serial first-pass window, 10 independent
    second-pass windows whose neighbours share their boundary anchor."""
    g = cases["plan168"]  # (this golden: first pass "gt", second pass without the anchors = refine_anchors=False)
    plan = P.two_pass_plan(168, [0], cams["plan168"], T=21, chunk_strategy="interp", first_pass_strategy="gt",
                           refine_anchors=False)
    assert plan["anchors"] == g["prior_inds"] and len(plan["anchors"]) == g["num_prior_frames"] == 20
    assert plan["pass1"][0] == g["pass1_chunks"] and len(plan["pass1"][0]) == 1
    assert plan["pass2"][0] == g["pass2_chunks"] and len(plan["pass2"][0]) == 10
    wins = plan["pass2"][0]
    for w0, w1 in zip(wins, wins[1:]):
        last_anchor = [s for s in w0 if s.startswith("!")][-1]
        assert w1[0] == last_anchor
    from seva.distributed import shard_windows
    shards = [shard_windows(len(wins), r, 8) for r in range(8)]
    assert sorted(i for s in shards for i in s) == list(range(10)) and max(len(s) for s in shards) == 2


def test_two_pass_plan_is_the_references_run_one_scene_composition():
    """Default `two_pass_plan` / `pipeline.plan_trajectory` == what the reference's run_one_scene composes in its 2-pass branch
    (eval.py:1653-1885): first pass "gt-nearest", second pass over the argsorted [inputs + anchors] pool with EVERY
    non-input frame as a target (the anchors are generated again).  Golden: window lists produced by the reference's own
    functions in that composition (tests/golden/g10_two_pass_plans.json, oracle/make_goldens_next.py:g10_two_pass)."""
    from seva import pipeline
    with open(os.path.join(GOLD, "g10_two_pass_plans.json")) as f:
        gold = json.load(f)
    z = np.load(os.path.join(GOLD, "g10_two_pass_cams.npz"))
    assert set(gold) >= {"orbit168", "orbit168_gt", "two_in80", "three60", "orbit300", "orbit100_nearest"}
    for name, g in gold.items():
        c2ws = torch.from_numpy(z[name])
        ins, n = g["input_ids"], g["n"]
        plan = P.two_pass_plan(n, ins, c2ws, T=g["T"], chunk_strategy=g["chunk_strategy"],
                               first_pass_strategy=g["first_pass_strategy"])
        assert plan["anchors"] == g["prior_inds"], name
        for k, (ours, ref) in {"pass1": (plan["pass1"], g["pass1"]), "pass2": (plan["pass2"], g["pass2"])}.items():
            assert list(ours[0]) == ref["chunks"], (name, k)
            assert [list(ours[1]), list(ours[2]), list(ours[3]), list(ours[4])] == \
                   [ref["input_inds"], ref["input_sels"], ref["test_inds"], ref["test_sels"]], (name, k)
        # the driver's typed plan: same windows, in trajectory frame ids
        tp = pipeline.plan_trajectory(c2ws, ins, T=g["T"], chunk_strategy=g["chunk_strategy"],
                                      first_pass_strategy=g["first_pass_strategy"])
        assert tp.anchor_ids == g["prior_inds"] and tp.pass1_serial == (g["first_pass_strategy"] != "gt")
        order = np.argsort(ins + g["prior_inds"]).tolist()
        pool = [(ins + g["prior_inds"])[o] for o in order]
        test = [i for i in range(n) if i not in ins]
        assert len(tp.pass2) == len(g["pass2"]["chunks"])
        for w, ii, ti, ts in zip(tp.pass2, g["pass2"]["input_inds"], g["pass2"]["test_inds"], g["pass2"]["test_sels"]):
            assert w.source_ids == [pool[j] for j in ii] and w.target_ids == [test[j] for j in ti] and w.target_slots == list(ts)
        # every non-input frame is a second-pass target exactly once; every anchor is among them
        tg = [f for w in tp.pass2 for f in w.target_ids]
        assert sorted(tg) == test and set(g["prior_inds"]) <= set(tg)
