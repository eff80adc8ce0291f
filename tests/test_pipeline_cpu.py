"""Config-4 driver (`seva.pipeline`): plan coverage of the 168-view orbit and, on gloo with two ranks, that a
SHARDED run of the two-pass trajectory reproduces the single-process run bit for bit (per-window RNG, anchor exchange,
round-robin gather).  The HIP operators are replaced by tests/fake_ops.py (host logic only; no GPU here) and the network
by a cheap deterministic stand-in -- what is under test is the orchestration, not the denoiser."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _patch_cpu(monkeypatch=None):
    """Main process: through pytest's monkeypatch (undone after the test); spawned workers: plain setattr."""
    import fake_ops
    import seva.ops as ops
    from seva import geometry
    from seva import sampling as S
    put = setattr if monkeypatch is None else monkeypatch.setattr
    for name in dir(fake_ops):
        if not name.startswith("_") and callable(getattr(fake_ops, name)) and hasattr(ops, name):
            put(ops, name, getattr(fake_ops, name))
    put(S, "_need_gpu", lambda *a: None)
    put(geometry, "_compute_device", lambda *a: torch.device("cpu"))


def _scene(n=168, hw=8):
    from seva import synthetic as synth
    c2ws = synth.orbit_c2w(n)
    Ks = synth.default_K(n)
    g = torch.Generator().manual_seed(3)
    lat = torch.randn(1, 4, hw, hw, generator=g)
    tok = torch.randn(1024, generator=g)
    return c2ws, Ks, lat, tok / tok.norm()


def _fake_net(x, t, cond, num_frames):
    """Deterministic, frame-coupled stand-in for SGMWrapper(model): depends on x, the timestep index, the Pluecker
    maps, the mask channel and (through the frame mean) on every frame of the window."""
    n = x.shape[0]
    xs = x.view(n // num_frames, num_frames, *x.shape[1:])
    mix = xs.mean(1, keepdim=True).expand_as(xs).reshape_as(x)
    return 0.3 * x + 0.2 * mix + 0.05 * cond["concat"][:, 1:5] + 0.01 * cond["concat"][:, :1] + 1e-4 * t.view(-1, 1, 1, 1).float()


def _run(group=None, n=168, **kw):
    from seva import pipeline
    c2ws, Ks, lat, tok = _scene(n)
    return pipeline.run_trajectory(_fake_net, lat, c2ws, Ks, [0], clip_token=tok, T=21, num_steps=3, seed=23,
                                   device="cpu", group=group, **kw)


@pytest.mark.parametrize("refine", [True, False])
def test_plan_covers_the_168_view_orbit(refine):
    """1 input + 167 targets, T=21, `interp`: 20 anchors, 1 first-pass window, 10 second-pass windows whose neighbours share
    their boundary anchor (reference planner facts).  Default (the reference's composition): the second pass
    generates every non-input frame, anchors included; `refine_anchors=False`: every frame generated exactly once."""
    from seva import pipeline
    c2ws, _, _, _ = _scene()
    plan = pipeline.plan_trajectory(c2ws, [0], T=21, refine_anchors=refine)
    assert len(plan.anchor_ids) == 20 and len(plan.pass1) == 1 and len(plan.pass2) == 10
    assert plan.anchor_ids == sorted(set(plan.anchor_ids)) and plan.anchor_ids[-1] == 167 and 0 not in plan.anchor_ids
    assert plan.pass1_serial  # the reference's default first-pass strategy, "gt-nearest", chains its windows
    assert not pipeline.plan_trajectory(c2ws, [0], T=21, first_pass_strategy="gt").pass1_serial
    gen2 = [f for w in plan.pass2 for f in w.target_ids]
    if refine:
        assert sorted(gen2) == list(range(1, 168))
    else:
        assert sorted(gen2 + plan.anchor_ids) == list(range(1, 168))
    assert sorted(f for w in plan.pass1 for f in w.target_ids) == plan.anchor_ids
    for a, b in zip(plan.pass2[:-1], plan.pass2[1:]):
        assert a.source_ids[-1] == b.source_ids[0]  # the shared "overlap" anchor
    for w in plan.pass1 + plan.pass2:
        assert len(w.slot_frame) == 21 and all(w.slot_is_input[s] for s in w.source_slots)
        assert [w.slot_frame[s] for s in w.target_slots] == w.target_ids
    assert [w.global_index for w in plan.pass1 + plan.pass2] == list(range(11))


def test_single_process_trajectory_is_deterministic_and_window_order_independent(monkeypatch):
    _patch_cpu(monkeypatch)
    a = _run()
    b = _run()
    assert torch.equal(a["latents"], b["latents"]) and torch.isfinite(a["latents"]).all()
    assert a["latents"].shape == (168, 4, 8, 8)
    # input frame untouched; anchors = second-pass samples by default (reference), first-pass samples with refine_anchors=False
    _, _, lat, _ = _scene()
    assert torch.equal(a["latents"][0], lat[0])
    assert float(a["latents"][1:].abs().mean()) > 0
    c = _run(refine_anchors=False)
    anchors = a["plan"].anchor_ids
    others = [f for f in range(1, 168) if f not in anchors]
    assert not torch.equal(a["latents"][anchors], c["latents"][anchors])
    assert torch.isfinite(c["latents"]).all() and c["latents"][others].abs().mean() > 0


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q, first_pass, extra=None):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    _patch_cpu()
    timers = {}
    res = _run(first_pass_strategy=first_pass, timers=timers, **(extra or {}))
    pairs = None
    if (extra or {}).get("cfg_split"):
        from seva import pipeline
        # the pair groups every rank holds, in creation order (new_group is collective: all ranks must create them identically)
        pairs = [tuple(dist.get_process_group_ranks(g)) for g in pipeline.cfg_pair_groups()]
    q.put((rank, res["latents"].numpy() if "latents" in res else None, sorted(timers), pairs))
    dist.destroy_process_group()


@pytest.mark.parametrize("first_pass", ["gt", "gt-nearest"])
def test_two_rank_sharded_trajectory_equals_sequential(first_pass, monkeypatch):
    from conftest import PKG
    here = os.path.dirname(os.path.abspath(__file__))
    os.environ["PYTHONPATH"] = os.pathsep.join([PKG, here, os.environ.get("PYTHONPATH", "")])
    _patch_cpu(monkeypatch)
    ref = _run(first_pass_strategy=first_pass)["latents"]
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, first_pass)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=300) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res[1][1] is None  # only rank 0 assembles the trajectory
    got = torch.from_numpy(res[0][1])
    assert torch.equal(got, ref), float((got - ref).abs().max())
    assert res[0][2] == ["exchange", "gather", "pass1", "pass2", "start", "windows"]


def test_reference_style_rgb_handoff_and_per_window_clip_token(monkeypatch):
    """handoff='rgb' + conditioner=: anchors go decode -> encode between the passes and every window's CLIP token is the mean
    embedding of ITS conditioning views (reference eval.py:1248, 1820-1829), here with linear stand-ins for the VAE / CLIP."""
    from seva import pipeline
    _patch_cpu(monkeypatch)
    c2ws, Ks, lat, _ = _scene(n=50)

    class FakeAE:
        def decode(self, z):  # (k,4,h,w) -> (k,3,h,w)
            return torch.tanh(z[:, :3] * 0.5 + 0.1 * z[:, 3:4])

        def encode(self, x):
            return torch.cat([x * 1.5, x.mean(1, keepdim=True)], 1)

    seen = []

    def fake_clip(imgs):  # (k,3,h,w) -> (k,1024)
        seen.append(imgs.shape[0])
        g = torch.Generator().manual_seed(0)
        proj = torch.randn(3 * 8 * 8, 1024, generator=g) * 0.05
        return imgs.reshape(imgs.shape[0], -1) @ proj

    rgb_in = FakeAE().decode(lat)
    res = pipeline.run_trajectory(_fake_net, lat, c2ws, Ks, [0], T=21, num_steps=2, seed=5, device="cpu", ae=FakeAE(),
                                  handoff="rgb", conditioner=fake_clip, input_rgb=rgb_in)
    plan = res["plan"]
    assert res["latents"].shape == (50, 4, 8, 8) and torch.isfinite(res["latents"]).all() and "rgb" in res
    # one conditioner call per window, over exactly that window's conditioning views
    assert seen == [len(w.source_ids) for w in plan.pass1 + plan.pass2]
    # anchors went through the RGB round trip: what pass 2 conditioned on is encode(decode(.)), whose 4th channel is the
    # mean of the first three divided by 1.5 under this stand-in
    # (visible in the final latents only when the second pass does not regenerate the anchors)
    seen.clear()
    res = pipeline.run_trajectory(_fake_net, lat, c2ws, Ks, [0], T=21, num_steps=2, seed=5, device="cpu", ae=FakeAE(),
                                  handoff="rgb", conditioner=fake_clip, input_rgb=rgb_in, refine_anchors=False)
    a = res["plan"].anchor_ids[0]
    assert torch.allclose(res["latents"][a][3], res["latents"][a][:3].mean(0) / 1.5, atol=1e-6)


def test_second_pass_schedule_with_cfg_split():
    """10 windows on 8 ranks: one round of 8 whole windows, then the 2 leftover windows on the pairs (0,1) and (2,3); without
    cfg_split two whole-window rounds.  Every window exactly once; a pair is two adjacent ranks, even rank first."""
    from seva.pipeline import second_pass_schedule
    s = second_pass_schedule(10, 8, True)
    assert [len(r) for r in s] == [8, 2] and s[1] == [(8, (0, 1)), (9, (2, 3))]
    assert second_pass_schedule(10, 8, False) == [[(i, (i,)) for i in range(8)], [(8, (0,)), (9, (1,))]]
    for n in range(1, 40):
        for world in (1, 2, 4, 6, 8):
            for split in (False, True):
                sch = second_pass_schedule(n, world, split)
                assert sorted(i for r in sch for i, _ in r) == list(range(n))
                for r in sch:
                    used = [q for _, ranks in r for q in ranks]
                    assert len(used) == len(set(used)) and all(0 <= q < world for q in used)
                    assert all(len(ranks) == 1 or (ranks[0] % 2 == 0 and ranks[1] == ranks[0] + 1) for _, ranks in r)
    # ceiling of BASELINE config 4 on 8 GPUs in window times: 1 + 2 without, 0.5 + 1 + 0.5 with CFG-split
    assert sum(0.5 if len(r[0][1]) == 2 else 1.0 for r in second_pass_schedule(10, 8, True)) == 1.5


@pytest.mark.parametrize("world,n", [(2, 100), (4, 168), (8, 168)])
def test_cfg_split_trajectory_equals_single_process(world, n, monkeypatch):
    """CFG-split: the first-pass window on the pair (0,1) and the leftover second-pass round on pairs,
    each rank of a pair running one half of every CFG batch with one all-gather per step -- the trajectory is bit for bit
    the single-process one (gloo; world 2: an odd number of second-pass windows so that the last one is split;
    world 4: the 168-view plan, 10 windows = 4 + 4 + a pair round of 2; world 8 = the target machine's rank count: the
    168-view plan as one round of 8 whole windows + a pair round of 2 on (0,1), (2,3)).  Every rank creates the same pair
    groups in the same order."""
    from conftest import PKG
    from seva import pipeline
    here = os.path.dirname(os.path.abspath(__file__))
    os.environ["PYTHONPATH"] = os.pathsep.join([PKG, here, os.environ.get("PYTHONPATH", "")])
    _patch_cpu(monkeypatch)
    c2ws, _, _, _ = _scene(n)
    n2 = len(pipeline.plan_trajectory(c2ws, [0], T=21).pass2)
    sched = pipeline.second_pass_schedule(n2, world, True)
    assert any(len(ranks) == 2 for r in sched for _, ranks in r), (n2, sched)
    ref = _run(n=n)["latents"]
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, "gt-nearest", dict(n=n, cfg_split=True))) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=600) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    got = torch.from_numpy(res[0][1])
    assert all(r[1] is None for r in res[1:])
    assert torch.equal(got, ref), float((got - ref).abs().max())
    # every rank holds world / 2 pair groups, created in the same order; entry rank // 2 is the rank's own pair (the entries of
    # pairs a rank does not belong to are non-member handles)
    for rank, _, _, pairs in res:
        assert len(pairs) == world // 2 and pairs[rank // 2] == (2 * (rank // 2), 2 * (rank // 2) + 1), (rank, pairs)
