"""Two-pass trajectory driver (`seva.pipeline`, BASELINE config 4) on the HIP path, one GPU: a 50-view orbit through the tiny
UNet and the FULL-SIZE 168-view plan through the 1.3B network at 576x576 -- every frame generated, the input frame untouched,
deterministic, and window results independent of the execution order (what makes sharding over ranks a pure scheduling
decision; the 2-rank run itself is tested on gloo)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from seva import _native
    _native.load()
    return torch.device("cuda:0")


def test_trajectory_on_gpu_tiny(dev):
    from test_model_gpu import _build
    from seva import pipeline
    from seva import synthetic as synth
    from seva.model import SGMWrapper
    net, _ = _build("tiny", dev)
    n, hw, T = 50, 16, 21
    c2ws, Ks = synth.orbit_c2w(n), synth.default_K(n)
    g = torch.Generator().manual_seed(3)
    lat = (torch.randn(1, 4, hw, hw, generator=g) * 0.9).to(dev)
    tok = torch.randn(1024, generator=g)
    tok = (tok / tok.norm()).to(dev)
    wrap = SGMWrapper(net)
    plan = pipeline.plan_trajectory(c2ws, [0], T=T)
    assert len(plan.pass1) >= 1 and len(plan.pass2) >= 2
    kw = dict(clip_token=tok, T=T, num_steps=3, device=dev, plan=plan)
    with torch.no_grad():
        a = pipeline.run_trajectory(wrap, lat, c2ws, Ks, [0], **kw)
        b = pipeline.run_trajectory(wrap, lat, c2ws, Ks, [0], **kw)
    assert a["latents"].shape == (n, 4, hw, hw) and torch.isfinite(a["latents"]).all()
    assert torch.equal(a["latents"], b["latents"])
    assert torch.equal(a["latents"][0], lat[0])
    # a pass-2 window run on its own (as another rank would) reproduces its frames bit for bit
    win = plan.pass2[-1]
    latents_of = {0: lat[0], **a["anchor_latents"]}
    g0 = torch.Generator().manual_seed(23)
    noises = [torch.randn((T, 4, hw, hw), generator=g0) for _ in range(len(plan.pass1) + len(plan.pass2))]
    with torch.no_grad():
        z = pipeline.run_window(win, latents_of, wrap, c2ws, Ks, hw=(hw, hw), num_steps=3, cfg=2.0, cfg_min=1.2, guider=1,
                                camera_scale=2.0, noise=noises[win.global_index],
                                step_seed=(23 * 1000003 + 7919 * (win.global_index + 1)) & 0x7FFFFFFFFFFF, clip_token=tok, device=dev)
    for fid, slot in zip(win.target_ids, win.target_slots):
        assert torch.equal(z[slot], a["latents"][fid])


def test_trajectory_168_views_full_size_one_gpu(dev):
    """BASELINE config 4's workload on ONE GPU: the 168-view orbit (1 input view, the reference's two-pass plan: 20 anchors,
    1 + 10 windows of 21 views) through the 1.3B network at 576x576 (latent 72x72), 2 sampler steps per window (the plan, the
    window assembly, the anchor hand-off and the reassembly are step-count independent; 50 steps is `bench.py --trajectory 168`).
    Every non-input frame is generated (anchors twice: pass 1, then again in pass 2 like the reference), everything finite,
    and a second-pass window re-run on its own -- as another rank would run it -- reproduces its frames bit for bit."""
    from test_model_gpu import _build
    from seva import pipeline
    from seva import synthetic as synth
    from seva.model import SGMWrapper
    net, _ = _build("full", dev)
    n, hw, T, steps = 168, 72, 21, 2
    c2ws, Ks = synth.orbit_c2w(n), synth.default_K(n)
    g = torch.Generator().manual_seed(23)
    lat = (torch.randn(1, 4, hw, hw, generator=g) * 0.18215 * 5.0).to(dev)
    tok = torch.randn(1024, generator=g)
    tok = (tok / tok.norm()).to(dev)
    wrap = SGMWrapper(net)
    plan = pipeline.plan_trajectory(c2ws, [0], T=T)
    assert len(plan.anchor_ids) == 20 and len(plan.pass1) == 1 and len(plan.pass2) == 10
    assert sorted(f for w in plan.pass2 for f in w.target_ids) == list(range(1, n))
    timers = {}
    with torch.no_grad():
        res = pipeline.run_trajectory(wrap, lat, c2ws, Ks, [0], clip_token=tok, T=T, num_steps=steps, device=dev, plan=plan,
                                      timers=timers)
    z = res["latents"]
    assert z.shape == (n, 4, hw, hw) and torch.isfinite(z).all() and torch.equal(z[0], lat[0])
    assert float(z[1:].abs().mean()) > 1e-3 and all(float(z[f].abs().max()) > 0 for f in range(1, n))
    # anchors: final frames are second-pass samples, distinct from the first-pass ones pass 2 conditioned on
    a0 = plan.anchor_ids[3]
    assert not torch.equal(z[a0], res["anchor_latents"][a0])
    win = plan.pass2[4]
    latents_of = {0: lat[0], **res["anchor_latents"]}
    g0 = torch.Generator().manual_seed(23)
    noises = [torch.randn((T, 4, hw, hw), generator=g0) for _ in range(len(plan.pass1) + len(plan.pass2))]
    with torch.no_grad():
        zz = pipeline.run_window(win, latents_of, wrap, c2ws, Ks, hw=(hw, hw), num_steps=steps, cfg=2.0, cfg_min=1.2, guider=1,
                                 camera_scale=2.0, noise=noises[win.global_index],
                                 step_seed=(23 * 1000003 + 7919 * (win.global_index + 1)) & 0x7FFFFFFFFFFF, clip_token=tok,
                                 device=dev)
    for fid, slot in zip(win.target_ids, win.target_slots):
        assert torch.equal(zz[slot], z[fid])
    print(f"\n168-view trajectory, 1.3B @ 576x576, {steps} steps/window: {len(plan.pass1)}+{len(plan.pass2)} windows in "
          f"{timers['gather'] - timers['start']:.2f} s")


def _cfg_split_worker(rank, world, port, q, n, hw, T, steps):
    import os
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from test_model_gpu import _build
    from seva import pipeline
    from seva import synthetic as synth
    from seva.model import SGMWrapper
    dev = torch.device("cuda:0")  # both ranks on the one card of this box: gloo carries the collectives (RCCL refuses that)
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    net, _ = _build("tiny", dev)
    c2ws, Ks = synth.orbit_c2w(n), synth.default_K(n)
    g = torch.Generator().manual_seed(3)
    lat = (torch.randn(1, 4, hw, hw, generator=g) * 0.9).to(dev)
    tok = torch.randn(1024, generator=g)
    tok = (tok / tok.norm()).to(dev)
    with torch.no_grad():
        res = pipeline.run_trajectory(SGMWrapper(net), lat, c2ws, Ks, [0], clip_token=tok, T=T, num_steps=steps, device=dev,
                                      cfg_split=True)
    q.put((rank, res["latents"].cpu().numpy() if "latents" in res else None,
           {f: v.cpu().numpy() for f, v in res["anchor_latents"].items()} if "anchor_latents" in res else None))
    dist.barrier()
    dist.destroy_process_group()


def test_cfg_split_on_the_hip_path_equals_single_process(dev):
    """CFG-split on REAL kernels: two ranks (both on this box's one card, gloo for the collectives) run the first-pass window
    and -- the trajectory has an odd number of second-pass windows -- the last second-pass window as CFG halves with one
    all-gather per step; the trajectory equals the single-process one BIT FOR BIT.  This is the property that makes the split a
    pure scheduling decision: a half batch is bitwise the corresponding half of the full batch through every HIP kernel."""
    import socket
    import torch.multiprocessing as mp
    from test_model_gpu import _build
    from seva import pipeline
    from seva import synthetic as synth
    from seva.model import SGMWrapper
    n, hw, T, steps = 100, 16, 21, 3
    c2ws, Ks = synth.orbit_c2w(n), synth.default_K(n)
    plan = pipeline.plan_trajectory(c2ws, [0], T=T)
    sched = pipeline.second_pass_schedule(len(plan.pass2), 2, True)
    assert any(len(ranks) == 2 for r in sched for _, ranks in r), sched
    net, _ = _build("tiny", dev)
    g = torch.Generator().manual_seed(3)
    lat = (torch.randn(1, 4, hw, hw, generator=g) * 0.9).to(dev)
    tok = torch.randn(1024, generator=g)
    tok = (tok / tok.norm()).to(dev)
    with torch.no_grad():
        ref_res = pipeline.run_trajectory(SGMWrapper(net), lat, c2ws, Ks, [0], clip_token=tok, T=T, num_steps=steps, device=dev)
    ref = ref_res["latents"].cpu()
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    import os
    from conftest import PKG, ROOT
    here = os.path.dirname(os.path.abspath(__file__))
    os.environ["PYTHONPATH"] = os.pathsep.join([PKG, ROOT, here, os.environ.get("PYTHONPATH", "")])  # for the spawned workers
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_cfg_split_worker, args=(r, 2, port, q, n, hw, T, steps)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=600) for _ in range(2)), key=lambda t: t[0])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    got = torch.from_numpy(res[0][1])
    assert res[1][1] is None and torch.isfinite(got).all()
    if not torch.equal(got, ref):  # say where: the first-pass anchors (the CFG-split window) or a second-pass window
        anc = {f: float((torch.from_numpy(v) - ref_res["anchor_latents"][f].cpu()).abs().max()) for f, v in res[0][2].items()}
        per_win = [(i, max(float((got[f] - ref[f]).abs().max()) for f in w.target_ids)) for i, w in enumerate(plan.pass2)]
        raise AssertionError(f"anchors (pass 1) max diff {max(anc.values()):.3e}; second-pass windows {per_win}")
