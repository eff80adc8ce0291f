"""GPU parity of the HIP-backed `seva.model` / `seva.sampling` against the CPU oracle and the
golden vectors generated from the reference (tests/golden, oracle/make_goldens.py).

Tolerance: BASELINE.json north_star asks for 1e-3 relative (fp16) per denoised latent; the metric
is rel-L2 = ||out - ref||_2 / ||ref||_2 over the whole tensor.  fp16 operand rounding alone gives
~1e-3 through the 1.3 B UNet, so network-level checks use NET_TOL below and the
measured value is printed.
"""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

from conftest import load_golden, rel_l2

NET_TOL = 1e-3      # north_star tolerance: rel-L2 per denoised latent, fp16 operands vs fp32 oracle
LAYER_TOL = 1e-3    # every layer output inside the network


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from seva import _native
    _native.load()
    return torch.device("cuda:0")


def _shapes(tag):
    g = load_golden(f"g0_keys_{tag}")
    return {str(k): tuple(int(s) for s in str(v).split(",")) for k, v in zip(g["keys"], g["shapes"])}


def _build(tag, dev, seed=0):
    from seva import synthetic as synth
    from seva.model import Seva, SevaParams
    params = SevaParams() if tag == "full" else SevaParams(model_channels=64)
    sd = synth.synth_state_dict(_shapes(tag), seed)
    with torch.device("meta"):
        net = Seva(params)
    net.load_state_dict(sd, strict=True, assign=True)
    return net.to(dev).eval(), sd


@pytest.fixture(scope="module")
def tiny(dev):
    return _build("tiny", dev)


def _layer_table(engine, trace, n):
    rows = []
    for key, t in engine.arena.bufs.items():
        name = key[0]
        if not name.startswith("out:"):
            continue
        pfx = name[4:]
        if pfx not in trace:
            continue
        ref = trace[pfx]  # NCHW
        got = t.float().cpu().view(ref.shape[0], ref.shape[2], ref.shape[3], ref.shape[1]).permute(0, 3, 1, 2)
        rows.append((pfx, rel_l2(got, ref), float(ref.abs().max())))
    return rows


def test_tiny_forward_vs_golden_and_layers(dev, tiny):
    from oracle import seva_ref as O
    from seva.model import SGMWrapper
    net, sd = tiny
    g = load_golden("g3_tiny_forward")
    T = int(g["T"])
    c = {k: g[k].to(dev) for k in ("crossattn", "concat", "dense_vector")}
    y = SGMWrapper(net)(g["x"].to(dev), g["t"].to(dev), c, num_frames=T)
    torch.cuda.synchronize()
    trace = {}
    c_cpu = {k: g[k] for k in ("crossattn", "concat", "dense_vector")}
    O.sgm_wrapper_forward(sd, g["x"], g["t"], c_cpu, T, trace=trace)
    rows = _layer_table(net.engine(), trace, g["x"].shape[0])
    print("\nlayer-wise rel-L2 (HIP vs oracle):")
    for pfx, err, mag in rows:
        print(f"  {pfx:28s} {err:.3e}  |ref|max {mag:.2f}")
    err = rel_l2(y.cpu(), g["y"])
    print(f"tiny forward vs reference golden: rel-L2 {err:.3e}")
    assert len(rows) >= 40
    bad = [(p, e) for p, e, _ in rows if not e < LAYER_TOL]
    assert not bad, f"layers beyond tolerance: {bad[:5]}"
    assert err < NET_TOL


@pytest.mark.parametrize("T,h,w", [(3, 8, 24), (2, 8, 8), (5, 16, 8)])
def test_tiny_forward_odd_shapes(dev, tiny, T, h, w):
    """Ragged sizes: M-tails in every GEMM, partial attention tiles, non-square images."""
    from oracle import seva_ref as O
    net, sd = tiny
    g = torch.Generator().manual_seed(T * 100 + h)
    n = 2 * T
    x = torch.randn(n, 11, h, w, generator=g)
    t = torch.randint(0, 1000, (n,), generator=g)
    y = torch.randn(n, 1, 1024, generator=g)
    dense = torch.randn(n, 6, h, w, generator=g)
    out = net(x.to(dev), t.to(dev), y.to(dev), dense.to(dev), num_frames=T)
    ref = O.seva_forward(sd, x, t, y, dense, T)
    err = rel_l2(out.cpu(), ref)
    print(f"T={T} {h}x{w}: rel-L2 {err:.3e}")
    assert err < NET_TOL


def test_tiny_forward_long_context(dev, tiny):
    """Context length 3 exercises the general cross-attention path (no single-token collapse)."""
    from oracle import seva_ref as O
    net, sd = tiny
    T, h, w = 2, 8, 8
    g = torch.Generator().manual_seed(5)
    n = 2 * T
    x, t = torch.randn(n, 11, h, w, generator=g), torch.randint(0, 1000, (n,), generator=g)
    y, dense = torch.randn(n, 3, 1024, generator=g), torch.randn(n, 6, h, w, generator=g)
    out = net(x.to(dev), t.to(dev), y.to(dev), dense.to(dev), num_frames=T)
    ref = O.seva_forward(sd, x, t, y, dense, T)
    err = rel_l2(out.cpu(), ref)
    print(f"context length 3: rel-L2 {err:.3e}")
    assert err < NET_TOL


def test_forward_is_deterministic_and_graph_replayable(dev, tiny):
    from seva import ops
    net, _ = tiny
    g = load_golden("g3_tiny_forward")
    T = int(g["T"])
    args = (torch.cat([g["x"], g["concat"]], 1).to(dev), g["t"].to(dev), g["crossattn"].to(dev),
            g["dense_vector"].to(dev))
    a = net(*args, num_frames=T).clone()
    b = net(*args, num_frames=T).clone()
    assert torch.equal(a, b)
    eng = net.engine()
    before = eng.arena.nbytes()
    out = torch.empty_like(a)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        x, concat = args[0][:, :4].contiguous(), args[0][:, 4:].contiguous()
        eng.forward(x, concat, args[1], args[2], args[3], T, out=out)  # warm on this stream
        gr = ops.Graph()
        gr.capture_begin()
        eng.forward(x, concat, args[1], args[2], args[3], T, out=out)
        gr.capture_end()
        out.zero_()
        gr.launch()
        gr.launch()
    s.synchronize()
    assert eng.arena.nbytes() == before, "forward allocated during replay"
    assert torch.equal(out, a)


def test_forward_is_repeatable_while_a_second_stream_keeps_the_card_busy(dev):
    """Round 3: with other work on the card (a second stream here, a second process in the CFG-split test) two kernels -- the modulated
    GroupNorm apply and the fused feed-forward's LayerNorm prologue -- computed a stale value in the last 16 lanes of a wave now and then
    (27 of 149 forwards differed; alone every run was bit-equal).  Fixed in the kernels (tests/test_isa_concurrency_cpu.py, DESIGN.md
    section 4); this is the run-time check: 60 forwards under load, every one bit-equal to the first."""
    import threading
    import time
    net, _ = _build("tiny", dev)
    eng = net.engine()
    eng.use_graph = False
    T, hw = 21, 16

    def inputs(seed):
        g = torch.Generator().manual_seed(seed)
        n = 2 * T
        return ((torch.randn(n, 4, hw, hw, generator=g) * 10).to(dev), torch.randn(n, 7, hw, hw, generator=g).to(dev),
                torch.full((n,), 700, dtype=torch.int64, device=dev), torch.randn(n, 1, 1024, generator=g).to(dev),
                torch.randn(n, 6, hw * 8, hw * 8, generator=g).to(dev))

    stop = []
    started = threading.Event()

    def load():
        s2 = torch.cuda.Stream()
        with torch.cuda.stream(s2):
            net2, _ = _build("tiny", dev, seed=1)
            eng2 = net2.engine()
            eng2.use_graph = False
            a2 = inputs(11)
            while not stop:
                for _ in range(10):
                    eng2.forward(*a2, T)
                s2.synchronize()
                started.set()

    th = threading.Thread(target=load)
    th.start()
    try:
        assert started.wait(120), "the load thread never ran"
        a = inputs(5)
        ref = eng.forward(*a, T).clone()
        differing = 0
        for _ in range(60):
            differing += int(not torch.equal(eng.forward(*a, T), ref))
            time.sleep(0)
    finally:
        stop.append(1)
        th.join()
    assert differing == 0, f"{differing} of 60 forwards differ from the first while a second stream runs"


def test_frame_sliced_execution_is_bitwise_identical(dev, tiny):
    """engine._slice_rows: running the token-wise chains a few frames at a time changes no bit of the output."""
    net, _ = tiny
    eng = net.engine()
    g = torch.Generator().manual_seed(19)
    T, h, w = 5, 16, 8
    n = 2 * T
    x, t = torch.randn(n, 11, h, w, generator=g).to(dev), torch.randint(0, 1000, (n,), generator=g).to(dev)
    y, dense = torch.randn(n, 1, 1024, generator=g).to(dev), torch.randn(n, 6, h, w, generator=g).to(dev)
    keep = (eng.slice_frames, eng.slice_min_bytes, eng.use_graph, eng.slice_attn)
    try:
        eng.use_graph = False
        eng.slice_attn = True
        eng.slice_frames = 0
        ref = eng.forward(x, None, t, y, dense, T).clone()
        for frames in (1, 3):
            eng.slice_frames, eng.slice_min_bytes = frames, 0
            out = eng.forward(x, None, t, y, dense, T)
            assert torch.equal(out, ref), frames
    finally:
        eng.slice_frames, eng.slice_min_bytes, eng.use_graph, eng.slice_attn = keep


def test_graph_replay_equals_eager(dev, tiny):
    """Engine default = hipGraph replay; must be bit-identical to the eager launch sequence, also when
    inputs change between calls (static input buffers are refreshed)."""
    net, _ = tiny
    eng = net.engine()
    g = torch.Generator().manual_seed(9)
    T, h, w = 2, 8, 8
    n = 2 * T
    for _ in range(3):
        x, t = torch.randn(n, 11, h, w, generator=g).to(dev), torch.randint(0, 1000, (n,), generator=g).to(dev)
        y, dense = torch.randn(n, 1, 1024, generator=g).to(dev), torch.randn(n, 6, h, w, generator=g).to(dev)
        a = eng.forward_graphed(x, None, t, y, dense, T)
        b = eng.forward(x, None, t, y, dense, T)
        assert torch.equal(a, b)
    assert len(eng._graphs) >= 1


def _sampler_run(net, dev, T, hw, steps, eps):
    from seva import sampling as S
    from seva import synthetic as synth
    from seva.model import SGMWrapper
    sc = synth.synth_scene(T, (hw, hw), (0,), seed=23)
    disc = S.DDPMDiscretization()
    den = S.DiscreteDenoiser(disc, num_idx=1000, device=dev)
    sampler = S.EulerEDMSampler(disc, S.MultiviewCFG(1.2), num_steps=steps, verbose=False, device=dev,
                                s_churn=0.0)
    it = iter(eps)
    sampler.noise_fn = lambda x: next(it).to(x.device)
    wrap = SGMWrapper(net)
    cond = {k: v.to(dev) for k, v in sc["cond"].items()}
    uc = {k: v.to(dev) for k, v in sc["uc"].items()}
    return sampler(lambda x, s, c: den(wrap, x, s, c, num_frames=T), sc["noise"].to(dev), scale=2.0,
                   cond=cond, uc=uc, verbose=False, c2w=sc["c2w"].to(dev), K=sc["K"].to(dev),
                   input_frame_mask=sc["input_frame_mask"].to(dev))


def test_tiny_sampler_loop_vs_golden(dev, tiny):
    net, _ = tiny
    g = load_golden("g7_loop_tiny")
    y = _sampler_run(net, dev, int(g["T"]), int(g["hw"]), int(g["steps"]), list(g["eps"]))
    err = rel_l2(y.cpu(), g["y"])
    print(f"tiny 4-step sampler loop vs reference golden: rel-L2 {err:.3e}")
    assert err < NET_TOL  # four chained network calls, still inside the per-latent tolerance (measured 6.6e-4)


def test_guiders_and_denoiser_vs_golden(dev, tiny):
    from seva import sampling as S
    from seva import synthetic as synth
    from seva.model import SGMWrapper
    net, _ = tiny
    g = load_golden("g6_guiders")
    T = int(g["T"])
    d, c2w, K, mask = g["d"].to(dev), g["c2w"].to(dev), g["K"].to(dev), g["mask"].bool().to(dev)
    sig = torch.full((T,), 24.2054, device=dev) + 1e-6
    assert rel_l2(S.VanillaCFG()(d, sig, 2.0).cpu(), g["y0"]) < 1e-6
    assert rel_l2(S.MultiviewCFG(1.2)(d, sig, 2.0, c2w, K, mask).cpu(), g["y1"]) < 1e-6
    assert rel_l2(S.MultiviewTemporalCFG(T, 1.2)(d, sig, 2.0, c2w, K, mask).cpu(), g["y2"]) < 1e-6
    g = load_golden("g5_denoiser")
    T = int(g["T"])
    sc = synth.synth_scene(T, tuple(g["x"].shape[-2:]), (0,), seed=int(g["seed"]))
    guider = S.MultiviewCFG(1.2)
    cond = {k: v.to(dev) for k, v in sc["cond"].items()}
    uc = {k: v.to(dev) for k, v in sc["uc"].items()}
    xin, sin, cin = guider.prepare_inputs(g["x"].to(dev), g["sigma"].to(dev), cond, uc)
    den = S.DiscreteDenoiser(S.DDPMDiscretization(), num_idx=1000, device=dev)
    out = den(SGMWrapper(net), xin, sin, cin, num_frames=T)
    err = rel_l2(out.cpu(), g["y"])
    print(f"denoiser call vs reference golden: rel-L2 {err:.3e}")
    assert err < NET_TOL


def test_cpu_tensors_fail_loudly(tiny):
    from seva._native import SevaNativeError
    net, _ = tiny
    with pytest.raises(SevaNativeError):
        net(torch.zeros(2, 11, 8, 8), torch.zeros(2, dtype=torch.int64), torch.zeros(2, 1, 1024),
            torch.zeros(2, 6, 8, 8), num_frames=2)


@pytest.fixture(scope="module")
def full(dev):
    return _build("full", dev)


def test_full_forward_vs_golden(dev, full):
    """BASELINE config 1 shapes (T=4, 32x32 latent, CFG batch 8), 1.3 B parameters."""
    from seva.model import SGMWrapper
    net, _ = full
    g = load_golden("g4_full_forward")
    T = int(g["T"])
    c = {k: g[k].to(dev) for k in ("crossattn", "concat", "dense_vector")}
    y = SGMWrapper(net)(g["x"].to(dev), g["t"].to(dev), c, num_frames=T)
    err = rel_l2(y.cpu(), g["y"])
    per = [rel_l2(y[i].cpu(), g["y"][i]) for i in range(y.shape[0])]
    print(f"1.3B forward vs reference golden: rel-L2 {err:.3e}; per latent max {max(per):.3e}")
    assert err < NET_TOL and max(per) < NET_TOL


def test_full_sampler_loop_vs_golden(dev, full):
    net, _ = full
    g = load_golden("g7_loop_full")
    y = _sampler_run(net, dev, int(g["T"]), int(g["hw"]), int(g["steps"]), list(g["eps"]))
    err = rel_l2(y.cpu(), g["y"])
    per = [rel_l2(y[i].cpu(), g["y"][i]) for i in range(y.shape[0])]
    print(f"1.3B 4-step sampler loop (config 1) vs reference golden: rel-L2 {err:.3e}; per latent {[f'{e:.2e}' for e in per]}")
    # Tolerances, explicitly: the north_star's 1e-3 is per denoised latent of ONE network call (asserted per latent in
    # test_full_forward_vs_golden and, at 576x576, tests/test_headline_gpu.py).  This is FOUR chained calls over the coarse
    # 4-step schedule (sigma 84.9 -> 24.2 -> 9.35 -> 3.10 -> 0): the whole result must still be inside 1e-3 (measured
    # 6.0e-4); a single latent of the chain may reach 2x the per-call tolerance (measured max 1.64e-3).
    assert err < NET_TOL and max(per) < 2 * NET_TOL


# ------------------------------------------------------------------ VAE decoder (parity UNPINNED)
def _vae(dev, block_out, seed=3):
    from oracle import vae_ref as V
    from seva import synthetic as synth
    from seva.modules.autoencoder import AutoEncoder, VaeWeights
    ae = AutoEncoder(chunk_size=1, random_init=True)
    if tuple(block_out) != tuple(ae.module.block_out):
        ae.module = VaeWeights(block_out=block_out)
    sd = synth.synth_state_dict({**V.decoder_shapes(block_out=block_out), **V.encoder_shapes(block_out=block_out)}, seed)
    ae.module.load_state_dict(sd)
    return ae.to(dev), sd


@pytest.mark.parametrize("block_out,n,h,w", [((64, 64, 128, 128), 2, 6, 6), ((128, 256, 512, 512), 1, 16, 16)])
def test_vae_decode_vs_restatement(dev, block_out, n, h, w):
    """HIP decoder vs oracle/vae_ref.py (our restatement of the published SD-2.1 VAE topology).
    diffusers itself is unavailable offline: this is self-consistency, not pinned parity."""
    from oracle import vae_ref as V
    ae, sd = _vae(dev, block_out)
    z = torch.randn(n, 4, h, w, generator=torch.Generator().manual_seed(1)) * 0.18215 * 4
    out = ae.decode(z.to(dev))
    ref = V.vae_decode(sd, z)
    assert out.shape == (n, 3, 8 * h, 8 * w)
    err = rel_l2(out.cpu(), ref)
    print(f"vae decode {block_out} {n}x{h}x{w}: rel-L2 {err:.3e}")
    assert err < 2e-3


def test_vae_decode_576_frame(dev):
    """One 576x576 frame (72x72 latent) through the full-width decoder."""
    from oracle import vae_ref as V
    ae, sd = _vae(dev, (128, 256, 512, 512))
    z = torch.randn(1, 4, 72, 72, generator=torch.Generator().manual_seed(2)) * 0.18215 * 4
    out = ae.decode(z.to(dev), 1)
    torch.set_num_threads(min(16, torch.get_num_threads()))
    ref = V.vae_decode(sd, z)
    err = rel_l2(out.cpu(), ref)
    print(f"vae decode 576x576 frame: rel-L2 {err:.3e}")
    assert out.shape == (1, 3, 576, 576) and err < 2e-3


def test_softmax_rows(dev):
    from seva import ops
    x = torch.randn(300, 200, device=dev) * 5
    out = torch.full((300, 256), float("nan"), device=dev, dtype=torch.float16)
    ops.softmax_rows(x, out, 200, 0.3)
    ref = torch.softmax(x * 0.3, -1)
    assert (out[:, :200].float() - ref).abs().max() < 1e-3 and out[:, 200:].abs().max() == 0


# ------------------------------------------------------------------ full-size (BASELINE headline shape) properties
def _headline_inputs(dev, T=21, hw=72, seed=5):
    from seva import synthetic as synth
    sc = synth.synth_scene(T, (hw, hw), (0,), seed=seed)
    g = torch.Generator().manual_seed(seed + 1)
    x = torch.randn(2 * T, 4, hw, hw, generator=g) * 3.0
    concat = torch.cat((sc["uc"]["concat"], sc["cond"]["concat"]), 0)
    y = torch.cat((sc["uc"]["crossattn"], sc["cond"]["crossattn"]), 0)
    dense = torch.cat((sc["uc"]["dense_vector"], sc["cond"]["dense_vector"]), 0)
    t = torch.full((2 * T,), 500, dtype=torch.int64)
    return [v.to(dev) for v in (x, concat, t, y, dense)]


def test_headline_shape_properties(dev, full):
    """T=21, 576x576 (latent 72x72), CFG batch 42 -- the shape the headline metric is quoted on.
    No CPU oracle at this size (76.9 TFLOP per call); size-independent properties instead:
      * determinism: two eager calls are bit-identical, and equal to the hipGraph replay;
      * batch-split invariance: the uncond and cond halves are independent scenes, every kernel is
        row/sample-independent, so running a half alone must reproduce its rows BIT-EXACTLY;
      * outputs finite, O(1) magnitude."""
    net, _ = full
    eng = net.engine()
    T = 21
    x, concat, t, y, dense = _headline_inputs(dev, T)
    a = eng.forward(x, concat, t, y, dense, T).clone()
    b = eng.forward(x, concat, t, y, dense, T).clone()
    assert torch.equal(a, b)
    c = eng.forward_graphed(x, concat, t, y, dense, T)
    assert torch.equal(a, c)
    assert torch.isfinite(a).all() and 1e-3 < float(a.abs().mean()) < 1e2
    lo = eng.forward(x[:T], concat[:T], t[:T], y[:T], dense[:T], T).clone()
    hi = eng.forward(x[T:], concat[T:], t[T:], y[T:], dense[T:], T).clone()
    assert torch.equal(lo, a[:T]) and torch.equal(hi, a[T:])


def test_headline_sampler_step_finite_and_replayable(dev, full):
    """One Euler-EDM step at the headline shape through the public sampling API."""
    net, _ = full
    T, hw = 21, 72
    from seva import sampling as S
    from seva import synthetic as synth
    from seva.model import SGMWrapper
    sc = synth.synth_scene(T, (hw, hw), (0,), seed=23)
    disc = S.DDPMDiscretization()
    den = S.DiscreteDenoiser(disc, num_idx=1000, device=dev)
    sampler = S.EulerEDMSampler(disc, S.MultiviewCFG(1.2), num_steps=50, verbose=False, device=dev)
    eps = torch.randn(T, 4, hw, hw, generator=torch.Generator().manual_seed(1)).to(dev)
    sampler.noise_fn = lambda x: eps
    wrap = SGMWrapper(net)
    cond = {k: v.to(dev) for k, v in sc["cond"].items()}
    uc = {k: v.to(dev) for k, v in sc["uc"].items()}
    gk = dict(c2w=sc["c2w"].to(dev), K=sc["K"].to(dev), input_frame_mask=sc["input_frame_mask"].to(dev))
    x, s_in, sigmas, n_sig, cond, uc = sampler.prepare_sampling_loop(sc["noise"].to(dev), cond, uc, None)
    assert n_sig == 51 and abs(float(sigmas[0]) - 84.916) < 1e-2
    f = lambda xx, ss, cc: den(wrap, xx, ss, cc, num_frames=T)
    x1 = sampler.sampler_step(s_in * sigmas[0], s_in * sigmas[1], f, x, 2.0, cond, uc, 0.0, **gk)
    x2 = sampler.sampler_step(s_in * sigmas[0], s_in * sigmas[1], f, x, 2.0, cond, uc, 0.0, **gk)
    assert torch.equal(x1, x2) and torch.isfinite(x1).all()
    # the input frame is pinned by the replace-blend: its denoised value is exact, so its Euler update
    # moves it along (x - latent)/sigma only -- check against the closed form
    lat = cond["replace"][0:1, :4]
    expect = x[0:1] + (sigmas[1] - (sigmas[0] + 1e-6)) * (x[0:1] - lat) / (sigmas[0] + 1e-6)
    assert (x1[0:1] - expect).abs().max() < 0.35 * float(x1[0:1].abs().max())


# ------------------------------------------------------------------ VAE encoder (parity UNPINNED)
@pytest.mark.parametrize("block_out,n,h,w", [((64, 64, 128, 128), 2, 48, 64), ((128, 256, 512, 512), 1, 128, 128)])
def test_vae_encode_vs_restatement(dev, block_out, n, h, w):
    """HIP encoder vs oracle/vae_ref.py (restatement of the published SD-2.1 VAE topology; self-consistency)."""
    from oracle import vae_ref as V
    ae, sd = _vae(dev, block_out)
    x = torch.rand(n, 3, h, w, generator=torch.Generator().manual_seed(4)) * 2 - 1
    out = ae.encode(x.to(dev))
    torch.set_num_threads(min(16, torch.get_num_threads()))
    ref = V.vae_encode(sd, x)
    assert out.shape == (n, 4, h // 8, w // 8)
    err = rel_l2(out.cpu(), ref)
    print(f"vae encode {block_out} {n}x{h}x{w}: rel-L2 {err:.3e}")
    assert err < 2e-3


def test_vae_encode_576_frame_and_chunking(dev):
    """576x576 frames through the full-width encoder; chunked encode == per-frame encode (bitwise)."""
    ae, _ = _vae(dev, (128, 256, 512, 512))
    x = (torch.rand(2, 3, 576, 576, generator=torch.Generator().manual_seed(6)) * 2 - 1).to(dev)
    z = ae.encode(x, 1)
    assert z.shape == (2, 4, 72, 72) and torch.isfinite(z).all()
    z0 = ae.encode(x[:1], 1)
    assert torch.equal(z[:1], z0)
    img = ae.decode(z, 1)
    assert img.shape == (2, 3, 576, 576) and torch.isfinite(img).all()


# ------------------------------------------------------------------ whole-step hipGraph (seva/_stepgraph.py)
def _loop(net, dev, T, hw, steps, eps, guider=None, inference=False, scale=2.0):
    from seva import sampling as S
    from seva import synthetic as synth
    from seva.model import SGMWrapper
    sc = synth.synth_scene(T, (hw, hw), (0,), seed=23)
    disc = S.DDPMDiscretization()
    den = S.DiscreteDenoiser(disc, num_idx=1000, device=dev)
    sampler = S.EulerEDMSampler(disc, guider or S.MultiviewCFG(1.2), num_steps=steps, verbose=False, device=dev,
                                s_churn=0.0)
    it = iter(eps)
    sampler.noise_fn = lambda x: next(it).to(x.device)
    wrap = SGMWrapper(net)

    def run():
        cond = {k: v.to(dev) for k, v in sc["cond"].items()}
        uc = {k: v.to(dev) for k, v in sc["uc"].items()}
        # VanillaCFG takes no camera arguments (reference eval.py passes them to the multiview guiders only)
        kw = {} if isinstance(sampler.guider, S.VanillaCFG) and not isinstance(sampler.guider, S.MultiviewCFG) else dict(
            c2w=sc["c2w"].to(dev), K=sc["K"].to(dev), input_frame_mask=sc["input_frame_mask"].to(dev))
        return sampler(lambda x, s, c: den(wrap, x, s, c, num_frames=T), sc["noise"].to(dev), scale=scale,
                       cond=cond, uc=uc, verbose=False, **kw)

    if inference:
        with torch.inference_mode():
            return run().clone(), sampler
    return run(), sampler


@pytest.mark.parametrize("guider_kind", [0, 1, 2])
def test_whole_step_graph_is_bit_identical_to_eager(dev, tiny, guider_kind, monkeypatch):
    """6-step loop: step 0 eager (warm-up), step 1 captures the WHOLE sampler step into one hipGraph, steps 2-5 replay it;
    the result must equal the all-eager loop bit for bit (same recorded eps on both sides)."""
    from seva import sampling as S
    net, _ = tiny
    T, hw, steps = 4, 16, 6
    g = torch.Generator().manual_seed(5)
    eps = [torch.randn(T, 4, hw, hw, generator=g) for _ in range(steps)]
    mk = lambda: [S.VanillaCFG(), S.MultiviewCFG(1.2), S.MultiviewTemporalCFG(T, 1.2)][guider_kind]  # noqa: E731
    monkeypatch.setenv("SEVA_STEPGRAPH", "0")
    monkeypatch.setenv("SEVA_HIPGRAPH", "0")
    ref, s0 = _loop(net, dev, T, hw, steps, eps, mk())
    assert s0._step_graphs.captures == 0
    monkeypatch.setenv("SEVA_STEPGRAPH", "1")
    monkeypatch.setenv("SEVA_HIPGRAPH", "1")
    got, s1 = _loop(net, dev, T, hw, steps, eps, mk())
    assert s1._step_graphs.captures == 1 and s1._step_graphs.graph.replays == steps - 1
    assert torch.equal(got, ref)
    # under torch.inference_mode() (how the reference's do_sample calls the sampler, eval.py:1242)
    got_i, s2 = _loop(net, dev, T, hw, steps, eps, mk(), inference=True)
    assert s2._step_graphs.captures == 1
    assert torch.equal(got_i, ref)


def test_to_d_and_euler_step_kernels(dev):
    """`seva_to_d_f32` / `seva_euler_step_f32` (the guider-without-frame_scale branch, reference sampling.py:24-25,366-368)."""
    from seva import ops
    from seva import sampling as S
    g = torch.Generator().manual_seed(3)
    n, shape = 5, (5, 4, 9, 7)
    x, den = torch.randn(shape, generator=g).to(dev), torch.randn(shape, generator=g).to(dev)
    sigma = (torch.rand(n, generator=g) * 10 + 0.1).to(dev)
    dt = (torch.randn(n, generator=g)).to(dev)
    d = S.to_d(x, sigma, den)
    ref_d = (x - den) / sigma[:, None, None, None]
    assert torch.allclose(d, ref_d, rtol=2e-6, atol=1e-6)
    out = torch.empty_like(x)
    ops.euler_step(x, den, sigma, dt, out)
    ref = x + dt[:, None, None, None] * ref_d
    assert torch.allclose(out, ref, rtol=2e-6, atol=2e-6)

    class PlainGuider:  # no frame_scale attribute: sampler_step takes the guider() + euler_step branch
        def prepare_inputs(self, x, s, c, uc):
            return S.VanillaCFG().prepare_inputs(x, s, c, uc)

        def __call__(self, x, sigma, scale):
            u, c = x.chunk(2)
            return u + scale * (c - u)

    sampler = S.EulerEDMSampler(S.DDPMDiscretization(), PlainGuider(), num_steps=4, device=dev)
    eps = torch.randn(shape, generator=g).to(dev)
    sampler.noise_fn = lambda t: eps
    fake_den = lambda xx, ss, cc: xx * 0.5  # noqa: E731
    s_in = torch.ones(n, device=dev)
    o = sampler.sampler_step(s_in * 3.0, s_in * 2.0, fake_den, x, 2.0, {}, {}, 0.0)
    sg = torch.full((n, 1, 1, 1), 3.0, device=dev)
    sh = sg * 1.0 + 1e-6  # fp32, as the sampler computes it
    xn = x + eps * (sh ** 2 - sg ** 2) ** 0.5
    dd = (xn - 0.5 * xn) / sh
    assert torch.allclose(o, xn + (2.0 - sh) * dd, rtol=1e-5, atol=1e-5)


def test_tracked_sampler_subclass_protocol(dev, tiny):
    """The way the reference's `GradioTrackedSampler` (seva/eval.py:1037-1089) drives the sampler: a subclass that calls
    `prepare_sampling_loop`, `get_sigma_gen` and `sampler_step` itself, compares `sigmas[i]` with Python floats, polls an abort flag per
    step and returns None when aborted.  Must give the same result as `__call__` (whole-step hipGraph included) and abort cleanly."""
    import threading
    from seva import sampling as S
    from seva import synthetic as synth
    from seva.model import SGMWrapper
    net, _ = tiny
    T, hw, steps = 4, 16, 5

    class Tracked(S.EulerEDMSampler):
        def __init__(self, *a, abort_event=None, **k):
            super().__init__(*a, **k)
            self.abort_event = abort_event
            self.steps_done = 0

        def __call__(self, denoiser, x, scale, cond, uc=None, num_steps=None, verbose=True, **guider_kwargs):
            uc = cond if uc is None else uc
            x, s_in, sigmas, num_sigmas, cond, uc = self.prepare_sampling_loop(x, cond, uc, num_steps)
            for i in self.get_sigma_gen(num_sigmas, verbose=verbose):
                gamma = min(self.s_churn / (num_sigmas - 1), 2 ** 0.5 - 1) if self.s_tmin <= sigmas[i] <= self.s_tmax else 0.0
                x = self.sampler_step(s_in * sigmas[i], s_in * sigmas[i + 1], denoiser, x, scale, cond, uc, gamma, **guider_kwargs)
                self.steps_done += 1
                if self.abort_event is not None and self.abort_event.is_set():
                    return None
            return x

    g = torch.Generator().manual_seed(13)
    eps = [torch.randn(T, 4, hw, hw, generator=g) for _ in range(steps)]
    sc = synth.synth_scene(T, (hw, hw), (0,), seed=23)
    disc = S.DDPMDiscretization()
    den = S.DiscreteDenoiser(disc, num_idx=1000, device=dev)
    wrap = SGMWrapper(net)

    def run(sampler):
        it = iter(eps)
        sampler.noise_fn = lambda x: next(it).to(x.device)
        cond = {k: v.to(dev) for k, v in sc["cond"].items()}
        uc = {k: v.to(dev) for k, v in sc["uc"].items()}
        with torch.inference_mode():
            out = sampler(lambda x, s, c: den(wrap, x, s, c, num_frames=T), sc["noise"].to(dev), scale=2.0, cond=cond, uc=uc,
                          verbose=False, c2w=sc["c2w"].to(dev), K=sc["K"].to(dev), input_frame_mask=sc["input_frame_mask"].to(dev))
            return None if out is None else out.clone()

    mk = lambda cls, **k: cls(disc, S.MultiviewCFG(1.2), num_steps=steps, verbose=False, device=dev, s_churn=0.0, **k)  # noqa: E731
    ref = run(mk(S.EulerEDMSampler))
    tracked = mk(Tracked)
    got = run(tracked)
    assert tracked.steps_done == steps and torch.equal(got, ref)
    ev = threading.Event()
    aborting = mk(Tracked, abort_event=ev)
    orig = aborting.sampler_step

    def step_then_abort(*a, **k):
        out = orig(*a, **k)
        if aborting.steps_done == 1:
            ev.set()
        return out

    aborting.sampler_step = step_then_abort
    assert run(aborting) is None and aborting.steps_done == 2
