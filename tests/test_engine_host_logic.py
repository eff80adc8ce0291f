"""CPU checks of the host side of the product: module tree == reference state_dict contract,
weight packing layouts, engine orchestration (with tests/fake_ops.py standing in for the HIP
kernels), sampler host logic, C-ABI symbol export.  No GPU work happens here."""
import ctypes
import os
import re

import pytest
import torch
import torch.nn.functional as F

import fake_ops
from conftest import ROOT, load_golden, rel_l2


def _shapes(tag):
    g = load_golden(f"g0_keys_{tag}")
    return {str(k): tuple(int(s) for s in str(v).split(",")) for k, v in zip(g["keys"], g["shapes"])}


@pytest.mark.parametrize("tag", ["full", "tiny"])
def test_state_dict_contract(tag):
    """Same 1146 keys, shapes and order as the reference Seva (meta device, bf16, assign=True)."""
    from seva.model import Seva, SevaParams
    params = SevaParams() if tag == "full" else SevaParams(model_channels=64)
    with torch.device("meta"):
        net = Seva(params).to(torch.bfloat16)
    sd = net.state_dict()
    ref = _shapes(tag)
    assert list(sd.keys()) == list(ref.keys())
    assert {k: tuple(v.shape) for k, v in sd.items()} == ref
    assert all(v.dtype == torch.bfloat16 for v in sd.values())
    if tag == "full":
        assert len(sd) == 1146 and sum(v.numel() for v in sd.values()) == 1263968004
    fake = {k: torch.empty(s, dtype=torch.bfloat16, device="meta") for k, s in ref.items()}
    missing, unexpected = net.load_state_dict(fake, strict=False, assign=True)
    assert not missing and not unexpected


def test_layout_names_match_reference_regimes():
    from seva._arch import build_layout
    from seva.model import SevaParams
    lay = build_layout(SevaParams())
    mv = [s for s in lay.all_specs() if s.kind == "mvt"]
    assert len(mv) == 16 and sum(1 for s in lay.all_specs() if s.kind == "res") == 22
    joint = sorted(s.prefix for s in mv if s.joint)
    assert joint == sorted(["middle_block.1"] + [f"output_blocks.{i}.1" for i in (3, 4, 5, 6, 7, 8)])
    assert [s.heads for s in mv if s.prefix.startswith("input_blocks")] == [5, 5, 10, 10, 20, 20]


def test_weight_packing_layouts():
    from seva._engine import interleave_geglu, pack_conv3x3
    g = torch.Generator().manual_seed(0)
    w, b, a = torch.randn(512, 64, generator=g), torch.randn(512, generator=g), torch.randn(10, 64, generator=g)
    wi, bi = interleave_geglu(w, b)
    out = torch.empty(10, 256)
    fake_ops.gemm(a.half(), wi.half(), bias=bi, out_f32=out, geglu=True)
    y = a.half().float() @ w.half().float().T + b
    assert torch.allclose(out, y[:, :256] * F.gelu(y[:, 256:]), atol=1e-5)
    wc, x = torch.randn(32, 64, 3, 3, generator=g), torch.randn(2, 64, 5, 6, generator=g)
    o = torch.empty(2 * 5 * 6, 32)
    fake_ops.conv3x3(x.permute(0, 2, 3, 1).contiguous().half(), pack_conv3x3(wc), out_f32=o)
    ref = F.conv2d(x.half().float(), wc.half().float(), None, padding=1)
    assert torch.allclose(o.view(2, 5, 6, 32).permute(0, 3, 1, 2), ref, atol=1e-4)
    assert pack_conv3x3(torch.randn(8, 11, 3, 3), 64).shape == (8, 576)


def _cpu_engine(tag="tiny"):
    """Engine on CPU with the HIP ops replaced by fake_ops (test-only construction)."""
    from seva import _engine, synthetic as synth
    from seva.model import Seva, SevaParams
    params = SevaParams() if tag == "full" else SevaParams(model_channels=64)
    sd = synth.synth_state_dict(_shapes(tag))
    with torch.device("meta"):
        net = Seva(params)
    net.load_state_dict(sd, strict=True, assign=True)
    orig = _engine.SevaEngine.__dict__["_resolve_device"]
    _engine.SevaEngine._resolve_device = staticmethod(lambda m: torch.device("cpu"))  # test seam
    try:
        eng = _engine.SevaEngine(net)
    finally:
        _engine.SevaEngine._resolve_device = orig
    return eng, sd


@pytest.fixture()
def patched(monkeypatch):
    from seva import _engine
    monkeypatch.setattr(_engine, "ops", fake_ops)
    monkeypatch.setattr(_engine, "require_cuda", lambda *a: None)


def test_engine_orchestration_vs_oracle(patched):
    from oracle import seva_ref as O
    eng, sd = _cpu_engine()
    g = load_golden("g3_tiny_forward")
    T = int(g["T"])
    y = eng.forward(g["x"], g["concat"], g["t"], g["crossattn"], g["dense_vector"], T)
    err = rel_l2(y, g["y"])
    trace = {}
    O.sgm_wrapper_forward(sd, g["x"], g["t"], {k: g[k] for k in ("crossattn", "concat", "dense_vector")}, T, trace=trace)
    worst = 0.0
    for key, t in eng.arena.bufs.items():
        if key[0].startswith("out:") and key[0][4:] in trace:
            ref = trace[key[0][4:]]
            got = t.view(ref.shape[0], ref.shape[2], ref.shape[3], ref.shape[1]).permute(0, 3, 1, 2)
            worst = max(worst, rel_l2(got, ref))
    print(f"emulated-kernel engine vs golden {err:.2e}; worst layer {worst:.2e}")
    assert err < 2e-3 and worst < 3e-3


@pytest.mark.parametrize("T,h,w,lc", [(3, 8, 24, 1), (2, 8, 8, 3)])
def test_engine_orchestration_odd_shapes(patched, T, h, w, lc):
    from oracle import seva_ref as O
    eng, sd = _cpu_engine()
    g = torch.Generator().manual_seed(T * 100 + h)
    n = 2 * T
    x, t = torch.randn(n, 11, h, w, generator=g), torch.randint(0, 1000, (n,), generator=g)
    y, dense = torch.randn(n, lc, 1024, generator=g), torch.randn(n, 6, h, w, generator=g)
    out = eng.forward(x, None, t, y, dense, T)
    assert rel_l2(out, O.seva_forward(sd, x, t, y, dense, T)) < 2e-3


@pytest.mark.parametrize("frames", [1, 2])
def test_engine_frame_sliced_chains_match_unsliced(patched, frames):
    """Frame-sliced execution of the LN -> GEMM -> ... chains (engine._slice_rows) is the same computation.  (Bitwise on
    the GPU, tests/test_model_gpu.py; the torch-CPU emulation picks GEMM / LayerNorm kernels by row count and the differences pass through f16 roundings, so this
    only guards against slicing the wrong rows, which would be an O(1) error.)"""
    eng, _ = _cpu_engine()
    g = torch.Generator().manual_seed(17)
    T, h, w = 3, 8, 16
    n = 2 * T
    x, t = torch.randn(n, 11, h, w, generator=g), torch.randint(0, 1000, (n,), generator=g)
    y, dense = torch.randn(n, 1, 1024, generator=g), torch.randn(n, 6, h, w, generator=g)
    eng.slice_frames = 0
    ref = eng.forward(x, None, t, y, dense, T).clone()
    eng.slice_frames, eng.slice_min_bytes, eng.slice_attn = frames, 0, True  # force slicing at every level
    out = eng.forward(x, None, t, y, dense, T)
    err = rel_l2(out, ref)
    print(f"sliced ({frames} frame) vs unsliced, emulated kernels: {err:.2e}")
    assert err < 2e-3


def test_sampler_host_logic_vs_golden(patched, monkeypatch):
    """Product sampler classes driven by the emulated kernels reproduce the reference loop."""
    from seva import sampling as S
    from seva import synthetic as synth
    monkeypatch.setattr(S, "ops", fake_ops)
    monkeypatch.setattr(S, "_need_gpu", lambda *a: None)
    eng, _ = _cpu_engine()
    g = load_golden("g7_loop_tiny")
    T, hw, steps = int(g["T"]), int(g["hw"]), int(g["steps"])
    sc = synth.synth_scene(T, (hw, hw), (0,), seed=int(g["scene_seed"]))
    disc = S.DDPMDiscretization()
    den = S.DiscreteDenoiser(disc, num_idx=1000, device="cpu")
    sampler = S.EulerEDMSampler(disc, S.MultiviewCFG(1.2), num_steps=steps, verbose=False, device="cpu")
    it = iter(list(g["eps"]))
    sampler.noise_fn = lambda x: next(it)
    net = lambda x, t, c, num_frames: eng.forward(x, c["concat"], t, c["crossattn"], c["dense_vector"], num_frames)
    out = sampler(lambda x, s, c: den(net, x, s, c, num_frames=T), sc["noise"].clone(), scale=2.0,
                  cond=sc["cond"], uc=sc["uc"], verbose=False, c2w=sc["c2w"], K=sc["K"],
                  input_frame_mask=sc["input_frame_mask"])
    assert rel_l2(out, g["y"]) < 4e-3
    # guiders / schedules against the reference goldens (pure host logic)
    gg = load_golden("g6_guiders")
    sig = torch.full((T,), 24.2054) + 1e-6
    args = (gg["c2w"], gg["K"], gg["mask"].bool())
    assert rel_l2(S.VanillaCFG()(gg["d"], sig, 2.0), gg["y0"]) < 1e-6
    assert rel_l2(S.MultiviewCFG(1.2)(gg["d"], sig, 2.0, *args), gg["y1"]) < 1e-6
    assert rel_l2(S.MultiviewTemporalCFG(T, 1.2)(gg["d"], sig, 2.0, *args), gg["y2"]) < 1e-6
    g1 = load_golden("g1_schedules")
    assert torch.equal(disc(50), g1["sig50"]) and torch.equal(disc(4), g1["sig4"])
    assert torch.equal(den.sigmas, g1["table"]) and torch.equal(den.sigma_to_idx(g1["sig50"][:-1]), g1["idx50"])
    with pytest.raises(ValueError):
        disc.get_sigmas(1001)
    with pytest.raises(ValueError):
        S.append_dims(torch.zeros(2, 2), 1)


def test_capi_exports_every_declared_symbol():
    from seva import _native
    hdr = open(os.path.join(ROOT, "include", "seva_hip.h")).read()
    declared = set(re.findall(r"\b(seva_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(_native.SYMBOLS), declared ^ set(_native.SYMBOLS)
    lib = ctypes.CDLL(_native.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    assert _native.load().seva_abi_version() == _native.ABI_VERSION


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "stable-virtual-camera_amd", "seva")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+\S*oracle", src, re.M), f
                assert "/root/reference" not in src, f


def test_hip_path_fails_loudly_on_cpu():
    from seva import ops, sampling as S
    from seva._native import SevaNativeError
    from seva.model import Seva, SevaParams
    net = Seva(SevaParams(model_channels=64))
    with pytest.raises(SevaNativeError):
        net(torch.zeros(2, 11, 8, 8), torch.zeros(2, dtype=torch.int64), torch.zeros(2, 1, 1024),
            torch.zeros(2, 6, 8, 8), num_frames=2)
    with pytest.raises(SevaNativeError):
        ops.gemm(torch.zeros(4, 64, dtype=torch.float16), torch.zeros(4, 64, dtype=torch.float16),
                 out_f32=torch.zeros(4, 4))
    with pytest.raises(SevaNativeError):
        S.VanillaCFG()(torch.zeros(4, 4, 2, 2), torch.ones(2), 2.0)


def test_vae_decoder_orchestration_vs_restatement(monkeypatch):
    """VAE decode engine (emulated kernels) vs oracle/vae_ref.py.  Parity with diffusers is UNPINNED."""
    from oracle import vae_ref as V
    from seva import _vae_engine, synthetic as synth
    from seva.modules.autoencoder import VaeDecoderWeights
    monkeypatch.setattr(_vae_engine, "ops", fake_ops)
    monkeypatch.setattr(_vae_engine, "require_cuda", lambda *a: None)
    monkeypatch.setattr(_vae_engine.VaeDecoderEngine, "_resolve_device", staticmethod(lambda w: torch.device("cpu")))
    full = {k: tuple(v.shape) for k, v in VaeDecoderWeights().state_dict().items()}
    assert full == V.decoder_shapes()
    small = (64, 64, 128, 128)  # same topology, narrower: keeps the CPU test fast
    wts = VaeDecoderWeights(block_out=small)
    shapes = V.decoder_shapes(block_out=small)
    assert {k: tuple(v.shape) for k, v in wts.state_dict().items()} == shapes
    sd = synth.synth_state_dict(shapes, 3)
    wts.load_state_dict(sd)
    eng = _vae_engine.VaeDecoderEngine(wts)
    z = torch.randn(2, 4, 6, 6, generator=torch.Generator().manual_seed(0)) * 0.18215 * 4
    out = eng.decode(z, 0.18215)
    ref = V.vae_decode(sd, z)
    assert out.shape == (2, 3, 48, 48)
    err = rel_l2(out, ref)
    print(f"vae decode (emulated kernels) vs restatement: {err:.2e}")
    assert err < 3e-3


def test_vae_encoder_orchestration_vs_restatement(monkeypatch):
    """VAE encode engine (emulated kernels) vs oracle/vae_ref.py.  Parity with diffusers is UNPINNED."""
    from oracle import vae_ref as V
    from seva import _vae_engine, synthetic as synth
    from seva.modules.autoencoder import VaeWeights
    monkeypatch.setattr(_vae_engine, "ops", fake_ops)
    monkeypatch.setattr(_vae_engine, "require_cuda", lambda *a: None)
    monkeypatch.setattr(_vae_engine.VaeEncoderEngine, "_resolve_device", staticmethod(lambda w: torch.device("cpu")))
    full = {k: tuple(v.shape) for k, v in VaeWeights().state_dict().items()}
    assert full == {**V.decoder_shapes(), **V.encoder_shapes()}
    small = (64, 64, 128, 128)
    wts = VaeWeights(block_out=small)
    shapes = {**V.decoder_shapes(block_out=small), **V.encoder_shapes(block_out=small)}
    assert {k: tuple(v.shape) for k, v in wts.state_dict().items()} == shapes
    sd = synth.synth_state_dict(shapes, 5)
    wts.load_state_dict(sd)
    eng = _vae_engine.VaeEncoderEngine(wts)
    x = torch.rand(2, 3, 48, 64, generator=torch.Generator().manual_seed(0)) * 2 - 1
    out = eng.encode(x, 0.18215)
    ref = V.vae_encode(sd, x)
    assert out.shape == (2, 4, 6, 8)
    err = rel_l2(out, ref)
    print(f"vae encode (emulated kernels) vs restatement: {err:.2e}")
    assert err < 3e-3
    with pytest.raises(ValueError):
        eng.encode(torch.zeros(1, 3, 20, 16), 0.18215)
