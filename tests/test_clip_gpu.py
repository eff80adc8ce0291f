"""CLIP ViT-H-14 image conditioner on the HIP path vs oracle/clip_ref.py.

PARITY UNPINNED: open_clip and kornia are not available offline and the reference holds no fixture for them; the oracle
restates their published algorithms (see its header), so these are self-consistency checks -- like the VAE's."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

from conftest import rel_l2


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from seva import _native
    _native.load()
    return torch.device("cuda:0")


def _conditioner(dev, params, seed=11):
    from oracle import clip_ref as CR
    from seva import synthetic as synth
    from seva.modules.conditioner import CLIPConditioner
    cond = CLIPConditioner(params, random_init=True)
    shapes = CR.vit_shapes(params.width, params.layers, params.patch_size, params.image_size,
                           int(params.width * params.mlp_ratio), params.embed_dim)
    assert {k: tuple(v.shape) for k, v in cond.module.state_dict().items()} == shapes
    sd = synth.synth_state_dict(shapes, seed)
    g = torch.Generator().manual_seed(seed)
    for k in ("visual.class_embedding", "visual.positional_embedding"):
        sd[k] = 0.02 * torch.randn(shapes[k], generator=g)
    sd["visual.proj"] = torch.randn(shapes["visual.proj"], generator=g) * params.width ** -0.5
    for k in shapes:  # not a `.weight` name: give the fused q/k/v projection a proper fan-in scale
        if k.endswith("in_proj_weight"):
            sd[k] = torch.randn(shapes[k], generator=g) * params.width ** -0.5
    cond.module.load_state_dict(sd, strict=True)
    return cond.to(dev), sd


@pytest.mark.parametrize("B,H,L,D", [(2, 16, 257, 80), (3, 2, 50, 64), (1, 3, 300, 128), (5, 1, 7, 2)])
def test_attention_small(dev, B, H, L, D):
    from seva import ops
    C = H * D
    g = torch.Generator().manual_seed(3)
    qkv = torch.randn(B, L, 3 * C, generator=g).half().to(dev)
    out = torch.full((B, L, C), float("nan"), device=dev, dtype=torch.float16)
    ops.attention_small(qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:], out, batch=B, heads=H, L=L, head_dim=D,
                        q_strides=(L * 3 * C, 3 * C), k_strides=(L * 3 * C, 3 * C), o_strides=(L * C, C),
                        scale=1.0 / math.sqrt(D))
    x = qkv.view(B, L, 3, H, D).permute(2, 0, 3, 1, 4).double()
    ref = (torch.softmax(x[0] @ x[1].transpose(-1, -2) / math.sqrt(D), -1) @ x[2]).transpose(1, 2).reshape(B, L, C)
    assert torch.isfinite(out).all() and rel_l2(out, ref) < 1e-3


@pytest.mark.parametrize("n,H,W", [(2, 96, 128), (1, 576, 576), (2, 300, 224), (1, 1000, 640)])
def test_clip_preprocess_vs_restatement(dev, n, H, W):
    """up-scaling (no blur), the 576x576 case of the hot path (3x3 Gaussian), mixed, and a 7-tap blur"""
    from oracle import clip_ref as CR
    from seva.modules.conditioner import ViTParams
    cond, _ = _conditioner(dev, ViTParams(width=320, layers=1, embed_dim=64))
    x = (torch.rand(n, 3, H, W, generator=torch.Generator().manual_seed(5)) * 2 - 1)
    got = cond.preprocess(x.to(dev)).cpu()
    ref = CR.preprocess(x)
    assert got.shape == ref.shape == (n, 3, 224, 224)
    assert (got - ref).abs().max() < 4e-3 and rel_l2(got, ref) < 1e-3  # f16 storage of the patch matrix


def test_clip_tiny_tower_vs_restatement(dev):
    from oracle import clip_ref as CR
    from seva.modules.conditioner import ViTParams
    p = ViTParams(width=320, layers=3, embed_dim=128)  # 4 heads of width 80, 257 tokens
    cond, sd = _conditioner(dev, p)
    x = (torch.rand(3, 3, 160, 200, generator=torch.Generator().manual_seed(7)) * 2 - 1)
    got = cond(x.to(dev)).cpu()
    ref = CR.clip_conditioner(sd, x, heads=p.heads)
    err = rel_l2(got, ref)
    print(f"\ntiny CLIP tower (3 layers, width 320): rel-L2 {err:.3e}")
    assert got.shape == (3, 128) and err < 2e-3
    # one frame alone reproduces its row bit for bit (no cross-frame coupling anywhere)
    assert torch.equal(cond(x[1:2].to(dev)).cpu(), got[1:2])


def test_clip_vit_h14_full_width(dev):
    """The real ViT-H-14 geometry (32 layers, width 1280, 16 heads x 80, MLP 5120, 1024-d output), one 576x576 frame."""
    import time
    from oracle import clip_ref as CR
    from seva.modules.conditioner import ViTParams
    p = ViTParams()
    cond, sd = _conditioner(dev, p)
    assert sum(v.numel() for v in sd.values()) == 632_076_800
    x = (torch.rand(1, 3, 576, 576, generator=torch.Generator().manual_seed(9)) * 2 - 1)
    got = cond(x.to(dev))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    got = cond(x.to(dev)).cpu()
    dt = time.perf_counter() - t0
    torch.set_num_threads(min(16, torch.get_num_threads()))
    ref = CR.clip_conditioner(sd, x, heads=p.heads)
    err = rel_l2(got, ref)
    print(f"\nCLIP ViT-H-14 (632 M params), one 576x576 frame: rel-L2 {err:.3e}; {dt * 1e3:.1f} ms per frame on the GPU")
    assert got.shape == (1, 1024) and err < 2e-3
