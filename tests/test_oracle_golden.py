"""Pins the CPU oracle (oracle/*.py) against golden vectors produced by the reference itself
(oracle/make_goldens.py).  CPU-only; these run under `-m "not gpu"`."""
import os
import sys

import pytest
import torch

from conftest import load_golden, rel_l2

from oracle import seva_ref as O
from oracle import sampling_ref as S
from seva import synthetic as synth

TOL = 2e-5  # fp32 CPU vs fp32 CPU; differences are accumulation order only


def _sd(keys_shapes, seed=0):
    return synth.synth_state_dict(keys_shapes, seed)


def _block_shapes(kind):
    """Shapes of the block-level modules used in g2_* (C=128, heads=2, ctx 1024)."""
    C = 128
    att = lambda ctx: {"to_q.weight": (C, C), "to_k.weight": (C, ctx), "to_v.weight": (C, ctx),
                       "to_out.0.weight": (C, C), "to_out.0.bias": (C,)}
    ff = lambda cin, cout: {"net.0.proj.weight": (8 * cin, cin), "net.0.proj.bias": (8 * cin,),
                            "net.2.weight": (cout, 4 * cin), "net.2.bias": (cout,)}
    ln = lambda: {"weight": (C,), "bias": (C,)}
    pre = lambda p, d: {f"{p}.{k}": v for k, v in d.items()}
    if kind == "attn_self":
        return att(C)
    if kind == "attn_cross":
        return att(1024)
    if kind == "ff":
        return ff(C, C)
    if kind == "tblock":
        d = {}
        d.update(pre("attn1", att(C))); d.update(pre("ff", ff(C, C))); d.update(pre("attn2", att(1024)))
        for n in ("norm1", "norm2", "norm3"):
            d.update(pre(n, ln()))
        return d
    if kind == "timemix":
        d = {}
        d.update(pre("norm_in", ln())); d.update(pre("ff_in", ff(C, C)))
        d.update(pre("attn1", att(C))); d.update(pre("ff", ff(C, C))); d.update(pre("attn2", att(1024)))
        for n in ("norm1", "norm2", "norm3"):
            d.update(pre(n, ln()))
        return d
    if kind == "mvt":
        d = {"norm.weight": (C,), "norm.bias": (C,), "proj_in.weight": (C, C), "proj_in.bias": (C,)}
        d.update(pre("transformer_blocks.0", _block_shapes("tblock")))
        d.update({"proj_out.weight": (C, C), "proj_out.bias": (C,)})
        d.update(pre("time_mix_blocks.0", _block_shapes("timemix")))
        return d
    raise KeyError(kind)


def _resblock_shapes(cin, cout, emb=256):
    d = {"in_layers.0.weight": (cin,), "in_layers.0.bias": (cin,),
         "in_layers.2.weight": (cout, cin, 3, 3), "in_layers.2.bias": (cout,),
         "emb_layers.1.weight": (cout, emb), "emb_layers.1.bias": (cout,),
         "dense_emb_layers.0.weight": (2 * cin, 6, 1, 1), "dense_emb_layers.0.bias": (2 * cin,),
         "out_layers.0.weight": (cout,), "out_layers.0.bias": (cout,),
         "out_layers.3.weight": (cout, cout, 3, 3), "out_layers.3.bias": (cout,)}
    if cin != cout:
        d.update({"skip_connection.weight": (cout, cin, 1, 1), "skip_connection.bias": (cout,)})
    return d


def _p(sd, prefix="m"):
    return {f"{prefix}.{k}": v for k, v in sd.items()}


def test_g1_schedules():
    g = load_golden("g1_schedules")
    assert torch.equal(S.ddpm_sigmas(4), g["sig4"])
    assert torch.equal(S.ddpm_sigmas(50), g["sig50"])
    assert torch.equal(S.ddpm_sigmas(1000), g["sig1000"])
    assert torch.equal(S.ddpm_sigmas(50, append_zero=False, flip=True), g["sig50_noappend_flip"])
    table = S.ddpm_sigmas(1000, append_zero=False, flip=True)
    assert torch.equal(table, g["table"])
    assert torch.equal(S.sigma_to_idx(table, g["sig50"][:-1]), g["idx50"])
    assert torch.equal(S.sigma_to_idx(table, g["sig50"][:-1] + 1e-6), g["idx50_hat"])
    # known answers quoted in SURVEY.md §8a/A2
    assert abs(float(g["sig50"][0]) - 84.916) < 1e-2 and abs(float(g["sig4"][1]) - 24.205) < 1e-2
    assert int(g["idx50"][0]) == 999 and int(g["idx50"][1]) == 979 and int(g["idx50"][-1]) == 19


def test_g2_attention_ff():
    g = load_golden("g2_attn_self")
    sd = _p(_sd(_block_shapes("attn_self"), 0))
    assert rel_l2(O.attention(sd, "m", g["x"], None), g["y"]) < TOL
    g = load_golden("g2_attn_cross")
    sd = _p(_sd(_block_shapes("attn_cross"), 1))
    assert rel_l2(O.attention(sd, "m", g["x"], g["ctx1"]), g["y1"]) < TOL
    assert rel_l2(O.attention(sd, "m", g["x"], g["ctx3"]), g["y3"]) < TOL
    g = load_golden("g2_ff")
    sd = _p(_sd(_block_shapes("ff"), 2))
    assert rel_l2(O.feedforward(sd, "m", g["x"]), g["y"]) < TOL


def test_g2_transformer_blocks():
    g = load_golden("g2_tblock")
    sd = _p(_sd(_block_shapes("tblock"), 3))
    assert rel_l2(O.transformer_block(sd, "m", g["x"], g["ctx"]), g["y"]) < TOL
    g = load_golden("g2_timemix")
    sd = _p(_sd(_block_shapes("timemix"), 4))
    assert rel_l2(O.timemix_block(sd, "m", g["x"], g["ctx"], int(g["T"])), g["y"]) < TOL
    for tag, joint in (("joint", True), ("frame", False)):
        g = load_golden(f"g2_mvt_{tag}")
        sd = _p(_sd(_block_shapes("mvt"), 6))
        y = O.multiview_transformer(sd, "m", g["x"], g["ctx"], int(g["T"]), joint)
        assert rel_l2(y, g["y"]) < TOL


def test_g2_conv_blocks():
    for tag, cin, cout in (("id", 64, 64), ("skip", 96, 64)):
        g = load_golden(f"g2_resblock_{tag}")
        sd = _p(_sd(_resblock_shapes(cin, cout), 10))
        assert rel_l2(O.resblock(sd, "m", g["x"], g["emb"], g["dense"]), g["y"]) < TOL
    g = load_golden("g2_updown")
    up = _p(_sd({"conv.weight": (64, 64, 3, 3), "conv.bias": (64,)}, 12))
    dn = _p(_sd({"op.weight": (64, 64, 3, 3), "op.bias": (64,)}, 13))
    assert rel_l2(O.upsample(up, "m", g["x"]), g["up"]) < TOL
    assert rel_l2(O.downsample(dn, "m", g["x"]), g["down"]) < TOL
    g = load_golden("g2_temb")
    assert torch.allclose(O.timestep_embedding(g["t"], 320), g["y320"], atol=1e-6)
    assert torch.allclose(O.timestep_embedding(g["t"], 64), g["y64"], atol=1e-6)


def _golden_shapes(tag):
    g = load_golden(f"g0_keys_{tag}")
    return {str(k): tuple(int(s) for s in str(v).split(",")) for k, v in zip(g["keys"], g["shapes"])}


def test_g3_tiny_forward():
    g = load_golden("g3_tiny_forward")
    sd = _sd(_golden_shapes("tiny"))
    c = {"crossattn": g["crossattn"], "concat": g["concat"], "dense_vector": g["dense_vector"]}
    y = O.sgm_wrapper_forward(sd, g["x"], g["t"], c, int(g["T"]))
    assert rel_l2(y, g["y"]) < 5e-5


def _tiny_network():
    sd = _sd(_golden_shapes("tiny"))
    return lambda x, idx, c, num_frames: O.sgm_wrapper_forward(sd, x, idx, c, num_frames)


def test_g5_denoiser_g6_guiders():
    g = load_golden("g5_denoiser")
    T = int(g["T"])
    sc = synth.synth_scene(T, tuple(g["x"].shape[-2:]), (0,), seed=int(g["seed"]))
    cat = {k: torch.cat((sc["uc"][k], sc["cond"][k]), 0) for k in sc["cond"]}
    table = S.ddpm_sigmas(1000, append_zero=False, flip=True)
    y = S.denoise(_tiny_network(), table, torch.cat([g["x"]] * 2), torch.cat([g["sigma"]] * 2), cat,
                  num_frames=T)
    assert rel_l2(y, g["y"]) < 5e-5
    g = load_golden("g6_guiders")
    T = int(g["T"])
    args = (g["c2w"], g["K"], g["mask"].bool(), T)
    assert rel_l2(S.guide(g["d"], 2.0, 0, 1.2, *args), g["y0"]) < 1e-6
    assert rel_l2(S.guide(g["d"], 2.0, 1, 1.2, *args), g["y1"]) < 1e-6
    assert rel_l2(S.guide(g["d"], 2.0, 2, 1.2, *args), g["y2"]) < 1e-6
    # the close frame (index 2 == input pose) must get cfg_min, the far ones cfg
    s = S.multiview_scale(2.0, 1.2, g["c2w"], g["K"], g["mask"].bool())
    assert s.tolist() == pytest.approx([1.2, 2.0, 1.2, 2.0])


def test_g7_tiny_loop():
    g = load_golden("g7_loop_tiny")
    T, hw, steps = int(g["T"]), int(g["hw"]), int(g["steps"])
    sc = synth.synth_scene(T, (hw, hw), (0,), seed=int(g["scene_seed"]))
    y = S.euler_edm_sample(_tiny_network(), sc["noise"], sc["cond"], sc["uc"], steps, 2.0,
                           list(g["eps"]), guider=1, cfg_min=1.2, c2w=sc["c2w"], K=sc["K"],
                           input_frame_mask=sc["input_frame_mask"])
    assert rel_l2(y, g["y"]) < 1e-4


@pytest.mark.slow
@pytest.mark.skipif(os.environ.get("SEVA_SLOW", "0") != "1", reason="1.3B oracle: set SEVA_SLOW=1")
def test_g4_full_forward():
    g = load_golden("g4_full_forward")
    sd = _sd(_golden_shapes("full"))
    c = {"crossattn": g["crossattn"], "concat": g["concat"], "dense_vector": g["dense_vector"]}
    y = O.sgm_wrapper_forward(sd, g["x"], g["t"], c, int(g["T"]))
    assert rel_l2(y, g["y"]) < 5e-5


# ---- conditioning geometry -----------------------------------------------------------------------
def test_geometry_plucker_matches_reference_golden():
    from oracle import geometry_ref as G
    g = load_golden("g8_plucker")
    a = G.get_plucker_coordinates(torch.tensor(g["a_w2c"][0]), torch.tensor(g["a_w2c"]), None, target_size=(9, 9))
    assert torch.allclose(a, torch.tensor(g["a_out"]), atol=2e-6, rtol=0)
    b = G.get_plucker_coordinates(torch.tensor(g["b_w2c"][1]), torch.tensor(g["b_w2c"]), torch.tensor(g["b_K"]), target_size=(12, 20))
    assert torch.allclose(b, torch.tensor(g["b_out"]), atol=2e-6, rtol=0)
    c = G.get_plucker_coordinates(torch.tensor(g["c_w2c"][0]), torch.tensor(g["c_w2c"]), torch.tensor(g["c_K"]), target_size=(8, 6))
    assert torch.allclose(c, torch.tensor(g["c_out"]), atol=2e-6, rtol=0)


def test_geometry_value_dict_matches_reference_golden():
    from oracle import geometry_ref as G
    g = load_golden("g8_value_dict")
    for tag in ("a", "b", "c"):
        H, W = (int(v) for v in g[f"{tag}_HW"])
        c2w_in = torch.tensor(g[f"{tag}_c2w_in"])
        T = c2w_in.shape[0]
        imgs = torch.zeros(T, 3, H, W)
        vd = G.get_value_dict(imgs, [int(i) for i in g[f"{tag}_in_idx"]], c2w_in, torch.tensor(g[f"{tag}_K"]),
                              G.to_hom_pose(c2w_in), 2.0)
        assert torch.equal(vd["cond_frames_mask"], torch.tensor(g[f"{tag}_mask"]))
        assert torch.allclose(vd["c2w"], torch.tensor(g[f"{tag}_c2w"]), atol=1e-6, rtol=0), tag
        assert torch.allclose(vd["plucker_coordinate"], torch.tensor(g[f"{tag}_plucker"]), atol=3e-6, rtol=0), tag
