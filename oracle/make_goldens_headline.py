"""Goldens at the shapes the headline metric is quoted on (dev container only; minutes of CPU).

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_goldens_headline.py [--only g9_T8,...]

Runs the REFERENCE (`/root/reference`, CPU fp32, imported read-only exactly as
`oracle/make_goldens.py` does) at latent 72x72 (= 576x576 pixels) with the name-keyed
synthetic 1.3B weights and stores OUTPUTS only -- every input regenerates from
`seva.synthetic` seeds (`headline_inputs` below is mirrored in tests/test_headline_gpu.py):

  g9_T8_forward   one SGMWrapper call, T=8,  CFG batch 16  (BASELINE config 2)   1.3 MB
  g9_T21_forward  one SGMWrapper call, T=21, CFG batch 42  (the metric's shape)  3.5 MB
  g9_T24_forward  one SGMWrapper call, T=24, CFG batch 48  (BASELINE config 3)   4.0 MB
  g9_T21_step     one EulerEDMSampler.sampler_step at T=21 (sigma index 30 of the 50-step
                  schedule, MultiviewCFG(1.2), scale 2.0) with the recorded eps       1.7 MB

Reference call sites: seva/model.py:176-234 (forward), seva/sampling.py:347-368 (step).
"""

from __future__ import annotations

import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_goldens as mg  # noqa: E402  (imports the reference + seva.synthetic)

rmodel, rsamp, synth = mg.rmodel, mg.rsamp, mg.synth

HW = 72
FORWARD_SEEDS = {8: 400, 21: 500, 24: 600}
STEP_T, STEP_SCENE_SEED, STEP_X_SEED, STEP_EPS_SEED, STEP_INDEX = 21, 23, 701, 702, 30


def headline_inputs(T, seed):
    """x (2T,4,72,72), t (2T,) int64, cond dict [uncond; cond] -- same recipe as G3/G4."""
    return mg._wrapper_inputs(T, HW, seed)


def full_net():
    with torch.device("meta"):
        meta = rmodel.Seva(rmodel.SevaParams())
    shapes = {k: tuple(v.shape) for k, v in meta.state_dict().items()}
    sd = synth.synth_state_dict(shapes, 0)
    net = rmodel.Seva(rmodel.SevaParams())
    net.load_state_dict(sd, strict=True, assign=True)
    return net.eval()


@torch.no_grad()
def forward_golden(net, T):
    x, t, c = headline_inputs(T, FORWARD_SEEDS[T])
    wrap = rmodel.SGMWrapper(net)
    t0 = time.time()
    y = wrap(x, t, c, num_frames=T)
    dt = time.time() - t0
    print(f"  reference 1.3B forward T={T}, 72x72, B={2 * T}: {dt:.1f}s", flush=True)
    mg.save(f"g9_T{T}_forward", y=y, T=T, hw=HW, seed=FORWARD_SEEDS[T], ref_seconds=dt,
            ref_threads=torch.get_num_threads())


@torch.no_grad()
def step_golden(net):
    T = STEP_T
    disc = rsamp.DDPMDiscretization()
    den = rsamp.DiscreteDenoiser(disc, num_idx=1000, device="cpu")
    sampler = rsamp.EulerEDMSampler(disc, rsamp.MultiviewCFG(1.2), num_steps=50,
                                    verbose=False, device="cpu", s_churn=0.0)
    sc = synth.synth_scene(T, (HW, HW), (0,), seed=STEP_SCENE_SEED)
    sigmas = disc(50)
    sigma, next_sigma = sigmas[STEP_INDEX], sigmas[STEP_INDEX + 1]
    # a plausible mid-trajectory state: clean-ish signal + sigma * noise
    x = mg.rnd(T, 4, HW, HW, seed=STEP_X_SEED) * float((sigma ** 2 + 1.0) ** 0.5)
    eps = mg.rnd(T, 4, HW, HW, seed=STEP_EPS_SEED)
    orig = mg._patch_randn([eps])
    wrap = rmodel.SGMWrapper(net)
    s_in = x.new_ones([T])
    t0 = time.time()
    try:
        out = sampler.sampler_step(
            s_in * sigma, s_in * next_sigma,
            lambda xx, ss, cc: den(wrap, xx, ss, cc, num_frames=T),
            x.clone(), 2.0, sc["cond"], sc["uc"], 0.0,
            c2w=sc["c2w"], K=sc["K"], input_frame_mask=sc["input_frame_mask"],
        )
    finally:
        rsamp.torch.randn_like = orig
    dt = time.time() - t0
    print(f"  reference sampler_step T={T}: {dt:.1f}s", flush=True)
    mg.save("g9_T21_step", y=out, T=T, hw=HW, sigma=sigma, next_sigma=next_sigma,
            step_index=STEP_INDEX, scene_seed=STEP_SCENE_SEED, x_seed=STEP_X_SEED,
            eps_seed=STEP_EPS_SEED, ref_seconds=dt)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="g9_T8,g9_T21,g9_T21_step,g9_T24")
    args = ap.parse_args()
    want = set(args.only.split(","))
    t0 = time.time()
    net = full_net()
    print(f"  synth 1.3B weights loaded in {time.time() - t0:.1f}s", flush=True)
    if "g9_T8" in want:
        forward_golden(net, 8)
    if "g9_T21" in want:
        forward_golden(net, 21)
    if "g9_T21_step" in want:
        step_golden(net)
    if "g9_T24" in want:
        forward_golden(net, 24)


if __name__ == "__main__":
    main()
