#!/usr/bin/env python3
"""Two identical eager sampler runs (tiny UNet, T=21, 16x16): where do they first differ?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stable-virtual-camera_amd")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import torch
from test_model_gpu import _build
from seva import sampling as S
from seva import synthetic as synth
from seva.model import SGMWrapper
dev = torch.device("cuda:0")
net, _ = _build("tiny", dev)
net.engine().use_graph = False
wrap = SGMWrapper(net)
T, hw, steps = 21, 16, 3
sc = synth.synth_scene(T, (hw, hw), (0,), seed=7)
disc = S.DDPMDiscretization()


def run(zero_eps):
    rec = {}
    den = S.DiscreteDenoiser(disc, num_idx=1000, device=dev)
    sampler = S.EulerEDMSampler(disc, S.MultiviewCFG(1.2), num_steps=steps, verbose=False, device=dev, s_churn=0.0)
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234)
    sampler.noise_fn = (lambda x: torch.zeros_like(x)) if zero_eps else (lambda x: torch.randn(x.shape, generator=gen, device=x.device, dtype=x.dtype))
    sampler._step_graphs.disabled = True
    cond = {k: v.to(dev) for k, v in sc["cond"].items()}
    uc = {k: v.to(dev) for k, v in sc["uc"].items()}
    kw = dict(c2w=sc["c2w"].to(dev), K=sc["K"].to(dev), input_frame_mask=sc["input_frame_mask"].to(dev))
    calls = []

    def denoise(a, s, c):
        out = den(wrap, a, s, c, num_frames=T)
        calls.append((a.clone(), s.clone(), out.clone()))
        return out

    x, s_in, sigmas, num_sigmas, cond, uc = sampler.prepare_sampling_loop(sc["noise"].to(dev).clone(), cond, uc, None)
    rec["x0"] = x.clone()
    xs = []
    for i in range(num_sigmas - 1):
        x = sampler.sampler_step(s_in * sigmas[i], s_in * sigmas[i + 1], denoise, x, 2.0, cond, uc, 0.0, **kw)
        xs.append(x.clone())
    rec["xs"], rec["calls"] = xs, calls
    rec["fs"] = sampler.guider.frame_scale(calls[0][2], s_in * sigmas[0], 2.0, **kw).flatten().clone()
    return rec


with torch.no_grad():
    for zero in (False, True):
        a, b = run(zero), run(zero)
        d = lambda u, v: float((u - v).abs().max())
        print(f"zero_eps={zero}: x0 {d(a['x0'], b['x0']):.2e}; step-0 network input {d(a['calls'][0][0], b['calls'][0][0]):.2e} sigma {d(a['calls'][0][1], b['calls'][0][1]):.2e} "
              f"denoised {d(a['calls'][0][2], b['calls'][0][2]):.2e}; frame scale {d(a['fs'], b['fs']):.2e}; x after steps {[f'{d(u, v):.2e}' for u, v in zip(a['xs'], b['xs'])]}")
        print("   frame scale:", [round(float(v), 3) for v in a["fs"][:6]], "...")
