#!/usr/bin/env python3
"""Single-process check of what CFG-split relies on: denoiser(full CFG batch) == cat(denoiser(uncond half), denoiser(cond half)),
through DiscreteDenoiser + SGMWrapper on the HIP path, with and without the network hipGraph."""
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
wrap = SGMWrapper(net)
T, hw = 21, 16
sc = synth.synth_scene(T, (hw, hw), (0,), seed=7)
cond = {k: v.to(dev) for k, v in sc["cond"].items()}
uc = {k: v.to(dev) for k, v in sc["uc"].items()}
disc = S.DDPMDiscretization()
den = S.DiscreteDenoiser(disc, num_idx=1000, device=dev)
guider = S.MultiviewCFG(1.2)
x = (torch.randn(T, 4, hw, hw, generator=torch.Generator().manual_seed(1)) * 10).to(dev)
sigma = torch.full((T,), 9.3527, device=dev)
for graph in (True, False):
    net.engine().use_graph = graph
    with torch.no_grad():
        for rep in range(3):
            xx, ss, cc = guider.prepare_inputs(x, sigma, cond, uc)
            full = den(wrap, xx, ss, dict(cc), num_frames=T).clone()
            halves = []
            for h in (0, 1):
                sl = slice(h * T, (h + 1) * T)
                xx, ss, cc = guider.prepare_inputs(x, sigma, cond, uc)
                ch = {k: (v[sl] if isinstance(v, torch.Tensor) and v.dim() > 0 and v.shape[0] == 2 * T else v) for k, v in cc.items()}
                halves.append(den(wrap, xx[sl], ss[sl], ch, num_frames=T).clone())
            both = torch.cat(halves)
            print(f"graph={graph} rep {rep}: equal = {torch.equal(both, full)}  max diff {float((both - full).abs().max()):.3e}  "
                  f"half0 {float((both[:T] - full[:T]).abs().max()):.2e} half1 {float((both[T:] - full[T:]).abs().max()):.2e}")
