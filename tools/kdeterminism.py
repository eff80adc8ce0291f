#!/usr/bin/env python3
"""Is the tiny UNet's forward bitwise repeatable at T=21, 16x16 (CFG batch 42)?  One subprocess per knob setting."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys
ROOT = os.environ["SEVA_ROOT"]
sys.path.insert(0, os.path.join(ROOT, "stable-virtual-camera_amd")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import torch
from test_model_gpu import _build
dev = torch.device("cuda:0")
net, _ = _build("tiny", dev)
eng = net.engine(); eng.use_graph = False
T, hw = 21, 16
g = torch.Generator().manual_seed(5); n = 2 * T
x = (torch.randn(n, 4, hw, hw, generator=g) * 10).to(dev); concat = torch.randn(n, 7, hw, hw, generator=g).to(dev)
t = torch.full((n,), 700, dtype=torch.int64, device=dev); y = torch.randn(n, 1, 1024, generator=g).to(dev)
dense = torch.randn(n, 6, hw * 8, hw * 8, generator=g).to(dev)
outs = []
layers = []
for r in range(6):
    outs.append(eng.forward(x, concat, t, y, dense, T).clone())
    layers.append({k[0]: v.clone() for k, v in eng.arena.bufs.items() if k[0].startswith("out:") or k[0] in ("res_mid",)})
d = [float((outs[i] - outs[0]).abs().max()) for i in range(1, 6)]
first = None
for name in layers[0]:
    if any(not torch.equal(layers[i][name], layers[0][name]) for i in range(1, 6)):
        first = name if first is None else first
names = [nm for nm in layers[0] if any(not torch.equal(layers[i][nm], layers[0][nm]) for i in range(1, 6))]
print(os.environ.get("TAG"), "max diff vs run 0:", ["%.2e" % v for v in d], "| differing buffers:", names[:6])
'''
for tag, env in [("default", {}), ("SPLIT=none", {"SEVA_SPLIT_PRECISION": "none"}), ("FOLD=0", {"SEVA_FOLD_SKIP": "0"}),
                 ("SPLITK=0", {"SEVA_CONV_SPLITK": "0"}), ("GNSTATS=0", {"SEVA_GN_FUSED_STATS": "0"}),
                 ("FF_FUSED=0", {"SEVA_FF_FUSED": "0"}), ("SPLIT=none,FOLD=0,SPLITK=0", {"SEVA_SPLIT_PRECISION": "none", "SEVA_FOLD_SKIP": "0", "SEVA_CONV_SPLITK": "0"})]:
    e = dict(os.environ, SEVA_ROOT=ROOT, TAG=tag, **env)
    r = subprocess.run([sys.executable, "-c", CHILD], env=e, capture_output=True, text=True)
    print((r.stdout.strip() or r.stderr.strip()[-400:]), flush=True)
