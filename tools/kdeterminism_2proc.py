#!/usr/bin/env python3
"""Forward determinism of the tiny UNet while ANOTHER process runs the same thing on the same card: which buffer differs first?"""
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
ref = None
bad = {}
nbad = 0
for r in range(int(os.environ.get("REPS", "60"))):
    out = eng.forward(x, concat, t, y, dense, T).clone()
    snap = {k[0] + str(k[1]): v.clone() for k, v in eng.arena.bufs.items() if k[0].startswith("out:") or k[0] in ("head", "emb_all", "ctxvec", "x16")}
    if ref is None:
        ref, ref_out = snap, out
        continue
    if not torch.equal(out, ref_out):
        nbad += 1
        for name in snap:  # arena order = first-use order
            if not torch.equal(snap[name], ref[name]):
                bad[name] = bad.get(name, 0) + 1
                break
print(os.environ.get("TAG"), "runs differing from run 0:", nbad, "| first differing buffer (count):", bad, flush=True)
'''
procs = []
for tag in ("procA", "procB"):
    e = dict(os.environ, SEVA_ROOT=ROOT, TAG=tag, **{k: v for k, v in [a.split("=") for a in sys.argv[1:]]})
    procs.append(subprocess.Popen([sys.executable, "-c", CHILD], env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
for p in procs:
    o, er = p.communicate()
    print(o.strip() or er.strip()[-600:], flush=True)
