#!/usr/bin/env python3
"""Forward repeatability of the tiny UNet while OTHER work runs on the card in the same process: a second thread drives a second
copy of the network on its own stream.  Prints how many of REPS forwards differ from the first and which buffer differs first.
(One process, no load: every run is bit-equal -- tools/kdeterminism.py.)  LOAD=0 runs without the second stream."""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stable-virtual-camera_amd")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import torch
from test_model_gpu import _build

dev = torch.device("cuda:0")
TAG = os.environ.get("MODEL", "tiny")   # MODEL=full HW=72: the 1.3 B network at the headline shape
T, hw = int(os.environ.get("T", "21")), int(os.environ.get("HW", "16"))


def inputs(seed):
    g = torch.Generator().manual_seed(seed); n = 2 * T
    return ((torch.randn(n, 4, hw, hw, generator=g) * 10).to(dev), torch.randn(n, 7, hw, hw, generator=g).to(dev),
            torch.full((n,), 700, dtype=torch.int64, device=dev), torch.randn(n, 1, 1024, generator=g).to(dev),
            torch.randn(n, 6, hw * 8, hw * 8, generator=g).to(dev))


stop = False


def load():
    s2 = torch.cuda.Stream()
    with torch.cuda.stream(s2):
        net, _ = _build(TAG, dev)
        eng = net.engine(); eng.use_graph = False
        x, concat, t, y, dense = inputs(11)
        while not stop:
            for _ in range(10): eng.forward(x, concat, t, y, dense, T)
            s2.synchronize()


net, _ = _build(TAG, dev)
eng = net.engine(); eng.use_graph = False
x, concat, t, y, dense = inputs(5)
th = None
if os.environ.get("LOAD", "1") == "1":
    th = threading.Thread(target=load); th.start(); time.sleep(10)
ref = None; bad = {}; nbad = 0
reps = int(os.environ.get("REPS", "150"))
for r in range(reps):
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
    if r % 10 == 0: print("run", r, "differing so far", nbad, flush=True)
stop = True
if th: th.join()
print(f"forwards differing from the first: {nbad} of {reps - 1} | first differing buffer (count): {bad}", flush=True)
