#!/usr/bin/env python3
"""Does running the two CFG halves of one network call as two CONCURRENT streams beat one batch-2T launch sequence?

The halves [uncond; cond] never interact inside the network (per-frame ops are per sample, the joint / temporal attention is per
half), so one call can be issued as two independent half-batch launch sequences whose kernels overlap on the chip: the drain of
one kernel (partly filled last round of workgroups) is covered by the other stream's kernel.  Measured here, at the headline
shape, eager and as one hipGraph with two branches:
  (a) one engine, batch 2T, one stream;
  (b) two engines (own activation arenas, same weights), batch T each, two streams.
"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "stable-virtual-camera_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import warnings; warnings.simplefilter("ignore")
import torch
from test_model_gpu import _build
from seva import synthetic as synth
from seva._engine import SevaEngine

T = int(os.environ.get("KCONC_T", "21"))
HW = int(os.environ.get("KCONC_HW", "72"))
REPS = int(os.environ.get("KCONC_REPS", "5"))
dev = torch.device("cuda:0")
net, _ = _build("full", dev)
sc = synth.synth_scene(T, (HW, HW), (0,), seed=500)
x = torch.randn(2 * T, 4, HW, HW, generator=torch.Generator().manual_seed(501)).to(dev)
c = {k: torch.cat((sc["uc"][k], sc["cond"][k]), 0).to(dev) for k in ("crossattn", "concat", "dense_vector")}
t = torch.full((2 * T,), 979, dtype=torch.int64, device=dev)

eng = SevaEngine(net); eng.use_graph = False
ea = SevaEngine(net); ea.use_graph = False
eb = SevaEngine(net); eb.use_graph = False
sa, sb = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
out_full = torch.empty(2 * T, 4, HW, HW, device=dev)
out_a = torch.empty(T, 4, HW, HW, device=dev)
out_b = torch.empty(T, 4, HW, HW, device=dev)


def full():
    eng.forward(x, c["concat"], t, c["crossattn"], c["dense_vector"], T, out=out_full)


def halves():
    cur = torch.cuda.current_stream(dev)
    sa.wait_stream(cur); sb.wait_stream(cur)
    with torch.cuda.stream(sa):
        ea.forward(x[:T], c["concat"][:T], t[:T], c["crossattn"][:T], c["dense_vector"][:T], T, out=out_a)
    with torch.cuda.stream(sb):
        eb.forward(x[T:], c["concat"][T:], t[T:], c["crossattn"][T:], c["dense_vector"][T:], T, out=out_b)
    cur.wait_stream(sa); cur.wait_stream(sb)


def timeit(fn, reps=REPS):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    ts.sort()
    return ts[len(ts) // 2], ts[0]


with torch.no_grad():
    full(); halves(); torch.cuda.synchronize()
    same = torch.equal(out_full[:T], out_a) and torch.equal(out_full[T:], out_b)
    print(f"halves bit-identical to the full batch: {same}", flush=True)
    for rnd in range(2):
        m, lo = timeit(full); print(f"[eager] one stream, batch {2 * T}: median {m:.2f} ms, min {lo:.2f}", flush=True)
        m, lo = timeit(halves); print(f"[eager] two streams, batch {T} each: median {m:.2f} ms, min {lo:.2f}", flush=True)
    # graphs
    g1 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g1, capture_error_mode="thread_local"):
        full()
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2, capture_error_mode="thread_local"):
        halves()
    for rnd in range(3):
        m, lo = timeit(g1.replay, 10); print(f"[graph] one stream, batch {2 * T}: median {m:.2f} ms, min {lo:.2f}", flush=True)
        m, lo = timeit(g2.replay, 10); print(f"[graph] two branches, batch {T} each: median {m:.2f} ms, min {lo:.2f}", flush=True)
