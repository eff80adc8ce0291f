#!/usr/bin/env python3
"""Which operator's OUTPUT changes between identical forwards of the tiny UNet while a second stream of the same process keeps the
card busy?  Every seva.ops call of the main thread is followed by a device sync and a byte-sum of all its tensor arguments; the first
call whose sums differ from the first pass, with equal sums for everything before it, is the operator that was not repeatable."""
import inspect, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stable-virtual-camera_amd")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import torch
from test_model_gpu import _build
from seva import ops

dev = torch.device("cuda:0")
TAG = os.environ.get("MODEL", "tiny")
T, hw = int(os.environ.get("T", "21")), int(os.environ.get("HW", "16"))
MAIN = threading.current_thread()


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
        a = inputs(11)
        while not stop:
            for _ in range(10): eng.forward(*a, T)
            s2.synchronize()


cur = []


def hsh(v):
    if not isinstance(v, torch.Tensor) or not v.is_cuda or not v.numel(): return None
    return (tuple(v.shape), str(v.dtype), int(v.contiguous().view(torch.uint8).to(torch.int64).sum().item()))


SKIP_KW = ("splitk_ws", "split_ws", "workspace")


def wrap(name, fn):
    def w(*a, **k):
        r = fn(*a, **k)
        if threading.current_thread() is not MAIN: return r
        torch.cuda.synchronize()
        cur.append((name, [(j, hsh(v)) for j, v in enumerate(a) if not (name == "groupnorm" and j == 5)] +
                    [(kk, hsh(vv)) for kk, vv in sorted(k.items()) if isinstance(vv, torch.Tensor) and kk not in SKIP_KW]))
        return r
    return w


NOT_OPS = ("check_handoffs", "set_knob", "prof_enable", "prof_collect", "channel_stats_shape", "splitk_workspace", "attention_split_workspace_numel",
           "groupnorm_workspace", "quantize_weight_fp8", "dequantize_weight_fp8", "to_fp8")
for name in dir(ops):
    fn = getattr(ops, name)
    if inspect.isfunction(fn) and fn.__module__ == "seva.ops" and not name.startswith("_") and name not in NOT_OPS:
        setattr(ops, name, wrap(name, fn))

net, _ = _build(TAG, dev)
eng = net.engine(); eng.use_graph = False
a = inputs(5)
th = None
if os.environ.get("LOAD", "1") == "1":
    th = threading.Thread(target=load); th.start(); time.sleep(10)
first = {}
for p in range(int(os.environ.get("PASSES", "60"))):
    cur.clear()
    eng.forward(*a, T)
    if p == 0:
        ref = list(cur); continue
    for i, (c, r) in enumerate(zip(cur, ref)):
        if c != r:
            diff = [u[0] for u, v in zip(c[1], r[1]) if u != v]
            key = (c[0], tuple(diff), tuple(str(u[1][0]) for u, v in zip(c[1], r[1]) if u != v))
            first[key] = first.get(key, 0) + 1
            if first[key] == 1: print("pass", p, "call", i, c[0], "differing arguments", diff, "all arguments:", [(u[0], u[1][0] if u[1] else None) for u in c[1]], flush=True)
            break
stop = True
if th: th.join()
print("first non-repeatable call per pass (operator, differing argument, shape): count", flush=True)
for k, v in sorted(first.items(), key=lambda kv: -kv[1]): print("  ", k, v, flush=True)
print("passes:", int(os.environ.get("PASSES", "60")), "calls per pass:", len(ref), flush=True)
