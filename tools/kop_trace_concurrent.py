#!/usr/bin/env python3
"""Which operator's OUTPUT changes between identical forwards of the tiny UNet while a second stream of the same process keeps the
card busy?  Every seva.ops call of the main thread is followed by a device sync and a byte-sum of all its tensor arguments; the first
call whose sums differ from the first pass, with equal sums for everything before it, is the operator that was not repeatable."""
import faulthandler, gc, inspect, os, sys, threading, time
faulthandler.enable()  # a SIGABRT / SIGSEGV of this process leaves the Python stacks of all threads on stderr
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


def load(net2):
    """The load generator: forwards of a SECOND engine on a second stream.  Everything it creates on the device dies HERE, before the
    thread returns: round 3's TAG=full run ended in `terminate called without an active exception` at interpreter exit, after its results
    were printed (profiles/r03_concurrency_op_trace_1p3b.log); the tiny run never did.  What the full run had and the tiny one had not: the
    1.3 B synthetic state_dict was generated INSIDE this thread (CPU tensors large enough for torch's intra-op thread team, owned by a
    thread that no longer exists at exit) and a 14 GB arena, its graph-capture pool and the side stream's events were left to the
    interpreter's teardown.  Now the weights are built on the main thread and this thread frees what it made."""
    s2 = torch.cuda.Stream()
    with torch.cuda.stream(s2):
        eng = net2.engine(); eng.use_graph = False
        a = inputs(11)
        while not stop:
            for _ in range(10): eng.forward(*a, T)
            s2.synchronize()
        s2.synchronize()
        del eng, a
    torch.cuda.synchronize()


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
    net2, _ = _build(TAG, dev)  # on the MAIN thread (see load())
    th = threading.Thread(target=load, args=(net2,)); th.start(); time.sleep(10)
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
# explicit teardown in a defined order, on the main thread, while the HIP runtime is certainly alive
del eng, net, a
if th: del net2
gc.collect()
torch.cuda.synchronize()
torch.cuda.empty_cache()
print("teardown: engines, arenas and streams released; exiting normally", flush=True)
