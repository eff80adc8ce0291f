"""Whole-step hipGraph for `EulerEDMSampler.sampler_step` (north_star: "sampler loop captured in hipGraph").

One denoising step -- noise perturbation, CFG input assembly, nearest-sigma lookup, the ~600 kernels of the
CFG-batched network call, denoiser combine, guidance and the Euler update (reference seva/sampling.py:347-368
plus everything it calls) -- is captured ONCE per trajectory into a hipGraph and replayed for the remaining steps.
Per-step values enter through static device buffers (x, eps, sigma, next_sigma), so the graph is the step as a
function, not a recording of one step's numbers.

PyTorch is used for what the task assigns to it -- device memory and streams: `torch.cuda.CUDAGraph` (= hipGraph on
ROCm) drives hipStreamBeginCapture / hipGraphInstantiate / hipGraphLaunch and gives the captured region a private
allocator pool, so the few host-sized torch ops inside the step (`torch.cat` of the cond dictionaries, the
argmin over the 1000-entry sigma table) keep stable addresses across replays.  All latent-sized math inside the
graph is libseva_hip.so kernels launched on the capturing stream.

What stays OUTSIDE the graph on purpose: the per-step Gaussian draw (`noise_fn`, default `torch.randn_like`): the
generator's Philox offset must advance per step exactly as in an eager run (and tests inject recorded eps), so
eps is drawn eagerly and copied into a static buffer -- graph and eager runs are bit-identical.
"""

from __future__ import annotations

import os

import torch


def enabled() -> bool:
    """SEVA_STEPGRAPH=0 disables whole-step capture (the network call alone is then replayed, SEVA_HIPGRAPH)."""
    return os.environ.get("SEVA_STEPGRAPH", "1") != "0" and os.environ.get("SEVA_HIPGRAPH", "1") != "0"


class StepGraph:
    """fn(*tensors) -> tensor, captured once; `inputs` are example tensors whose values are refreshed per replay."""

    def __init__(self, fn, inputs):
        dev = inputs[0].device
        # static buffers must be ordinary tensors even when the caller runs under torch.inference_mode()
        # (reference eval.py:1242): an inference tensor cannot be updated in place from outside that mode
        with torch.inference_mode(False):
            self.static_in = [torch.empty(t.shape, dtype=t.dtype, device=dev) for t in inputs]
        self._refresh(inputs)
        self.graph = torch.cuda.CUDAGraph()
        torch.cuda.synchronize(dev)
        # thread-local capture mode: HIP calls of OTHER threads (e.g. the RCCL watchdog of a multi-GPU job polling its events)
        # neither join nor invalidate this capture
        with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
            self.static_out = fn(*self.static_in)
        self.replays = 0

    def _refresh(self, inputs):
        for s, t in zip(self.static_in, inputs):
            s.copy_(t)

    def __call__(self, *inputs):
        self._refresh(inputs)
        self.graph.replay()
        self.replays += 1
        return self.static_out


class StepGraphCache:
    """One live graph per sampler: keyed by the identity of everything the captured step closes over.

    The key holds STRONG references (identity can then never be recycled by the allocator); a miss drops the old
    graph.  `state` per key: 0 = never seen -> run eagerly (warm-up: fills the engine arena, the guider's rule
    cache and every lazily packed weight; its result is the real step result), 1 = warmed -> capture + replay.
    """

    def __init__(self):
        self.key = None
        self.state = 0
        self.graph: StepGraph | None = None
        self.disabled = False
        self.captures = 0

    @staticmethod
    def _same(a, b) -> bool:
        if len(a) != len(b):
            return False
        for u, v in zip(a, b):
            if isinstance(u, (int, float, str, tuple, type(None))) or isinstance(v, (int, float, str, tuple, type(None))):
                if type(u) is not type(v) or u != v:
                    return False
            elif u is not v:
                return False
        return True

    def lookup(self, key):
        if self.key is None or not self._same(self.key, key):
            self.key, self.state, self.graph = key, 0, None
        return self

    def reset(self):
        self.key, self.state, self.graph = None, 0, None
