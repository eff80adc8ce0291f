"""pytest configuration: markers + import paths.

`gpu` tests need a real MI355X (run with `-m gpu` on the GPU box); everything else runs on CPU.
"""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "stable-virtual-camera_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLD = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (HIP kernels run)")
    config.addinivalue_line("markers", "slow: multi-GB CPU test (1.3B oracle)")


def load_golden(name: str) -> dict:
    z = np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=False)
    out = {}
    for k in z.files:
        a = z[k]
        out[k] = torch.from_numpy(a) if a.dtype.kind in "fiub" and a.ndim > 0 else a
    return out


@pytest.fixture(scope="session")
def golden():
    return load_golden


def rel_l2(a: torch.Tensor, b: torch.Tensor) -> float:
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


KNOB_NAMES = ("gemm_chunks", "gemm_dbg", "gemm_stagger", "gemm_bm", "gemm_bn", "gemm_astat",
              "attn_dbg", "attn_no_tr", "attn_two", "attn_split", "gn_min_iter", "conv_win")


@pytest.fixture
def knobs():
    """Set libseva_hip.so benchmark knobs for one test (seva_set_knob); every knob is back to 'unset' afterwards."""
    from seva import ops

    def set_(**kw):
        for k, v in kw.items():
            ops.set_knob(k, int(v))

    yield set_
    for k in KNOB_NAMES:
        ops.set_knob(k, -1)
