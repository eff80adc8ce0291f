"""Build-time checks on the emitted ISA of the two kernels that were not repeatable while OTHER work ran on the same MI355X (a second
stream or a second process; alone they always were): the Pluecker-modulated GroupNorm apply and the fused feed-forward's LayerNorm
prologue.  In both, a packed-fp32 instruction was the first reader of a register a memory-pipeline return had just written -- `v_pk_fma_f32
... op_sel` on the registers the modulation's loads returned into, `v_pk_add_f32` on the results of `ds_bpermute_b32` -- and now and then
computed with the register's previous content in the last 16 lanes of the wave (up to 25 % of the launches under load: DESIGN.md section 4,
profiles/r03_concurrency_*.log).  With a plain 32-bit first reader (two `v_mov_b32` building a real register pair; `v_add_f32`) both are
clean over thousands of launches.  A compiler upgrade that folded the broadcast back into op_sel, or re-packed the shuffle adds, would bring
the fault back silently; this test fails instead.  tests/test_model_gpu.py holds the run-time check.  (hipcc cross-compiles, ~40 s.)"""
import os
import re
import subprocess

import pytest

from conftest import PKG

HIPCC = "/opt/rocm/bin/hipcc"
REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def _regs(text):
    out = set()
    for m in REG.finditer(text):
        out.update([int(m.group(1))] if m.group(1) is not None else range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def _asm(tmp_path, name):
    src = os.path.join(PKG, "csrc", name + ".hip")
    out = tmp_path / (name + ".s")
    subprocess.run([HIPCC, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=fast", "-I", os.path.dirname(src),
                    "--cuda-device-only", "-S", src, "-o", str(out)], check=True, stderr=subprocess.DEVNULL)
    bodies, cur = {}, None
    for ln in out.read_text().split("\n"):
        m = re.match(r"^(_Z\w+):\s", ln)
        if m:
            cur = m.group(1); bodies[cur] = []
        elif cur is not None:
            t = ln.split(";")[0].strip()
            if t: bodies[cur].append(t)
            if t.startswith("s_endpgm"): cur = None
    return bodies


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
def test_groupnorm_modulation_reads_real_register_pairs(tmp_path):
    bodies = {k: v for k, v in _asm(tmp_path, "norm").items() if "gn_apply_kernelILb1E" in k}   # DENSE = true
    assert len(bodies) >= 3, list(bodies)
    for name, body in bodies.items():
        pk = [t for t in body if t.startswith("v_pk_fma_f32")]
        assert len(pk) >= 72, (name, len(pk))                    # the modulation is still packed: 4 channels x 6 components x 3 pixels
        assert not [t for t in pk if "op_sel" in t], (name, [t for t in pk if "op_sel" in t][:3])


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
def test_fused_ff_layernorm_prologue_adds_shuffle_results_with_plain_ops(tmp_path):
    bodies = {k: v for k, v in _asm(tmp_path, "ff_fused").items() if "ff_fused8_kernel" in k}
    assert len(bodies) == 4, list(bodies)
    for name, body in bodies.items():
        perm = [i for i, t in enumerate(body) if t.startswith("ds_bpermute_b32")]
        assert len(perm) >= 8, (name, len(perm))                 # two statistics x two shuffles x two row groups
        for i in perm:
            dst = _regs(body[i].split(None, 1)[1].split(",")[0])
            reader = next((t for t in body[i + 1:i + 60] if t.startswith("v_") and "," in t and _regs(t.split(None, 1)[1].split(",", 1)[1]) & dst), None)
            assert reader is not None and not reader.startswith("v_pk_"), (name, body[i], reader)
