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
import sys

import pytest

from conftest import PKG, ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_lint  # noqa: E402  (tools/isa_lint.py: the rule as a scanner over emitted ISA)

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


# ---- the rule, library-wide (round 4) ------------------------------------------------------------------------------------------
# No `v_pk_*_f32` instruction is the first reader of a VGPR written by a memory-pipeline return (VMEM load incl. scratch reloads,
# `ds_read*`, `ds_bpermute_b32` / `ds_permute_b32` / `ds_swizzle_b32`) in ANY kernel of libseva_hip.so.  hipcc forms packed fp32 math
# wherever it can, so the sources route loaded values through `first_read()` (seva_common.h: one in-place `v_mov_b32`), elementwise.hip is
# built without the vectorisers, and this test compiles every production source with the Makefile's flags and scans every kernel.
@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
@pytest.mark.parametrize("src", [os.path.basename(s) for s in isa_lint.production_sources()])
def test_no_packed_fp32_first_reader_of_a_memory_return_in_any_kernel(src):
    path = os.path.join(PKG, "csrc", src)
    bodies = isa_lint.kernels(isa_lint.compile_to_asm(path))
    bad = {name: isa_lint.lint_body(body) for name, body in bodies.items()}
    bad = {k: v for k, v in bad.items() if v}
    assert not bad, {k: [(ld, rd) for _, ld, rd in v[:3]] for k, v in bad.items()}
    if src not in ("capi.hip",):
        assert len(bodies) >= 1, src  # the scanner did see kernels


def test_lint_scanner_on_synthetic_streams():
    """The scanner itself: a packed first reader is found; a plain first reader, an overwritten register, a store's operands and an
    LDS-DMA (no register destination) are not."""
    L = isa_lint.lint_body
    assert L(["global_load_dwordx4 v[4:7], v[0:1], off", "s_waitcnt vmcnt(0)", "v_pk_fma_f32 v[8:9], v[4:5], v[10:11], v[12:13]"])
    assert L(["ds_bpermute_b32 v3, v1, v2", "s_waitcnt lgkmcnt(0)", "v_pk_add_f32 v[2:3], v[2:3], v[6:7]"])
    assert L(["scratch_load_dwordx2 v[20:21], off, off offset:16", "v_pk_mul_f32 v[0:1], v[20:21], v[2:3]"])
    assert not L(["global_load_dwordx4 v[4:7], v[0:1], off", "v_mov_b32 v4, v4", "v_mov_b32 v5, v5", "v_mov_b32 v6, v6", "v_mov_b32 v7, v7",
                  "v_pk_fma_f32 v[8:9], v[4:5], v[10:11], v[12:13]"])
    assert not L(["ds_read_b128 v[4:7], v1", "v_mfma_f32_16x16x32_f16 v[8:11], v[4:7], v[12:15], v[8:11]", "v_pk_add_f32 v[8:9], v[8:9], v[20:21]"])
    assert not L(["global_load_dword v4, v[0:1], off", "v_mov_b32 v4, 0", "v_pk_add_f32 v[4:5], v[4:5], v[6:7]"])
    assert not L(["global_load_lds_dwordx4 v[0:1], off", "v_pk_add_f32 v[0:1], v[0:1], v[2:3]"])
    assert not L(["global_load_dword v4, v[0:1], off", "global_store_dword v[2:3], v4, off", "v_pk_add_f32 v[4:5], v[4:5], v[6:7]"])


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
def test_lint_is_red_on_the_sources_before_the_two_fixes(tmp_path):
    """Round 3's two non-repeatable kernels, as they were BEFORE commit 031eb80 (`git show 251746c:...`): the scanner flags exactly the
    instructions that were at fault -- `v_pk_fma_f32` reading the modulation loads' registers in the modulated GroupNorm apply, and
    `v_pk_add_f32` reading `ds_bpermute_b32` results in the fused feed-forward's LayerNorm prologue."""
    rel = "stable-virtual-camera_amd/csrc/"
    old = {}
    for name, rev in (("norm.hip", "251746c"), ("ff_fused.hip", "12312b2"), ("seva_common.h", "251746c"), ("gemm_common.h", "251746c")):
        r = subprocess.run(["git", "-C", ROOT, "show", f"{rev}:{rel}{name}"], capture_output=True, text=True)
        if r.returncode != 0:
            pytest.skip("no git history here")
        old[name] = r.stdout
    d = tmp_path / "csrc"
    d.mkdir()
    (tmp_path / "include").mkdir()
    hdr = subprocess.run(["git", "-C", ROOT, "show", "251746c:include/seva_hip.h"], capture_output=True, text=True).stdout
    (tmp_path / "include" / "seva_hip.h").write_text(hdr)
    for name, text in old.items():
        (d / name).write_text(text.replace('"../../include/seva_hip.h"', '"../include/seva_hip.h"'))
    gn = isa_lint.kernels(isa_lint.compile_to_asm(str(d / "norm.hip")))
    hits = [h for k, b in gn.items() if "gn_apply_kernelILb1E" in k for h in isa_lint.lint_body(b)]
    assert any(rd.startswith("v_pk_fma_f32") and ld.startswith("global_load") for _, ld, rd in hits), hits[:4]
    ff = isa_lint.kernels(isa_lint.compile_to_asm(str(d / "ff_fused.hip")))
    hits = [h for k, b in ff.items() if "ff_fused8_kernelILi64E" in k for h in isa_lint.lint_body(b)]
    assert any(rd.startswith("v_pk_add_f32") and ld.startswith("ds_bpermute_b32") for _, ld, rd in hits), hits[:4]
