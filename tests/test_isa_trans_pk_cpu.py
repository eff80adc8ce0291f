"""Build-time check on the emitted ISA of the GroupNorm kernels: no packed-fp32 VALU instruction (v_pk_fma_f32 / v_pk_mul_f32 /
v_pk_add_f32) may issue while a transcendental result (v_exp_f32, v_rcp_f32, ...) of the same wave is still unconsumed.  With the two
interleaved, the Pluecker-modulated GroupNorm + SiLU apply kernel lost the last 16 lanes of one result register in ~10 % of its launches
while a second process was running on the same MI355X (never with one process; DESIGN.md section 4, profiles/r03_two_process_groupnorm.log).
The kernel now runs the SiLU and the modulation as two phases; a compiler upgrade that re-interleaved them would bring the fault back
silently -- this test fails instead.  tools/isa_trans_pk_lint.py is the same scan for any .s file.  (hipcc cross-compiles, ~15 s.)"""
import importlib.util
import os
import subprocess

import pytest

from conftest import PKG, ROOT

HIPCC = "/opt/rocm/bin/hipcc"


def _lint():
    spec = importlib.util.spec_from_file_location("isa_trans_pk_lint", os.path.join(ROOT, "tools", "isa_trans_pk_lint.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_the_scan_sees_an_interleaved_pair_and_accepts_a_consumed_one(tmp_path):
    lint = _lint()
    dirty = tmp_path / "dirty.s"
    dirty.write_text("k_dirty:\n\tv_rcp_f32_e32 v3, v2\n\tv_pk_fma_f32 v[10:11], v[4:5], v[6:7], v[8:9]\n\tv_mul_f32_e32 v3, v1, v3\n\ts_endpgm\n")
    clean = tmp_path / "clean.s"
    clean.write_text("k_clean:\n\tv_rcp_f32_e32 v3, v2\n\ts_nop 0\n\tv_mul_f32_e32 v3, v1, v3\n\tv_pk_fma_f32 v[10:11], v[4:5], v[6:7], v[8:9]\n\ts_endpgm\n")
    assert list(lint.lint(str(dirty))) == ["k_dirty"]
    assert lint.lint(str(clean)) == {}


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
def test_groupnorm_kernels_keep_packed_math_clear_of_pending_transcendentals(tmp_path):
    src = os.path.join(PKG, "csrc", "norm.hip")
    out = tmp_path / "norm.s"
    subprocess.run([HIPCC, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=fast", "-I", os.path.dirname(src),
                    "--cuda-device-only", "-S", src, "-o", str(out)], check=True, stderr=subprocess.DEVNULL)
    text = out.read_text()
    assert text.count("gn_apply_kernel") >= 5 and "v_pk_fma_f32" in text and "v_exp_f32" in text   # the scan has something to look at
    findings = {k: v for k, v in _lint().lint(str(out)).items() if "gn_apply_kernel" in k}
    assert not findings, {k: (len(v), v[0][:2]) for k, v in findings.items()}
