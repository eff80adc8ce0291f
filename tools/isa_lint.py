#!/usr/bin/env python3
"""ISA lint for libseva_hip.so: no packed-fp32 VALU instruction may be the FIRST READER of a register that a memory-pipeline
return has just written.

Round 3 found two kernels that were not repeatable while another kernel shared the CU: in both, a
`v_pk_*_f32` was the first instruction to read a VGPR written by a VMEM load / `ds_bpermute_b32`, and now and then computed with the
register's previous content in the last 16 lanes.  hipcc (ROCm 7.2) emits the pattern freely, so the rule is checked on the emitted
ISA of EVERY kernel of the production library (tests/test_isa_concurrency_cpu.py):

    for every VGPR written by  global_load_* / buffer_load_* (register destination) / flat_load_* / scratch_load_* /
                               ds_read* / ds_bpermute_b32 / ds_permute_b32 / ds_swizzle_b32 / ds_consume / ds_append
    the first later instruction (program order, until the register is overwritten) that reads it is not `v_pk_*_f32`.

    python tools/isa_lint.py [source.hip ...]        # default: every source of the production library; exit 1 on findings

The scan is linear over the kernel body (branch targets are not followed: a reader reached only through a back edge is seen when
the loop body is scanned from its top, which for a load inside the loop is the same text).
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "stable-virtual-camera_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=fast", "--cuda-device-only", "-S"]

REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")
MEM_RETURN = re.compile(r"^(global_load_|buffer_load_|flat_load_|scratch_load_|ds_read|ds_bpermute_b32|ds_permute_b32|ds_swizzle_b32|"
                        r"ds_consume|ds_append|global_atomic_\w+ .*sc0|buffer_atomic_\w+ .*sc0)")
NO_VGPR_DST = re.compile(r"^(global_store|buffer_store|flat_store|scratch_store|ds_write|ds_add|ds_max|ds_min|s_|v_cmp|v_cmpx|v_readlane|"
                         r"v_readfirstlane|global_atomic|buffer_atomic|ds_gws|buffer_wbl2|buffer_inv|global_load_lds|ds_nop)")
READS_DST = re.compile(r"^(v_fmac|v_mac|v_dot2c|v_dot4c|v_dot8c|v_pk_fmac|v_mfma|v_smfmac|v_movrel|v_permlane|v_writelane|v_cndmask_b32_dpp|"
                       r"v_mov_b32_dpp|v_mov_b32_sdwa|v_bfi|v_cvt_pk_fp8|v_cvt_pk_bf8|v_cvt_scalef32)")


def regs(text):
    out = set()
    for m in REG.finditer(text):
        out.update([int(m.group(1))] if m.group(1) is not None else range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def split_ops(t):
    parts = t.split(None, 1)
    if len(parts) < 2:
        return parts[0], []
    return parts[0], [o.strip() for o in parts[1].split(",")]


def dst_src(t):
    """(VGPRs written, VGPRs read) of one instruction line."""
    mn, ops = split_ops(t)
    if not ops:
        return set(), set()
    if t.startswith("buffer_load") and " lds" in t or mn.startswith("global_load_lds"):
        return set(), regs(" ".join(ops))                      # LDS-DMA: no register destination
    if NO_VGPR_DST.match(mn) and not (mn.startswith(("global_atomic", "buffer_atomic")) and "sc0" in t):
        return set(), regs(" ".join(ops))
    d = regs(ops[0])
    s = regs(" ".join(ops[1:]))
    if READS_DST.match(mn):
        s |= d
    return d, s


def kernels(asm_text):
    bodies, cur = {}, None
    for ln in asm_text.split("\n"):
        m = re.match(r"^(_Z\w+):\s", ln)
        if m:
            cur = m.group(1)
            bodies[cur] = []
        elif cur is not None:
            t = ln.split(";")[0].strip()
            if t and not t.startswith("."):
                bodies[cur].append(t)
            if t.startswith("s_endpgm"):
                cur = None
    return bodies


def lint_body(body, horizon=4000):
    """[(index, load text, reader text)] for every packed-fp32 first reader of a memory-pipeline return."""
    found = []
    info = [dst_src(t) for t in body]
    for i, t in enumerate(body):
        if not MEM_RETURN.match(t):
            continue
        d, _ = info[i]
        if not d:
            continue
        pending = set(d)
        for j in range(i + 1, min(len(body), i + 1 + horizon)):
            dj, sj = info[j]
            hit = pending & sj
            if hit:
                mn = body[j].split(None, 1)[0]
                if mn.startswith("v_pk_") and "f32" in mn:
                    found.append((i, t, body[j]))
                pending -= hit
            pending -= dj                                       # overwritten before any read: nothing to check
            if not pending:
                break
    return found


def file_flags(src):
    """Per-file flags of the Makefile (FLAGS_<stem> := ...): the lint looks at the code the library is built from."""
    mk = open(os.path.join(CSRC, "Makefile")).read()
    m = re.search(r"^FLAGS_%s\s*:=\s*(.+)$" % re.escape(os.path.splitext(os.path.basename(src))[0]), mk, re.M)
    return m.group(1).split() if m else []


def compile_to_asm(src, defines=()):
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.run([HIPCC, *FLAGS, *file_flags(src), *[f"-D{d}" for d in defines], "-I", os.path.dirname(src), src, "-o", out],
                       check=True, stderr=subprocess.DEVNULL)
        return open(out).read()


def production_sources():
    mk = open(os.path.join(CSRC, "Makefile")).read()
    m = re.search(r"^SRCS\s*:=\s*(.+)$", mk, re.M)
    return [os.path.join(CSRC, s) for s in m.group(1).split()]


def lint_source(src):
    out = {}
    for name, body in kernels(compile_to_asm(src)).items():
        f = lint_body(body)
        if f:
            out[name] = f
    return out


def main(argv):
    srcs = argv or production_sources()
    total = 0
    for src in srcs:
        res = lint_source(src)
        n = sum(len(v) for v in res.values())
        total += n
        print(f"{os.path.basename(src)}: {n} packed-fp32 first readers of memory-pipeline returns in {len(res)} kernels", flush=True)
        for name, f in sorted(res.items()):
            print(f"  {name}: {len(f)}")
            for i, ld, rd in f[:4]:
                print(f"      [{i}] {ld}   ->   {rd}")
    return 1 if total else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
