"""Build-time check of the workgroup-to-workgroup hand-off protocol of the split-K kernels (csrc/gemm_common.h), on the
emitted ISA: correctness rests on (1) EVERY payload store and load of the handed-off accumulators being an agent-scope
(`sc1`) access -- they go past the XCD-private L2 without a cache-wide fence -- and (2) the flag store coming after the
storing waves' `s_waitcnt vmcnt(0)` and the workgroup barrier.  A compiler upgrade that re-vectorised the relaxed atomics into
plain dwordx4 accesses, or moved the flag store, would break the protocol silently; this test would fail instead.
(hipcc cross-compiles for gfx950 without a GPU; ~20 s.)"""
import os
import re
import shutil
import subprocess

import pytest

from conftest import PKG

HIPCC = "/opt/rocm/bin/hipcc"
KERNEL = re.compile(r"^(_ZN\S*gemm_kernelILi128ELi(?:128|160)ELi1ELi0ELb0ELb0ELb0ELb0ELb1ELi4E(?:Lb0E)?E\S*):\s")  # MODE 1, SPLITK = true


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
def test_splitk_handoff_uses_agent_scope_accesses_and_orders_the_flag(tmp_path):
    src = os.path.join(PKG, "csrc", "gemm.hip")
    out = tmp_path / "gemm.s"
    subprocess.run([HIPCC, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=fast", "-I", os.path.dirname(src),
                    "--cuda-device-only", "-S", src, "-o", str(out)], check=True, stderr=subprocess.DEVNULL)
    lines = out.read_text().split("\n")
    bodies, cur = {}, None
    for ln in lines:
        m = KERNEL.match(ln)
        if m:
            cur = m.group(1)
            bodies[cur] = []
        elif cur is not None:
            bodies[cur].append(ln.strip())
            if ln.strip().startswith("s_endpgm"):
                cur = None
    assert len(bodies) == 2, list(bodies)  # the 128 x 128 and 128 x 160 split-K conv kernels
    for name, body in bodies.items():
        bn = 160 if "Li160E" in name else 128
        frags = (128 // 2 // 16) * (bn // 2 // 16)  # MI x NJ accumulator fragments per wave, 4 dwords each
        st = [i for i, t in enumerate(body) if t.startswith("global_store_dword ") and t.endswith("sc1")]
        ld = [i for i, t in enumerate(body) if t.startswith("global_load_dword ") and t.endswith("sc1")]
        assert len(st) >= 4 * frags + 1 and len(ld) >= 4 * frags + 1, (name, len(st), len(ld))
        # payload accesses address through a 64-bit VGPR pair ("v[a:b], vN, off" / "vN, v[a:b], off"), the flag through the
        # scalar base + VGPR offset form ("..., s[a:b]")
        pay_st = [i for i in st if ", off" in body[i]]
        flag_st = [i for i in st if re.search(r", s\[\d+:\d+\]", body[i])]
        pay_ld = [i for i in ld if ", off" in body[i]]
        poll_ld = [i for i in ld if re.search(r", s\[\d+:\d+\]", body[i])]
        assert len(pay_st) >= 4 * frags and len(pay_ld) >= 4 * frags and flag_st and poll_ld, (name, len(pay_st), len(pay_ld))
        # producer: after the LAST payload store comes s_waitcnt vmcnt(0), then a barrier, then the flag store
        flag = next((i for i in flag_st if i > pay_st[-1]), None)
        assert flag is not None, name
        between = body[pay_st[-1] + 1:flag]
        w = next((i for i, t in enumerate(between) if t.startswith("s_waitcnt") and "vmcnt(0)" in t), None)
        b = next((i for i, t in enumerate(between) if t == "s_barrier"), None)
        assert w is not None and b is not None and w < b, (name, between[:60])
        # consumer: a barrier separates the flag poll from the first payload load
        polls = [i for i in poll_ld if i < pay_ld[0]]
        assert polls and any(t == "s_barrier" for t in body[polls[-1]:pay_ld[0]]), name
        # nothing wider than a dword carries sc1 (a vectorised payload access would not be the measured protocol)
        assert not any(("dwordx" in t and t.endswith("sc1")) for t in body), name
