#!/bin/bash
# poll the shader clock while a long attention / GEMM loop runs
cd $GRAFT_REPO_ROOT 2>/dev/null || cd /root/repo
(for i in $(seq 1 30); do rocm-smi --showclocks 2>/dev/null | grep -i "sclk" | head -1; rocm-smi --showpower 2>/dev/null | grep -i "power" | head -1; sleep 0.3; done) > gpurun_out/clk_idle_then_load.log 2>&1 &
sleep 1.5
KATTN_SHAPE=2,10,27216 python - <<'PY' > gpurun_out/clk_run.log 2>&1
import os, sys, time
sys.path.insert(0, "stable-virtual-camera_amd")
import torch
from seva import ops
dev = torch.device("cuda:0")
B, H, L = 2, 10, 27216
C = 64 * H
qkv = torch.randn(B * L, 3 * C).half().to(dev)
out = torch.empty(B * L, C, device=dev, dtype=torch.float16)
c3 = 3 * C
t0 = time.time()
n = 0
while time.time() - t0 < 4.0:
    for _ in range(20):
        ops.attention(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], out, nb0=B, nb1=1, heads=H, lq=L, lk=L,
                      q_strides=(L * c3, 0, c3), k_strides=(L * c3, 0, c3), o_strides=(L * C, 0, C), q_prescaled=True)
    torch.cuda.synchronize(); n += 20
print("attention calls", n, "in", time.time() - t0)
a = torch.randn(13608, 5120).half().to(dev); w = torch.randn(1280, 5120).half().to(dev); o = torch.empty(13608, 1280, device=dev)
t0 = time.time()
while time.time() - t0 < 4.0:
    for _ in range(200):
        ops.gemm(a, w, out_f32=o)
    torch.cuda.synchronize()
print("gemm done")
PY
wait
cat gpurun_out/clk_idle_then_load.log | tr '\n' ' ' | sed 's/GPU\[0\]//g' | head -c 3000
