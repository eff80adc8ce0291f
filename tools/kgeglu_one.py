import os, sys, torch
sys.path.insert(0, "stable-virtual-camera_amd")
from seva import ops
dev = torch.device("cuda:0")
for M, N, K in [(13608, 10240, 1280), (54432, 5120, 640)]:
    a = torch.randn(M, K, device=dev).half()
    w = (torch.randn(N, K, device=dev) * 0.05).half()
    b = torch.randn(N, device=dev)
    o = torch.empty(M, N // 2, device=dev, dtype=torch.float16)
    for _ in range(3):
        ops.gemm(a, w, bias=b, out_f16=o, geglu=True)
torch.cuda.synchronize()
print("done")
