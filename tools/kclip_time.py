import sys, time, torch
sys.path.insert(0, "stable-virtual-camera_amd")
from seva.modules.conditioner import CLIPConditioner
dev = torch.device("cuda:0")
clip = CLIPConditioner(random_init=True).to(dev)
img = torch.rand(1, 3, 576, 576, device=dev) * 2 - 1
with torch.no_grad():
    for i in range(6):
        torch.cuda.synchronize(); t = time.perf_counter(); clip(img); torch.cuda.synchronize()
        print(f"clip call {i}: {(time.perf_counter() - t) * 1e3:.2f} ms", flush=True)
