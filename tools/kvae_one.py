#!/usr/bin/env python3
"""One warm VAE decode of 7 frames at 576x576 (random-init weights) -- the driver for rocprofv3 runs over the VAE path."""
import os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stable-virtual-camera_amd"))
import torch
warnings.filterwarnings("ignore")
from seva.modules.autoencoder import AutoEncoder
dev = torch.device("cuda:0")
ae = AutoEncoder(chunk_size=1, random_init=True).to(dev)
z = (torch.randn(7, 4, 72, 72, generator=torch.Generator().manual_seed(1)) * 0.18215).to(dev)
with torch.no_grad():
    ae.decode(z); torch.cuda.synchronize()
    for _ in range(3): ae.decode(z)
    torch.cuda.synchronize()
print("done")
