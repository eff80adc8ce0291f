#!/usr/bin/env python3
"""VAE decode / encode of 576x576 frames (random-init SD-2.1 topology): ms per frame vs frames per pass."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "stable-virtual-camera_amd"))
import torch
from seva.modules.autoencoder import AutoEncoder
import warnings; warnings.simplefilter("ignore")
dev = torch.device("cuda:0")
ae = AutoEncoder(chunk_size=1, random_init=True).to(dev)
g = torch.Generator().manual_seed(0)
batches = [int(v) for v in os.environ.get("KVAE_BATCHES", "1,3,7,21").split(",")]
zall = (torch.randn(max(batches), 4, 72, 72, generator=g) * 0.18215 * 4).to(dev)
ref = None
for n in batches:
    z = zall[:n]
    eng = ae.engine()
    with torch.no_grad():
        eng.decode(z, ae.scale_factor); torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 3 if n < 8 else 1
        for _ in range(reps):
            out = eng.decode(z, ae.scale_factor)
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    if ref is None:
        ref = out[:1].clone()
    same = torch.equal(out[:1], ref)
    print(f"decode {n:2d} frames per pass: {dt * 1e3:8.2f} ms = {dt / n * 1e3:6.2f} ms/frame; frame 0 bit-identical to the 1-frame pass: {same}; "
          f"arena {eng.arena.nbytes() / 2**30:.2f} GiB", flush=True)
if os.environ.get("KVAE_ENCODE", "1") == "1":
    x = (torch.rand(7, 3, 576, 576, generator=g) * 2 - 1).to(dev)
    for n in (1, 7):
        enc = ae.encoder_engine()
        with torch.no_grad():
            enc.encode(x[:n], ae.scale_factor); torch.cuda.synchronize()
            t0 = time.perf_counter(); enc.encode(x[:n], ae.scale_factor); torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"encode {n:2d} frames per pass: {dt * 1e3:8.2f} ms = {dt / n * 1e3:6.2f} ms/frame", flush=True)
