#!/usr/bin/env python3
"""Two ranks on one card (gloo): one window with CFG-split vs the same window unsplit on each rank; per-step comparison."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stable-virtual-camera_amd")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def worker(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from test_model_gpu import _build
    from seva import sampling as S
    from seva import synthetic as synth
    from seva.model import SGMWrapper
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    grp = dist.new_group([0, 1])
    net, _ = _build("tiny", dev)
    wrap = SGMWrapper(net)
    T, hw, steps = 21, 16, 3
    sc = synth.synth_scene(T, (hw, hw), (0,), seed=7)
    disc = S.DDPMDiscretization()

    def run(split, graph=True):
        den = S.DiscreteDenoiser(disc, num_idx=1000, device=dev)
        sampler = S.EulerEDMSampler(disc, S.MultiviewCFG(1.2), num_steps=steps, verbose=False, device=dev, s_churn=0.0)
        gen = torch.Generator(device=dev)
        gen.manual_seed(1234)
        sampler.noise_fn = lambda x: torch.randn(x.shape, generator=gen, device=x.device, dtype=x.dtype)
        sampler.cfg_split = (grp, rank) if split else None
        sampler._step_graphs.disabled = not graph
        cond = {k: v.to(dev) for k, v in sc["cond"].items()}
        uc = {k: v.to(dev) for k, v in sc["uc"].items()}
        kw = dict(c2w=sc["c2w"].to(dev), K=sc["K"].to(dev), input_frame_mask=sc["input_frame_mask"].to(dev))
        xs = []
        x, s_in, sigmas, num_sigmas, cond, uc = sampler.prepare_sampling_loop(sc["noise"].to(dev).clone(), cond, uc, None)
        for i in range(num_sigmas - 1):
            x = sampler.sampler_step(s_in * sigmas[i], s_in * sigmas[i + 1], lambda a, s, c: den(wrap, a, s, c, num_frames=T), x, 2.0,
                                     cond, uc, 0.0, **kw)
            xs.append(x.clone())
        return xs

    with torch.no_grad():
        a = run(False, True)
        a2 = run(False, True)
        b = run(False, False)
        b2 = run(False, False)
        net.engine().use_graph = False
        e = run(False, False)   # pure eager: no step graph, no network graph
        e2 = run(False, False)
        net.engine().use_graph = True
        c = run(True)
    if rank == 0:
        d = lambda u, v: [f"{float((u[i] - v[i]).abs().max()):.2e}" for i in range(len(u))]
        print("whole-step graph, run 1 vs run 2      ", d(a, a2))
        print("network-only graph, run 1 vs run 2    ", d(b, b2))
        print("pure eager, run 1 vs run 2            ", d(e, e2))
        print("whole-step graph vs pure eager        ", d(a2, e))
        print("network-only graph vs pure eager      ", d(b2, e))
        print("CFG-split (network graph) vs pure eager", d(c, e))
        print("CFG-split vs network-only graph       ", d(c, b2), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(worker, args=(2, port), nprocs=2, join=True)
