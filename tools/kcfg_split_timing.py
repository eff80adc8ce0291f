#!/usr/bin/env python3
"""Where does a CFG-split window spend its time?  Two ranks (gloo; both on card 0 unless two cards are visible), the 1.3B network at the
headline shape, one warm-up window and one timed window of 4 steps in CFG-split mode, then the same window unsplit on rank 0."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stable-virtual-camera_amd")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def worker(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from test_model_gpu import _build
    from seva import pipeline, sampling as S
    from seva import synthetic as synth
    from seva.model import SGMWrapper
    dev = torch.device("cuda", rank if torch.cuda.device_count() >= world else 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    net, _ = _build(os.environ.get("MODEL", "full"), dev)
    wrap = SGMWrapper(net)
    n, T, hw = 168, 21, int(os.environ.get("HW", "72"))
    c2ws, Ks = synth.orbit_c2w(n), synth.default_K(n)
    g = torch.Generator().manual_seed(23)
    lat = (torch.randn(1, 4, hw, hw, generator=g) * 0.18215 * 5.0).to(dev)
    tok = torch.randn(1024, generator=g); tok = (tok / tok.norm()).to(dev)
    plan = pipeline.plan_trajectory(c2ws, [0], T=T)
    pair = pipeline.cfg_pair_groups()[0]
    acc = {"gather": 0.0, "gathers": 0, "denoise": 0.0}
    real_gather = dist.all_gather_into_tensor

    def timed_gather(out, inp, group=None):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        real_gather(out, inp, group=group)
        torch.cuda.synchronize(); acc["gather"] += time.perf_counter() - t0; acc["gathers"] += 1
    dist.all_gather_into_tensor = timed_gather
    real_split = S.EulerEDMSampler._denoise_cfg_split

    def timed_split(self, *a):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = real_split(self, *a)
        torch.cuda.synchronize(); acc["denoise"] += time.perf_counter() - t0
        return r
    S.EulerEDMSampler._denoise_cfg_split = timed_split

    def window(steps, split, win=None):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        pipeline.run_window(win or plan.pass1[0], {f: lat[0] for f in range(n)}, wrap, c2ws, Ks, hw=(hw, hw), num_steps=steps, cfg=2.0, cfg_min=1.2,
                            guider=1, camera_scale=2.0, noise=torch.randn(T, 4, hw, hw), step_seed=1, clip_token=tok, device=dev,
                            cfg_split=(pair, rank) if split else None)
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    with torch.no_grad():
        if os.environ.get("WARM_PASS2", "0") == "1":
            window(2, False, plan.pass2[0])
            w0 = window(2, True, plan.pass2[0])
        else:
            w0 = window(2, True)
        dist.barrier()
        for k in acc: acc[k] = 0
        if os.environ.get("PROFILE", "0") == "1" and rank == 0:
            import cProfile, pstats, io
            pr = cProfile.Profile(); pr.enable()
            w1 = window(4, True)
            pr.disable()
            st = io.StringIO(); pstats.Stats(pr, stream=st).sort_stats("cumulative").print_stats(28); print(st.getvalue()[:6000], flush=True)
        else:
            w1 = window(4, True)
        dist.barrier()
        split_acc = dict(acc)
        if rank == 0:
            u0 = window(2, False); u1 = window(4, False)
    if rank == 0:
        print(f"CFG-split window: warm-up (2 steps) {w0:.2f} s; timed (4 steps) {w1:.3f} s = {w1 / 4 * 1e3:.1f} ms per step; inside: the half-batch denoise + gather "
              f"{split_acc['denoise'] / 4 * 1e3:.1f} ms per step, of which the all-gather {split_acc['gather'] / max(split_acc['gathers'], 1) * 1e3:.1f} ms ({split_acc['gathers']} gathers)")
        print(f"unsplit window on rank 0 alone: warm-up {u0:.2f} s; timed (4 steps) {u1:.3f} s = {u1 / 4 * 1e3:.1f} ms per step", flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(worker, args=(2, port), nprocs=2, join=True)
