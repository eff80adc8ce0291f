#!/usr/bin/env python3
"""bench.py -- denoising steps/sec of the Seva 1.3B hot path on MI355X (driver contract).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one Euler-EDM sampler step over one 21-view window at 576x576 (latent 72x72): noise
perturbation, CFG-batched network call on 2T = 42 frames, guidance combine, Euler update
(reference seva/sampling.py:347-368).  Inputs are synthetic (seva/synthetic.py) and resident in
HBM before the timed region; weights are random-init of the 1.3B architecture.

N > 1: the windows of a long trajectory are independent work units; every rank
denoises its own window with replicated weights ("weak" scaling), after one RCCL all-gather of the
anchor latents that adjacent windows share.  value = (steps of all ranks) / max-over-ranks time.

Extra objects on the JSON line: `roofline` (dominant kernel class, algorithmic FLOP / HIP-event
time measured live in one instrumented step) and `cpu_baseline` (the CPU oracle timed on the
host cores on BASELINE config 1, rank 0, N=1 only).

Outside `value`, so that the driver-run record also carries BASELINE.json's other configurations:
`other_configs` (N = 1: config 2 = T=8, config 3 = T=24 as one window, config 5 = fp8 at T=21, config 4 = the 168-view
two-pass trajectory on this one GPU with a few sampler steps per window) and, for N > 1, `trajectory`: the SAME 168-view
trajectory run across the N ranks with CFG-split (STRONG scaling: total work fixed; seva/pipeline.py) -- the number that
says something about config 4, which one-window-per-rank weak scaling does not.
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "stable-virtual-camera_amd"))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_FP16_MFMA_TFLOPS = 2500.0  # dense, MI355X_MICROARCH.md "Peak BF16/FP16 MFMA"
# reference-equivalent dense FLOP per denoising step (FlopCounterMode on the reference)
N_INPUT_VIEWS = 1          # synthetic scene: frame 0 is the input view
NUM_STEPS_PER_WINDOW = 50  # reference default (demo.py:292-306 num_steps)
FLOP_PER_STEP = {(21, 72): 7.691e13, (8, 72): 2.634e13, (24, 72): 8.999e13, (4, 32): 2.217e12}


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def build_model(device, shapes_only=False):
    from seva import synthetic as synth
    from seva.model import Seva, SevaParams

    with torch.device("meta"):
        net = Seva(SevaParams())
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    t0 = time.time()
    sd = synth.synth_state_dict(shapes, 0)
    log(f"[bench] synthetic 1.3B weights generated on CPU in {time.time() - t0:.1f}s")
    net.load_state_dict(sd, strict=True, assign=True)
    return net, sd


def make_sampler(net, device, T, hw, steps, seed):
    from seva import sampling as S
    from seva import synthetic as synth
    from seva.model import SGMWrapper

    sc = synth.synth_scene(T, (hw, hw), (0,), seed=seed)
    # step-invariant inputs assembled on the GPU: camera normalisation on the host, Pluecker maps
    # and the cond / uc channel assembly by HIP kernels; replaces the CPU-built dictionaries of synth_scene
    from seva import conditioning as Cn
    Cn.get_value_dict((hw * 8, hw * 8), [0], sc["c2w"][:, :3], sc["K"], sc["c2w"], 2.0, device=device)  # warm-up
    torch.cuda.synchronize()
    times = []
    for _ in range(3):  # median of three: the host part (camera normalisation, V + 1 tiny inverses) is noisy on a fresh box
        t0 = time.perf_counter()
        vd = Cn.get_value_dict((hw * 8, hw * 8), [0], sc["c2w"][:, :3], sc["K"], sc["c2w"], 2.0, device=device)
        lat = sc["cond"]["replace"][sc["input_frame_mask"], :4]
        cond_d, uc_d = Cn.assemble_cond(lat, sc["cond"]["crossattn"][0, 0], sc["input_frame_mask"], vd["plucker_coordinate"])
        torch.cuda.synchronize()
        times.append((time.perf_counter() - t0) * 1e3)
    make_sampler.cond_assembly_ms = sorted(times)[1]
    sc["cond"], sc["uc"], sc["c2w"] = cond_d, uc_d, vd["c2w"]
    disc = S.DDPMDiscretization()
    den = S.DiscreteDenoiser(disc, num_idx=1000, device=device)
    sampler = S.EulerEDMSampler(disc, S.MultiviewCFG(1.2), num_steps=steps, verbose=False,
                                device=device, s_churn=0.0)
    wrap = SGMWrapper(net)
    cond = {k: v.to(device) for k, v in sc["cond"].items()}
    uc = {k: v.to(device) for k, v in sc["uc"].items()}
    gk = dict(c2w=sc["c2w"].to(device), K=sc["K"].to(device),
              input_frame_mask=sc["input_frame_mask"].to(device))
    denoise = lambda x, s, c: den(wrap, x, s, c, num_frames=T)  # noqa: E731
    return sampler, denoise, sc["noise"].to(device), cond, uc, gk


def cpu_baseline(sd):
    """CPU oracle (restatement pinned to the reference, oracle/) on BASELINE config 1:
    T=4, 32x32 latent, CFG batch 8 -- bounded sample: 4 sampler steps (about 15 s on 16 host threads)."""
    from oracle import sampling_ref as SR
    from oracle import seva_ref as OR
    from seva import synthetic as synth

    T, hw, steps = 4, 32, 4
    sc = synth.synth_scene(T, (hw, hw), (0,), seed=23)
    net = lambda x, idx, c, num_frames: OR.sgm_wrapper_forward(sd, x, idx, c, num_frames)  # noqa: E731
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, 16))  # the GPU box grants a 16-core share per GPU
    torch.set_num_threads(cores)
    t0 = time.time()
    with torch.no_grad():
        SR.euler_edm_sample(net, sc["noise"], sc["cond"], sc["uc"], steps, 2.0, None, guider=1,
                            cfg_min=1.2, c2w=sc["c2w"], K=sc["K"],
                            input_frame_mask=sc["input_frame_mask"])
    dt = time.time() - t0
    return {
        "value": steps / dt, "unit": "denoising steps/s", "cores": cores, "kind": "port",
        "sample": f"{steps} Euler steps of BASELINE config 1 (T=4, 32x32 latent, CFG batch 8, fp32), "
                  f"{dt:.1f}s; {FLOP_PER_STEP[(4, 32)] * steps / dt / 1e12:.2f} TFLOP/s",
    }


def trajectory_mode(args, net, device, rank, world):
    """BASELINE config 4: an N-view orbit, 1 input view, two-pass `interp` plan (1 + 10 windows of 21 views at N=168),
    50 steps per window.  Reports BOTH definitions SURVEY §8e names: the shardable second-pass throughput and the
    whole-trajectory wall time including the first pass, the anchor all-gather and the gather to rank 0."""
    leg = trajectory_leg(net, device, rank, world, args.trajectory, args.latent, args.views, args.traj_steps,
                         not args.no_cfg_split)
    if rank == 0:
        print(json.dumps({
            "metric": f"novel views/sec, 1.3B Seva, {leg['workload']}",
            "value": leg["novel_views_per_sec"], "unit": "novel views/s (whole trajectory, first pass included)",
            "higher_is_better": True, "dtype": "f16", "data": "synthetic", **leg}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def steps_leg(net, device, T, hw, steps, seed=23):
    """steps/s of one window shape with the same procedure as the headline line (2 set-up steps incl. graph capture,
    1 warm-up, `steps` timed), for the `other_configs` object."""
    sampler, denoise, noise, cond, uc, gk = make_sampler(net, device, T, hw, steps + 3, seed)
    x, s_in, sigmas, _, cond, uc = sampler.prepare_sampling_loop(noise, cond, uc, None)
    with torch.no_grad():
        for i in range(3):
            x = sampler.sampler_step(s_in * sigmas[i], s_in * sigmas[i + 1], denoise, x, 2.0, cond, uc, 0.0, **gk)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(3, 3 + steps):
            x = sampler.sampler_step(s_in * sigmas[i], s_in * sigmas[i + 1], denoise, x, 2.0, cond, uc, 0.0, **gk)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    assert torch.isfinite(x).all()
    flop = FLOP_PER_STEP.get((T, hw))
    return {"views": T, "latent": hw, "steps": steps, "steps_per_sec": steps / dt, "ms_per_step": dt / steps * 1e3,
            "model_tflops": flop * steps / dt / 1e12 if flop else None,
            "hipgraph_whole_step": sampler._step_graphs.captures > 0}


def trajectory_leg(net, device, rank, world, n, hw, T, steps, cfg_split):
    """The n-view orbit (BASELINE config 4: n = 168 -> 20 anchors, 1 + 10 windows of 21 views) through
    seva.pipeline.run_trajectory on `world` ranks, `steps` sampler steps per window.  Max-over-ranks phase times."""
    from seva import pipeline
    from seva import synthetic as synth
    from seva.model import SGMWrapper

    c2ws, Ks = synth.orbit_c2w(n), synth.default_K(n)
    g = torch.Generator().manual_seed(23)
    lat = (torch.randn(1, 4, hw, hw, generator=g) * 0.18215 * 5.0).to(device)
    tok = torch.randn(1024, generator=g)
    tok = (tok / tok.norm()).to(device)
    wrap = SGMWrapper(net)
    plan = pipeline.plan_trajectory(c2ws, [0], T=T)
    timers: dict = {}
    with torch.no_grad():
        # untimed warm-up: one short window fills the engine arena on each rank (and, under CFG-split, the half-batch arena)
        pipeline.run_window(plan.pass2[0], {f: lat[0] for f in range(n)}, wrap, c2ws, Ks, hw=(hw, hw), num_steps=2, cfg=2.0,
                            cfg_min=1.2, guider=1, camera_scale=2.0, noise=torch.randn(T, 4, hw, hw), step_seed=1,
                            clip_token=tok, device=device)
        if world > 1 and cfg_split:
            pair = pipeline.cfg_pair_groups()[rank // 2] if rank // 2 < world // 2 else None
            if pair is not None:
                pipeline.run_window(plan.pass2[0], {f: lat[0] for f in range(n)}, wrap, c2ws, Ks, hw=(hw, hw), num_steps=2,
                                    cfg=2.0, cfg_min=1.2, guider=1, camera_scale=2.0, noise=torch.randn(T, 4, hw, hw),
                                    step_seed=1, clip_token=tok, device=device, cfg_split=(pair, rank % 2))
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        prof = None
        if os.environ.get("SEVA_BENCH_PROFILE_TRAJ") == "1":  # debugging aid: where the host time of the trajectory goes
            import cProfile
            prof = cProfile.Profile()
            prof.enable()
        res = pipeline.run_trajectory(wrap, lat, c2ws, Ks, [0], clip_token=tok, T=T, num_steps=steps, device=device,
                                      plan=plan, timers=timers, cfg_split=cfg_split and world > 1)
        if prof is not None:
            import io
            import pstats
            prof.disable()
            st = io.StringIO()
            pstats.Stats(prof, stream=st).sort_stats("cumulative").print_stats(30)
            print(f"[rank {rank}]", st.getvalue()[:5000], file=sys.stderr, flush=True)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
    t = torch.tensor([timers["pass1"] - timers["start"], timers["exchange"] - timers["pass1"],
                      timers["pass2"] - timers["exchange"], timers["gather"] - timers["pass2"],
                      timers["gather"] - timers["start"]], device=device, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    p1, ex, p2, ga, total = (float(v) for v in t)
    if rank == 0:
        assert torch.isfinite(res["latents"]).all()
    sched = pipeline.second_pass_schedule(len(plan.pass2), world, cfg_split and world > 1)
    return {
        "workload": f"{n}-view orbit, 1 input view, reference two-pass plan: {len(plan.anchor_ids)} anchors, {len(plan.pass1)}+"
                    f"{len(plan.pass2)} windows of {T} views @ {hw * 8}x{hw * 8}, {steps} sampler steps per window "
                    f"(BASELINE config 4 runs 50), anchors regenerated in pass 2 like the reference",
        "n_gpus": world, "scaling": "strong", "cfg_split": bool(cfg_split and world > 1), "steps_per_window": steps,
        "wall_s": total, "novel_views_per_sec": (n - 1) / total, "pass1_s": p1, "anchor_allgather_s": ex, "pass2_s": p2,
        "gather_s": ga, "pass2_steps_per_sec": len(plan.pass2) * steps / p2, "pass2_rounds": [len(r) for r in sched],
        "rank0_window_s": [[ps, i, round(sec, 4)] for ps, i, sec in timers.get("windows", [])],
        "window_times_ceiling": (0.5 if (cfg_split and world > 1) else 1.0) * len(plan.pass1)
                                + sum(0.5 if len(r[0][1]) == 2 else 1.0 for r in sched),
        "handoff": "anchor latents (no decode/encode round trip); VAE decode of the frames not included",
        # which definition a speed-up over one GPU refers to: WHOLE trajectory wall time, serial first pass included.
        # One GPU = (windows of pass 1 + pass 2) window-times; N GPUs = window_times_ceiling.  For the 168-view plan (1 + 10 windows) on
        # 8 GPUs that is 11 / 2.0 = 5.5x with CFG-split (3.7x without): north_star's >= 6x is NOT reachable by sharding whole windows and
        # the two CFG halves -- it needs intra-window sharding (heads / query blocks of the joint attention over more than two ranks)
        "speedup_definition": "whole-trajectory wall time vs one GPU (pass 1 + pass 2); ideal = (n_pass1 + n_pass2) / window_times_ceiling",
        "ideal_speedup_vs_one_gpu": (len(plan.pass1) + len(plan.pass2)) / ((0.5 if (cfg_split and world > 1) else 1.0) * len(plan.pass1)
                                                                       + sum(0.5 if len(r[0][1]) == 2 else 1.0 for r in sched)),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--views", type=int, default=21, help="frames per window (T)")
    ap.add_argument("--latent", type=int, default=72, help="latent side (576/8)")
    ap.add_argument("--trajectory", type=int, default=0,
                    help="BASELINE config 4 mode: generate an N-view orbit (1 input view) with the two-pass pipeline "
                         "(seva/pipeline.py), windows sharded over the ranks; prints its own JSON line")
    ap.add_argument("--traj-steps", type=int, default=50, help="sampler steps per window in --trajectory mode")
    ap.add_argument("--no-cfg-split", action="store_true",
                    help="--trajectory with N > 1: whole windows per rank only (no CFG-split of the first pass / leftover round)")
    ap.add_argument("--precision", choices=["f16", "fp8"], default="f16",
                    help="f16 = parity mode (default, the headline line); fp8 = BASELINE config 5: e4m3 weights + activations "
                         "on the block-scaled fp8 MFMA for the C >= 640 levels (separate accuracy class)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-vae", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the `other_configs` legs (N = 1) / the strong-scaling `trajectory` leg (N > 1)")
    ap.add_argument("--other-steps", type=int, default=5, help="timed steps of each `other_configs` steps/s leg")
    ap.add_argument("--leg-traj-steps", type=int, default=4,
                    help="sampler steps per window of the 168-view trajectory legs on the default line")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an AMD GPU (the HIP path has no CPU fallback)")
    # one rank per GPU.  Rehearsal on a one-GPU box only: SEVA_BENCH_DEVICE pins every rank to one card and
    # SEVA_BENCH_BACKEND=gloo replaces RCCL (which refuses two ranks on one device); the numbers of such a run mean nothing.
    dev_index = int(os.environ.get("SEVA_BENCH_DEVICE", local_rank))
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("SEVA_BENCH_BACKEND", "nccl")  # "nccl" is RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)  # bound to this rank's card before the first collective
        else:
            dist.init_process_group(backend)

    from seva import _native, ops
    from seva.distributed import exchange_anchor_latents

    _native.load()
    T, hw, K, Wm = args.views, args.latent, args.steps, args.warmup
    net, sd = build_model(device)
    net = net.to(device).eval()
    net.set_precision(args.precision)
    if args.trajectory:
        return trajectory_mode(args, net, device, rank, world)
    total_steps = K + Wm
    sampler, denoise, noise, cond, uc, gk = make_sampler(net, device, T, hw, max(total_steps, 2), 23 + rank)
    cond_assembly_ms = getattr(make_sampler, "cond_assembly_ms", None)  # (of the headline window; later legs overwrite the attribute)

    # anchors shared by adjacent windows (first-pass output; here: this rank's input-view latent)
    anchors = cond["replace"][:1, :4].contiguous()
    x, s_in, sigmas, num_sigmas, cond, uc = sampler.prepare_sampling_loop(noise, cond, uc, None)

    def step(i, xx):
        return sampler.sampler_step(s_in * sigmas[i], s_in * sigmas[i + 1], denoise, xx, 2.0, cond, uc,
                                    0.0, **gk)

    with torch.no_grad():
        # set-up (like a compile step, outside warm-up and timing): the first call of a trajectory runs eagerly and fills
        # every cache, the second captures the whole sampler step into one hipGraph; W warm-up and K timed steps replay it
        scratch = x.clone()
        for i in range(2):
            scratch = step(i, scratch)
        del scratch
        for i in range(Wm):
            x = step(i, x)
        if world > 1:
            exchange_anchor_latents(anchors)  # untimed: first use of the all-gather (RCCL sets channels up lazily)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        if world > 1:
            exchange_anchor_latents(anchors)
        for i in range(Wm, Wm + K):
            x = step(i, x)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        elapsed = time.perf_counter() - t0
    if not torch.isfinite(x).all():
        raise SystemExit("non-finite sampler state")
    tmax = torch.tensor([elapsed], device=device, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed = float(tmax.item())

    roofline = None
    if not args.no_roofline:
        eng = net.engine()
        # events bracket individual launches: this one instrumented step runs eagerly (no step graph, no network graph)
        was_graph, eng.use_graph = eng.use_graph, False
        was_step, sampler._step_graphs.disabled = sampler._step_graphs.disabled, True
        ops.prof_enable(True)
        with torch.no_grad():
            step(min(Wm + K, num_sigmas - 2), x)
        prof = ops.prof_collect()
        ops.prof_enable(False)
        eng.use_graph, sampler._step_graphs.disabled = was_graph, was_step
        mm = {k: prof[k] for k in ("gemm", "conv", "attention")}
        dom = max(mm, key=lambda k: mm[k]["ms"])
        d = mm[dom]
        ach = d["work"] / (d["ms"] * 1e-3) / 1e12 if d["ms"] > 0 else 0.0
        # HBM traffic of the dominant kernel class: measured by separate rocprofv3 --pmc passes of this same command
        # (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE; tools/traffic_from_pmc.py writes profiles/rNN_traffic.json,
        # stamped with the git revision and command it was taken on).  It cannot be collected from inside this process,
        # so the line names its source; `alg_bytes` (same unit, per launch) is computed live next to it.
        traffic, traffic_source = None, None
        tfiles = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_traffic.json"))
        if tfiles and (T, hw) == (21, 72):
            tj = json.load(open(os.path.join(ROOT, "profiles", tfiles[-1])))
            t = tj.get(dom)
            traffic = t["bytes_per_launch"] if t else None
            traffic_source = {"file": "profiles/" + tfiles[-1], "git": tj.get("git"), "command": tj.get("command")}
        roofline = {
            "bound": "mfma", "kernel": {"gemm": "gemm_kernel<plain|GEGLU>", "conv": "gemm_kernel<conv3x3>",
                                        "attention": "attn_kernel"}[dom],
            "achieved": ach, "peak": PEAK_FP16_MFMA_TFLOPS, "unit": "TFLOP/s",
            "frac": ach / PEAK_FP16_MFMA_TFLOPS, "traffic": traffic, "traffic_unit": "HBM bytes per launch (PMC)",
            "traffic_source": traffic_source,
            "alg_bytes": d["bytes"] / max(d["launches"], 1),
            "alg_flop": d["work"] / max(d["launches"], 1),
            "launches": d["launches"], "avg_launch_ms": d["ms"] / max(d["launches"], 1),
            "classes_ms": {k: round(v["ms"], 3) for k, v in prof.items()},
            "classes_tflops": {k: (v["work"] / (v["ms"] * 1e-3) / 1e12 if v["ms"] > 0 else 0.0)
                               for k, v in mm.items()},
            "classes_alg_gbps": {k: (v["bytes"] / (v["ms"] * 1e-3) / 1e9 if v["ms"] > 0 else 0.0)
                                 for k, v in prof.items()},
        }

    vae = None
    if rank == 0 and not args.no_vae:
        try:
            # VAE decode of finished latents (reference autoencoder.py:40-48, chunk_size=1), outside `value`
            from seva.modules.autoencoder import AutoEncoder

            ae = AutoEncoder(chunk_size=1, random_init=True).to(device)
            zl = (x[:7] / x[:7].std() * 0.18215).contiguous()  # 7 frames per pass (AutoEncoder's default execution chunk)
            with torch.no_grad():
                ae.decode(zl)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                img = ae.decode(zl)
                torch.cuda.synchronize()
            dtv = (time.perf_counter() - t1) / zl.shape[0]
            with torch.no_grad():  # encode of the input / anchor views (reference autoencoder.py:21-35), chunk_size=1
                ae.encode(img)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                ae.encode(img)
                torch.cuda.synchronize()
            dte = (time.perf_counter() - t1) / img.shape[0]
            vae = {"ms_per_frame": dtv * 1e3, "frames_per_sec": 1.0 / dtv, "frame": f"{img.shape[-2]}x{img.shape[-1]}",
                   "encode_ms_per_frame": dte * 1e3, "frames_per_pass": int(zl.shape[0]),
                   "weights": "random-init SD-2.1 VAE topology (parity unpinned)"}
            del ae
            # CLIP ViT-H-14 image conditioner (reference conditioner.py:36-39), once per window on the input views, outside `value`
            from seva.modules.conditioner import CLIPConditioner

            clip = CLIPConditioner(random_init=True).to(device)
            with torch.no_grad():
                clip(img[:1])
                clip(img[:1])
                ts = []
                for _ in range(3):  # median of three: the second call on a fresh box still pays allocator growth
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    clip(img[:1])
                    torch.cuda.synchronize()
                    ts.append((time.perf_counter() - t1) * 1e3)
            vae["clip_conditioner_ms_per_frame"] = sorted(ts)[1]
            del clip
        except Exception as e:  # noqa: BLE001 -- an auxiliary leg never costs the headline line
            vae = {"error": f"{type(e).__name__}: {e}"[:400]}

    # The headline line is complete here.  Everything below (other configs, the strong-scaling trajectory leg, the CPU baseline)
    # is reported next to it and must never cost it: a leg that raises is recorded as {"error": ...}, and a leg that hangs (a
    # collective some rank never reaches) is cut by a watchdog that prints the line as it stands and ends the process.
    flop = FLOP_PER_STEP.get((T, hw))
    value = world * K / elapsed
    out = {
        # the headline name is reserved for the headline shape; other shapes are labelled as what they are
        "metric": (f"denoising steps/sec, 1.3B Seva @ {T}x{hw * 8}x{hw * 8} views"
                   + ("" if (T, hw) == (21, 72) else " (NOT the 21x576x576 headline shape)")
                   + ("" if args.precision == "f16" else " [fp8 weights + activations, BASELINE config 5: not the f16 parity mode]")),
        "value": value, "unit": "denoising steps/s", "n_gpus": world, "steps": K, "warmup": Wm,
        "ms_per_step": elapsed / K * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f16" if args.precision == "f16" else "f8e4m3 (C>=640 levels) + f16 (C=320 level)",
        "data": "synthetic",
        "novel_views_per_sec": value * (T - N_INPUT_VIEWS) / NUM_STEPS_PER_WINDOW,
        "hipgraph": {"whole_step": sampler._step_graphs.captures > 0,
                     "step_replays": getattr(sampler._step_graphs.graph, "replays", 0),
                     "network_only": bool(net.engine().use_graph) and sampler._step_graphs.captures == 0},
        "config": {"workload": f"Seva 1.3B (1,263,968,004 params, random-init), one {T}-view window per GPU, "
                               f"{hw * 8}x{hw * 8} px (latent {hw}x{hw}), CFG batch {2 * T}, Euler-EDM step, "
                               "MultiviewCFG(1.2) cfg 2.0; novel_views_per_sec = value x (T - 1 input view) / 50 steps per window",
                   "views": T, "latent": hw, "windows": world,
                   "flop_per_step": flop,
                   "model_tflops": (flop * value / 1e12) if flop else None},
        "roofline": roofline, "cpu_baseline": None, "vae_decode": vae,
        "other_configs": None, "trajectory": None,
        "cond_assembly": {"ms": cond_assembly_ms,
                          "what": "camera normalisation (host) + Pluecker maps + cond/uc assembly (HIP), once per window, outside `value`; median of 3"},
    }
    print_lock = threading.Lock()
    state = {"printed": False, "leg": "none"}

    def emit():
        # the lock is held across print AND flush: a watchdog firing while the main thread is inside emit() waits for the
        # whole line instead of cutting it, and the line is printed exactly once
        with print_lock:
            if not state["printed"] and rank == 0:
                print(json.dumps(out), flush=True)
            state["printed"] = True

    def watchdog():
        # a hung auxiliary leg (e.g. a collective some rank never reaches): the headline line is printed as it stands, the leg
        # that was running is named, and the process ends NON-ZERO on every rank -- a hang is not reported as success
        out["legs_cut_short"] = (f"auxiliary leg '{state['leg']}' did not finish within {leg_limit:.0f} s; the line is printed "
                                 "without it and the process exits with code 3")
        emit()
        os._exit(3)

    leg_limit = float(os.environ.get("SEVA_BENCH_LEG_TIMEOUT", "900"))
    timer = threading.Timer(leg_limit, watchdog)
    timer.daemon = True
    timer.start()

    def leg(name, fn, *a):
        state["leg"] = name
        try:
            return fn(*a)
        except Exception as e:  # noqa: BLE001 -- reported, never fatal for the headline line
            return {"error": f"{type(e).__name__}: {e}"[:400]}

    def fp8_leg():
        net.set_precision("fp8")
        try:
            r8 = steps_leg(net, device, T, hw, args.other_steps)
            r8["dtype"] = "f8e4m3 (C>=640 levels) + f16 (C=320 level); separate accuracy class, not the parity mode"
            return r8
        finally:
            net.set_precision("f16")

    # whatever happens below -- an exception outside a leg included -- the headline line is printed
    try:
        if not args.no_other_configs and (T, hw) == (21, 72) and args.precision == "f16":
            if world == 1:
                # BASELINE configs 2, 3, 5 and (on this one GPU) 4 -- outside `value`
                other = {"config2_T8": leg("config2_T8", steps_leg, net, device, 8, hw, args.other_steps),
                         "config3_T24_one_window": leg("config3_T24", steps_leg, net, device, 24, hw, args.other_steps),
                         "config5_fp8_T21": leg("config5_fp8", fp8_leg)}
                other["config4_trajectory168_one_gpu"] = leg("config4_trajectory", trajectory_leg, net, device, rank, world, 168, hw, T,
                                                             args.leg_traj_steps, False)
                out["other_configs"] = other
            else:
                out["trajectory"] = leg("trajectory", trajectory_leg, net, device, rank, world, 168, hw, T, args.leg_traj_steps, True)

        if rank == 0 and world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = leg("cpu_baseline", cpu_baseline, sd)
        state["leg"] = "none"
    except BaseException as e:  # noqa: BLE001
        out["legs_error"] = f"{type(e).__name__}: {e}"[:400]
        raise
    finally:
        timer.cancel()
        emit()
    if world > 1:
        dist.barrier()  # rank 0 alone ran the VAE / CPU legs: nobody tears the communicator down under it
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
