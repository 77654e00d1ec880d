#!/usr/bin/env python3
"""bench.py -- images/sec of the candidate-evaluation hot path on MI355X.

A "step" is one pass of the hot path over one batch: B images sampled with the searched 4-step
DDIM schedule [153, 424, 926, 690] through the ADM-G ImageNet-64 UNet (classifier-guided when
--workload guided), ending in the fused uint8 NHWC pack.  Synthetic data: x_T ~ N(0,1), labels
~ U{0..999}, random-init weights of the real architecture (no checkpoint is reachable offline).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--workload guided|unguided|adm256|sd]

--workload adm256 = the 256x256 line of the north star: ADM LSUN-256 dynamic UNet (search_lsun_cat.sh:1; 552.8 M, 2242.9
GFLOP per image and evaluation), unconditional, uniform 5-step DDIM (the search's `--time_step 5` start candidate).

N > 1 is launched by the driver as `python -m torch.distributed.run --nproc-per-node N ... bench.py
--gpus N ...` (one rank per GPU, RCCL): the batch shards by image with no data-path collective
(weak scaling); timing is barrier + synchronize bracketed, MAX over ranks.  Rank 0 prints ONE JSON
line with the metric, the dominant kernel's roofline (HIP events on the launch stream; with classifier guidance
the gradient network runs concurrently on a second stream, so `roofline.isolated` adds the kernel's launch time from
one untimed batch with the two networks in sequence) and the CPU oracle baseline timed on the host cores.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

SCHEDULE = [153, 424, 926, 690]            # reference scripts/classifier_sample_generate_image.py:164
GFLOP_UNET = 219.64                        # per image per UNet eval
GFLOP_GUIDE = 93.85                        # classifier fwd + backward-data per image per step
PEAK_BF16_TFLOPS = 2500.0                  # dense MFMA bf16, MI355X_MICROARCH.md


SCHEDULE_256 = [0, 200, 400, 600, 800]     # space_timesteps(1000, "ddim5"): the start candidate of `--time_step 5` (search_lsun_cat.sh:9)
GFLOP_UNET_256 = 2242.87                   # ADM LSUN-256, per image per UNet eval


def pmc_traffic(workload):
    """HBM bytes per launch of the workload's dominant kernel, from the committed rocprofv3 --pmc passes of THIS round's
    build (profiles/r02/pmc_dominant_kernel_traffic.json: separate FETCH_SIZE / WRITE_SIZE passes over `bench.py --steps 1`
    with the guide's gfx950 corrections, reduced by tools/pmc_summary.py).  PMC counters cannot be read inside the timed
    run, so the figure is a measured constant of the build, refreshed whenever the kernel changes; None if absent."""
    tp = os.path.join(ROOT, "profiles", "r02", "pmc_dominant_kernel_traffic.json")
    if not os.path.exists(tp):
        return None
    with open(tp) as f:
        return json.load(f).get(workload, {}).get("hbm_bytes_per_launch")


def adm256_flags():
    from autodiffusion_amd.script_util import model_and_diffusion_defaults
    d = model_and_diffusion_defaults()
    d.update(attention_resolutions="32,16,8", class_cond=False, diffusion_steps=1000, dropout=0.1, image_size=256,
             learn_sigma=True, noise_schedule="linear", num_channels=256, num_head_channels=64, num_res_blocks=2,
             resblock_updown=True, use_fp16=True, use_scale_shift_norm=True, use_dynamic_unet=True)
    return d


def adm64_flags(class_cond=True, dynamic=False):
    from autodiffusion_amd.script_util import model_and_diffusion_defaults
    d = model_and_diffusion_defaults()
    d.update(attention_resolutions="32,16,8", class_cond=class_cond, diffusion_steps=1000, dropout=0.1,
             image_size=64, learn_sigma=True, noise_schedule="cosine", num_channels=192,
             num_head_channels=64, num_res_blocks=3, resblock_updown=True, use_new_attention_order=True,
             use_fp16=True, use_scale_shift_norm=True, use_dynamic_unet=dynamic)
    return d


def cpu_baseline(sample_batch=4, workload="adm64"):
    """The CPU oracle on BASELINE config 1 (ADM-64 unconditional, ddim4, batch 4, fp32) on this host's cores
    (workload="sd": the oracle's Stable-Diffusion v1 latent UNet, one evaluation of one 64x64 latent, scaled to the
    6-step guided candidate = 12 evaluations per finished latent).

    BASELINE.md section 4: 1 warm-up + median of 3.  The thread count is swept (one single-evaluation probe each over
    8 / 16 / 32 / 64 threads; the survey measured the REFERENCE at 0.44 images/s on 8 cores) and the fastest count runs the
    timed loops.  Oversubscribed counts are not probed: on a 256-thread host one evaluation took 150 s with 256 threads
    against 0.64 s with 16.  The whole leg stays under a minute."""
    import numpy as np
    from autodiffusion_amd.arch import build_unet_plan
    from oracle import nets, sampler as osm, schedule as osch
    if workload == "sd":
        from autodiffusion_amd.sd_arch import SD_V1, sd_unet_plan
        from oracle import sd_nets
        splan = sd_unet_plan(**SD_V1)
        g = torch.Generator().manual_seed(1234)
        SP = {}
        for k, shp in splan.param_shapes().items():
            if len(shp) >= 2:
                SP[k] = torch.randn(shp, generator=g) * (1.0 / float(np.prod(shp[1:])) ** 0.5)
            elif k.endswith("weight"):
                SP[k] = torch.ones(shp)
            else:
                SP[k] = torch.zeros(shp)
        cores = min(32, os.cpu_count() or 1)
        torch.set_num_threads(cores)
        xl, ctx, tl = torch.randn(1, 4, 64, 64, generator=g), torch.randn(1, 77, 768, generator=g), torch.tensor([500])
        runs = []
        with torch.no_grad():
            sd_nets.sd_unet_forward(SP, splan, xl, tl, ctx)  # warm-up
            for _ in range(2):
                t0 = time.time()
                sd_nets.sd_unet_forward(SP, splan, xl, tl, ctx)
                runs.append(time.time() - t0)
        dt = min(runs)
        evals = 2 * len(SD_CAND)
        return {"value": round(1.0 / (evals * dt), 5), "unit": "latents/sec", "cores": cores, "kind": "port",
                "sample": f"oracle (PyTorch-CPU fp32 restatement) SD v1 latent UNet: one evaluation of one 64x64 latent = "
                          f"{dt:.1f} s (best of 2 after a warm-up), x {evals} evaluations per finished latent "
                          f"(6 DDIM steps under classifier-free guidance)"}
    plan = build_unet_plan(64, 3, 192, 6, 3, (2, 4, 8), (1, 2, 3, 4), num_classes=None, num_head_channels=64,
                           use_scale_shift_norm=True, resblock_updown=True, use_new_attention_order=True)
    g = torch.Generator().manual_seed(1234)
    P = {}
    for k, shp in plan.param_shapes().items():
        if len(shp) >= 2:
            P[k] = torch.randn(shp, generator=g) * (1.0 / float(np.prod(shp[1:])) ** 0.5)
        elif k.endswith("weight"):
            P[k] = torch.ones(shp)
        else:
            P[k] = torch.zeros(shp)
    d = osch.OracleDiffusion(steps=1000, noise_schedule="cosine", learn_sigma=True, timestep_respacing="ddim4")
    x = torch.randn(sample_batch, 3, 64, 64, generator=torch.Generator().manual_seed(0))
    fn = lambda xx, t: nets.unet_forward(P, plan, xx, t)  # noqa: E731
    ncpu = os.cpu_count() or 1
    probe = {}
    t_eval = torch.zeros(sample_batch, dtype=torch.int64)
    for nt in sorted({n for n in (8, 16, 32, 64) if n <= ncpu} or {ncpu}):
        torch.set_num_threads(nt)
        with torch.no_grad():
            fn(x, t_eval)  # warm-up at this thread count
            t0 = time.time()
            fn(x, t_eval)
            probe[nt] = time.time() - t0
    best = min(probe, key=probe.get)
    torch.set_num_threads(best)
    runs = []
    for _ in range(3):
        t0 = time.time()
        osm.sample_loop(d, fn, x, use_ddim=True)
        runs.append(time.time() - t0)
    dt = sorted(runs)[1]
    return {"value": round(sample_batch / dt, 4), "unit": "images/sec", "cores": best, "kind": "port",
            "sample": f"oracle (PyTorch-CPU fp32 restatement), ADM-64 unconditional ddim4, batch of {sample_batch} images = "
                      f"{4 * sample_batch} UNet evals, median of 3 runs ({', '.join(f'{r:.1f}' for r in runs)} s) at the fastest of "
                      f"the probed thread counts {{{', '.join(f'{k}: {v:.2f} s/eval-batch' for k, v in sorted(probe.items()))}}}; "
                      f"host has {ncpu} logical cores; the reference itself: 0.44 images/s on 8 cores (BASELINE.md section 2)"}


SD_CAND = [94, 217, 354, 574, 834, 944]     # GD/sample_imagenet64_classifier_guidance_subnet.sh:11's 6-step candidate, sorted
SD_GFLOP_LATENT = 803.27                    # per latent per UNet evaluation


def run_sd(args, rank, world, dev, red_dev):
    """BASELINE config 4: one step = one candidate-evaluation batch of the Stable-Diffusion example -- N latents
    [N, 4, 64, 64] sampled with K = 6 searched DDIM steps under classifier-free guidance 7.5 (2 UNet evaluations per
    step, batched as 2N latents) through the v1 latent UNet; random-init weights, synthetic 77 x 768 conditioning.
    The VAE decode / CLIP encoder / pytorch_fid ends of the reference's get_cand_fid are not on this path."""
    import torch.distributed as dist
    from autodiffusion_amd import ops
    from autodiffusion_amd.sd_arch import SD_V1
    from autodiffusion_amd.sd_sampler import DDIMSampler, LatentDiffusion
    from autodiffusion_amd.sd_unet import UNetModel
    n = args.batch or 6
    unet = UNetModel(image_size=32, use_spatial_transformer=True, **SD_V1).to(dev)
    unet.set_torso(args.torso)
    unet.randomize_(1234).enable_graph().enable_splitk(os.environ.get("ADM_SD_SPLITK", "1") != "0")
    sampler = DDIMSampler(LatentDiffusion(unet, device=dev))
    g = torch.Generator(device=dev).manual_seed(99)
    c, uc = (torch.randn(n, 77, 768, device=dev, generator=g) for _ in range(2))

    def one_step(idx):
        x_T = torch.randn(n, 4, 64, 64, device=dev, generator=torch.Generator(device=dev).manual_seed(1000003 * idx + rank + 7))
        return sampler.sample(S=len(SD_CAND), batch_size=n, shape=[4, 64, 64], conditioning=c, verbose=False, eta=0.0, x_T=x_T,
                              unconditional_guidance_scale=7.5, unconditional_conditioning=uc, sampled_timestep=SD_CAND)[0]
    for w in range(max(1, args.warmup)):
        out = one_step(-1 - w)
    torch.cuda.synchronize()
    assert torch.isfinite(out).all()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s_ in range(args.steps):
        one_step(s_)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    roof = None
    if not args.no_kernel_events:  # per-launch HIP events need the eager path: one untimed evaluation outside the graph
        unet.enable_graph(False)
        unet.enable_splitk(False)   # the roofline kernel's launches are the one-pass ones
        ops.CONV_PROFILE = []
        x = torch.randn(2 * n, 4, 64, 64, device=dev)
        unet(x, torch.full((2 * n,), 500, device=dev, dtype=torch.int64), torch.cat([uc, c]))
        torch.cuda.synchronize()
        prof, ops.CONV_PROFILE = ops.CONV_PROFILE, None
        dom = [p for p in prof if p[3] == (6, 9, True, 2)]
        if dom:
            ms = sum(p[0].elapsed_time(p[1]) for p in dom)
            fl = sum(p[2] for p in dom)
            roof = {"bound": "mfma", "kernel": "conv_kernel<2, 4, 8, 2, 2, 9, 324, 2, 1> (fused GN+SiLU+conv3x3, 256-pixel x 128-channel tile)",
                    "achieved": round(fl / (ms * 1e-3) / 1e12, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(fl / (ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4), "traffic": pmc_traffic("sd"), "launches": len(dom),
                    "avg_launch_us": round(ms * 1e3 / len(dom), 2), "avg_launch_gflop": round(fl / len(dom) / 1e9, 3)}
    if rank == 0:
        value = world * n * args.steps / elapsed
        cpu = None
        if not args.no_cpu_baseline and world == 1:
            del unet, sampler
            torch.cuda.empty_cache()
            cpu = cpu_baseline(workload="sd")
        print(json.dumps({
            "metric": "latents/sec (node), Stable-Diffusion v1 latent UNet, 6-step searched DDIM, guidance 7.5",
            "value": round(value, 2), "unit": "latents/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16" if args.torso == "bf16" else "f16",
            "data": "synthetic (x_T ~ N(0,1), conditioning ~ N(0,1) [N,77,768], random-init weights of the SD v1 UNet architecture)",
            "config": {"workload": f"Stable-Diffusion v1 latent UNet (859.5 M), searched DDIM {SD_CAND}, classifier-free guidance 7.5, "
                                   f"{n} latents per GPU and step (64x64x4), {args.torso}, hipGraph replay; VAE / CLIP / FID not on this path",
                       "global_batch": world * n, "latent_size": 64, "sampler_steps": len(SD_CAND),
                       "parallelism": f"dp{world} (latent-sharded, no data-path collective)"},
            "model_tflops": round(value * 2 * len(SD_CAND) * SD_GFLOP_LATENT / 1e3, 1),
            "roofline": roof, "cpu_baseline": cpu}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=None, help="images per step and GPU (default 256; --workload sd: 6 latents)")
    ap.add_argument("--workload", default="auto", choices=["auto", "guided", "unguided", "adm256", "sd"],
                    help="auto/guided = the headline (ADM-G ImageNet-64, BASELINE configs[1]); adm256 = ADM LSUN-256 dynamic UNet, "
                         "uniform 5-step DDIM (the north star's 256x256 line); sd = BASELINE config 4 "
                         "(Stable-Diffusion v1 latent UNet, 6 searched DDIM steps, classifier-free guidance 7.5)")
    ap.add_argument("--torso", default="bf16", choices=["bf16", "fp16"],
                    help="16-bit element type of the UNet torso: bf16 (BASELINE configs[1] names it) or fp16 (the reference's own "
                         "torso type, libadm_hip_f16.so: same kernels, 11 mantissa bits; the classifier's backward network stays bf16)")
    ap.add_argument("--graph", action="store_true",
                    help="replay the UNet evaluation and the guidance gradient as captured hipGraphs (small batches: the host's "
                         "~60 ms of launch work per guided step is the floor below batch ~100); the roofline's per-launch events "
                         "then come from one extra eager batch after the timed region")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl (= RCCL) for real multi-GPU runs; gloo only to rehearse N ranks on one GPU")
    ap.add_argument("--with-fid", action="store_true",
                    help="also run the FID stage inside every step: HIP Inception-v3 pool3 of the step's uint8 batch (random weights: "
                         "the checkpoint is not in the image) + the float64 Gram accumulation; the headline line leaves it out "
                         "(BASELINE's metric is sampling throughput)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true")
    ap.add_argument("--conv-breakdown", action="store_true",
                    help="print per-shape conv time / TFLOP/s (HIP events) to stderr")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N")
    dev = torch.device(f"cuda:{local_rank % max(1, torch.cuda.device_count())}")
    torch.cuda.set_device(dev)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.dist_backend == "nccl":
            dist.init_process_group(backend="nccl", init_method="env://", device_id=dev)
        else:
            dist.init_process_group(backend="gloo", init_method="env://")
    red_dev = dev if args.dist_backend == "nccl" else torch.device("cpu")

    if args.workload == "sd":
        run_sd(args, rank, world, dev, red_dev)
        if world > 1:
            dist.destroy_process_group()
        return
    w256 = args.workload == "adm256"
    if args.batch is None:
        # 256x256: 64 images per step = 0.7 s of GPU time; the reference's own launch flag is 36 (search_lsun_cat.sh:2) and
        # throughput is flat beyond (every launch already fills the chip), so "as large as fits" would only stretch the run
        args.batch = 64 if w256 else 256

    from autodiffusion_amd import ops
    from autodiffusion_amd.evaluate import CandidateEvaluator
    from autodiffusion_amd.script_util import (args_to_dict, classifier_defaults, create_classifier,
                                               create_model_and_diffusion, model_and_diffusion_defaults)

    guided = args.workload in ("auto", "guided")
    flags = adm256_flags() if w256 else adm64_flags(class_cond=True)
    size = 256 if w256 else 64
    schedule = SCHEDULE_256 if w256 else SCHEDULE
    gflop_unet = GFLOP_UNET_256 if w256 else GFLOP_UNET
    # the launch mix's dominant conv symbol: (tiling variant, taps, map > 8x8, prologue) -- 192-wide tiles for ADM-64's
    # multiples of 192 channels, 128-wide tiles for LSUN-256's multiples of 256
    dom_key = (6, 9, True, 2) if w256 else (5, 9, True, 2)
    dom_name = ("conv_kernel<2, 4, 8, 2, 2, 9, 324, 2, 1> (fused GN+SiLU+conv3x3, 256-pixel x 128-channel tile)" if w256 else
                "conv_kernel<2, 4, 8, 3, 2, 9, 324, 2, 1> (fused GN+SiLU+conv3x3, 256-pixel x 192-channel tile)")
    model, diffusion = create_model_and_diffusion(**args_to_dict(argparse.Namespace(**flags),
                                                                 model_and_diffusion_defaults().keys()))
    model.to(dev).randomize_(1234).convert_to_fp16()
    model.set_torso(args.torso)
    classifier = None
    if guided:
        try:
            cf = classifier_defaults()
            cf.update(image_size=64, classifier_depth=4)
            classifier = create_classifier(**cf)
            classifier.to(dev).randomize_(4321)
            if not hasattr(classifier, "log_prob_grad"):
                raise NotImplementedError
        except (ImportError, NotImplementedError):
            if args.workload == "guided":
                raise
            classifier, guided = None, False

    ev = CandidateEvaluator(model, diffusion, classifier=classifier, image_size=size, use_ddim=True,
                            classifier_scale=1.0, class_cond=not w256, device=dev, use_graph=args.graph)
    ev.set_candidate(schedule)
    B = args.batch

    fid_net = fid_acc = None
    if args.with_fid:
        from autodiffusion_amd.fid import ActivationAccumulator
        from autodiffusion_amd.inception import InceptionV3
        fid_net = InceptionV3().to(dev)
        fid_net.weights_loaded = True     # random weights on purpose (throughput run): no warning
        fid_acc = ActivationAccumulator(2048, dev)

    def one_step(step_idx):
        # deterministic, layout-independent seeding per (step, rank)
        u8 = ev.sample_batch(B, seed=(1000003 * step_idx + rank))
        if fid_net is not None:
            fid_acc.add_from(fid_net.features, u8)
        return u8

    for w in range(args.warmup):
        one_step(-1 - w)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    if not args.no_kernel_events and not args.graph:
        ops.CONV_PROFILE = []
        # HIP events only around the dominant kernel's launches (all conv launches with --conv-breakdown): a pair of
        # events per launch costs the unguided workload 3.5 % when every conv carries one
        ops.CONV_PROFILE_KEY = None if args.conv_breakdown else dom_key
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(args.steps):
        one_step(s)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    roof = None
    if args.graph and not args.no_kernel_events:   # the replayed launches carry no events: one eager batch for the roofline
        model.enable_graph(False)
        if classifier is not None:
            classifier.enable_graph(False)
        ops.CONV_PROFILE, ops.CONV_PROFILE_KEY = [], (None if args.conv_breakdown else dom_key)
        t0e = time.perf_counter()
        one_step(args.steps + 1)
        torch.cuda.synchronize()
        eager_s = time.perf_counter() - t0e
    if ops.CONV_PROFILE is not None:
        prof, ops.CONV_PROFILE = ops.CONV_PROFILE, None
        # dominant kernel symbol: conv_kernel<2, 4, 8, 3, 2, 9, 324, 2, 1> = fused GN+SiLU prologue, 3x3 conv,
        # 256-pixel x 192-channel tile, 8 waves
        if args.conv_breakdown and rank == 0:
            agg = {}
            for e0, e1, f, key, shape in prof:
                a = agg.setdefault((key, shape), [0.0, 0.0, 0])
                a[0] += e0.elapsed_time(e1); a[1] += f; a[2] += 1
            for (key, shape), (ms_, fl_, cnt) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
                print(f"conv {key} nhwc_in={shape[:4]} cout={shape[4]} x{cnt // args.steps}: "
                      f"{ms_ / args.steps:8.2f} ms/batch {fl_ / ms_ / 1e9:7.1f} TFLOP/s", file=sys.stderr)
        dom = [p for p in prof if p[3] == dom_key]
        if dom:
            ms = sum(p[0].elapsed_time(p[1]) for p in dom)
            fl = sum(p[2] for p in dom)
            achieved = fl / (ms * 1e-3) / 1e12
            traffic = pmc_traffic("adm256" if w256 else ("guided" if guided else "unguided"))
            roof = {"bound": "mfma", "kernel": dom_name,
                    "achieved": round(achieved, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(achieved / PEAK_BF16_TFLOPS, 4), "traffic": traffic,
                    "launches": len(dom), "avg_launch_us": round(ms * 1e3 / len(dom), 2),
                    "avg_launch_gflop": round(fl / len(dom) / 1e9, 3),
                    "share_of_step_time": round(ms * 1e-3 / (eager_s if args.graph else elapsed), 3)}
            if args.graph:
                roof["how"] = "one eager batch after the timed region (the timed steps replay hipGraphs, which carry no events)"
            if guided and ev.active_diffusion.overlap_guidance:
                # in the timed region the guidance gradient runs on a second stream, so the launches above share the CUs
                # with its kernels (their event-bracketed time is not the kernel's own speed): one extra, untimed batch
                # with the two networks in sequence gives the kernel's duration when it owns the chip
                ev.active_diffusion.overlap_guidance = False
                ops.CONV_PROFILE = []
                one_step(args.steps)
                torch.cuda.synchronize()
                iso = [p for p in ops.CONV_PROFILE if p[3] == dom_key]
                ops.CONV_PROFILE = None
                ev.active_diffusion.overlap_guidance = True
                ims = sum(p[0].elapsed_time(p[1]) for p in iso)
                ifl = sum(p[2] for p in iso)
                roof["concurrency"] = "timed region: UNet and classifier-guidance kernels overlap on two HIP streams"
                roof["isolated"] = {"achieved": round(ifl / (ims * 1e-3) / 1e12, 2),
                                    "frac": round(ifl / (ims * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4),
                                    "avg_launch_us": round(ims * 1e3 / len(iso), 2), "launches": len(iso),
                                    "how": "one untimed batch after the timed region, networks in sequence on one stream"}

    if rank == 0:
        imgs = world * B * args.steps
        value = imgs / elapsed
        gflop_img = len(schedule) * (gflop_unet + (GFLOP_GUIDE if guided else 0.0))
        if w256:
            wl = (f"ADM LSUN-256 dynamic UNet (552.8 M), unconditional, uniform {len(schedule)}-step DDIM {schedule}, "
                  f"batch={B} per GPU, {args.torso}")
        else:
            wl = (("ADM-G ImageNet-64 classifier-guided" if guided else
                   "ADM ImageNet-64 class-conditional, UNGUIDED (classifier guidance not in this run)")
                  + f", searched 4-step DDIM {schedule}, batch={B} per GPU, {args.torso}")
        out = {
            "metric": ("images/sec (node), ADM LSUN-256 5-step DDIM" if w256 else
                       "images/sec (node), ADM-G ImageNet-64 4-step DDIM"),
            "value": round(value, 2), "unit": "images/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 2),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16" if args.torso == "bf16" else "f16",
            "data": ("synthetic (x_T ~ N(0,1), random-init weights of the ADM LSUN-256 architecture)" if w256 else
                     "synthetic (x_T ~ N(0,1), y ~ U{0..999}, random-init weights of the ADM-G-64 architecture)"),
            "config": {"workload": wl,
                       "global_batch": world * B, "image_size": size, "sampler_steps": len(schedule),
                       "parallelism": f"dp{world} (image-sharded, no data-path collective)",
                       "launch": "hipGraph replay" if args.graph else "eager",
                       "fid_stage_in_step": bool(args.with_fid)},
            "model_tflops": round(value * gflop_img / 1e3, 1),
            "roofline": roof,
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline()
            if w256:
                out["cpu_baseline"]["note"] = ("timed on BASELINE config 1 (64x64): one 256x256 evaluation is 2242.9 GFLOP, "
                                               "10.2 x a 64x64 one, too long for a bounded sample of this workload")
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
