#!/usr/bin/env python3
"""bench.py -- images/sec of the candidate-evaluation hot path on MI355X.

A "step" is one pass of the hot path over one batch: B images sampled with the searched 4-step
DDIM schedule [153, 424, 926, 690] through the ADM-G ImageNet-64 UNet (classifier-guided when
--workload guided), ending in the fused uint8 NHWC pack.  Synthetic data: x_T ~ N(0,1), labels
~ U{0..999}, random-init weights of the real architecture (no checkpoint is reachable offline).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--workload guided|unguided|adm128|adm256|sd|candidate]

--workload adm256 = the 256x256 line of the north star: ADM LSUN-256 dynamic UNet (search_lsun_cat.sh:1; 552.8 M, 2242.9
GFLOP per image and evaluation), unconditional, uniform 5-step DDIM (the search's `--time_step 5` start candidate);
`--skip-layers auto|<json>` gives every step a layer-skip list and `--with-fid` adds the pooled-FID stage: BASELINE config 5.
--workload adm128 = BASELINE config 3's per-GPU unit: ADM-G ImageNet-128 (configs/128_guided_sample.sh:1-3: 421.5 M UNet with
128 / 192 / 256-wide attention heads + the 128x128 depth-2 classifier), classifier-guided, a 10-step candidate, batch 32.
--workload candidate = ONE WHOLE get_cand_fid per step at the reference's own search flags
(search_imagenet64_classifier_guidance.sh:1-20: batch 100, 5000 images): sampling (hipGraph replay) + HIP Inception pool3 +
float64 Gram + Frechet distance on the device; reports candidates/hour and the reference's reset / sample / fid_time split.

After the timed region every workload produces one more, untimed batch and checks it (finite float sample, uint8 batch
that is not constant; `output_check` in the JSON line): a non-finite batch makes bench.py exit non-zero.

N > 1: either launch form works.  `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...` (one rank per
GPU, RCCL) runs the ranks directly; a bare `python bench.py --gpus N ...` (WORLD_SIZE unset) LAUNCHES ITSELF: the parent starts
that same torchrun command as a child process before anything touches the GPU, relays rank 0's line and exits with the
child's code.  The batch shards by image with no data-path collective
(weak scaling); timing is barrier + synchronize bracketed, MAX over ranks.  `--force-dist` builds a process group even at N = 1
(a world-size-1 nccl group: every collective of the timing protocol and of the pooled-FID stage then runs through RCCL on a
one-GPU box).  With the default workload (`auto`) the same process then runs the other BASELINE workloads as short secondary
lines (`secondary`: adm256, sd, adm128, population; `--secondary none` to skip).  Rank 0 prints ONE JSON
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
GFLOP_UNET_128 = 614.69                    # ADM-G ImageNet-128, per image per UNet eval
GFLOP_GUIDE_128 = 93.0                     # 128x128 classifier (depth 2): 46.5 GFLOP forward + as much backward-data


def pmc_traffic(workload):
    """(HBM bytes per launch of the workload's dominant kernel, where the figure comes from).  PMC counters cannot be read
    inside the timed run: the figure is a COMMITTED constant of the build, measured by separate rocprofv3 `--pmc FETCH_SIZE`
    / `--pmc WRITE_SIZE` passes over `bench.py --steps 1` with the guide's gfx950 corrections (tools/gpu_call.sh pmc,
    tools/pmc_summary.py, tools/pmc_traffic_json.py) and refreshed whenever the kernel changes -- this round's file first,
    the previous round's if this round has not re-collected the workload; (None, None) if absent.  The JSON line names the
    file as `roofline.traffic_source`, so that a reader does not take it for a live measurement."""
    for rnd in ("r04", "r03", "r02"):
        tp = os.path.join(ROOT, "profiles", rnd, "pmc_dominant_kernel_traffic.json")
        if not os.path.exists(tp):
            continue
        with open(tp) as f:
            ent = json.load(f).get(workload)
        if ent and ent.get("hbm_bytes_per_launch") is not None:
            return ent["hbm_bytes_per_launch"], (f"committed constant: profiles/{rnd}/pmc_dominant_kernel_traffic.json "
                                                 f"(rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of that round's build; not read live)")
    return None, None


def adm256_flags():
    from autodiffusion_amd.script_util import model_and_diffusion_defaults
    d = model_and_diffusion_defaults()
    d.update(attention_resolutions="32,16,8", class_cond=False, diffusion_steps=1000, dropout=0.1, image_size=256,
             learn_sigma=True, noise_schedule="linear", num_channels=256, num_head_channels=64, num_res_blocks=2,
             resblock_updown=True, use_fp16=True, use_scale_shift_norm=True, use_dynamic_unet=True)
    return d


def adm128_flags():
    """configs/128_guided_sample.sh:1 (the reference runs this model in fp32: `--use_fp16 False`; the HIP torso is 16-bit)."""
    from autodiffusion_amd.script_util import model_and_diffusion_defaults
    d = model_and_diffusion_defaults()
    d.update(attention_resolutions="32,16,8", class_cond=True, image_size=128, learn_sigma=True, num_channels=256,
             num_heads=4, num_res_blocks=2, resblock_updown=True, use_fp16=True, use_scale_shift_norm=True)
    return d


def auto_skip_layers(layer_num, steps, frac=0.1, seed=0):
    """A deterministic layer-skip candidate in the shape the progressive joint search produces (search_dynamic_unet_..._
    progressive.sh: `--max_prun 0.1`): every step skips round(frac * layer_num) layers drawn by a seeded RNG."""
    import random
    rng = random.Random(seed)
    k = max(1, round(frac * layer_num))
    return [sorted(rng.sample(range(layer_num), k)) for _ in range(steps)]


def check_output(sample, u8):
    """Untimed verification of one produced batch: finite float sample, uint8 NHWC batch that is an image (not constant).
    -> the `output_check` object of the JSON line; raises SystemExit(3) on a non-finite or degenerate batch."""
    ok_f = bool(torch.isfinite(sample).all().item())
    u = u8.to(torch.float32)
    std = float(u.std().item())
    res = {"finite": ok_f, "u8_shape": list(u8.shape), "u8_checksum": int(u8.to(torch.int64).sum().item()),
           "u8_mean": round(float(u.mean().item()), 3), "u8_std": round(std, 3),
           "how": "one extra untimed batch after the timed region, same code path and seed schedule"}
    if not ok_f or not (std > 0.0):
        print(json.dumps({"error": "bench.py: the produced batch is non-finite or constant", "output_check": res}), flush=True)
        raise SystemExit(3)
    return res


def _cpulist(text):
    out = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        out.extend(range(int(lo), int(hi or lo) + 1))
    return out


def gpu_local_cpus(index):
    """CPUs of the NUMA node GPU `index` hangs off (/sys/bus/pci/devices/<bdf>/local_cpulist), or None if it cannot be read."""
    try:
        p = torch.cuda.get_device_properties(index)
        bdf = f"{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
        with open(f"/sys/bus/pci/devices/{bdf}/local_cpulist") as f:
            return _cpulist(f.read()) or None
    except Exception:
        return None


def cpu_core_key(cpu):
    """The lowest-numbered hardware thread of the physical core `cpu` belongs to (its SMT siblings share the key)."""
    try:
        with open(f"/sys/devices/system/cpu/cpu{cpu}/topology/thread_siblings_list") as f:
            return min(_cpulist(f.read()))
    except Exception:
        return cpu


def rank_cpu_slice(cpus, local_rank, local_world, local_of=None, ndev=1, core_key=None):
    """The CPU slice of one local rank.  `cpus`: the CPUs the job may use; `local_of(device index)`: the CPUs local to that GPU's
    NUMA node (None = unknown).  Ranks whose GPUs share a NUMA node split that node's CPUs among themselves (a launch thread
    on the far socket pays the inter-socket hop on every doorbell write); without topology information the job's CPUs are cut
    into `local_world` contiguous slices.  `core_key(cpu)` groups SMT siblings so that a slice holds whole physical cores.
    Pure function (tests/test_bench_host.py)."""
    if core_key is not None:   # keep the hardware threads of one physical core in one slice: two ranks never share a core
        cpus = sorted(cpus, key=lambda c: (core_key(c), c))
    if local_of is not None:
        mine = local_of(local_rank % ndev)
        if mine:
            pool = [c for c in cpus if c in set(mine)]
            peers = [r for r in range(local_world) if local_of(r % ndev) == mine]
            if pool and local_rank in peers:
                per = max(1, len(pool) // len(peers))
                i = peers.index(local_rank)
                got = pool[i * per:(i + 1) * per]
                if got:
                    return got
    per = max(1, len(cpus) // local_world)
    return cpus[local_rank * per:(local_rank + 1) * per] or cpus


def pin_rank(local_rank, local_world):
    """One process per GPU on one node: every rank issues ~3400 launches per step from ONE Python thread; a rank whose
    launch thread is descheduled stalls its GPU queue and the MAX-over-ranks time.  Each rank gets its own slice of the CPUs this
    job may use -- of its GPU's NUMA node where the topology can be read (rank_cpu_slice) -- by sched_setaffinity, and a thread
    budget of that size for OMP / torch intra-op pools (ADM_BENCH_AFFINITY=0 leaves the process alone).  -> description for the
    JSON line."""
    if local_world <= 1 or os.environ.get("ADM_BENCH_AFFINITY", "1") == "0":
        return None
    try:
        cpus = sorted(os.sched_getaffinity(0))
    except AttributeError:
        return None
    ndev = max(1, torch.cuda.device_count())
    cache = {}

    def local_of(i):
        if i not in cache:
            cache[i] = gpu_local_cpus(i) if torch.cuda.is_available() else None
        return cache[i]
    mine = sorted(rank_cpu_slice(cpus, local_rank, local_world, local_of, ndev, cpu_core_key))
    numa = bool(local_of(local_rank % ndev))
    try:
        os.sched_setaffinity(0, mine)
    except OSError:
        return None
    nthr = max(1, min(len(mine), int(os.environ.get("OMP_NUM_THREADS", len(mine)))))
    os.environ["OMP_NUM_THREADS"] = str(nthr)
    torch.set_num_threads(nthr)
    def ranges(v):
        out, a = [], None
        for c in v + [None]:
            if a is None:
                a = b = c
            elif c is not None and c == b + 1:
                b = c
            else:
                out.append(f"{a}-{b}" if b != a else str(a))
                a = b = c
        return ",".join(out)
    return {"cpus": ranges(mine), "n_cpus": len(mine), "threads": nthr, "numa_local": numa}


class Ctx:
    """What one rank knows about the job: its place in it, its GPU, and whether the timing protocol's collectives run
    (`dist_on`: more than one rank, or --force-dist's world-size-1 group)."""

    def __init__(self, rank, world, local_rank, dev, red_dev, pin, dist_on, backend):
        self.rank, self.world, self.local_rank, self.dev, self.red_dev = rank, world, local_rank, dev, red_dev
        self.pin, self.dist_on, self.backend = pin, dist_on, backend


def timed(ctx, one_step, steps):
    """The contract's timed region: barrier + synchronize, EXACTLY `steps` steps, synchronize (this rank's own time), barrier +
    synchronize, MAX over ranks.  -> (elapsed_local, elapsed)."""
    import torch.distributed as dist
    torch.cuda.synchronize()
    if ctx.dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s_ in range(steps):
        one_step(s_)
    torch.cuda.synchronize()
    elapsed_local = time.perf_counter() - t0
    if ctx.dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if ctx.dist_on:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=ctx.red_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    return elapsed_local, elapsed


def gather_ranks(ctx, elapsed_local, extra=None):
    """Rank 0 collects (rank, device name, PCI bus id, CPU slice, per-rank elapsed seconds [+ extra]) from every rank and runs one
    all-reduce of ones on the timing protocol's device: `RCCL saw N ranks` becomes a recorded fact of the JSON line.
    None when no process group is up."""
    if not ctx.dist_on:
        return None
    import torch.distributed as dist
    dev = ctx.dev
    props = torch.cuda.get_device_properties(dev)
    bus = None
    if all(hasattr(props, a) for a in ("pci_domain_id", "pci_bus_id", "pci_device_id")):
        bus = f"{props.pci_domain_id:04x}:{props.pci_bus_id:02x}:{props.pci_device_id:02x}.0"
    me = {"rank": ctx.rank, "local_rank": ctx.local_rank, "device": torch.cuda.get_device_name(dev), "cuda_index": dev.index,
          "pci_bus_id": bus, "uuid": str(getattr(props, "uuid", "")) or None, "elapsed_s": round(elapsed_local, 4),
          "pid": os.getpid(), "cpu_affinity": ctx.pin}
    if extra:
        me.update(extra)
    every = [None] * ctx.world
    dist.all_gather_object(every, me)
    one = torch.ones(1, dtype=torch.float64, device=ctx.red_dev)
    dist.all_reduce(one)
    return {"ranks": every, "collective_check": {"backend": dist.get_backend(), "allreduce_sum_of_ones": float(one.item()),
                                                 "world_size": ctx.world}}


def hbm_peak_gb(dev):
    """Peak of the caching allocator's reserved HBM since the workload started (hipGraph pools included), GB."""
    return round(torch.cuda.max_memory_reserved(dev) / 1e9, 2)


# The error of the torso a line was produced with, next to its throughput: what the full-size parity tests ASSERT against the
# reference's own fp32 output (fixtures captured from the reference, tests/golden/) and what they last measured.
PARITY = {
    ("adm64", "bf16"): {"vs": "the reference's fp32 output on identical inputs (tests/golden/full_adm64.npz, full_loop64.npz)",
                        "asserted": "UNet eval rel < 2e-2; guided 4-step loop: sample rel < 2.5e-2, uint8 within 2 levels >= 90 %, within 8 >= 99 %",
                        "last_measured": "UNet eval 1.0e-2; guided loop 1.1e-2 relative, 95.5 % of uint8 pixels within 2 levels",
                        "test": "tests/test_hip_fullsize.py::test_adm64_unet_classifier_and_guided_loop_match_the_reference"},
    ("adm64", "fp16"): {"vs": "the reference's fp32 output on identical inputs (its own fp16 torso is 1.4e-3 away)",
                        "asserted": "UNet eval rel < 4e-3; guided loop (fp16 UNet + fp16 classifier): sample rel < 6e-3, uint8 within 2 levels >= 99.9 %",
                        "last_measured": "UNet eval 1.3e-3; guided loop <= 6e-3, >= 99.9 % within 2 levels",
                        "test": "tests/test_hip_fullsize.py::test_fp16_torso_matches_the_reference_at_its_own_precision"},
    ("adm128", "bf16"): {"vs": "the reference's fp32 output (tests/golden/full_adm128.npz)",
                         "asserted": "UNet eval rel < 2e-2 (fp16 torso < 4e-3); guided 10-step loop: sample rel < 2.5e-2, uint8 within 2 levels >= 90 %",
                         "test": "tests/test_hip_fullsize.py::test_adm128_unet_classifier_and_guided_10_step_loop_match_the_reference"},
    ("adm256", "bf16"): {"vs": "the reference's fp32 output (tests/golden/full_lsun256.npz; class-conditional: full_adm256cc.npz)",
                         "asserted": "UNet eval (with and without a layer-skip list) rel < 2e-2, output norm within 1 % (fp16 torso < 4e-3)",
                         "test": "tests/test_hip_fullsize.py::test_lsun256_dynamic_unet_matches_the_reference"},
    ("sd", "bf16"): {"vs": "the reference's fp32 output (tests/golden/full_sd_v1.npz)",
                     "asserted": "UNet eval rel < 2e-2, one-pass and split-K schedules (fp16 torso < 5e-3)",
                     "test": "tests/test_hip_fullsize.py::test_sd_v1_latent_unet_matches_the_reference"},
}


def parity_block(model_key, torso):
    p = PARITY.get((model_key, torso)) or PARITY.get((model_key, "bf16"))
    if p is None:
        return None
    out = dict(p)
    out["torso"] = torso
    out["fid_tolerance"] = "FID +-0.1 vs the reference is unpinned: no checkpoint / Inception weights / reference statistics in the image"
    return out


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

# the exact symbols of the launch classes that dominate a workload (rocprofv3 --kernel-trace names them; profiles/r04/)
CONV_SYMBOLS = {
    (5, 9, True, 2): "conv_kernel<2, 4, 8, 3, 2, 9, 324, 2, 1, false> (fused GN+SiLU+conv3x3, 256-pixel x 192-channel tile)",
    (6, 9, True, 2): "conv_kernel<2, 4, 8, 2, 2, 9, 324, 2, 1, false> (fused GN+SiLU+conv3x3, 256-pixel x 128-channel tile)",
    (6, 1, True, 0): "conv_kernel<2, 4, {8|4}, 2, 2, 1, {256|128}, 0, 1, false> (raw 1x1 conv, staged 128-channel tile: token projections)",
    (10, 1, True, 0): "conv1x1r_kernel<BM, 0> (raw 1x1 conv, activation tile resident in LDS)",
    (10, 1, True, 1): "conv1x1r_kernel<BM, 1> (GroupNorm-affine 1x1 conv, activation tile resident in LDS)",
}


def conv_class_name(key):
    if key in CONV_SYMBOLS:
        return CONV_SYMBOLS[key]
    v, taps, big, pro = key
    tile = {5: "192-channel 8-wave tile", 6: "128-channel 8-wave tile", 3: "16-channel tile", 7: "32x32x16-MFMA 192-channel tile",
            10: "LDS-resident 1x1 tile"}.get(v, f"variant {v}")
    return (f"conv launch class (variant {v}: {tile}; taps {taps}; {'maps > 8x8' if big else '8x8 maps'}; prologue "
            f"{ {0: 'raw', 1: 'affine', 2: 'GN+SiLU', 3: 'GN-backward epilogue', 4: 'GN+SiLU + folded skip 1x1'}.get(pro, pro) })")


def launch_mix(prof, top=4):
    """Conv launches of one profiled pass aggregated by launch class -> [{class, launches, ms, share_of_conv_time, tflops}], by time."""
    agg = {}
    for e0, e1, f, key, _shape in prof:
        a = agg.setdefault(key, [0.0, 0.0, 0])
        a[0] += e0.elapsed_time(e1); a[1] += f; a[2] += 1
    tot = sum(a[0] for a in agg.values()) or 1.0
    rows = sorted(agg.items(), key=lambda kv: -kv[1][0])[:top]
    return [{"key": list(k), "class": conv_class_name(k), "launches": c, "ms": round(ms, 3),
             "share_of_conv_time": round(ms / tot, 3), "tflops": round(fl / ms / 1e9, 1) if ms > 0 else None} for k, (ms, fl, c) in rows], tot


def roof_from(dom, kernel, traffic=None, traffic_src=None):
    ms = sum(p[0].elapsed_time(p[1]) for p in dom)
    fl = sum(p[2] for p in dom)
    achieved = fl / (ms * 1e-3) / 1e12
    return {"bound": "mfma", "kernel": kernel, "achieved": round(achieved, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
            "frac": round(achieved / PEAK_BF16_TFLOPS, 4), "traffic": traffic, "traffic_source": traffic_src, "launches": len(dom),
            "avg_launch_us": round(ms * 1e3 / len(dom), 2), "avg_launch_gflop": round(fl / len(dom) / 1e9, 3)}, ms


def run_sd(args, ctx):
    """BASELINE config 4: one step = one candidate-evaluation batch of the Stable-Diffusion example -- N latents
    [N, 4, 64, 64] sampled with K = 6 searched DDIM steps under classifier-free guidance 7.5 (2 UNet evaluations per
    step, batched as 2N latents) through the v1 latent UNet; random-init weights, synthetic 77 x 768 conditioning.
    The VAE decode / CLIP encoder / pytorch_fid ends of the reference's get_cand_fid are not on this path."""
    from autodiffusion_amd import ops
    from autodiffusion_amd.sd_arch import SD_V1
    from autodiffusion_amd.sd_sampler import DDIMSampler, LatentDiffusion
    from autodiffusion_amd.sd_unet import UNetModel
    rank, world, dev = ctx.rank, ctx.world, ctx.dev
    n = args.batch or 6
    unet = UNetModel(image_size=32, use_spatial_transformer=True, **SD_V1).to(dev)
    unet.set_torso(args.torso)
    unet.randomize_(1234).enable_graph().enable_splitk(os.environ.get("ADM_SD_SPLITK", "1") != "0")
    sampler = DDIMSampler(LatentDiffusion(unet, device=dev))
    g = torch.Generator(device=dev).manual_seed(99)
    c, uc = (torch.randn(n, 77, 768, device=dev, generator=g) for _ in range(2))
    last = {}

    def one_step(idx):
        x_T = torch.randn(n, 4, 64, 64, device=dev, generator=torch.Generator(device=dev).manual_seed(1000003 * idx + rank + 7))
        last["out"] = sampler.sample(S=len(SD_CAND), batch_size=n, shape=[4, 64, 64], conditioning=c, verbose=False, eta=0.0, x_T=x_T,
                                     unconditional_guidance_scale=7.5, unconditional_conditioning=uc, sampled_timestep=SD_CAND)[0]
    for w in range(max(1, args.warmup)):
        one_step(-1 - w)
    torch.cuda.synchronize()
    assert torch.isfinite(last["out"]).all()
    elapsed_local, elapsed = timed(ctx, one_step, args.steps)
    out = last["out"]
    rankinfo = gather_ranks(ctx, elapsed_local)
    # the latents of the last timed step (outside the timed region): finite and not constant
    lat_ok = bool(torch.isfinite(out).all().item())
    lat_std = float(out.float().std().item())
    chk = {"finite": lat_ok, "latent_shape": list(out.shape), "latent_mean": round(float(out.float().mean().item()), 5),
           "latent_std": round(lat_std, 5), "how": "the last timed step's latents, checked after the timed region"}
    if not lat_ok or not lat_std > 0.0:
        print(json.dumps({"error": "bench.py --workload sd: non-finite or constant latents", "output_check": chk}), flush=True)
        raise SystemExit(3)
    value = world * n * args.steps / elapsed
    model_tflops = value * 2 * len(SD_CAND) * SD_GFLOP_LATENT / 1e3
    roof = None
    if not args.no_kernel_events:  # per-launch HIP events need the eager path: one untimed evaluation outside the graph
        unet.enable_graph(False)
        ops.CONV_PROFILE, ops.CONV_PROFILE_KEY = [], None
        x = torch.randn(2 * n, 4, 64, 64, device=dev)
        t0e = time.perf_counter()
        unet(x, torch.full((2 * n,), 500, device=dev, dtype=torch.int64), torch.cat([uc, c]))
        torch.cuda.synchronize()
        eager_ms = (time.perf_counter() - t0e) * 1e3
        prof, ops.CONV_PROFILE = ops.CONV_PROFILE, None
        if prof:
            # the roofline kernel is the conv launch class with the largest share of this evaluation's conv time -- chosen from
            # the mix that actually ran, not assumed (round 3 named the 3x3 tile, 5.7 % of the kernel time)
            mix, conv_ms = launch_mix(prof)
            top = tuple(mix[0]["key"])
            dom = [p for p in prof if p[3] == top]
            tr, src = pmc_traffic("sd") if (n == 6 and top == (6, 9, True, 2)) else (None, None)
            roof, ms = roof_from(dom, conv_class_name(top), tr, src)
            roof.update({"share_of_conv_time": mix[0]["share_of_conv_time"], "launch_mix": mix,
                         "conv_ms_per_evaluation": round(conv_ms, 2), "eager_evaluation_ms": round(eager_ms, 2),
                         "whole_step": {"model_tflops": round(model_tflops, 1), "frac": round(model_tflops / PEAK_BF16_TFLOPS, 4),
                                        "what": "803.27 GFLOP per latent and evaluation x 12 evaluations per finished latent / measured step time"},
                         "how": f"one eager evaluation of {2 * n} latents after the timed region with HIP events on every conv launch (the timed steps "
                                "replay hipGraphs, which carry no events); attention / LayerNorm / GEGLU launches are not conv launches: "
                                "profiles/r04/bench_sd_kernel_stats.csv has the whole mix"})
    if rank != 0:
        return None
    cpu = None
    if not args.no_cpu_baseline and world == 1:
        del unet, sampler
        torch.cuda.empty_cache()
        cpu = cpu_baseline(workload="sd")
    line = {
        "metric": "latents/sec (node), Stable-Diffusion v1 latent UNet, 6-step searched DDIM, guidance 7.5",
        "value": round(value, 2), "unit": "latents/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "bf16" if args.torso == "bf16" else "f16",
        "data": "synthetic (x_T ~ N(0,1), conditioning ~ N(0,1) [N,77,768], random-init weights of the SD v1 UNet architecture)",
        "config": {"workload": f"Stable-Diffusion v1 latent UNet (859.5 M), searched DDIM {SD_CAND}, classifier-free guidance 7.5, "
                               f"{n} latents per GPU and step (64x64x4), {args.torso}, hipGraph replay; VAE / CLIP / FID not on this path",
                   "global_batch": world * n, "latent_size": 64, "sampler_steps": len(SD_CAND),
                   "parallelism": f"dp{world} (latent-sharded, no data-path collective)"},
        "model_tflops": round(model_tflops, 1),
        "roofline": roof, "output_check": chk, "parity": parity_block("sd", args.torso), "hbm_peak_gb": hbm_peak_gb(dev), "cpu_baseline": cpu}
    if rankinfo:
        line.update(rankinfo)
    return line


CAND_SEARCH_FLAGS = dict(batch_size=100, num_samples=5000)   # search_imagenet64_classifier_guidance.sh:2
CAND_LIST = [[153, 424, 926, 690], [85, 305, 572, 856], [137, 441, 647, 971], [62, 333, 690, 902],
             [201, 424, 744, 926], [17, 260, 519, 803]]        # 4-step candidates (the first = the headline schedule)


def synthetic_ref_stats(dim):
    import numpy as np
    from autodiffusion_amd.fid import FIDStatistics
    rng = np.random.RandomState(0)
    a = rng.randn(dim, dim) / 45.0
    return FIDStatistics(rng.randn(dim) * 0.1, a @ a.T + 0.1 * np.eye(dim))


def build_guided(flags, size, depth, torso, clf_torso, dev):
    """(model, diffusion, classifier) of an ADM-G configuration, random-init weights.  Any failure raises: a guided workload
    never degrades to an unguided one."""
    from autodiffusion_amd.script_util import (args_to_dict, classifier_defaults, create_classifier,
                                               create_model_and_diffusion, model_and_diffusion_defaults)
    model, diffusion = create_model_and_diffusion(**args_to_dict(argparse.Namespace(**flags), model_and_diffusion_defaults().keys()))
    model.to(dev).randomize_(1234).convert_to_fp16()
    model.set_torso(torso)
    cf = classifier_defaults()
    cf.update(image_size=size, classifier_depth=depth)   # configs/128_guided_sample.sh:2 / search_imagenet64...sh:6
    classifier = create_classifier(**cf)
    classifier.to(dev).randomize_(4321)
    classifier.set_torso(clf_torso)
    if not hasattr(classifier, "log_prob_grad"):
        raise SystemExit("bench.py: the classifier has no HIP guidance-gradient path (log_prob_grad)")
    return model, diffusion, classifier


def run_candidate(args, ctx):
    """One step = ONE WHOLE candidate evaluation, `EvolutionSearcher.get_cand_fid(cand, args)` at the reference's own search
    flags (search_imagenet64_classifier_guidance.sh:1-20, search_imagenet64_classifier_guidance.py:308-376): reset_diffusion,
    5000 classifier-guided ADM-G-64 images in batches of 100 (hipGraph replay: at this batch the host's ~60 ms of launch
    work per guided step is the floor otherwise), HIP Inception-v3 pool3 of every uint8 batch (random weights: the
    checkpoint is not in the image), float64 Gram on the matrix cores, and the Frechet distance on the device
    (`--fid_on_device`) against synthetic reference statistics.  N > 1: every rank evaluates its own candidate
    (population-parallel, statistics local, no data-path collective): weak scaling, value = candidates of all ranks / time."""
    import types
    import numpy as np
    from autodiffusion_amd import logger, ops
    from autodiffusion_amd.inception import pool3_features
    from autodiffusion_amd.search import EvolutionSearcher
    rank, world, dev = ctx.rank, ctx.world, ctx.dev
    bs = args.batch or CAND_SEARCH_FLAGS["batch_size"]
    nimg = args.images or CAND_SEARCH_FLAGS["num_samples"]
    model, diffusion, classifier = build_guided(adm64_flags(class_cond=True), 64, 4, args.torso, args.classifier_torso, dev)
    features, dim = pool3_features(dev, "", "tf1", allow_random=True)
    ref = synthetic_ref_stats(dim)
    use_graph = not args.no_graph and bs <= 128
    sargs = types.SimpleNamespace(max_epochs=1, select_num=10, population_num=50, m_prob=0.25, crossover_num=15, mutation_num=25,
                                  batch_size=bs, num_samples=nimg, image_size=64, use_ddim=True, clip_denoised=True,
                                  class_cond=True, classifier_scale=1.0, seed=0, time_step=4, use_ddim_init_x=True,
                                  fid_on_device=True, use_graph=use_graph, merge_batches=args.merge_batches)
    _log, logger.log = logger.log, (lambda *a_, **k_: None)   # the reference's per-batch "created N samples" lines belong to a search's log.txt
    searcher = EvolutionSearcher(sargs, model, diffusion, 4, classifier=classifier, features=features, feature_dim=dim,
                                 ref_stats=ref, population_parallel=True)
    splits, fids = [], []

    def one_step(idx):
        cand = CAND_LIST[(idx * world + rank) % len(CAND_LIST)]
        fids.append(searcher.get_cand_fid(cand=cand, args=sargs))
        splits.append(dict(searcher.last_times))
    for w in range(args.warmup):   # a warm-up candidate of a few batches: graph capture, kernel attributes, allocator pools
        sargs.num_samples = min(nimg, 3 * bs)
        one_step(-1 - w)
        sargs.num_samples = nimg
    del splits[:], fids[:]
    elapsed_local, elapsed = timed(ctx, one_step, args.steps)
    if not all(np.isfinite(fids)):
        print(json.dumps({"error": "bench.py --workload candidate: non-finite FID", "fids": [float(f) for f in fids]}), flush=True)
        raise SystemExit(3)
    rankinfo = gather_ranks(ctx, elapsed_local)
    # output check + roofline of the dominant kernel at this batch: one eager batch after the timed region
    ev = searcher._ev
    model.enable_graph(False)
    classifier.enable_graph(False)
    ev.set_candidate(CAND_LIST[0])
    roof = None
    if not args.no_kernel_events:
        ops.CONV_PROFILE, ops.CONV_PROFILE_KEY = [], (5, 9, True, 2)
    t0e = time.perf_counter()
    u8, sample = ev.sample_batch(bs, seed=12345 + rank, return_float=True)
    torch.cuda.synchronize()
    eager_s = time.perf_counter() - t0e
    if ops.CONV_PROFILE is not None:
        prof, ops.CONV_PROFILE, ops.CONV_PROFILE_KEY = ops.CONV_PROFILE, None, None
        if prof:
            tr, src = pmc_traffic("guided")
            roof, _ = roof_from(prof, CONV_SYMBOLS[(5, 9, True, 2)], None if tr is None else round(tr * bs / 256.0),
                                None if tr is None else src + f"; measured at batch 256 and scaled by {bs}/256 (the kernel's traffic is linear in the batch)")
            roof["how"] = (f"one eager batch of {bs} after the timed region (the timed candidates replay hipGraphs, which carry no events), "
                           f"{eager_s * 1e3:.0f} ms with per-launch events")
    chk = check_output(sample, u8)
    logger.log = _log
    if rank != 0:
        return None
    ncand = world * args.steps
    per_cand = elapsed / args.steps
    mean = lambda k: float(np.mean([t_[k] for t_ in splits]))   # noqa: E731
    gflop_img = len(CAND_LIST[0]) * (GFLOP_UNET + GFLOP_GUIDE)
    from autodiffusion_amd.evaluate import merge_policy
    mb = merge_policy(64, bs, args.merge_batches)[0]
    out = {
        "metric": "candidates/hour (node): one whole get_cand_fid (ADM-G ImageNet-64, 4-step guided DDIM, 5000 images, Inception pool3 + FID)",
        "value": round(ncand / elapsed * 3600.0, 1), "unit": "candidates/hour", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(per_cand * 1e3, 1), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "bf16" if args.torso == "bf16" else "f16",
        "data": "synthetic (x_T ~ N(0,1), y ~ U{0..999}, random-init weights of the ADM-G-64, classifier and Inception-v3 architectures, "
                "synthetic reference statistics): the FID VALUES mean nothing, the work is the real candidate's",
        "config": {"workload": f"get_cand_fid at the reference's search flags (search_imagenet64_classifier_guidance.sh: batch_size {bs}, "
                               f"num_samples {nimg}, 4-step candidates, classifier_scale 1.0), {mb} batches per pass (bitwise the same images), {args.torso}, "
                               f"{'hipGraph replay' if use_graph else 'eager launches'}, HIP Inception pool3 + f64 Gram + on-device Frechet distance",
                   "global_batch": world * bs, "image_size": 64, "sampler_steps": 4, "images_per_candidate": nimg,
                   "batches_per_pass": mb,
                   "parallelism": f"dp{world} (population-parallel: one whole candidate per rank and step, no data-path collective)",
                   "launch": "hipGraph replay" if use_graph else "eager"},
        "images_per_sec": round(ncand * nimg / elapsed, 1),
        "model_tflops": round(ncand * nimg / elapsed * gflop_img / 1e3, 1),
        "time_split_s": {"reset_time": round(mean("reset_time"), 4), "sample_time": round(mean("sample_time"), 3),
                         "fid_time": round(mean("fid_time"), 3), "per_candidate": round(per_cand, 3),
                         "note": "rank 0's mean over the timed candidates, the reference's own three timers "
                                 "(search_imagenet64_classifier_guidance.py:311, 366, 374); sample_time includes the Inception + Gram "
                                 "launches queued behind each batch, fid_time = pooled statistics + two f64 eigh + the scalar's D2H"},
        "fid_values": [round(float(f), 4) for f in fids],
        "roofline": roof, "output_check": chk, "parity": parity_block("adm64", args.torso), "hbm_peak_gb": hbm_peak_gb(dev),
    }
    if rankinfo:
        out.update(rankinfo)
    if not args.no_cpu_baseline and world == 1:
        out["cpu_baseline"] = cpu_baseline()
        v = out["cpu_baseline"]["value"]
        out["cpu_baseline"]["note"] = (f"sampling only, unguided, BASELINE config 1 (images/sec); a {nimg}-image guided candidate at that rate "
                                       f"is > {nimg / max(v, 1e-9) / 3600.0:.1f} host-hours")
    return out


POP_FLAGS = dict(population=64, images=64, batch=32, time_step=10)   # BASELINE configs[2]; num_samples <= 1000 per candidate (GD/README.md:22)


def run_population(args, ctx):
    """BASELINE configs[2] as written: one step = ONE EA EPOCH's population -- E = 64 ADM-G ImageNet-128 10-step candidates
    (configs/128_guided_sample.sh:1-3) -- generated with the reference's own `random` / `np.random` draw order on every rank
    (search_imagenet64_classifier_guidance.py:289-306: legality is the visited-set dedupe only, so generation never waits for a
    FID), assigned to ranks longest-processing-time-first (EvolutionSearcher.assign_candidates: 8 per GPU at N = 8), each
    evaluated WHOLE on its rank (reset_diffusion, classifier-guided sampling, HIP Inception pool3, f64 Gram, on-device Frechet
    distance), and ONE all_gather of the epoch's FIDs over the ranks (flush_pending; RCCL on a node).  The population is fixed as
    N grows: strong scaling; at N = 1 the line is the per-GPU unit of the split."""
    import random
    import types
    import numpy as np
    import torch.distributed as dist
    from autodiffusion_amd import logger
    from autodiffusion_amd.evaluate import graph_auto, merge_policy
    from autodiffusion_amd.inception import pool3_features
    from autodiffusion_amd.search import EvolutionSearcher
    rank, world, dev = ctx.rank, ctx.world, ctx.dev
    E = args.population or POP_FLAGS["population"]
    nimg = args.images or POP_FLAGS["images"]
    bs = args.batch or POP_FLAGS["batch"]
    K = args.sampler_steps or POP_FLAGS["time_step"]
    m64 = args.model == "adm64"
    size = 64 if m64 else 128
    model, diffusion, classifier = build_guided(adm64_flags(True) if m64 else adm128_flags(), size, 4 if m64 else 2,
                                                args.torso, args.classifier_torso, dev)
    features, dim = pool3_features(dev, "", "tf1", allow_random=True)
    merge, per_pass = merge_policy(size, bs, args.merge_batches, rounds=-(-nimg // bs))
    use_graph = not args.no_graph and graph_auto(size, per_pass)
    sargs = types.SimpleNamespace(max_epochs=1, select_num=10, population_num=E, m_prob=0.25, crossover_num=15, mutation_num=25,
                                  batch_size=bs, num_samples=nimg, image_size=size, use_ddim=True, clip_denoised=True,
                                  class_cond=True, classifier_scale=1.0, seed=0, time_step=K, use_ddim_init_x=False,
                                  fid_on_device=True, use_graph=use_graph, merge_batches=args.merge_batches)
    _log, logger.log = logger.log, (lambda *a_, **k_: None)
    searcher = EvolutionSearcher(sargs, model, diffusion, K, classifier=classifier, features=features, feature_dim=dim,
                                 ref_stats=synthetic_ref_stats(dim), population_parallel=True)
    flushes = []

    def epoch(idx, count):
        # every rank draws the same candidates: the seeds do not depend on the rank
        random.seed(7 + idx)
        np.random.seed(7 + idx)
        searcher.vis_dict, searcher.candidates = {}, []
        searcher.get_random_before_search(count)      # queues `count` candidates (population_parallel: nothing is evaluated yet)
        searcher.flush_pending()                      # LPT assignment, whole-candidate evaluation, one all_gather of the FIDs
        fids = [searcher.vis_dict[c]["fid"] for c in searcher.candidates]
        flushes.append(dict(searcher.last_flush, fids=fids))
    for w in range(max(1, args.warmup)):   # one candidate per rank: graph capture, kernel attributes, allocator pools
        epoch(-1 - w, world)
    del flushes[:]
    elapsed_local, elapsed = timed(ctx, lambda s_: epoch(s_, E), args.steps)
    logger.log = _log
    fl = flushes[-1]
    if not all(np.isfinite(f["fids"]).all() for f in flushes):
        print(json.dumps({"error": "bench.py --workload population: non-finite FID"}), flush=True)
        raise SystemExit(3)
    mine = {"assigned": int(np.mean([f["assigned"] for f in flushes])), "assigned_cost": int(np.mean([f["assigned_cost"] for f in flushes])),
            "evaluate_s": round(float(np.mean([f["evaluate_s"] for f in flushes])), 3)}
    rankinfo = gather_ranks(ctx, elapsed_local, extra=mine)
    if rank != 0:
        return None
    ncand = E * args.steps
    gflop_img = K * ((GFLOP_UNET + GFLOP_GUIDE) if m64 else (GFLOP_UNET_128 + GFLOP_GUIDE_128))
    out = {
        "metric": f"candidates/hour (node): EA population of {E} ADM-G ImageNet-{size} {K}-step candidates, whole-candidate evaluation per GPU",
        "value": round(ncand / elapsed * 3600.0, 1), "unit": "candidates/hour", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 1), "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "bf16" if args.torso == "bf16" else "f16",
        "data": f"synthetic (x_T ~ N(0,1), y ~ U{{0..999}}, random-init weights of the ADM-G-{size}, classifier and Inception-v3 architectures, "
                "synthetic reference statistics): the FID VALUES mean nothing, the work is the real population's",
        "config": {"workload": f"one EA epoch = population of {E} {K}-step candidates (random timestep subsets drawn by the search's own operator), "
                               f"{nimg} images per candidate in batches of {bs} ({merge} per pass), ADM-G ImageNet-{size} classifier-guided DDIM, {args.torso}, "
                               f"{'hipGraph replay' if use_graph else 'eager launches'}; HIP Inception pool3 + f64 Gram + on-device Frechet distance per candidate; "
                               f"LPT assignment over {world} rank(s), one all_gather of the {E} FIDs per epoch",
                   "population": E, "images_per_candidate": nimg, "global_batch": world * bs, "image_size": size, "sampler_steps": K,
                   "candidates_per_rank": [sum(1 for o in fl["owner"] if o == r) for r in range(world)],
                   "cost_per_rank": [sum(c for c, o in zip(fl["costs"], fl["owner"]) if o == r) for r in range(world)],
                   "parallelism": f"population-parallel over {world} rank(s): whole candidates per rank, no data-path collective inside a candidate",
                   "launch": "hipGraph replay" if use_graph else "eager"},
        "images_per_sec": round(ncand * nimg / elapsed, 1),
        "model_tflops": round(ncand * nimg / elapsed * gflop_img / 1e3, 1),
        "epoch_collective": fl["collective"],
        "fid_values_last_epoch": [round(float(f), 3) for f in fl["fids"][:8]] + (["..."] if E > 8 else []),
        "parity": parity_block("adm64" if m64 else "adm128", args.torso), "hbm_peak_gb": hbm_peak_gb(dev),
    }
    if not ctx.dist_on:
        out["per_rank"] = [mine]
    if rankinfo:
        out.update(rankinfo)
    return out


def run_images(args, ctx):
    """The image workloads: guided (auto) / unguided ADM-64, adm128, adm256.  One step = one batch (x --merge-batches) sampled."""
    import torch.distributed as dist  # noqa: F401
    from autodiffusion_amd import ops
    from autodiffusion_amd.evaluate import CandidateEvaluator
    from autodiffusion_amd.schedule import space_timesteps
    from autodiffusion_amd.script_util import args_to_dict, create_model_and_diffusion, model_and_diffusion_defaults
    rank, world, dev = ctx.rank, ctx.world, ctx.dev
    w256, w128 = args.workload == "adm256", args.workload == "adm128"
    B = args.batch
    if B is None:
        # 256x256: 64 images per step = 0.7 s of GPU time; the reference's own launch flag is 36 (search_lsun_cat.sh:2) and
        # throughput is flat beyond (every launch already fills the chip), so "as large as fits" would only stretch the run;
        # 128x128: the reference's launch flag (configs/128_guided_sample.sh:3)
        B = 64 if w256 else (32 if w128 else 256)
    guided = args.workload in ("auto", "guided", "adm128")
    flags = adm256_flags() if w256 else (adm128_flags() if w128 else adm64_flags(class_cond=True))
    if w256 and args.class_cond:
        flags["class_cond"] = True     # BASELINE configs[4] as SURVEY 8(d) writes it: search_lsun_cat.sh:1 + class_cond (554 M)
    size = 256 if w256 else (128 if w128 else 64)
    # adm128: the uniform 10-step grid, the start candidate of a `--time_step 10` search (the candidate's cost does not depend
    # on which timesteps it holds)
    schedule = SCHEDULE_256 if w256 else (sorted(space_timesteps(1000, "ddim10")) if w128 else SCHEDULE)
    if args.sampler_steps:
        schedule = sorted(space_timesteps(1000, f"ddim{args.sampler_steps}"))
    gflop_unet = GFLOP_UNET_256 if w256 else (GFLOP_UNET_128 if w128 else GFLOP_UNET)
    gflop_guide = GFLOP_GUIDE_128 if w128 else GFLOP_GUIDE
    # the launch mix's dominant conv symbol: (tiling variant, taps, map > 8x8, prologue) -- 192-wide tiles for ADM-64's
    # multiples of 192 channels, 128-wide tiles for the 256-multiples of ADM-128 / LSUN-256
    dom_key = (6, 9, True, 2) if (w256 or w128) else (5, 9, True, 2)
    dom_name = CONV_SYMBOLS[dom_key]
    classifier = None
    if guided:   # any failure here raises: the headline never degrades to an unguided run
        model, diffusion, classifier = build_guided(flags, size, 2 if w128 else 4, args.torso, args.classifier_torso, dev)
    else:
        model, diffusion = create_model_and_diffusion(**args_to_dict(argparse.Namespace(**flags), model_and_diffusion_defaults().keys()))
        model.to(dev).randomize_(1234).convert_to_fp16()
        model.set_torso(args.torso)
    class_cond = (not w256) or bool(args.class_cond)
    ev = CandidateEvaluator(model, diffusion, classifier=classifier, image_size=size, use_ddim=True,
                            classifier_scale=1.0, class_cond=class_cond, device=dev, use_graph=args.graph)
    skip_layers = None
    if args.skip_layers:
        if not getattr(model.plan, "dynamic", False):
            raise SystemExit("--skip-layers needs a dynamic-UNet workload (adm256)")
        skip_layers = (auto_skip_layers(model.layer_num, len(schedule)) if args.skip_layers == "auto"
                       else json.loads(args.skip_layers))
        if len(skip_layers) != len(schedule):
            raise SystemExit(f"--skip-layers: {len(skip_layers)} lists for {len(schedule)} steps")
        ev.set_candidate({"timesteps": list(schedule), "skip_layers": skip_layers})
    else:
        ev.set_candidate(schedule)

    fid_net = fid_acc = None
    if args.with_fid:
        from autodiffusion_amd.fid import ActivationAccumulator
        from autodiffusion_amd.inception import InceptionV3
        fid_net = InceptionV3().to(dev)
        fid_net.weights_loaded = True     # random weights on purpose (throughput run): no warning
        fid_acc = ActivationAccumulator(2048, dev)

    MB = max(1, args.merge_batches)   # reference batches per pass over the networks (bitwise the same images: sample_batches)

    def one_step(step_idx, return_float=False):
        # deterministic, layout-independent seeding per (step, rank)
        if MB > 1 and not return_float:
            u8s = ev.sample_batches(B, [1000003 * step_idx + rank + 7919 * j for j in range(MB)])
            if fid_net is not None:
                for u8 in u8s:
                    fid_acc.add_from(fid_net.features, u8)
            return u8s[0]
        res = ev.sample_batch(B, seed=(1000003 * step_idx + rank), return_float=return_float)
        u8 = res[0] if return_float else res
        if fid_net is not None:
            fid_acc.add_from(fid_net.features, u8)
        return res

    for w in range(args.warmup):
        one_step(-1 - w)
    torch.cuda.synchronize()
    events_live = not args.no_kernel_events and not args.graph
    if events_live:
        ops.CONV_PROFILE = []
        # HIP events only around the dominant kernel's launches (all conv launches with --conv-breakdown): a pair of
        # events per launch costs the unguided workload 3.5 % when every conv carries one
        ops.CONV_PROFILE_KEY = None if args.conv_breakdown else dom_key
    elapsed_local, elapsed = timed(ctx, one_step, args.steps)
    prof_timed, ops.CONV_PROFILE = ops.CONV_PROFILE, None
    fid_pooled = None
    if fid_acc is not None and w256:
        # BASELINE config 5's pooled-FID stage: ONE all-gather of the packed float64 (n, sum a, sum a a^T) over the ranks (RCCL over
        # xGMI; skipped without a process group) + the Frechet distance on the device; timed on its own, outside the sampling steps
        ref = synthetic_ref_stats(2048)
        torch.cuda.synchronize()
        tf0 = time.perf_counter()
        fidv = fid_acc.frechet_distance_device(ref)
        torch.cuda.synchronize()
        fid_pooled = {"images_pooled": world * B * MB * (args.steps + args.warmup), "seconds": round(time.perf_counter() - tf0, 4),
                      "finite": bool(fidv == fidv and abs(fidv) != float("inf")), "collective": fid_acc.last_collective,
                      "what": "packed f64 all-gather of (n, s1, s2) over the ranks + on-device Frechet distance (two f64 eigh), once per candidate"}
    rankinfo = gather_ranks(ctx, elapsed_local)

    roof = None
    eager_s = None
    prof = prof_timed
    if args.graph and not args.no_kernel_events:   # the replayed launches carry no events: one eager batch for the roofline
        model.enable_graph(False)
        if classifier is not None:
            classifier.enable_graph(False)
        ops.CONV_PROFILE, ops.CONV_PROFILE_KEY = [], (None if args.conv_breakdown else dom_key)
        t0e = time.perf_counter()
        one_step(args.steps + 1)
        torch.cuda.synchronize()
        eager_s = time.perf_counter() - t0e
        prof, ops.CONV_PROFILE = ops.CONV_PROFILE, None
    if prof is not None:
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
            traffic, traffic_src = pmc_traffic(args.workload if args.workload != "auto" else "guided")
            if traffic is not None and B * MB != {"adm256": 64, "adm128": 32}.get(args.workload, 256):
                traffic, traffic_src = None, None   # the committed figure is for the default batch
            roof, ms = roof_from(dom, dom_name, traffic, traffic_src)
            roof["share_of_step_time"] = round(ms * 1e-3 / (eager_s if args.graph else elapsed), 3)
            if args.graph:
                roof["how"] = "one eager batch after the timed region (the timed steps replay hipGraphs, which carry no events)"
            else:
                # what the event pairs cost the number they sit in: the same steps once more without them, untimed
                ne = min(3, args.steps)
                torch.cuda.synchronize()
                t0n = time.perf_counter()
                for s_ in range(ne):
                    one_step(args.steps + 10 + s_)
                torch.cuda.synchronize()
                plain_ms = (time.perf_counter() - t0n) / ne * 1e3
                timed_ms = elapsed / args.steps * 1e3
                roof["events_in_timed_region"] = {
                    "pairs_per_step": len(dom) // args.steps, "ms_per_step_with_events": round(timed_ms, 2),
                    "ms_per_step_without_events": round(plain_ms, 2), "cost_pct": round((timed_ms - plain_ms) / timed_ms * 100.0, 2),
                    "how": f"{ne} more steps right after the timed region with no events recorded (same seeds schedule, this rank); the pairs "
                           "bracket the dominant kernel's launches only (every conv launch carrying one was measured at 3.5 %)"}
            if guided and ev.active_diffusion.overlap_guidance:
                # in the timed region the guidance gradient runs on a second stream, so the launches above share the CUs
                # with its kernels (their event-bracketed time is not the kernel's own speed): one extra, untimed batch
                # with the two networks in sequence gives the kernel's duration when it owns the chip
                ev.active_diffusion.overlap_guidance = False
                ops.CONV_PROFILE, ops.CONV_PROFILE_KEY = [], dom_key
                one_step(args.steps)
                torch.cuda.synchronize()
                iso = [p for p in ops.CONV_PROFILE if p[3] == dom_key]
                ops.CONV_PROFILE = None
                ev.active_diffusion.overlap_guidance = True
                ims = sum(p[0].elapsed_time(p[1]) for p in iso)
                ifl = sum(p[2] for p in iso)
                roof["concurrency"] = ev.active_diffusion.describe_overlap() if hasattr(ev.active_diffusion, "describe_overlap") else \
                    "timed region: UNet and classifier-guidance kernels overlap on two HIP streams"
                roof["isolated"] = {"achieved": round(ifl / (ims * 1e-3) / 1e12, 2),
                                    "frac": round(ifl / (ims * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4),
                                    "avg_launch_us": round(ims * 1e3 / len(iso), 2), "launches": len(iso),
                                    "how": "one untimed batch after the timed region, networks in sequence on one stream"}
    ops.CONV_PROFILE = None
    # what the timed loop produces, verified on one more batch outside the timed region
    u8c, samplec = one_step(args.steps + 2, return_float=True)
    torch.cuda.synchronize()
    chk = check_output(samplec, u8c)
    if rank != 0:
        return None
    imgs = world * B * MB * args.steps
    value = imgs / elapsed
    gflop_img = len(schedule) * (gflop_unet + (gflop_guide if guided else 0.0))
    if w256:
        wl = (f"ADM LSUN-256 dynamic UNet ({'554 M, class-conditional' if args.class_cond else '552.8 M, unconditional'}), uniform {len(schedule)}-step DDIM {schedule}, "
              f"batch={B} per GPU, {args.torso}" + (f", layer-skip lists {skip_layers}" if skip_layers else ""))
    elif w128:
        wl = (f"ADM-G ImageNet-128 (421.5 M UNet, 128x128 depth-2 classifier) classifier-guided, {len(schedule)}-step DDIM "
              f"{schedule}, batch={B} per GPU, {args.torso}")
    else:
        wl = (("ADM-G ImageNet-64 classifier-guided" if guided else
               "ADM ImageNet-64 class-conditional, UNGUIDED (classifier guidance not in this run)")
              + f", searched 4-step DDIM {schedule}, batch={B} per GPU, {args.torso}")
    out = {
        "metric": (f"images/sec (node), ADM LSUN-256 {len(schedule)}-step DDIM" if w256 else
                   (f"images/sec (node), ADM-G ImageNet-128 {len(schedule)}-step DDIM" if w128 else
                    "images/sec (node), ADM-G ImageNet-64 4-step DDIM")),
        "value": round(value, 2), "unit": "images/sec", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 2),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16" if args.torso == "bf16" else "f16",
        "data": ("synthetic (x_T ~ N(0,1), random-init weights of the ADM LSUN-256 architecture)" if w256 else
                 f"synthetic (x_T ~ N(0,1), y ~ U{{0..999}}, random-init weights of the ADM-G-{size} architecture)"),
        "config": {"workload": wl + (f"; {MB} such batches per pass over the networks (bitwise the same images), i.e. {MB * B} images per step and GPU" if MB > 1 else ""),
                   "global_batch": world * B * MB, "batches_per_pass": MB, "image_size": size, "sampler_steps": len(schedule),
                   "parallelism": f"dp{world} (image-sharded, no data-path collective)",
                   "launch": "hipGraph replay" if args.graph else "eager",
                   "fid_stage_in_step": bool(args.with_fid)},
        "model_tflops": round(value * gflop_img / 1e3, 1),
        "roofline": roof, "output_check": chk,
        "parity": parity_block("adm256" if w256 else ("adm128" if w128 else "adm64"), args.torso),
        "hbm_peak_gb": hbm_peak_gb(dev),
    }
    if roof is not None:
        roof["whole_step"] = {"model_tflops": out["model_tflops"], "frac": round(out["model_tflops"] / PEAK_BF16_TFLOPS, 4)}
    if skip_layers:
        skipped = sum(len(s_) for s_ in skip_layers)
        out["config"]["skip_layers"] = skip_layers if len(skip_layers) <= 6 else f"{len(skip_layers)} lists, {len({tuple(sorted(s_)) for s_ in skip_layers})} distinct"
        out["model_tflops"] = None   # the per-layer FLOPs of the skipped layers are not tabulated: no TFLOP/s claim
        if roof is not None:
            roof["whole_step"] = None
        out["config"]["layers_evaluated"] = f"{len(schedule) * model.layer_num - skipped} of {len(schedule) * model.layer_num}"
    if args.graph:
        out["graphs"] = model.graph_report() if hasattr(model, "graph_report") else None
    if fid_pooled:
        out["fid_pooled"] = fid_pooled
    if rankinfo:
        out.update(rankinfo)
    if not args.no_cpu_baseline and world == 1:
        out["cpu_baseline"] = cpu_baseline()
        if w256 or w128:
            out["cpu_baseline"]["note"] = (f"timed on BASELINE config 1 (64x64): one {size}x{size} evaluation is {gflop_unet:.1f} GFLOP, "
                                           f"{gflop_unet / GFLOP_UNET:.1f} x a 64x64 one, too long for a bounded sample of this workload")
    return out


RUNNERS = {"sd": run_sd, "candidate": run_candidate, "population": run_population}
SECONDARY_DEFAULT = "adm256,sd,adm128,population"
SECONDARY_STEPS = {"adm256": (2, 1), "sd": (5, 1), "adm128": (2, 1), "population": (1, 1)}   # (steps, warmup) of a secondary line


def compact(line):
    """A secondary line: the same keys minus the long prose (data / how / notes) and the CPU leg."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "scaling", "dtype", "model_tflops", "images_per_sec",
            "hbm_peak_gb", "fid_pooled", "epoch_collective", "per_rank", "ranks", "collective_check")
    out = {k: line[k] for k in keep if k in line and line[k] is not None}
    out["workload"] = line["config"]["workload"]
    for k in ("candidates_per_rank", "cost_per_rank"):
        if k in line["config"]:
            out[k] = line["config"][k]
    r = line.get("roofline")
    if r:
        out["roofline"] = {k: r[k] for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "launches", "avg_launch_us", "avg_launch_gflop",
                                             "share_of_step_time", "share_of_conv_time", "whole_step") if k in r}
    c = line.get("output_check")
    if c:
        out["output_check"] = {k: v for k, v in c.items() if k != "how"}
    if "ranks" in out:
        out["ranks"] = [{k: r_.get(k) for k in ("rank", "elapsed_s", "assigned", "assigned_cost", "evaluate_s") if k in r_} for r_ in out["ranks"]]
    return out


def self_launch(n, argv):
    """`python bench.py --gpus N` with no launcher around it: start N ranks as `python -m torch.distributed.run ...` in a CHILD
    process (never exec, and before this process has touched the GPU: nothing here calls into HIP), let rank 0's JSON line
    through on the inherited stdout, and return the child's exit code."""
    import socket
    import subprocess
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on this pool: RCCL's intra-node setup needs it
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    print(f"bench.py: WORLD_SIZE unset and --gpus {n}: launching {n} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default 3; population: 1 epoch)")
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=None, help="images per step and GPU (default 256; adm128: 32; adm256: 64; "
                                                            "--workload sd: 6 latents; candidate: 100 per sampling batch; population: 32)")
    ap.add_argument("--workload", default="auto", choices=["auto", "guided", "unguided", "adm128", "adm256", "sd", "candidate", "population"],
                    help="auto/guided = the headline (ADM-G ImageNet-64, BASELINE configs[1]); adm128 = ADM-G ImageNet-128 guided, "
                         "10-step candidate (one sampling batch); adm256 = ADM LSUN-256 dynamic UNet, uniform 5-step DDIM (the north "
                         "star's 256x256 line; + --skip-layers / --with-fid / --class-cond = configs[4]); sd = configs[3] (Stable-Diffusion v1 latent "
                         "UNet, 6 searched DDIM steps, classifier-free guidance 7.5); candidate = one whole get_cand_fid per step; "
                         "population = configs[2]: one EA epoch of 64 ADM-G-128 10-step candidates, LPT-sharded over the ranks")
    ap.add_argument("--secondary", default=None,
                    help=f"comma list of workloads run after the headline in the same process and attached to its line as `secondary` "
                         f"(default with --workload auto: {SECONDARY_DEFAULT}; 'none' to skip)")
    ap.add_argument("--skip-layers", default=None,
                    help="dynamic-UNet workloads (adm256): one layer-skip list per step, as JSON ('[[1,5],[],...]'), or 'auto' = "
                         "10 %% of the layers per step, seeded (the shape `--max_prun 0.1` candidates have)")
    ap.add_argument("--sampler-steps", type=int, default=0, help="adm128 / adm256 / population: the uniform ddimK grid instead of the default K")
    ap.add_argument("--class-cond", action="store_true", help="adm256: the class-conditional 554 M network (configs[4] as SURVEY 8(d) writes it)")
    ap.add_argument("--images", type=int, default=None, help="images per candidate (candidate: default 5000; population: 64)")
    ap.add_argument("--population", type=int, default=None, help="--workload population: candidates per epoch (default 64)")
    ap.add_argument("--model", default="adm128", choices=["adm128", "adm64"], help="--workload population: the candidates' model")
    ap.add_argument("--no-graph", action="store_true", help="candidate / population: eager launches instead of hipGraph replay")
    ap.add_argument("--merge-batches", type=int, default=0,
                    help="reference batches evaluated per pass over the networks (images bitwise the same either way).  candidate / population: "
                         "0 = auto (evaluate.merge_policy, what get_cand_fid does), 1 = one batch per pass as the reference does; image workloads "
                         "(e.g. adm128, batch 32): default 1 = the reference's launch unit, K = what a search on this GPU runs at")
    ap.add_argument("--torso", default="bf16", choices=["bf16", "fp16"],
                    help="16-bit element type of the UNet torso: bf16 (BASELINE configs[1] names it) or fp16 (the reference's own "
                         "torso type, libadm_hip_f16.so: same kernels, 11 mantissa bits)")
    ap.add_argument("--classifier-torso", default="bf16", choices=["bf16", "fp16"],
                    help="the guidance classifier's element type, forward AND backward network (fp16: d(logits) runs scaled by 2^10)")
    ap.add_argument("--graph", action="store_true",
                    help="replay the UNet evaluation and the guidance gradient as captured hipGraphs (small batches: the host's "
                         "~60 ms of launch work per guided step is the floor below batch ~100); the roofline's per-launch events "
                         "then come from one extra eager batch after the timed region")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl (= RCCL) for real multi-GPU runs; gloo only to rehearse N ranks on one GPU")
    ap.add_argument("--force-dist", action="store_true",
                    help="build the process group even with one rank and run every collective of the protocol (and of the pooled-FID / "
                         "population stages) through it: a world-size-1 nccl group exercises RCCL on a one-GPU box")
    ap.add_argument("--with-fid", action="store_true",
                    help="also run the FID stage inside every step: HIP Inception-v3 pool3 of the step's uint8 batch (random weights: "
                         "the checkpoint is not in the image) + the float64 Gram accumulation; the headline line leaves it out "
                         "(BASELINE's metric is sampling throughput)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true")
    ap.add_argument("--conv-breakdown", action="store_true",
                    help="print per-shape conv time / TFLOP/s (HIP events) to stderr")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # the driver's own command shape (`python3 bench.py --gpus N ...`): become the launcher.  Nothing above touched the GPU.
        raise SystemExit(self_launch(args.gpus, sys.argv[1:]))

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    if args.steps is None:
        args.steps = 1 if args.workload == "population" else 3
    ndev = torch.cuda.device_count()
    if args.dist_backend == "nccl" and world > 1 and ndev < local_world:
        raise SystemExit(f"bench.py: {local_world} ranks on this node but {ndev} visible GPU(s): RCCL takes one rank per device "
                         "(--dist-backend gloo rehearses several ranks on one GPU)")
    pin = pin_rank(local_rank, local_world)   # before any thread pool exists
    dev = torch.device(f"cuda:{local_rank % max(1, ndev)}")
    torch.cuda.set_device(dev)
    dist_on = world > 1 or args.force_dist
    if dist_on:
        import torch.distributed as dist
        # this pool's host driver supports dmabuf IPC only; with the legacy mode RCCL's intra-node transport setup fails with
        # `hipIpcGetMemHandle: invalid argument`
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.force_dist:
            os.environ["ADM_FORCE_COLLECTIVES"] = "1"
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29577")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if args.dist_backend == "nccl":
            dist.init_process_group(backend="nccl", init_method="env://", device_id=dev)
        else:
            dist.init_process_group(backend="gloo", init_method="env://")
    red_dev = dev if args.dist_backend == "nccl" else torch.device("cpu")
    ctx = Ctx(rank, world, local_rank, dev, red_dev, pin, dist_on, args.dist_backend)

    def run(a):
        torch.cuda.reset_peak_memory_stats(dev)
        return RUNNERS.get(a.workload, run_images)(a, ctx)

    line = run(args)
    names = args.secondary if args.secondary is not None else (SECONDARY_DEFAULT if args.workload == "auto" else "none")
    names = [] if names in ("none", "") else [n_ for n_ in names.split(",") if n_]
    if names:
        import gc
        sec = {}
        for name in names:
            if name not in ("adm128", "adm256", "sd", "candidate", "population", "guided", "unguided"):
                raise SystemExit(f"--secondary: unknown workload {name}")
            sub = argparse.Namespace(**vars(args))
            sub.workload, sub.batch, sub.images, sub.merge_batches, sub.graph, sub.with_fid = name, None, None, 0, False, False
            sub.skip_layers, sub.sampler_steps, sub.class_cond, sub.conv_breakdown, sub.no_cpu_baseline = None, 0, False, False, True
            sub.steps, sub.warmup = SECONDARY_STEPS.get(name, (2, 1))
            if name == "adm128":
                # what a search on this model launches: get_cand_fid merges the reference's memory-driven batches of 32 up to the
                # 128-image pass cap (evaluate.merge_policy; bitwise the same images) -- the line's workload string says so;
                # `--workload adm128` on its own times ONE reference batch per pass
                from autodiffusion_amd.evaluate import merge_policy
                sub.merge_batches = merge_policy(128, 32)[0]
            gc.collect()
            torch.cuda.empty_cache()
            t0 = time.perf_counter()
            try:
                res = run(sub)
                if rank == 0:
                    sec[name] = compact(res)
                    sec[name]["wall_s"] = round(time.perf_counter() - t0, 1)
            except (Exception, SystemExit) as e:   # a secondary line never takes the headline down; its failure is on the line
                if rank == 0:
                    sec[name] = {"error": f"{type(e).__name__}: {e}"}
                if world > 1:
                    raise
        if rank == 0:
            line["secondary"] = sec
    if rank == 0:
        print(json.dumps(line), flush=True)
    if dist_on:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
