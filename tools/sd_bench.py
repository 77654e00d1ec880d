#!/usr/bin/env python3
"""Time the full-size Stable-Diffusion v1 latent UNet (859.5 M parameters, 64x64 latents, 77 x 768 context) on the HIP
path: ms per UNet evaluation, latents/s and model TFLOP/s (803.3 GFLOP per latent per evaluation).
GRAPH=1 replays a captured hipGraph.  BATCH (default 12 = the reference's n_samples 6 with classifier-free guidance), REPS, BREAKDOWN=1 for per-shape conv time.
SAMPLER=ddim|plms|dpm additionally times BASELINE config 4's candidate evaluation: K searched steps (K, default 6),
classifier-free guidance 7.5, N_SAMPLES latents per batch (default 6) -> finished latents/s."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from autodiffusion_amd import ops  # noqa: E402
from autodiffusion_amd.sd_arch import SD_V1  # noqa: E402
from autodiffusion_amd.sd_unet import UNetModel  # noqa: E402

DEV = "cuda:0"
GFLOP_PER_LATENT = 803.27


def main():
    b = int(os.environ.get("BATCH", "12"))
    reps = int(os.environ.get("REPS", "5"))
    m = UNetModel(image_size=32, use_spatial_transformer=True, **SD_V1).to(DEV)
    m.randomize_(1234)
    if os.environ.get("GRAPH") == "1":
        m.enable_graph()
    x = torch.randn(b, 4, 64, 64, device=DEV)
    t = torch.full((b,), 500, device=DEV, dtype=torch.int64)
    ctx = torch.randn(b, 77, 768, device=DEV)
    for _ in range(2):
        out = m(x, t, ctx)
    torch.cuda.synchronize()
    assert torch.isfinite(out).all()
    if os.environ.get("BREAKDOWN") == "1":
        ops.CONV_PROFILE = []
    t0 = time.perf_counter()
    for _ in range(reps):
        m(x, t, ctx)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"SD v1 UNet batch {b}: {dt * 1e3:.1f} ms / evaluation, {b / dt:.1f} latents/s, "
          f"{b / dt * GFLOP_PER_LATENT / 1e3:.1f} model TFLOP/s")
    if os.environ.get("SAMPLER"):
        sampler_bench(m)
    if ops.CONV_PROFILE:
        prof, ops.CONV_PROFILE = ops.CONV_PROFILE, None
        agg = {}
        for e0, e1, f, key, shape in prof:
            a = agg.setdefault((key, shape), [0.0, 0.0, 0])
            a[0] += e0.elapsed_time(e1); a[1] += f; a[2] += 1
        tot = sum(v[0] for v in agg.values()) / reps
        print(f"conv launches: {tot:.1f} ms of {dt * 1e3:.1f} ms")
        for (key, shape), (ms_, fl_, cnt) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:25]:
            print(f"  conv {key} nhwc_in={shape[:4]} cout={shape[4]} x{cnt // reps}: {ms_ / reps:8.2f} ms {fl_ / ms_ / 1e9:7.1f} TFLOP/s")


def sampler_bench(m):
    from autodiffusion_amd.sd_sampler import DDIMSampler, DPMSolverSampler, LatentDiffusion, PLMSSampler
    kind = os.environ["SAMPLER"]
    k, n = int(os.environ.get("K", "6")), int(os.environ.get("N_SAMPLES", "6"))
    reps = int(os.environ.get("REPS", "5"))
    ld = LatentDiffusion(m, device=DEV)
    cand = sorted([94, 834, 217, 944, 574, 354, 153, 690, 424, 926][:k])
    if kind == "dpm":
        cand = sorted(cand + [3], reverse=True)  # K+1 time points
    sampler = {"ddim": DDIMSampler, "plms": PLMSSampler, "dpm": DPMSolverSampler}[kind](ld)
    c, uc = torch.randn(n, 77, 768, device=DEV), torch.randn(n, 77, 768, device=DEV)

    def run(seed):
        x_T = torch.randn(n, 4, 64, 64, device=DEV, generator=torch.Generator(device=DEV).manual_seed(seed))
        return sampler.sample(S=k, batch_size=n, shape=[4, 64, 64], conditioning=c, verbose=False, eta=0.0, x_T=x_T,
                              unconditional_guidance_scale=7.5, unconditional_conditioning=uc, sampled_timestep=cand)[0]
    for w in range(2):
        out = run(-1 - w)
    torch.cuda.synchronize()
    assert torch.isfinite(out).all()
    t0 = time.perf_counter()
    for r in range(reps):
        run(r)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    evals = k + (1 if kind == "plms" else 0)
    print(f"SD v1 {kind} K={k} cfg 7.5, {n} latents/batch: {dt * 1e3:.1f} ms / batch, {n / dt:.2f} finished latents/s, "
          f"{2 * n * evals / dt * GFLOP_PER_LATENT / 1e3:.1f} model TFLOP/s ({evals} guided UNet evaluations)")


if __name__ == "__main__":
    main()
