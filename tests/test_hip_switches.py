"""Every NON-DEFAULT kernel path that an `ADM_*` switch (or the per-model attribute behind it) selects, held to the same parity
bounds as the default path -- DESIGN.md section 7 lists the switches and names the test of each.

The switches exist for same-box A/B measurements; a path that only a builder-run tool exercised could rot unseen (round-2 review,
hygiene).  Python-level switches are class / module attributes initialised from the environment: they are toggled here directly.  The two
switches read once inside the library (`ADM_CONV_NO_RESIDENT`, `ADM_CG_NO_LDS`) run in a child process with the variable set.
"""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from helpers import golden
from test_hip_fullsize import DEV, adm64, clf, guided_loop, rel

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


def test_unet_up_resblock_paths_and_sequential_guidance(monkeypatch):
    """ADM_UPCONV_PHASES=0 (one 9-tap launch instead of the four 2x2-tap phase convs), ADM_NO_VIRTUAL_UP (materialised upsample +
    plain conv) on the full-size ADM-G-64 UNet against the reference's fp32 output; ADM_OVERLAP_GUIDANCE=0 (eps and the guidance
    gradient in sequence on one stream) bitwise equal to the two-stream loop."""
    g = golden("full_adm64")
    model, diffusion = adm64()
    x, t, y = (torch.from_numpy(g[k]).to(DEV) for k in ("x", "t", "y"))
    base = model(x, t, y)
    assert model.upconv_phases and rel(base, g["out"]) < 2e-2
    model.upconv_phases = False
    one = model(x, t, y)
    model.upconv_phases = True
    monkeypatch.setenv("ADM_NO_VIRTUAL_UP", "1")
    mat = model(x, t, y)
    monkeypatch.delenv("ADM_NO_VIRTUAL_UP")
    r1, r2 = rel(one, g["out"]), rel(mat, g["out"])
    print(f"up-ResBlock convs: phases {rel(base, g['out']):.3e}, one 9-tap launch {r1:.3e}, materialised upsample {r2:.3e}")
    assert r1 < 2e-2 and r2 < 2e-2 and not torch.equal(one, base)
    assert rel(one, mat.cpu().numpy()) < 6e-3          # same taps, statistics taken by a separate pass in the materialised path
    gl = golden("full_loop64")
    c64 = clf(64, 4)
    x_T, yl = torch.from_numpy(gl["x_T"]), torch.from_numpy(gl["y"])
    s2, u2 = guided_loop(model, diffusion, c64, gl["cand"].tolist(), x_T, yl)
    from autodiffusion_amd.sampler import SpacedDiffusion
    assert SpacedDiffusion.overlap_guidance
    monkeypatch.setattr(SpacedDiffusion, "overlap_guidance", False)
    s1, u1 = guided_loop(model, diffusion, c64, gl["cand"].tolist(), x_T, yl)
    assert torch.equal(s1, s2) and torch.equal(u1, u2)


def test_resblocks_without_the_skip_connection_fold():
    """ADM_FOLD_SKIP=0: skip_connection as its own 1x1 launch and the residual operand of the out_layers conv, instead of extra one-tap
    K-steps of that conv (adm_conv_args.fold0) -- UNet and classifier (forward on the tape path; the backward network is the same)."""
    g = golden("full_adm64")
    x, t, y = (torch.from_numpy(g[k]).to(DEV) for k in ("x", "t", "y"))
    model, _ = adm64()
    assert model.fold_skip
    fold = model(x, t, y)
    assert any("w2f" in d for d in model._packed.blocks.values())
    model.fold_skip, model._packed = False, None      # the switch is read when the weights are packed
    two = model(x, t, y)
    assert not any("w2f" in d for d in model._packed.blocks.values())
    rf, r2 = rel(fold, g["out"]), rel(two, g["out"])
    print(f"ADM-G-64 UNet vs the reference's fp32 output: skip fold {rf:.3e}, two launches {r2:.3e}")
    assert rf < 2e-2 and r2 < 2e-2 and not torch.equal(fold, two) and rel(fold, two.cpu().numpy()) < 1.5e-2   # two bf16 evaluations, each ~1e-2 from fp32
    gc = golden("full_clf64")
    xc, tc, yc = (torch.from_numpy(gc[k]).to(DEV) for k in ("x", "t", "y"))
    c64 = clf(64, 4)
    gf = c64.log_prob_grad(xc, tc, yc, 1.0)
    c64.fold_skip, c64._packed = False, None
    g2 = c64.log_prob_grad(xc, tc, yc, 1.0)
    print(f"guidance gradient vs the reference's autograd: skip fold {rel(gf, gc['grad']):.3e}, two launches {rel(g2, gc['grad']):.3e}")
    assert rel(gf, gc["grad"]) < 5e-2 and rel(g2, gc["grad"]) < 5e-2 and not torch.equal(gf, g2)


def test_classifier_gn_backward_as_three_passes():
    """ADM_FUSE_GN_BWD=0: partial -> finalize -> apply instead of the backward conv's fused epilogue."""
    gc = golden("full_clf64")
    c64 = clf(64, 4)
    xc, tc, yc = (torch.from_numpy(gc[k]).to(DEV) for k in ("x", "t", "y"))
    assert c64.fuse_gn_bwd
    fused = c64.log_prob_grad(xc, tc, yc, 1.0)
    c64.fuse_gn_bwd = False
    three = c64.log_prob_grad(xc, tc, yc, 1.0)
    rf, r3 = rel(fused, gc["grad"]), rel(three, gc["grad"])
    print(f"guidance gradient vs the reference's autograd: fused epilogue {rf:.3e}, three passes {r3:.3e}")
    assert rf < 5e-2 and r3 < 5e-2 and rel(three, fused.cpu().numpy()) < 1.5e-2 and not torch.equal(three, fused)


def test_fid_accumulation_without_the_side_stream(monkeypatch):
    """ADM_FID_OVERLAP=0: features + Gram on the caller's stream; the float64 sums are bitwise those of the side-stream path."""
    from autodiffusion_amd.fid import ActivationAccumulator
    g = torch.Generator(device=DEV).manual_seed(0)
    u8 = torch.randint(0, 256, (6, 8, 8, 3), device=DEV, dtype=torch.uint8, generator=g)
    proj = torch.randn(192, 64, device=DEV, generator=g)

    def feats(b):
        return b.reshape(b.shape[0], -1).float() @ proj
    feats.stream_safe = True
    sums = []
    for overlap in (True, False):
        monkeypatch.setattr(ActivationAccumulator, "OVERLAP", overlap)
        acc = ActivationAccumulator(64, DEV)
        for i in range(3):
            acc.add_from(feats, u8[2 * i:2 * i + 2])
        n, s1, s2 = acc.pooled()
        sums.append((n, s1.clone(), s2.clone()))
    assert sums[0][0] == sums[1][0] == 6 and torch.equal(sums[0][1], sums[1][1]) and torch.equal(sums[0][2], sums[1][2])


def test_sd_unet_non_default_schedules(monkeypatch):
    """ADM_FOLD_SKIP=0, ADM_SD_FUSE_GEGLU=0 (projection -> tensor -> adm_geglu), ADM_SD_STRIDE2=0 (Downsample as stride-1 conv + pixel pick),
    ADM_SD_SPLITK (split-K on / off; ADM_SD_SPLITK_1X1=1: the wide 1x1 projections too), ADM_UPCONV_PHASES=0 on the reference-captured w320 fixture; ADM_SD_SPLIT_GUIDANCE=0 (one batch of
    2N instead of two half batches on two streams) bitwise equal through the DDIM sampler."""
    from autodiffusion_amd import sd_unet
    from test_hip_sd import _model, check
    from test_sd_oracle import sd_case
    from autodiffusion_amd.ops import splitk_1x1_for as ops_splitk_1x1
    g, plan, P = sd_case("sd_unet_w320")
    x, t, ctx = (torch.from_numpy(g[k]).to(DEV) for k in ("x", "t", "context"))
    m = _model(plan, P)
    base = m(x, t, ctx)
    check(base, g["out"], "sd w320 default")
    assert m.fuse_geglu and sd_unet.STRIDE2_TAPS and m.fold_skip
    m2 = _model(plan, P)
    m2.fold_skip = False                                  # ADM_FOLD_SKIP=0 (read when the weights are packed)
    nofold = m2(x, t, ctx)
    check(nofold, g["out"], "sd w320, skip_connection as its own launch")
    assert not torch.equal(nofold, base)
    del m2
    m.fuse_geglu = False
    two = m(x, t, ctx)
    m.fuse_geglu = True
    check(two, g["out"], "sd w320, GEGLU as a separate pass")
    assert not torch.equal(two, base)
    monkeypatch.setattr(sd_unet, "STRIDE2_TAPS", False)
    check(m(x, t, ctx), g["out"], "sd w320, Downsample as stride-1 conv + pick")
    monkeypatch.setattr(sd_unet, "STRIDE2_TAPS", True)
    check(m.enable_splitk(True)(x, t, ctx), g["out"], "sd w320, split-K")
    monkeypatch.setattr(sd_unet, "SPLITK_1X1", True)      # ADM_SD_SPLITK_1X1=1: the 1280-wide 1x1 projections of the 16x16 / 8x8 levels too
    from autodiffusion_amd import ops as _ops
    monkeypatch.setattr(_ops, "SPLITK_1X1_MIN_CHUNKS", 8)  # the fixture's widest level has 640 channels (SD v1: 1280): lower the threshold so that its
    widths = {b.inner for seq in plan.input_blocks + [plan.middle_block] + plan.output_blocks for b in seq if hasattr(b, "inner")}   # projections split
    assert any(ops_splitk_1x1(8, 8, k, c) > 1 for c in widths for k in (c, 4 * c)), widths
    check(m(x, t, ctx), g["out"], "sd w320, split-K incl. the wide 1x1 projections")
    monkeypatch.setattr(sd_unet, "SPLITK_1X1", False)
    m.enable_splitk(False)
    check(m.enable_upconv_phases(False)(x, t, ctx), g["out"], "sd w320, one-launch Upsample convs")
    m.enable_upconv_phases(True)
    from autodiffusion_amd.sd_sampler import DDIMSampler, LatentDiffusion
    sampler = DDIMSampler(LatentDiffusion(m, device=DEV))
    n = 2
    gen = torch.Generator(device=DEV).manual_seed(3)
    c, uc = (torch.randn(n, ctx.shape[1], ctx.shape[2], device=DEV, generator=gen) for _ in range(2))
    x_T = torch.randn(n, x.shape[1], x.shape[2], x.shape[3], device=DEV, generator=gen)
    outs = []
    for split in (True, False):
        monkeypatch.setattr(type(sampler), "split_guidance", split, raising=False)
        outs.append(sampler.sample(S=3, batch_size=n, shape=list(x.shape[1:]), conditioning=c, verbose=False, eta=0.0, x_T=x_T,
                                   unconditional_guidance_scale=7.5, unconditional_conditioning=uc, sampled_timestep=[100, 500, 900])[0])
    assert torch.isfinite(outs[0]).all() and torch.equal(outs[0], outs[1])


_CHILD = r"""
import sys, torch, torch.nn.functional as F
sys.path.insert(0, %r)
from autodiffusion_amd import ops
dev = "cuda:0"
g = torch.Generator().manual_seed(1)
# 1x1 conv that the resident-tile kernel would take (cin %% 64 == 0, cout >= 256): with ADM_CONV_NO_RESIDENT the staged kernel serves it
x = torch.randn(2, 16, 16, 128, generator=g).to(torch.bfloat16)
w = torch.randn(384, 128, 1, 1, generator=g) * 128 ** -0.5
b = 0.1 * torch.randn(384, generator=g)
got = ops.conv(x.to(dev), ops.pack_conv_weight(w.to(dev)), b.to(dev), 384, 1)
ref = F.conv2d(x.float().permute(0, 3, 1, 2), w.to(torch.bfloat16).float(), b).permute(0, 2, 3, 1)
e1 = float((got.float().cpu() - ref).norm() / ref.norm())
# Inception-style conv: with ADM_CG_NO_LDS the LDS-free kernel serves every layer
xi = torch.randn(2, 17, 17, 64, generator=g).to(torch.float16)
wi = torch.randn(96, 64, 3, 3, generator=g) * (64 * 9) ** -0.5
bi = 0.1 * torch.randn(96, generator=g)
goti = ops.conv2d(xi.to(dev), ops.pack_conv2d_weight(wi.to(dev), None, torch.float16), bi.to(dev), 3, 3, 1, (1, 1), relu=True)
refi = F.relu(F.conv2d(xi.float().permute(0, 3, 1, 2), wi.half().float(), bi, padding=1)).permute(0, 2, 3, 1)
e2 = float((goti.float().cpu() - refi).norm() / refi.norm())
# deep 1x1 conv on a 16x16 map (SD v1's 1280-wide projections): 128-pixel tiles by default, 256-pixel tiles with ADM_CONV_NO_SMALL1X1
xs = torch.randn(3, 16, 16, 1280, generator=g).to(torch.bfloat16)
ws = torch.randn(320, 1280, 1, 1, generator=g) * 1280 ** -0.5
bs = 0.1 * torch.randn(320, generator=g)
gots = ops.conv(xs.to(dev), ops.pack_conv_weight(ws.to(dev)), bs.to(dev), 320, 1)
refs = F.conv2d(xs.float().permute(0, 3, 1, 2), ws.to(torch.bfloat16).float(), bs).permute(0, 2, 3, 1)
e3 = float((gots.float().cpu() - refs).norm() / refs.norm())
import hashlib
# resident-tile 1x1 conv with few pixel tiles and several Cout blocks (+ residual, fused statistics): the blocks are shared out over
# neighbouring grid entries by default, one entry per pixel tile with ADM_C1_NO_CSPLIT
w3 = torch.randn(1152, 128, 1, 1, generator=g) * 128 ** -0.5
b3 = 0.1 * torch.randn(1152, generator=g)
r3 = torch.randn(2, 16, 16, 1152, generator=g).to(torch.bfloat16)
got3 = ops.conv(x.to(dev), ops.pack_conv_weight(w3.to(dev)), b3.to(dev), 1152, 1, res=r3.to(dev), want_stats=True)
ref3 = F.conv2d(x.float().permute(0, 3, 1, 2), w3.to(torch.bfloat16).float(), b3).permute(0, 2, 3, 1) + r3.float()
e4 = float((got3.float().cpu() - ref3).norm() / ref3.norm())
dg = lambda t: hashlib.sha1(t.cpu().contiguous().view(torch.int16).numpy().tobytes()).hexdigest()
# the Stable-Diffusion feed-forward's GEGLU projection (sd_unet.py fuse_geglu, the default): its epilogue exists on the resident-tile
# kernel only, so ADM_CONV_NO_RESIDENT must leave THIS conv there instead of failing every transformer block
wg = torch.randn(512, 128, generator=g) * 128 ** -0.5
bg = 0.1 * torch.randn(512, generator=g)
wgi, bgi = ops.geglu_interleave(wg, bg)
gotg = ops.conv(x.to(dev), ops.pack_conv_weight(wgi[:, :, None, None].to(dev)), bgi.to(dev), 512, 1, geglu=True)
u = F.linear(x.float(), wg.to(torch.bfloat16).float(), bg)
refg = u[..., :256] * F.gelu(u[..., 256:])
e5 = float((gotg.float().cpu() - refg).norm() / refg.norm())
print("ERR", e1, e2, e3, dg(gots), e4, dg(got3), hashlib.sha1(got3._adm_stats[0].cpu().numpy().tobytes()).hexdigest(), e5)
"""


_DIGESTS = {}


@pytest.mark.parametrize("var", ["ADM_CONV_NO_RESIDENT", "ADM_CG_NO_LDS", "ADM_CONV_NO_SMALL1X1", "ADM_C1_NO_CSPLIT", ""])
def test_library_level_switches_in_a_child_process(var):
    env = dict(os.environ)
    if var:
        env[var] = "1"
    r = subprocess.run([sys.executable, "-c", _CHILD % ROOT], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    fields = [ln for ln in r.stdout.splitlines() if ln.startswith("ERR")][0].split()[1:]
    e1, e2, e3 = (float(v) for v in fields[:3])
    print(var or "(default)", "1x1 conv rel", e1, "conv2d rel", e2, "deep 1x1 conv at 16x16 rel", e3)
    assert e1 < 4e-3 and e2 < 2e-3 and e3 < 4e-3, (var, e1, e2, e3)
    e4, e5 = float(fields[4]), float(fields[7])
    assert e4 < 4e-3 and e5 < 6e-3, (var, e4, e5)
    _DIGESTS[var] = (fields[3], fields[5], fields[6])
    if "ADM_CONV_NO_SMALL1X1" in _DIGESTS and "" in _DIGESTS:   # the tile size does not change a single bit of the result
        assert _DIGESTS["ADM_CONV_NO_SMALL1X1"][0] == _DIGESTS[""][0]
    if "ADM_C1_NO_CSPLIT" in _DIGESTS and "" in _DIGESTS:       # nor does the way a tile's Cout blocks are dealt to the grid (output and statistics)
        assert _DIGESTS["ADM_C1_NO_CSPLIT"][1:] == _DIGESTS[""][1:]


def test_cu_partitioned_streams_give_the_same_images():
    """ADM_CU_PARTITION=G (SpacedDiffusion.cu_partition_spec): the UNet and the guidance gradient on CU-masked streams -- disjoint
    parts of the chip, persistent grids sized to each part (adm_stream_create_cumask) -- produce bitwise the images of the default
    two-stream loop and of the sequential one: the partition changes where and when tiles run, never their arithmetic."""
    from autodiffusion_amd.sampler import cu_partition
    gl = golden("full_loop64")
    model, diffusion = adm64()
    c64 = clf(64, 4)
    x_T, yl = torch.from_numpy(gl["x_T"]), torch.from_numpy(gl["y"])
    cls = type(diffusion)
    base, base_u8 = guided_loop(model, diffusion, c64, gl["cand"].tolist(), x_T, yl)
    old = cls.cu_partition_spec
    try:
        for spec in ("96", "64"):
            cls.cu_partition_spec = spec
            us, gs, desc = cu_partition(DEV, spec)
            assert us.cuda_stream != gs.cuda_stream and "guidance gradient on" in desc
            s, u8 = guided_loop(model, diffusion, c64, gl["cand"].tolist(), x_T, yl)
            assert torch.equal(s, base) and torch.equal(u8, base_u8), spec
    finally:
        cls.cu_partition_spec = old
