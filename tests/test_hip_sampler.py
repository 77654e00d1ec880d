"""GPU parity of the full candidate-evaluation path through the reference-shaped interface:
create_model_and_diffusion / create_classifier -> reset_diffusion(cand) -> ddim / p sample loops
(with classifier guidance and per-step layer skipping) -> uint8 NHWC batch, against the golden
vectors captured from the reference (tests/golden/sampler_loops_*.npz) and the CPU oracle.

Tolerance (stated, and held to what is measured): 4 sampler steps through two bf16 networks with random fill-rule
weights, against the reference's fp32 result.  Required: relative Frobenius error of the final fp32 sample <= 2.5e-2
(measured 1.1e-2 .. 1.5e-2), and on the uint8 NHWC image >= 98.5 % of pixels within 8 levels, >= 90 % within 2
(measured: see the printed histograms; the BASELINE-size loops in test_hip_fullsize.py give 99.3 % / 95.5 %).
SURVEY section 7 proposed <= 2/255 for >= 99.9 % of pixels: out of reach for ANY 8-bit-mantissa torso -- one bf16 UNet
evaluation differs from fp32 by 1.0e-2 relative (the reference's own fp16 torso: 1.4e-3), 4 steps compound to 1.1e-2 of
an image whose range is 255 levels, i.e. an RMS error of ~1.5 levels with a tail; the same loop with the guidance
gradient of the fp32 oracle injected instead of the bf16 one has the same error (test_hip_fullsize.py prints both), so
the bf16 gradient is not what limits it.  FID parity is "unpinned" (no Inception graph / reference statistics offline).
"""
import copy

import numpy as np
import pytest
import torch

from helpers import filled, golden, plan_c64, plan_m32, plan_m64

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _load(model, plan):
    model.load_state_dict({k: torch.from_numpy(v) for k, v in filled(plan).items()})
    return model.to(DEV).eval()


def _setup_m64():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from autodiffusion_amd.script_util import (classifier_defaults, create_classifier,
                                               create_model_and_diffusion, model_and_diffusion_defaults)
    d = model_and_diffusion_defaults()
    d.update(image_size=64, num_channels=32, num_res_blocks=1, channel_mult="1,2,2", attention_resolutions="16",
             num_head_channels=32, class_cond=True, learn_sigma=True, resblock_updown=True,
             use_scale_shift_norm=True, use_new_attention_order=True, use_dynamic_unet=True,
             noise_schedule="cosine")
    model, diffusion = create_model_and_diffusion(**d)
    _load(model, plan_m64(dynamic=True))
    c = classifier_defaults()
    c.update(image_size=64, classifier_width=64, classifier_depth=1)
    clf = _load(create_classifier(**c), plan_c64())
    return model, diffusion, clf


def _check(sample, u8, g, tag):
    ref = torch.from_numpy(g[f"{tag}_sample"])
    r = ((sample.cpu() - ref).norm() / ref.norm()).item()
    print(tag, "rel fro", r)
    assert r < 2.5e-2, (tag, r)
    if f"{tag}_uint8" in g.files:
        d = np.abs(u8.cpu().numpy().astype(int) - g[f"{tag}_uint8"].astype(int))
        hist = {k: round(float((d <= k).mean()), 4) for k in (0, 1, 2, 4, 8)}
        print(tag, "uint8 within k levels", hist)
        assert hist[8] >= 0.985 and hist[2] >= 0.90, (tag, hist)


def test_guided_and_unguided_loops_match_reference_golden():
    from autodiffusion_amd.evaluate import CandidateEvaluator
    g = golden("sampler_loops_m64")
    model, diffusion, clf = _setup_m64()
    x_T, y = torch.from_numpy(g["x_T"]).to(DEV), torch.from_numpy(g["y"]).to(DEV)
    noises = [torch.from_numpy(n).to(DEV) for n in g["noises"]]
    for use_ddim, name in ((True, "ddim"), (False, "ddpm")):
        for classifier, tag in ((None, "u"), (clf, "g")):
            ev = CandidateEvaluator(model, diffusion, classifier, image_size=64, use_ddim=use_ddim, device=DEV)
            ev.set_candidate(g["cand"].tolist())
            d = ev.active_diffusion
            assert d.timestep_map == sorted(g["cand"].tolist()) and d.num_timesteps == 4
            # inject the reference's noise draws: step k uses noises[k]
            it = iter(noises)
            orig = torch.randn_like
            torch.randn_like = lambda x: next(it)
            try:
                fn = d.ddim_sample_loop if use_ddim else d.p_sample_loop
                sample = fn(ev._model_fn, (2, 3, 64, 64), noise=x_T, clip_denoised=True, model_kwargs={"y": y},
                            cond_fn=ev._cond_fn if classifier is not None else None, device=torch.device(DEV))
            finally:
                torch.randn_like = orig
            _check(sample, d.last_uint8_nhwc, g, f"{name}_{tag}")
            # the fused uint8 pack equals packing the returned fp32 sample
            from autodiffusion_amd import ops
            assert torch.equal(d.last_uint8_nhwc, ops.pack_u8_nhwc(sample))


def test_layer_skip_candidate_matches_reference_golden():
    from autodiffusion_amd.evaluate import CandidateEvaluator
    g = golden("sampler_loops_m64")
    model, diffusion, clf = _setup_m64()
    skip_layers = [[int(v) for v in s.split(",") if v] for s in g["skip_layers"]]
    ev = CandidateEvaluator(model, diffusion, clf, image_size=64, use_ddim=True, device=DEV)
    ev.set_candidate({"timesteps": g["cand"].tolist(), "skip_layers": skip_layers})
    x_T, y = torch.from_numpy(g["x_T"]).to(DEV), torch.from_numpy(g["y"]).to(DEV)
    sample = ev.active_diffusion.ddim_sample_loop(
        ev._model_fn, (2, 3, 64, 64), noise=x_T, model_kwargs={"y": y, "skip_layers": skip_layers},
        cond_fn=ev._cond_fn, device=torch.device(DEV))
    _check(sample, None, g, "ddim_g_skip")


def test_unconditional_uniform_ddim4_and_return_all_images():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from autodiffusion_amd.script_util import create_model_and_diffusion, model_and_diffusion_defaults
    g = golden("sampler_loops_m32_uncond")
    d = model_and_diffusion_defaults()
    d.update(image_size=32, num_channels=32, num_res_blocks=1, channel_mult="1,2,2", attention_resolutions="16,8",
             num_head_channels=32, class_cond=False, learn_sigma=True, resblock_updown=True,
             use_scale_shift_norm=True, use_new_attention_order=True, noise_schedule="cosine",
             timestep_respacing="ddim4")
    model, diffusion = create_model_and_diffusion(**d)
    _load(model, plan_m32(dynamic=False, class_cond=False))
    assert diffusion.timestep_map == [0, 250, 500, 750]
    x_T = torch.from_numpy(g["x_T"]).to(DEV)
    noises = [torch.from_numpy(n).to(DEV) for n in g["noises"]]
    for name, fn in (("ddim", diffusion.ddim_sample_loop), ("ddpm", diffusion.p_sample_loop)):
        it = iter(noises)
        orig = torch.randn_like
        torch.randn_like = lambda x: next(it)
        try:
            sample = fn(model, (2, 3, 32, 32), noise=x_T, clip_denoised=True, model_kwargs={})
        finally:
            torch.randn_like = orig
        ref = torch.from_numpy(g[f"{name}_sample"])
        r = ((sample.cpu() - ref).norm() / ref.norm()).item()
        print(name, "uncond rel fro", r)
        assert r < 2.5e-2
    imgs = diffusion.ddim_sample_loop(model, (2, 3, 32, 32), noise=x_T, model_kwargs={}, return_all_images=True)
    assert len(imgs) == 5 and torch.equal(imgs[0], x_T)  # AutoDiffusion yields the start noise first
    dd = copy.deepcopy(diffusion)
    assert dd.num_timesteps == 4


def test_two_stream_guidance_overlap_is_bit_identical():
    """eps(x_t) on the launch stream and the guidance gradient on a second HIP stream (SpacedDiffusion.overlap_guidance,
    the default) against the sequential order of the reference (gaussian_diffusion.py:258-326 then :381-393)."""
    from autodiffusion_amd.evaluate import CandidateEvaluator
    from autodiffusion_amd.sampler import SpacedDiffusion
    assert SpacedDiffusion.overlap_guidance is True
    model, diffusion, clf = _setup_m64()
    outs = []
    for use_ddim in (True, False):
        for overlap in (True, False, True):
            ev = CandidateEvaluator(model, diffusion, clf, image_size=64, use_ddim=use_ddim, device=DEV)
            ev.set_candidate([153, 424, 926, 690])
            ev.active_diffusion.overlap_guidance = overlap
            outs.append(ev.sample_batch(5, seed=11).clone())
        assert torch.equal(outs[-1], outs[-2]) and torch.equal(outs[-2], outs[-3])


def test_hipgraph_replay_of_unet_and_guidance_gradient_is_bit_identical():
    """CandidateEvaluator(use_graph=True): the UNet evaluation and the guidance gradient are captured once per (shapes,
    layer-skip set, launching stream) and replayed -- guided and unguided, ddim and ddpm, plain and layer-skip candidates,
    with the gradient on the side stream -- against the eager path, twice (second pass: pure replay)."""
    from autodiffusion_amd.evaluate import CandidateEvaluator
    model, diffusion, clf = _setup_m64()
    cases = [([153, 424, 926, 690], None), ([94, 217, 574], [[1], [], [0, 5]])]
    for use_ddim in (True, False):
        for classifier in (clf, None):
            want = []
            for graph in (False, True, True):
                model.enable_graph(False)
                clf.enable_graph(False)
                ev = CandidateEvaluator(model, diffusion, classifier, image_size=64, use_ddim=use_ddim, device=DEV, use_graph=graph)
                got = []
                for steps, skips in cases:
                    ev.set_candidate(steps if skips is None else {"timesteps": steps, "skip_layers": skips})
                    got.append(ev.sample_batch(3, seed=5).clone())
                    got.append(ev.sample_batch(3, seed=6).clone())
                if not want:
                    want = got
                for a, b in zip(got, want):
                    assert torch.equal(a, b), (use_ddim, classifier is not None, graph)
    assert len(model._packed.graphs) >= 3 and len(clf._packed.graphs) >= 1     # {no skips, [1], [0, 5]} x streams
    model.enable_graph(False)
    clf.enable_graph(False)


def test_graph_cache_grows_with_the_candidates_distinct_skip_sets():
    """A dynamic-UNet candidate carries one skip list per step: with more distinct lists than the LRU holds every evaluation
    of every batch would miss and recapture (the round-2 advisor's finding).  set_candidate() sizes the cache to the
    candidate (<= GRAPH_CACHE_MAX), so the second batch of a 14-set candidate is pure replay; beyond the cap the candidate
    runs eagerly.  Results equal the eager path's bitwise either way."""
    from autodiffusion_amd.evaluate import CandidateEvaluator
    model, diffusion, _ = _setup_m64()
    L = model.layer_num
    steps = sorted(range(40, 40 + 14 * 60, 60))
    import itertools
    skips = [list(c) for c in itertools.islice(itertools.combinations(range(L), 2), 14)]   # 14 distinct sets > the default LRU of 12
    assert len(skips) == 14
    cand = {"timesteps": steps, "skip_layers": skips}
    model.enable_graph(False)
    ev = CandidateEvaluator(model, diffusion, None, image_size=64, use_ddim=True, device=DEV, use_graph=False)
    ev.set_candidate(cand)
    want = ev.sample_batch(2, seed=9).clone()
    evg = CandidateEvaluator(model, diffusion, None, image_size=64, use_ddim=True, device=DEV, use_graph=True)
    evg.set_candidate(cand)
    assert model.GRAPH_CACHE >= 14 and not model._graph_eager
    assert torch.equal(evg.sample_batch(2, seed=9), want)
    keys = list(model._packed.graphs.keys())
    assert len(keys) == 14
    assert torch.equal(evg.sample_batch(2, seed=9), want)
    assert list(model._packed.graphs.keys()) == keys            # second batch: no eviction, no recapture
    # beyond the cap: eager evaluation (one log line), no thrash
    model.GRAPH_CACHE_MAX, old = 4, model.GRAPH_CACHE_MAX
    try:
        evg.set_candidate(cand)
        assert model._graph_eager
        assert torch.equal(evg.sample_batch(2, seed=9), want)
        assert list(model._packed.graphs.keys()) == keys
        evg.set_candidate(steps[:4])                            # a plain candidate fits again
        assert not model._graph_eager
    finally:
        model.GRAPH_CACHE_MAX = old
        model.GRAPH_CACHE = type(model).GRAPH_CACHE
        model.enable_graph(False)


def test_graph_pools_are_bounded_by_bytes_not_only_by_count():
    """Every captured graph owns a private activation pool (LSUN-256 at batch 64: > 10 GB each, and a layer-skip candidate brings one
    per distinct skip set): the replay path measures a capture's pool and falls back to eager launches when the candidate's
    `distinct sets x pool bytes` exceed GRAPH_POOL_FRACTION of the device's HBM (the round-3 advisor's finding: bounded by count
    alone, 13-32 such pools could exhaust the HBM mid-search).  Results equal the eager path's bitwise either way."""
    import itertools
    from autodiffusion_amd.evaluate import CandidateEvaluator
    model, diffusion, _ = _setup_m64()
    L = model.layer_num
    steps = sorted(range(40, 40 + 6 * 100, 100))
    skips = [list(c) for c in itertools.islice(itertools.combinations(range(L), 2), 6)]
    cand = {"timesteps": steps, "skip_layers": skips}
    model.enable_graph(False)
    ev = CandidateEvaluator(model, diffusion, None, image_size=64, use_ddim=True, device=DEV, use_graph=False)
    ev.set_candidate(cand)
    want = ev.sample_batch(2, seed=11).clone()
    evg = CandidateEvaluator(model, diffusion, None, image_size=64, use_ddim=True, device=DEV, use_graph=True)
    model._packed.__dict__.pop("graphs", None)
    old_frac, old_seen = model.GRAPH_POOL_FRACTION, model._pool_bytes_seen
    try:
        evg.set_candidate(cand)
        assert torch.equal(evg.sample_batch(2, seed=11), want)
        rep = model.graph_report()
        assert rep["cached_graphs"] == 6 and not rep["eager_fallback"] and rep["pool_bytes_largest"] > 0 and model._pool_bytes_seen > 0, rep
        pool = model._pool_bytes_seen
        # a budget of 3 such pools: 6 distinct sets do not fit -> the candidate runs eagerly (known BEFORE any capture now)
        model.GRAPH_POOL_FRACTION = 3.5 * pool / torch.cuda.get_device_properties(0).total_memory
        evg.set_candidate(cand)
        assert model._graph_eager and model.graph_report()["eager_fallback"]
        assert torch.equal(evg.sample_batch(2, seed=11), want)
        evg.set_candidate({"timesteps": steps[:3], "skip_layers": skips[:3]})      # 3 sets fit again
        assert not model._graph_eager
        # unknown pool size (first capture of a model) with a too-small budget: the first capture replays once, then eager
        model._packed.__dict__.pop("graphs", None)
        model._pool_bytes_seen = 0
        model.GRAPH_POOL_FRACTION = 1.5 * pool / torch.cuda.get_device_properties(0).total_memory
        evg.set_candidate(cand)
        assert not model._graph_eager
        assert torch.equal(evg.sample_batch(2, seed=11), want)
        assert model._graph_eager and len(model._packed.graphs) <= 1
    finally:
        model.GRAPH_POOL_FRACTION, model._pool_bytes_seen = old_frac, old_seen
        model._graph_eager = False
        model.enable_graph(False)


def test_denoised_fn_steps_and_loops_match_the_reference_capture():
    """`denoised_fn` (gaussian_diffusion.py:258-326 honours it; round 3 raised): applied to the predicted x_0 before the clip, between
    two launches of the step kernel.  Single steps (the step arithmetic alone: the reference's model output / gradient come from the
    fp32 oracle, golden-pinned) to 1e-5, and both guided loops through the HIP networks within the loops' bf16 bound."""
    from helpers import denoised_fn_fixture as dfn
    from autodiffusion_amd.evaluate import CandidateEvaluator
    from oracle import nets
    g = golden("sampler_denoised_m64")
    model, diffusion, clf = _setup_m64()
    ev = CandidateEvaluator(model, diffusion, clf, image_size=64, use_ddim=True, device=DEV)
    ev.set_candidate(g["cand"].tolist())
    d = ev.active_diffusion
    x, y = torch.from_numpy(g["x"]), torch.from_numpy(g["y"])
    P, CP = nets.params_from_numpy(filled(plan_m64(dynamic=True))), nets.params_from_numpy(filled(plan_c64()))
    for idx in (2, 0):
        t = torch.full((1,), idx, dtype=torch.int64)
        tm = torch.full((1,), d.timestep_map[idx], dtype=torch.int64)
        with torch.no_grad():
            mo = nets.unet_forward(P, plan_m64(dynamic=True), x, tm, y).to(DEV)
        gr = nets.classifier_grad(CP, plan_c64(), x, tm, y, 1.0).to(DEV)
        nz = torch.from_numpy(g[f"noise_i{idx}"]).to(DEV)
        orig = torch.randn_like
        torch.randn_like = lambda x_: nz
        try:
            for guided, cf in (("u", None), ("g", lambda x_, t_, **kw: gr)):
                for clip in ((True, False) if (idx == 2 and cf is not None) else (True,)):
                    tag = f"i{idx}_{guided}" + ("" if clip else "_noclip")
                    o = d.ddim_sample(lambda x_, t_, **kw: mo, x.to(DEV), t.to(DEV), clip_denoised=clip, denoised_fn=dfn, cond_fn=cf, eta=0.3)
                    for k_, key in (("sample", "sample"), ("pred_xstart", "x0")):
                        np.testing.assert_allclose(o[k_].cpu().numpy(), g[f"ddim_{tag}_{key}"], rtol=2e-4, atol=2e-5)
                    o = d.p_sample(lambda x_, t_, **kw: mo, x.to(DEV), t.to(DEV), clip_denoised=clip, denoised_fn=dfn, cond_fn=cf)
                    for k_, key in (("sample", "sample"), ("pred_xstart", "x0")):
                        np.testing.assert_allclose(o[k_].cpu().numpy(), g[f"ddpm_{tag}_{key}"], rtol=2e-4, atol=2e-5)
        finally:
            torch.randn_like = orig
    noises = [torch.from_numpy(n).to(DEV) for n in g["noises"]]
    for use_ddim, name in ((True, "ddim"), (False, "ddpm")):
        ev = CandidateEvaluator(model, diffusion, clf, image_size=64, use_ddim=use_ddim, device=DEV)
        ev.set_candidate(g["cand"].tolist())
        d = ev.active_diffusion
        it = iter(noises)
        orig = torch.randn_like
        torch.randn_like = lambda x_: next(it)
        try:
            fn = d.ddim_sample_loop if use_ddim else d.p_sample_loop
            sample = fn(ev._model_fn, (1, 3, 64, 64), noise=x.to(DEV), clip_denoised=True, denoised_fn=dfn, model_kwargs={"y": y.to(DEV)},
                        cond_fn=ev._cond_fn, device=torch.device(DEV))
        finally:
            torch.randn_like = orig
        _check(sample, None, g, f"{name}_loop")


def test_single_step_candidate_and_odd_batch_match_the_oracle():
    """Edge cases of the candidate space: a ONE-step candidate (reset_diffusion's K == 1 quirk: `posterior_log_variance_clipped` holds the raw
    variance, search_imagenet64_classifier_guidance.py:242-247 -- it feeds the learned-range interpolation of p_sample) and an odd batch
    (3 images: ragged image groups in every tile of the 8x8 level), guided, both samplers, against the oracle's loops (golden-pinned for
    the K == 1 tables, tests/test_oracle_golden.py)."""
    from autodiffusion_amd.evaluate import CandidateEvaluator
    from oracle import nets, sampler as osm, schedule as osch
    model, diffusion, clf = _setup_m64()
    P, CP = nets.params_from_numpy(filled(plan_m64(dynamic=True))), nets.params_from_numpy(filled(plan_c64()))
    g = torch.Generator().manual_seed(5)
    x_T, y = torch.randn(3, 3, 64, 64, generator=g), torch.tensor([1, 500, 999])
    nz = [torch.randn(3, 3, 64, 64, generator=g)]
    od = osch.OracleDiffusion(steps=1000, noise_schedule="cosine", learn_sigma=True).reset([500])
    for use_ddim in (True, False):
        ev = CandidateEvaluator(model, diffusion, clf, image_size=64, use_ddim=use_ddim, device=DEV)
        ev.set_candidate([500])
        d = ev.active_diffusion
        assert d.num_timesteps == 1 and d.timestep_map == [500]
        it = iter([n.to(DEV) for n in nz])
        orig = torch.randn_like
        torch.randn_like = lambda x_: next(it)
        try:
            fn = d.ddim_sample_loop if use_ddim else d.p_sample_loop
            sample = fn(ev._model_fn, (3, 3, 64, 64), noise=x_T.to(DEV), clip_denoised=True, model_kwargs={"y": y.to(DEV)},
                        cond_fn=ev._cond_fn, device=torch.device(DEV))
        finally:
            torch.randn_like = orig
        ref = osm.sample_loop(od, lambda x_, t_, y=None: nets.unet_forward(P, plan_m64(dynamic=True), x_, t_, y), x_T, use_ddim=use_ddim,
                              cond_fn=lambda x_, t_, y=None: nets.classifier_grad(CP, plan_c64(), x_, t_, y, 1.0), noises=nz, model_kwargs={"y": y})
        r = ((sample.cpu() - ref).norm() / ref.norm()).item()
        print("K = 1,", "ddim" if use_ddim else "ddpm", "batch 3: rel", r)
        assert torch.isfinite(sample).all() and r < 2.5e-2, r
        assert d.last_uint8_nhwc.shape == (3, 64, 64, 3)
        # the batch-of-3 rows equal their own single-image evaluations bitwise (no batch dependence at ragged sizes either)
        ev1 = CandidateEvaluator(model, diffusion, clf, image_size=64, use_ddim=use_ddim, device=DEV)
        ev1.set_candidate([500])
        it = iter([nz[0][2:3].to(DEV)])
        torch.randn_like = lambda x_: next(it)
        try:
            fn1 = ev1.active_diffusion.ddim_sample_loop if use_ddim else ev1.active_diffusion.p_sample_loop
            s1 = fn1(ev1._model_fn, (1, 3, 64, 64), noise=x_T[2:3].to(DEV), clip_denoised=True, model_kwargs={"y": y[2:3].to(DEV)},
                     cond_fn=ev1._cond_fn, device=torch.device(DEV))
        finally:
            torch.randn_like = orig
        assert torch.equal(s1, sample[2:3])
