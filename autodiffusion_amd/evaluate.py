"""Candidate evaluation: sample an image batch with a searched schedule (and layer-skip mask).

The device-side half of ``EvolutionSearcher.get_cand_fid`` (reference
search_imagenet64_classifier_guidance.py:308-366; dict candidates with per-step skip lists:
search_dynamic_unet_imagenet64_classifier_guidance_progressive.py:369-445; unconditional:
search_uncondition_model.py:312-368): reset_diffusion(cand) -> sample loop -> uint8 NHWC batch.
Differences from the reference, by design: images stay on the producing GPU
(no per-batch all_gather / D2H), and every batch is seeded per (seed) so results do not depend on
how batches are sharded over ranks.
"""
from __future__ import annotations

import copy
from typing import Optional, Sequence, Union

import torch

from .schedule import apply_candidate

NUM_CLASSES = 1000

# Images evaluated per pass over the networks when several of the reference's (memory-driven) batches are merged: the cap scales
# with the image area -- 256 images at 64x64 (the headline batch), 128 at 128x128, 64 at 256x256 -- so the activation working set of
# a pass stays near the headline's whatever the model (a 128x128 ADM-G pass of 256 images would carry 4 x the headline's tape).
PASS_IMAGES = {64: 256, 128: 128, 256: 64}


def pass_cap(image_size: int) -> int:
    for size in sorted(PASS_IMAGES):
        if image_size <= size:
            return PASS_IMAGES[size]
    return max(1, PASS_IMAGES[256] * 256 * 256 // (image_size * image_size))


def merge_policy(image_size: int, batch_size: int, requested: int = 0, rounds: int = None):
    """-> (reference batches per pass, images per pass).  requested > 0 is taken as given (--merge_batches K); 0 = auto:
    pass_cap(image_size) // batch_size, at least 1, at most the `rounds` a rank still has to run."""
    merge = int(requested) if requested and requested > 0 else max(1, pass_cap(image_size) // max(1, batch_size))
    if rounds is not None:
        merge = max(1, min(merge, rounds))
    return merge, merge * batch_size


def graph_auto(image_size: int, images_per_pass: int) -> bool:
    """`--use_graph auto`: hipGraph replay when the pass that is actually launched (the MERGED batch, which is what gets captured)
    is small enough for the host's launch rate to be the floor: up to 256 64x64-equivalents per pass (measured: batch 100 x 2
    at 64x64 replayed 6.78 s per candidate against 7.61 s eager; above that the GPU is the slower side)."""
    return images_per_pass * (image_size / 64.0) ** 2 <= 256


class CandidateEvaluator:
    def __init__(self, model, base_diffusion, classifier=None, *, image_size: int, use_ddim: bool = True,
                 clip_denoised: bool = True, class_cond: bool = True, classifier_scale: float = 1.0,
                 device=None, use_graph: bool = False):
        self.model = model
        self.classifier = classifier
        self.base_diffusion = base_diffusion
        self.active_diffusion = copy.deepcopy(base_diffusion)
        self.image_size = image_size
        self.use_ddim = use_ddim
        self.clip_denoised = clip_denoised
        self.class_cond = class_cond
        self.classifier_scale = classifier_scale
        self.device = device if device is not None else model.device
        self.skip_layers = None
        if use_graph:  # hipGraph replay of the UNet evaluation and of the guidance gradient (batches <= ~100: the host's
            model.enable_graph(True)   # ~60 ms of launch work per guided step is otherwise the floor)
            if classifier is not None and hasattr(classifier, "enable_graph"):
                classifier.enable_graph(True)

    def set_candidate(self, cand: Union[Sequence[int], dict]):
        """reset_diffusion(cand); dict candidates also carry one skip-layer list per step."""
        if isinstance(cand, dict):
            assert len(cand["timesteps"]) == len(cand["skip_layers"])
            self.skip_layers = [list(s) for s in cand["skip_layers"]]
            steps = cand["timesteps"]
        else:
            self.skip_layers = None
            steps = cand
        apply_candidate(self.active_diffusion, self.base_diffusion, steps)
        if hasattr(self.model, "plan_graphs"):   # one captured launch sequence per distinct skip set of this candidate
            self.model.plan_graphs(len({tuple(sorted(s)) for s in self.skip_layers}) if self.skip_layers is not None else 1)
        return self

    # closures with the reference's calling convention ------------------------------------------
    def _model_fn(self, x, t, y=None, skip_layers=None):
        yy = y if self.class_cond else None
        if skip_layers is not None:
            # the search script indexes by position in the ASCENDING timestep_map
            sl = skip_layers[self.active_diffusion.timestep_map.index(int(t[0]))]
            return self.model(x, t, yy, skip_layer=sl)
        return self.model(x, t, yy)

    def _cond_fn(self, x, t, y=None, skip_layers=None):
        assert y is not None
        return self.classifier.log_prob_grad(x, t, y, self.classifier_scale)

    def sample_batches(self, batch_size: int, seeds: Sequence[int]):
        """len(seeds) reference batches of `batch_size` images in ONE pass over the networks -> [uint8 NHWC [B, H, W, 3]] per seed.

        Bitwise the images of sample_batch(batch_size, seed) for every seed: each sub-batch draws its labels, x_T and per-step noise
        from its own generator, and an image's result does not depend on how many images ride along (tests/test_hip_bigbatch.py).
        The reference's search batch (100, a memory-driven flag) leaves the 16x16 / 8x8 levels with 200-300 tiles on 256 CUs; two
        batches per pass fill the chip like the headline's 256."""
        dev = self.device
        gens = [torch.Generator(device=dev).manual_seed(int(s_) & 0x7FFFFFFFFFFFFFFF) for s_ in seeds]
        shape1 = (batch_size, 3, self.image_size, self.image_size)
        classes, noise = [], []
        for g_ in gens:   # the draw order of sample_batch, per generator
            classes.append(torch.randint(low=0, high=NUM_CLASSES, size=(batch_size,), device=dev, generator=g_))
            noise.append(torch.randn(*shape1, device=dev, generator=g_))
        classes, x_T = torch.cat(classes, 0), torch.cat(noise, 0)
        d = self.active_diffusion
        d.generator = [(g_, batch_size) for g_ in gens]
        kwargs = {"y": classes}
        if self.skip_layers is not None:
            kwargs["skip_layers"] = self.skip_layers
        fn = d.ddim_sample_loop if self.use_ddim else d.p_sample_loop
        try:
            fn(self._model_fn, tuple(x_T.shape), noise=x_T, clip_denoised=self.clip_denoised, model_kwargs=kwargs,
               cond_fn=self._cond_fn if self.classifier is not None else None, device=dev)
        finally:
            d.generator = None
        self.last_classes = classes
        return list(d.last_uint8_nhwc.split(batch_size, 0))

    def sample_batch(self, batch_size: int, seed: Optional[int] = None, return_float: bool = False):
        """-> uint8 NHWC [B, H, W, 3] on the device (and the fp32 sample if return_float)."""
        dev = self.device
        gen = None
        if seed is not None:
            gen = torch.Generator(device=dev).manual_seed(int(seed) & 0x7FFFFFFFFFFFFFFF)
        classes = torch.randint(low=0, high=NUM_CLASSES, size=(batch_size,), device=dev, generator=gen)
        shape = (batch_size, 3, self.image_size, self.image_size)
        x_T = torch.randn(*shape, device=dev, generator=gen)
        d = self.active_diffusion
        d.generator = gen
        kwargs = {"y": classes}
        if self.skip_layers is not None:
            kwargs["skip_layers"] = self.skip_layers
        fn = d.ddim_sample_loop if self.use_ddim else d.p_sample_loop
        sample = fn(self._model_fn, shape, noise=x_T, clip_denoised=self.clip_denoised, model_kwargs=kwargs,
                    cond_fn=self._cond_fn if self.classifier is not None else None, device=dev)
        u8 = d.last_uint8_nhwc
        self.last_classes = classes
        if return_float:
            return u8, sample
        return u8
