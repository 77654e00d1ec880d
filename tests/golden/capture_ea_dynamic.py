#!/usr/bin/env python3
"""Capture the trajectory of the REFERENCE's dynamic ("progressive") evolutionary search -- candidates are
{'timesteps': [...], 'skip_layers': [[...], ...]} -- under a synthetic fitness, by importing the reference's own
search driver (GD/search_dynamic_unet_imagenet64_classifier_guidance_progressive.py).

Runs only in the build container (needs /root/reference); the GPU box never sees the reference.  Output:
ea_dynamic_trajectory.npz next to this script (candidate strings, fitness values, skip-layer ranges: data only).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/capture_ea_dynamic.py

As in capture_golden.py, empty stand-in modules are registered *in this capture process only* for two imports the
driver never uses on this path (``torchvision.transforms``, ``blobfile``); ``EvolutionSearcher.__init__`` needs
TensorFlow, so the instance is made with ``object.__new__`` and given the attributes the recorded methods read.
"""
import os
import random
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/examples/guided_diffusion"
sys.dont_write_bytecode = True
sys.path.insert(0, REF)

for missing in ("torchvision", "torchvision.transforms", "blobfile"):
    if missing not in sys.modules:
        sys.modules[missing] = types.ModuleType(missing)
sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]

import search_dynamic_unet_imagenet64_classifier_guidance_progressive as drv  # noqa: E402

LAYERS = 14  # layer_num of the 32x32 test model (tests/helpers.py::plan_m32(dynamic=True))


def fitness_of(cand):
    """Synthetic, deterministic: distance of the sorted timesteps to a target + a reward for skipped layers."""
    ts = np.sort(np.array(cand["timesteps"], dtype=np.float64))
    tgt = np.array([150.0, 420.0, 690.0, 930.0])
    k = min(len(ts), len(tgt))
    f = float(np.abs(ts[:k] - tgt[:k]).sum() / 10.0) + 3.0 * abs(len(ts) - len(tgt))
    skipped = sum(len(s) for s in cand["skip_layers"])
    return f - 0.05 * skipped + 0.001 * sum(sum(s) for s in cand["skip_layers"])


def run(max_epochs, seed, use_ddim_init_x):
    class A:
        pass
    args = A()
    args.max_epochs, args.select_num, args.population_num = max_epochs, 4, 10
    args.m_prob, args.crossover_num, args.mutation_num = 0.25, 3, 5
    args.max_fid, args.max_prun, args.min_prun = 48.0, 0.5, 0.1
    args.use_ddim_init_x, args.use_ddim, args.time_step = use_ddim_init_x, True, 4
    drv.args = args

    class _Log:
        @staticmethod
        def log(*a, **k):
            pass
    drv.logger = _Log

    class _Model:
        layer_num = LAYERS

    class _Diff:
        original_num_steps = 1000

    s = object.__new__(drv.EvolutionSearcher)
    s.args, s.model, s.base_diffusion, s.classifier = args, _Model(), _Diff(), None
    s.init_time_step = args.time_step
    s.max_index_number = args.time_step * LAYERS
    s.max_epochs, s.select_num, s.population_num = args.max_epochs, args.select_num, args.population_num
    s.m_prob, s.crossover_num, s.mutation_num = args.m_prob, args.crossover_num, args.mutation_num
    s.keep_top_k = {s.select_num: [], 50: []}
    s.epoch, s.candidates, s.vis_dict = 0, [], {}
    s.max_fid, s.max_prun, s.min_prun = args.max_fid, args.max_prun, args.min_prun
    s.rf_features, s.rf_lebal = [], []
    s.model_layers = LAYERS
    s.skip_layer_range = [0, 0]
    s.last_best_cand = None
    evaluated, ranges = [], []

    def fitness(cand=None, args=None):
        evaluated.append(str(cand))
        ranges.append(list(s.skip_layer_range))
        return fitness_of(cand)
    s.get_cand_fid = fitness
    random.seed(seed)
    np.random.seed(seed)
    s.search()
    top = s.keep_top_k[50]
    return dict(evaluated=np.array(evaluated), ranges=np.array(ranges, dtype=np.float64),
                final_candidates=np.array(s.candidates), top50=np.array(top),
                top50_fid=np.array([s.vis_dict[c]["fid"] for c in top], dtype=np.float64),
                final_range=np.array(s.skip_layer_range, dtype=np.float64))


if __name__ == "__main__":
    out = {}
    for tag, (ep, seed, init) in {"a": (9, 0, True), "b": (8, 3, False)}.items():
        r = run(ep, seed, init)
        print(tag, "evaluations:", len(r["evaluated"]), "final range:", r["final_range"], "first:", r["evaluated"][0])
        for k, v in r.items():
            out[f"{tag}_{k}"] = v
    out["layers"] = np.array(LAYERS)
    np.savez_compressed(os.path.join(HERE, "ea_dynamic_trajectory.npz"), **out)
