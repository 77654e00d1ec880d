#!/usr/bin/env python3
"""Evolutionary timestep search on the HIP evaluation path -- the reference's search launchers in one CLI.

Flags and defaults follow search_imagenet64_classifier_guidance.py:586-618 (classifier-guided) and
search_uncondition_model.py (``--without_classifier True``): ``--time_step 4 --max_epochs 10
--population_num 50 --mutation_num 25 --crossover_num 15 --m_prob 0.25 --use_ddim_init_x True`` ...
Result lines ("epoch = i : top k result", "No.j [..] fid = ..") go to ``<save_dir>/log.txt`` unchanged.

FID features: by default the bundled HIP Inception-v3 pool3 extractor (autodiffusion_amd/inception.py) with the weights
named by ``--inception_path`` (the pt_inception-2015-12-05 state_dict the reference's evaluator stack downloads; it is
not in this image).  With neither ``--features`` nor ``--inception_path`` the CLI exits: a search that ranks candidates on
random-weight features is meaningless; ``--inception_random True`` opts in for throughput runs and tests, and every FID line
of log.txt is then tagged "FID on RANDOM Inception weights".  ``--features pkg.module:factory``
swaps in any callable ``factory(device) -> (features, dim)`` with ``features(uint8 NHWC device batch) -> fp32 [B, dim]``.
``--ref_path`` is an .npz with ``mu``, ``sigma`` (written from the reference's pickled FIDStatistics).  ``--merge_batches`` (default 0 = auto: evaluate.merge_policy, 256 images per pass at 64x64, 128 at 128x128, 64 at 256x256) evaluates that many reference batches per pass over the networks -- bitwise the
same images (every sub-batch draws from its own generator, and an image's result does not depend on the batch it rides in), with the chip
filled like the headline batch.  ``--use_graph`` (default ``auto`` = on when the merged pass is at most 256 64x64-equivalents, e.g. the reference's batch 100 x 2) replays each UNet evaluation /
guidance gradient as a captured hipGraph: bit-identical, and at such batches the eager path is bound by the host's launch rate
(one whole candidate at batch 100: 7.29 s instead of 7.61 s, bench.py --workload candidate).  ``--population_parallel True`` shards whole
candidates over ranks; otherwise every candidate's images are sharded and the statistics pooled.

``--use_dynamic_unet True`` runs the joint timestep + layer-skip search of
search_dynamic_unet_imagenet64_classifier_guidance_progressive.py (flags ``--index_step``, ``--max_prun``,
``--min_prun`` as there, :717-748): candidates are {'timesteps': [...], 'skip_layers': [[...], ...]}.
"""
import argparse
import importlib
import os
import random
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from autodiffusion_amd import dist_util, logger  # noqa: E402
from autodiffusion_amd.schedule import space_timesteps  # noqa: E402
from autodiffusion_amd.script_util import (add_dict_to_argparser, args_to_dict, classifier_defaults,  # noqa: E402
                                           create_classifier, create_model_and_diffusion,
                                           model_and_diffusion_defaults)
from autodiffusion_amd.search import DynamicEvolutionSearcher, EvolutionSearcher  # noqa: E402


def create_argparser():
    defaults = dict(
        clip_denoised=True, num_samples=10000, batch_size=16, use_ddim=False, model_path="", save_dir="",
        time_step=100, seed=0, deterministic=False, local_rank=0, max_epochs=20, select_num=10, population_num=50,
        m_prob=0.1, crossover_num=25, mutation_num=35, classifier_path="", classifier_scale=1.0, max_fid=48.0,
        thres=0.2, use_ddim_init_x=False, search_space="", ref_path="", MASTER_PORT="12344", init_x="",
        without_classifier=False, features="", inception_path="", inception_input="tf1", inception_random=False, population_parallel=False, fid_on_device=False, use_graph="auto", merge_batches=0,
        index_step=None, max_prun=0.0, min_prun=0.0,
    )
    defaults.update(model_and_diffusion_defaults())
    defaults.update(classifier_defaults())
    parser = argparse.ArgumentParser()
    add_dict_to_argparser(parser, defaults)
    return parser


def build_search_space(args, diffusion):
    if args.search_space == "":
        return None
    core = sorted(eval(args.search_space))
    if args.use_ddim_init_x:
        core += list(space_timesteps(diffusion.original_num_steps, ("ddim" if args.use_ddim else "") + str(args.time_step)))
    r = int(diffusion.original_num_steps / 100)
    space = []
    for s in core:
        space += list(range(max(s - r, 0), min(s + r, diffusion.original_num_steps)))
    return sorted(set(space))


def main(argv=None):
    args = create_argparser().parse_args(argv)
    from autodiffusion_amd.script_util import str2bool
    from autodiffusion_amd.evaluate import graph_auto, merge_policy
    # `auto` is decided on the MERGED batch -- the pass that is launched is what a graph captures
    per_pass = merge_policy(args.image_size, args.batch_size, args.merge_batches, -(-args.num_samples // args.batch_size))[1]
    args.use_graph = graph_auto(args.image_size, per_pass) if str(args.use_graph).lower() == "auto" else str2bool(args.use_graph)
    os.environ.setdefault("MASTER_PORT", args.MASTER_PORT)
    torch.manual_seed(args.seed)
    np.random.seed(args.seed)
    random.seed(args.seed)
    dist_util.setup_dist()
    logger.configure(args.save_dir or None)
    logger.log(str(args))
    logger.log("creating model and diffusion...")
    model, diffusion = create_model_and_diffusion(**args_to_dict(args, model_and_diffusion_defaults().keys()))
    model.to(dist_util.dev())
    if args.model_path:
        model.load_state_dict(dist_util.load_state_dict(args.model_path, map_location="cpu"))
    else:
        model.randomize_(1234)
    if args.use_fp16:
        model.convert_to_fp16()
    model.eval()
    classifier = None
    if not args.without_classifier:
        classifier = create_classifier(**args_to_dict(args, classifier_defaults().keys()))
        classifier.to(dist_util.dev())
        if args.classifier_path:
            classifier.load_state_dict(dist_util.load_state_dict(args.classifier_path, map_location="cpu"))
        else:
            classifier.randomize_(4321)
    if args.features:
        mod, fn = args.features.split(":")
        features, dim = getattr(importlib.import_module(mod), fn)(dist_util.dev())
    else:
        from autodiffusion_amd.inception import pool3_features
        if not args.inception_path and not args.inception_random:
            raise SystemExit("search_ea.py: give --inception_path (the pt_inception-2015-12-05 state_dict) or --features "
                             "pkg.module:factory; FID on random Inception weights ranks candidates on a meaningless metric "
                             "(--inception_random True opts in for throughput runs and tests)")
        features, dim = pool3_features(dist_util.dev(), args.inception_path, args.inception_input,
                                       allow_random=args.inception_random)
        if args.inception_random and not args.inception_path:
            logger.log("WARNING: FID features come from an Inception-v3 with RANDOM weights (--inception_random True): "
                       "every fid value below is NOT a quality metric")
    search_space = build_search_space(args, diffusion)
    if search_space is not None:
        logger.log("search space: " + str(search_space))
    t = time.time()
    if args.use_dynamic_unet:
        searcher = DynamicEvolutionSearcher(args, model=model, base_diffusion=diffusion, time_step=args.time_step,
                                            classifier=classifier, index_step=args.index_step, features=features,
                                            feature_dim=dim, population_parallel=args.population_parallel)
    else:
        searcher = EvolutionSearcher(args, model=model, base_diffusion=diffusion, time_step=args.time_step,
                                     classifier=classifier, search_space=search_space, features=features,
                                     feature_dim=dim, population_parallel=args.population_parallel)
    searcher.search()
    logger.log("total searching time = {:.2f} hours".format((time.time() - t) / 3600))


if __name__ == "__main__":
    main()
