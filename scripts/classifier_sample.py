#!/usr/bin/env python3
"""Sample an .npz image batch with a searched timestep schedule (and optional per-step layer skipping).

The on-disk format either side of the hot path, as the reference's samplers write it
(scripts/classifier_sample.py:82-205, classifier_sample_prunedUNET.py:151-213, image_sample.py:96-159):
``<save_dir>/samples_{N}x{H}x{W}x3.npz`` with ``arr_0`` = uint8 NHWC images and ``arr_1`` = int64 labels
(when class-conditional); the same flags (``--use_timestep '[153, 424, 926, 690]'``, ``--use_ddim``,
``--classifier_scale``, ``--without_classifier`` ...), plus ``--skip_layers '[[1],[],[0,5],[2,3]]'`` for
dynamic UNets.  Without ``--model_path`` / ``--classifier_path`` the networks keep synthetic weights
(benchmarks, tests).  One process per GPU: ``python -m torch.distributed.run --nproc-per-node N ...``.
"""
import argparse
import os
import sys
import time

import numpy as np
import torch as th
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from autodiffusion_amd import dist_util, logger  # noqa: E402
from autodiffusion_amd.evaluate import CandidateEvaluator, merge_policy  # noqa: E402
from autodiffusion_amd.script_util import (add_dict_to_argparser, args_to_dict, classifier_defaults,  # noqa: E402
                                           create_classifier, create_model_and_diffusion,
                                           model_and_diffusion_defaults)


def create_argparser():
    defaults = dict(
        clip_denoised=True, num_samples=10000, batch_size=16, use_ddim=False, model_path="", classifier_path="",
        save_dir="", classifier_scale=1.0, use_timestep=None, skip_layers=None, MASTER_PORT="12344", use_mean=False,
        without_classifier=False, seed=0, merge_batches=0,
    )
    defaults.update(model_and_diffusion_defaults())
    defaults.update(classifier_defaults())
    parser = argparse.ArgumentParser()
    add_dict_to_argparser(parser, defaults)
    return parser


def main(argv=None):
    t1 = time.time()
    args = create_argparser().parse_args(argv)
    if args.use_mean and args.use_timestep is not None:
        ts = eval(args.use_timestep.replace(" ", ","))
        args.use_timestep = str([round(t) for t in ts])
    os.environ.setdefault("MASTER_PORT", args.MASTER_PORT)
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
        logger.log("loading classifier...")
        classifier = create_classifier(**args_to_dict(args, classifier_defaults().keys()))
        classifier.to(dist_util.dev())
        if args.classifier_path:
            classifier.load_state_dict(dist_util.load_state_dict(args.classifier_path, map_location="cpu"))
        else:
            classifier.randomize_(4321)
        classifier.eval()

    ev = CandidateEvaluator(model, diffusion, classifier, image_size=args.image_size, use_ddim=args.use_ddim,
                            clip_denoised=args.clip_denoised, class_cond=args.class_cond,
                            classifier_scale=args.classifier_scale, device=dist_util.dev())
    steps = sorted(eval(args.use_timestep)) if args.use_timestep is not None else sorted(diffusion.use_timesteps)
    if args.skip_layers is not None:
        ev.set_candidate({"timesteps": steps, "skip_layers": eval(args.skip_layers)})
    else:
        ev.set_candidate(steps)

    logger.log("sampling...")
    world, rank = dist_util.get_world_size(), dist_util.get_rank()
    all_images, all_labels = [], []
    batch_idx = 0
    # --merge_batches K (0 = auto: evaluate.merge_policy -- 256 images per pass at 64x64, 128 at 128x128, 64 at 256x256): K of the reference's batches per pass over the networks -- bitwise the same
    # images (every sub-batch draws from its own generator; an image's result does not depend on the batch it rides in), with the
    # chip filled like a batch of 256 (ADM-G-128 at the launch script's batch 32: +33 % images/s)
    rounds = -(-args.num_samples // (args.batch_size * world))
    merge, per_pass = merge_policy(args.image_size, args.batch_size, int(getattr(args, "merge_batches", 0) or 0), rounds)
    if merge > 1:
        logger.log(f"evaluating {merge} batches of {args.batch_size} per pass ({per_pass} images per pass; bitwise the images of separate passes)")
    while len(all_images) * args.batch_size < args.num_samples:
        k = max(1, min(merge, rounds - batch_idx))
        seeds = [args.seed * 1000003 + (batch_idx + j) * world + rank for j in range(k)]
        if k == 1:
            samples, labels = [ev.sample_batch(args.batch_size, seed=seeds[0])], [ev.last_classes]
        else:
            samples = ev.sample_batches(args.batch_size, seeds)
            labels = list(ev.last_classes.split(args.batch_size, 0))
        for sample, classes in zip(samples, labels):
            sample, classes = sample.contiguous(), classes.contiguous()
            gathered = [th.zeros_like(sample) for _ in range(world)]
            gathered_labels = [th.zeros_like(classes) for _ in range(world)]
            if world > 1:
                dist.all_gather(gathered, sample)
                dist.all_gather(gathered_labels, classes)
            else:
                gathered, gathered_labels = [sample], [classes]
            all_images.extend([s.cpu().numpy() for s in gathered])
            all_labels.extend([lab.cpu().numpy() for lab in gathered_labels])
            batch_idx += 1
            logger.log("created " + str(len(all_images) * args.batch_size) + " samples")

    arr = np.concatenate(all_images, axis=0)[: args.num_samples]
    label_arr = np.concatenate(all_labels, axis=0)[: args.num_samples]
    out_path = None
    if rank == 0:
        shape_str = "x".join(str(x) for x in arr.shape)
        out_path = os.path.join(logger.get_dir() or ".", "samples_" + shape_str + ".npz")
        logger.log("saving to " + str(out_path))
        if args.class_cond:
            np.savez(out_path, arr, label_arr)
        else:
            np.savez(out_path, arr)
    if world > 1:
        dist.barrier()
    logger.log("sampling complete")
    logger.log("total time: " + str(time.time() - t1))
    return out_path


if __name__ == "__main__":
    main()
