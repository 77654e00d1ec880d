#!/usr/bin/env python3
"""Unconditional / class-conditional sampling WITHOUT classifier guidance into an .npz image batch: the counterpart of
the reference's scripts/image_sample.py (:81-159; flags :162-178) on the HIP path.

Same flags (``--model_path``, ``--use_ddim``, ``--use_timestep '[0, 250, 500, 750]'`` for a searched subset,
``--timestep_respacing``, ``--clip_denoised``, ``--num_samples``, ``--batch_size``, ``--save_dir``, ``--port``), same output:
``<save_dir>/samples_{N}x{H}x{W}x3.npz`` with ``arr_0`` = uint8 NHWC images (+ ``arr_1`` = int64 labels when
``--class_cond True``), same log lines ("sampling...", "created N samples", "saving to ...", "sampling time: ...",
"sampling complete").  ``--skip_layers '[[1],[],...]'`` adds per-step layer skipping for ``--use_dynamic_unet True``
models; without ``--model_path`` the network keeps synthetic weights (benchmarks, tests).
One process per GPU: ``python -m torch.distributed.run --nproc-per-node N scripts/image_sample.py ...``.
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
from autodiffusion_amd.script_util import (add_dict_to_argparser, args_to_dict, create_model_and_diffusion,  # noqa: E402
                                           model_and_diffusion_defaults)


def create_argparser():
    defaults = dict(clip_denoised=True, num_samples=10000, batch_size=16, use_ddim=False, model_path="", port="12346",
                    save_dir="", use_timestep=None, skip_layers=None, seed=0, gpu="", merge_batches=0)
    defaults.update(model_and_diffusion_defaults())
    parser = argparse.ArgumentParser()
    add_dict_to_argparser(parser, defaults)
    return parser


def main(argv=None):
    args = create_argparser().parse_args(argv)
    os.environ.setdefault("MASTER_PORT", args.port)
    dist_util.setup_dist()
    logger.configure(args.save_dir or None)

    logger.log("creating model and diffusion...")
    model, diffusion = create_model_and_diffusion(**args_to_dict(args, model_and_diffusion_defaults().keys()))
    model.to(dist_util.dev())
    if args.model_path:
        model.load_state_dict(dist_util.load_state_dict(args.model_path, map_location="cpu"))
        logger.log('load from: ' + args.model_path)
    else:
        model.randomize_(1234)
    if args.use_fp16:
        model.convert_to_fp16()
    model.eval()

    ev = CandidateEvaluator(model, diffusion, None, image_size=args.image_size, use_ddim=args.use_ddim,
                            clip_denoised=args.clip_denoised, class_cond=args.class_cond, device=dist_util.dev())
    steps = sorted(eval(args.use_timestep)) if args.use_timestep is not None else sorted(diffusion.use_timesteps)
    if args.skip_layers is not None:
        ev.set_candidate({"timesteps": steps, "skip_layers": eval(args.skip_layers)})
    else:
        ev.set_candidate(steps)

    logger.log("sampling...")
    world, rank = dist_util.get_world_size(), dist_util.get_rank()
    all_images, all_labels = [], []
    batch_idx = 0
    t1 = time.time()
    # --merge_batches K (0 = auto: evaluate.merge_policy): K of the reference's batches per
    # pass over the network, bitwise the same images (scripts/classifier_sample.py)
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
            if world > 1:
                gathered = [th.zeros_like(sample) for _ in range(world)]
                gathered_labels = [th.zeros_like(classes) for _ in range(world)]
                dist.all_gather(gathered, sample)
                dist.all_gather(gathered_labels, classes)
            else:
                gathered, gathered_labels = [sample], [classes]
            all_images.extend([s.cpu().numpy() for s in gathered])
            all_labels.extend([lab.cpu().numpy() for lab in gathered_labels])
            batch_idx += 1
            logger.log("created " + str(len(all_images) * args.batch_size) + " samples")
    sample_time = time.time() - t1
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
    logger.log("sampling time: " + str(sample_time))
    logger.log("sampling complete")
    return out_path


if __name__ == "__main__":
    main()
