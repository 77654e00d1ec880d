#!/usr/bin/env python3
"""Host enqueue time of one guided 4-step DDIM batch against its GPU time (MI355X): at B = 8 the GPU work is small, so
the wall time is the pure host cost of the ~2,800 kernel launches (measured 60 ms = 21 us per launch through ctypes);
at B = 256 the GPU needs 367 ms, i.e. the host has a 6x margin and the step is GPU-bound."""
import os, sys, time, argparse
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from bench import adm64_flags, SCHEDULE
from autodiffusion_amd.evaluate import CandidateEvaluator
from autodiffusion_amd.script_util import (args_to_dict, classifier_defaults, create_classifier,
                                           create_model_and_diffusion, model_and_diffusion_defaults)
dev = torch.device("cuda:0")
flags = adm64_flags(class_cond=True)
model, diffusion = create_model_and_diffusion(**args_to_dict(argparse.Namespace(**flags), model_and_diffusion_defaults().keys()))
model.to(dev).randomize_(1234).convert_to_fp16()
cf = classifier_defaults(); cf.update(image_size=64, classifier_depth=4)
clf = create_classifier(**cf); clf.to(dev).randomize_(4321)
ev = CandidateEvaluator(model, diffusion, classifier=clf, image_size=64, use_ddim=True, clip_denoised=True, class_cond=True, classifier_scale=1.0, device=dev)
ev.set_candidate(SCHEDULE)
for B in (256, 8):
    ev.sample_batch(B, seed=1); torch.cuda.synchronize()
    t0 = time.perf_counter(); ev.sample_batch(B, seed=2); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"B={B}: host enqueue {1e3*(t1-t0):.1f} ms, total {1e3*(t2-t0):.1f} ms")
