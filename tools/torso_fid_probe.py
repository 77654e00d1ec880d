"""How far does the torso's rounding move a candidate's FID, against the sampling noise of the FID itself?  (GPU; parity of the
FEATURES with the reference's Inception is unpinned -- no weights in the image, and a random-weight Inception's pool3 is nearly
constant over images -- so this is a probe in a plain feature space, the 8 x 8 x 3 average-pooled pixels (192 dimensions, [0, 255]
scale), with random-init ADM-G-64 networks: a RATIO of two Frechet distances, not a FID.)

  a = FD( bf16-torso images  ||  fp16-torso images )   same candidate [153, 424, 926, 690], same seeds (labels, x_T), N images each
  b = FD( fp16-torso images of OTHER seeds  ||  fp16-torso images )   two independent N-image draws of one and the same sampler
  c = FD( bf16-torso images of the other seeds  ||  fp16-torso images )

a is what the 8-bit mantissa does to the statistics; b is what drawing another N images does.  a << b means the torso's error is
far inside the FID's own sampling noise at this N; c ~ b says the same from the other side."""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from autodiffusion_amd.evaluate import CandidateEvaluator  # noqa: E402
from autodiffusion_amd.fid import ActivationAccumulator, FIDStatistics  # noqa: E402


def stats_of(acc):
    st = acc.statistics(local=True)
    return FIDStatistics(st.mu, st.sigma)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=5000)
    ap.add_argument("--batch", type=int, default=250)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dim = 192

    def features(u8):   # uint8 NHWC [B, 64, 64, 3] -> fp32 [B, 192]: 8 x 8 block means (a tool's feature space, not the product path)
        x = u8.to(torch.float32).permute(0, 3, 1, 2)
        return torch.nn.functional.avg_pool2d(x, 8).reshape(x.shape[0], -1).contiguous()
    accs = {}
    for torso in ("bf16", "fp16"):
        model, diffusion, clf = bench.build_guided(bench.adm64_flags(True), 64, 4, torso, torso, dev)
        ev = CandidateEvaluator(model, diffusion, classifier=clf, image_size=64, use_ddim=True, classifier_scale=1.0, class_cond=True, device=dev)
        ev.set_candidate(bench.SCHEDULE)
        for tag, seed0 in (("A", 1000), ("B", 900000)):
            acc = ActivationAccumulator(dim, dev)
            for i in range(a.images // a.batch):
                acc.add(features(ev.sample_batch(a.batch, seed=seed0 + i)))
            accs[(torso, tag)] = acc
        del model, clf, ev
        torch.cuda.empty_cache()
    ref = stats_of(accs[("fp16", "A")])
    fd = lambda k: accs[k].frechet_distance_device(ref, local=True)  # noqa: E731
    out = {"images": a.images, "a_bf16_vs_fp16_same_seeds": fd(("bf16", "A")), "b_fp16_other_seeds_vs_fp16": fd(("fp16", "B")),
           "c_bf16_other_seeds_vs_fp16": fd(("bf16", "B")), "self": fd(("fp16", "A"))}
    out["a_over_b"] = out["a_bf16_vs_fp16_same_seeds"] / out["b_fp16_other_seeds_vs_fp16"]
    print(out)


if __name__ == "__main__":
    main()
