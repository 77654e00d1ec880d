#!/usr/bin/env python3
"""Where does the bf16 torso's 1e-2 come from, and would a MIXED torso (fp16 in some blocks, bf16 elsewhere) buy the reference-grade
error of the all-fp16 torso for less than its 1.8 % of throughput?  (Round-2 review, item 8.)

Measurement tool, not a product path: the full-size ADM-G-64 UNet is evaluated block sequence by block sequence through TWO models
holding the same weights -- one bf16, one fp16 -- and the tensors crossing a precision boundary are cast with torch (a cast is an
extra HBM pass a product path would have to fuse or pay for).  Every choice of "fp16 levels" is compared with the reference's fp32
output (tests/golden/full_adm64.npz, captured by importing the reference).  Levels: 0 = 64x64 ... 3 = 8x8 (+ middle block);
"enc" / "dec" = the input / output half of a level.  Second part: the classifier's guidance gradient with an fp16 backward network
(no loss scaling), against the reference's autograd gradient, with the magnitude range of every backward tensor.
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_hip_fullsize as T  # noqa: E402


def block_levels(plan):
    """(sequence, level, half) for every input / middle / output sequence: the level is the resolution index of its OUTPUT."""
    out, res = [], plan.image_size
    for seq in plan.input_blocks:
        b = seq[0]
        if getattr(b, "down", False):
            res //= 2
        out.append(("in", seq, plan.image_size // res))
    out.append(("mid", plan.middle_block, plan.image_size // res))
    for seq in plan.output_blocks:
        out.append(("out", seq, plan.image_size // res))
        if getattr(seq[-1], "up", False):
            res *= 2
    return out


def run_mixed(models, prs, x, t, y, use16):
    """use16(kind, ds) -> True: that sequence runs on the fp16 model."""
    mb, mh = models
    film = mb._embed(prs[0], t, y)      # fp32 embeddings: identical in both models
    hs, h = [], None
    seqs = block_levels(mb.plan)

    def cast(tn, dt):
        return tn if tn is None or tn.dtype == dt else tn.to(dt)
    for kind, seq, ds in seqs:
        m, pr = (mh, prs[1]) if use16(kind, ds) else (mb, prs[0])
        dt = m.compute_dtype
        if kind == "in":
            h = m._run_seq(pr, seq, cast(h, dt), None, film, set(), x_nchw=x)
            hs.append(h)
        elif kind == "mid":
            h = m._run_seq(pr, seq, cast(h, dt), None, film, set())
        else:
            h = m._run_seq(pr, seq, cast(h, dt), cast(hs.pop(), dt), film, set())
    m, pr = (mh, prs[1]) if use16("head", 1) else (mb, prs[0])
    from autodiffusion_amd import ops
    h = cast(h, m.compute_dtype)
    hd = pr.head
    aff = ops.gn_affine(h, hd["g"], hd["b"])
    return ops.conv(h, hd["w"], hd["cb"], mb.plan.out_channels, 9, aff=aff, silu=True, out_f32_nchw=True)


def main():
    g = T.golden("full_adm64")
    mb, _ = T.adm64()
    mh, _ = T.adm64()
    mh.set_torso("fp16")
    prs = (mb._packed or mb._prepare(), mh._packed or mh._prepare())
    x, t, y = (torch.from_numpy(g[k]).to(T.DEV) for k in ("x", "t", "y"))
    with torch.no_grad():
        cases = [
            ("all bf16", lambda k, ds: False),
            ("all fp16", lambda k, ds: True),
            ("fp16 at 64x64 only (25 % of the FLOPs)", lambda k, ds: ds == 1),
            ("fp16 at 64x64 + 32x32", lambda k, ds: ds <= 2),
            ("fp16 at 16x16 + 8x8 + middle", lambda k, ds: ds >= 4 and k != "head"),
            ("fp16 at 8x8 + middle only", lambda k, ds: ds >= 8 and k != "head"),
            ("fp16 encoder (input blocks + middle), bf16 decoder", lambda k, ds: k in ("in", "mid")),
            ("bf16 encoder, fp16 decoder + head", lambda k, ds: k in ("out", "head")),
            ("fp16 decoder at 64x64 + head only", lambda k, ds: (k == "out" and ds == 1) or k == "head"),
        ]
        print("ADM-G-64 UNet (295.9 M), B = 2, relative Frobenius error vs the reference's fp32 output:")
        for name, fn in cases:
            out = run_mixed((mb, mh), prs, x, t, y, fn)
            print(f"  {name:58s} {T.rel(out, g['out']):.3e}")
        print(f"  (the reference's own fp16 torso vs its fp32:                {T.rel(torch.from_numpy(g['out_fp16']), g['out']):.3e})")

    # ---- the classifier's backward network in fp16 (no loss scaling)
    from autodiffusion_amd import ops
    for size, depth, gold in ((64, 4, "full_clf64"), (128, 2, "full_adm128")):
        gc = T.golden(gold)
        xc, tc, yc = (torch.from_numpy(gc[k]).to(T.DEV) for k in ("x", "t", "y"))
        for torso in ("bf16", "fp16"):
            c = T.clf(size, depth)
            c.set_torso(torso)
            grad, logits = c.log_prob_grad(xc, tc, yc, 1.0, return_logits=True)
            fin = bool(torch.isfinite(grad).all())
            rl = float((logits.cpu() - torch.from_numpy(gc["logits"])).abs().max() / abs(gc["logits"]).max())
            print(f"{size}x{size} classifier (depth {depth}), {torso} forward AND backward network: logits {rl:.3e}, guidance gradient rel "
                  f"{T.rel(grad, gc['grad']):.3e}, finite {fin}, |grad| max {float(grad.abs().max()):.3e}, exact zeros {float((grad == 0).float().mean()):.4f}")
            if torso == "fp16":
                # magnitude range of the backward tensors: fp16's smallest normal is 6.1e-5, its subnormals end at 6e-8
                logits2, tape = c._forward_tape(xc, tc)
                dl = ops.logsoftmax_grad(logits2, yc.to(torch.int64).contiguous(), 1.0)
                gmax, gmed, tiny = [], [], []
                orig = ops.gn_bwd

                def spy(x_, dy, *a, **k):
                    v = dy.float().abs()
                    nz = v[v > 0]
                    gmax.append(float(v.max()))
                    gmed.append(float(nz.median()) if nz.numel() else 0.0)
                    tiny.append(float((v < 6.1e-5).float().mean()))
                    return orig(x_, dy, *a, **k)
                ops.gn_bwd = spy
                try:
                    c._backward_tape(tape, dl)
                finally:
                    ops.gn_bwd = orig
                print(f"    backward tensors entering the {len(gmax)} GroupNorm-backward passes: max |g| {min(gmax):.2e} .. {max(gmax):.2e}, median |g| "
                      f"{min(gmed):.2e} .. {max(gmed):.2e}, fraction below fp16's smallest normal (6.1e-5): {min(tiny):.3f} .. {max(tiny):.3f}")


if __name__ == "__main__":
    main()
