#!/usr/bin/env python3
"""Capture BASELINE-size golden vectors by importing the REFERENCE's own modules (build container only).

The architectures are the ones the reference's launch scripts name -- no checkpoint is reachable offline, so every
parameter comes from the deterministic fill rule (``oracle/fill.py``), regenerated on the test side from the
state-dict names.  Only inputs and expected outputs are stored:

    full_adm64.npz      ADM-G ImageNet-64 UNet, 295.9 M parameters (search_imagenet64_classifier_guidance.sh:1):
                        fp32 output, and the reference's own fp16-torso output (``convert_to_fp16``, unet.py:618-624) on
                        the same input -- the error the reference itself accepts, quoted next to ours in DESIGN section 4
    full_clf64.npz      64x64 classifier, depth 4 (65.4 M): logits + the cond_fn gradient (torch.autograd)
    full_loop64.npz     classifier-guided searched 4-step DDIM loop [153,424,926,690], B=2, both networks at full size
    full_adm128.npz     ADM-G ImageNet-128 UNet (421.5 M; configs/128_guided_sample.sh:1), 128x128 classifier depth 2
                        (logits + gradient), and a guided 10-step DDIM loop, B=1
    full_lsun256.npz    ADM LSUN-256 dynamic UNet (552.8 M; search_lsun_cat.sh:1), B=1, with and without a skip list
                        (output stored at every second pixel: 2 x 393 KB instead of 2 x 1.5 MB)
    full_adm256cc.npz   the same network class-conditional (554 M: search_lsun_cat.sh:1 + class_cond), B=1, with and
                        without a skip list, every second pixel
    full_sd_v1.npz      Stable-Diffusion v1 latent UNet (859.5 M; v1-inference_coco.yaml:29-44), one 64x64 latent

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/capture_fullsize.py [name ...]
"""
import copy
import os
import sys
import time
import types

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
GD = "/root/reference/examples/guided_diffusion"
SD = "/root/reference/examples/Stable Diffusion"
sys.dont_write_bytecode = True
sys.path.insert(0, GD)
sys.path.insert(0, ROOT)

from oracle.fill import fill_array  # noqa: E402

torch.set_num_threads(8)


def fill_module(mod):
    with torch.no_grad():
        for k, v in mod.state_dict().items():
            v.copy_(torch.from_numpy(fill_array(k, tuple(v.shape))))
    return mod.eval()


def save(name, **arrs):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrs.items()})
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB, keys={sorted(arrs)}", flush=True)


def rnd(shape, seed):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed))


def gd_imports():
    from guided_diffusion.script_util import (classifier_defaults, create_classifier, create_model_and_diffusion,
                                              model_and_diffusion_defaults)
    return classifier_defaults, create_classifier, create_model_and_diffusion, model_and_diffusion_defaults


def flags_adm64(dynamic=False):
    d = gd_imports()[3]()
    d.update(attention_resolutions="32,16,8", class_cond=True, diffusion_steps=1000, dropout=0.1, image_size=64,
             learn_sigma=True, noise_schedule="cosine", num_channels=192, num_head_channels=64, num_res_blocks=3,
             resblock_updown=True, use_new_attention_order=True, use_fp16=False, use_scale_shift_norm=True,
             use_dynamic_unet=dynamic)
    return d


def flags_adm128():
    d = gd_imports()[3]()
    d.update(attention_resolutions="32,16,8", class_cond=True, image_size=128, learn_sigma=True, num_channels=256,
             num_heads=4, num_res_blocks=2, resblock_updown=True, use_fp16=False, use_scale_shift_norm=True)
    return d


def flags_lsun256():
    d = gd_imports()[3]()
    d.update(attention_resolutions="32,16,8", class_cond=False, diffusion_steps=1000, dropout=0.1, image_size=256,
             learn_sigma=True, noise_schedule="linear", num_channels=256, num_head_channels=64, num_res_blocks=2,
             resblock_updown=True, use_fp16=False, use_scale_shift_norm=True, use_dynamic_unet=True)
    return d


def make_cond_fn(clf, scale):
    """The closure of search_imagenet64_classifier_guidance.py:319-326, restated around the reference classifier."""
    def cond_fn(x, t, y=None, **kw):
        with torch.enable_grad():
            x_in = x.detach().requires_grad_(True)
            logits = clf(x_in, t)
            lp = F.log_softmax(logits, dim=-1)
            sel = lp[range(len(logits)), y.view(-1)]
            return torch.autograd.grad(sel.sum(), x_in)[0] * scale
    return cond_fn


def import_search_driver():
    for missing in ("torchvision", "torchvision.transforms", "blobfile"):
        if missing not in sys.modules:
            sys.modules[missing] = types.ModuleType(missing)
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    import search_imagenet64_classifier_guidance as drv
    return drv


def clf_grad(clf, x, t, y):
    xin = x.clone().requires_grad_(True)
    logits = clf(xin, t)
    sel = F.log_softmax(logits, dim=-1)[range(len(y)), y.view(-1)]
    return logits.detach(), torch.autograd.grad(sel.sum(), xin)[0]


def cap_adm64():
    _, _, create_model_and_diffusion, _ = gd_imports()
    t0 = time.time()
    m, _ = create_model_and_diffusion(**flags_adm64())
    fill_module(m)
    x, t, y = rnd((2, 3, 64, 64), 61), torch.tensor([424, 926]), torch.tensor([207, 5])
    with torch.no_grad():
        out = m(x, t, y)
    print("adm64 fp32 forward done", time.time() - t0, flush=True)
    # the reference's own mixed-precision torso: conv weights of the three block lists in half, GroupNorm / softmax /
    # embeddings / head in fp32 (unet.py:618-624, nn.py:17-19, fp16_util.py:15-22)
    m.dtype = torch.float16  # what use_fp16=True sets at construction (unet.py:464)
    m.convert_to_fp16()
    with torch.no_grad():
        out16 = m(x, t, y)
    print("adm64 fp16 forward done", time.time() - t0, "rel", float((out16 - out).norm() / out.norm()), flush=True)
    save("full_adm64", x=x.numpy(), t=t.numpy(), y=y.numpy(), out=out.numpy(), out_fp16=out16.float().numpy())


def cap_clf64():
    classifier_defaults, create_classifier, _, _ = gd_imports()
    cf = classifier_defaults()
    cf.update(image_size=64, classifier_depth=4)
    clf = fill_module(create_classifier(**cf))
    x, t, y = rnd((2, 3, 64, 64), 62), torch.tensor([926, 153]), torch.tensor([5, 999])
    logits, g = clf_grad(clf, x, t, y)
    save("full_clf64", x=x.numpy(), t=t.numpy(), y=y.numpy(), logits=logits.numpy(), grad=g.numpy())


def cap_loop64():
    classifier_defaults, create_classifier, create_model_and_diffusion, _ = gd_imports()
    drv = import_search_driver()
    m, base = create_model_and_diffusion(**flags_adm64())
    fill_module(m)
    cf = classifier_defaults()
    cf.update(image_size=64, classifier_depth=4)
    clf = fill_module(create_classifier(**cf))
    s = object.__new__(drv.EvolutionSearcher)
    s.base_diffusion, s.active_diffusion = base, copy.deepcopy(base)
    cand = [153, 424, 926, 690]
    s.reset_diffusion(cand)
    diff = s.active_diffusion
    x, y = rnd((2, 3, 64, 64), 63), torch.tensor([281, 948])
    out = {"x_T": x.numpy(), "y": y.numpy(), "cand": np.array(cand)}
    for guided in (True, False):
        tag = "g" if guided else "u"
        torch.manual_seed(7)
        t0 = time.time()
        smp = diff.ddim_sample_loop(lambda xx, tt, y=None, **kw: m(xx, tt, y), (2, 3, 64, 64), noise=x, clip_denoised=True,
                                    model_kwargs={"y": y}, cond_fn=make_cond_fn(clf, 1.0) if guided else None,
                                    device=torch.device("cpu"))
        print("loop64", tag, time.time() - t0, flush=True)
        out[f"ddim_{tag}_sample"] = smp.numpy()
        out[f"ddim_{tag}_uint8"] = ((smp + 1) * 127.5).clamp(0, 255).to(torch.uint8).permute(0, 2, 3, 1).contiguous().numpy()
    save("full_loop64", **out)


def cap_adm128():
    classifier_defaults, create_classifier, create_model_and_diffusion, _ = gd_imports()
    drv = import_search_driver()
    m, base = create_model_and_diffusion(**flags_adm128())
    fill_module(m)
    x, t, y = rnd((1, 3, 128, 128), 64), torch.tensor([500]), torch.tensor([417])
    with torch.no_grad():
        out = m(x, t, y)
    cf = classifier_defaults()
    cf.update(image_size=128, classifier_depth=2, classifier_width=128)
    clf = fill_module(create_classifier(**cf))
    logits, g = clf_grad(clf, x, t, y)
    s = object.__new__(drv.EvolutionSearcher)
    s.base_diffusion, s.active_diffusion = base, copy.deepcopy(base)
    cand = [3, 77, 140, 251, 333, 480, 611, 702, 850, 999]
    s.reset_diffusion(cand)
    torch.manual_seed(7)
    t0 = time.time()
    smp = s.active_diffusion.ddim_sample_loop(lambda xx, tt, y=None, **kw: m(xx, tt, y), (1, 3, 128, 128), noise=x,
                                              clip_denoised=True, model_kwargs={"y": y}, cond_fn=make_cond_fn(clf, 1.0),
                                              device=torch.device("cpu"))
    print("loop128", time.time() - t0, flush=True)
    u8 = ((smp + 1) * 127.5).clamp(0, 255).to(torch.uint8).permute(0, 2, 3, 1).contiguous()
    save("full_adm128", x=x.numpy(), t=t.numpy(), y=y.numpy(), out=out.numpy(), logits=logits.numpy(), grad=g.numpy(),
         cand=np.array(cand), loop_sample=smp.numpy(), loop_uint8=u8.numpy())


def cap_lsun256():
    _, _, create_model_and_diffusion, _ = gd_imports()
    m, _ = create_model_and_diffusion(**flags_lsun256())
    fill_module(m)
    x, t = rnd((1, 3, 256, 256), 65), torch.tensor([333])
    skip = [1, 5, 20, m.layer_num - 2]
    with torch.no_grad():
        t0 = time.time()
        out = m(x, t, None, skip_layer=[])
        print("lsun256 forward", time.time() - t0, flush=True)
        out_s = m(x, t, None, skip_layer=skip)
    save("full_lsun256", x=x.numpy(), t=t.numpy(), skip=np.array(skip), layer_num=np.array(m.layer_num),
         out_sub=out[:, :, ::2, ::2].numpy(), out_skip_sub=out_s[:, :, ::2, ::2].numpy(),
         out_norm=np.array(float(out.double().norm())), out_skip_norm=np.array(float(out_s.double().norm())))


def cap_adm256cc():
    """BASELINE configs[4] as SURVEY 8(d) writes it: search_lsun_cat.sh:1 + class_cond (554 M: label_emb on the 6-level model)."""
    _, _, create_model_and_diffusion, _ = gd_imports()
    flags = flags_lsun256()
    flags["class_cond"] = True
    m, _ = create_model_and_diffusion(**flags)
    fill_module(m)
    x, t, y = rnd((1, 3, 256, 256), 68), torch.tensor([612]), torch.tensor([417])
    skip = [2, 9, 33, m.layer_num - 3]
    with torch.no_grad():
        out = m(x, t, y, skip_layer=[])
        out_s = m(x, t, y, skip_layer=skip)
    save("full_adm256cc", x=x.numpy(), t=t.numpy(), y=y.numpy(), skip=np.array(skip), layer_num=np.array(m.layer_num),
         params=np.array(sum(p.numel() for p in m.parameters())),
         out_sub=out[:, :, ::2, ::2].numpy(), out_skip_sub=out_s[:, :, ::2, ::2].numpy(),
         out_norm=np.array(float(out.double().norm())), out_skip_norm=np.array(float(out_s.double().norm())))


def cap_sd_v1():
    sys.path.insert(0, SD)
    _oc, _lc = types.ModuleType("omegaconf"), types.ModuleType("omegaconf.listconfig")
    _lc.ListConfig = type("ListConfig", (list,), {})
    _oc.listconfig = _lc
    sys.modules.setdefault("omegaconf", _oc)
    sys.modules.setdefault("omegaconf.listconfig", _lc)
    from ldm.modules.diffusionmodules.openaimodel import UNetModel
    net = UNetModel(image_size=32, in_channels=4, out_channels=4, model_channels=320, attention_resolutions=[4, 2, 1],
                    num_res_blocks=2, channel_mult=[1, 2, 4, 4], num_heads=8, use_spatial_transformer=True,
                    transformer_depth=1, context_dim=768, use_checkpoint=False, legacy=False)
    fill_module(net)
    x, ctx, t = rnd((1, 4, 64, 64), 66), rnd((1, 77, 768), 67), torch.tensor([637])
    with torch.no_grad():
        out = net(x, t, ctx)
    save("full_sd_v1", x=x.numpy(), t=t.numpy(), context=ctx.numpy(), out=out.numpy())


ALL = dict(adm64=cap_adm64, clf64=cap_clf64, loop64=cap_loop64, adm128=cap_adm128, lsun256=cap_lsun256, adm256cc=cap_adm256cc, sd_v1=cap_sd_v1)

if __name__ == "__main__":
    for name in (sys.argv[1:] or list(ALL)):
        t0 = time.time()
        ALL[name]()
        print(f"[{name}] {time.time() - t0:.1f} s", flush=True)
