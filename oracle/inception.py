"""Oracle: the FID Inception-v3 pool3 feature extractor (TEST INFRASTRUCTURE) -- PARITY UNPINNED.

The reference scores candidates with Inception-v3 2048-d pool3 features from THIRD-PARTY code that is absent here:
  * guided_diffusion side: a frozen TensorFlow graph ``classify_image_graph_def.pb`` fetched from a URL
    (reference evaluations/evaluator_v1.py:20, 652-679; run at :263-269 on uint8 NHWC batches fed as floats in [0, 255]);
  * Stable-Diffusion side: ``pytorch_fid.inception.InceptionV3([3])`` (reference scripts/search_ea.py:43, 95-127, 171-182;
    pip dependency ``pytorch-fid``, whose weights ``pt_inception-2015-12-05`` are a port of that same graph, also URL-fetched).
Neither the graph, nor the package, nor the weights are in the image, and no reference test or fixture holds an
Inception output: this file restates the PUBLISHED architecture of ``pytorch_fid.inception`` (FID variant of
torchvision's Inception3: BasicConv2d = conv(no bias) + BatchNorm(eps 1e-3) + ReLU; average pools that do not count the
padding in Mixed_5*/6*/7b; a max pool in Mixed_7c's pool branch) with the torchvision state-dict names, as plain
PyTorch-CPU fp32 functional ops.  It checks that the HIP extractor computes THIS network; that this network equals the
reference's features is unpinned until the weights and a golden activation exist.

Input conventions (``prepare``):
  * "pt"  (pytorch_fid): float NCHW in [0, 1] -> bilinear 299x299 (align_corners=False, half-pixel centres) -> 2x - 1
  * "tf1" (the frozen graph): NHWC values in [0, 255] -> TF1 ResizeBilinear (align_corners=False, NO half-pixel
    offset: src = dst * in/out) -> (x - 128) / 128
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

from .fill import fill_array

BN_EPS = 1e-3


def _c(shapes, name, cin, cout, kh, kw):
    shapes[name + ".conv.weight"] = (cout, cin, kh, kw)
    for leaf in ("weight", "bias", "running_mean", "running_var"):
        shapes[name + ".bn." + leaf] = (cout,)


def state_shapes() -> dict:
    """name -> shape of every tensor the extractor reads (torchvision Inception3 names, fc / AuxLogits left out)."""
    s: dict = {}
    _c(s, "Conv2d_1a_3x3", 3, 32, 3, 3)
    _c(s, "Conv2d_2a_3x3", 32, 32, 3, 3)
    _c(s, "Conv2d_2b_3x3", 32, 64, 3, 3)
    _c(s, "Conv2d_3b_1x1", 64, 80, 1, 1)
    _c(s, "Conv2d_4a_3x3", 80, 192, 3, 3)
    for name, cin, pf in (("Mixed_5b", 192, 32), ("Mixed_5c", 256, 64), ("Mixed_5d", 288, 64)):
        _c(s, name + ".branch1x1", cin, 64, 1, 1)
        _c(s, name + ".branch5x5_1", cin, 48, 1, 1)
        _c(s, name + ".branch5x5_2", 48, 64, 5, 5)
        _c(s, name + ".branch3x3dbl_1", cin, 64, 1, 1)
        _c(s, name + ".branch3x3dbl_2", 64, 96, 3, 3)
        _c(s, name + ".branch3x3dbl_3", 96, 96, 3, 3)
        _c(s, name + ".branch_pool", cin, pf, 1, 1)
    _c(s, "Mixed_6a.branch3x3", 288, 384, 3, 3)
    _c(s, "Mixed_6a.branch3x3dbl_1", 288, 64, 1, 1)
    _c(s, "Mixed_6a.branch3x3dbl_2", 64, 96, 3, 3)
    _c(s, "Mixed_6a.branch3x3dbl_3", 96, 96, 3, 3)
    for name, c7 in (("Mixed_6b", 128), ("Mixed_6c", 160), ("Mixed_6d", 160), ("Mixed_6e", 192)):
        _c(s, name + ".branch1x1", 768, 192, 1, 1)
        _c(s, name + ".branch7x7_1", 768, c7, 1, 1)
        _c(s, name + ".branch7x7_2", c7, c7, 1, 7)
        _c(s, name + ".branch7x7_3", c7, 192, 7, 1)
        _c(s, name + ".branch7x7dbl_1", 768, c7, 1, 1)
        _c(s, name + ".branch7x7dbl_2", c7, c7, 7, 1)
        _c(s, name + ".branch7x7dbl_3", c7, c7, 1, 7)
        _c(s, name + ".branch7x7dbl_4", c7, c7, 7, 1)
        _c(s, name + ".branch7x7dbl_5", c7, 192, 1, 7)
        _c(s, name + ".branch_pool", 768, 192, 1, 1)
    _c(s, "Mixed_7a.branch3x3_1", 768, 192, 1, 1)
    _c(s, "Mixed_7a.branch3x3_2", 192, 320, 3, 3)
    _c(s, "Mixed_7a.branch7x7x3_1", 768, 192, 1, 1)
    _c(s, "Mixed_7a.branch7x7x3_2", 192, 192, 1, 7)
    _c(s, "Mixed_7a.branch7x7x3_3", 192, 192, 7, 1)
    _c(s, "Mixed_7a.branch7x7x3_4", 192, 192, 3, 3)
    for name, cin in (("Mixed_7b", 1280), ("Mixed_7c", 2048)):
        _c(s, name + ".branch1x1", cin, 320, 1, 1)
        _c(s, name + ".branch3x3_1", cin, 384, 1, 1)
        _c(s, name + ".branch3x3_2a", 384, 384, 1, 3)
        _c(s, name + ".branch3x3_2b", 384, 384, 3, 1)
        _c(s, name + ".branch3x3dbl_1", cin, 448, 1, 1)
        _c(s, name + ".branch3x3dbl_2", 448, 384, 3, 3)
        _c(s, name + ".branch3x3dbl_3a", 384, 384, 1, 3)
        _c(s, name + ".branch3x3dbl_3b", 384, 384, 3, 1)
        _c(s, name + ".branch_pool", cin, 192, 1, 1)
    return s


def fill_params() -> dict:
    """Synthetic weights by the shared fill rule; running_var made positive (1 + 0.5 u), gamma 1 + 0.2 u."""
    out = {}
    for k, shp in state_shapes().items():
        a = fill_array(k, shp)
        if k.endswith("running_var"):
            a = (1.0 + 5.0 * a).astype(np.float32)      # 0.1 u -> 1 + 0.5 u
        elif k.endswith(".conv.weight"):
            a = (a * np.float32(1.4)).astype(np.float32)  # keeps activations O(1) through ~50 ReLU layers
        out[k] = torch.from_numpy(a)
    return out


def prepare(images: torch.Tensor, mode: str = "pt") -> torch.Tensor:
    """-> float32 NCHW [N, 3, 299, 299] in about [-1, 1]."""
    if mode == "pt":
        x = F.interpolate(images.float(), size=(299, 299), mode="bilinear", align_corners=False)
        return 2 * x - 1
    if mode == "tf1":
        x = images.float().permute(0, 3, 1, 2)
        n, c, h, w = x.shape
        ys = torch.arange(299, dtype=torch.float32) * (h / 299.0)
        xs = torch.arange(299, dtype=torch.float32) * (w / 299.0)
        y0 = ys.floor().long().clamp(max=h - 1); y1 = (y0 + 1).clamp(max=h - 1); fy = (ys - y0.float()).view(1, 1, -1, 1)
        x0 = xs.floor().long().clamp(max=w - 1); x1 = (x0 + 1).clamp(max=w - 1); fx = (xs - x0.float()).view(1, 1, 1, -1)
        top = x[:, :, y0][:, :, :, x0] * (1 - fx) + x[:, :, y0][:, :, :, x1] * fx
        bot = x[:, :, y1][:, :, :, x0] * (1 - fx) + x[:, :, y1][:, :, :, x1] * fx
        return ((top * (1 - fy) + bot * fy) - 128.0) / 128.0
    raise ValueError(mode)


def _bc(p, name, x, stride=1, padding=0):
    x = F.conv2d(x, p[name + ".conv.weight"], None, stride=stride, padding=padding)
    x = F.batch_norm(x, p[name + ".bn.running_mean"], p[name + ".bn.running_var"], p[name + ".bn.weight"],
                     p[name + ".bn.bias"], training=False, eps=BN_EPS)
    return F.relu(x)


def _avg(x):
    return F.avg_pool2d(x, 3, stride=1, padding=1, count_include_pad=False)


def _a(p, n, x):
    b1 = _bc(p, n + ".branch1x1", x)
    b5 = _bc(p, n + ".branch5x5_2", _bc(p, n + ".branch5x5_1", x), padding=2)
    b3 = _bc(p, n + ".branch3x3dbl_3", _bc(p, n + ".branch3x3dbl_2", _bc(p, n + ".branch3x3dbl_1", x), padding=1), padding=1)
    bp = _bc(p, n + ".branch_pool", _avg(x))
    return torch.cat([b1, b5, b3, bp], 1)


def _b(p, n, x):
    b3 = _bc(p, n + ".branch3x3", x, stride=2)
    bd = _bc(p, n + ".branch3x3dbl_3", _bc(p, n + ".branch3x3dbl_2", _bc(p, n + ".branch3x3dbl_1", x), padding=1), stride=2)
    return torch.cat([b3, bd, F.max_pool2d(x, 3, stride=2)], 1)


def _cblk(p, n, x):
    b1 = _bc(p, n + ".branch1x1", x)
    b7 = _bc(p, n + ".branch7x7_1", x)
    b7 = _bc(p, n + ".branch7x7_2", b7, padding=(0, 3))
    b7 = _bc(p, n + ".branch7x7_3", b7, padding=(3, 0))
    bd = _bc(p, n + ".branch7x7dbl_1", x)
    bd = _bc(p, n + ".branch7x7dbl_2", bd, padding=(3, 0))
    bd = _bc(p, n + ".branch7x7dbl_3", bd, padding=(0, 3))
    bd = _bc(p, n + ".branch7x7dbl_4", bd, padding=(3, 0))
    bd = _bc(p, n + ".branch7x7dbl_5", bd, padding=(0, 3))
    bp = _bc(p, n + ".branch_pool", _avg(x))
    return torch.cat([b1, b7, bd, bp], 1)


def _d(p, n, x):
    b3 = _bc(p, n + ".branch3x3_2", _bc(p, n + ".branch3x3_1", x), stride=2)
    b7 = _bc(p, n + ".branch7x7x3_1", x)
    b7 = _bc(p, n + ".branch7x7x3_2", b7, padding=(0, 3))
    b7 = _bc(p, n + ".branch7x7x3_3", b7, padding=(3, 0))
    b7 = _bc(p, n + ".branch7x7x3_4", b7, stride=2)
    return torch.cat([b3, b7, F.max_pool2d(x, 3, stride=2)], 1)


def _e(p, n, x, max_pool_branch):
    b1 = _bc(p, n + ".branch1x1", x)
    b3 = _bc(p, n + ".branch3x3_1", x)
    b3 = torch.cat([_bc(p, n + ".branch3x3_2a", b3, padding=(0, 1)), _bc(p, n + ".branch3x3_2b", b3, padding=(1, 0))], 1)
    bd = _bc(p, n + ".branch3x3dbl_2", _bc(p, n + ".branch3x3dbl_1", x), padding=1)
    bd = torch.cat([_bc(p, n + ".branch3x3dbl_3a", bd, padding=(0, 1)), _bc(p, n + ".branch3x3dbl_3b", bd, padding=(1, 0))], 1)
    pooled = F.max_pool2d(x, 3, stride=1, padding=1) if max_pool_branch else _avg(x)
    bp = _bc(p, n + ".branch_pool", pooled)
    return torch.cat([b1, b3, bd, bp], 1)


def forward(p: dict, x: torch.Tensor, upto: int = 3):
    """x: prepared NCHW [N,3,299,299] -> list of the block outputs 0..upto (pytorch_fid's output_blocks order):
    [N,64,73,73], [N,192,35,35], [N,768,17,17], [N,2048,1,1]."""
    outs = []
    x = _bc(p, "Conv2d_1a_3x3", x, stride=2)
    x = _bc(p, "Conv2d_2a_3x3", x)
    x = _bc(p, "Conv2d_2b_3x3", x, padding=1)
    x = F.max_pool2d(x, 3, stride=2)
    outs.append(x)
    if upto >= 1:
        x = _bc(p, "Conv2d_3b_1x1", x)
        x = _bc(p, "Conv2d_4a_3x3", x)
        x = F.max_pool2d(x, 3, stride=2)
        outs.append(x)
    if upto >= 2:
        for n in ("Mixed_5b", "Mixed_5c", "Mixed_5d"):
            x = _a(p, n, x)
        x = _b(p, "Mixed_6a", x)
        for n in ("Mixed_6b", "Mixed_6c", "Mixed_6d", "Mixed_6e"):
            x = _cblk(p, n, x)
        outs.append(x)
    if upto >= 3:
        x = _d(p, "Mixed_7a", x)
        x = _e(p, "Mixed_7b", x, False)
        x = _e(p, "Mixed_7c", x, True)
        outs.append(F.adaptive_avg_pool2d(x, (1, 1)))
    return outs


def pool3(p: dict, images: torch.Tensor, mode: str = "pt") -> torch.Tensor:
    """[N, 2048] float32 features."""
    with torch.no_grad():
        return forward(p, prepare(images, mode))[3].flatten(1)
