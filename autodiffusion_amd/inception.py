"""FID Inception-v3 pool3 features on the MI355X HIP layers (csrc/adm_convg.hip) -- row A11 `compute_activations` / (f1).

Stands where the reference calls third-party feature extractors (both absent offline, weights URL-fetched):
  * guided_diffusion: ``Evaluator_v1.compute_activations(batches, batch_size)`` -> ``sess.run`` of a frozen TensorFlow
    Inception graph on uint8 NHWC batches (evaluations/evaluator_v1.py:252-280, 665-679)        -> ``features(u8)``
  * Stable Diffusion: ``pytorch_fid.inception.InceptionV3([block_idx])`` applied to float NCHW batches in [0, 1]
    (scripts/search_ea.py:43, 95-127, 171-182)                                                  -> ``InceptionV3.forward``

The network is the published FID variant of Inception-v3 (pytorch_fid.inception / torchvision Inception3 names):
BasicConv2d = Conv2d(bias=False) + BatchNorm(eps 1e-3) + ReLU, folded here into packed 16-bit weights (scale) and an
fp32 bias; every branch of a Mixed block writes its channel slice of the block's NHWC output directly; activations stay
on the device from the sampler's uint8 batch to the float64 FID sums (fid.ActivationAccumulator).

PARITY UNPINNED: no Inception weights, graph or golden activation exist in the image.  ``load_state_dict`` takes the
``pt_inception-2015-12-05`` / torchvision key layout (or pytorch_fid's ``blocks.i.j`` names); until it is called the
module holds random weights and says so.  tests/test_hip_inception.py checks the HIP network against a CPU restatement
of the same architecture (oracle/inception.py) with synthetic weights -- that the architecture equals the reference's
frozen graph is documented, not tested.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, List, Sequence, Tuple

import torch

from . import logger, ops
from ._lib import AdmError

BN_EPS = 1e-3

# (name, cin, cout, kh, kw) of every BasicConv2d, in execution order
_A = lambda n, cin, pf: [(n + ".branch1x1", cin, 64, 1, 1), (n + ".branch5x5_1", cin, 48, 1, 1), (n + ".branch5x5_2", 48, 64, 5, 5),
                         (n + ".branch3x3dbl_1", cin, 64, 1, 1), (n + ".branch3x3dbl_2", 64, 96, 3, 3),
                         (n + ".branch3x3dbl_3", 96, 96, 3, 3), (n + ".branch_pool", cin, pf, 1, 1)]
_C = lambda n, c7: [(n + ".branch1x1", 768, 192, 1, 1), (n + ".branch7x7_1", 768, c7, 1, 1), (n + ".branch7x7_2", c7, c7, 1, 7),
                    (n + ".branch7x7_3", c7, 192, 7, 1), (n + ".branch7x7dbl_1", 768, c7, 1, 1), (n + ".branch7x7dbl_2", c7, c7, 7, 1),
                    (n + ".branch7x7dbl_3", c7, c7, 1, 7), (n + ".branch7x7dbl_4", c7, c7, 7, 1), (n + ".branch7x7dbl_5", c7, 192, 1, 7),
                    (n + ".branch_pool", 768, 192, 1, 1)]
_E = lambda n, cin: [(n + ".branch1x1", cin, 320, 1, 1), (n + ".branch3x3_1", cin, 384, 1, 1), (n + ".branch3x3_2a", 384, 384, 1, 3),
                     (n + ".branch3x3_2b", 384, 384, 3, 1), (n + ".branch3x3dbl_1", cin, 448, 1, 1), (n + ".branch3x3dbl_2", 448, 384, 3, 3),
                     (n + ".branch3x3dbl_3a", 384, 384, 1, 3), (n + ".branch3x3dbl_3b", 384, 384, 3, 1), (n + ".branch_pool", cin, 192, 1, 1)]
CONVS: List[Tuple[str, int, int, int, int]] = (
    [("Conv2d_1a_3x3", 3, 32, 3, 3), ("Conv2d_2a_3x3", 32, 32, 3, 3), ("Conv2d_2b_3x3", 32, 64, 3, 3),
     ("Conv2d_3b_1x1", 64, 80, 1, 1), ("Conv2d_4a_3x3", 80, 192, 3, 3)]
    + _A("Mixed_5b", 192, 32) + _A("Mixed_5c", 256, 64) + _A("Mixed_5d", 288, 64)
    + [("Mixed_6a.branch3x3", 288, 384, 3, 3), ("Mixed_6a.branch3x3dbl_1", 288, 64, 1, 1), ("Mixed_6a.branch3x3dbl_2", 64, 96, 3, 3),
       ("Mixed_6a.branch3x3dbl_3", 96, 96, 3, 3)]
    + _C("Mixed_6b", 128) + _C("Mixed_6c", 160) + _C("Mixed_6d", 160) + _C("Mixed_6e", 192)
    + [("Mixed_7a.branch3x3_1", 768, 192, 1, 1), ("Mixed_7a.branch3x3_2", 192, 320, 3, 3), ("Mixed_7a.branch7x7x3_1", 768, 192, 1, 1),
       ("Mixed_7a.branch7x7x3_2", 192, 192, 1, 7), ("Mixed_7a.branch7x7x3_3", 192, 192, 7, 1), ("Mixed_7a.branch7x7x3_4", 192, 192, 3, 3)]
    + _E("Mixed_7b", 1280) + _E("Mixed_7c", 2048))

# pytorch_fid keeps the layers in nn.Sequential blocks: its own state_dict says blocks.<i>.<j>.<rest>
_FID_BLOCKS = [["Conv2d_1a_3x3", "Conv2d_2a_3x3", "Conv2d_2b_3x3"], ["Conv2d_3b_1x1", "Conv2d_4a_3x3"],
               ["Mixed_5b", "Mixed_5c", "Mixed_5d", "Mixed_6a", "Mixed_6b", "Mixed_6c", "Mixed_6d", "Mixed_6e"],
               ["Mixed_7a", "Mixed_7b", "Mixed_7c"]]


def param_shapes() -> "OrderedDict[str, tuple]":
    s: "OrderedDict[str, tuple]" = OrderedDict()
    for name, cin, cout, kh, kw in CONVS:
        s[name + ".conv.weight"] = (cout, cin, kh, kw)
        for leaf in ("weight", "bias", "running_mean", "running_var"):
            s[name + ".bn." + leaf] = (cout,)
    return s


def _pad32(c: int) -> int:
    return (c + 31) // 32 * 32


class InceptionV3:
    """pytorch_fid.inception.InceptionV3's surface on the HIP layers: ``InceptionV3([block_idx])(x)`` -> list of feature
    maps (float32 NCHW; block 3 = the 2048-d pool3 features as [N, 2048, 1, 1])."""

    DEFAULT_BLOCK_INDEX = 3
    BLOCK_INDEX_BY_DIM = {64: 0, 192: 1, 768: 2, 2048: 3}
    CHUNK = 320   # images per pass: the 8x8 and 17x17 levels need this many to fill 256 CUs (100: -16 %); bounded by the 2 GiB buffer-offset range of the 299x299x32 input (375 images) and ~16 MB of transient activations per image

    def __init__(self, output_blocks: Sequence[int] = (DEFAULT_BLOCK_INDEX,), resize_input: bool = True,
                 normalize_input: bool = True, requires_grad: bool = False, use_fid_inception: bool = True,
                 dtype: torch.dtype = torch.float16):
        if requires_grad:
            raise NotImplementedError("the HIP Inception extractor is inference-only (the reference never trains it)")
        if not use_fid_inception:
            raise NotImplementedError("only the FID variant (pt_inception-2015-12-05 weights) is built")
        if dtype not in (torch.float16, torch.bfloat16):
            raise ValueError("dtype must be torch.float16 or torch.bfloat16")
        self.output_blocks = sorted(output_blocks)
        self.last_needed_block = max(output_blocks)
        assert self.last_needed_block <= 3, "Last possible output block index is 3"
        self.resize_input, self.normalize_input = resize_input, normalize_input
        self.compute_dtype = dtype
        self.training = False
        self.weights_loaded = False
        g = torch.Generator().manual_seed(0)
        self._params: "OrderedDict[str, torch.Tensor]" = OrderedDict()
        for k, shp in param_shapes().items():
            if k.endswith("conv.weight"):
                fan_in = shp[1] * shp[2] * shp[3]
                self._params[k] = (torch.rand(shp, generator=g) * 2 - 1) * (3.0 / fan_in) ** 0.5
            elif k.endswith(("bn.weight", "running_var")):
                self._params[k] = torch.ones(shp)
            else:
                self._params[k] = torch.zeros(shp)
        self._packed: Dict[str, tuple] = {}

    # --- nn.Module-like surface -------------------------------------------------
    def eval(self):
        return self

    def state_dict(self):
        return OrderedDict(self._params)

    def parameters(self):
        return iter(self._params.values())

    def to(self, device):
        for k in self._params:
            self._params[k] = self._params[k].to(device)
        self._packed = {}
        return self

    def cuda(self, device=None):
        return self.to("cuda" if device is None else device)

    @property
    def device(self):
        return next(iter(self._params.values())).device

    def load_state_dict(self, sd, strict: bool = True):
        """Takes torchvision / pt_inception names (``Mixed_5b.branch1x1.conv.weight``) or pytorch_fid's own
        (``blocks.2.0.branch1x1.conv.weight``); the classifier head (``fc.*``), ``AuxLogits.*`` and
        ``num_batches_tracked`` entries of a full checkpoint are not part of the extractor and are skipped."""
        renamed = {}
        for k, v in sd.items():
            if k.startswith("blocks."):
                _, i, j, rest = k.split(".", 3)
                if int(j) >= len(_FID_BLOCKS[int(i)]):
                    continue
                k = _FID_BLOCKS[int(i)][int(j)] + "." + rest
            if k.startswith(("fc.", "AuxLogits.")) or k.endswith("num_batches_tracked"):
                continue
            renamed[k] = v
        missing = [k for k in self._params if k not in renamed]
        unexpected = [k for k in renamed if k not in self._params]
        if strict and (missing or unexpected):
            raise RuntimeError(f"Error(s) in loading state_dict: missing keys {missing[:5]}... unexpected keys {unexpected[:5]}...")
        for k, v in renamed.items():
            if k not in self._params:
                continue
            v = torch.as_tensor(v)
            if tuple(v.shape) != tuple(self._params[k].shape):
                raise RuntimeError(f"size mismatch for {k}: {tuple(v.shape)} vs {tuple(self._params[k].shape)}")
            self._params[k] = v.detach().to(device=self._params[k].device, dtype=torch.float32).clone()
        self._packed = {}
        self.weights_loaded = not missing
        return missing, unexpected

    # --- layers -------------------------------------------------------------------
    def _prepare(self):
        if self._packed:
            return
        if self.device.type != "cuda":
            raise AdmError("InceptionV3: parameters must live on a GPU (the HIP path has no CPU fallback); call .to('cuda')")
        if not self.weights_loaded:
            logger.warn("WARNING: InceptionV3 holds RANDOM weights (the pt_inception-2015-12-05 checkpoint is fetched from a URL by "
                        "pytorch_fid and is not in this image): load_state_dict() it before reading FID values")
        p = self._params
        for name, cin, cout, kh, kw in CONVS:
            scale = p[name + ".bn.weight"] / torch.sqrt(p[name + ".bn.running_var"] + BN_EPS)
            bias = (p[name + ".bn.bias"] - p[name + ".bn.running_mean"] * scale).contiguous()
            self._packed[name] = (ops.pack_conv2d_weight(p[name + ".conv.weight"], scale, self.compute_dtype), bias, kh, kw)

    def _bc(self, name, x, stride=1, pad=(0, 0), out=None):
        w, b, kh, kw = self._packed[name]
        cout = w.shape[0]
        if out is None and cout % 32:
            # the consumer reads whole 32-channel steps: keep the pad channels (x zero weights) finite
            buf = torch.zeros((x.shape[0], (x.shape[1] + 2 * pad[0] - kh) // stride + 1, (x.shape[2] + 2 * pad[1] - kw) // stride + 1,
                               _pad32(cout)), dtype=x.dtype, device=x.device)
            ops.conv2d(x, w, b, kh, kw, stride, pad, True, out=buf[..., :cout])
            return buf
        return ops.conv2d(x, w, b, kh, kw, stride, pad, True, out=out)

    def _new(self, x, c, stride=1):
        n, h, w, _ = x.shape
        if stride == 2:
            h, w = (h - 3) // 2 + 1, (w - 3) // 2 + 1
        return torch.empty((n, h, w, c), dtype=x.dtype, device=x.device)

    def _mixed_a(self, n, x, pf):
        y = self._new(x, 224 + pf)
        self._bc(n + ".branch1x1", x, out=y[..., 0:64])
        self._bc(n + ".branch5x5_2", self._bc(n + ".branch5x5_1", x), pad=(2, 2), out=y[..., 64:128])
        t = self._bc(n + ".branch3x3dbl_2", self._bc(n + ".branch3x3dbl_1", x), pad=(1, 1))
        self._bc(n + ".branch3x3dbl_3", t, pad=(1, 1), out=y[..., 128:224])
        self._bc(n + ".branch_pool", ops.pool2d(x, 3, 1, 1, "avg"), out=y[..., 224:224 + pf])
        return y

    def _mixed_b(self, n, x):
        y = self._new(x, 768, stride=2)
        self._bc(n + ".branch3x3", x, stride=2, out=y[..., 0:384])
        t = self._bc(n + ".branch3x3dbl_2", self._bc(n + ".branch3x3dbl_1", x), pad=(1, 1))
        self._bc(n + ".branch3x3dbl_3", t, stride=2, out=y[..., 384:480])
        ops.pool2d(x, 3, 2, 0, "max", out=y[..., 480:768])
        return y

    def _mixed_c(self, n, x):
        y = self._new(x, 768)
        self._bc(n + ".branch1x1", x, out=y[..., 0:192])
        t = self._bc(n + ".branch7x7_2", self._bc(n + ".branch7x7_1", x), pad=(0, 3))
        self._bc(n + ".branch7x7_3", t, pad=(3, 0), out=y[..., 192:384])
        t = self._bc(n + ".branch7x7dbl_2", self._bc(n + ".branch7x7dbl_1", x), pad=(3, 0))
        t = self._bc(n + ".branch7x7dbl_4", self._bc(n + ".branch7x7dbl_3", t, pad=(0, 3)), pad=(3, 0))
        self._bc(n + ".branch7x7dbl_5", t, pad=(0, 3), out=y[..., 384:576])
        self._bc(n + ".branch_pool", ops.pool2d(x, 3, 1, 1, "avg"), out=y[..., 576:768])
        return y

    def _mixed_d(self, n, x):
        y = self._new(x, 1280, stride=2)
        self._bc(n + ".branch3x3_2", self._bc(n + ".branch3x3_1", x), stride=2, out=y[..., 0:320])
        t = self._bc(n + ".branch7x7x3_2", self._bc(n + ".branch7x7x3_1", x), pad=(0, 3))
        t = self._bc(n + ".branch7x7x3_3", t, pad=(3, 0))
        self._bc(n + ".branch7x7x3_4", t, stride=2, out=y[..., 320:512])
        ops.pool2d(x, 3, 2, 0, "max", out=y[..., 512:1280])
        return y

    def _mixed_e(self, n, x, pool_mode):
        y = self._new(x, 2048)
        self._bc(n + ".branch1x1", x, out=y[..., 0:320])
        t = self._bc(n + ".branch3x3_1", x)
        self._bc(n + ".branch3x3_2a", t, pad=(0, 1), out=y[..., 320:704])
        self._bc(n + ".branch3x3_2b", t, pad=(1, 0), out=y[..., 704:1088])
        t = self._bc(n + ".branch3x3dbl_2", self._bc(n + ".branch3x3dbl_1", x), pad=(1, 1))
        self._bc(n + ".branch3x3dbl_3a", t, pad=(0, 1), out=y[..., 1088:1472])
        self._bc(n + ".branch3x3dbl_3b", t, pad=(1, 0), out=y[..., 1472:1856])
        self._bc(n + ".branch_pool", ops.pool2d(x, 3, 1, 1, pool_mode), out=y[..., 1856:2048])
        return y

    def _body(self, x, upto: int):
        """x: prepared [N,299,299,32] 16-bit NHWC -> block outputs 0..upto (NHWC 16-bit; block 3: fp32 [N, 2048])."""
        outs = []
        x = self._bc("Conv2d_1a_3x3", x, stride=2)
        x = self._bc("Conv2d_2a_3x3", x)
        x = self._bc("Conv2d_2b_3x3", x, pad=(1, 1))
        x = ops.pool2d(x, 3, 2, 0, "max")
        outs.append(x)
        if upto >= 1:
            x = self._bc("Conv2d_3b_1x1", x)
            x = self._bc("Conv2d_4a_3x3", x)
            x = ops.pool2d(x, 3, 2, 0, "max")
            outs.append(x)
        if upto >= 2:
            x = self._mixed_a("Mixed_5b", x, 32)
            x = self._mixed_a("Mixed_5c", x, 64)
            x = self._mixed_a("Mixed_5d", x, 64)
            x = self._mixed_b("Mixed_6a", x)
            for n in ("Mixed_6b", "Mixed_6c", "Mixed_6d", "Mixed_6e"):
                x = self._mixed_c(n, x)
            outs.append(x)
        if upto >= 3:
            x = self._mixed_d("Mixed_7a", x)
            x = self._mixed_e("Mixed_7b", x, "avg")
            x = self._mixed_e("Mixed_7c", x, "max")   # the FID graph's last block pools with a max
            outs.append(ops.global_avgpool_f32(x))
        return outs

    # --- entry points ---------------------------------------------------------------
    def forward(self, inp: torch.Tensor):
        """pytorch_fid contract: float32 NCHW [N, 3, H, W] in [0, 1] -> list of float32 NCHW maps of the requested blocks."""
        self._prepare()
        if inp.dim() != 4 or inp.shape[1] != 3:
            raise AdmError(f"InceptionV3: expected [N, 3, H, W], got {tuple(inp.shape)}")
        if not self.resize_input and tuple(inp.shape[2:]) != (299, 299):
            raise AdmError("InceptionV3(resize_input=False) needs 299 x 299 inputs")
        inp = inp.to(device=self.device, dtype=torch.float32).contiguous()
        scale, shift = (2.0, -1.0) if self.normalize_input else (1.0, 0.0)
        parts: List[List[torch.Tensor]] = []
        for i in range(0, inp.shape[0], self.CHUNK):
            x = ops.resize_bilinear(inp[i:i + self.CHUNK], 299, 299, 32, "f32_nchw", True, scale, shift, self.compute_dtype)
            outs = self._body(x, self.last_needed_block)
            got = []
            for b in self.output_blocks:
                o = outs[b]
                got.append(o.view(o.shape[0], -1, 1, 1) if b == 3 else o[..., :(64, 192, 768)[b]].permute(0, 3, 1, 2).float())
            parts.append(got)
        return [torch.cat([p[j] for p in parts], 0) for j in range(len(self.output_blocks))]

    __call__ = forward

    def features(self, u8_nhwc: torch.Tensor, mode: str = "tf1") -> torch.Tensor:
        """The search drivers' plug (search.EvolutionSearcher(features=...)): uint8 NHWC device batch -> fp32 [B, 2048]
        pool3.  mode "tf1": the frozen graph's own input handling (values in [0, 255], TensorFlow-1 bilinear resize,
        (x - 128) / 128; evaluator_v1.py:263-269); "pt": pytorch_fid's (x / 255, half-pixel bilinear, 2x - 1)."""
        self._prepare()
        if u8_nhwc.dtype != torch.uint8 or u8_nhwc.dim() != 4 or u8_nhwc.shape[3] != 3:
            raise AdmError(f"InceptionV3.features: expected uint8 [N, H, W, 3], got {u8_nhwc.dtype} {tuple(u8_nhwc.shape)}")
        if mode not in ("tf1", "pt"):
            raise ValueError(f"mode must be 'tf1' or 'pt', got {mode!r}")
        u8_nhwc = u8_nhwc.to(self.device).contiguous()
        half, scale, shift = (False, 1.0 / 128.0, -1.0) if mode == "tf1" else (True, 2.0 / 255.0, -1.0)
        outs = []
        for i in range(0, u8_nhwc.shape[0], self.CHUNK):
            x = ops.resize_bilinear(u8_nhwc[i:i + self.CHUNK], 299, 299, 32, "u8_nhwc", half, scale, shift, self.compute_dtype)
            outs.append(self._body(x, 3)[3])
        return torch.cat(outs, 0)


class Evaluator_v1:
    """``compute_activations`` of the reference's evaluator object (evaluations/evaluator_v1.py:249-280) over the HIP
    extractor: uint8 NHWC numpy (or device) array -> (pool_3 [N, 2048], spatial) as numpy, for the reference's host
    ``cal_fid`` (:730-753).  The spatial (sFID) features are not used by cal_fid: an empty array stands in."""

    def __init__(self, model: InceptionV3, mode: str = "tf1"):
        self.model, self.mode = model, mode

    def warmup(self):
        import numpy as np
        self.compute_activations(np.zeros([8, 64, 64, 3], dtype=np.uint8), 8)

    def compute_activations(self, batches, batch_size):
        import numpy as np
        preds = []
        for i in range(0, batches.shape[0], batch_size):
            b = torch.as_tensor(batches[i:i + batch_size]).to(torch.uint8)
            preds.append(self.model.features(b, self.mode).cpu().numpy())
        return np.concatenate(preds, axis=0), np.zeros((batches.shape[0], 0), dtype=np.float32)


InceptionV3.features.stream_safe = True   # every launch goes to torch's current stream (fid.ActivationAccumulator.add_from)


def pool3_features(device, weights_path: str = "", mode: str = "tf1", dtype: torch.dtype = torch.float16,
                   allow_random: bool = False):
    """``--features`` factory of scripts/search_ea.py: -> (features(uint8 NHWC device batch) -> fp32 [B, 2048], 2048).
    weights_path: a state_dict file in the pt_inception-2015-12-05 / torchvision / pytorch_fid key layout (.pth read with
    torch.load(weights_only=True), or .safetensors).  Without it the extractor would score candidates on RANDOM weights --
    a meaningless metric that looks like a result in log.txt -- so that needs an explicit allow_random=True (benchmarks,
    tests); the returned callable then carries `random_weights = True` and the searchers tag every FID line they log.
    `features.stream_safe = True`: every launch goes to torch's current stream (fid.ActivationAccumulator.add_from)."""
    if not weights_path and not allow_random:
        raise ValueError("pool3_features: no Inception-v3 checkpoint given (--inception_path): FID values on random weights are "
                         "meaningless; pass allow_random=True (--inception_random True) for throughput runs and tests")
    m = InceptionV3(dtype=dtype).to(device)
    if weights_path:
        if weights_path.endswith(".safetensors"):
            from safetensors.torch import load_file
            sd = load_file(weights_path)
        else:
            sd = torch.load(weights_path, map_location="cpu", weights_only=True)
        m.load_state_dict(sd)
    if not weights_path:
        m.weights_loaded = True   # asked for: the searchers' per-line tag replaces the first-use warning

    def features(u8):
        return m.features(u8, mode)
    features.random_weights = not weights_path
    features.stream_safe = True
    features.net = m
    return features, 2048
