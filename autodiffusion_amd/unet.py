"""ADM UNet / dynamic (layer-skip) UNet forward on hand-written HIP kernels.

Host-side mirror of the reference model interface (``UNetModel`` reference
guided_diffusion/unet.py:396-665, ``Dynamic_UNetModel`` dynamic_unet.py:416-702):
same constructor bookkeeping (via ``arch.build_unet_plan``), the same
state-dict key layout (so the public ``64x64_diffusion.pt`` etc. load
unchanged), ``__call__(x, timesteps, y=None[, skip_layer=[]])`` taking and
returning fp32 NCHW tensors.

Everything between is the MI355X engine: activations are bf16 NHWC in HBM, every
op is a libadm_hip.so launch (ops.py); PyTorch only owns the memory and the stream.
Launch sequence per ResBlock (reference unet.py:236-256):
    gn_partial + gn_finalize           in_layers GroupNorm statistics -> affine
    conv3x3 [affine+SiLU prologue]     in_layers conv   (virtual concat of the UNet skip)
    gn_partial + gn_finalize           out_layers GroupNorm + FiLM (1+scale, shift) -> affine
    [conv1x1]                          skip_connection when channels change
    conv3x3 [affine+SiLU, +residual]   out_layers conv + skip
AttentionBlock (unet.py:299-305): gn -> conv1x1 [affine prologue] -> flash attention ->
conv1x1 [+residual].  The 36 emb_layers Linear projections run as ONE fp32 GEMM per step.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, List, Optional, Sequence

import os

import torch

from . import ops
from ._lib import AdmError
from .arch import AttnSpec, HeadSpec, ResBlockSpec, ResampleSpec, StemSpec, UNetPlan

_ZERO_INIT_SUFFIXES = ("out_layers.3.weight", "out_layers.3.bias", "proj_out.weight", "proj_out.bias")


_warned_precision = set()


def warn_compute_dtype(what: str, flag: str):
    """One loud line per process and flag: the caller asked for the reference's fp32 model; the HIP engine has ONE compute
    type (bf16 operands, fp32 accumulation / GroupNorm statistics / softmax / embeddings / sampler step)."""
    if flag in _warned_precision:
        return
    _warned_precision.add(flag)
    import warnings
    from . import logger
    msg = (f"{what}: {flag}=False asks for the reference's fp32 network; the MI355X HIP path computes with a 16-bit torso "
           "(bf16 by default, fp16 with model.set_torso('fp16') / ADM_TORSO=fp16) and fp32 accumulation, GroupNorm "
           "statistics, softmax and embeddings: one evaluation differs from the reference's fp32 result by ~1e-2 relative in "
           "bf16, 1.3e-3 in fp16 (the reference's own fp16 torso: 1.4e-3; tests/test_hip_fullsize.py)")
    logger.warn("WARNING: " + msg)
    warnings.warn(msg, stacklevel=3)


class HipModule:
    """Minimal parameter container with the subset of nn.Module the search drivers use."""

    compute_dtype = torch.bfloat16  # what the kernels compute in, whatever `dtype` (the reference's attribute) says

    def set_torso(self, torso: str):
        """16-bit element type of activations and weights between kernels: "bf16" (default; BASELINE config 2) or "fp16"
        (the reference's own torso type under use_fp16=True: 11 mantissa bits instead of 8, the same kernels built for
        IEEE half, libadm_hip_f16.so).  Accumulation, GroupNorm statistics, softmax, embeddings, sampler stay fp32."""
        if torso not in ("bf16", "fp16"):
            raise ValueError(f"torso must be 'bf16' or 'fp16', got {torso!r}")
        self.compute_dtype = torch.float16 if torso == "fp16" else torch.bfloat16
        self._packed = None
        return self

    def __init__(self, plan: UNetPlan, use_fp16: bool):
        self.plan = plan
        if os.environ.get("ADM_TORSO", "bf16") == "fp16" and not getattr(self, "with_backward", False):
            self.compute_dtype = torch.float16   # the classifier's backward network stays bf16 (gradient range)
        self.dtype = torch.float16 if use_fp16 else torch.float32  # reference attribute (unet.py:464)
        self._params: "OrderedDict[str, torch.Tensor]" = OrderedDict()
        self._packed = None
        self.training = False
        g = torch.Generator().manual_seed(0)
        for name, shape in plan.param_shapes().items():
            self._params[name] = self._init_param(name, shape, g)

    @staticmethod
    def _init_param(name, shape, g):
        # zero_module'd tensors start at zero like the reference; the rest get a fan-in scaled
        # uniform draw (checkpoints overwrite everything; only the zero pattern is behavioural)
        if name.endswith(_ZERO_INIT_SUFFIXES) or name in ("out.2.weight", "out.2.bias"):
            return torch.zeros(shape)
        leaf = name.rsplit(".", 1)[-1]
        if len(shape) >= 2:
            fan_in = 1
            for s in shape[1:]:
                fan_in *= s
            bound = (1.0 / fan_in) ** 0.5
            return (torch.rand(shape, generator=g) * 2 - 1) * bound
        if leaf == "weight":
            return torch.ones(shape)
        return torch.zeros(shape)

    # --- nn.Module-like surface -------------------------------------------------
    def state_dict(self):
        return OrderedDict(self._params)

    def load_state_dict(self, sd, strict=True):
        missing = [k for k in self._params if k not in sd]
        unexpected = [k for k in sd if k not in self._params]
        if strict and (missing or unexpected):
            raise RuntimeError(f"Error(s) in loading state_dict: missing keys {missing[:5]}... "
                               f"unexpected keys {unexpected[:5]}...")
        for k, v in sd.items():
            if k not in self._params:
                continue
            v = torch.as_tensor(v)
            if tuple(v.shape) != tuple(self._params[k].shape):
                raise RuntimeError(f"size mismatch for {k}: {tuple(v.shape)} vs {tuple(self._params[k].shape)}")
            self._params[k] = v.detach().to(device=self._params[k].device, dtype=torch.float32).clone()
        self._packed = None
        return missing, unexpected

    def parameters(self):
        return iter(self._params.values())

    def named_parameters(self):
        return iter(self._params.items())

    def to(self, device):
        device = torch.device(device)
        for k in self._params:
            self._params[k] = self._params[k].to(device)
        self._packed = None
        return self

    def cuda(self, device=None):
        return self.to("cuda" if device is None else device)

    def eval(self):
        self.training = False
        return self

    def train(self, mode=True):
        if mode:
            raise NotImplementedError("the HIP engine is inference-only (AutoDiffusion is training-free)")
        return self

    def requires_grad_(self, flag=False):
        return self

    def randomize_(self, seed: int = 1234, std_scale: float = 1.0):
        """Synthetic weights for benchmarks (no checkpoint is reachable offline): every tensor,
        including the zero-initialised ones, ~ N(0, std_scale^2 / fan_in); GroupNorm gamma = 1."""
        for i, (k, v) in enumerate(self._params.items()):
            g = torch.Generator(device=v.device).manual_seed(seed + i)
            if v.dim() >= 2:
                fan_in = v[0].numel()
                self._params[k] = torch.randn(v.shape, generator=g, device=v.device) * (std_scale / fan_in ** 0.5)
            elif k.rsplit(".", 1)[-1] == "weight":
                self._params[k] = torch.ones_like(v)
            else:
                self._params[k] = torch.randn(v.shape, generator=g, device=v.device) * 0.02
        self._packed = None
        return self

    def convert_to_fp16(self):
        """Reference API (unet.py:618-624).  The HIP torso always computes in bf16 with fp32
        accumulate/GroupNorm/softmax; this only records the reference's dtype attribute."""
        self.dtype = torch.float16
        return self

    def convert_to_fp32(self):
        warn_compute_dtype(type(self).__name__ + ".convert_to_fp32()", "use_fp16")
        self.dtype = torch.float32
        return self

    @property
    def device(self):
        return next(iter(self._params.values())).device

    def __call__(self, *a, **kw):
        return self.forward(*a, **kw)


class _Prep:
    """Device-resident, kernel-ready parameters derived from the state dict."""
    pass


def _graph_pool_bytes(graph, dev, reserved0, free0) -> int:
    """Bytes of the private allocator pool a just-captured graph owns: the segments tagged with the graph's pool id in the caching
    allocator's snapshot; if the snapshot does not tell (older / newer torch), the growth of reserved memory or the drop of free
    device memory across the capture, whichever is larger."""
    try:
        pid = tuple(graph.pool())
        got = sum(seg["total_size"] for seg in torch.cuda.memory_snapshot() if tuple(seg.get("segment_pool_id", (0, 0))) == pid
                  and seg.get("device", dev.index) == dev.index)
        if got > 0:
            return int(got)
    except Exception:
        pass
    return int(max(0, torch.cuda.memory_reserved(dev) - reserved0, free0 - torch.cuda.mem_get_info(dev)[0]))


class AdmNet(HipModule):
    """Blocks shared by the UNet and the classifier half-UNet."""

    def __init__(self, plan: UNetPlan, use_fp16: bool = False):
        super().__init__(plan, use_fp16)
        self.image_size = plan.image_size
        self.in_channels = plan.in_channels
        self.model_channels = plan.model_channels
        self.out_channels = plan.out_channels
        self.num_classes = plan.num_classes
        if plan.dynamic:
            self.layer_num = plan.layer_num

    # ResBlock skip_connection folded into the out_layers conv's K loop (adm_conv_args.fold0); read when the weights are packed
    fold_skip = os.environ.get("ADM_FOLD_SKIP", "1") != "0"
    with_backward = False  # classifier: also pack the backward-data weight images
    grad_scale = 1.0       # classifier with an fp16 backward network: static power-of-two scale of d(logits) (classifier.py)

    # ------------------------------------------------------------------ hipGraph replay
    # One UNet evaluation is ~700 launches, one guidance gradient ~1000: ~18 us of host time each through ctypes.  At the
    # headline batch (256) the GPU is the slower side; at the reference's search batch (100) and below the host is
    # (tools/host_overhead.py: ~60 ms of enqueue per guided step whatever the batch).  enable_graph() captures an
    # evaluation once per (entry point, input shapes, layer-skip set, launching stream) and replays it: bit-identical
    # outputs.  Graphs (with their private allocator pools) live in the packed-weight object and die with it; at most
    # GRAPH_CACHE of them are kept (layer-skip candidates bring a new launch sequence per distinct skip list).
    # up-ResBlocks: conv3x3(upsample2x(.)) as four 2x2-tap phase convs on the half-resolution tensor (4/9 of the MACs; x 1.5
    # at batch 256, tools/upconv_bench.py).  A per-model choice, never a function of the batch: the pre-summed taps round
    # differently from the nine separate ones, and an image's result must not depend on how many ride along.
    upconv_phases = ops.UPCONV_PHASES

    use_graph = False
    GRAPH_CACHE = 12        # graphs kept per model (LRU); plan_graphs() raises it to the active candidate's needs
    GRAPH_CACHE_MAX = 32    # ... up to this many
    # ... and never more than this fraction of the device's HBM in graph pools: every graph owns a PRIVATE activation pool
    # (LSUN-256 at batch 64: one first-level activation is 2.1 GB, a forward's pool > 10 GB; a 13-32-step layer-skip candidate brings
    # one launch sequence per distinct skip set).  Pools are measured at capture (reserved-memory delta).
    GRAPH_POOL_FRACTION = float(os.environ.get("ADM_GRAPH_POOL_FRACTION", "0.45"))
    _graph_eager = False    # the active candidate needs more graphs than fit (count or bytes): evaluate eagerly
    _planned_sets = 1
    _pool_bytes_seen = 0    # largest pool a capture of this model has needed so far

    def enable_graph(self, flag: bool = True):
        self.use_graph = bool(flag)
        return self

    def _graph_budget(self) -> int:
        dev = self.device
        if dev.type != "cuda":
            return 0
        return int(self.GRAPH_POOL_FRACTION * torch.cuda.get_device_properties(dev).total_memory)

    def plan_graphs(self, distinct_sets: int):
        """Called per candidate (CandidateEvaluator.set_candidate) with the number of DISTINCT layer-skip sets its steps use:
        each needs its own captured launch sequence.  An LRU smaller than that count would miss on every evaluation of every
        batch (recapture = 2 warm-up runs + capture + sync: several times slower than eager), so the cache grows to the
        candidate's needs up to GRAPH_CACHE_MAX graphs AND GRAPH_POOL_FRACTION of HBM in pools (the pool size of this model's
        captures is known after the first one; _graphed re-checks at every capture); beyond either bound the candidate is
        evaluated eagerly, with one log line."""
        from . import logger
        distinct_sets = max(1, int(distinct_sets))
        self._planned_sets = distinct_sets
        too_many = distinct_sets > self.GRAPH_CACHE_MAX
        too_big = self._pool_bytes_seen > 0 and distinct_sets * self._pool_bytes_seen > self._graph_budget()
        if not (too_many or too_big):
            self.GRAPH_CACHE = max(type(self).GRAPH_CACHE, distinct_sets)
            self._graph_eager = False
        else:
            if not self._graph_eager and self.use_graph:
                why = (f"{distinct_sets} distinct layer-skip sets > {self.GRAPH_CACHE_MAX} cached graphs" if too_many else
                       f"{distinct_sets} graphs x {self._pool_bytes_seen / 1e9:.1f} GB of private pool > {self._graph_budget() / 1e9:.0f} GB budget")
                logger.log(f"hipGraph replay off for this candidate: {why} (eager launches instead of recapturing every step)")
            self._graph_eager = True
        return self

    def graph_report(self):
        """What the replay path holds: bench.py puts it on the JSON line of a --graph run."""
        graphs = getattr(self._packed, "graphs", None) if getattr(self, "_packed", None) is not None else None
        nb = [e[3] for e in graphs.values()] if graphs else []
        return {"cached_graphs": len(nb), "pool_gb_total": round(sum(nb) / 1e9, 3), "pool_gb_largest": round(max(nb) / 1e9, 3) if nb else 0.0,
                "pool_bytes_largest": max(nb) if nb else 0, "pool_gb_of_a_capture": round(self._pool_bytes_seen / 1e9, 3),
                "budget_gb": round(self._graph_budget() / 1e9, 1), "planned_distinct_sets": self._planned_sets,
                "eager_fallback": bool(self._graph_eager)}

    def _graphed(self, key, fn, inputs):
        """Replay (capturing at first use) fn(*inputs) -> tensor or tuple of tensors; inputs are device tensors."""
        pr = self._packed
        graphs = pr.__dict__.setdefault("graphs", OrderedDict())
        dev = inputs[0].device
        # ... and the per-model launch-sequence switches: toggling one after the first replay must not keep the old capture
        key = key + tuple((tuple(t.shape), t.dtype) for t in inputs) + (torch.cuda.current_stream(dev).cuda_stream,
                                                                       self.upconv_phases, getattr(self, "fuse_gn_bwd", None))
        entry = graphs.get(key)
        if entry is None:
            static_in = [t.clone() for t in inputs]
            cur = torch.cuda.current_stream(dev)
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(cur)
            with torch.cuda.stream(side):  # first calls size per-kernel attributes; they must not land in the capture
                for _ in range(2):
                    fn(*static_in)
            cur.wait_stream(side)
            reserved0, free0 = torch.cuda.memory_reserved(dev), torch.cuda.mem_get_info(dev)[0]
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                out = fn(*static_in)
            pool = _graph_pool_bytes(graph, dev, reserved0, free0)
            self._pool_bytes_seen = max(self._pool_bytes_seen, pool)
            entry = graphs[key] = (graph, static_in, out, pool)
            budget = self._graph_budget()
            while len(graphs) > 1 and (len(graphs) > self.GRAPH_CACHE or sum(e[3] for e in graphs.values()) > budget):
                graphs.popitem(last=False)
            if self._planned_sets * pool > budget:
                # this candidate's launch sequences do not fit together: replay this one, then go eager (plan_graphs logs why)
                for s_, t in zip(static_in, inputs):
                    s_.copy_(t)
                graph.replay()
                res = tuple(o.clone() for o in out) if isinstance(out, tuple) else out.clone()
                graphs.pop(key, None)
                self.plan_graphs(self._planned_sets)
                return res
        else:
            graphs.move_to_end(key)
        graph, static_in, out = entry[:3]
        for s_, t in zip(static_in, inputs):
            s_.copy_(t)
        graph.replay()
        return tuple(o.clone() for o in out) if isinstance(out, tuple) else out.clone()

    # ------------------------------------------------------------------ weight preparation
    def _prepare(self):
        P = self._params
        dev = self.device
        if dev.type != "cuda":
            raise AdmError("UNetModel: parameters are on the CPU; call .to(device) first "
                           "(the HIP path has no CPU fallback)")
        pr = _Prep()
        cd = self.compute_dtype
        pack = lambda w: ops.pack_conv_weight(w, cd)          # noqa: E731
        pack_bwd = lambda w: ops.pack_conv_weight_bwd(w, cd)  # noqa: E731
        f32 = lambda k: P[k].to(torch.float32).contiguous()  # noqa: E731
        pr.te0_w, pr.te0_b = f32("time_embed.0.weight"), f32("time_embed.0.bias")
        pr.te2_w, pr.te2_b = f32("time_embed.2.weight"), f32("time_embed.2.bias")
        pr.label = f32("label_emb.weight") if self.plan.num_classes is not None else None
        # all emb_layers in one [sum(2*cout), emb_dim] matrix; per-block column offsets
        ws, bs, off = [], [], 0
        pr.film_off: Dict[str, int] = {}
        pr.blocks: Dict[str, dict] = {}
        for b in self.plan.all_blocks():
            p = b.prefix
            if isinstance(b, StemSpec):
                # stem on the MFMA conv kernel: input channels zero-padded to one 32-channel chunk
                w = P[f"{p}.weight"].to(torch.float32)
                wpad = torch.zeros((b.cout, 32, 3, 3), dtype=torch.float32, device=dev)
                wpad[:, :b.cin] = w
                pr.blocks[p] = dict(w=pack(wpad), b=f32(f"{p}.bias"))
            elif isinstance(b, ResBlockSpec):
                ws.append(f32(f"{p}.emb_layers.1.weight"))
                bs.append(f32(f"{p}.emb_layers.1.bias"))
                pr.film_off[p] = off
                off += (2 if b.scale_shift else 1) * b.cout   # (scale | shift), or the additive embedding of use_scale_shift_norm=False
                d = dict(
                    g1=f32(f"{p}.in_layers.0.weight"), b1=f32(f"{p}.in_layers.0.bias"),
                    w1=pack(P[f"{p}.in_layers.2.weight"]), c1b=f32(f"{p}.in_layers.2.bias"),
                    g2=f32(f"{p}.out_layers.0.weight"), b2=f32(f"{p}.out_layers.0.bias"),
                    w2=pack(P[f"{p}.out_layers.3.weight"]), c2b=f32(f"{p}.out_layers.3.bias"),
                )
                if b.has_skip_conv:
                    d["ws"] = pack(P[f"{p}.skip_connection.weight"])
                    d["wsb"] = f32(f"{p}.skip_connection.bias")
                    if self.fold_skip and not (b.up or b.down):   # skip_connection as extra K-steps of the out_layers conv (ops.conv(fold=))
                        d["w2f"] = ops.fold_weights(d["w2"], d["ws"])
                        d["c2fb"] = (d["c2b"] + d["wsb"]).contiguous()
                elif b.up:   # up-ResBlock: the first conv reads a 2x upsample -> four 2x2-tap phase convs (ops.pack_conv_weight_up)
                    d["w1_up"] = ops.pack_conv_weight_up(P[f"{p}.in_layers.2.weight"], cd)
                pr.blocks[p] = d
            elif isinstance(b, ResampleSpec):
                if b.use_conv and b.down:     # 3x3 stride-2 conv at its own 9 taps per output pixel: the general conv2d kernel
                    pr.blocks[p] = dict(w2d=ops.pack_conv2d_weight(P[f"{p}.op.weight"], None, cd), b=f32(f"{p}.op.bias"))
                elif b.use_conv:              # conv3x3(nearest 2x): the virtual-upsample conv, as four 2x2-tap phase convs
                    pr.blocks[p] = dict(w=pack(P[f"{p}.conv.weight"]), b=f32(f"{p}.conv.bias"),
                                        w_up=ops.pack_conv_weight_up(P[f"{p}.conv.weight"], cd))
            elif isinstance(b, AttnSpec):
                pr.blocks[p] = dict(
                    g=f32(f"{p}.norm.weight"), b=f32(f"{p}.norm.bias"),
                    wqkv=pack(P[f"{p}.qkv.weight"]), bqkv=f32(f"{p}.qkv.bias"),
                    wproj=pack(P[f"{p}.proj_out.weight"]), bproj=f32(f"{p}.proj_out.bias"),
                )
        pr.film_w = torch.cat(ws, dim=0).contiguous()
        pr.film_b = torch.cat(bs, dim=0).contiguous()
        pr.film_total = off
        if self.with_backward:
            zmax = 0
            for b in self.plan.all_blocks():
                p = b.prefix
                d = pr.blocks[p]
                if isinstance(b, StemSpec):
                    # the LAST backward-data conv: it undoes EncoderUNetModel.grad_scale in its fp32 epilogue (adm_conv_args.out_scale),
                    # not in these 16-bit weights -- 2^-10 would push every |w| < 2^-4 of an fp16 image below the smallest normal
                    d["w_bwd"] = pack_bwd(P[f"{p}.weight"])
                    zmax = max(zmax, b.cin, b.cout)
                elif isinstance(b, ResBlockSpec):
                    d["w1_bwd"] = pack_bwd(P[f"{p}.in_layers.2.weight"])
                    d["w2_bwd"] = pack_bwd(P[f"{p}.out_layers.3.weight"])
                    if b.has_skip_conv:
                        d["ws_bwd"] = pack_bwd(P[f"{p}.skip_connection.weight"])
                    zmax = max(zmax, b.cin, b.cout)
                elif isinstance(b, ResampleSpec):   # Downsample conv of a classifier_resblock_updown=False classifier
                    d["w_bwd"] = pack_bwd(P[f"{p}.op.weight"])
                    zmax = max(zmax, b.channels)
                elif isinstance(b, AttnSpec):
                    d["wqkv_bwd"] = pack_bwd(P[f"{p}.qkv.weight"])
                    d["wproj_bwd"] = pack_bwd(P[f"{p}.proj_out.weight"])
                    zmax = max(zmax, 3 * b.channels)
            pr.zero_bias = torch.zeros(zmax, dtype=torch.float32, device=dev)
        self._prepare_head(pr, P, f32)
        self._packed = pr
        return pr

    def _prepare_head(self, pr, P, f32):
        pass

    # ------------------------------------------------------------------ blocks
    def _resblock(self, pr, s: ResBlockSpec, x0, x1, film, skipped, tape=None):
        d = pr.blocks[s.prefix]
        mode = "up" if s.up else ("down" if s.down else None)
        if skipped:  # dynamic_unet.py:245-250: body bypassed, x_upd + skip_connection kept
            xs = ops.resample(x0, mode) if mode else x0
            if s.has_skip_conv:
                return ops.conv(xs, d["ws"], d["wsb"], s.cout, 1, x1=x1, want_stats=True)
            return xs
        st1 = st2 = None
        if tape is not None:
            a1, b1, st1 = ops.gn_affine(x0, d["g1"], d["b1"], x1, want_stats=True)
            aff1 = (a1, b1)
        else:
            aff1 = ops.gn_affine(x0, d["g1"], d["b1"], x1)
        virtual_up = (mode == "up" and not s.has_skip_conv and tape is None and x0.shape[1] >= 8
                      and not os.environ.get("ADM_NO_VIRTUAL_UP"))  # A/B switch for measurements
        if virtual_up:
            # h_upd(in_layers[:-1](x)) and x_upd(x) are never materialised: both convs read the half-resolution
            # tensor through a nearest-neighbour 2x upsample (adm_conv_args.in_up / res_up)
            assert x1 is None
            h = ops.conv(x0, d["w1"], d["c1b"], s.cout, 9, aff=aff1, silu=True, in_up=True, want_stats=True,
                         w_up=d.get("w1_up") if self.upconv_phases else None)
            xs, xs1 = x0, None
        elif mode:
            assert x1 is None
            h_in = ops.resample(x0, mode, aff1)
            xs = ops.resample(x0, mode)
            h = ops.conv(h_in, d["w1"], d["c1b"], s.cout, 9, want_stats=True)
            xs1 = None
        else:
            h = ops.conv(x0, d["w1"], d["c1b"], s.cout, 9, x1=x1, aff=aff1, silu=True, want_stats=True)
            xs, xs1 = x0, x1
        off = pr.film_off[s.prefix]
        if tape is not None:
            add2 = None
            if s.scale_shift:
                a2, b2, st2 = ops.gn_affine(h, d["g2"], d["b2"], film=film[:, off:], film_stride=pr.film_total, want_stats=True)
            else:   # out_layers(h + emb_out): the backward pass needs e to correct the stored-h sums (ops.gn_bwd(norm_add=))
                add2 = film[:, off:off + s.cout]
                a2, b2, st2 = ops.gn_affine(h, d["g2"], d["b2"], add=add2, want_stats=True)
            aff2 = (a2, b2)
            tape.append(("res", s, dict(x=x0, aff1=aff1, st1=st1, h1=h, aff2=aff2, st2=st2, add2=add2)))
        elif not s.scale_shift:
            # use_scale_shift_norm=False (reference unet.py:251-254): out_layers(h + emb_out).  h + e is never written: the statistics
            # of x + e[n, c] follow from the conv epilogue's per-channel sums and e folds into the next conv's prologue affine
            aff2 = ops.gn_affine(h, d["g2"], d["b2"], add=film[:, off:off + s.cout])
        else:
            aff2 = ops.gn_affine(h, d["g2"], d["b2"], film=film[:, off:], film_stride=pr.film_total)
        if s.has_skip_conv:
            if "w2f" in d and mode is None and ops.fold_ok(h.shape[1], h.shape[2]):
                # `self.skip_connection(x) + h` (reference unet.py:256) inside the out_layers conv: no 1x1 launch, no residual operand
                return ops.conv(h, d["w2f"], d["c2fb"], s.cout, 9, aff=aff2, silu=True, fold=(xs, xs1), want_stats=True)
            res = ops.conv(xs, d["ws"], d["wsb"], s.cout, 1, x1=xs1)
        else:
            res = xs
        return ops.conv(h, d["w2"], d["c2b"], s.cout, 9, aff=aff2, silu=True, res=res, res_up=virtual_up, want_stats=True)

    def _resample(self, pr, s: ResampleSpec, x):
        """Downsample / Upsample of a resblock_updown=False model (reference unet.py:78-141)."""
        if not s.use_conv:
            return ops.resample(x, "down" if s.down else "up")
        d = pr.blocks[s.prefix]
        if s.down:
            return ops.conv2d(x, d["w2d"], d["b"], 3, 3, stride=2, pad=(1, 1), relu=False)
        if x.shape[1] >= 8 and x.shape[2] >= 8:
            return ops.conv(x, d["w"], d["b"], s.channels, 9, in_up=True, want_stats=True,
                            w_up=d["w_up"] if self.upconv_phases else None)
        return ops.conv(ops.resample(x, "up"), d["w"], d["b"], s.channels, 9, want_stats=True)   # maps below 8x8: materialised

    def _attention(self, pr, s: AttnSpec, x, skipped, tape=None):
        if skipped:  # dynamic_unet.py:316-318
            return x
        d = pr.blocks[s.prefix]
        n, hh, ww, c = x.shape
        if tape is not None:
            a_, b_, st = ops.gn_affine(x, d["g"], d["b"], want_stats=True)
            aff = (a_, b_)
        else:
            aff = ops.gn_affine(x, d["g"], d["b"])
        qkv = ops.conv(x, d["wqkv"], d["bqkv"], 3 * c, 1, aff=aff, silu=False)
        if tape is not None:
            a, lse = ops.attention(qkv.view(n, hh * ww, 3 * c), s.num_heads, s.new_order, want_lse=True)
            tape.append(("attn", s, dict(x=x, aff=aff, st=st, qkv=qkv, a=a, lse=lse)))
        else:
            a = ops.attention(qkv.view(n, hh * ww, 3 * c), s.num_heads, s.new_order)
        return ops.conv(a.view(n, hh, ww, c), d["wproj"], d["bproj"], c, 1, res=x, want_stats=True)

    def _run_seq(self, pr, seq, h, skip, film, skip_ids, x_nchw=None, tape=None):
        first = True
        for blk in seq:
            if isinstance(blk, StemSpec):
                d = pr.blocks[blk.prefix]
                h = ops.conv(ops.nchw_to_nhwc_pad(x_nchw, 32, self.compute_dtype), d["w"], d["b"], blk.cout, 9, want_stats=True)
                if tape is not None:
                    tape.append(("stem", blk, {}))
            elif isinstance(blk, ResBlockSpec):
                h = self._resblock(pr, blk, h, skip if first else None, film, blk.layer_id in skip_ids, tape)
            elif isinstance(blk, ResampleSpec):
                h = self._resample(pr, blk, h)
                if tape is not None:
                    tape.append(("down", blk, {}))
            else:
                h = self._attention(pr, blk, h, blk.layer_id in skip_ids, tape)
            first = False
        return h

    # ------------------------------------------------------------------ forward
    def _embed(self, pr, timesteps, y):
        plan = self.plan
        assert (y is not None) == (plan.num_classes is not None), \
            "must specify y if and only if the model is class-conditional"
        e = ops.timestep_embedding(timesteps, plan.model_channels)
        e = ops.linear_f32(e, pr.te0_w, pr.te0_b)
        if y is not None:
            assert y.shape == (timesteps.shape[0],)
            e = ops.linear_f32(e, pr.te2_w, pr.te2_b, silu_in=True, table=pr.label,
                               idx=y.to(torch.int64).contiguous())
        else:
            e = ops.linear_f32(e, pr.te2_w, pr.te2_b, silu_in=True)
        return ops.linear_f32(e, pr.film_w, pr.film_b, silu_in=True)  # every block's (scale | shift)


class UNetModel(AdmNet):
    def __init__(self, plan: UNetPlan, use_fp16: bool = False):
        if plan.encoder_only:
            raise ValueError("use EncoderUNetModel for classifier plans")
        super().__init__(plan, use_fp16)

    def _prepare_head(self, pr, P, f32):
        h = self.plan.head
        pr.head = dict(g=f32(f"{h.prefix}.0.weight"), b=f32(f"{h.prefix}.0.bias"),
                       w=ops.pack_conv_weight(P[f"{h.prefix}.2.weight"], self.compute_dtype), cb=f32(f"{h.prefix}.2.bias"))

    def forward(self, x, timesteps, y=None, skip_layer: Sequence[int] = ()):
        """x fp32 [N,C,H,W], timesteps [N] (original-process timesteps), y int64 [N] or None."""
        pr = self._packed or self._prepare()
        if not x.is_cuda:
            raise AdmError("UNetModel.forward: x must be a device tensor (no CPU fallback)")
        skip_ids = set(int(s) for s in skip_layer) if self.plan.dynamic else set()
        if skip_layer and not self.plan.dynamic:
            raise TypeError("skip_layer needs a dynamic UNet (use_dynamic_unet=True)")
        x = x.to(torch.float32).contiguous()
        if self.use_graph and not self._graph_eager and ops.CONV_PROFILE is None and timesteps.is_cuda:
            ins = (x, timesteps.contiguous()) + (() if y is None else (y.contiguous(),))
            return self._graphed(("unet", tuple(sorted(skip_ids))),
                                 lambda x_, t_, y_=None: self._forward(pr, x_, t_, y_, skip_ids), ins)
        return self._forward(pr, x, timesteps, y, skip_ids)

    def _forward(self, pr, x, timesteps, y, skip_ids):
        with torch.no_grad():
            film = self._embed(pr, timesteps, y)
            hs: List[torch.Tensor] = []
            h = None
            for seq in self.plan.input_blocks:
                h = self._run_seq(pr, seq, h, None, film, skip_ids, x_nchw=x)
                hs.append(h)
            h = self._run_seq(pr, self.plan.middle_block, h, None, film, skip_ids)
            for seq in self.plan.output_blocks:
                h = self._run_seq(pr, seq, h, hs.pop(), film, skip_ids)
            hd = pr.head
            aff = ops.gn_affine(h, hd["g"], hd["b"])
            return ops.conv(h, hd["w"], hd["cb"], self.plan.out_channels, 9, aff=aff, silu=True,
                            out_f32_nchw=True)


Dynamic_UNetModel = UNetModel  # the plan's `dynamic` flag carries the difference
