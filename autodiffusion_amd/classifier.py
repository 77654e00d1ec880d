"""Noisy-image classifier (EncoderUNetModel) forward and its input gradient on HIP kernels.

Mirror of the reference classifier interface (``EncoderUNetModel`` guided_diffusion/unet.py:685-896,
``AttentionPool2d`` :22-51, built by ``create_classifier`` script_util.py:257-295) plus the one
quantity candidate evaluation needs from it -- the guidance gradient of ``cond_fn``
(search_imagenet64_classifier_guidance.py:319-326):

    log_prob_grad(x, t, y, s) = s * d/dx sum_n log_softmax(f(x, t))[n, y_n]

The reference runs torch.autograd; here the backward network is explicit (data gradients only):
every conv's backward-data is the same fused MFMA conv kernel with transposed, tap-flipped weights;
GroupNorm(+FiLM)+SiLU, attention and the attention pool have dedicated backward kernels
(csrc/adm_backward.hip, csrc/adm_attention_bwd.hip).  Activations and gradients between kernels are
bf16 NHWC; accumulation, GroupNorm statistics, softmax and the logits are fp32.

Every ``create_classifier`` flag is built: the pools "adaptive" / "spatial" / "spatial_v2" (unet.py:826-856, 880-896;
csrc/adm_clfhead.hip), ``classifier_use_scale_shift_norm=False`` (GroupNorm of h + emb with only h stored) and
``classifier_resblock_updown=False`` (the stride-2 Downsample conv: backward = zero-insert + transposed, flipped conv).
"""
from __future__ import annotations

import torch

from . import ops
from ._lib import AdmError
from .arch import AttnPoolSpec, PoolHeadSpec, UNetPlan
from .unet import AdmNet


class EncoderUNetModel(AdmNet):
    with_backward = True
    fuse_gn_bwd = ops.FUSE_GN_BWD   # a per-model choice (never by batch): the fused sums are taken per 256-pixel tile

    def __init__(self, plan: UNetPlan, use_fp16: bool = False):
        if not plan.encoder_only or not isinstance(plan.head, (AttnPoolSpec, PoolHeadSpec)):
            raise ValueError("EncoderUNetModel needs an encoder plan with a pool head")
        super().__init__(plan, use_fp16)

    # The backward-data network's tensors are tiny: d(logits) <= 1, and 40 layers later the medians sit at 4e-6 .. 5e-4, with up to
    # 100 % of a tensor's elements below fp16's smallest normal (6.1e-5; tools/mixed_torso_probe.py, profiles/r03/mixed_torso_probe.log).
    # bf16 has fp32's exponent range and needs nothing.  With an fp16 torso (set_torso("fp16"): 11 mantissa bits, the guidance
    # gradient's error against the reference's autograd 2.0e-2 -> 4e-3) the whole chain is LINEAR in d(logits), so it runs on
    # d(logits) * 2^10 -- every tensor back in the normal range, max |g| * 2^10 ~ 3, far from 65504 -- and the last backward conv (the
    # stem's, which writes the fp32 NCHW gradient) multiplies by 2^-10 in its fp32 epilogue (adm_conv_args.out_scale): exact, no extra
    # pass, no dynamic loss-scale state, and the stem's fp16 weights stay in the normal range (folding 2^-10 into THEM, as round 3
    # did, made every |w| < 2^-4 subnormal).
    FP16_GRAD_SCALE = 1024.0

    @property
    def grad_scale(self):
        return self.FP16_GRAD_SCALE if self.compute_dtype == torch.float16 else 1.0

    # ------------------------------------------------------------------ head weights
    def _prepare_head(self, pr, P, f32):
        h = self.plan.head
        p = h.prefix
        if isinstance(h, PoolHeadSpec):
            t = lambda k: f32(k).t().contiguous()   # noqa: E731  (backward-data: the transposed Linear)
            if h.kind == "adaptive":
                wc = f32(f"{p}.3.weight").reshape(h.out_dim, h.channels).contiguous()
                pr.head = dict(g=f32(f"{p}.0.weight"), b=f32(f"{p}.0.bias"), wc=wc, bc=f32(f"{p}.3.bias"), wc_t=wc.t().contiguous())
            else:
                last = "2" if h.kind == "spatial" else "3"
                pr.head = dict(w0=f32(f"{p}.0.weight"), b0=f32(f"{p}.0.bias"), w0_t=t(f"{p}.0.weight"),
                               wc=f32(f"{p}.{last}.weight"), bc=f32(f"{p}.{last}.bias"), wc_t=t(f"{p}.{last}.weight"))
                if h.kind == "spatial_v2":
                    pr.head.update(g1=f32(f"{p}.1.weight"), b1=f32(f"{p}.1.bias"))
            return
        wc = f32(f"{p}.2.c_proj.weight").reshape(h.out_dim, h.channels).contiguous()
        pr.head = dict(
            g=f32(f"{p}.0.weight"), b=f32(f"{p}.0.bias"), pos=f32(f"{p}.2.positional_embedding"),
            wqkv=ops.pack_conv_weight(P[f"{p}.2.qkv_proj.weight"], self.compute_dtype), bqkv=f32(f"{p}.2.qkv_proj.bias"),
            wqkv_bwd=ops.pack_conv_weight_bwd(P[f"{p}.2.qkv_proj.weight"], self.compute_dtype),
            wc=wc, bc=f32(f"{p}.2.c_proj.bias"), wc_t=wc.t().contiguous(),
        )
        pr.zero_bias = torch.zeros(max(pr.zero_bias.numel(), 3 * h.channels), dtype=torch.float32,
                                   device=pr.zero_bias.device)

    # ------------------------------------------------------------------ forward
    def _features(self, pr, x, timesteps, tape):
        """-> (final map, feat): feat = the concatenated per-block channel means of the spatial pools (unet.py:884-893), else None."""
        film = self._embed(pr, timesteps, None)
        hs = self.plan.head
        spatial = isinstance(hs, PoolHeadSpec) and hs.kind.startswith("spatial")
        feat = torch.empty((x.shape[0], hs.feature_size), dtype=torch.float32, device=x.device) if spatial else None
        col = 0

        def keep(h):
            nonlocal col
            if spatial:
                ops.channel_mean(h, out=feat, col=col)
                if tape is not None:
                    tape.append(("feat", None, dict(col=col, shape=tuple(h.shape), dtype=h.dtype)))
                col += h.shape[3]
        h = None
        for seq in self.plan.input_blocks:
            h = self._run_seq(pr, seq, h, None, film, (), x_nchw=x, tape=tape)
            keep(h)
        h = self._run_seq(pr, self.plan.middle_block, h, None, film, (), tape=tape)
        keep(h)
        assert not spatial or col == hs.feature_size
        return h, feat

    def _pool_head_forward(self, pr, hs: PoolHeadSpec, h, feat, tape):
        hd = pr.head
        if hs.kind == "adaptive":     # GN -> SiLU -> AdaptiveAvgPool2d((1, 1)) -> conv1x1 -> Flatten
            if tape is not None:
                a_, b_, st = ops.gn_affine(h, hd["g"], hd["b"], want_stats=True)
                tape.append(("adaptive", hs, dict(h=h, aff=(a_, b_), st=st)))
            else:
                a_, b_ = ops.gn_affine(h, hd["g"], hd["b"])
            return ops.linear_f32(ops.channel_mean(h, (a_, b_)), hd["wc"], hd["bc"])
        z1 = ops.linear_f32(feat, hd["w0"], hd["b0"])
        if hs.kind == "spatial":      # Linear -> ReLU -> Linear
            if tape is not None:
                tape.append(("spatial", hs, dict(z1=z1)))
            return ops.linear_f32(ops.vec_act(z1, "relu"), hd["wc"], hd["bc"])
        y, st = ops.vec_gn(z1, hd["g1"], hd["b1"])    # Linear -> GroupNorm32(32, 2048) -> SiLU -> Linear
        if tape is not None:
            tape.append(("spatial_v2", hs, dict(z1=z1, y=y, st=st)))
        return ops.linear_f32(y, hd["wc"], hd["bc"], silu_in=True)

    def _head_forward(self, pr, h, tape, feat=None):
        hs = self.plan.head
        if isinstance(hs, PoolHeadSpec):
            return self._pool_head_forward(pr, hs, h, feat, tape)
        hd = pr.head
        n, hh, ww, c = h.shape
        if hh * ww != 64:
            raise AdmError(f"attention pool: {hh}x{ww} final map unsupported (reference classifiers end at 8x8)")
        t = hh * ww + 1
        tpad = 128
        if tape is not None:
            a_, b_, st = ops.gn_affine(h, hd["g"], hd["b"], want_stats=True)
        else:
            a_, b_ = ops.gn_affine(h, hd["g"], hd["b"])
            st = None
        tok = ops.pool_prep(h, (a_, b_), hd["pos"], tpad)
        # qkv_proj as a 1x1 conv over the padded token rows, viewed as 8x8 maps (rows are independent)
        qkv = ops.conv(tok.view(n * tpad // 64, 8, 8, c), hd["wqkv"], hd["bqkv"], 3 * c, 1)
        qkv = qkv.view(n, tpad, 3 * c)
        a0, wts = ops.pool_attn_fwd(qkv, t, hs.num_heads)
        logits = ops.linear_f32(a0, hd["wc"], hd["bc"])
        if tape is not None:
            tape.append(("pool", hs, dict(h=h, aff=(a_, b_), st=st, qkv=qkv, wts=wts, t=t, tpad=tpad)))
        return logits

    def forward(self, x, timesteps):
        """x fp32 [N,3,H,W], timesteps [N] -> logits fp32 [N, 1000].

        If ``x`` requires grad (inside ``th.enable_grad()``), the logits carry a torch.autograd node whose backward
        runs the explicit backward-data network: the reference's own ``cond_fn`` closure
        (search_imagenet64_classifier_guidance.py:319-326 -- ``classifier(x_in, t)``, ``log_softmax``, ``autograd.grad``)
        works unchanged on this object."""
        pr = self._packed or self._prepare()
        if not x.is_cuda:
            raise AdmError("EncoderUNetModel.forward: x must be a device tensor (no CPU fallback)")
        if torch.is_grad_enabled() and x.requires_grad:
            return _ClassifierFn.apply(x, timesteps, self)
        with torch.no_grad():
            x = x.to(torch.float32).contiguous()
            h, feat = self._features(pr, x, timesteps, None)
            return self._head_forward(pr, h, None, feat)

    # ------------------------------------------------------------------ backward-data
    def _bwd_conv(self, pr, dy, w_bwd, cout, taps):
        return ops.conv(dy, w_bwd, pr.zero_bias, cout, taps)

    def _forward_tape(self, x, timesteps):
        """Forward with the activations the backward-data network needs -> (logits, tape)."""
        pr = self._packed or self._prepare()
        with torch.no_grad():
            x = x.detach().to(torch.float32).contiguous()
            tape = []
            h, feat = self._features(pr, x, timesteps, tape)
            logits = self._head_forward(pr, h, tape, feat)
        return logits, tape

    def _backward_tape(self, tape, dl):
        """dl = d(loss)/d(logits) fp32 [N, 1000] -> d(loss)/dx fp32 [N,3,H,W] (data gradients only)."""
        pr = self._packed
        with torch.no_grad():
            g = dfeat = None
            for kind, s, t in reversed(tape):
                if kind == "adaptive":
                    hd = pr.head
                    n, hh, ww, c = t["h"].shape
                    dact = ops.bcast_add(ops.linear_f32(dl, hd["wc_t"], None), t["h"].shape, t["h"].dtype, 1.0 / (hh * ww))
                    g = ops.gn_bwd(t["h"], dact, t["aff"], t["st"], silu=True)
                elif kind == "spatial":
                    hd = pr.head
                    dfeat = ops.linear_f32(ops.vec_act(t["z1"], "relu", dy=ops.linear_f32(dl, hd["wc_t"], None)), hd["w0_t"], None)
                elif kind == "spatial_v2":
                    hd = pr.head
                    dy = ops.vec_act(t["y"], "silu", dy=ops.linear_f32(dl, hd["wc_t"], None))
                    dfeat = ops.linear_f32(ops.vec_gn_bwd(t["z1"], hd["g1"], t["st"], dy), hd["w0_t"], None)
                elif kind == "feat":     # this block's channel means fed the spatial head: d mean / d h = 1 / HW on every pixel
                    shp = t["shape"]
                    g = ops.bcast_add(dfeat, shp, t["dtype"], 1.0 / (shp[1] * shp[2]), add=g, col=t["col"])
                elif kind == "down":     # Downsample's 3x3 stride-2 conv: stride-1 backward conv over the zero-inserted gradient
                    g = self._bwd_conv(pr, ops.resample(g, "zero2"), pr.blocks[s.prefix]["w_bwd"], s.channels, 9)
                elif kind == "pool":
                    hd = pr.head
                    n, hh, ww, c = t["h"].shape
                    da0 = ops.linear_f32(dl, hd["wc_t"], None)
                    dqkv = ops.pool_attn_bwd(t["qkv"], t["wts"], da0, t["t"], s.num_heads)
                    dtok = self._bwd_conv(pr, dqkv.view(n * t["tpad"] // 64, 8, 8, 3 * c), hd["wqkv_bwd"], c, 1)
                    dact = ops.pool_prep_bwd(dtok.view(n, t["tpad"], c), hh, ww)
                    g = ops.gn_bwd(t["h"], dact, t["aff"], t["st"], silu=True)
                elif kind == "attn":
                    d = pr.blocks[s.prefix]
                    n, hh, ww, c = t["x"].shape
                    da = self._bwd_conv(pr, g, d["wproj_bwd"], c, 1)
                    dqkv = ops.attention_bwd(t["qkv"].view(n, hh * ww, 3 * c), t["a"], da.view(n, hh * ww, c),
                                             t["lse"], s.num_heads, s.new_order)
                    dgn = self._bwd_conv(pr, dqkv.view(n, hh, ww, 3 * c), d["wqkv_bwd"], c, 1)
                    g = ops.gn_bwd(t["x"], dgn, t["aff"], t["st"], silu=False, add=g)
                elif kind == "res":
                    d = pr.blocks[s.prefix]
                    # on maps >= 16x16 the backward conv's epilogue already turns its output into dz = dy * SiLU'(a x + b) and
                    # leaves the (sum dz, sum dz x) slabs: the GroupNorm backward then is finalize + apply, no partial pass
                    fuse2 = self.fuse_gn_bwd and _gnb_ok(t["h1"])
                    if fuse2:
                        dz2 = ops.conv(g, d["w2_bwd"], pr.zero_bias, s.cout, 9, gnb=(t["h1"], t["aff2"]))
                        dh1 = ops.gn_bwd(t["h1"], dz2, t["aff2"], t["st2"], silu=True, partial=dz2._adm_stats[0], norm_add=t["add2"])
                    else:
                        d_act2 = self._bwd_conv(pr, g, d["w2_bwd"], s.cout, 9)
                        dh1 = ops.gn_bwd(t["h1"], d_act2, t["aff2"], t["st2"], silu=True, norm_add=t["add2"])
                    if s.down:
                        d_in = self._bwd_conv(pr, dh1, d["w1_bwd"], s.cin, 9)
                        g = ops.gn_bwd(t["x"], d_in, t["aff1"], t["st1"], silu=True, dy_half=True, add=g, add_half=True)
                    elif s.up:
                        raise NotImplementedError("up-sampling ResBlocks do not occur in the encoder")
                    else:
                        dskip = self._bwd_conv(pr, g, d["ws_bwd"], s.cin, 1) if s.has_skip_conv else g
                        if self.fuse_gn_bwd and _gnb_ok(t["x"]):
                            dz1 = ops.conv(dh1, d["w1_bwd"], pr.zero_bias, s.cin, 9, gnb=(t["x"], t["aff1"]))
                            g = ops.gn_bwd(t["x"], dz1, t["aff1"], t["st1"], silu=True, add=dskip, partial=dz1._adm_stats[0])
                        else:
                            d_in = self._bwd_conv(pr, dh1, d["w1_bwd"], s.cin, 9)
                            g = ops.gn_bwd(t["x"], d_in, t["aff1"], t["st1"], silu=True, add=dskip)
                elif kind == "stem":
                    d = pr.blocks[s.prefix]
                    g = ops.conv(g, d["w_bwd"], pr.zero_bias, s.cin, 9, out_f32_nchw=True,
                                 out_scale=None if self.grad_scale == 1.0 else 1.0 / self.grad_scale)
            return g

    def log_prob_grad(self, x, timesteps, y, scale: float = 1.0, return_logits: bool = False):
        """scale * grad_x sum_n log_softmax(f(x,t))[n, y_n]; fp32 [N,3,H,W]."""
        if not x.is_cuda:
            raise AdmError("EncoderUNetModel.log_prob_grad: x must be a device tensor (no CPU fallback)")
        if self.use_graph and ops.CONV_PROFILE is None and timesteps.is_cuda:
            self._packed or self._prepare()
            ins = (x.detach().to(torch.float32).contiguous(), timesteps.contiguous(), y.to(torch.int64).contiguous())
            g, logits = self._graphed(("grad", float(scale)), lambda x_, t_, y_: self._log_prob_grad(x_, t_, y_, scale), ins)
            return (g, logits) if return_logits else g
        g, logits = self._log_prob_grad(x, timesteps, y, scale)
        return (g, logits) if return_logits else g

    def _log_prob_grad(self, x, timesteps, y, scale):
        logits, tape = self._forward_tape(x, timesteps)
        with torch.no_grad():
            dl = ops.logsoftmax_grad(logits, y.to(torch.int64).contiguous(), scale * self.grad_scale)
        return self._backward_tape(tape, dl), logits


def _gnb_ok(x) -> bool:
    """A GroupNorm input whose backward-data conv can carry the GroupNorm-backward epilogue (adm_conv prologue 3)."""
    _, h, w, c = x.shape
    return h >= 16 and w >= 16 and (h * w) % 256 == 0 and c % 8 == 0


class _ClassifierFn(torch.autograd.Function):
    """torch.autograd bridge over the explicit backward network: forward = the HIP classifier with its tape kept,
    backward(dlogits) = EncoderUNetModel._backward_tape.  Only d/dx exists (the engine is inference-only)."""

    @staticmethod
    def forward(ctx, x, timesteps, net):
        logits, tape = net._forward_tape(x, timesteps)
        ctx.net, ctx.tape, ctx.x_dtype = net, tape, x.dtype
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        tape, ctx.tape = ctx.tape, None  # one backward per forward, like a graph without retain_graph
        if tape is None:
            raise RuntimeError("the HIP classifier's activations were already released (backward called twice)")
        dl = dlogits.detach().to(torch.float32)
        if ctx.net.grad_scale != 1.0:
            dl = dl * ctx.net.grad_scale     # undone inside the network (the stem's backward conv scales its fp32 output by 1 / grad_scale)
        g = ctx.net._backward_tape(tape, dl.contiguous())
        return g.to(ctx.x_dtype), None, None
