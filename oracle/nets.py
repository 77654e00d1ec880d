"""Oracle: ADM UNet / dynamic UNet / classifier forward on the CPU (TEST INFRASTRUCTURE).

PyTorch-CPU float32 functional restatement, driven by the same architecture
plan as the HIP engine (``autodiffusion_amd.arch``).  Parameters are a plain
``{reference state-dict key: float32 tensor}`` mapping.

Restates:
  * ``timestep_embedding`` -- reference guided_diffusion/nn.py:103-121
  * ``GroupNorm32`` -- nn.py:17-19 (float32 statistics, 32 groups, eps 1e-5)
  * ``ResBlock._forward`` -- unet.py:236-256 (+ dynamic skip branch,
    dynamic_unet.py:245-250); ``Downsample`` / ``Upsample`` -- unet.py:78-141
  * ``AttentionBlock._forward`` -- unet.py:299-305 (+ dynamic_unet.py:316-318)
  * ``QKVAttention`` / ``QKVAttentionLegacy`` -- unet.py:361-393 / 328-358
  * ``UNetModel.forward`` -- unet.py:634-665;  ``Dynamic_UNetModel.forward``
    -- dynamic_unet.py:673-702
  * ``EncoderUNetModel.forward`` + ``AttentionPool2d`` -- unet.py:873-896, 22-51
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Sequence

import torch
import torch.nn.functional as F

from autodiffusion_amd.arch import (AttnPoolSpec, AttnSpec, HeadSpec, PoolHeadSpec, ResBlockSpec, ResampleSpec, StemSpec,
                                    UNetPlan, GN_GROUPS)

Params = Dict[str, torch.Tensor]


def sinusoid_embedding(t: torch.Tensor, dim: int, max_period: float = 10000.0) -> torch.Tensor:
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(half, dtype=torch.float32) / half)
    ang = t[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(ang), torch.sin(ang)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


def group_norm(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    return F.group_norm(x.float(), GN_GROUPS, w, b, eps=1e-5)


def _silu(x):
    return x * torch.sigmoid(x)


def _down(x):
    return F.avg_pool2d(x, kernel_size=2, stride=2)


def _up(x):
    return F.interpolate(x, scale_factor=2, mode="nearest")


def resblock(P: Params, s: ResBlockSpec, x: torch.Tensor, emb: torch.Tensor,
             skipped: bool = False) -> torch.Tensor:
    p = s.prefix

    def skip_path(v):
        if s.has_skip_conv:
            return F.conv2d(v, P[f"{p}.skip_connection.weight"], P[f"{p}.skip_connection.bias"])
        return v

    resample = _up if s.up else (_down if s.down else None)
    if skipped:  # layer-skip search space: body bypassed, skip path kept
        return skip_path(resample(x) if resample else x)
    h = _silu(group_norm(x, P[f"{p}.in_layers.0.weight"], P[f"{p}.in_layers.0.bias"]))
    if resample is not None:
        h = resample(h)
        x = resample(x)
    h = F.conv2d(h, P[f"{p}.in_layers.2.weight"], P[f"{p}.in_layers.2.bias"], padding=1)
    e = F.linear(_silu(emb), P[f"{p}.emb_layers.1.weight"], P[f"{p}.emb_layers.1.bias"])
    e = e[:, :, None, None]
    if s.scale_shift:
        scale, shift = torch.chunk(e, 2, dim=1)
        h = group_norm(h, P[f"{p}.out_layers.0.weight"], P[f"{p}.out_layers.0.bias"]) * (1 + scale) + shift
    else:
        h = group_norm(h + e, P[f"{p}.out_layers.0.weight"], P[f"{p}.out_layers.0.bias"])
    h = F.conv2d(_silu(h), P[f"{p}.out_layers.3.weight"], P[f"{p}.out_layers.3.bias"], padding=1)
    return skip_path(x) + h


def qkv_attention(qkv: torch.Tensor, n_heads: int, new_order: bool) -> torch.Tensor:
    """qkv [N, 3*H*C, T] -> [N, H*C, T]; softmax in float32; scale on q and k."""
    bs, width, length = qkv.shape
    ch = width // (3 * n_heads)
    if new_order:
        q, k, v = qkv.chunk(3, dim=1)
        q = q.reshape(bs * n_heads, ch, length)
        k = k.reshape(bs * n_heads, ch, length)
        v = v.reshape(bs * n_heads, ch, length)
    else:
        q, k, v = qkv.reshape(bs * n_heads, ch * 3, length).split(ch, dim=1)
    scale = 1.0 / math.sqrt(math.sqrt(ch))
    w = torch.einsum("bct,bcs->bts", q * scale, k * scale)
    w = torch.softmax(w.float(), dim=-1).type(w.dtype)
    a = torch.einsum("bts,bcs->bct", w, v)
    return a.reshape(bs, -1, length)


def attention_block(P: Params, s: AttnSpec, x: torch.Tensor, skipped: bool = False) -> torch.Tensor:
    if skipped:
        return x
    p = s.prefix
    b, c, *spatial = x.shape
    xf = x.reshape(b, c, -1)
    qkv = F.conv1d(group_norm(xf, P[f"{p}.norm.weight"], P[f"{p}.norm.bias"]),
                   P[f"{p}.qkv.weight"], P[f"{p}.qkv.bias"])
    h = qkv_attention(qkv, s.num_heads, s.new_order)
    h = F.conv1d(h, P[f"{p}.proj_out.weight"], P[f"{p}.proj_out.bias"])
    return (xf + h).reshape(b, c, *spatial)


def time_embedding(P: Params, plan: UNetPlan, t: torch.Tensor, y: Optional[torch.Tensor]):
    e = sinusoid_embedding(t, plan.model_channels)
    e = F.linear(e, P["time_embed.0.weight"], P["time_embed.0.bias"])
    e = F.linear(_silu(e), P["time_embed.2.weight"], P["time_embed.2.bias"])
    if plan.num_classes is not None:
        assert y is not None and y.shape == (t.shape[0],)
        e = e + P["label_emb.weight"][y]
    else:
        assert y is None, "must specify y if and only if the model is class-conditional"
    return e


def _run_seq(P, seq, h, emb, skip_ids):
    for blk in seq:
        if isinstance(blk, StemSpec):
            h = F.conv2d(h, P[f"{blk.prefix}.weight"], P[f"{blk.prefix}.bias"], padding=1)
        elif isinstance(blk, ResBlockSpec):
            h = resblock(P, blk, h, emb, skipped=blk.layer_id in skip_ids)
        elif isinstance(blk, AttnSpec):
            h = attention_block(P, blk, h, skipped=blk.layer_id in skip_ids)
        elif isinstance(blk, ResampleSpec):   # Downsample / Upsample of resblock_updown=False models: reference unet.py:78-141
            if blk.down:
                h = (F.conv2d(h, P[f"{blk.prefix}.op.weight"], P[f"{blk.prefix}.op.bias"], stride=2, padding=1)
                     if blk.use_conv else _down(h))
            else:
                h = _up(h)
                if blk.use_conv:
                    h = F.conv2d(h, P[f"{blk.prefix}.conv.weight"], P[f"{blk.prefix}.conv.bias"], padding=1)
        else:
            raise TypeError(blk)
    return h


def attention_pool(P: Params, s: AttnPoolSpec, x: torch.Tensor) -> torch.Tensor:
    p = f"{s.prefix}.2"
    b, c = x.shape[:2]
    x = x.reshape(b, c, -1)
    x = torch.cat([x.mean(dim=-1, keepdim=True), x], dim=-1)
    x = x + P[f"{p}.positional_embedding"][None]
    x = F.conv1d(x, P[f"{p}.qkv_proj.weight"], P[f"{p}.qkv_proj.bias"])
    x = qkv_attention(x, s.num_heads, new_order=True)
    x = F.conv1d(x, P[f"{p}.c_proj.weight"], P[f"{p}.c_proj.bias"])
    return x[:, :, 0]


def unet_forward(P: Params, plan: UNetPlan, x: torch.Tensor, t: torch.Tensor,
                 y: Optional[torch.Tensor] = None, skip_layer: Sequence[int] = ()) -> torch.Tensor:
    """x [N,C,H,W] float32, t [N] (already mapped to original timesteps) -> [N,out,H,W]."""
    skip_ids = set(skip_layer) if plan.dynamic else set()
    emb = time_embedding(P, plan, t, y)
    h = x.float()
    hs = []
    for seq in plan.input_blocks:
        h = _run_seq(P, seq, h, emb, skip_ids)
        hs.append(h)
    h = _run_seq(P, plan.middle_block, h, emb, skip_ids)
    if plan.encoder_only:
        head = plan.head
        if isinstance(head, PoolHeadSpec):   # reference unet.py:826-856, 880-896
            p = head.prefix
            if head.kind == "adaptive":
                h = _silu(group_norm(h, P[f"{p}.0.weight"], P[f"{p}.0.bias"])).mean(dim=(2, 3), keepdim=True)
                return F.conv2d(h, P[f"{p}.3.weight"], P[f"{p}.3.bias"]).flatten(1)
            f = torch.cat([r.mean(dim=(2, 3)) for r in hs] + [h.mean(dim=(2, 3))], dim=-1)   # every input block (the stem is block 0) + the middle
            z = F.linear(f, P[f"{p}.0.weight"], P[f"{p}.0.bias"])
            if head.kind == "spatial":
                return F.linear(F.relu(z), P[f"{p}.2.weight"], P[f"{p}.2.bias"])
            return F.linear(_silu(group_norm(z, P[f"{p}.1.weight"], P[f"{p}.1.bias"])), P[f"{p}.3.weight"], P[f"{p}.3.bias"])
        h = _silu(group_norm(h, P[f"{head.prefix}.0.weight"], P[f"{head.prefix}.0.bias"]))
        return attention_pool(P, head, h)
    for seq in plan.output_blocks:
        h = torch.cat([h, hs.pop()], dim=1)
        h = _run_seq(P, seq, h, emb, skip_ids)
    head: HeadSpec = plan.head
    h = _silu(group_norm(h, P[f"{head.prefix}.0.weight"], P[f"{head.prefix}.0.bias"]))
    return F.conv2d(h, P[f"{head.prefix}.2.weight"], P[f"{head.prefix}.2.bias"], padding=1)


def classifier_grad(P: Params, plan: UNetPlan, x: torch.Tensor, t: torch.Tensor,
                    y: torch.Tensor, scale: float = 1.0) -> torch.Tensor:
    """cond_fn: scale * d/dx sum_n log_softmax(f(x,t))[n, y_n].

    Reference search_imagenet64_classifier_guidance.py:319-326.
    """
    with torch.enable_grad():
        x_in = x.detach().requires_grad_(True)
        logits = unet_forward(P, plan, x_in, t)
        logp = F.log_softmax(logits, dim=-1)
        sel = logp[range(len(logits)), y.view(-1)]
        return torch.autograd.grad(sel.sum(), x_in)[0] * scale


def params_from_numpy(sd) -> Params:
    return {k: torch.from_numpy(v.copy()).float() for k, v in sd.items()}
