"""Oracle: Stable-Diffusion latent UNet forward on the CPU (TEST INFRASTRUCTURE -- never imported by the product).

PyTorch-CPU float32 functional restatement driven by ``autodiffusion_amd.sd_arch``; parameters are a plain
``{reference state-dict key: float32 tensor}`` mapping.  Pinned by tests/golden/sd_unet_*.npz, captured from the
reference's own modules (tests/golden/capture_sd.py).

Restates (reference "Stable Diffusion"/ldm/...):
  * ``UNetModel.forward`` -- modules/diffusionmodules/openaimodel.py:710-742
  * ``ResBlock._forward`` (no scale-shift, no up/down) -- openaimodel.py:236-262
  * ``Downsample`` / ``Upsample`` (conv_resample) -- openaimodel.py:118-145 / 78-104
  * ``timestep_embedding`` -- modules/diffusionmodules/util.py:151-171;  ``GroupNorm32`` -- util.py:214-216
  * ``SpatialTransformer`` / ``BasicTransformerBlock`` / ``CrossAttention`` / ``GEGLU`` -- modules/attention.py:218-260,
    196-215, 152-194, 36-44
"""
from __future__ import annotations

from typing import Dict

import torch
import torch.nn.functional as F

from autodiffusion_amd.sd_arch import (SDDownSpec, SDResBlockSpec, SDStemSpec, SDTransformerSpec, SDUNetPlan,
                                       SDUpSpec)
from oracle.nets import _silu, group_norm, sinusoid_embedding

Params = Dict[str, torch.Tensor]


def resblock(P: Params, s: SDResBlockSpec, x, emb):
    p = s.prefix
    h = _silu(group_norm(x, P[f"{p}.in_layers.0.weight"], P[f"{p}.in_layers.0.bias"]))
    h = F.conv2d(h, P[f"{p}.in_layers.2.weight"], P[f"{p}.in_layers.2.bias"], padding=1)
    e = F.linear(_silu(emb), P[f"{p}.emb_layers.1.weight"], P[f"{p}.emb_layers.1.bias"])
    h = h + e[:, :, None, None]
    h = _silu(group_norm(h, P[f"{p}.out_layers.0.weight"], P[f"{p}.out_layers.0.bias"]))
    h = F.conv2d(h, P[f"{p}.out_layers.3.weight"], P[f"{p}.out_layers.3.bias"], padding=1)
    if s.has_skip_conv:
        x = F.conv2d(x, P[f"{p}.skip_connection.weight"], P[f"{p}.skip_connection.bias"])
    return x + h


def cross_attention(P: Params, p: str, x, context, heads: int):
    """x [B, T, C], context [B, S, Cc] -> [B, T, C]; softmax(q k^T * d_head^-0.5) v, heads split as (h d)."""
    q = F.linear(x, P[f"{p}.to_q.weight"])
    k = F.linear(context, P[f"{p}.to_k.weight"])
    v = F.linear(context, P[f"{p}.to_v.weight"])
    b, t, inner = q.shape
    d = inner // heads

    def split(z):
        return z.reshape(b, z.shape[1], heads, d).permute(0, 2, 1, 3)

    q, k, v = split(q), split(k), split(v)
    w = torch.softmax(torch.einsum("bhid,bhjd->bhij", q, k) * d ** -0.5, dim=-1)
    o = torch.einsum("bhij,bhjd->bhid", w, v).permute(0, 2, 1, 3).reshape(b, t, inner)
    return F.linear(o, P[f"{p}.to_out.0.weight"], P[f"{p}.to_out.0.bias"])


def transformer_block(P: Params, p: str, x, context, heads: int):
    c = x.shape[-1]

    def ln(z, k):
        return F.layer_norm(z, (c,), P[f"{p}.{k}.weight"], P[f"{p}.{k}.bias"], eps=1e-5)

    x = cross_attention(P, f"{p}.attn1", ln(x, "norm1"), ln(x, "norm1"), heads) + x
    x = cross_attention(P, f"{p}.attn2", ln(x, "norm2"), context, heads) + x
    u = F.linear(ln(x, "norm3"), P[f"{p}.ff.net.0.proj.weight"], P[f"{p}.ff.net.0.proj.bias"])
    a, gate = u.chunk(2, dim=-1)
    return F.linear(a * F.gelu(gate), P[f"{p}.ff.net.2.weight"], P[f"{p}.ff.net.2.bias"]) + x


def spatial_transformer(P: Params, s: SDTransformerSpec, x, context):
    p = s.prefix
    b, c, hh, ww = x.shape
    h = F.group_norm(x, 32, P[f"{p}.norm.weight"], P[f"{p}.norm.bias"], eps=1e-6)
    h = F.conv2d(h, P[f"{p}.proj_in.weight"], P[f"{p}.proj_in.bias"])
    h = h.reshape(b, s.inner, hh * ww).permute(0, 2, 1)
    for d in range(s.depth):
        h = transformer_block(P, f"{p}.transformer_blocks.{d}", h, context, s.heads)
    h = h.permute(0, 2, 1).reshape(b, s.inner, hh, ww)
    return x + F.conv2d(h, P[f"{p}.proj_out.weight"], P[f"{p}.proj_out.bias"])


def _run_seq(P, seq, h, emb, context):
    for blk in seq:
        if isinstance(blk, SDStemSpec):
            h = F.conv2d(h, P[f"{blk.prefix}.weight"], P[f"{blk.prefix}.bias"], padding=1)
        elif isinstance(blk, SDResBlockSpec):
            h = resblock(P, blk, h, emb)
        elif isinstance(blk, SDTransformerSpec):
            h = spatial_transformer(P, blk, h, context)
        elif isinstance(blk, SDDownSpec):
            h = F.conv2d(h, P[f"{blk.prefix}.op.weight"], P[f"{blk.prefix}.op.bias"], stride=2, padding=1)
        elif isinstance(blk, SDUpSpec):
            h = F.interpolate(h, scale_factor=2, mode="nearest")
            h = F.conv2d(h, P[f"{blk.prefix}.conv.weight"], P[f"{blk.prefix}.conv.bias"], padding=1)
        else:
            raise TypeError(blk)
    return h


def sd_unet_forward(P: Params, plan: SDUNetPlan, x, t, context):
    """x [N, C, H, W] float32 latents, t [N] timesteps, context [N, S, context_dim] -> [N, out, H, W]."""
    e = sinusoid_embedding(t, plan.model_channels)
    e = F.linear(e, P["time_embed.0.weight"], P["time_embed.0.bias"])
    emb = F.linear(_silu(e), P["time_embed.2.weight"], P["time_embed.2.bias"])
    h = x.float()
    context = context.float()
    hs = []
    for seq in plan.input_blocks:
        h = _run_seq(P, seq, h, emb, context)
        hs.append(h)
    h = _run_seq(P, plan.middle_block, h, emb, context)
    for seq in plan.output_blocks:
        h = torch.cat([h, hs.pop()], dim=1)
        h = _run_seq(P, seq, h, emb, context)
    h = _silu(group_norm(h, P["out.0.weight"], P["out.0.bias"]))
    return F.conv2d(h, P["out.2.weight"], P["out.2.bias"], padding=1)
