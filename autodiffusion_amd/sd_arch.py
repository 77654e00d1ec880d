"""Architecture plan of the Stable-Diffusion latent UNet.

Pure metadata, like ``arch.py``: block order, channel counts and the reference state-dict key prefixes of
``ldm.modules.diffusionmodules.openaimodel.UNetModel`` with ``use_spatial_transformer=True``,
``conv_resample=True``, ``resblock_updown=False``, ``use_scale_shift_norm=False`` (the only form the reference's
configs instantiate: "Stable Diffusion"/configs/stable-diffusion/v1-inference*.yaml).  Topology follows
openaimodel.py:500-693; ``SpatialTransformer`` follows ldm/modules/attention.py:196-260.
Interpreted by the HIP engine (``sd_unet.py``) and by the CPU oracle (``oracle/sd_nets.py``).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Sequence, Tuple, Union


@dataclass(frozen=True)
class SDStemSpec:
    prefix: str
    cin: int
    cout: int

    def param_shapes(self):
        return {f"{self.prefix}.weight": (self.cout, self.cin, 3, 3), f"{self.prefix}.bias": (self.cout,)}


@dataclass(frozen=True)
class SDResBlockSpec:
    prefix: str
    cin: int
    cout: int
    emb_dim: int

    @property
    def has_skip_conv(self) -> bool:
        return self.cin != self.cout

    def param_shapes(self):
        p = self.prefix
        s = {
            f"{p}.in_layers.0.weight": (self.cin,), f"{p}.in_layers.0.bias": (self.cin,),
            f"{p}.in_layers.2.weight": (self.cout, self.cin, 3, 3), f"{p}.in_layers.2.bias": (self.cout,),
            f"{p}.emb_layers.1.weight": (self.cout, self.emb_dim), f"{p}.emb_layers.1.bias": (self.cout,),
            f"{p}.out_layers.0.weight": (self.cout,), f"{p}.out_layers.0.bias": (self.cout,),
            f"{p}.out_layers.3.weight": (self.cout, self.cout, 3, 3), f"{p}.out_layers.3.bias": (self.cout,),
        }
        if self.has_skip_conv:
            s[f"{p}.skip_connection.weight"] = (self.cout, self.cin, 1, 1)
            s[f"{p}.skip_connection.bias"] = (self.cout,)
        return s


@dataclass(frozen=True)
class SDTransformerSpec:
    """SpatialTransformer: GroupNorm(eps 1e-6) - 1x1 proj_in - depth x BasicTransformerBlock - 1x1 proj_out (+x)."""
    prefix: str
    channels: int
    heads: int
    d_head: int
    context_dim: int
    depth: int = 1

    @property
    def inner(self) -> int:
        return self.heads * self.d_head

    def param_shapes(self):
        p, c, i = self.prefix, self.channels, self.inner
        s = {f"{p}.norm.weight": (c,), f"{p}.norm.bias": (c,),
             f"{p}.proj_in.weight": (i, c, 1, 1), f"{p}.proj_in.bias": (i,),
             f"{p}.proj_out.weight": (c, i, 1, 1), f"{p}.proj_out.bias": (c,)}
        for d in range(self.depth):
            b = f"{p}.transformer_blocks.{d}"
            for a, kv in (("attn1", i), ("attn2", self.context_dim)):
                s[f"{b}.{a}.to_q.weight"] = (i, i)
                s[f"{b}.{a}.to_k.weight"] = (i, kv)
                s[f"{b}.{a}.to_v.weight"] = (i, kv)
                s[f"{b}.{a}.to_out.0.weight"] = (i, i)
                s[f"{b}.{a}.to_out.0.bias"] = (i,)
            s[f"{b}.ff.net.0.proj.weight"] = (8 * i, i)
            s[f"{b}.ff.net.0.proj.bias"] = (8 * i,)
            s[f"{b}.ff.net.2.weight"] = (i, 4 * i)
            s[f"{b}.ff.net.2.bias"] = (i,)
            for k in ("norm1", "norm2", "norm3"):
                s[f"{b}.{k}.weight"] = (i,)
                s[f"{b}.{k}.bias"] = (i,)
        return s


@dataclass(frozen=True)
class SDDownSpec:
    """Downsample(conv_resample): 3x3 conv, stride 2, pad 1 (openaimodel.py:118-145); keys ``<prefix>.op.*``."""
    prefix: str
    channels: int

    def param_shapes(self):
        c = self.channels
        return {f"{self.prefix}.op.weight": (c, c, 3, 3), f"{self.prefix}.op.bias": (c,)}


@dataclass(frozen=True)
class SDUpSpec:
    """Upsample(conv_resample): nearest x2 then 3x3 conv (openaimodel.py:78-104); keys ``<prefix>.conv.*``."""
    prefix: str
    channels: int

    def param_shapes(self):
        c = self.channels
        return {f"{self.prefix}.conv.weight": (c, c, 3, 3), f"{self.prefix}.conv.bias": (c,)}


SDBlock = Union[SDStemSpec, SDResBlockSpec, SDTransformerSpec, SDDownSpec, SDUpSpec]


@dataclass
class SDUNetPlan:
    in_channels: int
    out_channels: int
    model_channels: int
    emb_dim: int
    context_dim: int
    input_blocks: List[List[SDBlock]] = field(default_factory=list)
    middle_block: List[SDBlock] = field(default_factory=list)
    output_blocks: List[List[SDBlock]] = field(default_factory=list)

    def all_blocks(self):
        for seq in self.input_blocks:
            yield from seq
        yield from self.middle_block
        for seq in self.output_blocks:
            yield from seq

    def param_shapes(self) -> Dict[str, Tuple[int, ...]]:
        mc, e = self.model_channels, self.emb_dim
        s = {"time_embed.0.weight": (e, mc), "time_embed.0.bias": (e,),
             "time_embed.2.weight": (e, e), "time_embed.2.bias": (e,)}
        for b in self.all_blocks():
            s.update(b.param_shapes())
        s.update({"out.0.weight": (mc,), "out.0.bias": (mc,),
                  "out.2.weight": (self.out_channels, mc, 3, 3), "out.2.bias": (self.out_channels,)})
        return s


def sd_unet_plan(in_channels: int, model_channels: int, out_channels: int, num_res_blocks: int,
                 attention_resolutions: Sequence[int], channel_mult: Sequence[int] = (1, 2, 4, 8),
                 num_heads: int = -1, num_head_channels: int = -1, transformer_depth: int = 1,
                 context_dim: int = None, legacy: bool = True) -> SDUNetPlan:
    assert context_dim is not None, "the latent UNet is built with use_spatial_transformer=True: context_dim is required"
    assert (num_heads == -1) != (num_head_channels == -1), "set exactly one of num_heads / num_head_channels"
    emb = model_channels * 4
    plan = SDUNetPlan(in_channels, out_channels, model_channels, emb, context_dim)

    def heads_for(ch):
        if num_head_channels == -1:
            h, d = num_heads, ch // num_heads
        else:
            h, d = ch // num_head_channels, num_head_channels
        if legacy:
            d = ch // h
        return h, d

    def transformer(prefix, ch):
        h, d = heads_for(ch)
        return SDTransformerSpec(prefix, ch, h, d, context_dim, transformer_depth)

    plan.input_blocks.append([SDStemSpec("input_blocks.0.0", in_channels, model_channels)])
    chans = [model_channels]
    ch, ds = model_channels, 1
    for level, mult in enumerate(channel_mult):
        for _ in range(num_res_blocks):
            i = len(plan.input_blocks)
            seq: List[SDBlock] = [SDResBlockSpec(f"input_blocks.{i}.0", ch, mult * model_channels, emb)]
            ch = mult * model_channels
            if ds in attention_resolutions:
                seq.append(transformer(f"input_blocks.{i}.1", ch))
            plan.input_blocks.append(seq)
            chans.append(ch)
        if level != len(channel_mult) - 1:
            i = len(plan.input_blocks)
            plan.input_blocks.append([SDDownSpec(f"input_blocks.{i}.0", ch)])
            chans.append(ch)
            ds *= 2
    plan.middle_block = [SDResBlockSpec("middle_block.0", ch, ch, emb), transformer("middle_block.1", ch),
                         SDResBlockSpec("middle_block.2", ch, ch, emb)]
    for level, mult in list(enumerate(channel_mult))[::-1]:
        for k in range(num_res_blocks + 1):
            ich = chans.pop()
            i = len(plan.output_blocks)
            seq = [SDResBlockSpec(f"output_blocks.{i}.0", ch + ich, model_channels * mult, emb)]
            ch = model_channels * mult
            if ds in attention_resolutions:
                seq.append(transformer(f"output_blocks.{i}.{len(seq)}", ch))
            if level and k == num_res_blocks:
                seq.append(SDUpSpec(f"output_blocks.{i}.{len(seq)}", ch))
                ds //= 2
            plan.output_blocks.append(seq)
    assert ch == model_channels
    return plan


SD_V1 = dict(in_channels=4, out_channels=4, model_channels=320, attention_resolutions=(4, 2, 1), num_res_blocks=2,
             channel_mult=(1, 2, 4, 4), num_heads=8, transformer_depth=1, context_dim=768, legacy=False)
