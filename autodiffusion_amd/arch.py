"""Architecture plans for the ADM UNet / dynamic UNet / classifier half-UNet.

A *plan* is pure metadata: the ordered list of blocks, their channel counts,
layer ids and the reference state-dict key prefixes.  It performs no arithmetic.
Both the HIP engine (``autodiffusion_amd.unet``) and the CPU oracle
(``oracle/nets.py``) interpret the same plan, so parameter names and shapes are
identical to the reference checkpoints:

* block topology follows ``UNetModel.__init__``
  (reference ``guided_diffusion/unet.py:396-616``),
* ``layer_id`` numbering follows ``Dynamic_UNetModel.__init__``
  (reference ``guided_diffusion/dynamic_unet.py:507-655``),
* the classifier follows ``EncoderUNetModel.__init__``
  (reference ``guided_diffusion/unet.py:685-857``).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple, Union

GN_GROUPS = 32  # reference nn.py:93-100 (GroupNorm32(32, channels))


@dataclass(frozen=True)
class ResBlockSpec:
    prefix: str          # e.g. "input_blocks.3.0"
    cin: int
    cout: int
    emb_dim: int
    up: bool = False
    down: bool = False
    scale_shift: bool = True
    layer_id: int = -1

    @property
    def has_skip_conv(self) -> bool:
        return self.cin != self.cout

    def param_shapes(self) -> Dict[str, Tuple[int, ...]]:
        p = self.prefix
        emb_out = 2 * self.cout if self.scale_shift else self.cout
        s = {
            f"{p}.in_layers.0.weight": (self.cin,),
            f"{p}.in_layers.0.bias": (self.cin,),
            f"{p}.in_layers.2.weight": (self.cout, self.cin, 3, 3),
            f"{p}.in_layers.2.bias": (self.cout,),
            f"{p}.emb_layers.1.weight": (emb_out, self.emb_dim),
            f"{p}.emb_layers.1.bias": (emb_out,),
            f"{p}.out_layers.0.weight": (self.cout,),
            f"{p}.out_layers.0.bias": (self.cout,),
            f"{p}.out_layers.3.weight": (self.cout, self.cout, 3, 3),
            f"{p}.out_layers.3.bias": (self.cout,),
        }
        if self.has_skip_conv:
            s[f"{p}.skip_connection.weight"] = (self.cout, self.cin, 1, 1)
            s[f"{p}.skip_connection.bias"] = (self.cout,)
        return s


@dataclass(frozen=True)
class AttnSpec:
    prefix: str
    channels: int
    num_heads: int
    new_order: bool      # QKVAttention (True) vs QKVAttentionLegacy (False)
    layer_id: int = -1

    @property
    def head_dim(self) -> int:
        return self.channels // self.num_heads

    def param_shapes(self) -> Dict[str, Tuple[int, ...]]:
        p, c = self.prefix, self.channels
        return {
            f"{p}.norm.weight": (c,),
            f"{p}.norm.bias": (c,),
            f"{p}.qkv.weight": (3 * c, c, 1),
            f"{p}.qkv.bias": (3 * c,),
            f"{p}.proj_out.weight": (c, c, 1),
            f"{p}.proj_out.bias": (c,),
        }


@dataclass(frozen=True)
class StemSpec:
    prefix: str          # "input_blocks.0.0"
    cin: int
    cout: int

    def param_shapes(self):
        return {
            f"{self.prefix}.weight": (self.cout, self.cin, 3, 3),
            f"{self.prefix}.bias": (self.cout,),
        }


@dataclass(frozen=True)
class HeadSpec:
    """GN -> SiLU -> conv3x3 (UNet ``out``)."""
    prefix: str          # "out"
    cin: int
    cout: int

    def param_shapes(self):
        p = self.prefix
        return {
            f"{p}.0.weight": (self.cin,),
            f"{p}.0.bias": (self.cin,),
            f"{p}.2.weight": (self.cout, self.cin, 3, 3),
            f"{p}.2.bias": (self.cout,),
        }


@dataclass(frozen=True)
class AttnPoolSpec:
    """GN -> SiLU -> AttentionPool2d (classifier ``out``, pool="attention")."""
    prefix: str          # "out"
    channels: int
    spatial: int         # feature-map side
    head_dim: int
    out_dim: int

    @property
    def num_heads(self) -> int:
        return self.channels // self.head_dim

    def param_shapes(self):
        p, c = self.prefix, self.channels
        return {
            f"{p}.0.weight": (c,),
            f"{p}.0.bias": (c,),
            f"{p}.2.positional_embedding": (c, self.spatial ** 2 + 1),
            f"{p}.2.qkv_proj.weight": (3 * c, c, 1),
            f"{p}.2.qkv_proj.bias": (3 * c,),
            f"{p}.2.c_proj.weight": (self.out_dim, c, 1),
            f"{p}.2.c_proj.bias": (self.out_dim,),
        }


@dataclass(frozen=True)
class PoolHeadSpec:
    """The classifier's other heads (reference unet.py:826-856): pool = "adaptive" (GN -> SiLU -> AdaptiveAvgPool2d(1) -> conv1x1 ->
    Flatten), "spatial" (Linear(F, 2048) -> ReLU -> Linear over the concatenated per-block channel means, F = ``_feature_size``:
    the stem's, every input block's and the middle block's channels) or "spatial_v2" (Linear -> GroupNorm32(32, 2048) -> SiLU -> Linear)."""
    prefix: str          # "out"
    kind: str
    channels: int        # of the final feature map
    feature_size: int    # F (spatial pools)
    out_dim: int
    hidden: int = 2048

    def param_shapes(self):
        p, c, o, f, hd = self.prefix, self.channels, self.out_dim, self.feature_size, self.hidden
        if self.kind == "adaptive":
            return {f"{p}.0.weight": (c,), f"{p}.0.bias": (c,), f"{p}.3.weight": (o, c, 1, 1), f"{p}.3.bias": (o,)}
        if self.kind == "spatial":
            return {f"{p}.0.weight": (hd, f), f"{p}.0.bias": (hd,), f"{p}.2.weight": (o, hd), f"{p}.2.bias": (o,)}
        return {f"{p}.0.weight": (hd, f), f"{p}.0.bias": (hd,), f"{p}.1.weight": (hd,), f"{p}.1.bias": (hd,),
                f"{p}.3.weight": (o, hd), f"{p}.3.bias": (o,)}


@dataclass(frozen=True)
class ResampleSpec:
    """``Downsample`` / ``Upsample`` of a UNet built with ``resblock_updown=False`` (reference unet.py:78-141): a 3x3 stride-2 conv
    (key ``<prefix>.op``) / nearest-neighbour 2x + 3x3 conv (key ``<prefix>.conv``) when ``use_conv`` (the constructor's
    ``conv_resample``, True in every factory), else AvgPool2d(2) / plain nearest 2x.  Never a skippable layer (no layer_id:
    dynamic_unet.py:556-561, 642-645 count only the ResBlock forms)."""
    prefix: str
    channels: int
    down: bool
    use_conv: bool = True
    layer_id: int = -1

    def param_shapes(self):
        if not self.use_conv:
            return {}
        k = f"{self.prefix}.op" if self.down else f"{self.prefix}.conv"
        return {f"{k}.weight": (self.channels, self.channels, 3, 3), f"{k}.bias": (self.channels,)}


Block = Union[StemSpec, ResBlockSpec, AttnSpec, ResampleSpec]


@dataclass
class UNetPlan:
    image_size: int
    in_channels: int
    model_channels: int
    out_channels: int
    emb_dim: int
    num_classes: Optional[int]
    input_blocks: List[List[Block]] = field(default_factory=list)
    middle_block: List[Block] = field(default_factory=list)
    output_blocks: List[List[Block]] = field(default_factory=list)
    head: Union[HeadSpec, AttnPoolSpec, None] = None
    layer_num: int = 0
    dynamic: bool = False
    encoder_only: bool = False

    def all_blocks(self):
        for seq in self.input_blocks:
            yield from seq
        yield from self.middle_block
        for seq in self.output_blocks:
            yield from seq

    def param_shapes(self) -> Dict[str, Tuple[int, ...]]:
        """Reference state-dict keys -> shapes, in the reference's order."""
        m, e = self.model_channels, self.emb_dim
        s: Dict[str, Tuple[int, ...]] = {
            "time_embed.0.weight": (e, m),
            "time_embed.0.bias": (e,),
            "time_embed.2.weight": (e, e),
            "time_embed.2.bias": (e,),
        }
        if self.num_classes is not None:
            s["label_emb.weight"] = (self.num_classes, e)
        for b in self.all_blocks():
            s.update(b.param_shapes())
        if self.head is not None:
            s.update(self.head.param_shapes())
        return s


def _num_heads(ch: int, num_heads: int, num_head_channels: int) -> int:
    # reference unet.py:276-283
    if num_head_channels == -1:
        return num_heads
    if ch % num_head_channels != 0:
        raise AssertionError(
            f"q,k,v channels {ch} is not divisible by num_head_channels {num_head_channels}"
        )
    return ch // num_head_channels


def build_unet_plan(
    image_size: int,
    in_channels: int,
    model_channels: int,
    out_channels: int,
    num_res_blocks: int,
    attention_resolutions: Sequence[int],
    channel_mult: Sequence[float] = (1, 2, 4, 8),
    num_classes: Optional[int] = None,
    num_heads: int = 1,
    num_head_channels: int = -1,
    num_heads_upsample: int = -1,
    use_scale_shift_norm: bool = False,
    resblock_updown: bool = False,
    use_new_attention_order: bool = False,
    dynamic: bool = False,
    encoder_only: bool = False,
    pool: str = "attention",
    conv_resample: bool = True,
) -> UNetPlan:
    """Mirror of the reference constructors' bookkeeping (no tensors)."""
    if not resblock_updown and not conv_resample and len(channel_mult) > 1 and encoder_only:
        raise NotImplementedError("classifier with AvgPool2d down-sampling (conv_resample=False): create_classifier never builds it")
    if num_heads_upsample == -1:
        num_heads_upsample = num_heads
    emb_dim = model_channels * 4
    plan = UNetPlan(
        image_size=image_size, in_channels=in_channels, model_channels=model_channels,
        out_channels=out_channels, emb_dim=emb_dim, num_classes=num_classes,
        dynamic=dynamic, encoder_only=encoder_only,
    )
    ch = input_ch = int(channel_mult[0] * model_channels)
    plan.input_blocks.append([StemSpec("input_blocks.0.0", in_channels, ch)])
    chans = [ch]
    ds = 1
    lid = 0
    idx = 1
    for level, mult in enumerate(channel_mult):
        for _ in range(num_res_blocks):
            cout = int(mult * model_channels)
            seq: List[Block] = [ResBlockSpec(f"input_blocks.{idx}.0", ch, cout, emb_dim,
                                             scale_shift=use_scale_shift_norm, layer_id=lid)]
            lid += 1
            ch = cout
            if ds in attention_resolutions:
                seq.append(AttnSpec(f"input_blocks.{idx}.1", ch,
                                    _num_heads(ch, num_heads, num_head_channels),
                                    use_new_attention_order, layer_id=lid))
                lid += 1
            plan.input_blocks.append(seq)
            chans.append(ch)
            idx += 1
        if level != len(channel_mult) - 1:
            if resblock_updown:
                plan.input_blocks.append([ResBlockSpec(f"input_blocks.{idx}.0", ch, ch, emb_dim,
                                                       down=True, scale_shift=use_scale_shift_norm,
                                                       layer_id=lid)])
                lid += 1
            else:
                plan.input_blocks.append([ResampleSpec(f"input_blocks.{idx}.0", ch, down=True, use_conv=conv_resample)])
            chans.append(ch)
            ds *= 2
            idx += 1
    plan.middle_block = [
        ResBlockSpec("middle_block.0", ch, ch, emb_dim, scale_shift=use_scale_shift_norm, layer_id=lid),
        AttnSpec("middle_block.1", ch, _num_heads(ch, num_heads, num_head_channels),
                 use_new_attention_order, layer_id=lid + 1),
        ResBlockSpec("middle_block.2", ch, ch, emb_dim, scale_shift=use_scale_shift_norm, layer_id=lid + 2),
    ]
    lid += 3
    if encoder_only:
        if pool == "attention":
            if num_head_channels == -1:
                raise AssertionError("attention pool needs num_head_channels")
            plan.head = AttnPoolSpec("out", ch, image_size // ds, num_head_channels, out_channels)
        elif pool in ("adaptive", "spatial", "spatial_v2"):
            plan.head = PoolHeadSpec("out", pool, ch, sum(chans) + ch, out_channels)   # chans: the stem + every input block; + the middle block
        else:
            raise NotImplementedError(f"Unexpected {pool} pooling")
        plan.layer_num = lid
        return plan
    oidx = 0
    for level, mult in list(enumerate(channel_mult))[::-1]:
        for i in range(num_res_blocks + 1):
            ich = chans.pop()
            cout = int(model_channels * mult)
            seq = [ResBlockSpec(f"output_blocks.{oidx}.0", ch + ich, cout, emb_dim,
                                scale_shift=use_scale_shift_norm, layer_id=lid)]
            lid += 1
            ch = cout
            sub = 1
            if ds in attention_resolutions:
                seq.append(AttnSpec(f"output_blocks.{oidx}.{sub}", ch,
                                    _num_heads(ch, num_heads_upsample, num_head_channels),
                                    use_new_attention_order, layer_id=lid))
                lid += 1
                sub += 1
            if level and i == num_res_blocks:
                if resblock_updown:
                    seq.append(ResBlockSpec(f"output_blocks.{oidx}.{sub}", ch, ch, emb_dim, up=True,
                                            scale_shift=use_scale_shift_norm, layer_id=lid))
                    lid += 1
                else:
                    seq.append(ResampleSpec(f"output_blocks.{oidx}.{sub}", ch, down=False, use_conv=conv_resample))
                ds //= 2
            plan.output_blocks.append(seq)
            oidx += 1
    plan.head = HeadSpec("out", input_ch, out_channels)
    plan.layer_num = lid
    return plan
