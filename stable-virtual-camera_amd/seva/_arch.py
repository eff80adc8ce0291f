"""Static layout of the Seva UNet derived from `SevaParams` (reference: seva/model.py:39-174).

Pure host logic: turns the hyper-parameters into a flat list of layer specs (with state_dict
prefixes, channel counts, resolution level and the attention regime of every transformer).
`seva.model` uses it to create parameter holders with the reference's 1146 key names and
`seva._engine` uses it to schedule HIP kernels.
"""

from __future__ import annotations

from dataclasses import dataclass, field


@dataclass
class ConvSpec:  # plain 3x3 conv (input stem)
    prefix: str
    cin: int
    cout: int
    kind: str = "conv"


@dataclass
class ResSpec:  # seva/modules/layers.py:86-139
    prefix: str
    cin: int
    cout: int
    kind: str = "res"


@dataclass
class MvtSpec:  # seva/modules/transformer.py:168-247
    prefix: str
    channels: int
    heads: int
    name: str
    joint: bool
    depth: int
    kind: str = "mvt"


@dataclass
class DownSpec:  # layers.py:49-58
    prefix: str
    channels: int
    kind: str = "down"


@dataclass
class UpSpec:  # layers.py:35-46
    prefix: str
    channels: int
    kind: str = "up"


@dataclass
class Layout:
    input_blocks: list = field(default_factory=list)  # list[list[spec]]
    middle: list = field(default_factory=list)
    output_blocks: list = field(default_factory=list)
    skip_channels: list = field(default_factory=list)  # channels pushed by each input block
    final_channels: int = 0
    time_embed_dim: int = 0

    def all_specs(self):
        for blk in self.input_blocks:
            yield from blk
        yield from self.middle
        for blk in self.output_blocks:
            yield from blk


def build_layout(p) -> Layout:
    """Walk the UNet exactly as the reference constructor does (model.py:54-168)."""
    lay = Layout(time_embed_dim=p.model_channels * 4)
    mc = p.model_channels
    nlev = len(p.channel_mult)
    lay.input_blocks.append([ConvSpec("input_blocks.0.0", p.in_channels, mc)])
    skip = [mc]
    ch, ds, idx = mc, 1, 1

    def mvt(prefix, c, side, level):
        name = f"{side}_ds{ds}"
        return MvtSpec(prefix, c, c // p.num_head_channels, name, name in p.unflatten_names,
                       p.transformer_depth[level])

    for level, mult in enumerate(p.channel_mult):
        for _ in range(p.num_res_blocks):
            cout = mult * mc
            blk = [ResSpec(f"input_blocks.{idx}.0", ch, cout)]
            ch = cout
            if ds in p.attention_resolutions:
                blk.append(mvt(f"input_blocks.{idx}.1", ch, "input", level))
            lay.input_blocks.append(blk)
            skip.append(ch)
            idx += 1
        if level != nlev - 1:
            lay.input_blocks.append([DownSpec(f"input_blocks.{idx}.0", ch)])
            skip.append(ch)
            ds *= 2
            idx += 1
    lay.skip_channels = list(skip)

    lay.middle = [
        ResSpec("middle_block.0", ch, ch),
        mvt("middle_block.1", ch, "middle", nlev - 1),
        ResSpec("middle_block.2", ch, ch),
    ]

    oidx = 0
    for level in reversed(range(nlev)):
        mult = p.channel_mult[level]
        for i in range(p.num_res_blocks + 1):
            ich = skip.pop()
            cout = mc * mult
            blk = [ResSpec(f"output_blocks.{oidx}.0", ch + ich, cout)]
            ch = cout
            j = 1
            if ds in p.attention_resolutions:
                blk.append(mvt(f"output_blocks.{oidx}.{j}", ch, "output", level))
                j += 1
            if level and i == p.num_res_blocks:
                blk.append(UpSpec(f"output_blocks.{oidx}.{j}", ch))
                ds //= 2
            lay.output_blocks.append(blk)
            oidx += 1
    lay.final_channels = ch
    return lay
