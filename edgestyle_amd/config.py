"""Architecture configs for the EdgeStyle hot path (SD1.5 UNet / ControlNet / VAE / fusion blocks).

The reference never states these numbers itself: they are the `config.json` of the SD1.5 checkpoints it
loads by name (README.md:131-135) and it hard-codes their consequences at
model/edgestyle_multicontrolnet.py:73-102 (residual channel/size table) and export_onnx.py:131-148
(`[2,4,64,64]`, `[2,77,768]`, `[2,320,64,64]`).  `sd15()` reproduces those; `tiny()` is a width/size-reduced
variant with the same topology used by the CPU tests so the oracle finishes in seconds.
"""
from dataclasses import dataclass, field, asdict
from typing import Tuple, List


@dataclass(frozen=True)
class UNetConfig:
    in_channels: int = 4
    out_channels: int = 4
    block_out_channels: Tuple[int, ...] = (320, 640, 1280, 1280)
    layers_per_block: int = 2
    # diffusers' SD1.5 config calls this `attention_head_dim: 8` but uses it as the HEAD COUNT
    num_heads: int = 8
    cross_attention_dim: int = 768
    norm_num_groups: int = 32
    norm_eps: float = 1e-5
    # which down blocks carry transformer layers (CrossAttnDownBlock2D x3, DownBlock2D)
    down_has_attn: Tuple[bool, ...] = (True, True, True, False)
    sample_size: int = 64          # latent H=W at the reference's 512x512 (TT:25)
    # ControlNet-only
    conditioning_channels: int = 3
    conditioning_embedding_out_channels: Tuple[int, ...] = (16, 32, 96, 256)

    @property
    def time_embed_dim(self) -> int:
        return self.block_out_channels[0] * 4

    @property
    def up_has_attn(self) -> Tuple[bool, ...]:
        return tuple(reversed(self.down_has_attn))

    def residual_table(self, sample_size: int = None) -> List[Tuple[int, int]]:
        """(channels, size) of the 12 down residuals + 1 mid residual.

        For sd15() at sample_size 64 this is exactly the table the reference hard-wires at
        model/edgestyle_multicontrolnet.py:73-102.
        """
        s = sample_size or self.sample_size
        out = [(self.block_out_channels[0], s)]
        for i, c in enumerate(self.block_out_channels):
            for _ in range(self.layers_per_block):
                out.append((c, s))
            if i != len(self.block_out_channels) - 1:
                s //= 2
                out.append((c, s))
        out.append((self.block_out_channels[-1], s))   # mid
        return out

    def to_dict(self):
        return asdict(self)


@dataclass(frozen=True)
class VAEConfig:
    in_channels: int = 3
    out_channels: int = 3
    latent_channels: int = 4
    block_out_channels: Tuple[int, ...] = (128, 256, 512, 512)
    layers_per_block: int = 2
    norm_num_groups: int = 32
    norm_eps: float = 1e-6
    scaling_factor: float = 0.18215

    @property
    def scale(self) -> int:
        return 2 ** (len(self.block_out_channels) - 1)

    def to_dict(self):
        return asdict(self)


def sd15_unet() -> UNetConfig:
    return UNetConfig()


def sd15_vae() -> VAEConfig:
    return VAEConfig()


def tiny_unet(sample_size: int = 16) -> UNetConfig:
    """Same topology, 1/5 of the width, 16x16 latents (128x128 images)."""
    return UNetConfig(block_out_channels=(64, 128, 256, 256), num_heads=4, cross_attention_dim=64,
                      sample_size=sample_size, conditioning_embedding_out_channels=(16, 32, 32, 64))


def tiny_vae() -> VAEConfig:
    return VAEConfig(block_out_channels=(32, 64, 64, 64))


NUM_CONTROLNETS = 6
# model/edgestyle_multicontrolnet.py callers: TT:50, TR:63, APP:40, EX:30
CONTROLNET_PATTERN = [0, None, 1, None, 1, None]
