"""State-dict layouts (diffusers key names) + deterministic random-init generator + on-disk layout I/O.

No trained weights exist offline, so every config runs on seeded random-init weights whose
KEY NAMES and SHAPES are those of the checkpoints the reference loads:

* UNet / ControlNet / AutoencoderKL keys: diffusers==0.26.3 module tree (pinned requirements-jetson.txt:25),
  as consumed by model/controllora.py:600-632 (`_skip_layers`, tie_weights).
* LoRA keys `<linear>.lora_layer.{down,up}.weight`: model/controllora.py:577-606.
* Fusion keys `multi_controlnet_down_blocks.{i}.*` / `multi_controlnet_mid_block.*`:
  model/edgestyle_multicontrolnet.py:23-53,173-193.
* Directory layout `<dir>/diffusion_pytorch_model.safetensors` + `<dir>/controlnet_{idx}/`:
  model/edgestyle_multicontrolnet.py:240-282,380-398.

Each tensor is drawn from its own generator seeded by crc32(key) ^ seed, so a key's values do not depend on
which other keys are generated (the oracle, the HIP path and the golden-fixture writer all see the same bits).
"""
import json
import os
import zlib
from collections import OrderedDict
from typing import Dict, Tuple

import torch

from .config import UNetConfig, VAEConfig, NUM_CONTROLNETS

Shapes = "OrderedDict[str, Tuple[int, ...]]"

SKIP_LAYERS = ("conv_in", "time_proj", "time_embedding", "class_embedding", "down_blocks", "mid_block")  # CL:443-450


# ----------------------------------------------------------------------------------------------------------------
# shape enumerators
# ----------------------------------------------------------------------------------------------------------------
def _conv(sh, name, cin, cout, k, bias=True):
    sh[f"{name}.weight"] = (cout, cin, k, k)
    if bias:
        sh[f"{name}.bias"] = (cout,)


def _lin(sh, name, cin, cout, bias=True):
    sh[f"{name}.weight"] = (cout, cin)
    if bias:
        sh[f"{name}.bias"] = (cout,)


def _norm(sh, name, c):
    sh[f"{name}.weight"] = (c,)
    sh[f"{name}.bias"] = (c,)


def _resnet(sh, name, cin, cout, temb_dim):
    _norm(sh, f"{name}.norm1", cin)
    _conv(sh, f"{name}.conv1", cin, cout, 3)
    if temb_dim:
        _lin(sh, f"{name}.time_emb_proj", temb_dim, cout)
    _norm(sh, f"{name}.norm2", cout)
    _conv(sh, f"{name}.conv2", cout, cout, 3)
    if cin != cout:
        _conv(sh, f"{name}.conv_shortcut", cin, cout, 1)


def _transformer(sh, name, c, cross_dim):
    _norm(sh, f"{name}.norm", c)
    _conv(sh, f"{name}.proj_in", c, c, 1)
    tb = f"{name}.transformer_blocks.0"
    _norm(sh, f"{tb}.norm1", c)
    for p in ("to_q", "to_k", "to_v"):
        _lin(sh, f"{tb}.attn1.{p}", c, c, bias=False)
    _lin(sh, f"{tb}.attn1.to_out.0", c, c)
    _norm(sh, f"{tb}.norm2", c)
    _lin(sh, f"{tb}.attn2.to_q", c, c, bias=False)
    _lin(sh, f"{tb}.attn2.to_k", cross_dim, c, bias=False)
    _lin(sh, f"{tb}.attn2.to_v", cross_dim, c, bias=False)
    _lin(sh, f"{tb}.attn2.to_out.0", c, c)
    _norm(sh, f"{tb}.norm3", c)
    _lin(sh, f"{tb}.ff.net.0.proj", c, 8 * c)
    _lin(sh, f"{tb}.ff.net.2", 4 * c, c)
    _conv(sh, f"{name}.proj_out", c, c, 1)


def _encoder_shapes(sh, cfg: UNetConfig):
    """conv_in, time_embedding, down_blocks, mid_block — the part a ControlNet shares with the UNet (CL:623-632)."""
    ch = cfg.block_out_channels
    td = cfg.time_embed_dim
    _conv(sh, "conv_in", cfg.in_channels, ch[0], 3)
    _lin(sh, "time_embedding.linear_1", ch[0], td)
    _lin(sh, "time_embedding.linear_2", td, td)
    cin = ch[0]
    for i, cout in enumerate(ch):
        for j in range(cfg.layers_per_block):
            _resnet(sh, f"down_blocks.{i}.resnets.{j}", cin, cout, td)
            if cfg.down_has_attn[i]:
                _transformer(sh, f"down_blocks.{i}.attentions.{j}", cout, cfg.cross_attention_dim)
            cin = cout
        if i != len(ch) - 1:
            _conv(sh, f"down_blocks.{i}.downsamplers.0.conv", cout, cout, 3)
    _resnet(sh, "mid_block.resnets.0", ch[-1], ch[-1], td)
    _transformer(sh, "mid_block.attentions.0", ch[-1], cfg.cross_attention_dim)
    _resnet(sh, "mid_block.resnets.1", ch[-1], ch[-1], td)


def unet_shapes(cfg: UNetConfig) -> Shapes:
    sh = OrderedDict()
    _encoder_shapes(sh, cfg)
    ch = cfg.block_out_channels
    td = cfg.time_embed_dim
    rev = list(reversed(ch))
    prev = rev[0]
    n = len(ch)
    for i, cout in enumerate(rev):
        cin_skip_block = rev[min(i + 1, n - 1)]
        for j in range(cfg.layers_per_block + 1):
            skip = cin_skip_block if j == cfg.layers_per_block else cout
            rin = prev if j == 0 else cout
            _resnet(sh, f"up_blocks.{i}.resnets.{j}", rin + skip, cout, td)
            if cfg.up_has_attn[i]:
                _transformer(sh, f"up_blocks.{i}.attentions.{j}", cout, cfg.cross_attention_dim)
        if i != n - 1:
            _conv(sh, f"up_blocks.{i}.upsamplers.0.conv", cout, cout, 3)
        prev = cout
    _norm(sh, "conv_norm_out", ch[0])
    _conv(sh, "conv_out", ch[0], cfg.out_channels, 3)
    return sh


def controlnet_own_shapes(cfg: UNetConfig, uses_vae: bool) -> Shapes:
    """Keys a ControlNet owns beyond the UNet encoder: zero-convs (+ cond embedding unless uses_vae)."""
    sh = OrderedDict()
    if not uses_vae:
        ce = cfg.conditioning_embedding_out_channels
        _conv(sh, "controlnet_cond_embedding.conv_in", cfg.conditioning_channels, ce[0], 3)
        for i in range(len(ce) - 1):
            _conv(sh, f"controlnet_cond_embedding.blocks.{2 * i}", ce[i], ce[i], 3)
            _conv(sh, f"controlnet_cond_embedding.blocks.{2 * i + 1}", ce[i], ce[i + 1], 3)
        _conv(sh, "controlnet_cond_embedding.conv_out", ce[-1], cfg.block_out_channels[0], 3)
    for i, (c, _) in enumerate(cfg.residual_table()[:-1]):
        _conv(sh, f"controlnet_down_blocks.{i}", c, c, 1)
    c = cfg.block_out_channels[-1]
    _conv(sh, "controlnet_mid_block", c, c, 1)
    return sh


def controlnet_shapes(cfg: UNetConfig) -> Shapes:
    """A full stand-alone ControlNetModel (the openpose net, TT:247-250)."""
    sh = OrderedDict()
    _encoder_shapes(sh, cfg)
    sh.update(controlnet_own_shapes(cfg, uses_vae=False))
    return sh


def lora_shapes(cfg: UNetConfig, rank: int) -> Shapes:
    """LoRA A/B on every Linear under SKIP_LAYERS (CL:577-593; conv LoRA off, TR:278-283)."""
    enc = OrderedDict()
    _encoder_shapes(enc, cfg)
    sh = OrderedDict()
    for k, s in enc.items():
        if k.endswith(".weight") and len(s) == 2:
            base = k[: -len(".weight")]
            sh[f"{base}.lora_layer.down.weight"] = (rank, s[1])
            sh[f"{base}.lora_layer.up.weight"] = (s[0], rank)
    return sh


def controllora_saved_shapes(cfg: UNetConfig, rank: int, uses_vae: bool = True) -> Shapes:
    """What `ControlLoRAModel.state_dict()` keeps on disk (CL:600-606)."""
    sh = controlnet_own_shapes(cfg, uses_vae)
    sh.update(lora_shapes(cfg, rank))
    return sh


def vae_shapes(cfg: VAEConfig) -> Shapes:
    sh = OrderedDict()
    ch = cfg.block_out_channels
    # encoder
    _conv(sh, "encoder.conv_in", cfg.in_channels, ch[0], 3)
    cin = ch[0]
    for i, cout in enumerate(ch):
        for j in range(cfg.layers_per_block):
            _resnet(sh, f"encoder.down_blocks.{i}.resnets.{j}", cin, cout, 0)
            cin = cout
        if i != len(ch) - 1:
            _conv(sh, f"encoder.down_blocks.{i}.downsamplers.0.conv", cout, cout, 3)
    for side, last in (("encoder", ch[-1]), ("decoder", ch[-1])):
        _resnet(sh, f"{side}.mid_block.resnets.0", last, last, 0)
        a = f"{side}.mid_block.attentions.0"
        _norm(sh, f"{a}.group_norm", last)
        for p in ("to_q", "to_k", "to_v", "to_out.0"):
            _lin(sh, f"{a}.{p}", last, last)
        _resnet(sh, f"{side}.mid_block.resnets.1", last, last, 0)
    _norm(sh, "encoder.conv_norm_out", ch[-1])
    _conv(sh, "encoder.conv_out", ch[-1], 2 * cfg.latent_channels, 3)
    _conv(sh, "quant_conv", 2 * cfg.latent_channels, 2 * cfg.latent_channels, 1)
    _conv(sh, "post_quant_conv", cfg.latent_channels, cfg.latent_channels, 1)
    # decoder
    rev = list(reversed(ch))
    _conv(sh, "decoder.conv_in", cfg.latent_channels, rev[0], 3)
    cin = rev[0]
    for i, cout in enumerate(rev):
        for j in range(cfg.layers_per_block + 1):
            _resnet(sh, f"decoder.up_blocks.{i}.resnets.{j}", cin, cout, 0)
            cin = cout
        if i != len(ch) - 1:
            _conv(sh, f"decoder.up_blocks.{i}.upsamplers.0.conv", cout, cout, 3)
    _norm(sh, "decoder.conv_norm_out", ch[0])
    _conv(sh, "decoder.conv_out", ch[0], cfg.out_channels, 3)
    return sh


def fusion_shapes(cfg: UNetConfig, num_nets: int = NUM_CONTROLNETS, sample_size: int = None) -> Shapes:
    """13 ControlNetBlocks (MC:23-53, MC:103-114).  LN affine planes have shape [C,H,W] (MC:34-36,44-46)."""
    sh = OrderedDict()
    table = cfg.residual_table(sample_size)
    half = num_nets // 2
    for i, (c, s) in enumerate(table):
        p = f"multi_controlnet_down_blocks.{i}" if i < len(table) - 1 else "multi_controlnet_mid_block"
        sh[f"{p}.first_conv.weight"] = (c * half, 2, 1, 1)
        sh[f"{p}.first_conv.bias"] = (c * half,)
        sh[f"{p}.first_normalization.weight"] = (c * half, s, s)
        sh[f"{p}.first_normalization.bias"] = (c * half, s, s)
        sh[f"{p}.second_conv.weight"] = (c, half, 1, 1)
        sh[f"{p}.second_conv.bias"] = (c,)
        sh[f"{p}.second_normalization.weight"] = (c, s, s)
        sh[f"{p}.second_normalization.bias"] = (c, s, s)
        sh[f"{p}.third_conv.weight"] = (c, 1, 1, 1)
        sh[f"{p}.third_conv.bias"] = (c,)
    return sh


# ----------------------------------------------------------------------------------------------------------------
# deterministic random init
# ----------------------------------------------------------------------------------------------------------------
def _init_one(key: str, shape, seed: int, device="cpu") -> torch.Tensor:
    dev = torch.device(device)
    g = torch.Generator(device=dev)          # cpu: bit-identical everywhere; cuda: fast path for bench weights
    g.manual_seed((zlib.crc32(key.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)
    x = torch.randn(shape, generator=g, dtype=torch.float32, device=dev)
    is_norm = any(t in key for t in (".norm", "_norm", "normalization"))
    if key.endswith(".bias"):
        return x * 0.02
    if is_norm:                      # GroupNorm / LayerNorm gains ~ N(1, 0.02)  
        return 1.0 + 0.02 * x
    if ".lora_layer.down." in key:
        return x / (shape[1] ** 0.5)
    if ".lora_layer.up." in key:
        return x * (0.5 / (shape[1] ** 0.5))
    fan_in = 1
    for d in shape[1:]:
        fan_in *= d
    gain = 1.0
    if key.startswith("controlnet_down_blocks") or key.startswith("controlnet_mid_block"):
        gain = 0.5                   # zero-convs are deliberately NON-zero (else the ControlNets contribute nothing)
    if "multi_controlnet" in key:    # grouped 1x1 convs of the fusion block: fan_in 2 / 3 / 1
        gain = 1.0
    return x * (gain / (fan_in ** 0.5))


def random_state_dict(shapes: Shapes, seed: int = 0, prefix: str = "", device="cpu") -> Dict[str, torch.Tensor]:
    """fp32 tensors; `prefix` only salts the seed (so unet / openpose / vae differ), keys stay unprefixed."""
    return OrderedDict((k, _init_one(prefix + k, s, seed, device)) for k, s in shapes.items())


# ----------------------------------------------------------------------------------------------------------------
# on-disk layout (MC:213-282 / MC:356-430, CL:600-614)
# ----------------------------------------------------------------------------------------------------------------
WEIGHTS_NAME = "diffusion_pytorch_model.safetensors"


def save_model_dir(path: str, state_dict: Dict[str, torch.Tensor], config: dict = None):
    from safetensors.torch import save_file
    os.makedirs(path, exist_ok=True)
    save_file({k: v.contiguous() for k, v in state_dict.items()}, os.path.join(path, WEIGHTS_NAME),
              metadata={"format": "pt"})
    if config is not None:
        with open(os.path.join(path, "config.json"), "w") as f:
            json.dump(config, f, indent=2)


def load_model_dir(path: str):
    from safetensors.torch import load_file
    if not os.path.isdir(path):
        raise ValueError(f"Provided path ({path}) should be a directory")   # MC:359-362
    sd = load_file(os.path.join(path, WEIGHTS_NAME), device="cpu")
    cfg = None
    cj = os.path.join(path, "config.json")
    if os.path.exists(cj):
        with open(cj) as f:
            cfg = json.load(f)
    return sd, cfg
