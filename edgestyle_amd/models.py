"""The reference's model call surface, backed by the HIP executors.

Class names, constructor/`from_pretrained` kwargs, call signatures, return conventions and ValueErrors mirror what
test_text2image_pretrained_openpose.py (TT:224-275) and model/edgestyle_pipeline.py touch:

  UNet2DConditionModel, ControlNetModel, AutoencoderKL           <- diffusers classes the reference imports
  CachedControlNetModel, ControlLoRAModel, FusedControlLoRAModel <- model/controllora.py
  ControlNetBlock, EdgeStyleMultiControlNetModel                 <- model/edgestyle_multicontrolnet.py

Tensors cross this boundary as the reference passes them (NCHW, fp32/fp16, any device); inside, activations are
NHWC fp16/bf16 and every arithmetic op is a C-ABI kernel launch.  Residual tensors returned by the ControlNet
classes are NCHW-shaped zero-copy views of NHWC memory (torch channels_last), so feeding them back into
`UNet2DConditionModel` costs no conversion.

`StepRunner` is the fused per-step fast path the pipeline uses: nets that share weights run as one batched pass
(3 openpose passes = one batch-3N pass, 2 clothes-LoRA passes = one batch-2N pass), conditioning scales are
applied inside the fusion kernel, text K/V projections are computed once per image.
"""
import json
import os
from types import SimpleNamespace
from typing import Any, Dict, List, Optional, Sequence, Tuple, Union

import torch

from . import engine as E
from . import ops
from . import weights as W
from .config import UNetConfig, VAEConfig, sd15_unet, sd15_vae
from .lib import EdgeStyleHipError

_ENC_PREFIXES = ("conv_in", "time_embedding", "down_blocks", "mid_block")


class _Config(SimpleNamespace):
    def __contains__(self, k):
        return k in self.__dict__

    def get(self, k, default=None):
        return self.__dict__.get(k, default)


def _compute_dtype(torch_dtype) -> torch.dtype:
    """fp32 requests run under the reference's autocast policy (TT:327): fp16 storage, fp32 accumulate/statistics."""
    return torch.bfloat16 if torch_dtype == torch.bfloat16 else torch.float16


def unet_config_from_json(cfg: Optional[dict]) -> UNetConfig:
    """Accepts this repo's config.json or a diffusers UNet/ControlNet config.json."""
    if not cfg:
        return sd15_unet()
    if "down_has_attn" in cfg:
        kw = {k: (tuple(v) if isinstance(v, list) else v) for k, v in cfg.items() if k in UNetConfig.__dataclass_fields__}
        return UNetConfig(**kw)
    base = sd15_unet()
    heads = cfg.get("num_attention_heads") or cfg.get("attention_head_dim", base.num_heads)
    if isinstance(heads, (list, tuple)):
        heads = heads[0]
    types = cfg.get("down_block_types")
    has = tuple(t.startswith("CrossAttn") for t in types) if types else base.down_has_attn
    return UNetConfig(
        in_channels=cfg.get("in_channels", 4), out_channels=cfg.get("out_channels", 4),
        block_out_channels=tuple(cfg.get("block_out_channels", base.block_out_channels)),
        layers_per_block=cfg.get("layers_per_block", 2), num_heads=heads,
        cross_attention_dim=cfg.get("cross_attention_dim", 768), norm_num_groups=cfg.get("norm_num_groups", 32),
        norm_eps=cfg.get("norm_eps", 1e-5), down_has_attn=has, sample_size=cfg.get("sample_size", 64) or 64,
        conditioning_channels=cfg.get("conditioning_channels", 3),
        conditioning_embedding_out_channels=tuple(cfg.get("conditioning_embedding_out_channels", (16, 32, 96, 256))))


def vae_config_from_json(cfg: Optional[dict]) -> VAEConfig:
    if not cfg:
        return sd15_vae()
    kw = {k: (tuple(v) if isinstance(v, list) else v) for k, v in cfg.items() if k in VAEConfig.__dataclass_fields__}
    return VAEConfig(**kw)


def _as_nhwc(x: torch.Tensor, dtype, device, cpad: Optional[int] = None) -> torch.Tensor:
    """NCHW tensor (any dtype/device/contiguity) -> NHWC `dtype` on `device`.  channels_last views pass through."""
    if x.device.type != torch.device(device).type:
        x = x.to(device)
    N, C, H, Wd = x.shape
    v = x.permute(0, 2, 3, 1)
    if x.dtype == dtype and v.is_contiguous() and (cpad is None or cpad == C):
        return v
    return ops.nchw_to_nhwc(x.float().contiguous(), dtype, cpad)


def _as_nchw_view(x: torch.Tensor) -> torch.Tensor:
    return x.permute(0, 3, 1, 2)


def _guess_level_scales(n: int) -> List[float]:
    """guess_mode: torch.logspace(-1, 0, 13) over the 12 down + 1 mid residuals (CL:256-264)."""
    return torch.logspace(-1, 0, n).tolist()


def _timestep_tensor(timestep, n: int, device) -> torch.Tensor:
    """CL:134-148: scalar / 0-d / [n] timestep -> fp32 [n] device tensor."""
    if not torch.is_tensor(timestep):
        return torch.full((n,), float(timestep), dtype=torch.float32, device=device)
    t = timestep.reshape(-1).to(device=device, dtype=torch.float32)
    return t.expand(n).contiguous() if t.numel() == 1 else t


class _HipModel:
    """Common plumbing: fp32 CPU state dict + lazily built device engine."""
    _engine = None
    device = torch.device("cpu")

    def to(self, device=None, dtype=None, **kw):
        if isinstance(device, torch.dtype):
            device, dtype = None, device
        if dtype is not None:
            self.dtype = _compute_dtype(dtype)
            self._engine = None
        if device is not None:
            device = torch.device(device)
            if device != self.device:
                self.device = device
                self._engine = None
        return self

    def eval(self):
        return self

    def requires_grad_(self, flag=False):
        return self

    def parameters(self):
        return iter(self.state_dict().values())

    def _require_gpu(self):
        if self.device.type != "cuda":
            raise EdgeStyleHipError(f"{type(self).__name__} runs only on an MI355X: call .to('cuda') first "
                                    "(there is no CPU fallback; the CPU restatement lives in oracle/ for tests)")


# ----------------------------------------------------------------------------------------------------------------
class UNet2DConditionModel(_HipModel):
    def __init__(self, state_dict: Dict[str, torch.Tensor], cfg: UNetConfig = None, torch_dtype=None):
        self.cfg = cfg or sd15_unet()
        self._sd = state_dict
        self.dtype = _compute_dtype(torch_dtype)
        c = self.cfg
        self.config = _Config(in_channels=c.in_channels, out_channels=c.out_channels, sample_size=c.sample_size,
                              block_out_channels=c.block_out_channels, layers_per_block=c.layers_per_block,
                              cross_attention_dim=c.cross_attention_dim, attention_head_dim=c.num_heads,
                              norm_num_groups=c.norm_num_groups, norm_eps=c.norm_eps, time_cond_proj_dim=None,
                              flip_sin_to_cos=True, freq_shift=0, act_fn="silu")

    @classmethod
    def from_pretrained(cls, path, subfolder: Optional[str] = None, torch_dtype=None, **kw):
        d = os.path.join(path, subfolder) if subfolder else path
        sd, cfg = W.load_model_dir(d)
        return cls(sd, unet_config_from_json(cfg), torch_dtype)

    def state_dict(self):
        return self._sd

    def save_pretrained(self, path):
        W.save_model_dir(path, self._sd, self.cfg.to_dict())

    @property
    def engine(self) -> E.UNet:
        self._require_gpu()
        if self._engine is None:
            self._engine = E.UNet(self._sd, self.cfg, self.dtype, self.device)
        return self._engine

    def __call__(self, sample, timestep, encoder_hidden_states, down_block_additional_residuals=None,
                 mid_block_additional_residual=None, return_dict: bool = True, **unused):
        """PL:500-510. sample [N,4,h,w] -> noise_pred [N,4,h,w] (same dtype as the compute dtype)."""
        eng = self.engine
        N = sample.shape[0]
        x = _as_nhwc(sample, self.dtype, self.device, eng.in_pad)
        tproj = eng.time_proj(_timestep_tensor(timestep, N, self.device))
        ctx = eng.context(encoder_hidden_states.to(self.device, self.dtype).contiguous())
        down = mid = None
        if down_block_additional_residuals is not None:
            down = [_as_nhwc(r, self.dtype, self.device) for r in down_block_additional_residuals]
            mid = _as_nhwc(mid_block_additional_residual, self.dtype, self.device)
        out = _as_nchw_view(eng.forward(x, tproj, ctx, down, mid))
        return SimpleNamespace(sample=out) if return_dict else (out,)


# ----------------------------------------------------------------------------------------------------------------
class ControlNetModel(_HipModel):
    """diffusers ControlNetModel surface (the openpose net, TT:247-250) + the cached-cond shortcut of
    CachedControlNetModel.forward (CL:199-203) — the two differ only in that branch, which is data-driven."""
    uses_vae = False

    def __init__(self, state_dict: Dict[str, torch.Tensor], cfg: UNetConfig = None, torch_dtype=None, **config_kw):
        self.cfg = cfg or sd15_unet()
        self._own = dict(state_dict)
        self.dtype = _compute_dtype(torch_dtype)
        self.config = _Config(global_pool_conditions=False, controlnet_conditioning_channel_order="rgb",
                              uses_vae=self.uses_vae, in_channels=self.cfg.in_channels,
                              block_out_channels=self.cfg.block_out_channels,
                              cross_attention_dim=self.cfg.cross_attention_dim, **config_kw)

    @classmethod
    def from_pretrained(cls, path, torch_dtype=None, **kw):
        sd, cfg = W.load_model_dir(path)
        extra = {k: v for k, v in (cfg or {}).items() if k in ("lora_linear_rank", "lora_conv2d_rank")}
        m = cls(sd, unet_config_from_json(cfg), torch_dtype, **extra)
        if cfg and cfg.get("uses_vae"):
            m.config.uses_vae = True
            m.uses_vae = True
        return m

    # -- weights ------------------------------------------------------------------------------------------------
    def full_state_dict(self):
        return self._own

    def state_dict(self):
        return self._own

    def save_pretrained(self, path, **kw):
        cfg = self.cfg.to_dict()
        cfg.update(uses_vae=bool(self.config.uses_vae), lora_linear_rank=self.config.get("lora_linear_rank", 0))
        W.save_model_dir(path, self.state_dict(), cfg)

    @property
    def engine(self) -> E.ControlNet:
        self._require_gpu()
        if self._engine is None:
            self._engine = E.ControlNet(self.full_state_dict(), self.cfg, self.dtype, self.device,
                                        uses_vae=bool(self.config.uses_vae))
        return self._engine

    # -- conditioning -------------------------------------------------------------------------------------------
    def preprocess_image(self, image: torch.Tensor, noise: Optional[torch.Tensor] = None,
                         generator: Optional[torch.Generator] = None) -> torch.Tensor:
        """CL:289-290: run the conditioning embedding once per image. image [N,3,H,W] -> [N,C0,H/8,W/8] (NCHW view)."""
        eng = self.engine
        img = _as_nhwc(image, self.dtype, self.device, 8)
        return _as_nchw_view(eng.embed_cond(img))

    def __call__(self, sample, timestep, encoder_hidden_states, controlnet_cond, conditioning_scale: float = 1.0,
                 guess_mode: bool = False, return_dict: bool = True, **unused):
        order = self.config.controlnet_conditioning_channel_order
        if order == "bgr":
            controlnet_cond = torch.flip(controlnet_cond, dims=[1])
        elif order != "rgb":
            raise ValueError(f"unknown `controlnet_conditioning_channel_order`: {order}")   # CL:124-126
        eng = self.engine
        N = sample.shape[0]
        x = _as_nhwc(sample, self.dtype, self.device, eng.in_pad)
        if tuple(controlnet_cond.shape[2:]) != tuple(sample.shape[2:]):     # CL:199-203
            controlnet_cond = self.preprocess_image(controlnet_cond)
        cond = _as_nhwc(controlnet_cond, self.dtype, self.device)
        tproj = eng.time_proj(_timestep_tensor(timestep, N, self.device))
        ctx = eng.context(encoder_hidden_states.to(self.device, self.dtype).contiguous())
        ls = _guess_level_scales(len(eng.zero) + 1) if guess_mode else None           # CL:256-264
        res = eng.forward(x, tproj, ctx, [cond], out_scale=float(conditioning_scale), level_scales=ls)
        down = [_as_nchw_view(r) for r in res[:-1]]
        mid = _as_nchw_view(res[-1])
        if not return_dict:
            return down, mid
        return SimpleNamespace(down_block_res_samples=down, mid_block_res_sample=mid)


class CachedControlNetModel(ControlNetModel):
    pass


class _LatentDist:
    def __init__(self, vae, moments):
        self.vae, self.moments = vae, moments

    def sample(self, generator: Optional[torch.Generator] = None, noise: Optional[torch.Tensor] = None):
        N, h, w, _ = self.moments.shape
        Lc = self.vae.cfg.latent_channels
        if noise is None:
            noise = torch.randn((N, Lc, h, w), generator=generator, dtype=torch.float32)
        z = ops.vae_sample(self.moments, noise.to(self.moments.device), Lc, Lc, 1.0)
        return _as_nchw_view(z)

    def sample_scaled_nhwc(self, noise: torch.Tensor, lpad: int, scaling: float):
        return ops.vae_sample(self.moments, noise.to(self.moments.device), self.vae.cfg.latent_channels, lpad, scaling)


class AutoencoderKL(_HipModel):
    def __init__(self, state_dict, cfg: VAEConfig = None, torch_dtype=None):
        self.cfg = cfg or sd15_vae()
        self._sd = state_dict
        self.dtype = _compute_dtype(torch_dtype)
        self.config = _Config(scaling_factor=self.cfg.scaling_factor, latent_channels=self.cfg.latent_channels,
                              block_out_channels=self.cfg.block_out_channels)

    @classmethod
    def from_pretrained(cls, path, subfolder: Optional[str] = None, torch_dtype=None, **kw):
        d = os.path.join(path, subfolder) if subfolder else path
        sd, cfg = W.load_model_dir(d)
        return cls(sd, vae_config_from_json(cfg), torch_dtype)

    def state_dict(self):
        return self._sd

    def save_pretrained(self, path):
        W.save_model_dir(path, self._sd, self.cfg.to_dict())

    @property
    def engine(self) -> E.VAE:
        self._require_gpu()
        if self._engine is None:
            self._engine = E.VAE(self._sd, self.cfg, self.dtype, self.device)
        return self._engine

    def encode(self, x: torch.Tensor):
        mom = self.engine.encode_moments(_as_nhwc(x, self.dtype, self.device, 8))
        return SimpleNamespace(latent_dist=_LatentDist(self, mom))

    def decode_nhwc(self, z_nhwc: torch.Tensor, unscaled_latents: bool = False) -> torch.Tensor:
        return self.engine.decode(z_nhwc, unscaled_latents)

    def decode(self, z: torch.Tensor, return_dict: bool = True, generator=None):
        """PL:552-557: z [B,4,h,w] (already divided by scaling_factor) -> image [B,3,8h,8w] in [-1,1]."""
        img = _as_nchw_view(self.engine.decode(_as_nhwc(z, self.dtype, self.device, self.engine.lat_pad)))
        return SimpleNamespace(sample=img) if return_dict else (img,)


# ----------------------------------------------------------------------------------------------------------------
class ControlLoRAModel(CachedControlNetModel):
    """ControlNet whose encoder parameters ARE the UNet's (tie_weights, CL:623-632) plus rank-r LoRA deltas on every
    Linear under conv_in/time_embedding/down_blocks/mid_block (CL:443-450, 577-593).  Its own state dict holds only
    the LoRA matrices and the 13 zero-convs (CL:600-606).  With uses_vae the conditioning embedding is
    vae.encode(x).sample() * scaling_factor -> conv_in (CL:28-42; conv_vae_out IS conv_in, CL:595-598).

    The HIP engine folds W + B.A into PRIVATE fp16 copies at build time; the tied UNet tensors are never mutated
    (the reference's fuse() does mutate them, CL:728-777 — a hazard we deliberately do not reproduce)."""
    _skip_layers = list(W.SKIP_LAYERS)

    def __init__(self, state_dict=None, cfg: UNetConfig = None, torch_dtype=None, lora_linear_rank: int = 4,
                 lora_conv2d_rank: int = 0, uses_vae: bool = False, **kw):
        if lora_conv2d_rank:
            raise NotImplementedError("conv LoRA is never enabled by the reference (TR:278-283)")
        self.uses_vae = uses_vae
        super().__init__(state_dict or {}, cfg, torch_dtype, lora_linear_rank=lora_linear_rank,
                         lora_conv2d_rank=lora_conv2d_rank)
        self.config.uses_vae = uses_vae
        self._unet_sd = None
        self._vae = None

    @classmethod
    def from_pretrained(cls, path, torch_dtype=None, **kw):
        sd, cfg = W.load_model_dir(path)
        cfg = cfg or {}
        return cls(sd, unet_config_from_json(cfg), torch_dtype, lora_linear_rank=cfg.get("lora_linear_rank", 4),
                   uses_vae=bool(cfg.get("uses_vae", False)))

    @classmethod
    def from_unet(cls, unet: UNet2DConditionModel, conditioning_channels: int = 3, lora_linear_rank: int = 4,
                  lora_conv2d_rank: int = 0, autoencoder: Optional[AutoencoderKL] = None, seed: int = 0, **kw):
        """CL:642-725: fresh LoRA (A ~ N(0,1/r), B = 0) and zero-initialised zero-convs, tied to `unet`."""
        shapes = W.controllora_saved_shapes(unet.cfg, lora_linear_rank, uses_vae=autoencoder is not None)
        sd = {}
        g = torch.Generator().manual_seed(seed)
        for k, s in shapes.items():
            if ".lora_layer.down." in k:
                sd[k] = torch.randn(s, generator=g) / lora_linear_rank
            elif k.startswith("controlnet_cond_embedding") and not k.endswith("conv_out.weight") and k.endswith(".weight"):
                sd[k] = torch.randn(s, generator=g) * (1.0 / (s[1] * 9) ** 0.5)
            else:
                sd[k] = torch.zeros(s)
        m = cls(sd, unet.cfg, unet.dtype, lora_linear_rank=lora_linear_rank, uses_vae=autoencoder is not None)
        if autoencoder is not None:
            m.set_autoencoder(autoencoder)
        m.tie_weights(unet)
        m.to(unet.device)
        return m

    def tie_weights(self, unet: UNet2DConditionModel):
        self._unet_sd = unet.state_dict()       # aliased, not copied
        self._engine = None

    def set_autoencoder(self, autoencoder: Optional[AutoencoderKL]):
        """CL:634-640 — without the reference's in-place zero_module(conv_in) side effect on an un-flagged net."""
        self._vae = autoencoder
        if autoencoder is not None:
            self.uses_vae = True
            self.config.uses_vae = True

    def state_dict(self):
        """CL:600-606: non-encoder keys + LoRA keys only."""
        return {k: v for k, v in self._own.items()
                if k.split(".")[0] not in self._skip_layers or ".lora_layer." in k}

    def load_state_dict(self, state_dict, strict: bool = True):
        self._own = dict(state_dict)            # CL:608-614: tied keys are back-filled from the UNet at build time
        self._engine = None

    def full_state_dict(self):
        if self._unet_sd is None:
            raise EdgeStyleHipError("ControlLoRAModel.tie_weights(unet) must be called before use (TT:259-261)")
        sd = {k: v for k, v in self._unet_sd.items() if k.split(".")[0] in _ENC_PREFIXES}
        sd.update(self._own)
        return sd

    def preprocess_image(self, image: torch.Tensor, noise: Optional[torch.Tensor] = None,
                         generator: Optional[torch.Generator] = None) -> torch.Tensor:
        if not self.config.uses_vae:
            return super().preprocess_image(image)
        if self._vae is None:
            raise ValueError("vae must be provided if any of the controlnets uses a vae")     # MC:388-391
        vae = self._vae
        if vae.device != self.device:
            vae.to(self.device)
        dist = vae.encode(image).latent_dist
        N = image.shape[0]
        h, w = dist.moments.shape[1:3]
        if noise is None:   # the reference draws from the global RNG here (CL:39); we take a generator instead
            noise = torch.randn((N, vae.cfg.latent_channels, h, w), generator=generator, dtype=torch.float32)
        z = dist.sample_scaled_nhwc(noise, self.engine.in_pad, vae.cfg.scaling_factor)
        return _as_nchw_view(self.engine.embed_latent(z))

    def fuse(self) -> "FusedControlLoRAModel":
        """CL:739-777, into private tensors."""
        sd = E.fold_lora(self.full_state_dict())
        m = FusedControlLoRAModel(sd, self.cfg, self.dtype, uses_vae=bool(self.config.uses_vae))
        m._vae = self._vae
        m.to(self.device)
        return m


class FusedControlLoRAModel(ControlLoRAModel):
    """Plain ControlNet holding W+BA (CL:292-375)."""

    def __init__(self, state_dict, cfg=None, torch_dtype=None, uses_vae: bool = False, **kw):
        super().__init__(state_dict, cfg, torch_dtype, lora_linear_rank=0, uses_vae=uses_vae)

    def full_state_dict(self):
        return self._own

    def state_dict(self):
        return self._own


# ----------------------------------------------------------------------------------------------------------------
class ControlNetBlock:
    """Parameter container for one fusion block (MC:23-53); the arithmetic is es_fusion_block."""

    def __init__(self, output_channel: int, size: Tuple[int, int], num_controlnets: int):
        if num_controlnets != 6:
            raise ValueError("the HIP fusion kernel implements the reference's 6-net configuration")
        self.output_channel, self.size, self.num_controlnets = output_channel, tuple(size), num_controlnets


class EdgeStyleMultiControlNetModel(_HipModel):
    """model/edgestyle_multicontrolnet.py:66-430."""

    def __init__(self, controlnets: Sequence[ControlNetModel], cfg: UNetConfig = None, sample_size: int = None):
        self.nets = list(controlnets)
        self.cfg = cfg or self.nets[0].cfg
        table = self.cfg.residual_table(sample_size)
        # MC:73-102 (hard-wired SD1.5@512 in the reference; derived from the config here, identical for sd15())
        self.down_output_channels = [c for c, _ in table[:-1]]
        self.down_sizes = [(s, s) for _, s in table[:-1]]
        self.mid_output_channels, self.mid_size = table[-1][0], (table[-1][1],) * 2
        self.multi_controlnet_down_blocks = [ControlNetBlock(c, (s, s), len(self.nets)) for c, s in table[:-1]]
        self.multi_controlnet_mid_block = ControlNetBlock(table[-1][0], (table[-1][1],) * 2, len(self.nets))
        self._sample_size = sample_size
        self._fusion_sd = None
        self.dtype = self.nets[0].dtype
        self.device = self.nets[0].device
        self.config = _Config(global_pool_conditions=False)

    # -- weights (MC:173-211) -----------------------------------------------------------------------------------
    def state_dict(self):
        return dict(self._fusion_sd or {})

    def load_state_dict(self, state_dict, *a, **kw):
        want = W.fusion_shapes(self.cfg, len(self.nets), self._sample_size)
        sd = {k: v for k, v in state_dict.items()
              if k.startswith("multi_controlnet_down_blocks.") or k.startswith("multi_controlnet_mid_block.")}
        missing = [k for k in want if k not in sd]
        if missing:
            raise RuntimeError(f"Missing key(s) in state_dict: {missing[:4]}...")
        for k, s in want.items():
            if tuple(sd[k].shape) != tuple(s):
                raise RuntimeError(f"size mismatch for {k}: {tuple(sd[k].shape)} vs {tuple(s)}")
        self._fusion_sd = sd
        self._engine = None

    def to(self, device=None, dtype=None, **kw):
        super().to(device, dtype)
        seen = set()
        for n in self.nets:
            if id(n) not in seen:
                seen.add(id(n))
                n.to(device, dtype)
        return self

    def fuse(self):
        """MC:284-287"""
        done = {}
        for i, n in enumerate(self.nets):
            if isinstance(n, ControlLoRAModel) and not isinstance(n, FusedControlLoRAModel):
                if id(n) not in done:
                    done[id(n)] = n.fuse()
                self.nets[i] = done[id(n)]

    def save_pretrained(self, save_directory, save_pattern=None, **kw):
        """MC:213-282"""
        if os.path.isfile(save_directory):
            raise ValueError(f"Provided path ({save_directory}) should be a directory, not a file")
        W.save_model_dir(save_directory, self.state_dict())
        save_pattern = save_pattern or [None] * len(self.nets)
        saved = []
        for i, net in enumerate(self.nets):
            if save_pattern[i] is not None and save_pattern[i] not in saved:
                net.save_pretrained(os.path.join(save_directory, f"controlnet_{save_pattern[i]}"))
                saved.append(save_pattern[i])

    @classmethod
    def from_pretrained(cls, pretrained_model_path, **kwargs):
        """MC:289-430: fusion weights at <dir>/diffusion_pytorch_model.safetensors, LoRA nets at <dir>/controlnet_{idx}."""
        controlnet_class = kwargs.pop("controlnet_class", ControlNetModel)
        torch_dtype = kwargs.pop("torch_dtype", None)
        if not os.path.isdir(pretrained_model_path):
            raise ValueError(f"Provided path ({pretrained_model_path}) should be a directory")
        if "load_pattern" not in kwargs:
            raise ValueError("load_pattern must be provided")
        load_pattern = kwargs["load_pattern"]
        static = kwargs.get("static_controlnets", [None] * len(load_pattern))
        vae = kwargs.get("vae")
        loaded, nets = {}, []
        for i, load in enumerate(load_pattern):
            if load is not None:
                if load not in loaded:
                    net = controlnet_class.from_pretrained(os.path.join(pretrained_model_path, f"controlnet_{load}"),
                                                           torch_dtype=torch_dtype)
                    if net.config.uses_vae:
                        if vae is None:
                            raise ValueError("vae must be provided if any of the controlnets uses a vae")
                        net.set_autoencoder(vae)
                    loaded[load] = net
                nets.append(loaded[load])
            else:
                nets.append(static[i])
        for i, n in enumerate(nets):
            if n is None:
                raise ValueError(f"All controlnets must be provided. controlnet {i} is None.")
        model = cls(nets, sample_size=kwargs.get("sample_size"))
        sd, _ = W.load_model_dir(pretrained_model_path)
        model.load_state_dict(sd)
        return model

    # -- compute ------------------------------------------------------------------------------------------------
    @property
    def engine(self) -> E.Fusion:
        self._require_gpu()
        if self._fusion_sd is None:
            raise EdgeStyleHipError("fusion block weights not loaded (load_state_dict / from_pretrained)")
        if self._engine is None:
            self._engine = E.Fusion(self._fusion_sd, self.cfg, self.dtype, self.device, self._sample_size)
        return self._engine

    def groups(self) -> List[Tuple[ControlNetModel, List[int]]]:
        """Nets that are the same object share one batched pass: [(net, [positions])]."""
        out: List[Tuple[ControlNetModel, List[int]]] = []
        for i, n in enumerate(self.nets):
            for g in out:
                if g[0] is n:
                    g[1].append(i)
                    break
            else:
                out.append((n, [i]))
        return out

    def __call__(self, sample, timestep, encoder_hidden_states, controlnet_cond: List[torch.Tensor],
                 conditioning_scale: List[float], guess_mode: bool = False, return_dict: bool = True, **unused):
        """MC:116-171.  Returns (12 tensors, 1 tensor), NCHW-shaped.  guess_mode is handed to every net (MC:136-149)."""
        if len(controlnet_cond) != len(self.nets) or len(conditioning_scale) != len(self.nets):
            raise ValueError("controlnet_cond and conditioning_scale must have one entry per ControlNet")
        N = sample.shape[0]
        dev, dt = self.device, self.dtype
        ehs = encoder_hidden_states.to(dev, dt).contiguous()
        res_per_net, bs = [None] * len(self.nets), [None] * len(self.nets)
        for net, pos in self.groups():
            eng = net.engine
            k = len(pos)
            x = _as_nhwc(sample, dt, dev, eng.in_pad)
            conds = []
            for p in pos:
                c = controlnet_cond[p]
                if tuple(c.shape[2:]) != tuple(sample.shape[2:]):
                    c = net.preprocess_image(c)
                conds.append(_as_nhwc(c, dt, dev))
            tproj = eng.time_proj(_timestep_tensor(timestep, N, dev).repeat(k))
            ctx = eng.context(ehs.repeat(k, 1, 1) if k > 1 else ehs)
            res = eng.forward(x, tproj, ctx, conds,
                              level_scales=_guess_level_scales(len(eng.zero) + 1) if guess_mode else None)
            for j, p in enumerate(pos):
                res_per_net[p] = [r[j * N:] for r in res]
                bs[p] = [r.stride(0) for r in res]
        outs = self.engine.forward(res_per_net, bs, N, [float(s) for s in conditioning_scale])
        down = [_as_nchw_view(o) for o in outs[:-1]]
        mid = _as_nchw_view(outs[-1])
        if not return_dict:
            return down, mid
        return SimpleNamespace(down_block_res_samples=down, mid_block_res_sample=mid)

    forward = __call__


# ----------------------------------------------------------------------------------------------------------------
class StepState:
    """Every device buffer a (captured) denoising step reads that outlives one pipeline call: the cross-attention K/V
    projections of the text states, the batch-concatenated condition embeddings / replicated sample of the grouped
    conv_in launch, and the per-call time-projection table.  One StepState belongs to ONE `pipeline._Loop` (one graph):
    the buffers are allocated once, refilled in place, and `signature()` lists their addresses so the loop re-captures
    its graph if any of them ever moves (a graph replays the pointers it was captured with)."""

    def __init__(self):
        self.ctx_unet: Optional[List[torch.Tensor]] = None
        self.ctx_nets: Optional[List[List[torch.Tensor]]] = None
        self.ctx_grouped: Optional[List[torch.Tensor]] = None
        self.ctx_guess: Optional[List[List[torch.Tensor]]] = None
        self.cond_cat = None
        self.cond_src = None
        self.tproj_table = self.tproj_cur = self.tproj_gen = None
        self.fused_zero: Optional[List[torch.Tensor]] = None      # StepRunner.prepare_fused_zero

    def signature(self) -> Tuple[int, ...]:
        sig = []

        def walk(o):
            if torch.is_tensor(o):
                sig.append(o.data_ptr())
            elif isinstance(o, (list, tuple)):
                for x in o:
                    walk(x)
            else:
                sig.append(0)
        for name in ("ctx_unet", "ctx_nets", "ctx_grouped", "ctx_guess", "cond_cat", "tproj_table", "tproj_cur",
                     "tproj_gen", "fused_zero"):
            walk(getattr(self, name))
        return tuple(sig)


class StepRunner:
    """controlnet -> unet for one timestep == OnnxUNetAndControlnets.forward (export_onnx.py:43-74), with all
    per-image constants (text K/V projections, conditioning embeddings) hoisted out and static buffers (StepState) so
    that the call sequence is hipGraph-capturable.  Inputs/outputs NHWC.

    `controlnet` is the reference's fused 6-net EdgeStyleMultiControlNetModel, or ONE plain ControlNetModel (PL:338-351:
    its 13 residuals go to the UNet without fusion blocks - BASELINE configs[0])."""

    def __init__(self, unet: UNet2DConditionModel, controlnet):
        self.unet, self.controlnet = unet, controlnet
        self.device, self.dtype = unet.device, unet.dtype
        self.single = isinstance(controlnet, ControlNetModel)
        # nets sharing weights run as ONE batched chain (measured: 7 unbatched chains are 12 % slower at batch 1)
        self.groups = [(controlnet, [0])] if self.single else controlnet.groups()
        self.kmax = max(len(p) for _, p in self.groups)
        # execution of the four independent encoder chains of a step:
        #   "grouped" — one lockstep pass of grouped launches over the batch-concatenated activations (default)
        #   "serial"  — one chain after the other (profiling; also the fall-back for shapes whose groups do not tile)
        # (a third mode, one HIP stream per chain, measured 15.2 vs 13.1 ms per step and was removed in round 3)
        self.mode = os.environ.get("ES_CHAIN_MODE", "serial" if os.environ.get("ES_SERIAL") == "1" else "grouped")
        if self.mode not in ("grouped", "serial"):
            raise ValueError(f"ES_CHAIN_MODE={self.mode!r}: 'grouped' or 'serial'")
        self._grouped = None
        # every buffer a step reads between calls lives in a StepState; `state` is the one the next step() uses
        # (the pipeline switches it to the running loop's; stand-alone callers get this default one)
        self.state = StepState()
        self.keep_debug = False          # tests: keep the last step's decoder inputs (fused residuals) in self.debug
        self.debug = None

    @classmethod
    def from_state_dicts(cls, ws: Dict[str, Dict[str, torch.Tensor]], ucfg: UNetConfig, dtype, device,
                         rank: int = 4, sample_size: int = None):
        """Builds the reference's net list [lora0, pose, lora1, pose, lora1, pose] (TT:252-261) from raw dicts."""
        unet = UNet2DConditionModel(ws["unet"], ucfg, dtype).to(device)
        pose = ControlNetModel(ws["openpose"], ucfg, dtype).to(device)
        l0 = ControlLoRAModel(ws["lora0"], ucfg, dtype, lora_linear_rank=rank, uses_vae=True)
        l1 = ControlLoRAModel(ws["lora1"], ucfg, dtype, lora_linear_rank=rank, uses_vae=True)
        for n in (l0, l1):
            n.tie_weights(unet)
            n.to(device)
        mc = EdgeStyleMultiControlNetModel([l0, pose, l1, pose, l1, pose], ucfg, sample_size)
        mc.load_state_dict(ws["fusion"])
        mc.to(device)
        return cls(unet, mc)

    def set_context(self, ehs: torch.Tensor, guess_mode: bool = False, n_cn: Optional[int] = None):
        """ehs: [N,77,D] device dtype. Computes every cross-attention K/V projection once (constant over the loop),
        in place into the current StepState's buffers (allocated on first use): C-ABI launches only, so the whole
        preparation can be recorded into a native plan (edgestyle_amd/native.py).  The K/V tensors of the encoder
        transformers live batch-concatenated in group order [net groups..., UNet] (what the grouped lockstep pass reads);
        the per-net and UNet contexts are views of their slices.  The k nets of a weight-sharing group see the same text
        states: one launch writes their k copies (es_gemm_desc.x_nmod).
        guess_mode: the ControlNets see the last n_cn rows of ehs — under CFG only the conditional half (PL:453-459)."""
        st = self.state
        N = ehs.shape[0]
        ue = self.unet.engine
        utr = ue.transformers()
        n_enc = len(self.groups[0][0].engine.transformers())
        T77, dev, dt = ehs.shape[1], self.device, self.dtype
        ks = [len(pos) for _, pos in self.groups]
        if guess_mode:
            Nc = n_cn or N
            ehs_c = ehs[N - Nc:]
            if st.ctx_unet is None or st.ctx_unet[0].shape[0] != N:
                st.ctx_unet = [torch.empty((N, T77, t.kv2.cout), dtype=dt, device=dev) for t in utr]
            for t, o in zip(utr, st.ctx_unet):
                t.context(ehs, o)
            if st.ctx_guess is None or st.ctx_guess[0][0].shape[0] != ks[0] * Nc:
                st.ctx_guess = [[torch.empty((k * Nc, T77, t.kv2.cout), dtype=dt, device=dev) for t in net.engine.transformers()]
                                for (net, _), k in zip(self.groups, ks)]
            for (net, _), k, outs in zip(self.groups, ks, st.ctx_guess):
                for t, o in zip(net.engine.transformers(), outs):
                    t.context(ehs_c, o, rep=k)
            return
        ntot = (sum(ks) + 1) * N
        if st.ctx_grouped is None or st.ctx_grouped[0].shape[0] != ntot:
            st.ctx_grouped = [torch.empty((ntot, T77, utr[i].kv2.cout), dtype=dt, device=dev) for i in range(n_enc)]
            st.ctx_nets, a = [], 0
            for k in ks:
                st.ctx_nets.append([g[a:a + k * N] for g in st.ctx_grouped])
                a += k * N
            st.ctx_unet = [g[a:a + N] for g in st.ctx_grouped] + \
                [torch.empty((N, T77, t.kv2.cout), dtype=dt, device=dev) for t in utr[n_enc:]]
        for (net, _), k, outs in zip(self.groups, ks, st.ctx_nets):
            for t, o in zip(net.engine.transformers(), outs):
                t.context(ehs, o, rep=k)
        for t, o in zip(utr, st.ctx_unet):
            t.context(ehs, o)

    def _grouped_encoder(self, N: int):
        ue = self.unet.engine
        encs = [net.engine for net, _ in self.groups] + [ue]
        counts = [len(pos) * N for _, pos in self.groups] + [N]
        if self._grouped is None or self._grouped.counts != counts:
            self._grouped = E.GroupedEncoder(encs, counts)
        return self._grouped

    def set_time_table(self, timesteps: torch.Tensor, N: int):
        """timesteps: fp32 device [T].  Fills tproj_table[s] = what GroupedEncoder.time_proj would produce at step s
        (16 launches per call instead of ~20 per step); the StepState's buffers are reused in place while T is unchanged."""
        ge = self._grouped_encoder(N)
        st = self.state
        T = int(timesteps.shape[0])
        shape = (T, ge.ntot, ge.width)
        if st.tproj_table is None or tuple(st.tproj_table.shape) != shape:
            st.tproj_table = torch.zeros(shape, dtype=self.dtype, device=self.device)
            st.tproj_cur = torch.zeros(shape[1:], dtype=self.dtype, device=self.device)
        es = st.tproj_table.element_size()
        a = 0
        for e, n in zip(ge.encs, ge.counts):
            proj = e.time_proj(timesteps)                                # [T, width_e]
            for j in range(n):                                           # every sample row of the group: a strided copy
                ops.memcpy2d(st.tproj_table[0, a + j].data_ptr(), ge.ntot * ge.width * es, proj.data_ptr(),
                             e.tproj_width * es, e.tproj_width * es, T)
            a += n

    def set_conds(self, conds: Sequence[torch.Tensor]):
        """conds: the embedded conditions [N,h,w,C0] of this call (constant over the loop)."""
        N, H, W, C0 = conds[0].shape
        ge = self._grouped_encoder(N)
        st = self.state
        if st.cond_cat is None or tuple(st.cond_cat.shape) != (ge.ntot, H, W, C0):
            st.cond_cat = torch.zeros((ge.ntot, H, W, C0), dtype=self.dtype, device=self.device)
        st.cond_src = [c.data_ptr() for c in conds]      # the step only trusts cond_cat for these very buffers
        a = 0
        for _, pos in self.groups:
            for p in pos:
                ops.memcpy(st.cond_cat[a:a + N], conds[p])
                a += N                                   # the UNet's slot stays zero

    def refresh_cond(self, p: int, cond: torch.Tensor):
        """One net's embedded condition changed (per-step re-sampling, pipeline._Loop.resample_conds): its slot of the
        batch-concatenated tensor the grouped conv_in launch reads follows."""
        st = self.state
        if st.cond_cat is None:
            return
        N = cond.shape[0]
        a = 0
        for _, pos in self.groups:
            for q in pos:
                if q == p:
                    ops.memcpy(st.cond_cat[a:a + N], cond)
                    return
                a += N

    def clear_time_table(self):
        self.state.tproj_table = self.state.tproj_cur = None

    @property
    def n_nets(self) -> int:
        return 1 if self.single else len(self.controlnet.nets)

    def _fuse(self, results, N: int, scales, scales_dev, addends=None):
        """Per-group ControlNet residuals -> the 13 tensors the UNet adds (MC:151-169).  Single ControlNet: they are the
        net's own (already scaled) residuals."""
        if self.single:
            return results[0]
        nn = self.n_nets
        res_per_net, bs = [None] * nn, [None] * nn
        for gi, (net, pos) in enumerate(self.groups):
            res = results[gi]
            for j, p in enumerate(pos):
                res_per_net[p] = [r[j * N:] for r in res]
                bs[p] = [r.stride(0) for r in res]
        return self.controlnet.engine.forward(res_per_net, bs, N, scales, scales_dev, addends=addends)

    def _cn_scale(self, scales, scales_dev):
        """Single ControlNet: the conditioning scale rides on the zero-conv epilogues (CL:266-270)."""
        if not self.single:
            return {}
        return dict(out_scale=float(scales[0]), out_scale_dev=None if scales_dev is None else scales_dev[0:1])

    def _step_guess(self, x, t_rows, conds, scales, scales_dev, out):
        """guess_mode (CL:256-264): the 13 residual levels are scaled 0.1..1 log-spaced.  Under CFG (PL:453-459,
        487-497) the ControlNets additionally run on the conditional half only (conds then hold B samples) and the
        unconditional half of the UNet gets no residuals."""
        st = self.state
        N = x.shape[0]
        Bc = conds[0].shape[0]
        B = N - Bc                                       # first row of the ControlNet batch inside x (0 without CFG)
        ue = self.unet.engine
        xc = x[B:]
        ls = _guess_level_scales(len(ue.cfg.residual_table()))
        results = {}
        for gi, (net, pos) in enumerate(self.groups):
            eng = net.engine
            tproj = eng.time_proj(t_rows[: len(pos) * Bc])
            results[gi] = eng.forward(xc, tproj, st.ctx_guess[gi], [conds[p] for p in pos], level_scales=ls,
                                      **self._cn_scale(scales, scales_dev))
        fused = self._fuse(results, Bc, scales, scales_dev)
        tproj = ue.time_proj(t_rows[:N])
        skips, h = ue.encode(x, tproj, st.ctx_unet)
        for s_, f in zip(skips, fused[:-1]):              # torch.cat([zeros, d]) + skip == add into the cond half
            ops.add(s_[B:], f.reshape(s_[B:].shape), out=s_[B:])
        ops.add(h[B:], fused[-1].reshape(h[B:].shape), out=h[B:])
        return ue.forward(x, tproj, st.ctx_unet, out=out, encoded=(skips, h))

    # ---- steps outside every control-guidance window (PL:419-427: controlnet_keep = 0 for all nets) ---------------------
    def prepare_fused_zero(self, N: int):
        """What the 13 fusion blocks produce when every net's conditioning scale is 0: the reference still runs the six
        ControlNets, multiplies their residuals by cond_scale * keep = 0 (PL:464-470, CL:266-270) and sends the zeros
        through interleave + ControlNetBlock (MC:151-169) - whose biases and LayerNorm planes make the result NON-zero, but
        the same for every sample, timestep and image: computed here once per StepState (eagerly, outside any capture),
        added by step_unet_only in place of six encoder passes and the fusion launches."""
        st = self.state
        if self.single:
            st.fused_zero = None
            return
        fe = self.controlnet.engine
        if st.fused_zero is not None and st.fused_zero[0].shape[0] == N:
            return
        zs = [torch.zeros((N, s * s, c), dtype=self.dtype, device=self.device) for c, s in fe.table]
        res_per_net = [zs] * self.n_nets
        bs = [[z.stride(0) for z in zs]] * self.n_nets
        st.fused_zero = [f.clone() for f in fe.forward(res_per_net, bs, N, [0.0] * self.n_nets, None)]

    def step_unet_only(self, x: torch.Tensor, t_rows: torch.Tensor, out: Optional[torch.Tensor] = None,
                       step_idx: Optional[torch.Tensor] = None) -> torch.Tensor:
        """A denoising step in which no ControlNet contributes (all controlnet_keep = 0): the UNet alone, its skip / mid
        tensors plus the constant fused residuals of prepare_fused_zero (a single plain ControlNet: plus nothing)."""
        st = self.state
        N = x.shape[0]
        ue = self.unet.engine
        tproj = None
        if self.mode == "grouped" and step_idx is not None and st.tproj_table is not None:
            ge = self._grouped_encoder(N)
            if st.tproj_table.shape[1] == ge.ntot:
                T = st.tproj_table.shape[0]
                ops.gather_row(st.tproj_table.view(T, -1).view(torch.float32), step_idx, st.tproj_cur.view(-1).view(torch.float32))
                tproj = st.tproj_cur[ge.ntot - N:]
        if tproj is None:
            tproj = ue.time_proj(t_rows[:N])
        enc = ue.encode(x, tproj, st.ctx_unet)
        if self.single:
            return ue.forward(x, tproj, st.ctx_unet, out=out, encoded=enc)
        if st.fused_zero is None or st.fused_zero[0].shape[0] != N:
            raise EdgeStyleHipError("step_unet_only: call prepare_fused_zero(N) first (outside graph capture)")
        fz = st.fused_zero
        return ue.forward(x, tproj, st.ctx_unet, fz[:-1], fz[-1], out=out, encoded=enc)

    def step(self, x: torch.Tensor, t_rows: torch.Tensor, conds: Sequence[torch.Tensor], scales: Sequence[float],
             scales_dev: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
             step_idx: Optional[torch.Tensor] = None, guess_mode: bool = False) -> torch.Tensor:
        """x: [N,h,w,8] NHWC; t_rows: fp32 device [kmax*N] (all equal to the timestep); conds: n_nets x [N,h,w,C0] NHWC.

        The three batched ControlNet passes and the UNet's own down+mid path do not depend on each other (the
        residuals are only added after the UNet's down path, PL:500-510): by default they run in lockstep as grouped
        launches (_step_grouped); ES_CHAIN_MODE=serial runs them one after the other."""
        st = self.state
        N = x.shape[0]
        if guess_mode:
            return self._step_guess(x, t_rows, conds, scales, scales_dev, out)
        ue = self.unet.engine
        results = {}
        if self.mode == "grouped":
            out_t = self._step_grouped(x, t_rows, conds, scales, scales_dev, out, step_idx)
            if out_t is not None:
                return out_t

        def cn_chain(gi):
            net, pos = self.groups[gi]
            eng = net.engine
            tproj = eng.time_proj(t_rows[: len(pos) * N])
            results[gi] = eng.forward(x, tproj, st.ctx_nets[gi], [conds[p] for p in pos], **self._cn_scale(scales, scales_dev))

        def unet_chain():
            tproj = ue.time_proj(t_rows[:N])
            results["unet"] = (tproj, ue.encode(x, tproj, st.ctx_unet))

        chains = [lambda gi=gi: cn_chain(gi) for gi in range(len(self.groups))] + [unet_chain]
        for fn in chains:
            fn()
        fused = self._fuse(results, N, scales, scales_dev)
        tproj, enc = results["unet"]
        if self.keep_debug:
            self.debug = dict(fused=[f.clone() for f in fused])
        return ue.forward(x, tproj, st.ctx_unet, fused[:-1], fused[-1], out=out, encoded=enc)

    def _step_grouped(self, x, t_rows, conds, scales, scales_dev, out, step_idx=None):
        """The three batched ControlNet passes and the UNet encoder as ONE lockstep pass of grouped launches."""
        st = self.state
        N = x.shape[0]
        ue = self.unet.engine
        ge = self._grouped_encoder(N)
        encs, counts = ge.encs, ge.counts
        hw_min = (x.shape[1] >> (len(ue.cfg.block_out_channels) - 1)) * (x.shape[2] >> (len(ue.cfg.block_out_channels) - 1))
        if not ge.groupable(hw_min):
            return None                                   # tiny shapes: groups do not tile in 128-pixel units
        ncn = sum(counts[:-1])
        c0 = ue.conv_in.cout
        h0 = torch.empty((ge.ntot, x.shape[1], x.shape[2], c0), dtype=x.dtype, device=x.device)
        if st.cond_cat is not None and st.cond_cat.shape == h0.shape and [c.data_ptr() for c in conds] == st.cond_src:
            # sample = conv_in(sample) + cond (CL:197-203) for every net, and the UNet's conv_in, in one grouped launch:
            # every slot reads the same sample tensor (x_rep), the conditions are batch-concatenated in slot order
            ops.conv_gemm(x, [e.conv_in for e in encs], residual=st.cond_cat, group_n=counts, out=h0, x_rep=ge.ntot // N, wide=True,
                          gn_groups=ue.cfg.norm_num_groups)
        else:
            a = 0
            wide = ops.wide_stream(x.dtype)
            if wide:                                      # the sums' low parts, slice by slice like h0 (the UNet's slot has none: zero)
                h0._lo = torch.zeros_like(h0)
            G = ue.cfg.norm_num_groups                    # ... and the GroupNorm statistics of the slices, into one table (ops.GN_HANDOVER)
            hw = x.shape[1] * x.shape[2]
            gnp = torch.empty((ge.ntot, 2 * (hw // 64), G, 2), dtype=torch.float32, device=x.device) \
                if ops.gn_handover(hw, c0, G) else None
            gkw = lambda a_: dict(gn_groups=G, gn_part=gnp[a_:a_ + N]) if gnp is not None else {}
            for net, pos in self.groups:                  # sample = conv_in(sample) + cond   (CL:197-203)
                for p in pos:
                    ops.conv_gemm(x, net.engine.conv_in, residual=conds[p], out=h0[a:a + N], wide=True,
                                  out_lo=h0._lo[a:a + N] if wide else None, **gkw(a))
                    a += N
            ops.conv_gemm(x, ue.conv_in, out=h0[a:a + N], **gkw(a))
            if gnp is not None:
                h0._gnp = (gnp, G)
        if step_idx is not None and st.tproj_table is not None and st.tproj_table.shape[1] == ge.ntot:
            # one gather instead of 4 x (sinusoid + 3 linears) per step; fp16/bf16 rows moved as fp32 words
            T = st.tproj_table.shape[0]
            ops.gather_row(st.tproj_table.view(T, -1).view(torch.float32), step_idx, st.tproj_cur.view(-1).view(torch.float32))
            tproj = st.tproj_cur
        else:
            if st.tproj_gen is None or st.tproj_gen.shape != (ge.ntot, ge.width):
                st.tproj_gen = torch.zeros((ge.ntot, ge.width), dtype=self.dtype, device=self.device)
            tproj = ge.time_proj(t_rows, st.tproj_gen)
        cn_counts = counts[:-1]
        cn_engs = encs[:-1]
        nn = self.n_nets

        def zero_and_fuse(levels, srcs, addends, first_level):
            """zero-convs of the given levels (grouped over the ControlNets) -> their fusion blocks (+ the UNet's own tensors)"""
            res = [ops.conv_gemm(srcs[k][:ncn], [e.zero[lvl] if lvl < len(e.zero) else e.zero_mid for e in cn_engs], group_n=cn_counts)
                   for k, lvl in enumerate(levels)]
            res_per_net, bs = [None] * nn, [None] * nn
            a = 0
            for net, pos in self.groups:
                for p in pos:
                    res_per_net[p] = [r[a:] for r in res]
                    bs[p] = [r.stride(0) for r in res]
                    a += N
            # the fusion kernel adds the UNet's own skip / mid tensors: its outputs ARE the decoder's inputs
            return self.controlnet.engine.forward(res_per_net, bs, N, scales, scales_dev, addends=addends, first_level=first_level)

        skips, h = ge.run(h0, tproj, st.ctx_grouped)
        enc = ([s[ncn:] for s in skips], h[ncn:])
        if self.single:
            # one ControlNet: skip + scale * zero_conv(cn_skip) straight out of the zero-conv epilogue (PL:500-510)
            kw = self._cn_scale(scales, scales_dev)
            e = cn_engs[0]
            fused = [ops.conv_gemm(s[:ncn], e.zero[i], residual=enc[0][i], **kw) for i, s in enumerate(skips)]
            fused.append(ops.conv_gemm(h[:ncn], e.zero_mid, residual=enc[1], **kw))
            return ue.forward(x, tproj[ncn:], st.ctx_unet, fused[:-1], fused[-1], out=out, encoded=enc, presummed=True)
        srcs = skips + [h]
        fused = zero_and_fuse(list(range(len(srcs))), srcs, enc[0] + [enc[1]], 0)
        if self.keep_debug:
            self.debug = dict(presummed=[f.clone() for f in fused], skips=[e.clone() for e in enc[0] + [enc[1]]])
        return ue.forward(x, tproj[ncn:], st.ctx_unet, fused[:-1], fused[-1], out=out, encoded=enc, presummed=True)

    def step_nchw(self, sample, timestep, ehs, conds, scales):
        """Convenience for tests: NCHW fp32 in / NCHW out (on a StepState of its own: a pipeline loop's is left alone)."""
        self.state = StepState()
        N = sample.shape[0]
        dev, dt = self.device, self.dtype
        self.set_context(ehs.to(dev, dt).contiguous())
        x = _as_nhwc(sample, dt, dev, self.unet.engine.in_pad)
        t_rows = torch.full((self.kmax * N,), float(timestep), dtype=torch.float32, device=dev)
        c = [_as_nhwc(ci, dt, dev) for ci in conds]
        return _as_nchw_view(self.step(x, t_rows, c, [float(s) for s in scales]))
