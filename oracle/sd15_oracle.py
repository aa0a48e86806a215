"""CPU ORACLE — test infrastructure only.  PINNED for the arithmetic the reference itself owns (ControlNetBlock, interleave_*:
tests/golden/ref_fusion.safetensors holds outputs of the reference's OWN code, tests/golden/make_golden_ref_fusion.py);
PARITY UNPINNED for the diffusers-owned blocks (see below).

A plain-PyTorch fp32 restatement of EdgeStyle's 6-condition multi-ControlNet SD1.5 denoising path.  It is the
checker for the HIP path: only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import
it; the product package `edgestyle_amd` never does.

What is restated from the reference's own source (behaviour line by line):
  * ControlNetBlock / interleave / EdgeStyleMultiControlNetModel.forward
        model/edgestyle_multicontrolnet.py:23-63, 116-171, 479-514
  * CachedControlNetModel.forward (cached-cond shortcut), VAEControlNetConditioningEmbedding, LoRA attach set,
    weight tying, fuse          model/controllora.py:28-42, 58-290, 529-598, 623-632, 728-777
  * the pipeline loop, CFG, prepare_image / prepare_latents
        model/edgestyle_pipeline.py:329-330, 352-398, 419-427, 435-522, 552-572, 585-664

What is restated from the PUBLISHED ARCHITECTURE of the third-party dependency `diffusers==0.26.3`
(requirements-jetson.txt:25; absent from /root/reference and not installable offline — SURVEY.md §8c):
UNet2DConditionModel, ControlNetModel, AutoencoderKL, LoRACompatibleLinear, DDIMScheduler, Timesteps.
No golden vector, known-answer test or importable reference pins those blocks here, so their parity is
**unpinned**; state-dict key names are diffusers-exact so that one real checkpoint + one real diffusers run
can pin them later.  The reference's only numeric anchor is the tolerance policy export_onnx.py:329-334
(rtol 1e-3 / atol 1e-5, observed miss 9.2e-4 abs, README.md:237-251).

Everything is functional: `fn(sd, prefix, x, ...)` with `sd` a {key: fp32 tensor} dict.
"""
import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]


# ----------------------------------------------------------------------------------------------------------------
# primitives
# ----------------------------------------------------------------------------------------------------------------
def conv(sd: SD, p: str, x, stride=1, padding=1):
    return F.conv2d(x, sd[p + ".weight"], sd.get(p + ".bias"), stride=stride, padding=padding)


def linear(sd: SD, p: str, x):
    """LoRACompatibleLinear: y = xW^T + b + (x A^T) B^T, no alpha scaling (CL:577-593, network_alpha=None)."""
    y = F.linear(x, sd[p + ".weight"], sd.get(p + ".bias"))
    dk = p + ".lora_layer.down.weight"
    if dk in sd:
        y = y + F.linear(F.linear(x, sd[dk]), sd[p + ".lora_layer.up.weight"])
    return y


def group_norm(sd: SD, p: str, x, groups, eps):
    return F.group_norm(x, groups, sd[p + ".weight"], sd[p + ".bias"], eps)


def layer_norm(sd: SD, p: str, x, eps=1e-5):
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], eps)


def timestep_sinusoid(t: torch.Tensor, dim: int) -> torch.Tensor:
    """diffusers Timesteps(dim, flip_sin_to_cos=True, downscale_freq_shift=0) -> [cos | sin] (CL:150)."""
    half = dim // 2
    freqs = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32, device=t.device) / half)
    args = t.float()[:, None] * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


def time_embedding(sd: SD, t: torch.Tensor, cfg, batch: int) -> torch.Tensor:
    """CL:134-157: scalar/0-d timestep -> expand to batch -> sinusoid -> linear_1 -> SiLU -> linear_2."""
    if not torch.is_tensor(t):
        t = torch.tensor([t], dtype=torch.float32)
    t = t.reshape(-1).float()
    if t.numel() == 1:
        t = t.expand(batch)
    emb = timestep_sinusoid(t, cfg.block_out_channels[0])
    emb = linear(sd, "time_embedding.linear_1", emb)
    emb = F.silu(emb)
    return linear(sd, "time_embedding.linear_2", emb)


def resnet(sd: SD, p: str, x, temb, groups, eps):
    h = F.silu(group_norm(sd, p + ".norm1", x, groups, eps))
    h = conv(sd, p + ".conv1", h)
    if temb is not None:
        h = h + linear(sd, p + ".time_emb_proj", F.silu(temb))[:, :, None, None]
    h = F.silu(group_norm(sd, p + ".norm2", h, groups, eps))
    h = conv(sd, p + ".conv2", h)
    if (p + ".conv_shortcut.weight") in sd:
        x = conv(sd, p + ".conv_shortcut", x, padding=0)
    return x + h


def attention(sd: SD, p: str, x, ctx, heads):
    """diffusers Attention: q/k/v (no bias in UNet, bias in VAE), softmax(qk^T/sqrt(d))v, to_out.0."""
    q = linear(sd, p + ".to_q", x)
    k = linear(sd, p + ".to_k", ctx)
    v = linear(sd, p + ".to_v", ctx)
    b, s, c = q.shape
    d = c // heads
    q = q.view(b, s, heads, d).transpose(1, 2)
    k = k.view(b, -1, heads, d).transpose(1, 2)
    v = v.view(b, -1, heads, d).transpose(1, 2)
    o = F.scaled_dot_product_attention(q, k, v)
    o = o.transpose(1, 2).reshape(b, s, c)
    return linear(sd, p + ".to_out.0", o)


def transformer(sd: SD, p: str, x, ehs, heads, groups):
    """Transformer2DModel (use_linear_projection=False) with one BasicTransformerBlock; GroupNorm eps 1e-6."""
    b, c, hh, ww = x.shape
    res = x
    h = group_norm(sd, p + ".norm", x, groups, 1e-6)
    h = conv(sd, p + ".proj_in", h, padding=0)
    h = h.permute(0, 2, 3, 1).reshape(b, hh * ww, c)
    tb = p + ".transformer_blocks.0"
    n = layer_norm(sd, tb + ".norm1", h)
    h = attention(sd, tb + ".attn1", n, n, heads) + h
    n = layer_norm(sd, tb + ".norm2", h)
    h = attention(sd, tb + ".attn2", n, ehs, heads) + h
    n = layer_norm(sd, tb + ".norm3", h)
    g = linear(sd, tb + ".ff.net.0.proj", n)
    hidden, gate = g.chunk(2, dim=-1)                     # GEGLU: hidden * gelu(gate)
    f = linear(sd, tb + ".ff.net.2", hidden * F.gelu(gate))
    h = f + h
    h = h.reshape(b, hh, ww, c).permute(0, 3, 1, 2)
    h = conv(sd, p + ".proj_out", h, padding=0)
    return h + res


def encoder_forward(sd: SD, cfg, sample, emb, ehs):
    """down_blocks + mid_block shared by UNet and ControlNet (CL:205-238). Returns (skips tuple, mid sample)."""
    g, eps = cfg.norm_num_groups, cfg.norm_eps
    skips = [sample]
    ch = cfg.block_out_channels
    for i in range(len(ch)):
        for j in range(cfg.layers_per_block):
            sample = resnet(sd, f"down_blocks.{i}.resnets.{j}", sample, emb, g, eps)
            if cfg.down_has_attn[i]:
                sample = transformer(sd, f"down_blocks.{i}.attentions.{j}", sample, ehs, cfg.num_heads, g)
            skips.append(sample)
        if i != len(ch) - 1:
            sample = conv(sd, f"down_blocks.{i}.downsamplers.0.conv", sample, stride=2, padding=1)
            skips.append(sample)
    sample = resnet(sd, "mid_block.resnets.0", sample, emb, g, eps)
    sample = transformer(sd, "mid_block.attentions.0", sample, ehs, cfg.num_heads, g)
    sample = resnet(sd, "mid_block.resnets.1", sample, emb, g, eps)
    return skips, sample


# ----------------------------------------------------------------------------------------------------------------
# UNet2DConditionModel.forward with additional residuals (called at PL:500-510)
# ----------------------------------------------------------------------------------------------------------------
def unet_forward(sd: SD, cfg, sample, t, ehs, down_res: Optional[Sequence] = None, mid_res=None):
    emb = time_embedding(sd, t, cfg, sample.shape[0])
    h = conv(sd, "conv_in", sample)
    skips, h = encoder_forward(sd, cfg, h, emb, ehs)
    if down_res is not None:
        skips = [s + r for s, r in zip(skips, down_res)]
    if mid_res is not None:
        h = h + mid_res
    g, eps = cfg.norm_num_groups, cfg.norm_eps
    n = len(cfg.block_out_channels)
    for i in range(n):
        for j in range(cfg.layers_per_block + 1):
            h = torch.cat([h, skips.pop()], dim=1)
            h = resnet(sd, f"up_blocks.{i}.resnets.{j}", h, emb, g, eps)
            if cfg.up_has_attn[i]:
                h = transformer(sd, f"up_blocks.{i}.attentions.{j}", h, ehs, cfg.num_heads, g)
        if i != n - 1:
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")
            h = conv(sd, f"up_blocks.{i}.upsamplers.0.conv", h)
    h = F.silu(group_norm(sd, "conv_norm_out", h, g, eps))
    return conv(sd, "conv_out", h)


# ----------------------------------------------------------------------------------------------------------------
# ControlNet (CL:58-290)
# ----------------------------------------------------------------------------------------------------------------
def cond_embedding(sd: SD, cfg, cond):
    """diffusers ControlNetConditioningEmbedding: conv_in, (conv, conv s2) x3, conv_out; SiLU between."""
    p = "controlnet_cond_embedding"
    h = F.silu(conv(sd, p + ".conv_in", cond))
    nblk = 2 * (len(cfg.conditioning_embedding_out_channels) - 1)
    for i in range(nblk):
        h = F.silu(conv(sd, f"{p}.blocks.{i}", h, stride=2 if i % 2 == 1 else 1))
    return conv(sd, p + ".conv_out", h)


def vae_cond_embedding(sd_cn: SD, vae_sd: SD, vae_cfg, cond, noise):
    """VAEControlNetConditioningEmbedding.forward (CL:38-42).  `conv_vae_out` IS the net's conv_in (CL:36,595-598).

    `noise` replaces the global-RNG draw of latent_dist.sample() so the oracle is deterministic.
    """
    z = vae_encode_sample(vae_sd, vae_cfg, cond, noise)
    z = z * vae_cfg.scaling_factor
    return conv(sd_cn, "conv_in", z)


def controlnet_forward(sd: SD, cfg, sample, t, ehs, cond, conditioning_scale=1.0, embed_fn=None, guess_mode=False):
    """CachedControlNetModel.forward.  `cond` with latent spatial size is used as-is (CL:199-203)."""
    emb = time_embedding(sd, t, cfg, sample.shape[0])
    h = conv(sd, "conv_in", sample)
    if cond.shape[2:] != h.shape[2:]:
        cond = embed_fn(cond) if embed_fn is not None else cond_embedding(sd, cfg, cond)
    h = h + cond
    skips, h = encoder_forward(sd, cfg, h, emb, ehs)
    down = [conv(sd, f"controlnet_down_blocks.{i}", s, padding=0) for i, s in enumerate(skips)]
    mid = conv(sd, "controlnet_mid_block", h, padding=0)
    if guess_mode:                                         # CL:256-264 (global_pool_conditions is False)
        ls = torch.logspace(-1, 0, len(down) + 1) * conditioning_scale
        down = [d * s for d, s in zip(down, ls)]
        mid = mid * ls[-1]
    else:
        down = [d * conditioning_scale for d in down]      # CL:266-270
        mid = mid * conditioning_scale
    return down, mid


def fuse_lora(sd: SD) -> SD:
    """ControlLoRAModel.fuse (CL:728-777): W <- W + B.A into a PRIVATE copy (never the tied UNet tensors)."""
    out = {}
    for k, v in sd.items():
        if ".lora_layer." in k:
            continue
        out[k] = v
    for k in sd:
        if k.endswith(".lora_layer.down.weight"):
            base = k[: -len(".lora_layer.down.weight")]
            out[base + ".weight"] = sd[base + ".weight"] + sd[base + ".lora_layer.up.weight"] @ sd[k]
    return out


def tie_weights(cn_sd: SD, unet_sd: SD) -> SD:
    """ControlLoRAModel.tie_weights (CL:623-632): encoder params alias the UNet's."""
    out = dict(cn_sd)
    for k, v in unet_sd.items():
        if k.split(".")[0] in ("conv_in", "time_embedding", "down_blocks", "mid_block"):
            out[k] = v
    return out


# ----------------------------------------------------------------------------------------------------------------
# EdgeStyle fusion (MC:23-63, 116-171, 479-514)
# ----------------------------------------------------------------------------------------------------------------
def interleave_tensors(tensors: Sequence[torch.Tensor]) -> torch.Tensor:
    stacked = torch.stack(list(tensors), dim=1)                       # [B, n, C, H, W]
    b, n, c, h, w = stacked.shape
    return stacked.permute(0, 2, 1, 3, 4).contiguous().view(b, -1, h, w)   # channel = c*n + net


def controlnet_block(sd: SD, p: str, x):
    c3 = sd[p + ".first_conv.weight"].shape[0]
    c1 = sd[p + ".second_conv.weight"].shape[0]
    x = F.conv2d(x, sd[p + ".first_conv.weight"], sd[p + ".first_conv.bias"], groups=c3)
    x = F.layer_norm(x, x.shape[1:], sd[p + ".first_normalization.weight"], sd[p + ".first_normalization.bias"])
    x = F.silu(x)
    x = F.conv2d(x, sd[p + ".second_conv.weight"], sd[p + ".second_conv.bias"], groups=c1)
    x = F.layer_norm(x, x.shape[1:], sd[p + ".second_normalization.weight"], sd[p + ".second_normalization.bias"])
    x = F.silu(x)
    return F.conv2d(x, sd[p + ".third_conv.weight"], sd[p + ".third_conv.bias"], groups=c1)


def multicontrolnet_forward(fusion_sd: SD, nets: Sequence[Tuple[SD, object]], sample, t, ehs,
                            conds: Sequence[torch.Tensor], scales: Sequence[float], guess_mode=False):
    """EdgeStyleMultiControlNetModel.forward (MC:116-171). `nets` = 6 x (state_dict, cfg); conds pre-embedded."""
    downs, mids = [], []
    for (sd, cfg), cond, scale in zip(nets, conds, scales):
        d, m = controlnet_forward(sd, cfg, sample, t, ehs, cond, scale, guess_mode=guess_mode)      # MC:136-149
        downs.append(d)
        mids.append(m)
    down = [interleave_tensors(level) for level in zip(*downs)]
    mid = interleave_tensors(mids)
    down = [controlnet_block(fusion_sd, f"multi_controlnet_down_blocks.{i}", x) for i, x in enumerate(down)]
    mid = controlnet_block(fusion_sd, "multi_controlnet_mid_block", mid)
    return down, mid


# ----------------------------------------------------------------------------------------------------------------
# AutoencoderKL
# ----------------------------------------------------------------------------------------------------------------
def _vae_attn(sd: SD, p: str, x, groups, eps):
    b, c, h, w = x.shape
    res = x
    y = group_norm(sd, p + ".group_norm", x.view(b, c, h * w), groups, eps).transpose(1, 2)
    y = attention(sd, p, y, y, heads=1)
    return y.transpose(1, 2).reshape(b, c, h, w) + res


def _vae_mid(sd: SD, p: str, x, g, eps):
    x = resnet(sd, p + ".resnets.0", x, None, g, eps)
    x = _vae_attn(sd, p + ".attentions.0", x, g, eps)
    return resnet(sd, p + ".resnets.1", x, None, g, eps)


def vae_encode_moments(sd: SD, cfg, x):
    g, eps = cfg.norm_num_groups, cfg.norm_eps
    h = conv(sd, "encoder.conv_in", x)
    n = len(cfg.block_out_channels)
    for i in range(n):
        for j in range(cfg.layers_per_block):
            h = resnet(sd, f"encoder.down_blocks.{i}.resnets.{j}", h, None, g, eps)
        if i != n - 1:
            h = F.pad(h, (0, 1, 0, 1))                       # asymmetric pad, conv padding 0
            h = conv(sd, f"encoder.down_blocks.{i}.downsamplers.0.conv", h, stride=2, padding=0)
    h = _vae_mid(sd, "encoder.mid_block", h, g, eps)
    h = F.silu(group_norm(sd, "encoder.conv_norm_out", h, g, eps))
    h = conv(sd, "encoder.conv_out", h)
    return conv(sd, "quant_conv", h, padding=0)


def vae_encode_sample(sd: SD, cfg, x, noise):
    """encode(x).latent_dist.sample() with the normal draw supplied by the caller (CL:39)."""
    mean, logvar = vae_encode_moments(sd, cfg, x).chunk(2, dim=1)
    logvar = logvar.clamp(-30.0, 20.0)
    return mean + torch.exp(0.5 * logvar) * noise


def vae_cond_from_moments(sd_cn: SD, vae_cfg, moments, noise):
    """VAEControlNetConditioningEmbedding.forward (CL:38-42) from cached encoder moments: the encode is deterministic, only
    latent_dist.sample() changes from call to call - which is what the STOCK pipeline does every denoising step (see pipeline)."""
    mean, logvar = moments.chunk(2, dim=1)
    z = mean + torch.exp(0.5 * logvar.clamp(-30.0, 20.0)) * noise
    return conv(sd_cn, "conv_in", z * vae_cfg.scaling_factor)


def vae_decode(sd: SD, cfg, z):
    g, eps = cfg.norm_num_groups, cfg.norm_eps
    h = conv(sd, "post_quant_conv", z, padding=0)
    h = conv(sd, "decoder.conv_in", h)
    h = _vae_mid(sd, "decoder.mid_block", h, g, eps)
    n = len(cfg.block_out_channels)
    for i in range(n):
        for j in range(cfg.layers_per_block + 1):
            h = resnet(sd, f"decoder.up_blocks.{i}.resnets.{j}", h, None, g, eps)
        if i != n - 1:
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")
            h = conv(sd, f"decoder.up_blocks.{i}.upsamplers.0.conv", h)
    h = F.silu(group_norm(sd, "decoder.conv_norm_out", h, g, eps))
    return conv(sd, "decoder.conv_out", h)


# ----------------------------------------------------------------------------------------------------------------
# DDIM (SD1.5 scheduler_config.json semantics; PL:382-385, 520-522)
# ----------------------------------------------------------------------------------------------------------------
class DDIM:
    """scaled_linear betas 0.00085->0.012, 1000 train steps, steps_offset=1, 'leading' spacing,
    set_alpha_to_one=False, clip_sample=False, eta=0, epsilon prediction."""

    def __init__(self, num_train=1000, beta_start=0.00085, beta_end=0.012, steps_offset=1):
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train, dtype=torch.float32) ** 2
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.final_alpha_cumprod = self.alphas_cumprod[0]
        self.num_train = num_train
        self.steps_offset = steps_offset
        self.init_noise_sigma = 1.0

    def set_timesteps(self, n: int):
        self.n = n
        ratio = self.num_train // n
        self.timesteps = (torch.arange(0, n) * ratio).round().flip(0).long() + self.steps_offset
        return self.timesteps

    def step(self, eps, t: int, x):
        prev_t = t - self.num_train // self.n
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
        x0 = (x - (1 - a_t).sqrt() * eps) / a_t.sqrt()
        return a_prev.sqrt() * x0 + (1 - a_prev).sqrt() * eps


# ----------------------------------------------------------------------------------------------------------------
# the pipeline (PL:91-582), with pre-embedded conditions (the cached semantics, PL:660-662)
# ----------------------------------------------------------------------------------------------------------------
def denoise_step(unet_sd, unet_cfg, fusion_sd, nets, sample, t, ehs, conds, scales, guess_mode=False, cfg_on=True):
    """One controlnet->unet evaluation == OnnxUNetAndControlnets.forward (export_onnx.py:43-74).
    fusion_sd None = the single-ControlNet form of the pipeline (nets and conds hold one entry).
    guess_mode with CFG (PL:453-459, 487-497): the ControlNets see only the conditional half (conds are then [B]
    tensors) and the unconditional half of the UNet gets zero residuals."""
    if fusion_sd is None:
        # a plain ControlNetModel as `controlnet` (PL:338-351, PL:426, PL:464-470): one net, scalar scale, its 13
        # residuals go to the UNet as they are (no fusion blocks) — BASELINE configs[0]
        (sd, ncfg), = nets
        cond, = conds
        scale = scales[0] if isinstance(scales, (list, tuple)) else scales

        def cn(x, e):
            return controlnet_forward(sd, ncfg, x, t, e, cond, scale, guess_mode=guess_mode)
    else:
        def cn(x, e):
            return multicontrolnet_forward(fusion_sd, nets, x, t, e, conds, scales, guess_mode)
    if guess_mode and cfg_on:
        down, mid = cn(sample.chunk(2)[1], ehs.chunk(2)[1])
        down = [torch.cat([torch.zeros_like(d), d]) for d in down]
        mid = torch.cat([torch.zeros_like(mid), mid])
    else:
        down, mid = cn(sample, ehs)
    return unet_forward(unet_sd, unet_cfg, sample, t, ehs, down, mid)


def pipeline(unet_sd, unet_cfg, fusion_sd, nets, vae_sd, vae_cfg, latents, prompt_embeds, negative_prompt_embeds,
             conds, num_inference_steps=50, guidance_scale=7.5, scales=None, control_guidance_start=0.0,
             control_guidance_end=1.0, decode=True, on_step=None, guess_mode=False, cond_resample=None):
    """EdgeStyleStableDiffusionControlNetPipeline.__call__ (PL:91-582) for pre-embedded `conds`
    (6 x [B,C0,h,w]; duplicated for CFG here exactly like PL:657-658 does before embedding... the caller passes
    already-embedded tensors of batch 2B when CFG is on and guess_mode is off, B otherwise).

    cond_resample: {net index: (encoder moments [N,2L,h,w], noise table [T,N,L,h,w])} - the STOCK
    StableDiffusionControlNetPipeline the reference's test script drives (TT:263-272) hands the raw condition IMAGE to the
    ControlNets at every step, so CachedControlNetModel.forward embeds it every step (CL:199-203) and a VAE-conditioned net
    draws a fresh latent_dist.sample() each time (CL:38-42): condition k of step i is conv_in((mean + std * noise[i]) * sf)."""
    n_nets = len(nets)
    scales = list(scales) if scales is not None else [1.0] * n_nets
    cfg_on = guidance_scale > 1.0                              # do_classifier_free_guidance
    ehs = torch.cat([negative_prompt_embeds, prompt_embeds]) if cfg_on else prompt_embeds   # PL:329-330
    sched = DDIM()
    timesteps = sched.set_timesteps(num_inference_steps)
    latents = latents * sched.init_noise_sigma                  # PL:626
    starts = [control_guidance_start] * n_nets
    ends = [control_guidance_end] * n_nets
    T = len(timesteps)
    for i, t in enumerate(timesteps.tolist()):
        keep = [1.0 - float(i / T < s or (i + 1) / T > e) for s, e in zip(starts, ends)]   # PL:419-427
        x = torch.cat([latents] * 2) if cfg_on else latents     # PL:443-447
        cond_scale = [c * k for c, k in zip(scales, keep)]      # PL:464-470
        if cond_resample:
            conds = list(conds)
            for k, (mom, table) in cond_resample.items():       # CL:199-203 + CL:38-42 at every step
                conds[k] = vae_cond_from_moments(nets[k][0], vae_cfg, mom, table[i])
        eps = denoise_step(unet_sd, unet_cfg, fusion_sd, nets, x, t, ehs, conds, cond_scale, guess_mode, cfg_on)
        if cfg_on:
            e_u, e_t = eps.chunk(2)
            eps = e_u + guidance_scale * (e_t - e_u)            # PL:513-517
        latents = sched.step(eps, t, latents)                   # PL:520-522
        if on_step is not None:
            on_step(i, t, latents, eps)
    if not decode:
        return latents
    img = vae_decode(vae_sd, vae_cfg, latents / vae_cfg.scaling_factor)   # PL:552-557
    return (img / 2 + 0.5).clamp(0, 1)                          # PL:570-572 postprocess, output_type="pt"


# ----------------------------------------------------------------------------------------------------------------
# UniPCMultistepScheduler (the scheduler every reference caller swaps in: TT:273, APP:118, INF:538) — restated
# from the published algorithm of diffusers==0.26.3 `schedulers/scheduling_unipc_multistep.py` (B(h) form "bh2",
# solver_order 2, predict_x0, lower_order_final, epsilon prediction; betas/steps_offset/timestep_spacing taken from
# the SD1.5 scheduler config via from_config).  PARITY UNPINNED like the other [D] blocks.
# ----------------------------------------------------------------------------------------------------------------
class UniPC:
    def __init__(self, num_train=1000, beta_start=0.00085, beta_end=0.012, steps_offset=1, solver_order=2,
                 timestep_spacing="leading"):
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train, dtype=torch.float32) ** 2
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.num_train, self.steps_offset, self.solver_order = num_train, steps_offset, solver_order
        self.timestep_spacing = timestep_spacing
        self.init_noise_sigma = 1.0

    def set_timesteps(self, n: int):
        import numpy as np
        if self.timestep_spacing == "linspace":
            ts = np.linspace(0, self.num_train - 1, n + 1).round()[::-1][:-1].copy().astype(np.int64)
        elif self.timestep_spacing == "leading":
            ratio = self.num_train // (n + 1)
            ts = (np.arange(0, n + 1) * ratio).round()[::-1][:-1].copy().astype(np.int64) + self.steps_offset
        else:
            raise ValueError(self.timestep_spacing)
        ac = self.alphas_cumprod.double().numpy()
        sig = ((1 - ac) / ac) ** 0.5
        sigmas = np.interp(ts, np.arange(0, len(sig)), sig)
        sigmas = np.concatenate([sigmas, [((1 - ac[0]) / ac[0]) ** 0.5]])
        self.sigmas = torch.from_numpy(sigmas).double()
        self.timesteps = torch.from_numpy(ts)
        self.model_outputs = [None] * self.solver_order
        self.lower_order_nums, self.last_sample, self.step_index, self.this_order = 0, None, 0, 1
        return self.timesteps

    @staticmethod
    def _alpha_sigma(sigma):
        alpha = 1.0 / (sigma ** 2 + 1.0) ** 0.5
        return alpha, sigma * alpha

    def _rb(self, rks, hh, order):
        h_phi_1 = torch.expm1(hh)
        h_phi_k = h_phi_1 / hh - 1
        B_h = torch.expm1(hh)                                   # bh2
        fact, R, b = 1, [], []
        for i in range(1, order + 1):
            R.append(torch.pow(rks, i - 1))
            b.append(h_phi_k * fact / B_h)
            fact *= i + 1
            h_phi_k = h_phi_k / hh - 1 / fact
        return torch.stack(R), torch.stack(b), h_phi_1, B_h

    def _lams(self, idx):
        a, s = self._alpha_sigma(self.sigmas[idx])
        return a, s, torch.log(a) - torch.log(s)

    def _uni_p(self, sample, order):
        m0, x = self.model_outputs[-1], sample
        a_t, s_t, l_t = self._lams(self.step_index + 1)
        a_0, s_0, l_0 = self._lams(self.step_index)
        h = l_t - l_0
        rks, D1s = [], []
        for i in range(1, order):
            _, _, l_i = self._lams(self.step_index - i)
            rk = (l_i - l_0) / h
            rks.append(rk)
            D1s.append((self.model_outputs[-(i + 1)] - m0) / rk)
        rks.append(torch.tensor(1.0, dtype=torch.float64))
        R, b, h_phi_1, B_h = self._rb(torch.stack(rks), -h, order)
        x_t = (s_t / s_0) * x - a_t * h_phi_1 * m0
        if D1s:
            rhos_p = torch.tensor([0.5], dtype=torch.float64) if order == 2 else torch.linalg.solve(R[:-1, :-1], b[:-1])
            x_t = x_t - a_t * B_h * sum(r * d for r, d in zip(rhos_p, D1s))
        return x_t

    def _uni_c(self, model_t, last_sample, order):
        m0, x = self.model_outputs[-1], last_sample
        a_t, s_t, l_t = self._lams(self.step_index)
        a_0, s_0, l_0 = self._lams(self.step_index - 1)
        h = l_t - l_0
        rks, D1s = [], []
        for i in range(1, order):
            _, _, l_i = self._lams(self.step_index - (i + 1))
            rk = (l_i - l_0) / h
            rks.append(rk)
            D1s.append((self.model_outputs[-(i + 1)] - m0) / rk)
        rks.append(torch.tensor(1.0, dtype=torch.float64))
        R, b, h_phi_1, B_h = self._rb(torch.stack(rks), -h, order)
        rhos_c = torch.tensor([0.5], dtype=torch.float64) if order == 1 else torch.linalg.solve(R, b)
        corr = sum(r * d for r, d in zip(rhos_c[:-1], D1s)) if D1s else 0.0
        return (s_t / s_0) * x - a_t * h_phi_1 * m0 - a_t * B_h * (corr + rhos_c[-1] * (model_t - m0))

    def step(self, eps, t: int, sample):
        sample = sample.double()
        a_t, s_t = self._alpha_sigma(self.sigmas[self.step_index])
        x0 = (sample - s_t * eps.double()) / a_t                                # convert_model_output (epsilon, predict_x0)
        if self.step_index > 0 and self.last_sample is not None:
            sample = self._uni_c(x0, self.last_sample, self.this_order)
        self.model_outputs = self.model_outputs[1:] + [x0]
        this_order = min(self.solver_order, len(self.timesteps) - self.step_index)   # lower_order_final
        self.this_order = min(this_order, self.lower_order_nums + 1)
        self.last_sample = sample
        prev = self._uni_p(sample, self.this_order)
        if self.lower_order_nums < self.solver_order:
            self.lower_order_nums += 1
        self.step_index += 1
        return prev.float()
