"""StableDiffusionControlNetPipeline / EdgeStyleStableDiffusionControlNetPipeline call surface over the HIP path.

Follows model/edgestyle_pipeline.py:91-664 step for step (citations inline).  What changes is *how* it runs:

* the six conditioning images are embedded once per call (the cached semantics, PL:660-662 / CL:289-290);
  the stock pipeline the reference's test script uses would re-run a VAE encode with a fresh global-RNG sample
  every step (CL:38-42, CL:199-203) — non-deterministic and 6.7 TFLOP/step of redundant work — so both pipeline
  names here use the cached semantics and the VAE sample noise comes from `generator` (or `cond_noise=`);
* one denoising step (6 ControlNet passes as 3 batched passes -> 13 fusion blocks -> UNet -> CFG -> DDIM) is
  captured once into a hipGraph and replayed; timestep, DDIM coefficients and conditioning scales are device tables
  indexed by a device-side step counter, so a replay needs no host work;
* latents stay fp32 NHWC on the device for the whole loop; VAE decode and the [0,1] post-process are kernels.
"""
from dataclasses import dataclass
from types import SimpleNamespace
from typing import Any, Callable, Dict, List, Optional, Sequence, Union

import os

import numpy as np
import torch

from . import ops
from .lib import EdgeStyleHipError
from .models import (AutoencoderKL, ControlNetModel, EdgeStyleMultiControlNetModel, StepRunner, StepState,
                     UNet2DConditionModel, _as_nhwc, _as_nchw_view)
from .schedulers import DDIMScheduler, UniPCMultistepScheduler


from collections import OrderedDict

# id(device generator) -> (generator, state we left it in, its CPU stand-in).  torch.Generator cannot be weakly referenced,
# so the cache is a small LRU instead of a global that pins every generator ever seen.
_HOST_GENS: "OrderedDict[int, tuple]" = OrderedDict()
_HOST_GENS_MAX = 8


def _host_generator(generator):
    """torch.Generator on a device -> a CPU generator seeded with its initial_seed() (one per device generator object,
    so successive draws advance like the original would); lists are mapped element-wise; CPU generators pass through.

    A caller who calls `gen.manual_seed(s)` again between two pipeline calls (the usual way to reproduce a run; the
    reference draws from the device generator itself, TT:274, so that reproduces it there) must get the same images here:
    every use advances the DEVICE generator by one draw and remembers the state it was left in; a generator found in any
    other state has been re-seeded (even to the same seed) and its stand-in restarts from initial_seed()."""
    if generator is None:
        return None
    if isinstance(generator, (list, tuple)):
        return [_host_generator(g) for g in generator]
    if generator.device.type == "cpu":
        return generator
    key = id(generator)
    hit = _HOST_GENS.pop(key, None)
    if hit is None or hit[0] is not generator or not torch.equal(hit[1], generator.get_state()):
        host = torch.Generator().manual_seed(generator.initial_seed())
    else:
        host = hit[2]
    torch.empty(1, device=generator.device).normal_(generator=generator)      # mark: "seen in this state"
    _HOST_GENS[key] = (generator, generator.get_state().clone(), host)
    while len(_HOST_GENS) > _HOST_GENS_MAX:
        _HOST_GENS.popitem(last=False)
    return host


@dataclass
class StableDiffusionPipelineOutput:
    images: Any
    nsfw_content_detected: Optional[List[bool]] = None


LOOP_GRAPH = os.environ.get("ES_LOOP_GRAPH", "1") == "1"     # all steps of a call as one hipGraph (from the second call on)
# steps outside every net's control-guidance window (PL:419-427) run the UNet alone (+ the constant fusion-of-zeros residuals)
WINDOW_SKIP = os.environ.get("ES_WINDOW_SKIP", "1") == "1"


class _Loop:
    """Static device buffers + the captured step graph for one (B, cfg, h, w, guess) shape.  Everything the graph reads
    between calls is owned here (the StepState included) and refilled in place; `sig` records the addresses the graph
    was captured with, and a call whose buffers differ re-captures (never replays stale pointers)."""

    def __init__(self, pipe, B: int, cfg_on: bool, h: int, w: int, guess: bool = False):
        dev, dt = pipe.device, pipe.dtype
        unet = pipe.unet
        self.B, self.cfg_on, self.h, self.w = B, cfg_on, h, w
        self.guess = guess               # guess_mode: log-spaced level scales; under CFG ControlNets see the cond half only
        self.N = N = 2 * B if cfg_on else B
        Lc, Lp = unet.cfg.in_channels, unet.engine.in_pad
        c0 = unet.cfg.block_out_channels[0]
        self.runner = pipe._runner
        self.state = StepState()
        nn = self.nn = self.runner.n_nets
        k = self.runner.kmax
        self.latents = torch.zeros((B, h, w, Lc), dtype=torch.float32, device=dev)
        self.model_in = torch.zeros((N, h, w, Lp), dtype=dt, device=dev)
        self.noise = torch.zeros((N, h, w, unet.cfg.out_channels), dtype=dt, device=dev)
        self.conds = [torch.zeros((B if (guess and cfg_on) else N, h, w, c0), dtype=dt, device=dev) for _ in range(nn)]
        self.ehs = torch.zeros((N, 77, unet.cfg.cross_attention_dim), dtype=dt, device=dev)
        self.step_idx = torch.zeros((1,), dtype=torch.int32, device=dev)
        self.t_rows = torch.zeros((k * N,), dtype=torch.float32, device=dev)
        self.scales_cur = torch.ones((nn,), dtype=torch.float32, device=dev)
        self.hist = [torch.zeros((B, h, w, Lc), dtype=torch.float32, device=dev) for _ in range(3)]   # UniPC state
        self.unipc = False
        self.t_table = None
        self.scale_table = None
        self.coef = None
        self.graph = None
        self.prep_graph = None           # text K/V projections + condition slots + time table, captured like the step
        self.decode_graph = None         # VAE decode + [0,1] post-process
        self.image = None                # the decode graph's output buffer
        self.ts_dev = None               # fp32 [T] timesteps the time table is built from
        self.sig = None
        self.captures = 0                # how often this loop's graph was (re)captured (tests)
        self.loop_graph = None           # all steps of a call as one graph (captured with the step graph)
        self.guidance_scale = None
        self.steps = None
        self.resample = {}               # net index -> dict(net, mom, table, cur): VAE conditions re-sampled at every step
        self.skip = None                 # per step: no ControlNet contributes (every controlnet_keep is 0, PL:419-427)
        self.graph_unet = None           # the captured UNet-only step of those steps

    def signature(self):
        return self.state.signature() + tuple(t.data_ptr() for t in (self.t_table, self.scale_table, self.coef, self.ts_dev)) + \
            (ops.LANE,)

    def set_model_in(self):
        """self.latents (fp32 NHWC) -> the networks' input: both CFG halves, compute dtype, channel-padded (PL:443-447)."""
        ops.latents_to_input(self.latents, self.model_in, self.cfg_on)

    def resample_conds(self):
        """The stock pipeline's per-step condition embedding (CL:199-203 -> CL:38-42) for the VAE-conditioned nets, from the
        cached encoder moments: row `step` of the net's noise table -> (mean + std * noise) * scaling_factor -> conv_in."""
        for i, r in self.resample.items():
            ops.gather_row(r["table"], self.step_idx, r["cur"].view(-1))
            vae = r["net"]._vae
            z = ops.vae_sample(r["mom"], r["cur"], vae.cfg.latent_channels, r["net"].engine.in_pad, vae.cfg.scaling_factor)
            r["net"].engine.embed_latent(z, out=self.conds[i])
            self.runner.refresh_cond(i, self.conds[i])

    def one_step(self):
        """Everything between PL:435 and PL:522 for the step selected by the device counter."""
        if self.resample:
            self.runner.state = self.state
            self.resample_conds()
        ops.gather_row(self.t_table, self.step_idx, self.t_rows)
        ops.gather_row(self.scale_table, self.step_idx, self.scales_cur)
        self.runner.state = self.state
        self.runner.step(self.model_in, self.t_rows, self.conds, [1.0] * self.nn, self.scales_cur, out=self.noise,
                         step_idx=self.step_idx, guess_mode=self.guess)
        self._scheduler_step()

    def one_step_unet(self):
        """A step outside every net's control-guidance window (PL:419-427 sets controlnet_keep = 0): the six encoder passes and
        the fusion launches are skipped, not multiplied by zero - the UNet plus the constant fusion-of-zeros residuals."""
        ops.gather_row(self.t_table, self.step_idx, self.t_rows)
        self.runner.state = self.state
        self.runner.step_unet_only(self.model_in, self.t_rows, out=self.noise, step_idx=self.step_idx)
        self._scheduler_step()

    def step_of(self, i: int):
        return self.one_step_unet if (self.skip is not None and self.skip[i]) else self.one_step

    def rehearse(self, fn):
        """Run a step function eagerly for its side effects on scratch sizes / lazy initialisation only: the loop's state
        (latents, network input, step counter, scheduler history) is put back afterwards."""
        saved = [t.clone() for t in (self.latents, self.model_in, self.step_idx, *self.hist)]
        fn()
        for dst, src in zip((self.latents, self.model_in, self.step_idx, *self.hist), saved):
            dst.copy_(src)

    def _scheduler_step(self):
        if self.unipc:
            ops.cfg_unipc_step(self.noise, self.latents, self.hist[0], self.hist[1], self.hist[2], self.model_in,
                               self.coef, self.step_idx, float(self.guidance_scale), self.cfg_on)
        else:
            ops.cfg_ddim_step(self.noise, self.latents, self.model_in, self.coef, self.step_idx,
                              float(self.guidance_scale), self.cfg_on)
        ops.incr(self.step_idx)


class StableDiffusionControlNetPipeline:
    """Keeps the constructor / from_pretrained / __call__ surface TT:263-275 and TT:326-359 use."""

    # The STOCK pipeline hands the raw condition images to the ControlNets at every step: CachedControlNetModel.forward embeds
    # them each time (CL:199-203) and a VAE-conditioned net draws a FRESH latent_dist.sample() per step (CL:38-42) - what the
    # reference's test script actually runs (TT:263-272).  The encoder passes are deterministic, so this class caches their
    # moments and redoes only `(mean + std * noise_t) * sf -> conv_in` per step (a [N,4,h,w] sample and a 4 -> C0 convolution
    # per net instead of the reference's three VAE encodes per step).  EdgeStyleStableDiffusionControlNetPipeline embeds once
    # (PL:629-664) and keeps False.  `resample_cond_each_step=` of __call__ overrides either way.
    resample_cond_each_step = True

    def __init__(self, vae: AutoencoderKL, text_encoder=None, tokenizer=None, unet: UNet2DConditionModel = None,
                 controlnet: Union[EdgeStyleMultiControlNetModel, ControlNetModel] = None, scheduler=None,
                 safety_checker=None, feature_extractor=None, image_encoder=None, requires_safety_checker: bool = False):
        if unet is None or controlnet is None or vae is None:
            raise ValueError("vae, unet and controlnet are required")
        if isinstance(controlnet, (list, tuple)):
            # the stock pipeline wraps a list into diffusers' MultiControlNetModel (plain sum of residuals); the reference
            # never takes that path (TT:252-258 builds the fused EdgeStyleMultiControlNetModel), so it is not built here
            raise ValueError("pass an EdgeStyleMultiControlNetModel (the reference's fused 6-net model) or one ControlNetModel")
        self.vae, self.text_encoder, self.tokenizer = vae, text_encoder, tokenizer
        self.unet, self.controlnet = unet, controlnet
        self.scheduler = scheduler or DDIMScheduler()
        self.safety_checker = None
        self.vae_scale_factor = vae.cfg.scale
        self.device = unet.device
        self.dtype = unet.dtype
        self.use_graph = True
        self._runner = None
        self._loops: Dict[Any, _Loop] = {}
        self._last_loop: Optional[_Loop] = None
        self.collect_timing = False      # tools: HIP events at the phase boundaries of a call -> self.timing (ms)
        self.timing: Dict[str, float] = {}

    @classmethod
    def from_pretrained(cls, path=None, **components):
        components.pop("torch_dtype", None)
        if "scheduler" not in components:
            components["scheduler"] = DDIMScheduler()
        return cls(**{k: v for k, v in components.items() if k in (
            "vae", "text_encoder", "tokenizer", "unet", "controlnet", "scheduler", "safety_checker",
            "feature_extractor", "image_encoder", "requires_safety_checker")})

    def to(self, device):
        self.device = torch.device(device)
        for m in (self.unet, self.controlnet, self.vae):
            m.to(self.device)
        if self.text_encoder is not None and hasattr(self.text_encoder, "to"):
            self.text_encoder.to("cpu")          # CLIP text encode is outside the hot path
        self._runner = None
        self._loops.clear()
        self._last_loop = None
        return self

    @property
    def _execution_device(self):
        return self.device

    # ------------------------------------------------------------------------------------------------------
    def encode_prompt(self, prompt, negative_prompt, do_cfg, prompt_embeds=None, negative_prompt_embeds=None):
        """PL:315-325.  With tensors given this is a pass-through; strings need the transformers CLIP encoder."""
        if prompt_embeds is None:
            if self.tokenizer is None or self.text_encoder is None:
                raise ValueError("pass prompt_embeds/negative_prompt_embeds, or build the pipeline with a "
                                 "tokenizer and text_encoder")
            prompt_embeds = self._clip(prompt)
        if do_cfg and negative_prompt_embeds is None:
            if self.tokenizer is None or self.text_encoder is None:
                raise ValueError("negative_prompt_embeds required for guidance_scale > 1")
            n = prompt_embeds.shape[0]
            neg = negative_prompt if negative_prompt is not None else ""
            negative_prompt_embeds = self._clip([neg] * n if isinstance(neg, str) else neg)
        return prompt_embeds, negative_prompt_embeds

    def _clip(self, text):
        text = [text] if isinstance(text, str) else list(text)
        tok = self.tokenizer(text, padding="max_length", max_length=self.tokenizer.model_max_length,
                             truncation=True, return_tensors="pt")
        with torch.no_grad():
            return self.text_encoder(tok.input_ids)[0].float()

    @property
    def _single(self) -> bool:
        """A plain ControlNetModel as `controlnet` (PL:338-351) instead of the fused multi-net model."""
        return isinstance(self.controlnet, ControlNetModel)

    @property
    def _nets(self):
        return [self.controlnet] if self._single else self.controlnet.nets

    def check_inputs(self, image, prompt, prompt_embeds, controlnet_conditioning_scale, starts, ends):
        n = len(self._nets)
        if prompt is None and prompt_embeds is None:
            raise ValueError("Provide either `prompt` or `prompt_embeds`.")
        if prompt is not None and prompt_embeds is not None:
            raise ValueError("Cannot forward both `prompt` and `prompt_embeds`.")
        if self._single:
            if isinstance(image, (list, tuple)) and not (len(image) == 1 or not torch.is_tensor(image[0])):
                raise ValueError("a single ControlNet takes one conditioning image")
        elif not isinstance(image, (list, tuple)) or len(image) != n:
            raise ValueError(f"For multiple controlnets: `image` must be a list of {n} conditioning images")
        if isinstance(controlnet_conditioning_scale, (list, tuple)) and len(controlnet_conditioning_scale) != n:
            raise ValueError("`controlnet_conditioning_scale` must have one entry per ControlNet")
        if len(starts) != n or len(ends) != n:
            raise ValueError("`control_guidance_start`/`end` must have one entry per ControlNet")
        for s, e in zip(starts, ends):
            if s >= e:
                raise ValueError(f"control guidance start: {s} cannot be larger or equal to control guidance end: {e}.")
            if s < 0.0 or e > 1.0:
                raise ValueError("control guidance start/end must lie in [0, 1]")

    def prepare_image(self, image, batch_size, do_cfg, net, noise=None, generator=None, num_images_per_prompt: int = 1):
        """PL:629-664: -> fp32 [b,3,H,W] -> repeat -> CFG duplicate -> one-time embedding -> NHWC [N,h,w,C0]."""
        if not torch.is_tensor(image):
            arr = np.asarray(image, dtype=np.float32)
            if arr.ndim == 3:
                arr = arr[None]
            if arr.max() > 1.5:
                arr = arr / 255.0                       # VaeImageProcessor.preprocess(do_normalize=False)
            image = torch.from_numpy(arr).permute(0, 3, 1, 2)
        image = image.to(torch.float32)
        if image.shape[0] == 1:
            image = image.repeat_interleave(batch_size, dim=0)          # PL:647-653
        else:                                                           # image batch == prompt batch: one per prompt
            image = image.repeat_interleave(num_images_per_prompt, dim=0)
            if image.shape[0] != batch_size:
                raise ValueError("condition image batch must be 1 or equal to the prompt batch")
        if do_cfg:
            image = torch.cat([image] * 2)                              # PL:657-658
        if tuple(image.shape[1:2]) == (3,):
            emb = net.preprocess_image(image.to(self.device), noise=noise, generator=_host_generator(generator))
        else:
            emb = image                                                 # already embedded [N,C0,h,w]
        return _as_nhwc(emb, self.dtype, self.device)

    def prepare_images(self, images, batch_size, do_cfg, cond_noise=None, generator=None, num_images_per_prompt: int = 1):
        """The six one-time condition embeddings (PL:352-377, 629-664) with the redundant work removed: the reference
        embeds the CFG-duplicated batch net by net, so every deterministic encoder runs twice on identical pixels
        (both CFG halves) and the nets that share an encoder (3 x VAE, 3 x openpose stack) run it separately.  Here
        each shared encoder runs ONCE on the un-duplicated images of all its nets; only the VAE sampling noise is
        per CFG half, drawn in net order for the duplicated batch exactly as before.  Falls back to per-net
        prepare_image for anything that is not an RGB image tensor handled by the known net classes."""
        from .models import ControlLoRAModel
        nets = self._nets
        generator = _host_generator(generator)
        rgb = []
        for img in images:
            if not torch.is_tensor(img):
                rgb = None
                break
            rgb.append(img.dim() == 4 and img.shape[1] == 3 and img.shape[0] in (1, batch_size))
        if rgb is None or not all(rgb):
            return [self.prepare_image(img, batch_size, do_cfg, net, noise=None if cond_noise is None else cond_noise[i],
                                       generator=generator, num_images_per_prompt=num_images_per_prompt)
                    for i, (img, net) in enumerate(zip(images, nets))]
        rep = 2 if do_cfg else 1
        N = batch_size * rep
        self._cond_moments = {}                      # net index -> encoder moments [N,h,w,2L] of its image (per-step re-sampling)
        imgs = [img.to(torch.float32).repeat_interleave(batch_size, dim=0) if img.shape[0] == 1 else img.to(torch.float32)
                for img in images]
        out = [None] * len(nets)
        groups = {}                                  # shared encoder -> net indices
        for i, net in enumerate(nets):
            if isinstance(net, ControlLoRAModel) and net.config.uses_vae:
                if net._vae is None:
                    raise ValueError("vae must be provided if any of the controlnets uses a vae")     # MC:388-391
                groups.setdefault(("vae", id(net._vae)), []).append(i)
            else:
                groups.setdefault(("stack", id(net)), []).append(i)
        noise = {}
        for i, net in enumerate(nets):               # sampling noise in net order, for the CFG-duplicated batch (CL:39)
            if isinstance(net, ControlLoRAModel) and net.config.uses_vae:
                vae = net._vae
                hh, ww = imgs[i].shape[2] // vae.cfg.scale, imgs[i].shape[3] // vae.cfg.scale
                nz = None if cond_noise is None else cond_noise[i]
                if nz is not None and nz.dim() == 5:
                    nz = nz[0]                               # a per-step table: the embedding made here is step 0's
                if nz is None:
                    nz = torch.randn((N, vae.cfg.latent_channels, hh, ww), generator=generator, dtype=torch.float32)
                noise[i] = nz
        for (kind, _), idx in groups.items():
            batch = torch.cat([imgs[i] for i in idx]).to(self.device)
            if kind == "vae":
                vae = nets[idx[0]]._vae
                if vae.device != self.device:
                    vae.to(self.device)
                dist = vae.encode(batch).latent_dist                    # one encoder pass for all its nets, no CFG copy
                for k, i in enumerate(idx):
                    net = nets[i]
                    mom = dist.moments[k * batch_size:(k + 1) * batch_size]
                    mom = torch.cat([mom] * rep) if rep > 1 else mom
                    self._cond_moments[i] = mom.contiguous()
                    z = ops.vae_sample(mom.contiguous(), noise[i].to(self.device), vae.cfg.latent_channels,
                                       net.engine.in_pad, vae.cfg.scaling_factor)
                    out[i] = net.engine.embed_latent(z)
            else:
                net = nets[idx[0]]
                emb = _as_nhwc(net.preprocess_image(batch), self.dtype, self.device)
                for k, i in enumerate(idx):
                    e = emb[k * batch_size:(k + 1) * batch_size]
                    out[i] = torch.cat([e] * rep) if rep > 1 else e
        return out                                   # NHWC [N,h,w,C0], compute dtype, on device

    def prepare_latents(self, batch_size, channels, h, w, generator, latents=None):
        """PL:585-627.  Latents are always drawn on the HOST RNG stream (results then do not depend on the device's
        Philox stream, and the CPU oracle can reproduce them): a device generator, as TT:274 passes
        (`torch.Generator(device).manual_seed(42)`), is re-seated as a CPU generator with the same initial seed."""
        generator = _host_generator(generator)
        shape = (batch_size, channels, h, w)
        if isinstance(generator, list) and len(generator) != batch_size:
            raise ValueError(f"You have passed a list of generators of length {len(generator)}, but requested an "
                             f"effective batch size of {batch_size}.")
        if latents is None:
            if isinstance(generator, list):
                latents = torch.cat([torch.randn((1,) + shape[1:], generator=g, dtype=torch.float32)
                                     for g in generator])
            else:
                latents = torch.randn(shape, generator=generator, dtype=torch.float32)
        elif tuple(latents.shape) != shape:
            raise ValueError(f"latents shape {tuple(latents.shape)} != {shape}")
        return latents.to(torch.float32) * self.scheduler.init_noise_sigma

    # ------------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def __call__(self, prompt: Union[str, List[str], None] = None, image=None, height: Optional[int] = None,
                 width: Optional[int] = None, num_inference_steps: int = 50, timesteps=None,
                 guidance_scale: float = 7.5, negative_prompt=None, num_images_per_prompt: int = 1,
                 eta: float = 0.0, generator=None, latents: Optional[torch.Tensor] = None,
                 prompt_embeds: Optional[torch.Tensor] = None,
                 negative_prompt_embeds: Optional[torch.Tensor] = None, output_type: str = "pil",
                 return_dict: bool = True, controlnet_conditioning_scale: Union[float, List[float]] = 1.0,
                 guess_mode: bool = False, control_guidance_start: Union[float, List[float]] = 0.0,
                 control_guidance_end: Union[float, List[float]] = 1.0,
                 callback_on_step_end: Optional[Callable] = None, cond_noise: Optional[Sequence] = None,
                 resample_cond_each_step: Optional[bool] = None, ip_adapter_image=None, cross_attention_kwargs=None, clip_skip: Optional[int] = None,
                 callback_on_step_end_tensor_inputs: Sequence[str] = ("latents",), **kwargs):
        # the reference's signature (PL:92-120) in full; what this path does not implement is refused, never ignored
        if kwargs:
            raise TypeError(f"unexpected keyword argument(s) {sorted(kwargs)}: this pipeline implements PL:92-120 without the "
                            "deprecated `callback` / `callback_steps` and without extensions")
        if ip_adapter_image is not None:
            raise NotImplementedError("ip_adapter_image: IP-Adapter conditioning is not part of this path (no caller of the reference uses it)")
        if cross_attention_kwargs is not None:
            raise NotImplementedError("cross_attention_kwargs: attention processors / LoRA scale are not part of this path "
                                      "(LoRA deltas are folded into private weight copies at load time, CL:728-777)")
        if clip_skip is not None:
            raise NotImplementedError("clip_skip is not implemented: pass prompt_embeds of the layer you want (PL:106-107)")
        if list(callback_on_step_end_tensor_inputs) != ["latents"]:
            raise NotImplementedError("callback_on_step_end receives `latents` only (the step keeps no other tensor alive)")
        if eta != 0.0:
            raise NotImplementedError("only the deterministic DDIM update (eta = 0) is implemented")
        if timesteps is not None:
            raise NotImplementedError("custom timesteps are not supported (the reference path itself is broken: PL:697)")
        if self.device.type != "cuda":
            raise EdgeStyleHipError("the pipeline runs only on an MI355X: call .to('cuda') first")
        if not isinstance(self.scheduler, (DDIMScheduler, UniPCMultistepScheduler)):
            raise EdgeStyleHipError("the fused step kernels implement DDIM (the BASELINE metric) and UniPC (TT:273); "
                                    "assign edgestyle_amd.schedulers.DDIMScheduler or UniPCMultistepScheduler")
        nn = len(self._nets)
        # PL:243-265 broadcast guidance windows
        if not isinstance(control_guidance_start, list):
            control_guidance_start = [control_guidance_start] * nn
        if not isinstance(control_guidance_end, list):
            control_guidance_end = [control_guidance_end] * nn
        self.check_inputs(image, prompt, prompt_embeds, controlnet_conditioning_scale, control_guidance_start,
                          control_guidance_end)
        if self._single and not isinstance(image, (list, tuple)):
            image = [image]
        if isinstance(controlnet_conditioning_scale, (int, float)):                 # PL:295-300
            controlnet_conditioning_scale = [float(controlnet_conditioning_scale)] * nn
        do_cfg = guidance_scale > 1.0
        prompt_embeds, negative_prompt_embeds = self.encode_prompt(prompt, negative_prompt, do_cfg, prompt_embeds,
                                                                   negative_prompt_embeds)
        if num_images_per_prompt != 1:
            prompt_embeds = prompt_embeds.repeat_interleave(num_images_per_prompt, dim=0)
            if negative_prompt_embeds is not None:
                negative_prompt_embeds = negative_prompt_embeds.repeat_interleave(num_images_per_prompt, dim=0)
        B = prompt_embeds.shape[0]
        ehs = torch.cat([negative_prompt_embeds, prompt_embeds]) if do_cfg else prompt_embeds     # PL:329-330

        if self._runner is None:
            self._runner = StepRunner(self.unet, self.controlnet)

        ev = []

        def mark(name):
            if self.collect_timing:
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                ev.append((name, e))
        mark("start")
        # PL:352-377 — condition images, embedded ONCE
        guess = bool(guess_mode)
        # per-step re-sampling of the VAE conditions (the stock pipeline's semantics, class attribute above): which nets, and
        # their noise tables [T, N, L, h, w] - the caller's (cond_noise[i] 5-D) or drawn per step in net order from `generator`.
        # A 4-D cond_noise[i] pins ONE sample: that net keeps the embed-once form.
        want_rs = self.resample_cond_each_step if resample_cond_each_step is None else bool(resample_cond_each_step)
        rs_tables = {}
        if want_rs and not guess and isinstance(image, (list, tuple)):
            from .models import ControlLoRAModel
            Nn = B * (2 if do_cfg else 1)
            gen_h = _host_generator(generator)
            cand = [i for i, (img, net) in enumerate(zip(image, self._nets))
                    if isinstance(net, ControlLoRAModel) and net.config.uses_vae and torch.is_tensor(img) and img.dim() == 4
                    and img.shape[1] == 3 and (cond_noise is None or cond_noise[i] is None or cond_noise[i].dim() == 5)]
            if cand and all(torch.is_tensor(img) and img.dim() == 4 and img.shape[1] == 3 for img in image):
                sc = self.vae.cfg.scale
                shapes = {i: (Nn, self._nets[i]._vae.cfg.latent_channels, image[i].shape[2] // sc, image[i].shape[3] // sc) for i in cand}
                given = {i: cond_noise[i] for i in cand if cond_noise is not None and cond_noise[i] is not None}
                for i, tb in given.items():
                    if tuple(tb.shape) != (num_inference_steps,) + shapes[i]:
                        raise ValueError(f"cond_noise[{i}]: a per-step table must be [num_inference_steps, N, L, h, w] = "
                                         f"{(num_inference_steps,) + shapes[i]}, got {tuple(tb.shape)}")
                drawn = {i: [] for i in cand if i not in given}
                for _ in range(num_inference_steps if drawn else 0):          # the reference's draw order: step by step, net by net
                    for i in drawn:
                        drawn[i].append(torch.randn(shapes[i], generator=gen_h, dtype=torch.float32))
                rs_tables = {i: (given[i].float() if i in given else torch.stack(drawn[i])) for i in cand}
                cond_noise = [rs_tables.get(i, None if cond_noise is None else cond_noise[i]) for i in range(len(image))]
        conds = self.prepare_images(image, B, do_cfg and not guess, cond_noise, generator,
                                    num_images_per_prompt)                                      # PL:352-377, 657-658
        h, w = conds[0].shape[1:3]
        # PL:377: the reference takes height / width from the (embedded) condition images and never reads the arguments;
        # here they are honoured as a check: a request for another size than the condition images define is refused
        scale = self.vae.cfg.scale
        for name, asked, have in (("height", height, h * scale), ("width", width, w * scale)):
            if asked is not None and int(asked) != have:
                raise ValueError(f"`{name}`={asked} but the condition images define {have} pixels (latent {have // scale}): "
                                 "the output size follows the condition images (PL:377); resize them instead")

        mark("cond_embed")
        # PL:382-398
        ts = self.scheduler.set_timesteps(num_inference_steps)
        T = len(ts)
        lat = self.prepare_latents(B, self.unet.cfg.in_channels, h, w, generator, latents)

        key = (B, do_cfg, h, w, guess)
        loop = self._loops.get(key)
        if loop is None:
            loop = self._loops[key] = _Loop(self, B, do_cfg, h, w, guess)
        self._last_loop = loop
        N, k = loop.N, self._runner.kmax
        rs_key = tuple(sorted(rs_tables))
        rs_changed = tuple(sorted(loop.resample)) != rs_key or any(loop.resample[i]["table"].shape[0] != T for i in rs_key)
        if rs_changed:
            loop.resample = {i: dict(net=self._nets[i], mom=torch.empty_like(self._cond_moments[i]),
                                     table=torch.empty((T, rs_tables[i][0].numel()), dtype=torch.float32, device=self.device),
                                     cur=torch.empty(tuple(rs_tables[i].shape[1:]), dtype=torch.float32, device=self.device))
                             for i in rs_key}
        for i in rs_key:
            loop.resample[i]["mom"].copy_(self._cond_moments[i])
            loop.resample[i]["table"].copy_(rs_tables[i].reshape(T, -1))
        # PL:419-427 controlnet_keep folded into a per-step scale table
        keep = [[1.0 - float(i / T < s or (i + 1) / T > e)
                 for s, e in zip(control_guidance_start, control_guidance_end)] for i in range(T)]
        scale_table = torch.tensor([[c * kk for c, kk in zip(controlnet_conditioning_scale, row)] for row in keep],
                                   dtype=torch.float32)
        # steps in which NO net contributes (all keep = 0) skip the six encoder passes and the fusion launches altogether
        # (ES_WINDOW_SKIP=0: run them with scale 0 like the reference does; guess_mode keeps that form)
        skip = tuple(WINDOW_SKIP and not guess and all(k == 0.0 for k in row) for row in keep)
        dev = self.device
        unipc = isinstance(self.scheduler, UniPCMultistepScheduler)
        regraph = (loop.steps != T) or (loop.guidance_scale != float(guidance_scale)) or (loop.graph is None and loop.graph_unet is None) \
            or loop.unipc != unipc or loop.skip != skip or rs_changed
        loop.skip = skip
        loop.steps, loop.guidance_scale, loop.unipc = T, float(guidance_scale), unipc
        cw = 12 if unipc else 4
        if loop.t_table is None or loop.t_table.shape[0] != T or loop.coef.shape[1] != cw:
            loop.t_table = torch.empty((T, k * N), dtype=torch.float32, device=dev)
            loop.scale_table = torch.empty((T, nn), dtype=torch.float32, device=dev)
            loop.coef = torch.empty((T, cw), dtype=torch.float32, device=dev)
            loop.ts_dev = torch.empty((T,), dtype=torch.float32, device=dev)
        for hbuf in loop.hist:
            hbuf.zero_()
        loop.t_table.copy_(ts.float()[:, None].expand(T, k * N))
        loop.scale_table.copy_(scale_table)
        loop.coef.copy_(self.scheduler.coef_table())
        loop.step_idx.zero_()
        loop.latents.copy_(lat.permute(0, 2, 3, 1))
        loop.set_model_in()                                                       # PL:443-447
        for dst, src in zip(loop.conds, conds):
            dst.copy_(src)
        loop.ehs.copy_(ehs.to(dev, self.dtype))
        # every buffer the step reads between calls belongs to THIS loop (StepState): filled in place, and the graph
        # is re-captured whenever one of them sits at another address than at capture time
        runner = self._runner
        loop.ts_dev.copy_(ts.float())

        def prep():
            runner.state = loop.state
            runner.set_context(loop.ehs, guess, loop.conds[0].shape[0])
            if runner.mode == "grouped" and not guess:
                runner.set_conds(loop.conds)                              # conv_in + cond of all nets as one launch
                if os.environ.get("ES_TIME_TABLE", "1") == "1":
                    runner.set_time_table(loop.ts_dev, N)                 # every step's time projections, once per call
        graphs = self.use_graph and callback_on_step_end is None
        if graphs and not regraph and loop.prep_graph is not None and loop.sig is not None:
            loop.prep_graph.replay()
        else:
            prep()
        if any(skip):
            runner.state = loop.state
            runner.prepare_fused_zero(N)                                  # constants of the fusion weights: once per loop
        if loop.signature() != loop.sig:
            regraph = True

        mark("prep")
        # PL:435-543 — the denoising loop
        if not graphs:
            for i in range(T):
                loop.step_of(i)()
                if callback_on_step_end is not None:
                    out = callback_on_step_end(self, i, int(ts[i]), {"latents": _as_nchw_view(loop.latents)})
                    if out and "latents" in out:
                        # PL:529-531: the returned latents are what the next step's networks, the scheduler and the
                        # decode see
                        loop.latents.copy_(out["latents"].permute(0, 2, 3, 1))
                        loop.set_model_in()
        else:
            start = 0
            if regraph:
                loop.step_of(0)()                     # eager step 0: sizes the split-K workspace, warms kernels
                start = 1
                kinds = set(skip)                     # which step forms this call replays: full (False), UNet-only (True)
                if (not skip[0]) in kinds:
                    loop.rehearse(loop.one_step_unet if not skip[0] else loop.one_step)      # the other form, state put back
                torch.cuda.synchronize()
                saved = loop.step_idx.clone()
                loop.graph = loop.graph_unet = None
                for kind in sorted(kinds):
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g):
                        (loop.one_step_unet if kind else loop.one_step)()
                    if kind:
                        loop.graph_unet = g
                    else:
                        loop.graph = g
                loop.step_idx.copy_(saved)            # capture does not execute; keep the counter where it was
                gp = torch.cuda.CUDAGraph()           # the per-call preparation replays from the next call on
                with torch.cuda.graph(gp):
                    prep()
                loop.prep_graph, loop.decode_graph = gp, None
                loop.loop_graph = None
                loop.sig = loop.signature()
                loop.captures += 1
            elif LOOP_GRAPH and loop.loop_graph is None:
                # second call in a row with the same configuration: capture all T steps as ONE graph (BASELINE configs[2]:
                # "hipGraph-captured scheduler loop").  Every step is the same launch list - the device step counter picks
                # its table rows -, so from now on a call costs one graph launch instead of T (a graph launch is ~30 us of
                # device idle on this stack).  Not at the first call: a caller that alternates configurations re-captures
                # the one-step graph on every switch and would pay T host walks each time.
                saved = loop.step_idx.clone()
                gl = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gl):
                    for i in range(T):
                        loop.step_of(i)()
                loop.step_idx.copy_(saved)
                loop.loop_graph = gl
            if start == 0 and loop.loop_graph is not None:
                loop.loop_graph.replay()
            else:
                for i in range(start, T):
                    (loop.graph_unet if skip[i] else loop.graph).replay()

        mark("loop")
        if output_type == "latent":
            img = _as_nchw_view(loop.latents).clone()
        else:
            # PL:552-557 decode(latents / scaling_factor); PL:570-572 (x/2+0.5).clamp(0,1)
            # model_in[:B] already holds the final latents in the compute dtype, channel-padded (es_cfg_ddim_step)
            def decode():
                dec = self.vae.decode_nhwc(loop.model_in[:B], unscaled_latents=True)
                return ops.nhwc_to_nchw(dec, channels=3, scale=0.5, shift=0.5, clamp01=True)
            if graphs:
                if loop.decode_graph is None:
                    decode()                          # eager once: kernels warm, scratch sized outside the capture
                    torch.cuda.synchronize()
                    gd = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(gd):
                        loop.image = decode()
                    loop.decode_graph = gd
                loop.decode_graph.replay()
                img = loop.image.clone()              # the graph's output buffer is overwritten by the next call
            else:
                img = decode()
            if output_type in ("np", "pil"):
                arr = img.permute(0, 2, 3, 1).cpu().numpy()
                if output_type == "pil":
                    from PIL import Image
                    img = [Image.fromarray((a * 255).round().astype("uint8")) for a in arr]
                else:
                    img = arr
            elif output_type != "pt":
                raise ValueError(f"unknown output_type {output_type}")
        mark("decode")
        if ev:
            torch.cuda.synchronize()
            self.timing = {b[0]: a[1].elapsed_time(b[1]) for a, b in zip(ev[:-1], ev[1:])}
        if not return_dict:
            return (img, None)
        return StableDiffusionPipelineOutput(images=img, nsfw_content_detected=None)


    def profile_one_step(self, gemm_replay_iters: int = 0, around_replay=None):
        """Re-capture the step graph of the most recent call with in-kernel timing stamps on every es_conv_gemm
        launch, replay ONE step, and return [(meta, seconds)] (ops.Profiler).  Used by bench.py's roofline leg.

        gemm_replay_iters > 0 additionally times the SAME launch list as the production kernels run it (no stamp
        atomics): only the es_conv_gemm launches (with their split-K reduces), captured as one graph and replayed that
        many times between two HIP events on the launching stream; the average per replay lands in
        `self.last_gemm_replay_ms`.  `around_replay` = (start(expected_ms), stop()): called right before / after that event-timed
        region (bench.py samples the sustained clock and package power there; expected_ms from three warm-up replays)."""
        if not self._loops:
            raise EdgeStyleHipError("run the pipeline once before profiling")
        loop = self._last_loop or list(self._loops.values())[-1]
        prof = ops.Profiler(self.device)
        torch.cuda.synchronize()
        ops.PROFILE = prof
        try:
            loop.step_idx.zero_()
            loop.one_step()                  # eager serial warm-up: sizes lane-0 scratch outside the capture
            prof.meta.clear()
            prof.descs.clear()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                loop.one_step()
        finally:
            ops.PROFILE = None
        prof.reset()
        loop.step_idx.zero_()
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        res = prof.results()
        self.last_gemm_replay_ms = self.last_conv3_replay_ms = None
        if gemm_replay_iters > 0:
            def timed(only, hooks=None):
                prof.replay_gemms(only)      # eager once (same workspace sizes as the step: nothing grows)
                torch.cuda.synchronize()
                g2 = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g2):
                    prof.replay_gemms(only)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(3):
                    g2.replay()
                e1.record()
                torch.cuda.synchronize()
                if hooks is not None:
                    hooks[0](e0.elapsed_time(e1) / 3 * gemm_replay_iters)      # the expected length of the timed region, ms
                try:
                    e0.record()
                    for _ in range(gemm_replay_iters):
                        g2.replay()
                    e1.record()
                    torch.cuda.synchronize()
                finally:
                    if hooks is not None:
                        hooks[1]()
                return e0.elapsed_time(e1) / gemm_replay_iters
            self.last_gemm_replay_ms = timed(0, around_replay)
            self.last_conv3_replay_ms = timed(3)     # the 3x3 convolutions alone (with their split-K reduces)
        loop.step_idx.zero_()
        del g
        return res


class EdgeStyleStableDiffusionControlNetPipeline(StableDiffusionControlNetPipeline):
    """model/edgestyle_pipeline.py:57-664 — same call surface; the condition images are embedded ONCE per call (prepare_image,
    PL:629-664, with CachedControlNetModel's shortcut CL:199-203), where the stock pipeline re-samples the VAE conditions at
    every step (StableDiffusionControlNetPipeline.resample_cond_each_step)."""
    resample_cond_each_step = False
