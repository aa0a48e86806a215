"""Builder of a native execution context (es_ctx, include/edgestyle_hip.h) from a loaded pipeline.

The C ABI's step-level entry points — es_denoise_step (== OnnxUNetAndControlnets.forward, export_onnx.py:43-74),
es_denoise_loop (== the loop of model/edgestyle_pipeline.py:435-543) and es_vae_decode (PL:552-572) — execute recorded
launch lists (es_plan) against static device buffers.  Something has to build those once: pack the weights, allocate the
buffers and walk the model while the plans record.  That is this module; it is the only part that needs Python.  After
`NativeEngine(...)` returns, `engine.ctx` is a plain `es_ctx*` that any host can drive with raw device pointers
(INTEGRATION.md shows the C calls); the methods below do exactly that through ctypes and nothing else.

The recorded pointers live in torch's graph-private memory pools (every plan is recorded inside a `torch.cuda.graph`
capture, which is also what keeps the activations of a plan from being handed to anybody else); the engine keeps those
graphs alive for as long as the context exists.
"""
import ctypes as C
from typing import Optional, Sequence

import torch

from . import lib as L
from . import ops
from .lib import EdgeStyleHipError
from .schedulers import DDIMScheduler


def _p(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


class NativeEngine:
    def __init__(self, pipe, batch_size: int = 1, guidance: bool = True, num_inference_steps: int = 50,
                 height: Optional[int] = None, width: Optional[int] = None, use_graphs: bool = True):
        if not isinstance(pipe.scheduler, DDIMScheduler):
            raise EdgeStyleHipError("the native loop implements the DDIM update (the BASELINE metric's scheduler)")
        self.pipe, self.lib = pipe, L.load()
        dev = pipe.device
        ucfg, vcfg = pipe.unet.cfg, pipe.vae.cfg
        h = (height // vcfg.scale) if height else ucfg.sample_size
        w = (width // vcfg.scale) if width else ucfg.sample_size
        B, T = batch_size, num_inference_steps
        nn = len(pipe._nets)
        c0 = ucfg.block_out_channels[0]
        # 1. one ordinary pipeline call on placeholder inputs: builds the loop (static buffers), warms every kernel and
        #    sizes the split-K workspace
        g = torch.Generator().manual_seed(0)
        pe = torch.randn(B, 77, ucfg.cross_attention_dim, generator=g) * 0.5
        conds = [torch.randn(1, c0, h, w, generator=g) * 0.3 for _ in range(nn)]
        kw = dict(prompt_embeds=pe, negative_prompt_embeds=pe if guidance else None, image=conds if nn > 1 else conds[0],
                  latents=torch.randn(B, ucfg.in_channels, h, w, generator=g), guidance_scale=7.5 if guidance else 1.0,
                  num_inference_steps=T)
        pipe(output_type="pt", **kw)
        loop = self.loop = pipe._loops[(B, guidance, h, w, False)]
        runner = self.runner = pipe._runner
        if runner.mode == "streams":
            raise EdgeStyleHipError("ES_CHAIN_MODE=streams forks HIP streams inside a step: not expressible as one launch list")
        N = loop.N
        self.B, self.N, self.T, self.h, self.w, self.nn = B, N, T, h, w, nn
        self.dtype = pipe.dtype
        self.ts_dev = torch.zeros((T,), dtype=torch.float32, device=dev)
        self.ts_dev.copy_(pipe.scheduler.set_timesteps(T).float())
        grouped_tables = runner.mode == "grouped"

        def prep():
            runner.state = loop.state
            runner.set_context(loop.ehs)
            if grouped_tables:
                runner.set_conds(loop.conds)
                runner.set_time_table(self.ts_dev, N)

        def generic():
            runner.state = loop.state
            runner.set_context(loop.ehs)
            if grouped_tables:
                runner.set_conds(loop.conds)
            runner.step(loop.model_in, loop.t_rows, loop.conds, [1.0] * nn, loop.scales_cur, out=loop.noise, step_idx=None)

        out = {}

        def decode():
            dec = pipe.vae.decode_nhwc(loop.model_in[:B], unscaled_latents=True)
            out["img"] = ops.nhwc_to_nchw(dec, channels=3, scale=0.5, shift=0.5, clamp01=True)

        self._keep = []
        ctx = C.c_void_p()
        L.check(self.lib.es_ctx_create(dev.index or 0, C.byref(ctx)), "es_ctx_create")
        self.ctx = ctx
        geo = L.CtxGeometry(B=B, cfg=int(guidance), h=h, w=w, latent_channels=ucfg.in_channels,
                            latent_pad=pipe.unet.engine.in_pad, n_conds=nn, n_steps=T, dtype=L.ES_F16 if self.dtype == torch.float16 else L.ES_BF16)
        L.check(self.lib.es_ctx_set_geometry(ctx, C.byref(geo)), "es_ctx_set_geometry")
        self.plan_sizes, self.plan_forks = {}, {}
        for which, fn in ((L.PLAN_PREP, prep), (L.PLAN_STEP, loop.one_step), (L.PLAN_STEP_GENERIC, generic), (L.PLAN_DECODE, decode)):
            fn()                                     # eager: anything that allocates scratch does it outside the capture
            torch.cuda.synchronize()
            plan = C.c_void_p(self.lib.es_plan_create())
            graph = torch.cuda.CUDAGraph()
            L.check(self.lib.es_plan_begin_record(plan), "es_plan_begin_record")
            try:
                with torch.cuda.graph(graph):
                    fn()
            finally:
                L.check(self.lib.es_plan_end_record(plan), "es_plan_end_record")
            self._keep.append(graph)                 # owns the memory the plan's pointers refer to
            self.plan_sizes[which] = self.lib.es_plan_size(plan)
            self.plan_forks[which] = self.lib.es_plan_count(plan, L.PLAN_SIDE_BEGIN)
            L.check(self.lib.es_ctx_set_plan(ctx, which, plan), "es_ctx_set_plan")
        self.image = out["img"]
        binds = {L.BUF_SAMPLE: loop.model_in, L.BUF_T_ROWS: loop.t_rows, L.BUF_EHS: loop.ehs, L.BUF_SCALES: loop.scales_cur,
                 L.BUF_NOISE: loop.noise, L.BUF_LATENTS: loop.latents, L.BUF_STEP_IDX: loop.step_idx, L.BUF_T_TABLE: loop.t_table,
                 L.BUF_SCALE_TABLE: loop.scale_table, L.BUF_COEF: loop.coef, L.BUF_TIMESTEPS: self.ts_dev, L.BUF_IMAGE: self.image}
        for i, c in enumerate(loop.conds):
            binds[L.BUF_COND0 + i] = c
        for slot, t in binds.items():
            L.check(self.lib.es_ctx_bind(ctx, slot, _p(t), t.numel() * t.element_size()), "es_ctx_bind")
        ac = pipe.scheduler.alphas_cumprod.float().contiguous()
        L.check(self.lib.es_ctx_set_alphas_cumprod(ctx, ac.numpy().ctypes.data_as(C.POINTER(C.c_float)), ac.numel()),
                "es_ctx_set_alphas_cumprod")
        self.set_options(use_graphs=use_graphs)
        loop.graph = None                            # the pipeline's own graph of this loop saw other table contents: re-capture
        loop.sig = None

    # -- thin ctypes drivers: raw device pointers in, raw device pointers out ------------------------------------------
    def set_options(self, cond_scales: Optional[Sequence[float]] = None, control_guidance_start: float = 0.0,
                    control_guidance_end: float = 1.0, use_graphs: bool = True):
        arr = None
        if cond_scales is not None:
            arr = (C.c_float * 6)(*([float(s) for s in cond_scales] + [1.0] * (6 - len(cond_scales))))
        L.check(self.lib.es_ctx_set_options(self.ctx, arr, control_guidance_start, control_guidance_end, int(use_graphs)),
                "es_ctx_set_options")

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def denoise_step(self, sample: torch.Tensor, t: float, ehs: torch.Tensor, cond_embeds: Sequence[torch.Tensor],
                     scales: Optional[Sequence[float]] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """sample [N,h,w,latent_pad] dtype NHWC, ehs [N,77,D] dtype, cond_embeds n x [N,h,w,C0] dtype -> noise [N,h,w,4]."""
        loop = self.loop
        for a, b in [(sample, loop.model_in), (ehs, loop.ehs)] + list(zip(cond_embeds, loop.conds)):
            if a.shape != b.shape or a.dtype != b.dtype or not a.is_contiguous() or not a.is_cuda:
                raise EdgeStyleHipError(f"denoise_step: expected a contiguous device tensor {tuple(b.shape)} {b.dtype}")
        if out is None:
            out = torch.empty_like(loop.noise)
        ptrs = (C.c_void_p * len(cond_embeds))(*[c.data_ptr() for c in cond_embeds])
        sc = None if scales is None else (C.c_float * 6)(*([float(s) for s in scales] + [1.0] * (6 - len(scales))))
        L.check(self.lib.es_denoise_step(self.ctx, _p(sample), float(t), _p(ehs), ptrs, sc, _p(out), self._stream()),
                "es_denoise_step")
        return out

    def set_conds(self, cond_embeds: Sequence[torch.Tensor]):
        """The loop reads the condition embeddings last placed in the context's slots."""
        for dst, src in zip(self.loop.conds, cond_embeds):
            ops.memcpy(dst, src.contiguous())

    def denoise_loop(self, latents: torch.Tensor, ehs: torch.Tensor, guidance_scale: float,
                     timesteps: Optional[Sequence[float]] = None) -> torch.Tensor:
        """latents fp32 [B,h,w,L] NHWC (updated in place and returned); ehs [N,77,D] dtype."""
        if timesteps is None:
            timesteps = self.pipe.scheduler.set_timesteps(self.T).tolist()
        ts = (C.c_float * len(timesteps))(*[float(t) for t in timesteps])
        if latents.shape != self.loop.latents.shape or latents.dtype != torch.float32 or not latents.is_contiguous():
            raise EdgeStyleHipError(f"denoise_loop: latents must be fp32 contiguous {tuple(self.loop.latents.shape)}")
        L.check(self.lib.es_denoise_loop(self.ctx, _p(latents), _p(ehs.contiguous()), float(guidance_scale), ts, len(timesteps),
                                         self._stream()), "es_denoise_loop")
        return latents

    def vae_decode(self, latents: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        if out is None:
            out = torch.empty_like(self.image)
        L.check(self.lib.es_vae_decode(self.ctx, _p(latents), _p(out), self._stream()), "es_vae_decode")
        return out

    def close(self):
        if self.ctx:
            self.lib.es_ctx_destroy(self.ctx)
            self.ctx = None
            self._keep.clear()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
