"""Builders of a native execution context (es_ctx, include/edgestyle_hip.h): NativeEngine records one from a loaded pipeline
(this host walks the model); NativeContext, at the end of the file, has the library build it from raw state dicts
(es_load_weights: no model walk in Python at all).

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
                 height: Optional[int] = None, width: Optional[int] = None, use_graphs: bool = True,
                 embed_conditions: bool = True, guess_mode: bool = False):
        """guess_mode: the recorded step is guess_mode's (CL:256-264, PL:453-459, 487-497).
        embed_conditions: also record ES_PLAN_CONDS (es_prepare_conds: RGB condition images -> embeddings), when every
        net's conditioning path is one this builder knows (VAE-conditioned ControlLoRA nets, conv-stack ControlNets)."""
        if not isinstance(pipe.scheduler, DDIMScheduler):
            raise EdgeStyleHipError("the native loop implements the DDIM update (the BASELINE metric's scheduler)")
        self.pipe, self.lib = pipe, L.load()
        self._cond_scales = [1.0] * 6
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
                  num_inference_steps=T, guess_mode=bool(guess_mode))
        pipe(output_type="pt", **kw)
        guess = self.guess = bool(guess_mode)
        loop = self.loop = pipe._loops[(B, guidance, h, w, guess)]
        runner = self.runner = pipe._runner
        N = loop.N
        # the coefficient table the recorded scheduler call reads: room for UniPC's 12 columns (es_ctx_set_scheduler), the DDIM
        # rows the pipeline call left there in front
        self._coef = torch.zeros((T * 12,), dtype=torch.float32, device=dev)
        self._coef[:T * 4].copy_(loop.coef.reshape(-1))
        loop.coef = self._coef[:T * 4].view(T, 4)
        self.B, self.N, self.T, self.h, self.w, self.nn = B, N, T, h, w, nn
        self.dtype = pipe.dtype
        self.ts_dev = torch.zeros((T,), dtype=torch.float32, device=dev)
        self.ts_dev.copy_(pipe.scheduler.set_timesteps(T).float())
        grouped_tables = runner.mode == "grouped" and not guess
        n_cn = loop.conds[0].shape[0]

        def prep():
            runner.state = loop.state
            runner.set_context(loop.ehs, guess, n_cn)
            if grouped_tables:
                runner.set_conds(loop.conds)
                runner.set_time_table(self.ts_dev, N)

        def generic():
            runner.state = loop.state
            runner.set_context(loop.ehs, guess, n_cn)
            if grouped_tables:
                runner.set_conds(loop.conds)
            runner.step(loop.model_in, loop.t_rows, loop.conds, [1.0] * nn, loop.scales_cur, out=loop.noise, step_idx=None,
                        guess_mode=guess)

        out = {}

        def decode():
            dec = pipe.vae.decode_nhwc(loop.model_in[:B], unscaled_latents=True)
            out["img"] = ops.nhwc_to_nchw(dec, channels=3, scale=0.5, shift=0.5, clamp01=True)

        # -- es_prepare_conds: static inputs + the embedding walk of pipeline.prepare_images as C-ABI launches only --------
        self.cond_img, self.cond_noise = [None] * nn, [None] * nn
        conds_fn = self._conds_fn(pipe, loop, B, guidance and not guess, h, w) if embed_conditions else None

        self._keep = []
        ctx = C.c_void_p()
        L.check(self.lib.es_ctx_create(dev.index or 0, C.byref(ctx)), "es_ctx_create")
        self.ctx = ctx
        geo = L.CtxGeometry(B=B, cfg=int(guidance), h=h, w=w, latent_channels=ucfg.in_channels,
                            latent_pad=pipe.unet.engine.in_pad, n_conds=nn, n_steps=T, dtype=L.ES_F16 if self.dtype == torch.float16 else L.ES_BF16,
                            guess_mode=int(guess))
        L.check(self.lib.es_ctx_set_geometry(ctx, C.byref(geo)), "es_ctx_set_geometry")
        self._geo = geo
        self.plan_sizes = {}
        todo = [(L.PLAN_PREP, prep), (L.PLAN_STEP, loop.one_step), (L.PLAN_STEP_GENERIC, generic), (L.PLAN_DECODE, decode)]
        if conds_fn is not None:
            todo.append((L.PLAN_CONDS, conds_fn))
        if grouped_tables:
            # steps outside the control-guidance window (PL:419-427): the UNet alone + the fusion-of-zeros constants
            runner.state = loop.state
            runner.prepare_fused_zero(N)
            todo.append((L.PLAN_STEP_UNET, loop.one_step_unet))
        for which, fn in todo:
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
            L.check(self.lib.es_ctx_set_plan(ctx, which, plan), "es_ctx_set_plan")
        self.image = out["img"]
        binds = {L.BUF_SAMPLE: loop.model_in, L.BUF_T_ROWS: loop.t_rows, L.BUF_EHS: loop.ehs, L.BUF_SCALES: loop.scales_cur,
                 L.BUF_NOISE: loop.noise, L.BUF_LATENTS: loop.latents, L.BUF_STEP_IDX: loop.step_idx, L.BUF_T_TABLE: loop.t_table,
                 L.BUF_SCALE_TABLE: loop.scale_table, L.BUF_COEF: self._coef, L.BUF_TIMESTEPS: self.ts_dev, L.BUF_IMAGE: self.image}
        for i in range(3):
            binds[L.BUF_HIST0 + i] = loop.hist[i]
        for i, c in enumerate(loop.conds):
            binds[L.BUF_COND0 + i] = c
            if conds_fn is not None:
                binds[L.BUF_COND_IMG0 + i] = self.cond_img[i]
                if self.cond_noise[i] is not None:
                    binds[L.BUF_COND_NOISE0 + i] = self.cond_noise[i]
        self._binds = binds
        for slot, t in binds.items():
            L.check(self.lib.es_ctx_bind(ctx, slot, _p(t), t.numel() * t.element_size()), "es_ctx_bind")
        ac = pipe.scheduler.alphas_cumprod.float().contiguous()
        L.check(self.lib.es_ctx_set_alphas_cumprod(ctx, ac.numpy().ctypes.data_as(C.POINTER(C.c_float)), ac.numel()),
                "es_ctx_set_alphas_cumprod")
        self.set_options(use_graphs=use_graphs)
        # The recorded plans hold raw pointers into this loop's static buffers and StepState tensors (time-projection table,
        # batch-concatenated conditions, text K/V projections): the loop becomes PRIVATE to the engine.  It leaves the
        # pipeline's cache - a later pipe(...) with the same (B, cfg, h, w) builds its own loop and may replace ITS tables
        # (another num_inference_steps re-allocates tproj_table) without freeing anything a plan still reads - and every
        # tensor the plans can reference is pinned here for the life of the context.
        pipe._loops.pop((B, guidance, h, w, guess), None)
        if getattr(pipe, "_last_loop", None) is loop:
            pipe._last_loop = None
        from .models import StepState
        pinned = []

        def pin(o):
            if torch.is_tensor(o):
                pinned.append(o)
            elif isinstance(o, (list, tuple)):
                for x in o:
                    pin(x)
        for v in vars(loop.state).values():
            pin(v)
        for v in vars(loop).values():
            pin(v)
        self._keep.append(pinned)
        if runner.state is loop.state:
            runner.state = StepState()               # stand-alone StepRunner calls get a fresh default state again

    def _conds_fn(self, pipe, loop, B, guidance, h, w):
        """The one-time conditioning embedding (PL:352-377, 629-664; CL:28-42, 289-290) as pipeline.prepare_images does it -
        each shared encoder once over the un-duplicated images of all its nets, the VAE sampling noise per CFG half -
        on static inputs and through C-ABI launches only, so that a plan can record it.  None if a net's conditioning
        path is not one of the two known ones."""
        from .models import ControlLoRAModel, ControlNetModel
        nets = pipe._nets
        dev, dt = pipe.device, self.dtype
        rep = 2 if guidance else 1
        N = B * rep
        groups = {}
        for i, net in enumerate(nets):
            if isinstance(net, ControlLoRAModel) and net.config.uses_vae:
                if net._vae is None:
                    return None
                groups.setdefault(("vae", id(net._vae)), []).append(i)
            elif isinstance(net, ControlNetModel):
                groups.setdefault(("stack", id(net)), []).append(i)
            else:
                return None
        scale = pipe.vae.cfg.scale
        H, W = h * scale, w * scale
        gbufs = []
        for (kind, _), idx in groups.items():
            gb = torch.zeros((len(idx) * B, 3, H, W), dtype=torch.float32, device=dev)
            gbufs.append(gb)
            for k, i in enumerate(idx):
                self.cond_img[i] = gb[k * B:(k + 1) * B]
                if kind == "vae":
                    self.cond_noise[i] = torch.zeros((N, nets[i]._vae.cfg.latent_channels, h, w), dtype=torch.float32, device=dev)

        def fn():
            for ((kind, _), idx), gb in zip(groups.items(), gbufs):
                x8 = ops.nchw_to_nhwc(gb, dt, 8)
                if kind == "vae":
                    vae = nets[idx[0]]._vae
                    mom = vae.engine.encode_moments(x8)               # one encoder pass for all its nets, no CFG copy
                    for k, i in enumerate(idx):
                        mk = mom[k * B:(k + 1) * B]
                        if rep > 1:
                            mn = torch.empty((N,) + tuple(mk.shape[1:]), dtype=mk.dtype, device=dev)
                            ops.memcpy(mn[:B], mk)
                            ops.memcpy(mn[B:], mk)
                        else:
                            mn = mk
                        z = ops.vae_sample(mn, self.cond_noise[i], vae.cfg.latent_channels, nets[i].engine.in_pad, vae.cfg.scaling_factor)
                        ops.memcpy(loop.conds[i], nets[i].engine.embed_latent(z))
                else:
                    emb = nets[idx[0]].engine.embed_cond(x8)
                    for k, i in enumerate(idx):
                        e = emb[k * B:(k + 1) * B]
                        ops.memcpy(loop.conds[i][:B], e)
                        if rep > 1:
                            ops.memcpy(loop.conds[i][B:], e)
        return fn

    # -- thin ctypes drivers: raw device pointers in, raw device pointers out ------------------------------------------
    def set_options(self, cond_scales: Optional[Sequence[float]] = None, control_guidance_start: float = 0.0,
                    control_guidance_end: float = 1.0, use_graphs=True):
        """use_graphs: False = launch by launch, True = one hipGraph per plan, 2 = the whole loop as one hipGraph."""
        arr = None
        if cond_scales is not None:
            arr = (C.c_float * 6)(*([float(s) for s in cond_scales] + [1.0] * (6 - len(cond_scales))))
            self._cond_scales = list(arr)
        self._cg, self._use_graphs = (float(control_guidance_start), float(control_guidance_end)), int(use_graphs)
        L.check(self.lib.es_ctx_set_options(self.ctx, arr, control_guidance_start, control_guidance_end, int(use_graphs)),
                "es_ctx_set_options")

    def set_scheduler(self, scheduler: int):
        """L.SCHED_DDIM | L.SCHED_UNIPC: the update es_denoise_loop applies (the recorded step list is the same)."""
        L.check(self.lib.es_ctx_set_scheduler(self.ctx, int(scheduler)), "es_ctx_set_scheduler")
        self._scheduler = int(scheduler)

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

    def prepare_conds(self, images: Sequence[torch.Tensor], noise: Optional[Sequence[Optional[torch.Tensor]]] = None):
        """images: n x device fp32 [B,3,H,W]; noise: per net None or device fp32 [N,L,h,w] (VAE-conditioned nets need it)."""
        if L.PLAN_CONDS not in self.plan_sizes:
            raise EdgeStyleHipError("this context was built without ES_PLAN_CONDS")
        imgs = [im.to(self.pipe.device, torch.float32).contiguous() for im in images]
        nz = [None if (noise is None or noise[i] is None) else noise[i].to(self.pipe.device, torch.float32).contiguous()
              for i in range(len(imgs))]
        for i, im in enumerate(imgs):
            if im.shape != self.cond_img[i].shape:
                raise EdgeStyleHipError(f"prepare_conds: image {i} must be {tuple(self.cond_img[i].shape)}")
            if nz[i] is not None and self.cond_noise[i] is not None and nz[i].shape != self.cond_noise[i].shape:
                raise EdgeStyleHipError(f"prepare_conds: noise {i} must be {tuple(self.cond_noise[i].shape)}")
        ip = (C.c_void_p * len(imgs))(*[im.data_ptr() for im in imgs])
        npz = (C.c_void_p * len(imgs))(*[None if z is None else z.data_ptr() for z in nz])
        L.check(self.lib.es_prepare_conds(self.ctx, ip, npz, self._stream()), "es_prepare_conds")
        self._live = (imgs, nz)               # the copies are asynchronous: keep the sources until the next call

    def vae_decode(self, latents: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        if out is None:
            out = torch.empty_like(self.image)
        L.check(self.lib.es_vae_decode(self.ctx, _p(latents), _p(out), self._stream()), "es_vae_decode")
        return out

    # -- context image: everything es_ctx_load needs to run this context without Python ---------------------------------
    def save(self, path: str) -> dict:
        """Write a context image for es_ctx_load (include/edgestyle_hip.h).

        Relocation is TYPED: the library names the pointer fields of every recorded call (es_plan_pointer_fields) and exactly
        those 8-byte words are rewritten as offsets into one arena; a non-null pointer that is not inside a live allocator
        segment of this device is an error here, not a fault at load time.  The arena keeps the layout of every referenced
        allocator segment (an activation freed during a capture is an inactive block the captured kernels still write, and
        may straddle today's block boundaries), but only memory that the plans READ and never WRITE - packed weights, norm
        and fusion parameters, tables built at load time - is stored in the file; everything some call writes (activations,
        split-K slabs, scratch, static input slots) is produced at run time and travels as zero-filled space."""
        import struct
        import numpy as np
        torch.cuda.synchronize()
        hip = C.CDLL("libamdhip64.so")
        hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        dev_index = self.pipe.device.index or 0
        segs, blks = [], []
        for seg in torch.cuda.memory_snapshot():
            if seg.get("device", dev_index) != dev_index:
                continue
            segs.append((seg["address"], seg["total_size"]))
            a = seg["address"]
            for bk in seg.get("blocks", []):
                blks.append((bk.get("address", a), bk["size"]))
                a = bk.get("address", a) + bk["size"]
        segs.sort()
        blks.sort()
        starts = np.array([a for a, _ in segs], dtype=np.uint64)
        ends = starts + np.array([n for _, n in segs], dtype=np.uint64)
        bstarts = np.array([a for a, _ in blks], dtype=np.uint64)
        bends = bstarts + np.array([n for _, n in blks], dtype=np.uint64)

        def locate(addrs, st, en):
            idx = np.searchsorted(st, addrs, side="right").astype(np.int64) - 1
            ok = (idx >= 0) & (addrs < en[np.clip(idx, 0, len(en) - 1)])
            return np.where(ok, idx, -1)

        # pointer-field tables of the library, by op kind
        fields = {}

        def fields_of(kind):
            if kind not in fields:
                offs, uses = (C.c_int32 * 64)(), (C.c_int32 * 64)()
                elem = C.c_int32(0)
                n = self.lib.es_plan_pointer_fields(kind, offs, uses, 64, C.byref(elem))
                if n > 64:
                    raise EdgeStyleHipError("save: pointer-field table overflow")
                fields[kind] = ([(offs[i], uses[i]) for i in range(n)], elem.value)
            return fields[kind]

        plans, used = {}, set()
        blk_use = {}                                   # block index -> OR of the uses of every pointer into it
        for which in range(L.PLAN_COUNT):
            pl = self.lib.es_ctx_plan(self.ctx, which)
            if not pl:
                continue
            n = self.lib.es_plan_export(pl, None, 0)
            raw = (C.c_char * n)()
            self.lib.es_plan_export(pl, raw, n)
            img = np.frombuffer(raw, dtype=np.uint8).copy()
            n_ops = int(img[:8].view(np.uint64)[0])
            table = img[16:16 + 24 * n_ops].view(np.uint64).reshape(n_ops, 3)
            blob0 = 16 + 24 * n_ops
            rel = []
            for kind, off, nbytes in table.tolist():
                kind = int(np.int64(kind))
                fl, elem = fields_of(kind)
                if not fl:
                    continue
                reps = max(1, nbytes // elem) if elem else 1
                for r in range(reps):
                    for fo, use in fl:
                        pos = int(off) + r * elem + fo
                        if pos + 8 > int(off) + int(nbytes):
                            raise EdgeStyleHipError("save: a pointer field lies outside its record (ABI mismatch)")
                        addr = int(img[blob0 + pos:blob0 + pos + 8].view(np.uint64)[0])
                        if addr == 0:
                            continue
                        a1 = np.array([addr], dtype=np.uint64)
                        si = int(locate(a1, starts, ends)[0])
                        if si < 0:
                            raise EdgeStyleHipError(f"save: op kind {kind} holds a device pointer {addr:#x} outside every allocator "
                                                    "segment of this device - it cannot be relocated")
                        rel.append((pos, si, addr - int(starts[si])))
                        used.add(si)
                        bi = int(locate(a1, bstarts, bends)[0])
                        if bi >= 0:
                            blk_use[bi] = blk_use.get(bi, 0) | use
            plans[which] = (img, rel)
        binds = {}
        for slot, t in self._binds.items():
            a1 = np.array([t.data_ptr()], dtype=np.uint64)
            si = int(locate(a1, starts, ends)[0])
            if si < 0:
                raise EdgeStyleHipError("save: a bound buffer is not a live allocation")
            used.add(si)
            binds[slot] = (si, t.data_ptr() - int(starts[si]), t.numel() * t.element_size())
            bi = int(locate(a1, bstarts, bends)[0])
            if bi >= 0:
                blk_use[bi] = blk_use.get(bi, 0) | 2          # filled by the host entry points at run time
        # arena layout: one slot per referenced segment
        off, arena = {}, 0
        for si in sorted(used):
            off[si] = arena
            arena += (int(ends[si] - starts[si]) + 255) // 256 * 256
        # data extents: blocks that are only ever read (small blocks travel in any case: tables / flags a few KB large)
        extents = []
        for bi, use in sorted(blk_use.items()):
            nb = int(bends[bi] - bstarts[bi])
            if (use & 2) and nb > (64 << 10):
                continue
            si = int(locate(np.array([bstarts[bi]], dtype=np.uint64), starts, ends)[0])
            extents.append((int(bstarts[bi]), off[si] + int(bstarts[bi]) - int(starts[si]), nb))
        g = self._geo
        ac = self.pipe.scheduler.alphas_cumprod.float().contiguous().numpy()
        with open(path, "wb") as f:
            f.write(b"ESCTX\x03\x00\x00" + struct.pack("<IIQ", L.ABI_VERSION, len(used), arena))
            f.write(bytes(g))
            # options record of format 3: + the scheduler es_denoise_loop applies and the length of an f64 schedule (none from
            # this host: a loaded UniPC context derives its sigmas from the library's own SD1.5 table in double, as this one does)
            f.write(struct.pack("<6fffiIiI", *self._cond_scales, self._cg[0], self._cg[1], int(self._use_graphs), len(ac),
                                int(getattr(self, "_scheduler", L.SCHED_DDIM)), 0))
            f.write(ac.tobytes())
            if len(ac) & 1:
                f.write(b"\0" * 4)
            for si in sorted(used):
                f.write(struct.pack("<QQ", off[si], int(ends[si] - starts[si])))
            for which in range(L.PLAN_COUNT):
                if which not in plans:
                    f.write(struct.pack("<Q", 0))
                    continue
                img, rel = plans[which]
                f.write(struct.pack("<Q", len(img)))
                f.write(img.tobytes())
                f.write(struct.pack("<Q", len(rel)))
                for blob_off, si, o in rel:
                    f.write(struct.pack("<QQ", blob_off, off[si] + o))
            nslots = L.BUF_COUNT
            for slot in range(nslots):
                if slot in binds:
                    si, o, nb = binds[slot]
                    f.write(struct.pack("<qQ", off[si] + o, nb))
                else:
                    f.write(struct.pack("<qQ", -1, 0))
            f.write(struct.pack("<Q", len(extents)))
            for _, ao, nb in extents:
                f.write(struct.pack("<QQ", ao, nb))
            host = np.empty(64 << 20, dtype=np.uint8)
            data_bytes = 0
            for src, _, nb in extents:
                done = 0
                while done < nb:
                    n = min(nb - done, host.nbytes)
                    rc = hip.hipMemcpy(host.ctypes.data, C.c_void_p(src + done), n, 2)
                    if rc != 0:
                        raise EdgeStyleHipError(f"save: hipMemcpy failed ({rc})")
                    f.write(host[:n].tobytes())
                    done += n
                data_bytes += nb
        return dict(blocks=len(used), arena_bytes=arena, data_bytes=data_bytes, extents=len(extents),
                    relocations={w: len(r) for w, (_, r) in plans.items()})

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


# ======================================================================================================================
# es_load_weights: the same context, built by the library itself from raw tensors (no model walk in Python)
# ======================================================================================================================
def model_config(ucfg, vcfg, text_tokens: int = 77) -> "L.ModelConfig":
    """es_model_config of a UNetConfig / VAEConfig pair (edgestyle_amd/config.py)."""
    m = L.ModelConfig()
    m.in_channels, m.out_channels = ucfg.in_channels, ucfg.out_channels
    m.n_blocks = len(ucfg.block_out_channels)
    for i, c in enumerate(ucfg.block_out_channels):
        m.block_out_channels[i] = c
        m.down_has_attn[i] = int(ucfg.down_has_attn[i])
    m.layers_per_block, m.num_heads = ucfg.layers_per_block, ucfg.num_heads
    m.cross_attention_dim, m.norm_num_groups, m.norm_eps = ucfg.cross_attention_dim, ucfg.norm_num_groups, ucfg.norm_eps
    m.n_cond_embed = len(ucfg.conditioning_embedding_out_channels)
    for i, c in enumerate(ucfg.conditioning_embedding_out_channels):
        m.cond_embed_channels[i] = c
    m.conditioning_channels, m.text_tokens = ucfg.conditioning_channels, text_tokens
    m.vae_n_blocks = len(vcfg.block_out_channels)
    for i, c in enumerate(vcfg.block_out_channels):
        m.vae_block_out_channels[i] = c
    m.vae_layers_per_block, m.vae_latent_channels = vcfg.layers_per_block, vcfg.latent_channels
    m.vae_norm_num_groups, m.vae_norm_eps, m.vae_scaling_factor = vcfg.norm_num_groups, vcfg.norm_eps, vcfg.scaling_factor
    return m


_TENSOR_DT = {torch.float32: L.ES_F32, torch.float16: L.ES_F16, torch.bfloat16: L.ES_BF16}


def state_dict_descriptors(sd):
    """{key: tensor (host, or on the GPU the context is built for)} -> (es_state_dict, keep-alive list): descriptors only."""
    arr = (L.Tensor * max(len(sd), 1))()
    keep = [arr]
    for i, (k, v) in enumerate(sd.items()):
        if v.device.type not in ("cpu", "cuda") or v.dtype not in _TENSOR_DT:
            raise EdgeStyleHipError(f"es_load_weights takes host tensors in fp32 / fp16 / bf16: {k} is {v.dtype} on {v.device}")
        v = v.contiguous()
        kb = k.encode()
        keep += [v, kb]
        arr[i].key, arr[i].data, arr[i].ndim, arr[i].dtype = kb, v.data_ptr(), v.dim(), _TENSOR_DT[v.dtype]
        if v.dim() < 1 or v.dim() > 4:
            raise EdgeStyleHipError(f"{k}: tensors of 1..4 dimensions")
        for j, s in enumerate(v.shape):
            arr[i].shape[j] = s
    return L.StateDict(arr, len(sd)), keep


class NativeContext:
    """A context built by es_load_weights (include/edgestyle_hip.h) from state dicts in the reference's key layout, and thin
    ctypes drivers of the step-level entry points on raw device pointers.  Nothing of edgestyle_amd's model code (models.py,
    engine.py, ops.py) is involved: torch only hands over host pointers at build time and device pointers at run time.

    ws: {"unet", "vae", "fusion", <controlnet names>} state dicts; `controlnets`: [(name in ws, L.NET_*)] the distinct nets;
    net_of_cond: which of them serves each of the six condition slots (TT:252-258) - or ONE entry for a single ControlNet whose
    residuals go to the UNet without fusion blocks (PL:338-351; no "fusion" dict then).  device < 0: dry build (no GPU)."""

    def __init__(self, ws, ucfg, vcfg, batch_size: int = 1, guidance: bool = True, num_inference_steps: int = 50,
                 height: Optional[int] = None, width: Optional[int] = None, dtype=torch.float16, device: int = 0,
                 controlnets=(("lora0", L.NET_CONTROL_LORA_VAE), ("openpose", L.NET_CONTROLNET), ("lora1", L.NET_CONTROL_LORA_VAE)),
                 net_of_cond=(0, 1, 2, 1, 2, 1), guess_mode: bool = False):
        self.lib = L.load()
        self.ucfg, self.vcfg = ucfg, vcfg
        h = (height // vcfg.scale) if height else ucfg.sample_size
        w = (width // vcfg.scale) if width else ucfg.sample_size
        nn = len(net_of_cond)
        self.B, self.N, self.T, self.h, self.w, self.nn = batch_size, batch_size * (2 if guidance else 1), num_inference_steps, h, w, nn
        self.dtype, self.device = dtype, device
        wts = L.Weights()
        keep = []
        for name in ("unet", "vae", "fusion"):
            if name == "fusion" and nn == 1:
                continue
            d, k = state_dict_descriptors(ws[name])
            setattr(wts, name, d)
            keep.append(k)
        for i, (name, kind) in enumerate(controlnets):
            d, k = state_dict_descriptors(ws[name])
            wts.controlnet[i], wts.controlnet_kind[i] = d, kind
            keep.append(k)
        wts.n_controlnets = len(controlnets)
        for i, n in enumerate(net_of_cond):
            wts.net_of_cond[i] = n
        geo = L.CtxGeometry(B=batch_size, cfg=int(guidance), h=h, w=w, latent_channels=ucfg.in_channels,
                            latent_pad=(ucfg.in_channels + 7) // 8 * 8, n_conds=nn, n_steps=num_inference_steps,
                            dtype=L.ES_F16 if dtype == torch.float16 else L.ES_BF16, guess_mode=int(bool(guess_mode)))
        mc = model_config(ucfg, vcfg)
        ctx = C.c_void_p()
        L.check(self.lib.es_load_weights(C.byref(wts), C.byref(mc), C.byref(geo), device, C.byref(ctx)), "es_load_weights")
        self.ctx = ctx
        self.uses_noise = [controlnets[n][1] == L.NET_CONTROL_LORA_VAE for n in net_of_cond]
        del keep

    def plan_size(self, which: int) -> int:
        return self.lib.es_ctx_plan_size(self.ctx, which)

    def set_options(self, cond_scales: Optional[Sequence[float]] = None, control_guidance_start: float = 0.0,
                    control_guidance_end: float = 1.0, use_graphs=True):
        arr = None if cond_scales is None else (C.c_float * 6)(*([float(s) for s in cond_scales] + [1.0] * (6 - len(cond_scales))))
        L.check(self.lib.es_ctx_set_options(self.ctx, arr, control_guidance_start, control_guidance_end, int(use_graphs)),
                "es_ctx_set_options")

    def set_scheduler(self, scheduler: int):
        """L.SCHED_DDIM | L.SCHED_UNIPC (UniPCMultistepScheduler with the SD1.5 config, TT:273)."""
        L.check(self.lib.es_ctx_set_scheduler(self.ctx, int(scheduler)), "es_ctx_set_scheduler")

    def set_alphas_cumprod(self, alphas_cumprod):
        """The scheduler's schedule (default: the library's own SD1.5 table, which equals torch's cumprod to 1e-6 only - hand
        over the scheduler's tensor where the DDIM coefficients must match a Python host's bit for bit)."""
        ac = alphas_cumprod.float().contiguous()
        L.check(self.lib.es_ctx_set_alphas_cumprod(self.ctx, ac.numpy().ctypes.data_as(C.POINTER(C.c_float)), ac.numel()),
                "es_ctx_set_alphas_cumprod")

    @staticmethod
    def _stream():
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def denoise_step(self, sample, t: float, ehs, cond_embeds, scales=None, out=None):
        """sample [N,h,w,latent_pad], ehs [N,77,D], cond_embeds 6 x [N,h,w,C0] (compute dtype, contiguous, device) -> [N,h,w,4]"""
        if out is None:
            out = torch.empty((self.N, self.h, self.w, self.ucfg.out_channels), dtype=self.dtype, device=sample.device)
        ptrs = (C.c_void_p * self.nn)(*[c.data_ptr() for c in cond_embeds])
        sc = None if scales is None else (C.c_float * self.nn)(*[float(s) for s in scales])
        L.check(self.lib.es_denoise_step(self.ctx, _p(sample), float(t), _p(ehs), ptrs, sc, _p(out), self._stream()), "es_denoise_step")
        return out

    def prepare_conds(self, images, noise=None):
        """images: 6 x device fp32 [B,3,H,W]; noise: per slot None or device fp32 [N,L,h,w]"""
        self._live = ([im.contiguous() for im in images], [None if (noise is None or z is None) else z.contiguous() for z in (noise or [None] * self.nn)])
        ip = (C.c_void_p * self.nn)(*[im.data_ptr() for im in self._live[0]])
        npz = (C.c_void_p * self.nn)(*[None if z is None else z.data_ptr() for z in self._live[1]])
        L.check(self.lib.es_prepare_conds(self.ctx, ip, npz, self._stream()), "es_prepare_conds")

    def denoise_loop(self, latents, ehs, guidance_scale: float, timesteps):
        """latents fp32 [B,h,w,L] NHWC device (in place); timesteps: host list of length n_steps"""
        ts = (C.c_float * len(timesteps))(*[float(t) for t in timesteps])
        L.check(self.lib.es_denoise_loop(self.ctx, _p(latents), _p(ehs), float(guidance_scale), ts, len(timesteps), self._stream()),
                "es_denoise_loop")
        return latents

    def vae_decode(self, latents, out=None):
        if out is None:
            s = self.vcfg.scale
            out = torch.empty((self.B, 3, self.h * s, self.w * s), dtype=torch.float32, device=latents.device)
        L.check(self.lib.es_vae_decode(self.ctx, _p(latents), _p(out), self._stream()), "es_vae_decode")
        return out

    def close(self):
        if getattr(self, "ctx", None):
            self.lib.es_ctx_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def plan_records(lib, ctx, which: int):
    """[(op kind, argument record with every pointer field replaced by 0 / 1 = null / set)] of one plan of a context: what two
    builders must agree on call by call (the addresses themselves belong to each builder's allocator)."""
    import numpy as np
    pl = lib.es_ctx_plan(ctx, which)
    if not pl:
        return []
    n = lib.es_plan_export(pl, None, 0)
    raw = (C.c_char * n)()
    lib.es_plan_export(pl, raw, n)
    img = np.frombuffer(raw, dtype=np.uint8).copy()
    n_ops = int(img[:8].view(np.uint64)[0])
    table = img[16:16 + 24 * n_ops].view(np.uint64).reshape(n_ops, 3)
    blob0 = 16 + 24 * n_ops
    out, fields = [], {}
    for kind, off, nbytes in table.tolist():
        kind = int(np.int64(kind))
        if kind not in fields:
            offs, uses, elem = (C.c_int32 * 64)(), (C.c_int32 * 64)(), C.c_int32(0)
            k = lib.es_plan_pointer_fields(kind, offs, uses, 64, C.byref(elem))
            fields[kind] = ([offs[i] for i in range(k)], elem.value)
        fl, elem = fields[kind]
        rec = img[blob0 + int(off):blob0 + int(off) + int(nbytes)].copy()
        for r in range(max(1, int(nbytes) // elem) if elem else 1):
            for fo in fl:
                pos = r * elem + fo
                v = int(rec[pos:pos + 8].view(np.uint64)[0])
                rec[pos:pos + 8] = 0
                rec[pos] = 1 if v else 0
        out.append((kind, rec.tobytes()))
    return out
