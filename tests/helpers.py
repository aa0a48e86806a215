"""Shared test scaffolding: seeded random-init weight sets in the reference's on-disk key layout, and the
oracle-vs-HIP single-step check used by __graft_entry__.smoke()."""
import torch

from edgestyle_amd import config as C, weights as W


def make_weights(ucfg, vcfg, rank=4, seed=0):
    """UNet, openpose ControlNet, two ControlLoRA nets (saved keys only), fusion blocks, VAE — fp32 CPU."""
    return dict(
        unet=W.random_state_dict(W.unet_shapes(ucfg), seed, "unet."),
        openpose=W.random_state_dict(W.controlnet_shapes(ucfg), seed, "openpose."),
        lora0=W.random_state_dict(W.controllora_saved_shapes(ucfg, rank), seed, "controlnet_0."),
        lora1=W.random_state_dict(W.controllora_saved_shapes(ucfg, rank), seed, "controlnet_1."),
        fusion=W.random_state_dict(W.fusion_shapes(ucfg), seed, "fusion."),
        vae=W.random_state_dict(W.vae_shapes(vcfg), seed, "vae."),
    )


def quantize(sd, dtype=torch.float16):
    """Round weights through the storage dtype so oracle and HIP path see the same parameter values."""
    return {k: v.to(dtype).float() for k, v in sd.items()}


def oracle_nets(ws, ucfg):
    """The reference's 6-net list [agnostic-LoRA, pose, clothes-LoRA, pose, clothes-LoRA, pose] (TT:252-258) with
    LoRA nets tied to the UNet encoder (TT:259-261)."""
    from oracle import sd15_oracle as O
    n0 = O.tie_weights(ws["lora0"], ws["unet"])
    n1 = O.tie_weights(ws["lora1"], ws["unet"])
    op = ws["openpose"]
    return [(n0, ucfg), (op, ucfg), (n1, ucfg), (op, ucfg), (n1, ucfg), (op, ucfg)]


def rel_err(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-6))


def to_nhwc(x, device, dtype=torch.float16, cpad=None):
    x = x.permute(0, 2, 3, 1).contiguous()
    if cpad is not None and cpad != x.shape[-1]:
        x = torch.nn.functional.pad(x, (0, cpad - x.shape[-1]))
    return x.to(device=device, dtype=dtype)


def tiny_step_check(device="cuda:0", seed=0):
    """One full 6-cond multi-ControlNet + UNet step (== export_onnx.py:43-74) on the tiny config: HIP vs oracle.
    Returns (max abs error, error relative to the reference's largest magnitude) of noise_pred."""
    from oracle import sd15_oracle as O
    from edgestyle_amd import engine as E
    ucfg, vcfg = C.tiny_unet(), C.tiny_vae()
    ws = {k: quantize(v) for k, v in make_weights(ucfg, vcfg, seed=seed).items()}
    g = torch.Generator().manual_seed(42)
    N, s, c0 = 2, ucfg.sample_size, ucfg.block_out_channels[0]
    x = torch.randn(N, 4, s, s, generator=g).half().float()
    ehs = (torch.randn(N, 77, ucfg.cross_attention_dim, generator=g) * 0.5).half().float()
    conds = [(torch.randn(N, c0, s, s, generator=g) * 0.3).half().float() for _ in range(6)]
    scales = [1.0, 0.8, 1.0, 1.0, 0.5, 1.0]
    t = 501
    ref = O.denoise_step(ws["unet"], ucfg, ws["fusion"], oracle_nets(ws, ucfg), x, t, ehs, conds, scales)

    from edgestyle_amd.models import StepRunner
    runner = StepRunner.from_state_dicts(ws, ucfg, torch.float16, device)
    out = runner.step_nchw(x.to(device), t, ehs.to(device), [c.to(device) for c in conds], scales)
    torch.cuda.synchronize()
    err = (out.float().cpu() - ref).abs().max()
    return float(err), float(err / ref.abs().max())


# ----------------------------------------------------------------------------------------------------------------
# full-size (SD1.5 width, 64x64 latents = the benchmarked configuration) fixtures: inputs are regenerated from seeds
# (torch's CPU generator), only the oracle's OUTPUTS are committed under tests/golden/ (make_golden_full.py)
# ----------------------------------------------------------------------------------------------------------------
FULL_RANK = 32          # LoRA rank of the reference's training default (TR:275)
FULL_STEP_T = 501
FULL_STEP_SCALES = [1.0, 0.8, 1.0, 1.0, 0.5, 1.0]


def full_weights(seed=0, keys=("unet", "openpose", "lora0", "lora1", "fusion", "vae")):
    """fp16-rounded fp32 CPU weights of the full SD1.5-shaped model set (about 6 GB)."""
    ucfg, vcfg = C.sd15_unet(), C.sd15_vae()
    shapes = dict(unet=(W.unet_shapes, (ucfg,), "unet."), openpose=(W.controlnet_shapes, (ucfg,), "openpose."),
                  lora0=(W.controllora_saved_shapes, (ucfg, FULL_RANK), "controlnet_0."),
                  lora1=(W.controllora_saved_shapes, (ucfg, FULL_RANK), "controlnet_1."),
                  fusion=(W.fusion_shapes, (ucfg,), "fusion."), vae=(W.vae_shapes, (vcfg,), "vae."))
    out = {}
    for k in keys:
        fn, a, prefix = shapes[k]
        sd = W.random_state_dict(fn(*a), seed, prefix)
        out[k] = {kk: vv.half().float() for kk, vv in sd.items()}
        del sd
    return ucfg, vcfg, out


def full_step_inputs(seed=42):
    """One 6-cond CFG step at the benchmarked geometry: N = 2 samples, 64x64 latents, 77 x 768 text states."""
    ucfg = C.sd15_unet()
    g = torch.Generator().manual_seed(seed)
    N, s, c0 = 2, ucfg.sample_size, ucfg.block_out_channels[0]
    x = torch.randn(N, 4, s, s, generator=g).half().float()
    ehs = (torch.randn(N, 77, ucfg.cross_attention_dim, generator=g) * 0.5).half().float()
    conds = [(torch.randn(N, c0, s, s, generator=g) * 0.3).half().float() for _ in range(6)]
    return x, ehs, conds


def full_pipeline_inputs(seed=43, B=1):
    """4-step DDIM pipeline at 512x512 (BASELINE configs[1] geometry): latents, prompt / negative embeds, six
    pre-embedded conditions [1,320,64,64]."""
    ucfg = C.sd15_unet()
    g = torch.Generator().manual_seed(seed)
    s, c0 = ucfg.sample_size, ucfg.block_out_channels[0]
    lat = torch.randn(B, 4, s, s, generator=g)
    pe = (torch.randn(B, 77, ucfg.cross_attention_dim, generator=g) * 0.5).half().float()
    ne = (torch.randn(B, 77, ucfg.cross_attention_dim, generator=g) * 0.5).half().float()
    conds = [(torch.randn(1, c0, s, s, generator=g) * 0.3).half().float() for _ in range(6)]
    return lat, pe, ne, conds


def single_cn_inputs(seed=44):
    """BASELINE configs[0]: UNet + ONE openpose ControlNet, 512x512, 4 DDIM steps; the pose image is a raw
    [1,3,512,512] tensor in [0,1] (TT:29-48) that the net's own conv stack embeds."""
    ucfg, vcfg = C.sd15_unet(), C.sd15_vae()
    g = torch.Generator().manual_seed(seed)
    s = ucfg.sample_size
    lat = torch.randn(1, 4, s, s, generator=g)
    pe = (torch.randn(1, 77, ucfg.cross_attention_dim, generator=g) * 0.5).half().float()
    ne = (torch.randn(1, 77, ucfg.cross_attention_dim, generator=g) * 0.5).half().float()
    pose = torch.rand(1, 3, s * vcfg.scale, s * vcfg.scale, generator=g).half().float()
    return lat, pe, ne, pose


def full_rgb_inputs(seed=49):
    """The RGB-image form of the call (TT:328-359): six [1,3,512,512] condition images - IMG tensors in [-1,1] for the
    three VAE-conditioned LoRA nets (even slots), COND tensors in [0,1] for the three pose nets (odd slots, TT:29-48) -,
    the latent_dist.sample() noise of each LoRA net for the CFG-duplicated batch N = 2 (CL:39), latents, text states."""
    ucfg, vcfg = C.sd15_unet(), C.sd15_vae()
    g = torch.Generator().manual_seed(seed)
    s = ucfg.sample_size
    px = s * vcfg.scale
    imgs = []
    for i in range(6):
        u = torch.rand(1, 3, px, px, generator=g)
        # smooth images (blocks of 8 pixels blended with noise): closer to photographs than white noise is
        u = 0.7 * torch.nn.functional.interpolate(u[:, :, ::8, ::8], scale_factor=8.0, mode="bilinear", align_corners=False) + 0.3 * u
        imgs.append(((u * 2 - 1) if i % 2 == 0 else u).half().float())
    noise = [torch.randn(2, vcfg.latent_channels, s, s, generator=g).half().float() if i % 2 == 0 else None for i in range(6)]
    lat = torch.randn(1, 4, s, s, generator=g)
    pe = (torch.randn(1, 77, ucfg.cross_attention_dim, generator=g) * 0.5).half().float()
    ne = (torch.randn(1, 77, ucfg.cross_attention_dim, generator=g) * 0.5).half().float()
    return imgs, noise, lat, pe, ne


def psnr(a, b, peak=1.0):
    import math
    mse = float(((a.float().cpu() - b.float().cpu()) ** 2).mean())
    return 10 * math.log10(peak * peak / max(mse, 1e-20))


# ----------------------------------------------------------------------------------------------------------------
# The Python builder's launch lists WITHOUT a GPU: the library's dry recorder (es_plan_set_dry) validates and records
# every C-ABI call and launches nothing, so the model walk of edgestyle_amd (engine.py / models.py / pipeline._Loop /
# native.py) can run on host tensors.  Used to hold es_load_weights' plans against the Python host's, call by call.
# ----------------------------------------------------------------------------------------------------------------
def python_dry_context(ws, ucfg, vcfg, B=1, guidance=True, T=6, dtype=torch.float16,
                       controlnets=(("lora0", 1), ("openpose", 0), ("lora1", 1)), net_of_cond=(0, 1, 2, 1, 2, 1), rank=4, guess=False):
    """-> (lib, es_ctx) holding the plans the Python host records for this configuration (addresses are host
    addresses of scratch tensors: only good for plan_records).  guess: NativeEngine(guess_mode=True)'s recording."""
    import ctypes as Ct
    from types import SimpleNamespace
    from edgestyle_amd import lib as L, ops, models as M, native as Nat
    from edgestyle_amd.pipeline import EdgeStyleStableDiffusionControlNetPipeline, _Loop
    lib = L.load()
    saved = (ops._stream, M._HipModel._require_gpu, ops._get_workspace)
    ops._stream = lambda: None
    M._HipModel._require_gpu = lambda self: None
    wsb = {}

    def get_ws(nbytes, device):
        t = wsb.get("t")
        if t is None or t.numel() * 4 < nbytes:
            t = wsb["t"] = torch.empty(max(nbytes // 4, 1), dtype=torch.float32)
        return t
    ops._get_workspace = get_ws
    try:
        # controlnets: (name in ws, kind) with kind as include/edgestyle_hip.h ES_NET_*: 0 ControlNetModel, 1 ControlLoRA through
        # the VAE, 2 ControlLoRA with its own conv-stack embedding
        unet = M.UNet2DConditionModel(ws["unet"], ucfg, dtype).to("cpu")
        vae = M.AutoencoderKL(ws["vae"], vcfg, dtype)
        distinct = []
        for name, kind in controlnets:
            if kind == 0:
                net = M.ControlNetModel(ws[name], ucfg, dtype)
            else:
                net = M.ControlLoRAModel(ws[name], ucfg, dtype, lora_linear_rank=rank, uses_vae=kind == 1)
                net.tie_weights(unet)
                if kind == 1:
                    net.set_autoencoder(vae)
            distinct.append(net)
        if len(net_of_cond) == 1:                       # one ControlNet: 13 residuals straight to the UNet (PL:338-351)
            mc = distinct[net_of_cond[0]]
        else:
            mc = M.EdgeStyleMultiControlNetModel([distinct[i] for i in net_of_cond], ucfg, None)
            mc.load_state_dict(ws["fusion"])
        runner = M.StepRunner(unet, mc)
        pipe = EdgeStyleStableDiffusionControlNetPipeline(vae=vae, unet=runner.unet, controlnet=runner.controlnet)
        pipe._runner = runner
        h = w = ucfg.sample_size
        loop = _Loop(pipe, B, guidance, h, w, guess)
        N, k, nn = loop.N, runner.kmax, len(net_of_cond)
        loop.t_table = torch.zeros((T, k * N)); loop.scale_table = torch.zeros((T, nn)); loop.coef = torch.zeros((T, 4))
        loop.ts_dev = torch.zeros((T,)); loop.guidance_scale, loop.steps = (7.5 if guidance else 1.0), T
        eng = SimpleNamespace(cond_img=[None] * nn, cond_noise=[None] * nn, dtype=dtype)
        conds_fn = Nat.NativeEngine._conds_fn(eng, pipe, loop, B, guidance and not guess, h, w)
        n_cn = loop.conds[0].shape[0]
        image = {}

        def prep():
            runner.state = loop.state
            runner.set_context(loop.ehs, guess, n_cn)
            if not guess:
                runner.set_conds(loop.conds)
                runner.set_time_table(loop.ts_dev, N)

        def generic():
            runner.state = loop.state
            runner.set_context(loop.ehs, guess, n_cn)
            if not guess:
                runner.set_conds(loop.conds)
            runner.step(loop.model_in, loop.t_rows, loop.conds, [1.0] * nn, loop.scales_cur, out=loop.noise, step_idx=None,
                        guess_mode=guess)

        def decode():
            dec = vae.decode_nhwc(loop.model_in[:B], unscaled_latents=True)
            image["t"] = ops.nhwc_to_nchw(dec, channels=3, scale=0.5, shift=0.5, clamp01=True)
        ctx = Ct.c_void_p()
        L.check(lib.es_ctx_create(0, Ct.byref(ctx)), "es_ctx_create")
        keep = [runner, vae, pipe, loop, eng, image, wsb]
        if len(net_of_cond) != 1 and not guess:        # StepRunner.prepare_fused_zero without its launch: the constants' buffers
            loop.state.fused_zero = [torch.zeros((N, s_ * s_, c_), dtype=dtype) for c_, s_ in mc.engine.table]
        for which, fn in ((L.PLAN_PREP, prep), (L.PLAN_STEP, loop.one_step), (L.PLAN_STEP_GENERIC, generic), (L.PLAN_DECODE, decode),
                          (L.PLAN_CONDS, conds_fn), (L.PLAN_STEP_UNET, loop.one_step_unet)):
            if guess and which == L.PLAN_STEP_UNET:    # (not recorded in guess_mode: native.NativeEngine)
                continue
            plan = Ct.c_void_p(lib.es_plan_create())
            L.check(lib.es_plan_begin_record(plan), "begin")
            lib.es_plan_set_dry(1)
            try:
                fn()
            finally:
                lib.es_plan_set_dry(0)
                lib.es_plan_end_record(plan)
            L.check(lib.es_ctx_set_plan(ctx, which, plan), "set_plan")
        return lib, ctx, keep
    finally:
        ops._stream, M._HipModel._require_gpu, ops._get_workspace = saved


def describe_record(kind, rec):
    """A recorded call as text (diagnostics of a plan mismatch)."""
    import ctypes as Ct
    from edgestyle_amd import lib as L
    st = {1: L.GemmDesc, 2: L.XsDesc, 3: L.AttnDesc, 4: L.GnDesc, 6: L.LnDesc, 7: L.FusionDesc, 8: L.FusionDesc}.get(kind)
    if st is None or len(rec) < Ct.sizeof(st):
        return f"kind {kind}: {rec.hex()}"
    d = st.from_buffer_copy(rec[:Ct.sizeof(st)])
    out = []
    for name, tp in st._fields_:
        v = getattr(d, name)
        if hasattr(v, "__len__"):
            v = list(v)
        out.append(f"{name}={v}")
    return f"kind {kind}: " + " ".join(out)


def diff_plans(lib, ctx_a, ctx_b, which):
    """None if the two contexts hold the same calls in plan `which` (pointer fields compared as null / set), else a text."""
    from edgestyle_amd.native import plan_records
    a, b = plan_records(lib, ctx_a, which), plan_records(lib, ctx_b, which)
    for i, (ra, rb) in enumerate(zip(a, b)):
        if ra != rb:
            return f"plan {which}: call {i} of {len(a)} / {len(b)} differs\n  A {describe_record(*ra)}\n  B {describe_record(*rb)}"
    if len(a) != len(b):
        return f"plan {which}: {len(a)} vs {len(b)} calls"
    return None


def plan_constants(lib, ctx, which):
    """The persistent data every call of a plan reads - packed weights, biases, LayerNorm column sums, GroupNorm / fusion
    parameters - as host bytes, in call order.  Only for contexts whose recorded addresses are HOST addresses
    (python_dry_context, es_load_weights(device=-2))."""
    import ctypes as Ct
    import numpy as np
    from edgestyle_amd import lib as L
    pl = lib.es_ctx_plan(ctx, which)
    n = lib.es_plan_export(pl, None, 0)
    raw = (Ct.c_char * n)()
    lib.es_plan_export(pl, raw, n)
    img = bytes(raw)
    n_ops = int.from_bytes(img[:8], "little")
    blob0 = 16 + 24 * n_ops
    out = []

    def rd(addr, nbytes):
        if addr:
            out.append(Ct.string_at(addr, nbytes))
    for i in range(n_ops):
        kind, off, nb = (int.from_bytes(img[16 + 24 * i + 8 * j:24 + 24 * i + 8 * j], "little", signed=True) for j in range(3))
        rec = img[blob0 + off:blob0 + off + nb]
        if kind == 1:
            d = L.GemmDesc.from_buffer_copy(rec)
            ws_ = [d.w] if d.ngroups <= 1 else list(d.w_g)[:d.ngroups]
            bs_ = [d.bias] if d.ngroups <= 1 else list(d.bias_g)[:d.ngroups]
            cs_ = [d.ln_colsum] if d.ngroups <= 1 else list(d.ln_colsum_g)[:d.ngroups]
            for w in ws_:
                rd(w, d.rows_padded * d.Kpad * 2)
            for b in bs_ + cs_:
                rd(b, d.rows_padded * 4)
        elif kind == 2:
            d = L.XsDesc.from_buffer_copy(rec)
            for w in ([d.w] if d.ngroups <= 1 else list(d.w_g)[:d.ngroups]):
                rd(w, d.rows_padded * d.K * 2)
            for b in ([d.bias] if d.ngroups <= 1 else list(d.bias_g)[:d.ngroups]):
                rd(b, d.rows_padded * 4)
        elif kind == 4:
            d = L.GnDesc.from_buffer_copy(rec)
            gs = [d.gamma, d.beta] if d.ngroups <= 1 else list(d.gamma_g)[:d.ngroups] + list(d.beta_g)[:d.ngroups]
            for g in gs:
                rd(g, (d.C1 + d.C2) * 4)
        elif kind == 8:
            sz = Ct.sizeof(L.FusionDesc)
            for k in range(nb // sz):
                d = L.FusionDesc.from_buffer_copy(rec[k * sz:(k + 1) * sz])
                c, hw = d.C, d.HW
                for p, nbytes in ((d.w1, c * 24), (d.b1, c * 12), (d.g1, hw * c * 6), (d.be1, hw * c * 6), (d.w2, c * 12), (d.b2, c * 4),
                                  (d.g2, hw * c * 2), (d.be2, hw * c * 2), (d.w3, c * 4), (d.b3, c * 4)):
                    rd(p, nbytes)
    return out


# ----------------------------------------------------------------------------------------------------------------
# Inputs of tests/golden/ref_fusion.safetensors (make_golden_ref_fusion.py: outputs of the REFERENCE's own ControlNetBlock
# and interleave functions): the 13 (channels, size) pairs of the reference's residual table (MC:73-102) at batch 2
# ----------------------------------------------------------------------------------------------------------------
REF_FUSION_LEVELS = [(320, 64)] * 3 + [(320, 32)] + [(640, 32)] * 2 + [(640, 16)] + [(1280, 16)] * 2 + [(1280, 8)] * 3 + [(1280, 8)]


def ref_fusion_sample(y: torch.Tensor) -> torch.Tensor:
    """the stored part of a block's output [N, C, S, S]: every 4th channel on an 8 x 8 pixel grid (the LayerNorms couple every
    element of a sample, so an error anywhere shows everywhere; the full-tensor sum and abs-sum are stored beside it)"""
    st = max(y.shape[-1] // 8, 1)
    return y[:, ::4, ::st, ::st]


def ref_fusion_case(i: int, N: int = 2):
    """-> (state dict of ControlNetBlock(C, (S, S), 6) with the reference's parameter names, six residuals [N, C, S, S]);
    parameters and residuals are rounded through fp16 so that the HIP path sees the same values."""
    C, S = REF_FUSION_LEVELS[i]
    g = torch.Generator().manual_seed(9000 + i)

    def q(x):
        return x.half().float()
    sd = {
        "first_conv.weight": q(torch.randn(3 * C, 2, 1, 1, generator=g) * 0.7),
        "first_conv.bias": q(torch.randn(3 * C, generator=g) * 0.1),
        "first_normalization.weight": q(1 + 0.1 * torch.randn(3 * C, S, S, generator=g)),
        "first_normalization.bias": q(0.1 * torch.randn(3 * C, S, S, generator=g)),
        "second_conv.weight": q(torch.randn(C, 3, 1, 1, generator=g) * 0.6),
        "second_conv.bias": q(torch.randn(C, generator=g) * 0.1),
        "second_normalization.weight": q(1 + 0.1 * torch.randn(C, S, S, generator=g)),
        "second_normalization.bias": q(0.1 * torch.randn(C, S, S, generator=g)),
        "third_conv.weight": q(torch.randn(C, 1, 1, 1, generator=g)),
        "third_conv.bias": q(torch.randn(C, generator=g) * 0.1),
    }
    res = [q(torch.randn(N, C, S, S, generator=g) * (0.5 + 0.25 * k)) for k in range(6)]
    return sd, res


def ref_interleave_cases():
    g = torch.Generator().manual_seed(9100)
    a = [torch.randn(2, 8, 4, 4, generator=g) for _ in range(6)]
    b = [torch.randn(1, 5, 3, 2, generator=g) for _ in range(3)]
    return a, b
