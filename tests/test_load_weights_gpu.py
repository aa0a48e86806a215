"""es_load_weights (include/edgestyle_hip.h): a context built by the library itself from raw state-dict tensors -
no model walk in Python - driven through ctypes with raw device pointers, against the context the Python host builds
(NativeEngine) and the pipeline: bit for bit.  What is replaced on the loading side: MC:173-211, 289-430 (fusion weights,
per-net directories), CL:600-632, 728-777 (tie_weights, LoRA state dict, fuse) and the diffusers module construction of
TT:224-261."""
import ctypes as C
import dataclasses

import pytest
import torch

from edgestyle_amd import config as Cfg, lib as L
from tests.helpers import make_weights, quantize, diff_plans

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def both():
    from edgestyle_amd.models import StepRunner, AutoencoderKL
    from edgestyle_amd.pipeline import EdgeStyleStableDiffusionControlNetPipeline
    from edgestyle_amd.native import NativeEngine, NativeContext
    ucfg, vcfg = dataclasses.replace(Cfg.tiny_unet(), sample_size=64), Cfg.tiny_vae()
    ws = {k: quantize(v) for k, v in make_weights(ucfg, vcfg, seed=5).items()}
    runner = StepRunner.from_state_dicts(ws, ucfg, torch.float16, DEV)
    vae = AutoencoderKL(ws["vae"], vcfg).to(DEV)
    for n in runner.controlnet.nets:
        if getattr(n.config, "uses_vae", False):
            n.set_autoencoder(vae)
    pipe = EdgeStyleStableDiffusionControlNetPipeline(vae=vae, unet=runner.unet, controlnet=runner.controlnet).to(DEV)
    T = 5
    eng = NativeEngine(pipe, batch_size=1, num_inference_steps=T)
    nat = NativeContext(ws, ucfg, vcfg, batch_size=1, guidance=True, num_inference_steps=T, device=0)
    nat.set_alphas_cumprod(pipe.scheduler.alphas_cumprod)      # torch's cumprod bits (the library's own table is equal to 1e-6)
    yield pipe, eng, nat, ws, ucfg, vcfg, T
    nat.close()
    eng.close()


def _inputs(ucfg, vcfg, seed):
    g = torch.Generator().manual_seed(seed)
    s, c0 = ucfg.sample_size, ucfg.block_out_channels[0]
    px = s * vcfg.scale
    lat = torch.randn(1, 4, s, s, generator=g)
    pe = (torch.randn(1, 77, ucfg.cross_attention_dim, generator=g) * 0.5).half().float()
    ne = (torch.randn(1, 77, ucfg.cross_attention_dim, generator=g) * 0.5).half().float()
    conds = [(torch.randn(1, c0, s, s, generator=g) * 0.3).half().float() for _ in range(6)]
    imgs = [(torch.rand(1, 3, px, px, generator=g) * (2 if i % 2 == 0 else 1) - (1 if i % 2 == 0 else 0)).half().float() for i in range(6)]
    noise = [torch.randn(2, vcfg.latent_channels, s, s, generator=g).half().float() if i % 2 == 0 else None for i in range(6)]
    return lat, pe, ne, conds, imgs, noise


def test_the_two_builders_record_the_same_calls(both):
    pipe, eng, nat, ws, ucfg, vcfg, T = both
    lib = L.load()
    for which in range(L.PLAN_COUNT):
        assert diff_plans(lib, eng.ctx, nat.ctx, which) is None


def test_es_denoise_step_of_a_natively_built_context_equals_the_python_step_bitwise(both):
    from edgestyle_amd.models import _as_nhwc
    pipe, eng, nat, ws, ucfg, vcfg, T = both
    lat, pe, ne, conds, _, _ = _inputs(ucfg, vcfg, 31)
    x = torch.cat([lat, lat]).half().float()
    ehs = torch.cat([ne, pe])
    sample = _as_nhwc(x, torch.float16, DEV, pipe.unet.engine.in_pad).contiguous()
    ehs_d = ehs.to(DEV, torch.float16).contiguous()
    cond_d = [_as_nhwc(c.repeat(2, 1, 1, 1), torch.float16, DEV).contiguous() for c in conds]
    scales = [1.0, 0.8, 1.0, 1.0, 0.5, 1.0]
    want = eng.denoise_step(sample, 441.0, ehs_d, cond_d, scales).clone()
    for use_graphs in (True, False):
        nat.set_options(use_graphs=use_graphs)
        got = nat.denoise_step(sample, 441.0, ehs_d, cond_d, scales)
        torch.cuda.synchronize()
        assert torch.isfinite(got.float()).all() and float(got.float().abs().max()) > 1e-3
        assert torch.equal(got, want), float((got.float() - want.float()).abs().max())


def test_rgb_images_to_decoded_image_through_the_native_context_equal_the_pipeline_bitwise(both):
    """es_prepare_conds -> es_denoise_loop -> es_vae_decode of the natively built context == pipe(image=rgb, cond_noise=...)."""
    pipe, eng, nat, ws, ucfg, vcfg, T = both
    lat, pe, ne, _, imgs, noise = _inputs(ucfg, vcfg, 37)
    gs = 6.0
    kw = dict(prompt_embeds=pe, negative_prompt_embeds=ne, image=imgs, latents=lat, guidance_scale=gs, num_inference_steps=T,
              cond_noise=noise)
    want_lat = pipe(output_type="latent", **kw).images.clone()
    want_img = pipe(output_type="pt", **kw).images.clone()
    ehs = torch.cat([ne, pe]).to(DEV, torch.float16).contiguous()
    x = lat.permute(0, 2, 3, 1).contiguous().to(DEV)
    ts = pipe.scheduler.set_timesteps(T).tolist()
    for use_graphs in (True, False, 2):
        nat.set_options(use_graphs=use_graphs)
        nat.prepare_conds([im.to(DEV) for im in imgs], [None if z is None else z.to(DEV) for z in noise])
        got = nat.denoise_loop(x.clone(), ehs, gs, ts)
        img = nat.vae_decode(got)
        torch.cuda.synchronize()
        assert torch.equal(got.permute(0, 3, 1, 2), want_lat), float((got.permute(0, 3, 1, 2) - want_lat).abs().max())
        assert torch.equal(img, want_img)
    # the condition embeddings themselves, slot by slot, against the Python host's context
    eng.prepare_conds([im.to(DEV) for im in imgs], [None if z is None else z.to(DEV) for z in noise])
    torch.cuda.synchronize()
    lib = L.load()
    for i in range(6):
        nb = C.c_size_t(0)
        p = lib.es_ctx_buffer(nat.ctx, L.BUF_COND0 + i, C.byref(nb))
        mine = torch.empty(nb.value // 2, dtype=torch.float16, device=DEV)
        L.check(lib.es_memcpy(C.c_void_p(mine.data_ptr()), C.c_void_p(p), nb.value, None), "es_memcpy")
        torch.cuda.synchronize()
        assert torch.equal(mine, eng.loop.conds[i].reshape(-1))


def test_unipc_through_the_native_loop_equals_the_pipeline_bitwise(both):
    """es_ctx_set_scheduler(ES_SCHED_UNIPC): the scheduler the reference's callers assign (TT:273, APP:118).  The coefficient
    rows are derived in C (csrc/plan.hip) exactly as schedulers.UniPCMultistepScheduler derives them; the recorded step list is
    the DDIM one with its scheduler call re-issued as es_cfg_unipc_step.  Both contexts (es_load_weights', the Python host's)
    against pipe.scheduler = UniPCMultistepScheduler, graphs and launch by launch, and back to DDIM."""
    from edgestyle_amd.schedulers import UniPCMultistepScheduler
    from edgestyle_amd.models import _as_nhwc
    pipe, eng, nat, ws, ucfg, vcfg, T = both
    lat, pe, ne, conds, _, _ = _inputs(ucfg, vcfg, 47)
    gs = 4.0
    kw = dict(prompt_embeds=pe, negative_prompt_embeds=ne, image=conds, latents=lat, guidance_scale=gs, num_inference_steps=T)
    want_ddim = pipe(output_type="latent", **kw).images.clone()
    old = pipe.scheduler
    pipe.scheduler = UniPCMultistepScheduler.from_config(old.config)
    try:
        want = pipe(output_type="latent", **kw).images.clone()
        ts = pipe.scheduler.set_timesteps(T).tolist()
    finally:
        pipe.scheduler = old
    assert not torch.equal(want, want_ddim)
    cd = [_as_nhwc(c.repeat(2, 1, 1, 1), torch.float16, DEV).contiguous() for c in conds]
    ehs = torch.cat([ne, pe]).to(DEV, torch.float16).contiguous()
    x = lat.permute(0, 2, 3, 1).contiguous().to(DEV)
    lib = L.load()
    for i in range(6):
        nb = C.c_size_t(0)
        p = lib.es_ctx_buffer(nat.ctx, L.BUF_COND0 + i, C.byref(nb))
        L.check(lib.es_memcpy(C.c_void_p(p), C.c_void_p(cd[i].data_ptr()), nb.value, None), "es_memcpy")
    eng.set_conds(cd)
    torch.cuda.synchronize()
    for ctx in (nat, eng):
        ctx.set_scheduler(L.SCHED_UNIPC)
        try:
            for use_graphs in (True, False, 2):
                ctx.set_options(use_graphs=use_graphs)
                got = ctx.denoise_loop(x.clone(), ehs, gs, ts)
                torch.cuda.synchronize()
                assert torch.equal(got.permute(0, 3, 1, 2), want), (type(ctx).__name__, use_graphs, float((got.permute(0, 3, 1, 2) - want).abs().max()))
        finally:
            ctx.set_scheduler(L.SCHED_DDIM)
            ctx.set_options(use_graphs=True)
        got = ctx.denoise_loop(x.clone(), ehs, gs, old.set_timesteps(T).tolist())
        torch.cuda.synchronize()
        assert torch.equal(got.permute(0, 3, 1, 2), want_ddim)


def test_a_context_without_guidance_and_two_images_per_call(both):
    """B = 2, no classifier-free guidance: another geometry through both builders."""
    from edgestyle_amd.native import NativeContext
    pipe, eng, nat, ws, ucfg, vcfg, T = both
    g = torch.Generator().manual_seed(41)
    s, c0 = ucfg.sample_size, ucfg.block_out_channels[0]
    lat = torch.randn(2, 4, s, s, generator=g)
    pe = (torch.randn(2, 77, ucfg.cross_attention_dim, generator=g) * 0.5).half().float()
    conds = [(torch.randn(1, c0, s, s, generator=g) * 0.3).half().float() for _ in range(6)]
    want = pipe(prompt_embeds=pe, image=conds, latents=lat, guidance_scale=1.0, num_inference_steps=3, output_type="latent").images.clone()
    n2 = NativeContext(ws, ucfg, vcfg, batch_size=2, guidance=False, num_inference_steps=3, device=0)
    n2.set_alphas_cumprod(pipe.scheduler.alphas_cumprod)
    try:
        from edgestyle_amd.models import _as_nhwc
        lib = L.load()
        for i, c in enumerate(conds):                          # pre-embedded conditions straight into the slots
            src = _as_nhwc(c.repeat(2, 1, 1, 1), torch.float16, DEV).contiguous()
            nb = C.c_size_t(0)
            p = lib.es_ctx_buffer(n2.ctx, L.BUF_COND0 + i, C.byref(nb))
            assert nb.value == src.numel() * 2
            L.check(lib.es_memcpy(C.c_void_p(p), C.c_void_p(src.data_ptr()), nb.value, None), "es_memcpy")
        torch.cuda.synchronize()
        x = lat.permute(0, 2, 3, 1).contiguous().to(DEV)
        got = n2.denoise_loop(x, pe.to(DEV, torch.float16).contiguous(), 1.0, pipe.scheduler.set_timesteps(3).tolist())
        torch.cuda.synchronize()
        assert torch.equal(got.permute(0, 3, 1, 2), want)
    finally:
        n2.close()


def test_compiled_host_from_checkpoint_directories_to_image(both, tmp_path):
    """examples/checkpoint_host.cpp: a C++ program that memory-maps safetensors files in the reference's directory layout
    (MC:213-282, 380-398: fusion blocks + controlnet_{0,1}/ of LoRA nets; diffusers dirs for UNet / VAE / openpose), hands the
    tensors to es_load_weights and runs conditions -> loop -> decode.  No Python in that process; its output equals the
    pipeline's here bit for bit."""
    import os
    import shutil
    import subprocess
    import numpy as np
    from edgestyle_amd import weights as W
    pipe, eng, nat, ws, ucfg, vcfg, T = both
    lat, pe, ne, _, imgs, noise = _inputs(ucfg, vcfg, 43)
    gs = 5.5
    kw = dict(prompt_embeds=pe, negative_prompt_embeds=ne, image=imgs, latents=lat, guidance_scale=gs, num_inference_steps=T,
              cond_noise=noise)
    want_lat = pipe(output_type="latent", **kw).images.float().cpu()
    want_img = pipe(output_type="pt", **kw).images.float().cpu()
    # the checkpoint directories, written the way the reference's save_pretrained methods lay them out
    d = {k: str(tmp_path / k) for k in ("unet", "vae", "multi", "openpose")}
    W.save_model_dir(d["unet"], ws["unet"], ucfg.to_dict())
    W.save_model_dir(d["vae"], {k: v.half() for k, v in ws["vae"].items()}, vcfg.to_dict())          # an fp16 checkpoint
    W.save_model_dir(d["openpose"], ws["openpose"], ucfg.to_dict())
    W.save_model_dir(d["multi"], ws["fusion"])
    W.save_model_dir(os.path.join(d["multi"], "controlnet_0"), ws["lora0"], dict(ucfg.to_dict(), uses_vae=True, lora_linear_rank=4))
    W.save_model_dir(os.path.join(d["multi"], "controlnet_1"), ws["lora1"], dict(ucfg.to_dict(), uses_vae=True, lora_linear_rank=4))
    s = ucfg.sample_size
    Hpx = s * vcfg.scale
    with open(str(tmp_path / "in.bin"), "wb") as f:
        f.write(np.array([1, s, s, 4, ucfg.cross_attention_dim, 6, T] + [int(z is not None) for z in noise], dtype=np.int32).tobytes())
        f.write(np.float32(gs).tobytes())
        f.write(pipe.scheduler.set_timesteps(T).float().numpy().tobytes())
        f.write(lat.permute(0, 2, 3, 1).contiguous().numpy().astype(np.float32).tobytes())
        f.write(torch.cat([ne, pe]).half().numpy().tobytes())
        for im, nz in zip(imgs, noise):
            f.write(im.numpy().astype(np.float32).tobytes())
            if nz is not None:
                f.write(nz.numpy().astype(np.float32).tobytes())
        ac = pipe.scheduler.alphas_cumprod.float().numpy()
        f.write(np.int32(len(ac)).tobytes())
        f.write(ac.tobytes())
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = str(tmp_path / "checkpoint_host")
    libdir = os.path.join(root, "edgestyle_amd", "lib")
    c = subprocess.run([hipcc, "-O1", "-I" + os.path.join(root, "include"), os.path.join(root, "examples", "checkpoint_host.cpp"),
                        "-L" + libdir, "-ledgestyle_hip", "-Wl,-rpath," + libdir, "-o", exe], capture_output=True, text=True, timeout=600)
    assert c.returncode == 0, c.stderr[-2000:]
    r = subprocess.run([exe, d["unet"], d["vae"], d["multi"], d["openpose"], str(tmp_path / "in.bin"), str(tmp_path / "out.bin")],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    raw = np.fromfile(str(tmp_path / "out.bin"), dtype=np.float32)
    nl = s * s * 4
    got_lat = torch.from_numpy(raw[:nl].reshape(1, s, s, 4).copy()).permute(0, 3, 1, 2)
    got_img = torch.from_numpy(raw[nl:].reshape(1, 3, Hpx, Hpx).copy())
    assert torch.equal(got_lat, want_lat), float((got_lat - want_lat).abs().max())
    assert torch.equal(got_img, want_img)


def test_single_controlnet_context_equals_the_pipeline_bitwise(both):
    """n_conds = 1 (BASELINE configs[0], PL:338-351): one openpose ControlNet, its RGB pose image embedded by the net's own
    conv stack (es_prepare_conds), 13 residuals added to the UNet's skips without fusion blocks."""
    from edgestyle_amd.native import NativeContext
    from edgestyle_amd.pipeline import StableDiffusionControlNetPipeline
    pipe, eng, nat, ws, ucfg, vcfg, T = both
    pose = pipe.controlnet.nets[1]
    p1 = StableDiffusionControlNetPipeline(vae=pipe.vae, unet=pipe.unet, controlnet=pose).to(DEV)
    g = torch.Generator().manual_seed(59)
    s = ucfg.sample_size
    lat = torch.randn(1, 4, s, s, generator=g)
    pe = (torch.randn(1, 77, ucfg.cross_attention_dim, generator=g) * 0.5).half().float()
    ne = (torch.randn(1, 77, ucfg.cross_attention_dim, generator=g) * 0.5).half().float()
    img = torch.rand(1, 3, s * vcfg.scale, s * vcfg.scale, generator=g).half().float()
    kw = dict(prompt_embeds=pe, negative_prompt_embeds=ne, image=img, latents=lat, guidance_scale=5.0, num_inference_steps=3,
              controlnet_conditioning_scale=0.8)
    want_lat = p1(output_type="latent", **kw).images.clone()
    want_img = p1(output_type="pt", **kw).images.clone()
    n1 = NativeContext({k: v for k, v in ws.items() if k != "fusion"}, ucfg, vcfg, num_inference_steps=3, device=0,
                       controlnets=(("openpose", L.NET_CONTROLNET),), net_of_cond=(0,))
    try:
        n1.set_alphas_cumprod(p1.scheduler.alphas_cumprod)
        n1.set_options(cond_scales=[0.8])
        n1.prepare_conds([img.to(DEV)], [None])
        x = lat.permute(0, 2, 3, 1).contiguous().to(DEV)
        got = n1.denoise_loop(x, torch.cat([ne, pe]).to(DEV, torch.float16).contiguous(), 5.0, p1.scheduler.set_timesteps(3).tolist())
        out = n1.vae_decode(got)
        torch.cuda.synchronize()
        assert torch.equal(got.permute(0, 3, 1, 2), want_lat), float((got.permute(0, 3, 1, 2) - want_lat).abs().max())
        assert torch.equal(out, want_img)
    finally:
        n1.close()


def test_bf16_context_equals_the_python_step_bitwise():
    """Compute dtype bf16 (BASELINE configs[4]'s): weights packed as bf16 by both builders, one denoising step bit for bit."""
    from edgestyle_amd.models import StepRunner, _as_nhwc, _as_nchw_view
    from edgestyle_amd.native import NativeContext
    ucfg, vcfg = dataclasses.replace(Cfg.tiny_unet(), sample_size=64), Cfg.tiny_vae()
    ws = {k: {kk: vv.bfloat16().float() for kk, vv in v.items()} for k, v in make_weights(ucfg, vcfg, seed=7).items()}
    runner = StepRunner.from_state_dicts(ws, ucfg, torch.bfloat16, DEV)
    g = torch.Generator().manual_seed(61)
    s, c0 = ucfg.sample_size, ucfg.block_out_channels[0]
    x = torch.randn(2, 4, s, s, generator=g).bfloat16().float()
    ehs = (torch.randn(2, 77, ucfg.cross_attention_dim, generator=g) * 0.5).bfloat16().float()
    conds = [(torch.randn(2, c0, s, s, generator=g) * 0.3).bfloat16().float() for _ in range(6)]
    scales = [1.0, 0.8, 1.0, 1.0, 0.5, 1.0]
    want = runner.step_nchw(x.to(DEV), 301, ehs.to(DEV), [c.to(DEV) for c in conds], scales).clone()
    assert want.dtype == torch.bfloat16 and runner.mode == "grouped"
    nat = NativeContext(ws, ucfg, vcfg, batch_size=1, guidance=True, num_inference_steps=3, dtype=torch.bfloat16, device=0)
    try:
        got = nat.denoise_step(_as_nhwc(x, torch.bfloat16, DEV, 8).contiguous(), 301.0, ehs.to(DEV, torch.bfloat16).contiguous(),
                               [_as_nhwc(c, torch.bfloat16, DEV).contiguous() for c in conds], scales)
        torch.cuda.synchronize()
        assert torch.equal(_as_nchw_view(got), want), float((_as_nchw_view(got).float() - want.float()).abs().max())
    finally:
        nat.close()


def test_contexts_release_their_arena_and_graphs(both):
    """es_ctx_destroy frees the arena, the instantiated graphs and the pinned staging: building, running and destroying contexts
    in a row leaves the device's free memory where it was."""
    from edgestyle_amd.native import NativeContext
    pipe, eng, nat, ws, ucfg, vcfg, T = both
    lat, pe, ne, conds, imgs, noise = _inputs(ucfg, vcfg, 67)
    ehs = torch.cat([ne, pe]).to(DEV, torch.float16).contiguous()
    x = lat.permute(0, 2, 3, 1).contiguous().to(DEV)
    ims = [im.to(DEV) for im in imgs]
    nz = [None if z is None else z.to(DEV) for z in noise]
    ts = pipe.scheduler.set_timesteps(3).tolist()
    torch.cuda.synchronize()
    free0 = None
    outs = []
    for it in range(4):
        c = NativeContext(ws, ucfg, vcfg, batch_size=1, guidance=True, num_inference_steps=3, device=0)
        for use_graphs in (True, 2):
            c.set_options(use_graphs=use_graphs)
            c.prepare_conds(ims, nz)
            got = c.denoise_loop(x.clone(), ehs, 6.0, ts)
            img = c.vae_decode(got)
        torch.cuda.synchronize()
        outs.append(img.clone())
        c.close()
        torch.cuda.synchronize()
        free = torch.cuda.mem_get_info()[0]
        if it == 0:
            free0 = free                      # after the first round: allocator pools of this process are warm
        else:
            assert abs(free - free0) <= 64 << 20, (it, free0 - free)
    assert all(torch.equal(o, outs[0]) for o in outs[1:])


def test_es_ctx_save_of_a_native_context_loads_in_a_torch_free_process(both, tmp_path):
    """es_ctx_save (C, csrc/plan.hip) writes a natively built context as a context image: es_ctx_load brings it up again - here,
    and in a child process that imports neither torch nor this package (tests/run_ctx_image.py: ctypes on the library and the HIP
    runtime only) - and every output equals the pipeline's bit for bit.  Checkpoints -> es_load_weights -> es_ctx_save ->
    es_ctx_load: no Python model code anywhere on that route."""
    import os
    import subprocess
    import sys
    import numpy as np
    pipe, eng, nat, ws, ucfg, vcfg, T = both
    lat, pe, ne, _, imgs, noise = _inputs(ucfg, vcfg, 71)
    gs = 6.5
    kw = dict(prompt_embeds=pe, negative_prompt_embeds=ne, image=imgs, latents=lat, guidance_scale=gs, num_inference_steps=T,
              cond_noise=noise)
    want_lat = pipe(output_type="latent", **kw).images.float().cpu()
    want_img = pipe(output_type="pt", **kw).images.float().cpu()
    lib = L.load()
    path = str(tmp_path / "native.esctx")
    L.check(lib.es_ctx_save(nat.ctx, path.encode()), "es_ctx_save")
    arena = lib.es_ctx_arena_bytes(nat.ctx)
    assert 1 << 20 < os.path.getsize(path) < arena                 # weights and tables travel, activations / slabs / slots do not
    assert lib.es_ctx_save(eng.ctx, str(tmp_path / "no.esctx").encode()) != 0 and b"does not own" in lib.es_last_error()
    # in this process
    ctx2 = C.c_void_p()
    L.check(lib.es_ctx_load(path.encode(), 0, C.byref(ctx2)), "es_ctx_load")
    try:
        ims = [im.to(DEV) for im in imgs]
        nz = [None if z is None else z.to(DEV) for z in noise]
        ip = (C.c_void_p * 6)(*[im.data_ptr() for im in ims])
        npz = (C.c_void_p * 6)(*[None if z is None else z.data_ptr() for z in nz])
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        x = lat.permute(0, 2, 3, 1).contiguous().to(DEV)
        ehs = torch.cat([ne, pe]).to(DEV, torch.float16).contiguous()
        ts = pipe.scheduler.set_timesteps(T).tolist()
        tsa = (C.c_float * T)(*[float(t) for t in ts])
        out = torch.empty_like(want_img, device=DEV)
        L.check(lib.es_prepare_conds(ctx2, ip, npz, stream), "es_prepare_conds")
        L.check(lib.es_denoise_loop(ctx2, C.c_void_p(x.data_ptr()), C.c_void_p(ehs.data_ptr()), gs, tsa, T, stream), "es_denoise_loop")
        L.check(lib.es_vae_decode(ctx2, C.c_void_p(x.data_ptr()), C.c_void_p(out.data_ptr()), stream), "es_vae_decode")
        torch.cuda.synchronize()
        assert torch.equal(x.permute(0, 3, 1, 2).cpu(), want_lat) and torch.equal(out.cpu(), want_img)
        # a loaded context can be saved again: same bytes
        path2 = str(tmp_path / "again.esctx")
        L.check(lib.es_ctx_save(ctx2, path2.encode()), "es_ctx_save")
        assert open(path, "rb").read() == open(path2, "rb").read()
    finally:
        lib.es_ctx_destroy(ctx2)
    # in a child process without torch
    arrs = dict(n_conds=np.int64(6), latents=lat.permute(0, 2, 3, 1).contiguous().numpy(), ehs=torch.cat([ne, pe]).half().numpy(),
                guidance_scale=np.float32(gs), timesteps=pipe.scheduler.set_timesteps(T).float().numpy())
    for i, (im, z) in enumerate(zip(imgs, noise)):
        arrs[f"img{i}"] = im.numpy()
        if z is not None:
            arrs[f"noise{i}"] = z.numpy()
    np.savez(str(tmp_path / "in.npz"), **arrs)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "run_ctx_image.py"), path, str(tmp_path / "in.npz"), str(tmp_path / "out.npz")],
                       cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    o = np.load(str(tmp_path / "out.npz"))
    assert torch.equal(torch.from_numpy(o["latents"]).permute(0, 3, 1, 2), want_lat)
    assert torch.equal(torch.from_numpy(o["image"]), want_img)


def test_a_saved_unipc_context_is_still_a_unipc_context(both, tmp_path):
    """es_ctx_save stores the scheduler es_denoise_loop applies (image format 3; ADVICE r3: a UniPC context used to come back as
    a DDIM one, silently): the reloaded context's loop equals the unsaved UniPC context's bit for bit, and differs from DDIM.  A
    failing save leaves no file under the final name."""
    import os
    from edgestyle_amd.schedulers import UniPCMultistepScheduler
    from edgestyle_amd.models import _as_nhwc
    pipe, eng, nat, ws, ucfg, vcfg, T = both
    lat, pe, ne, conds, _, _ = _inputs(ucfg, vcfg, 83)
    gs = 5.0
    lib = L.load()
    ts = UniPCMultistepScheduler.from_config(pipe.scheduler.config).set_timesteps(T).tolist()
    cd = [_as_nhwc(c.repeat(2, 1, 1, 1), torch.float16, DEV).contiguous() for c in conds]
    ehs = torch.cat([ne, pe]).to(DEV, torch.float16).contiguous()
    x = lat.permute(0, 2, 3, 1).contiguous().to(DEV)

    def fill_conds(ctx):
        for i in range(6):
            nb = C.c_size_t(0)
            p = lib.es_ctx_buffer(ctx, L.BUF_COND0 + i, C.byref(nb))
            L.check(lib.es_memcpy(C.c_void_p(p), C.c_void_p(cd[i].data_ptr()), nb.value, None), "es_memcpy")
    fill_conds(nat.ctx)
    torch.cuda.synchronize()
    want_ddim = nat.denoise_loop(x.clone(), ehs, gs, ts).clone()
    nat.set_scheduler(L.SCHED_UNIPC)
    path = str(tmp_path / "unipc.esctx")
    try:
        want = nat.denoise_loop(x.clone(), ehs, gs, ts).clone()
        torch.cuda.synchronize()
        assert not torch.equal(want, want_ddim)
        L.check(lib.es_ctx_save(nat.ctx, path.encode()), "es_ctx_save")
    finally:
        nat.set_scheduler(L.SCHED_DDIM)
    assert not os.path.exists(path + ".tmp")
    bad = str(tmp_path / "missing_dir" / "x.esctx")
    assert lib.es_ctx_save(nat.ctx, bad.encode()) != 0 and not os.path.exists(bad)
    ctx2 = C.c_void_p()
    L.check(lib.es_ctx_load(path.encode(), 0, C.byref(ctx2)), "es_ctx_load")
    try:
        fill_conds(ctx2)
        y = x.clone()
        tsa = (C.c_float * T)(*[float(t) for t in ts])
        stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        L.check(lib.es_denoise_loop(ctx2, C.c_void_p(y.data_ptr()), C.c_void_p(ehs.data_ptr()), gs, tsa, T, stream), "es_denoise_loop")
        torch.cuda.synchronize()
        assert torch.equal(y, want)
    finally:
        lib.es_ctx_destroy(ctx2)


def test_the_profiling_recipe_runs_the_native_loop_under_rocprofv3(tmp_path):
    """Round-3's host SIGSEGV under rocprofv3 sits in hipGraphLaunch: the HIP runtime replays a graph's
    pre-built AQL packets ("graph packet capture") and rocprofiler-sdk's queue interception faults on them (SIGSEGV in a memcpy
    below hipGraphLaunch, or a malformed AQL packet and a hung finalisation) - at the batch-8 size of the native leg, never
    outside the profiler, and not with DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 (profiles/r04_rocprof_graph_fault.txt).  tools/prof_bench.sh
    and every tools/collect_*.sh set that variable; this test holds the recipe together on a small context: per-plan graphs and
    the whole-loop graph of an es_load_weights context under `rocprofv3 --kernel-trace`.
    This context (sample_size 64, B = 2) is SMALLER than the geometry the fault was seen at: the failing size itself is covered by
    the recorded `tools/prof_bench.sh` runs of the batch-8 bench (profiles/r04_rocprof_graph_fault.txt, section 2), not by this test.
    Since round 5 the library also guards itself: under a queue-intercepting profiler WITHOUT the variable its contexts replay launch
    by launch (es_ctx_graph_hazard, tests/test_host_cpu.py)."""
    import os
    import shutil
    import subprocess
    if shutil.which("rocprofv3") is None:
        pytest.skip("rocprofv3 is not on PATH")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DEBUG_CLR_GRAPH_PACKET_CAPTURE="0", TMPDIR="/tmp")
    r = subprocess.run(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", str(tmp_path / "prof"), "-o", "run", "--",
                        "python3", os.path.join(root, "tests", "run_native_loop.py")], cwd=root, env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0 and "native loop under the profiler: ok" in r.stdout, (r.returncode, r.stdout[-2000:], r.stderr[-3000:])


def test_control_guidance_window_through_the_native_loop_equals_the_pipeline_bitwise(both):
    """es_denoise_loop outside the control-guidance window (PL:419-427) replays ES_PLAN_STEP_UNET - the UNet alone plus the
    fusion-of-zeros constants, no ControlNet pass - exactly where pipeline._Loop replays its UNet-only graph: both contexts
    (es_load_weights', the Python host's) against pipe(control_guidance_start / end), per-plan graphs, launch by launch and the
    whole loop as one graph (its step pattern is part of the graph's key)."""
    from edgestyle_amd.models import _as_nhwc
    pipe, eng, nat, ws, ucfg, vcfg, T = both
    lat, pe, ne, conds, _, _ = _inputs(ucfg, vcfg, 91)
    gs = 5.5
    lib = L.load()
    assert lib.es_ctx_plan_size(nat.ctx, L.PLAN_STEP_UNET) > 50 and lib.es_ctx_plan_size(nat.ctx, L.PLAN_STEP_UNET) < lib.es_ctx_plan_size(nat.ctx, L.PLAN_STEP)
    assert diff_plans(lib, eng.ctx, nat.ctx, L.PLAN_STEP_UNET) is None
    cd = [_as_nhwc(c.repeat(2, 1, 1, 1), torch.float16, DEV).contiguous() for c in conds]
    ehs = torch.cat([ne, pe]).to(DEV, torch.float16).contiguous()
    x = lat.permute(0, 2, 3, 1).contiguous().to(DEV)
    ts = pipe.scheduler.set_timesteps(T).tolist()
    for i in range(6):
        nb = C.c_size_t(0)
        p = lib.es_ctx_buffer(nat.ctx, L.BUF_COND0 + i, C.byref(nb))
        L.check(lib.es_memcpy(C.c_void_p(p), C.c_void_p(cd[i].data_ptr()), nb.value, None), "es_memcpy")
    eng.set_conds(cd)
    torch.cuda.synchronize()
    kw = dict(prompt_embeds=pe, negative_prompt_embeds=ne, image=conds, latents=lat, guidance_scale=gs, num_inference_steps=T,
              output_type="latent")
    want_open = pipe(**kw).images.clone()
    try:
        for start, end in ((0.0, 0.5), (0.45, 1.0)):
            want = pipe(control_guidance_start=start, control_guidance_end=end, **kw).images.clone()
            assert any(pipe._last_loop.skip) and not torch.equal(want, want_open)
            for ctx in (nat, eng):
                for use_graphs in (True, False, 2, 2):
                    ctx.set_options(control_guidance_start=start, control_guidance_end=end, use_graphs=use_graphs)
                    got = ctx.denoise_loop(x.clone(), ehs, gs, ts)
                    torch.cuda.synchronize()
                    assert torch.equal(got.permute(0, 3, 1, 2), want), (type(ctx).__name__, start, end, use_graphs)
    finally:
        for ctx in (nat, eng):
            ctx.set_options(use_graphs=True)
    got = nat.denoise_loop(x.clone(), ehs, gs, ts)                     # window open again: the full step everywhere
    torch.cuda.synchronize()
    assert torch.equal(got.permute(0, 3, 1, 2), want_open)


@pytest.mark.parametrize("guidance", [True, False])
def test_guess_mode_context_equals_the_pipeline_bitwise(both, guidance):
    """es_ctx_geometry.guess_mode = 1 (CL:256-264; PL:453-459, 487-497): under CFG the ControlNets run on the conditional half
    only - their condition slots hold B rows - the 13 levels are scaled 0.1..1 log-spaced and the fused residuals go into the
    conditional half of the UNet's skips.  es_load_weights' context against NativeEngine(guess_mode=True) call by call, and both
    against pipe(guess_mode=True) bit for bit: per-plan graphs, launch by launch, one graph for the loop; RGB condition images
    through es_prepare_conds; a control-guidance window (no UNet-only plan in guess_mode: ES_PLAN_STEP with all scales 0)."""
    from edgestyle_amd.native import NativeEngine, NativeContext
    pipe, _, _, ws, ucfg, vcfg, T = both
    lat, pe, ne, conds, imgs, noise = _inputs(ucfg, vcfg, 71)
    gs = 5.0 if guidance else 1.0
    lib = L.load()
    eng = NativeEngine(pipe, batch_size=1, guidance=guidance, num_inference_steps=T, guess_mode=True)
    nat = NativeContext(ws, ucfg, vcfg, batch_size=1, guidance=guidance, num_inference_steps=T, device=0, guess_mode=True)
    nat.set_alphas_cumprod(pipe.scheduler.alphas_cumprod)
    try:
        for which in range(L.PLAN_COUNT):
            if which == L.PLAN_STEP_UNET:
                assert lib.es_ctx_plan_size(nat.ctx, which) == -1 and lib.es_ctx_plan_size(eng.ctx, which) == -1
                continue
            assert diff_plans(lib, eng.ctx, nat.ctx, which) is None
        noise1 = [None if z is None else z[:1].contiguous() for z in noise]      # one CFG half only: the ControlNets' batch is B
        kw = dict(prompt_embeds=pe, negative_prompt_embeds=ne if guidance else None, image=imgs, latents=lat, guidance_scale=gs,
                  num_inference_steps=T, cond_noise=noise1, guess_mode=True)
        want_lat = pipe(output_type="latent", **kw).images.clone()
        want_img = pipe(output_type="pt", **kw).images.clone()
        plain = pipe(output_type="latent", **dict(kw, guess_mode=False, cond_noise=noise if guidance else noise1)).images
        assert not torch.equal(plain, want_lat)
        want_win = pipe(output_type="latent", control_guidance_end=0.5, **kw).images.clone()
        assert not torch.equal(want_win, want_lat)
        ehs = (torch.cat([ne, pe]) if guidance else pe).to(DEV, torch.float16).contiguous()
        x = lat.permute(0, 2, 3, 1).contiguous().to(DEV)
        ts = pipe.scheduler.set_timesteps(T).tolist()
        for ctx in (nat, eng):
            for use_graphs in (True, False, 2):
                ctx.set_options(use_graphs=use_graphs)
                ctx.prepare_conds([im.to(DEV) for im in imgs], [None if z is None else z.to(DEV) for z in noise1])
                got = ctx.denoise_loop(x.clone(), ehs, gs, ts)
                img = ctx.vae_decode(got)
                torch.cuda.synchronize()
                assert torch.equal(got.permute(0, 3, 1, 2), want_lat), (type(ctx).__name__, use_graphs, float((got.permute(0, 3, 1, 2) - want_lat).abs().max()))
                assert torch.equal(img, want_img)
            ctx.set_options(control_guidance_end=0.5, use_graphs=True)
            got = ctx.denoise_loop(x.clone(), ehs, gs, ts)
            torch.cuda.synchronize()
            assert torch.equal(got.permute(0, 3, 1, 2), want_win), type(ctx).__name__
    finally:
        nat.close()
        eng.close()
