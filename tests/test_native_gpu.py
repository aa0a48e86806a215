"""The step-level C ABI (include/edgestyle_hip.h: es_ctx / es_plan / es_denoise_step / es_denoise_loop / es_vae_decode)
driven through ctypes with raw device pointers, against the Python host path that built the context — bit for bit —
and against the CPU oracle.  What is replaced: OnnxUNetAndControlnets.forward (export_onnx.py:43-74), the loop of
model/edgestyle_pipeline.py:435-543 and the decode of PL:552-572."""
import ctypes as C

import pytest
import torch

from edgestyle_amd import config as Cfg, lib as L
from tests.helpers import make_weights, quantize, oracle_nets, psnr

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def built():
    import dataclasses
    from edgestyle_amd.models import StepRunner, AutoencoderKL
    from edgestyle_amd.pipeline import EdgeStyleStableDiffusionControlNetPipeline
    from edgestyle_amd.native import NativeEngine
    # tiny width, but 64x64 latents: the groups tile in 128-pixel units, so the step is the grouped lockstep one
    ucfg, vcfg = dataclasses.replace(Cfg.tiny_unet(), sample_size=64), Cfg.tiny_vae()
    ws = {k: quantize(v) for k, v in make_weights(ucfg, vcfg, seed=2).items()}
    runner = StepRunner.from_state_dicts(ws, ucfg, torch.float16, DEV)
    vae = AutoencoderKL(ws["vae"], vcfg).to(DEV)
    pipe = EdgeStyleStableDiffusionControlNetPipeline(vae=vae, unet=runner.unet, controlnet=runner.controlnet).to(DEV)
    eng = NativeEngine(pipe, batch_size=1, num_inference_steps=6)
    return pipe, eng, ws, ucfg, vcfg


def _inputs(ucfg, seed):
    g = torch.Generator().manual_seed(seed)
    s, c0 = ucfg.sample_size, ucfg.block_out_channels[0]
    lat = torch.randn(1, 4, s, s, generator=g)
    pe = (torch.randn(1, 77, ucfg.cross_attention_dim, generator=g) * 0.5).half().float()
    ne = (torch.randn(1, 77, ucfg.cross_attention_dim, generator=g) * 0.5).half().float()
    conds = [(torch.randn(1, c0, s, s, generator=g) * 0.3).half().float() for _ in range(6)]
    return lat, pe, ne, conds


def test_plans_hold_the_launch_lists(built):
    pipe, eng, ws, ucfg, vcfg = built
    lib = L.load()
    n_step = lib.es_ctx_plan_size(eng.ctx, L.PLAN_STEP)
    assert n_step == eng.plan_sizes[L.PLAN_STEP] and 200 < n_step < 600
    assert lib.es_ctx_plan_size(eng.ctx, L.PLAN_PREP) > 20 and lib.es_ctx_plan_size(eng.ctx, L.PLAN_DECODE) > 50
    assert lib.es_ctx_plan_size(eng.ctx, L.PLAN_STEP_GENERIC) > n_step      # + context and time embedding


def test_es_denoise_step_raw_pointers_equals_python_step_bitwise_and_oracle(built):
    """es_denoise_step(ctx, sample, t, ehs, cond_embeds[6], scales[6], out) called through ctypes with data_ptr()s."""
    from oracle import sd15_oracle as O
    from edgestyle_amd.models import _as_nhwc, _as_nchw_view
    pipe, eng, ws, ucfg, vcfg = built
    lat, pe, ne, conds = _inputs(ucfg, 11)
    x = torch.cat([lat, lat]).half().float()
    ehs = torch.cat([ne, pe])
    conds2 = [c.repeat(2, 1, 1, 1) for c in conds]
    scales = [1.0, 0.8, 1.0, 1.0, 0.5, 1.0]
    t = 441
    ref = O.denoise_step(ws["unet"], ucfg, ws["fusion"], oracle_nets(ws, ucfg), x, t, ehs, conds2, scales)
    runner = pipe._runner
    want = runner.step_nchw(x.to(DEV), t, ehs.to(DEV), [c.to(DEV) for c in conds2], scales).clone()
    assert runner.mode == "grouped" and runner._grouped_encoder(2).groupable(8 * 8)
    # the C call: contiguous device buffers, raw pointers
    lib = L.load()
    sample = _as_nhwc(x, torch.float16, DEV, pipe.unet.engine.in_pad).contiguous()
    ehs_d = ehs.to(DEV, torch.float16).contiguous()
    cond_d = [_as_nhwc(c, torch.float16, DEV).contiguous() for c in conds2]
    out = torch.zeros((2, ucfg.sample_size, ucfg.sample_size, 4), dtype=torch.float16, device=DEV)
    ptrs = (C.c_void_p * 6)(*[c.data_ptr() for c in cond_d])
    sc = (C.c_float * 6)(*scales)
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for use_graphs in (1, 0, 1):
        eng.set_options(use_graphs=bool(use_graphs))
        out.zero_()
        rc = lib.es_denoise_step(eng.ctx, C.c_void_p(sample.data_ptr()), float(t), C.c_void_p(ehs_d.data_ptr()), ptrs, sc,
                                 C.c_void_p(out.data_ptr()), stream)
        assert rc == 0, lib.es_last_error()
        torch.cuda.synchronize()
        got = _as_nchw_view(out)
        assert torch.equal(got, want), float((got.float() - want.float()).abs().max())
    assert float((got.float().cpu() - ref).abs().max()) < 2e-2
    # a second call with another timestep / prompt: nothing of the first one sticks
    want2 = runner.step_nchw(x.to(DEV), 101, torch.flip(ehs, [0]).to(DEV), [c.to(DEV) for c in conds2], scales).clone()
    ehs2 = torch.flip(ehs, [0]).to(DEV, torch.float16).contiguous()
    assert lib.es_denoise_step(eng.ctx, C.c_void_p(sample.data_ptr()), 101.0, C.c_void_p(ehs2.data_ptr()), ptrs, sc,
                               C.c_void_p(out.data_ptr()), stream) == 0
    torch.cuda.synchronize()
    assert torch.equal(_as_nchw_view(out), want2)


def test_es_denoise_loop_and_vae_decode_equal_the_pipeline_bitwise(built):
    from oracle import sd15_oracle as O
    from edgestyle_amd.models import _as_nhwc
    pipe, eng, ws, ucfg, vcfg = built
    lat, pe, ne, conds = _inputs(ucfg, 23)
    gs, T = 5.0, eng.T
    kw = dict(prompt_embeds=pe, negative_prompt_embeds=ne, image=conds, latents=lat, guidance_scale=gs, num_inference_steps=T)
    want_lat = pipe(output_type="latent", **kw).images.clone()
    want_img = pipe(output_type="pt", **kw).images.clone()
    # native: conditions into the context's slots, then the loop and the decode with raw pointers
    eng.set_options(use_graphs=True)
    eng.set_conds([_as_nhwc(c.repeat(2, 1, 1, 1), torch.float16, DEV) for c in conds])
    ehs = torch.cat([ne, pe]).to(DEV, torch.float16).contiguous()
    x = lat.permute(0, 2, 3, 1).contiguous().to(DEV)
    for use_graphs in (True, False, 2, 2):                   # 2: preparation + all steps as ONE graph (second pass: cached)
        eng.set_options(use_graphs=use_graphs)
        got = eng.denoise_loop(x.clone(), ehs, gs)
        img = eng.vae_decode(got)
        torch.cuda.synchronize()
        assert torch.equal(got.permute(0, 3, 1, 2), want_lat), float((got.permute(0, 3, 1, 2) - want_lat).abs().max())
        assert torch.equal(img, want_img)
    # another guidance scale on the same context (the scheduler node of the step graph carries it: re-instantiated)
    eng.set_options(use_graphs=True)
    want3 = pipe(output_type="latent", **dict(kw, guidance_scale=2.5)).images
    got3 = eng.denoise_loop(x.clone(), ehs, 2.5)
    assert torch.equal(got3.permute(0, 3, 1, 2), want3)
    eng.set_options(use_graphs=2)                            # ... and of the whole-loop graph
    assert torch.equal(eng.denoise_loop(x.clone(), ehs, 2.5).permute(0, 3, 1, 2), want3)
    eng.set_options(use_graphs=True)
    # conditioning scales / control-guidance window live in the context (PL:419-427, 464-470)
    scales = [1.0, 0.5, 1.0, 1.0, 0.7, 1.0]
    want4 = pipe(output_type="latent", **dict(kw, controlnet_conditioning_scale=scales, control_guidance_end=0.5)).images
    eng.set_options(cond_scales=scales, control_guidance_end=0.5)
    got4 = eng.denoise_loop(x.clone(), ehs, gs)
    assert torch.equal(got4.permute(0, 3, 1, 2), want4)
    eng.set_options(cond_scales=[1.0] * 6)
    # and the whole thing against the oracle
    ref = O.pipeline(ws["unet"], ucfg, ws["fusion"], oracle_nets(ws, ucfg), ws["vae"], vcfg, lat, pe, ne,
                     [c.repeat(2, 1, 1, 1) for c in conds], num_inference_steps=T, guidance_scale=gs)
    assert psnr(want_img, ref) >= 40.0
    with pytest.raises(L.EdgeStyleHipError):
        eng.denoise_loop(x.clone(), ehs, gs, timesteps=[981.0, 961.0])      # built for T steps


def test_engine_survives_pipeline_calls_with_another_step_count(built):
    """The plans hold raw pointers into the static buffers of the loop the engine was built on (time-projection table, condition
    slots, text K/V projections).  That loop is private to the engine: a pipeline call with the same (batch, cfg, size) but
    ANOTHER num_inference_steps - which re-allocates the pipeline's own time table - must neither free nor overwrite anything
    the native plans read."""
    from edgestyle_amd.models import _as_nhwc
    pipe, eng, ws, ucfg, vcfg = built
    lat, pe, ne, conds = _inputs(ucfg, 57)
    gs, T = 5.0, eng.T
    kw = dict(prompt_embeds=pe, negative_prompt_embeds=ne, image=conds, latents=lat, guidance_scale=gs)
    want = pipe(output_type="latent", num_inference_steps=T, **kw).images.clone()
    assert all(lp is not eng.loop for lp in pipe._loops.values())
    for other_T in (T + 5, 3):                               # grows, then shrinks the pipeline's tables
        pipe(output_type="latent", num_inference_steps=other_T, **kw)
        junk = [torch.randn(1 << 20, device=DEV) for _ in range(8)]      # anything freed would be handed out again here
        eng.set_conds([_as_nhwc(c.repeat(2, 1, 1, 1), torch.float16, DEV) for c in conds])
        ehs = torch.cat([ne, pe]).to(DEV, torch.float16).contiguous()
        for use_graphs in (True, False):
            eng.set_options(use_graphs=use_graphs)
            got = eng.denoise_loop(lat.permute(0, 2, 3, 1).contiguous().to(DEV), ehs, gs)
            torch.cuda.synchronize()
            assert torch.equal(got.permute(0, 3, 1, 2), want), (other_T, use_graphs)
        del junk
    eng.set_options(use_graphs=True)


@pytest.mark.parametrize("guidance", [True, False])
def test_native_context_for_a_single_controlnet_and_without_cfg(built, guidance):
    """es_ctx with n_conds = 1 (BASELINE configs[0]: one plain ControlNetModel) and with CFG off (N = B): loop and decode
    equal the pipeline bit for bit."""
    from edgestyle_amd.models import _as_nhwc
    from edgestyle_amd.pipeline import StableDiffusionControlNetPipeline
    from edgestyle_amd.native import NativeEngine
    pipe, _eng, ws, ucfg, vcfg = built
    pose = pipe.controlnet.nets[1]
    p1 = StableDiffusionControlNetPipeline(vae=pipe.vae, unet=pipe.unet, controlnet=pose).to(DEV)
    eng = NativeEngine(p1, batch_size=2, guidance=guidance, num_inference_steps=4)
    try:
        g = torch.Generator().manual_seed(5)
        s, c0 = ucfg.sample_size, ucfg.block_out_channels[0]
        lat = torch.randn(2, 4, s, s, generator=g)
        pe = (torch.randn(2, 77, ucfg.cross_attention_dim, generator=g) * 0.5).half().float()
        ne = (torch.randn(2, 77, ucfg.cross_attention_dim, generator=g) * 0.5).half().float()
        cond = (torch.randn(1, c0, s, s, generator=g) * 0.3).half().float()
        gs = 6.0 if guidance else 1.0
        kw = dict(prompt_embeds=pe, negative_prompt_embeds=ne if guidance else None, image=cond, latents=lat,
                  guidance_scale=gs, num_inference_steps=4, controlnet_conditioning_scale=0.7)
        want_lat = p1(output_type="latent", **kw).images.clone()
        want_img = p1(output_type="pt", **kw).images.clone()
        N = 4 if guidance else 2
        eng.set_options(cond_scales=[0.7])
        eng.set_conds([_as_nhwc(cond.repeat(N, 1, 1, 1), torch.float16, DEV)])
        ehs = (torch.cat([ne, pe]) if guidance else pe).to(DEV, torch.float16).contiguous()
        got = eng.denoise_loop(lat.permute(0, 2, 3, 1).contiguous().to(DEV), ehs, gs)
        img = eng.vae_decode(got)
        assert torch.equal(got.permute(0, 3, 1, 2), want_lat), float((got.permute(0, 3, 1, 2) - want_lat).abs().max())
        assert torch.equal(img, want_img), float((img - want_img).abs().max())
    finally:
        eng.close()


def test_es_prepare_conds_from_rgb_images_equals_the_pipeline_bitwise(built):
    """es_prepare_conds(ctx, images[6], noise[6]) == prepare_image + the one-time conditioning embedding (PL:629-664,
    CL:28-42, 289-290): raw RGB condition images (LoRA nets: [-1,1] through the VAE encoder + latent_dist.sample() with the
    caller's noise; pose nets: [0,1] through the conv stack), then es_denoise_loop + es_vae_decode - the whole try-on behind
    device pointers - against the pipeline called with the same images and `cond_noise`, bit for bit."""
    from edgestyle_amd.native import NativeEngine
    pipe, eng0, ws, ucfg, vcfg = built
    assert L.PLAN_CONDS not in eng0.plan_sizes             # built before the LoRA nets had an autoencoder: nothing to record
    for net in pipe.controlnet.nets:
        if getattr(net.config, "uses_vae", False):
            net.set_autoencoder(pipe.vae)                  # TT:252-258 (vae= of from_pretrained)
    eng = NativeEngine(pipe, batch_size=1, num_inference_steps=6)
    assert eng.plan_sizes[L.PLAN_CONDS] > 50
    g = torch.Generator().manual_seed(77)
    s = ucfg.sample_size
    H = s * vcfg.scale
    lat = torch.randn(1, 4, s, s, generator=g)
    pe = (torch.randn(1, 77, ucfg.cross_attention_dim, generator=g) * 0.5).half().float()
    ne = (torch.randn(1, 77, ucfg.cross_attention_dim, generator=g) * 0.5).half().float()
    nets = pipe.controlnet.nets
    imgs, noise = [], []
    for net in nets:
        uses_vae = bool(getattr(net.config, "uses_vae", False))
        im = torch.rand(1, 3, H, H, generator=g)
        imgs.append(im * 2 - 1 if uses_vae else im)                       # TT:29-48
        noise.append(torch.randn(2, vcfg.latent_channels, s, s, generator=g) if uses_vae else None)
    gs, T = 5.0, eng.T
    kw = dict(prompt_embeds=pe, negative_prompt_embeds=ne, image=imgs, latents=lat, guidance_scale=gs, num_inference_steps=T,
              cond_noise=noise)
    want_lat = pipe(output_type="latent", **kw).images.clone()
    want_img = pipe(output_type="pt", **kw).images.clone()
    ehs = torch.cat([ne, pe]).to(DEV, torch.float16).contiguous()
    x = lat.permute(0, 2, 3, 1).contiguous().to(DEV)
    for use_graphs in (True, False):
        eng.set_options(use_graphs=use_graphs)
        eng.prepare_conds([im.to(DEV) for im in imgs], [None if z is None else z.to(DEV) for z in noise])
        got = eng.denoise_loop(x.clone(), ehs, gs)
        img = eng.vae_decode(got)
        torch.cuda.synchronize()
        assert torch.equal(got.permute(0, 3, 1, 2), want_lat), float((got.permute(0, 3, 1, 2) - want_lat).abs().max())
        assert torch.equal(img, want_img)
    # a VAE-conditioned net without its sampling noise is refused (the library has no RNG)
    with pytest.raises(L.EdgeStyleHipError):
        eng.prepare_conds([im.to(DEV) for im in imgs], None)
    eng.close()


def test_context_image_saved_here_runs_in_a_process_without_torch(built, tmp_path):
    """NativeEngine.save(path) -> es_ctx_load(path) in a child process that imports neither torch nor this package
    (tests/run_ctx_image.py: ctypes on the library and the HIP runtime only), and in a compiled C++ host
    (examples/tryon_host.cpp): RGB condition images -> es_prepare_conds -> es_denoise_loop -> es_vae_decode there equals the
    pipeline here, bit for bit.  The image carries the packed weights,
    the static buffers and the relocated launch lists."""
    import os
    import subprocess
    import sys
    import numpy as np
    from edgestyle_amd.native import NativeEngine
    pipe, eng0, ws, ucfg, vcfg = built
    for net in pipe.controlnet.nets:
        if getattr(net.config, "uses_vae", False):
            net.set_autoencoder(pipe.vae)
    T, gs = 4, 6.0
    eng = NativeEngine(pipe, batch_size=1, num_inference_steps=T)
    try:
        g = torch.Generator().manual_seed(91)
        s = ucfg.sample_size
        H = s * vcfg.scale
        lat = torch.randn(1, 4, s, s, generator=g)
        pe = (torch.randn(1, 77, ucfg.cross_attention_dim, generator=g) * 0.5).half().float()
        ne = (torch.randn(1, 77, ucfg.cross_attention_dim, generator=g) * 0.5).half().float()
        imgs, noise = [], []
        for net in pipe.controlnet.nets:
            uses_vae = bool(getattr(net.config, "uses_vae", False))
            im = torch.rand(1, 3, H, H, generator=g)
            imgs.append(im * 2 - 1 if uses_vae else im)
            noise.append(torch.randn(2, vcfg.latent_channels, s, s, generator=g) if uses_vae else None)
        kw = dict(prompt_embeds=pe, negative_prompt_embeds=ne, image=imgs, latents=lat, guidance_scale=gs, num_inference_steps=T,
                  cond_noise=noise)
        want_lat = pipe(output_type="latent", **kw).images.float().cpu()
        want_img = pipe(output_type="pt", **kw).images.float().cpu()
        path = str(tmp_path / "ctx.esctx")
        info = eng.save(path)
        # typed relocation (es_plan_pointer_fields); only memory the plans never write travels as data
        assert info["relocations"][L.PLAN_STEP] > 1000 and info["data_bytes"] < os.path.getsize(path) < info["arena_bytes"]
        arrs = dict(n_conds=np.int64(len(imgs)), latents=lat.permute(0, 2, 3, 1).contiguous().numpy(),
                    ehs=torch.cat([ne, pe]).half().numpy(), guidance_scale=np.float32(gs),
                    timesteps=pipe.scheduler.set_timesteps(T).float().numpy())
        for i, (im, nz) in enumerate(zip(imgs, noise)):
            arrs[f"img{i}"] = im.numpy()
            if nz is not None:
                arrs[f"noise{i}"] = nz.numpy()
        np.savez(str(tmp_path / "in.npz"), **arrs)
        # the same inputs as one flat file for the C++ host (examples/tryon_host.cpp documents the layout)
        with open(str(tmp_path / "in.bin"), "wb") as f:
            has_noise = [int(z is not None) for z in noise]
            f.write(np.array([1, s, s, 4, ucfg.cross_attention_dim, len(imgs), T] + has_noise, dtype=np.int32).tobytes())
            f.write(np.float32(gs).tobytes())
            f.write(arrs["timesteps"].astype(np.float32).tobytes())
            f.write(arrs["latents"].astype(np.float32).tobytes())
            f.write(arrs["ehs"].astype(np.float16).tobytes())
            for im, nz in zip(imgs, noise):
                f.write(im.numpy().astype(np.float32).tobytes())
                if nz is not None:
                    f.write(nz.numpy().astype(np.float32).tobytes())
    finally:
        eng.close()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "run_ctx_image.py"), path, str(tmp_path / "in.npz"),
                        str(tmp_path / "out.npz")], cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    out = np.load(str(tmp_path / "out.npz"))
    got_lat = torch.from_numpy(out["latents"]).permute(0, 3, 1, 2)
    assert torch.equal(got_lat, want_lat), float((got_lat - want_lat).abs().max())
    assert torch.equal(torch.from_numpy(out["image"]), want_img)
    # and a host with no Python at all: examples/tryon_host.cpp compiled against include/edgestyle_hip.h
    import shutil
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = str(tmp_path / "tryon_host")
    libdir = os.path.join(root, "edgestyle_amd", "lib")
    c = subprocess.run([hipcc, "-O1", "-I" + os.path.join(root, "include"), os.path.join(root, "examples", "tryon_host.cpp"),
                        "-L" + libdir, "-ledgestyle_hip", "-Wl,-rpath," + libdir, "-o", exe], capture_output=True, text=True, timeout=600)
    assert c.returncode == 0, c.stderr[-2000:]
    r = subprocess.run([exe, path, str(tmp_path / "in.bin"), str(tmp_path / "out.bin")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    raw = np.fromfile(str(tmp_path / "out.bin"), dtype=np.float32)
    nl = s * s * 4
    got_lat = torch.from_numpy(raw[:nl].reshape(1, s, s, 4).copy()).permute(0, 3, 1, 2)
    got_img = torch.from_numpy(raw[nl:].reshape(1, 3, H, H).copy())
    assert torch.equal(got_lat, want_lat) and torch.equal(got_img, want_img)
