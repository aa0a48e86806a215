"""End-to-end parity of the HIP pipeline against the CPU oracle pipeline (tiny config), PSNR >= 40 dB on the decoded
image (north_star tolerance), graph replay == eager, drop-in call-surface checks."""
import math
import os

import pytest
import torch

from edgestyle_amd import config as C
from tests.helpers import make_weights, quantize, rel_err, oracle_nets

pytestmark = pytest.mark.gpu
DEV = "cuda"


def psnr(a, b, peak=1.0):
    mse = float(((a.float().cpu() - b.float().cpu()) ** 2).mean())
    return 10 * math.log10(peak * peak / max(mse, 1e-20))


@pytest.fixture(scope="module")
def built():
    from edgestyle_amd.models import (UNet2DConditionModel, ControlNetModel, ControlLoRAModel, AutoencoderKL,
                                      EdgeStyleMultiControlNetModel)
    from edgestyle_amd.pipeline import StableDiffusionControlNetPipeline
    ucfg, vcfg = C.tiny_unet(), C.tiny_vae()
    ws = {k: quantize(v) for k, v in make_weights(ucfg, vcfg).items()}
    unet = UNet2DConditionModel(ws["unet"], ucfg, torch.float16)
    vae = AutoencoderKL(ws["vae"], vcfg)
    pose = ControlNetModel(ws["openpose"], ucfg, torch.float16)
    l0 = ControlLoRAModel(ws["lora0"], ucfg, lora_linear_rank=4, uses_vae=True)
    l1 = ControlLoRAModel(ws["lora1"], ucfg, lora_linear_rank=4, uses_vae=True)
    for n in (l0, l1):
        n.set_autoencoder(vae)
        n.tie_weights(unet)
    mc = EdgeStyleMultiControlNetModel([l0, pose, l1, pose, l1, pose])
    mc.load_state_dict(ws["fusion"])
    pipe = StableDiffusionControlNetPipeline.from_pretrained(None, vae=vae, unet=unet, controlnet=mc,
                                                             safety_checker=None, torch_dtype=torch.float16)
    pipe = pipe.to(DEV)
    return pipe, ws, ucfg, vcfg


def _inputs(ucfg, B, seed=42):
    g = torch.Generator().manual_seed(seed)
    s, c0 = ucfg.sample_size, ucfg.block_out_channels[0]
    lat = torch.randn(B, 4, s, s, generator=g)
    pe = (torch.randn(B, 77, ucfg.cross_attention_dim, generator=g) * 0.5).half().float()
    ne = (torch.randn(B, 77, ucfg.cross_attention_dim, generator=g) * 0.5).half().float()
    conds = [(torch.randn(1, c0, s, s, generator=g) * 0.3).half().float() for _ in range(6)]
    return lat, pe, ne, conds


@pytest.mark.parametrize("B,steps,gs", [(1, 4, 7.5), (2, 10, 3.5), (1, 3, 1.0)])
def test_pipeline_vs_oracle_psnr(built, B, steps, gs):
    from oracle import sd15_oracle as O
    pipe, ws, ucfg, vcfg = built
    lat, pe, ne, conds = _inputs(ucfg, B)
    N = 2 * B if gs > 1 else B
    oconds = [c.repeat(N, 1, 1, 1) for c in conds]
    ref = O.pipeline(ws["unet"], ucfg, ws["fusion"], oracle_nets(ws, ucfg), ws["vae"], vcfg, lat, pe, ne, oconds,
                     num_inference_steps=steps, guidance_scale=gs)
    out = pipe(prompt_embeds=pe, negative_prompt_embeds=ne, image=conds, latents=lat, guidance_scale=gs,
               num_inference_steps=steps, output_type="pt").images
    assert out.shape == ref.shape
    p = psnr(out, ref)
    assert p >= 40.0, p
    # second call reuses the captured graph and static buffers; eager must agree bit for bit with replay
    out2 = pipe(prompt_embeds=pe, negative_prompt_embeds=ne, image=conds, latents=lat, guidance_scale=gs,
                num_inference_steps=steps, output_type="pt").images
    assert torch.equal(out, out2)
    # ... and that second call captured all steps as ONE graph, which the third call only replays
    assert pipe._last_loop.loop_graph is not None
    out4 = pipe(prompt_embeds=pe, negative_prompt_embeds=ne, image=conds, latents=lat, guidance_scale=gs,
                num_inference_steps=steps, output_type="pt").images
    assert torch.equal(out, out4)
    pipe.use_graph = False
    try:
        out3 = pipe(prompt_embeds=pe, negative_prompt_embeds=ne, image=conds, latents=lat, guidance_scale=gs,
                    num_inference_steps=steps, output_type="pt").images
    finally:
        pipe.use_graph = True
    assert torch.equal(out, out3)


def test_pipeline_raw_images_guidance_window_and_outputs(built):
    """[1,3,H,W] condition images go through VAE-encode / openpose conv stack once (PL:660-662); control guidance
    window (PL:419-427) and scale list honoured; latent / np / pil outputs."""
    from oracle import sd15_oracle as O
    pipe, ws, ucfg, vcfg = built
    g = torch.Generator().manual_seed(7)
    s = ucfg.sample_size
    res = s * 8
    lat, pe, ne, _ = _inputs(ucfg, 1, seed=5)
    imgs = [(torch.rand(1, 3, res, res, generator=g) * (2 if i % 2 == 0 else 1) - (1 if i % 2 == 0 else 0)).half().float()
            for i in range(6)]
    noise = [torch.randn(2, 4, s, s, generator=g) if i % 2 == 0 else None for i in range(6)]
    nets = oracle_nets(ws, ucfg)
    oconds = []
    for i in range(6):
        im2 = torch.cat([imgs[i]] * 2)
        if i % 2 == 0:
            oconds.append(O.vae_cond_embedding(nets[i][0], ws["vae"], vcfg, im2, noise[i]))
        else:
            oconds.append(O.cond_embedding(ws["openpose"], ucfg, im2))
    scales = [1.0, 0.5, 1.0, 1.0, 0.7, 1.0]
    ref = O.pipeline(ws["unet"], ucfg, ws["fusion"], nets, ws["vae"], vcfg, lat, pe, ne, oconds, num_inference_steps=5,
                     guidance_scale=5.0, scales=scales, control_guidance_start=0.0, control_guidance_end=0.6,
                     decode=False)
    kw = dict(prompt_embeds=pe, negative_prompt_embeds=ne, image=imgs, latents=lat, guidance_scale=5.0,
              num_inference_steps=5, controlnet_conditioning_scale=scales, control_guidance_end=0.6, cond_noise=noise)
    out = pipe(output_type="latent", **kw).images
    assert rel_err(out, ref) < 3e-2
    arr = pipe(output_type="np", **kw).images
    assert arr.shape == (1, res, res, 3) and arr.min() >= 0 and arr.max() <= 1
    pil = pipe(output_type="pil", **kw).images
    assert pil[0].size == (res, res)


def test_multicontrolnet_and_unet_call_surface(built):
    """EdgeStyleMultiControlNetModel.__call__ (MC:116-171) + UNet2DConditionModel.__call__ chained the way
    OnnxUNetAndControlnets.forward does (export_onnx.py:43-74), NCHW in / NCHW out."""
    from oracle import sd15_oracle as O
    pipe, ws, ucfg, vcfg = built
    g = torch.Generator().manual_seed(11)
    s, c0 = ucfg.sample_size, ucfg.block_out_channels[0]
    x = torch.randn(2, 4, s, s, generator=g).half().float()
    ehs = (torch.randn(2, 77, ucfg.cross_attention_dim, generator=g) * 0.5).half().float()
    conds = [(torch.randn(2, c0, s, s, generator=g) * 0.3).half().float() for _ in range(6)]
    scales = [1.0] * 6
    down, mid = pipe.controlnet(x.to(DEV), 261, encoder_hidden_states=ehs.to(DEV),
                                controlnet_cond=[c.to(DEV) for c in conds], conditioning_scale=scales,
                                guess_mode=False, return_dict=False)
    assert len(down) == 12 and [tuple(d.shape) for d in down] == [(2, c, sz, sz) for c, sz in ucfg.residual_table()[:-1]]
    rd, rm = O.multicontrolnet_forward(ws["fusion"], oracle_nets(ws, ucfg), x, 261, ehs, conds, scales)
    for a, b in zip(down + [mid], rd + [rm]):
        assert rel_err(a, b) < 2e-2
    noise = pipe.unet(x.to(DEV), 261, encoder_hidden_states=ehs.to(DEV), down_block_additional_residuals=down,
                      mid_block_additional_residual=mid, return_dict=False)[0]
    ref = O.unet_forward(ws["unet"], ucfg, x, 261, ehs, rd, rm)
    assert tuple(noise.shape) == (2, 4, s, s) and rel_err(noise, ref) < 2e-2


@pytest.mark.parametrize("gs", [5.0, 1.0])
def test_pipeline_guess_mode_vs_oracle(built, gs):
    """guess_mode (CL:256-264 log-spaced level scales; under CFG the ControlNets run on the conditional half only and
    the unconditional half gets zero residuals, PL:453-459, 487-497)."""
    from oracle import sd15_oracle as O
    pipe, ws, ucfg, vcfg = built
    B, steps = 1, 3
    lat, pe, ne, conds = _inputs(ucfg, B, seed=7)
    ref = O.pipeline(ws["unet"], ucfg, ws["fusion"], oracle_nets(ws, ucfg), ws["vae"], vcfg, lat, pe, ne,
                     [c.repeat(B, 1, 1, 1) for c in conds], num_inference_steps=steps, guidance_scale=gs, guess_mode=True)
    out = pipe(prompt_embeds=pe, negative_prompt_embeds=ne, image=conds, latents=lat, guidance_scale=gs,
               num_inference_steps=steps, output_type="pt", guess_mode=True).images
    assert psnr(out, ref) >= 40.0
    plain = pipe(prompt_embeds=pe, negative_prompt_embeds=ne, image=conds, latents=lat, guidance_scale=gs,
                 num_inference_steps=steps, output_type="pt").images
    assert not torch.equal(out, plain)                      # the mode does change the result
    # the multi-ControlNet call surface hands guess_mode to every net (MC:136-149)
    g = torch.Generator().manual_seed(3)
    s = ucfg.sample_size
    x = torch.randn(2, 4, s, s, generator=g).half().float()
    ehs = torch.cat([ne, pe]).repeat(1, 1, 1)[:2]
    cc = [c.repeat(2, 1, 1, 1) for c in conds]
    down, mid = pipe.controlnet(x, 500, ehs, cc, [1.0] * 6, guess_mode=True, return_dict=False)
    rd, rm = O.multicontrolnet_forward(ws["fusion"], oracle_nets(ws, ucfg), x, 500, ehs, cc, [1.0] * 6, guess_mode=True)
    assert (mid.float().cpu() - rm).abs().max() < 2e-2 * max(1.0, float(rm.abs().max()))
    assert (down[0].float().cpu() - rd[0]).abs().max() < 2e-2 * max(1.0, float(rd[0].abs().max()))


def test_errors_match_reference_behaviour(built):
    pipe, ws, ucfg, vcfg = built
    lat, pe, ne, conds = _inputs(ucfg, 1)
    with pytest.raises(ValueError):
        pipe(prompt_embeds=pe, negative_prompt_embeds=ne, image=conds[:5], latents=lat)       # 5 images for 6 nets
    with pytest.raises(ValueError):
        pipe(image=conds, latents=lat)                                                        # no prompt at all
    with pytest.raises(ValueError):
        pipe(prompt_embeds=pe, negative_prompt_embeds=ne, image=conds, latents=lat, control_guidance_start=0.8,
             control_guidance_end=0.2)
    # what the reference's signature (PL:92-120) offers and this path does not implement is refused, never ignored
    base = dict(prompt_embeds=pe, negative_prompt_embeds=ne, image=conds, latents=lat, num_inference_steps=2)
    with pytest.raises(TypeError):
        pipe(callback_steps=1, **base)
    for kw in (dict(clip_skip=1), dict(cross_attention_kwargs={"scale": 0.5}), dict(ip_adapter_image=conds[0]),
               dict(callback_on_step_end_tensor_inputs=["latents", "prompt_embeds"]), dict(eta=0.5), dict(timesteps=[10, 5])):
        with pytest.raises(NotImplementedError):
            pipe(**base, **kw)
    px = ucfg.sample_size * vcfg.scale
    with pytest.raises(ValueError):
        pipe(height=px + 8 * vcfg.scale, **base)                 # another size than the condition images define (PL:377)
    out = pipe(height=px, width=px, output_type="latent", **base).images      # the matching size is accepted
    assert out.shape[-1] == ucfg.sample_size


def test_pipeline_bf16_non_default_size_config5_analogue():
    """BASELINE config 5 analogue at tiny width: bf16, latent size != the reference's hard-wired 64 (MC:73-102 is
    generalised from the config).  Parity target is the oracle (the reference itself raises at this size).
    bf16 keeps 8 mantissa bits, so the bar is PSNR >= 30 dB here (fp16: >= 40 dB above)."""
    import dataclasses
    from oracle import sd15_oracle as O
    from edgestyle_amd.models import StepRunner
    from edgestyle_amd.pipeline import EdgeStyleStableDiffusionControlNetPipeline
    from edgestyle_amd.models import AutoencoderKL
    ucfg = dataclasses.replace(C.tiny_unet(), sample_size=24)
    vcfg = C.tiny_vae()
    ws = {k: quantize(v, torch.bfloat16) for k, v in make_weights(ucfg, vcfg, seed=3).items()}
    runner = StepRunner.from_state_dicts(ws, ucfg, torch.bfloat16, DEV)
    vae = AutoencoderKL(ws["vae"], vcfg, torch.bfloat16).to(DEV)
    pipe = EdgeStyleStableDiffusionControlNetPipeline(vae=vae, unet=runner.unet, controlnet=runner.controlnet).to(DEV)
    g = torch.Generator().manual_seed(8)
    s, c0 = 24, ucfg.block_out_channels[0]
    lat = torch.randn(2, 4, s, s, generator=g)
    pe = (torch.randn(2, 77, ucfg.cross_attention_dim, generator=g) * 0.5).bfloat16().float()
    ne = (torch.randn(2, 77, ucfg.cross_attention_dim, generator=g) * 0.5).bfloat16().float()
    conds = [(torch.randn(1, c0, s, s, generator=g) * 0.3).bfloat16().float() for _ in range(6)]
    ref = O.pipeline(ws["unet"], ucfg, ws["fusion"], oracle_nets(ws, ucfg), ws["vae"], vcfg, lat, pe, ne,
                     [c.repeat(4, 1, 1, 1) for c in conds], num_inference_steps=3, guidance_scale=5.0)
    out = pipe(prompt_embeds=pe, negative_prompt_embeds=ne, image=conds, latents=lat, guidance_scale=5.0,
               num_inference_steps=3, output_type="pt").images
    assert out.shape == (2, 3, 192, 192)
    p = psnr(out, ref)
    print("bf16 PSNR", p)
    assert p >= 30.0, p


def test_pipeline_unipc_scheduler_like_the_reference_callers(built):
    """`pipeline.scheduler = UniPCMultistepScheduler.from_config(pipeline.scheduler.config)` (TT:273): the fused UniPC
    step kernel vs the oracle's UniPC restatement, same loop otherwise."""
    from oracle import sd15_oracle as O
    from edgestyle_amd.schedulers import UniPCMultistepScheduler, DDIMScheduler
    pipe, ws, ucfg, vcfg = built
    lat, pe, ne, conds = _inputs(ucfg, 1, seed=21)
    steps, gs = 8, 4.0
    nets = oracle_nets(ws, ucfg)
    oconds = [c.repeat(2, 1, 1, 1) for c in conds]
    sch = O.UniPC()
    ts = sch.set_timesteps(steps)
    x = lat.clone()
    ehs = torch.cat([ne, pe])
    for t in ts.tolist():
        eps = O.denoise_step(ws["unet"], ucfg, ws["fusion"], nets, torch.cat([x] * 2), t, ehs, oconds, [1.0] * 6)
        e_u, e_t = eps.chunk(2)
        x = sch.step(e_u + gs * (e_t - e_u), t, x)
    old = pipe.scheduler
    pipe.scheduler = UniPCMultistepScheduler.from_config(getattr(old, "config", None))
    try:
        out = pipe(prompt_embeds=pe, negative_prompt_embeds=ne, image=conds, latents=lat, guidance_scale=gs,
                   num_inference_steps=steps, output_type="latent").images
        out2 = pipe(prompt_embeds=pe, negative_prompt_embeds=ne, image=conds, latents=lat, guidance_scale=gs,
                    num_inference_steps=steps, output_type="latent").images
    finally:
        pipe.scheduler = old
    assert torch.equal(out, out2)
    assert rel_err(out, x) < 3e-2, rel_err(out, x)


def test_cli_counterpart_of_the_reference_test_script(tmp_path):
    """`python -m edgestyle_amd.cli` with the flags of test_inference.sh on seeded random-init model directories in
    the reference's on-disk layout and synthetic subject/head/openpose/clothes JPEGs -> 3x3 grid like TT:360-365."""
    import numpy as np
    from PIL import Image
    from edgestyle_amd import cli
    rng = np.random.RandomState(0)
    for root in ("source", "target"):
        for kind in ("subject", "agnostic", "head", "openpose", "clothes"):
            d = tmp_path / root / kind
            d.mkdir(parents=True)
            for name in ("0.jpg", "1.jpg", "2.jpg"):
                Image.fromarray(rng.randint(0, 255, (160, 144, 3), dtype=np.uint8)).save(str(d / name))
    args = cli.parse_args(["--random_init", str(tmp_path / "models"), "--tiny", "--controllora_use_vae",
                           "--mixed_precision", "fp16", "--source_path", str(tmp_path / "source"),
                           "--source_image_name", "1.jpg", "--target_path", str(tmp_path / "target"),
                           "--target_image_name", "0.jpg", "--target_path2", str(tmp_path / "target"),
                           "--target_image_name2", "2.jpg", "--result_path", str(tmp_path / "out"),
                           "--image_result_name", "result.jpg", "--num_inference_steps", "4"])
    out = cli.main(args)
    grid = Image.open(out)
    assert grid.size == (3 * 128, 3 * 128)
    assert sorted(os.listdir(tmp_path / "models" / "EdgeStyle" / "controlnet")) == [
        "controlnet_0", "controlnet_1", "diffusion_pytorch_model.safetensors"]
    x = cli.load_image(str(tmp_path / "source" / "head" / "1.jpg"), 128, True)
    assert x.shape == (1, 3, 128, 128) and float(x.min()) >= -1 and float(x.max()) <= 1


def test_try_on_service_batched_equals_individually_served(built):
    """edgestyle_amd/serve.py over the real (tiny) pipeline: three concurrent requests run as one batch of 2 + one of 1;
    each image matches the one the request gets when served alone (different tile plans per batch: PSNR, not bits)."""
    from edgestyle_amd.serve import TryOnService, TryOnRequest
    pipe, ws, ucfg, vcfg = built
    s, c0 = ucfg.sample_size, ucfg.block_out_channels[0]

    def req(seed):
        g = torch.Generator().manual_seed(500 + seed)
        return TryOnRequest([(torch.randn(1, c0, s, s, generator=g) * 0.3).half().float() for _ in range(6)],
                            (torch.randn(1, 77, ucfg.cross_attention_dim, generator=g) * 0.5).half().float(),
                            (torch.randn(1, 77, ucfg.cross_attention_dim, generator=g) * 0.5).half().float(),
                            5.0, 4, seed)
    solo = TryOnService(pipe, max_batch=1, max_wait_s=0.0, vae_scale=1)
    want = [solo.submit(req(i)).result(timeout=300) for i in range(3)]
    solo.shutdown()
    svc = TryOnService(pipe, max_batch=2, max_wait_s=1.0, batch_sizes=(1, 2), vae_scale=1)
    futs = [svc.submit(req(i)) for i in range(3)]
    got = [f.result(timeout=300) for f in futs]
    svc.shutdown()
    assert svc.stats["calls"] == 2 and svc.stats["images"] == 3
    for a, b in zip(got, want):
        assert a.shape == b.shape and psnr(a, b) >= 45.0


def test_pipeline_odd_batch_matches_single_images(built):
    """B = 3 (not a captured serving size, not a multiple of anything): every image of the batch equals the image the same
    inputs give alone (tile plans differ with the batch: PSNR, not bits)."""
    pipe, ws, ucfg, vcfg = built
    lat, pe, ne, conds = _inputs(ucfg, 3, seed=7)
    conds3 = [c.repeat(3, 1, 1, 1) * torch.tensor([1.0, 0.5, -0.7])[:, None, None, None] for c in conds]
    conds3 = [c.half().float() for c in conds3]
    out = pipe(prompt_embeds=pe, negative_prompt_embeds=ne, image=conds3, latents=lat, guidance_scale=6.0,
               num_inference_steps=4, output_type="pt").images
    assert out.shape[0] == 3 and bool(torch.isfinite(out).all())
    for i in range(3):
        one = pipe(prompt_embeds=pe[i:i + 1], negative_prompt_embeds=ne[i:i + 1], image=[c[i:i + 1] for c in conds3],
                   latents=lat[i:i + 1], guidance_scale=6.0, num_inference_steps=4, output_type="pt").images
        assert psnr(out[i:i + 1], one) >= 45.0, i


# ----------------------------------------------------------------------------------------------------------------
# graph / static-buffer ownership (a captured graph replays the pointers it was captured with)
# ----------------------------------------------------------------------------------------------------------------
def _call(pipe, lat, pe, ne, conds, **kw):
    a = dict(prompt_embeds=pe, negative_prompt_embeds=ne, image=conds, latents=lat, guidance_scale=5.0,
             num_inference_steps=3, output_type="pt")
    a.update(kw)
    return pipe(**a).images


def test_guess_mode_second_call_with_other_prompt_reads_its_own_text_states(built):
    """Two guess_mode calls of the same shape with different prompt embeddings: the replayed graph must see the second
    call's text K/V projections (they are refilled in place in buffers the loop owns), i.e. equal the eager run."""
    pipe, ws, ucfg, vcfg = built
    lat, pe, ne, conds = _inputs(ucfg, 1, seed=31)
    _, pe2, ne2, _ = _inputs(ucfg, 1, seed=32)
    out1 = _call(pipe, lat, pe, ne, conds, guess_mode=True)
    out2 = _call(pipe, lat, pe2, ne2, conds, guess_mode=True)
    assert not torch.equal(out1, out2)
    pipe.use_graph = False
    try:
        eager2 = _call(pipe, lat, pe2, ne2, conds, guess_mode=True)
        eager1 = _call(pipe, lat, pe, ne, conds, guess_mode=True)
    finally:
        pipe.use_graph = True
    assert torch.equal(out2, eager2) and torch.equal(out1, eager1)


@pytest.mark.parametrize("mode", ["serial", "grouped"])
def test_batch_switch_1_2_1_replays_cached_graphs_without_recapture(built, mode):
    """B = 1, 2, 1 through one pipeline: every loop owns its buffers, so the second B = 1 call replays the graph captured
    by the first (no re-capture), and every result equals the eager run."""
    pipe, ws, ucfg, vcfg = built
    pipe._runner = None
    pipe._loops.clear()
    lat1, pe1, ne1, conds = _inputs(ucfg, 1, seed=41)
    lat2, pe2, ne2, _ = _inputs(ucfg, 2, seed=42)
    lat3, pe3, ne3, _ = _inputs(ucfg, 1, seed=43)
    first = _call(pipe, lat1, pe1, ne1, conds)
    pipe._runner.mode = mode
    pipe._loops.clear()
    a = _call(pipe, lat1, pe1, ne1, conds)
    loop1 = pipe._last_loop
    assert loop1.captures == 1
    b = _call(pipe, lat2, pe2, ne2, conds)
    c = _call(pipe, lat3, pe3, ne3, conds)
    a2 = _call(pipe, lat1, pe1, ne1, conds)
    assert pipe._last_loop is loop1 and loop1.captures == 1, "the cached B=1 graph was re-captured"
    assert torch.equal(a, a2) and torch.equal(a, first)
    pipe.use_graph = False
    try:
        assert torch.equal(b, _call(pipe, lat2, pe2, ne2, conds))
        assert torch.equal(c, _call(pipe, lat3, pe3, ne3, conds))
    finally:
        pipe.use_graph = True
        pipe._runner = None
        pipe._loops.clear()


def test_callback_latents_feed_the_next_step_and_the_decode(built):
    """PL:529-531: latents returned by callback_on_step_end replace the loop's latents — for the scheduler, for the
    next step's networks and for the VAE decode.  Checked against the oracle loop with the same intervention."""
    from oracle import sd15_oracle as O
    pipe, ws, ucfg, vcfg = built
    lat, pe, ne, conds = _inputs(ucfg, 1, seed=51)
    steps, gs = 4, 5.0
    nets = oracle_nets(ws, ucfg)
    oconds = [c.repeat(2, 1, 1, 1) for c in conds]
    sch = O.DDIM()
    ts = sch.set_timesteps(steps)
    x = lat.clone()
    ehs = torch.cat([ne, pe])
    for i, t in enumerate(ts.tolist()):
        eps = O.denoise_step(ws["unet"], ucfg, ws["fusion"], nets, torch.cat([x] * 2), t, ehs, oconds, [1.0] * 6)
        e_u, e_t = eps.chunk(2)
        x = sch.step(e_u + gs * (e_t - e_u), t, x)
        if i in (1, steps - 1):
            x = x * 0.5 + 0.3
    ref = (O.vae_decode(ws["vae"], vcfg, x / vcfg.scaling_factor) / 2 + 0.5).clamp(0, 1)

    def cb(p, i, t, kw):
        return {"latents": kw["latents"] * 0.5 + 0.3} if i in (1, steps - 1) else {}
    out = pipe(prompt_embeds=pe, negative_prompt_embeds=ne, image=conds, latents=lat, guidance_scale=gs,
               num_inference_steps=steps, output_type="pt", callback_on_step_end=cb).images
    got = psnr(out, ref)
    assert got >= 40.0
    plain = _call(pipe, lat, pe, ne, conds, num_inference_steps=steps)
    assert psnr(plain, ref) < got - 8.0                 # the intervention is visible


def test_device_generator_like_the_reference_test_script(built):
    """TT:274 hands `torch.Generator(device).manual_seed(42)` to the pipeline: accepted, and (documented) drawn on a
    host generator with the same seed, so it equals a CPU generator seeded alike."""
    pipe, ws, ucfg, vcfg = built
    _, pe, ne, conds = _inputs(ucfg, 1, seed=61)
    kw = dict(prompt_embeds=pe, negative_prompt_embeds=ne, image=conds, guidance_scale=5.0, num_inference_steps=3,
              output_type="pt")
    with torch.autocast("cuda"):                            # TT:327
        a = pipe(generator=torch.Generator(DEV).manual_seed(42), **kw).images
    b = pipe(generator=torch.Generator().manual_seed(42), **kw).images
    assert torch.equal(a, b) and bool(torch.isfinite(a).all())
    # one device generator across calls: draws advance like the original's would; re-seeding it (even to the SAME seed, the
    # usual way to reproduce a run) restarts the stream, as it does for the reference that draws from the device generator
    gen = torch.Generator(DEV).manual_seed(7)
    first = pipe(generator=gen, **kw).images
    second = pipe(generator=gen, **kw).images
    assert not torch.equal(first, second)
    gen.manual_seed(7)
    assert torch.equal(pipe(generator=gen, **kw).images, first)
    assert torch.equal(pipe(generator=gen, **kw).images, second)


def test_num_images_per_prompt_repeats_a_per_prompt_image_batch(built):
    """PL:647-653: an image batch equal to the prompt batch is repeated num_images_per_prompt times."""
    pipe, ws, ucfg, vcfg = built
    lat, pe, ne, conds = _inputs(ucfg, 2, seed=71)
    conds2 = [(c.repeat(2, 1, 1, 1) * torch.tensor([1.0, -0.5])[:, None, None, None]).half().float() for c in conds]
    lat4 = torch.cat([lat, lat + 0.1]).index_select(0, torch.tensor([0, 2, 1, 3]))
    out = pipe(prompt_embeds=pe, negative_prompt_embeds=ne, image=conds2, latents=lat4, guidance_scale=5.0,
               num_inference_steps=3, output_type="pt", num_images_per_prompt=2).images
    assert out.shape[0] == 4
    one = pipe(prompt_embeds=pe[1:2], negative_prompt_embeds=ne[1:2], image=[c[1:2] for c in conds2], latents=lat4[3:4],
               guidance_scale=5.0, num_inference_steps=3, output_type="pt").images
    assert psnr(out[3:4], one) >= 45.0


# ----------------------------------------------------------------------------------------------------------------
# one plain ControlNetModel as `controlnet` (PL:338-351) — BASELINE configs[0] at tiny width, vs the live oracle
# ----------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("guess,gs", [(False, 7.5), (False, 1.0), (True, 5.0)])
def test_single_controlnet_pipeline_vs_oracle(built, guess, gs):
    from oracle import sd15_oracle as O
    from edgestyle_amd.pipeline import StableDiffusionControlNetPipeline
    pipe, ws, ucfg, vcfg = built
    pose = pipe.controlnet.nets[1]
    p1 = StableDiffusionControlNetPipeline(vae=pipe.vae, unet=pipe.unet, controlnet=pose).to(DEV)
    g = torch.Generator().manual_seed(81)
    s = ucfg.sample_size
    lat, pe, ne, _ = _inputs(ucfg, 1, seed=82)
    img = torch.rand(1, 3, s * 8, s * 8, generator=g).half().float()
    N = 2 if (gs > 1 and not guess) else 1
    steps = 4
    ref = O.pipeline(ws["unet"], ucfg, None, [(ws["openpose"], ucfg)], ws["vae"], vcfg, lat, pe, ne,
                     [img.repeat(N, 1, 1, 1)], num_inference_steps=steps, guidance_scale=gs, scales=[0.8],
                     control_guidance_end=0.75, guess_mode=guess)
    out = p1(prompt_embeds=pe, negative_prompt_embeds=ne, image=img, latents=lat, guidance_scale=gs,
             num_inference_steps=steps, output_type="pt", controlnet_conditioning_scale=0.8, control_guidance_end=0.75,
             guess_mode=guess).images
    assert psnr(out, ref) >= 40.0
    with pytest.raises(ValueError):
        p1(prompt_embeds=pe, negative_prompt_embeds=ne, image=[img, img], latents=lat)


def test_string_prompts_through_a_random_init_clip_text_encoder(built, tmp_path):
    """`prompt=` / `negative_prompt=` (TT:328-338): tokenizer + CLIPTextModel from transformers, built offline from a
    config (random init) and a synthetic byte-level BPE vocabulary; the embeddings the pipeline derives equal the ones
    computed by hand, and the image equals the prompt_embeds= call."""
    import json
    from transformers import CLIPTextConfig, CLIPTextModel, CLIPTokenizer
    pipe, ws, ucfg, vcfg = built
    chars = [chr(c) for c in range(ord("a"), ord("z") + 1)] + [",", "."]
    vocab = {}
    for ch in chars:
        vocab[ch] = len(vocab)
    for ch in chars:
        vocab[ch + "</w>"] = len(vocab)
    vocab["<|startoftext|>"] = len(vocab)
    vocab["<|endoftext|>"] = len(vocab)
    (tmp_path / "vocab.json").write_text(json.dumps(vocab))
    (tmp_path / "merges.txt").write_text("#version: 0.2\n")
    tok = CLIPTokenizer(str(tmp_path / "vocab.json"), str(tmp_path / "merges.txt"), model_max_length=77)
    torch.manual_seed(0)
    enc = CLIPTextModel(CLIPTextConfig(vocab_size=len(vocab), hidden_size=ucfg.cross_attention_dim, intermediate_size=128,
                                       num_hidden_layers=2, num_attention_heads=4, max_position_embeddings=77,
                                       bos_token_id=vocab["<|startoftext|>"], eos_token_id=vocab["<|endoftext|>"],
                                       pad_token_id=vocab["<|endoftext|>"])).eval()
    from edgestyle_amd.pipeline import StableDiffusionControlNetPipeline
    p2 = StableDiffusionControlNetPipeline(vae=pipe.vae, text_encoder=enc, tokenizer=tok, unet=pipe.unet,
                                           controlnet=pipe.controlnet).to(DEV)
    lat, _, _, conds = _inputs(ucfg, 1, seed=91)
    prompt, neg = "edgestyle, red, dress some text", "blurry"
    with torch.no_grad():
        def emb(t):
            ids = tok([t], padding="max_length", max_length=77, truncation=True, return_tensors="pt").input_ids
            return enc(ids)[0].float()
        pe, ne = emb(prompt), emb(neg)
    assert pe.shape == (1, 77, ucfg.cross_attention_dim) and not torch.equal(pe, ne)
    a = p2(prompt=prompt, negative_prompt=neg, image=conds, latents=lat, guidance_scale=5.0, num_inference_steps=3,
           output_type="pt").images
    b = p2(prompt_embeds=pe, negative_prompt_embeds=ne, image=conds, latents=lat, guidance_scale=5.0,
           num_inference_steps=3, output_type="pt").images
    assert torch.equal(a, b)
    with pytest.raises(ValueError):
        p2(prompt=prompt, prompt_embeds=pe, image=conds, latents=lat)


@pytest.mark.parametrize("start,end", [(0.0, 0.5), (0.5, 1.0), (0.34, 0.67)])
def test_control_guidance_windows_skip_the_controlnets_and_match_the_oracle(built, start, end):
    """PL:419-427: outside [control_guidance_start, control_guidance_end] every controlnet_keep is 0.  The reference still runs
    the six nets and multiplies by 0 (the fusion blocks then see zeros and return their bias / LayerNorm-plane constants);
    here such a step replays a UNet-only graph plus those constants.  Against the oracle, which follows
    the reference literally; eager == graph replay == whole-loop graph; and close to the scale-0 form (ES_WINDOW_SKIP=0)."""
    from oracle import sd15_oracle as O
    from edgestyle_amd import pipeline as P
    pipe, ws, ucfg, vcfg = built
    lat, pe, ne, conds = _inputs(ucfg, 1, seed=11)
    T = 6
    ref = O.pipeline(ws["unet"], ucfg, ws["fusion"], oracle_nets(ws, ucfg), ws["vae"], vcfg, lat, pe, ne,
                     [c.repeat(2, 1, 1, 1) for c in conds], num_inference_steps=T, guidance_scale=6.0,
                     control_guidance_start=start, control_guidance_end=end)
    kw = dict(prompt_embeds=pe, negative_prompt_embeds=ne, image=conds, latents=lat, guidance_scale=6.0, num_inference_steps=T,
              control_guidance_start=start, control_guidance_end=end, output_type="pt")
    out = pipe(**kw).images
    loop = pipe._last_loop
    want_skip = tuple((i / T < start) or ((i + 1) / T > end) for i in range(T))
    assert loop.skip == want_skip and any(want_skip) and not all(want_skip)
    assert loop.graph is not None and loop.graph_unet is not None
    assert psnr(out, ref) >= 40.0, psnr(out, ref)
    out2 = pipe(**kw).images                               # captures the whole loop (both step forms in one graph)
    out3 = pipe(**kw).images                               # replays it
    assert loop.loop_graph is not None and torch.equal(out, out2) and torch.equal(out, out3)
    pipe.use_graph = False
    try:
        out4 = pipe(**kw).images
    finally:
        pipe.use_graph = True
    assert torch.equal(out, out4)
    P.WINDOW_SKIP = False
    try:
        out5 = pipe(**kw).images                           # the reference's form: all nets run, scales 0
    finally:
        P.WINDOW_SKIP = True
    assert pipe._last_loop.skip == (False,) * T
    assert psnr(out5, ref) >= 40.0 and psnr(out, out5) >= 45.0
    out6 = pipe(**kw).images                               # and back: the loop re-captures for the other pattern
    assert torch.equal(out, out6)


def test_stock_pipeline_resamples_the_vae_conditions_every_step(built):
    """VERDICT r3 missing 2: what the reference's test script really runs (TT:263-272) is the STOCK
    StableDiffusionControlNetPipeline - the raw condition images reach CachedControlNetModel.forward at every step, which embeds
    them every time (CL:199-203), and a VAE-conditioned net draws a fresh latent_dist.sample() per step (CL:38-42).  Here the
    encoder moments are cached and only sample + conv_in run per step, from a per-step noise table (device table, row picked by
    the step counter inside the captured step).  Against the oracle doing CL:38-42 at every step; eager == graphs; the
    EdgeStyle class (embed once) differs, and equals the stock class with resample_cond_each_step=False."""
    from oracle import sd15_oracle as O
    from edgestyle_amd.pipeline import EdgeStyleStableDiffusionControlNetPipeline
    pipe, ws, ucfg, vcfg = built
    assert pipe.resample_cond_each_step is True and EdgeStyleStableDiffusionControlNetPipeline.resample_cond_each_step is False
    g = torch.Generator().manual_seed(23)
    s, T = ucfg.sample_size, 5
    res = s * 8
    lat, pe, ne, _ = _inputs(ucfg, 1, seed=9)
    imgs = [(torch.rand(1, 3, res, res, generator=g) * (2 if i % 2 == 0 else 1) - (1 if i % 2 == 0 else 0)).half().float()
            for i in range(6)]
    tables = [torch.randn(T, 2, 4, s, s, generator=g) if i % 2 == 0 else None for i in range(6)]
    nets = oracle_nets(ws, ucfg)
    oconds, rs = [], {}
    for i in range(6):
        im2 = torch.cat([imgs[i]] * 2)
        if i % 2 == 0:
            mom = O.vae_encode_moments(ws["vae"], vcfg, im2)
            rs[i] = (mom, tables[i])
            oconds.append(O.vae_cond_from_moments(nets[i][0], vcfg, mom, tables[i][0]))
            assert torch.allclose(oconds[-1], O.vae_cond_embedding(nets[i][0], ws["vae"], vcfg, im2, tables[i][0]), atol=1e-5)
        else:
            oconds.append(O.cond_embedding(ws["openpose"], ucfg, im2))
    ref = O.pipeline(ws["unet"], ucfg, ws["fusion"], nets, ws["vae"], vcfg, lat, pe, ne, oconds, num_inference_steps=T,
                     guidance_scale=5.0, cond_resample=rs)
    ref_once = O.pipeline(ws["unet"], ucfg, ws["fusion"], nets, ws["vae"], vcfg, lat, pe, ne, oconds, num_inference_steps=T,
                          guidance_scale=5.0)
    kw = dict(prompt_embeds=pe, negative_prompt_embeds=ne, image=imgs, latents=lat, guidance_scale=5.0, num_inference_steps=T,
              cond_noise=tables, output_type="pt")
    out = pipe(**kw).images
    assert sorted(pipe._last_loop.resample) == [0, 2, 4]
    assert psnr(out, ref) >= 40.0, psnr(out, ref)
    assert psnr(out, ref_once) < psnr(out, ref) - 3.0                     # it is NOT the embed-once result
    out2, out3 = pipe(**kw).images, pipe(**kw).images                      # whole-loop graph captured, then replayed
    assert pipe._last_loop.loop_graph is not None and torch.equal(out, out2) and torch.equal(out, out3)
    pipe.use_graph = False
    try:
        assert torch.equal(out, pipe(**kw).images)
    finally:
        pipe.use_graph = True
    once = pipe(resample_cond_each_step=False, **kw).images               # the EdgeStyle class's semantics: row 0, embedded once
    assert not pipe._last_loop.resample and psnr(once, ref_once) >= 40.0
    es = EdgeStyleStableDiffusionControlNetPipeline(vae=pipe.vae, unet=pipe.unet, controlnet=pipe.controlnet).to(DEV)
    assert torch.equal(es(**kw).images, once)
    # tables drawn from the generator: deterministic per seed, different from the embed-once call of the same seed
    kw2 = {k: v for k, v in kw.items() if k != "cond_noise"}
    a = pipe(generator=torch.Generator().manual_seed(3), **kw2).images
    b = pipe(generator=torch.Generator().manual_seed(3), **kw2).images
    c = pipe(generator=torch.Generator().manual_seed(3), resample_cond_each_step=False, **kw2).images
    assert torch.equal(a, b) and not torch.equal(a, c)
