"""Parity at the BENCHMARKED size: SD1.5 width (320/640/1280/1280 channels, head_dim 40/80/160), 64x64 latents, the
grouped 14-sample lockstep step that bench.py times, the 4-step 512x512 pipeline with the full-size VAE decode, the
single-ControlNet pipeline of BASELINE configs[0] and the batch-8 graph run of configs[2] — HIP path vs the CPU oracle
run live on this box AND vs the fixtures committed under tests/golden/ (tests/golden/make_golden_full.py).

Tolerances (fp16 path vs fp32 oracle; the reference's own cross-backend policy is export_onnx.py:329-334, observed miss
9.2e-4 abs in fp32, README.md:237-251): one step noise_pred <= 2e-2 max abs and <= 2e-2 relative to the tensor's max;
the 13 fused residuals <= 2e-2 relative; decoded image PSNR >= 40 dB (north_star); batch-8 vs batch-1 >= 45 dB.
Parity stays UNPINNED against diffusers itself (oracle/sd15_oracle.py header)."""
import os

import pytest
import torch
from safetensors.torch import load_file

from tests import helpers as H

pytestmark = pytest.mark.gpu
DEV = "cuda"
GOLD = os.path.join(os.path.dirname(__file__), "golden")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# configs[4] floors: 1 dB under what the GPU box measures (profiles/r0*_fullsize_parity.jsonl; VERDICT r3 weak 2: the round-3 floors
# sat 3-8 dB under the measurements).  Round 3 / 4 measured 38.0-38.5 dB (2 steps, vs the fp32 oracle), 36.1-36.7 dB (batch 4 vs
# batch 1) and, for the 12-step fixture, the value recorded by test_config4_batch4_768_bf16_12_steps_vs_golden.
CONFIG4_PSNR_FLOOR_2 = 37.0
CONFIG4_PSNR_FLOOR_B4_VS_B1 = 35.0
CONFIG4_PSNR_FLOOR_12 = 42.0        # measured 43.03 dB (12 DDIM steps; latents 2.0e-2 relative, flat from step 1 on)
CONFIG4_STEP_REL_CEIL = 1.5e-2      # one step's noise_pred relative to the tensor's max: measured 1.08e-2


def record(name, **vals):
    """Measured errors go to gpurun_out/ (scratch) so DESIGN.md can quote what the GPU box saw."""
    import json
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "fullsize_parity.jsonl"), "a") as f:
            f.write(json.dumps(dict(test=name, **{k: (round(v, 6) if isinstance(v, float) else v) for k, v in vals.items()})) + "\n")
    except OSError:
        pass


@pytest.fixture(scope="module")
def full():
    from edgestyle_amd.models import (UNet2DConditionModel, ControlNetModel, ControlLoRAModel, AutoencoderKL,
                                      EdgeStyleMultiControlNetModel)
    from edgestyle_amd.pipeline import EdgeStyleStableDiffusionControlNetPipeline
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    ucfg, vcfg, ws = H.full_weights()
    unet = UNet2DConditionModel(ws["unet"], ucfg, torch.float16)
    vae = AutoencoderKL(ws["vae"], vcfg)
    pose = ControlNetModel(ws["openpose"], ucfg, torch.float16)
    nets = []
    for key in ("lora0", "lora1"):
        n = ControlLoRAModel(ws[key], ucfg, lora_linear_rank=H.FULL_RANK, uses_vae=True)
        n.set_autoencoder(vae)
        n.tie_weights(unet)                                   # TT:259-261
        nets.append(n)
    mc = EdgeStyleMultiControlNetModel([nets[0], pose, nets[1], pose, nets[1], pose], ucfg)   # TT:50, TT:252-258
    mc.load_state_dict(ws["fusion"])
    pipe = EdgeStyleStableDiffusionControlNetPipeline(vae=vae, unet=unet, controlnet=mc).to(DEV)
    return dict(pipe=pipe, ws=ws, ucfg=ucfg, vcfg=vcfg, unet=unet, vae=vae, pose=pose)


def test_full_width_grouped_step_vs_oracle_and_golden(full):
    """One 6-cond CFG step (== export_onnx.py:43-74) exactly as bench.py runs it: N = 2, grouped lockstep launches over
    14 samples ([2,6,4,2]), head_dim 40/80/160, 160-wide tiles, LN / ffo / shortcut folds, split-K at K = 11520."""
    from oracle import sd15_oracle as O
    from edgestyle_amd.models import StepRunner
    pipe, ws, ucfg = full["pipe"], full["ws"], full["ucfg"]
    x, ehs, conds = H.full_step_inputs()
    runner = StepRunner(pipe.unet, pipe.controlnet)
    assert runner.mode == "grouped" and runner._grouped_encoder(2).groupable(8 * 8)
    assert runner._grouped_encoder(2).counts == [2, 6, 4, 2]
    runner.keep_debug = True
    out = runner.step_nchw(x.to(DEV), H.FULL_STEP_T, ehs.to(DEV), [c.to(DEV) for c in conds], H.FULL_STEP_SCALES)
    torch.cuda.synchronize()
    out = out.float().cpu()
    dbg = runner.debug
    assert dbg is not None and "presummed" in dbg, "the grouped path did not run"
    got_res = [(p.float() - s.float()).permute(0, 3, 1, 2).cpu() for p, s in zip(dbg["presummed"], dbg["skips"])]

    with torch.no_grad():
        down, mid = O.multicontrolnet_forward(ws["fusion"], H.oracle_nets(ws, ucfg), x, H.FULL_STEP_T, ehs, conds,
                                              H.FULL_STEP_SCALES)
        ref = O.unet_forward(ws["unet"], ucfg, x, H.FULL_STEP_T, ehs, down, mid)
    err = float((out - ref).abs().max())
    lv = [H.rel_err(g_, r_) for g_, r_ in zip(got_res, down + [mid])]
    record("full_step", noise_max_abs=err, noise_rel=H.rel_err(out, ref), noise_ref_max=float(ref.abs().max()),
           fused_rel_max=max(lv), fused_rel=[round(v, 5) for v in lv])
    assert err <= 2e-2, err
    assert H.rel_err(out, ref) <= 2e-2
    for lvl, v in enumerate(lv):
        assert v <= 2e-2, (lvl, v)

    gold = load_file(os.path.join(GOLD, "full_step.safetensors"))
    assert float((out - gold["noise_pred"]).abs().max()) <= 2e-2
    for lvl, g_ in enumerate(got_res):
        corner = gold[f"fused{lvl}"]
        c, s = corner.shape[1], corner.shape[2]
        assert H.rel_err(g_[:, :c, :s, :s], corner) <= 3e-2, lvl


def test_full_size_pipeline4_and_vae_decode_vs_oracle_and_golden(full):
    """BASELINE configs[1] geometry, 4 DDIM steps, CFG 7.5, graph-replayed, including the full-size VAE decode
    (128 ch @ 512x512): PSNR >= 40 dB vs the live oracle pipeline and vs the committed fixture."""
    from oracle import sd15_oracle as O
    pipe, ws, ucfg, vcfg = full["pipe"], full["ws"], full["ucfg"], full["vcfg"]
    lat, pe, ne, pc = H.full_pipeline_inputs()
    kw = dict(prompt_embeds=pe, negative_prompt_embeds=ne, image=pc, latents=lat, guidance_scale=7.5,
              num_inference_steps=4)
    img = pipe(output_type="pt", **kw).images.float().cpu()
    lat_out = pipe(output_type="latent", **kw).images.float().cpu()
    gold = load_file(os.path.join(GOLD, "full_pipeline4.safetensors"))
    assert img.shape == (1, 3, 512, 512)
    p_gold = H.psnr(img, gold["image"].float())
    assert p_gold >= 40.0, p_gold
    assert H.rel_err(lat_out, gold["latents_out"]) <= 3e-2
    # the decode alone, on the oracle's own final latents: isolates the VAE path at its real size
    dec = pipe.vae.decode(gold["latents_out"].to(DEV) / vcfg.scaling_factor, return_dict=False)[0]
    dec = (dec.float().cpu() / 2 + 0.5).clamp(0, 1)
    assert H.psnr(dec, gold["image"].float()) >= 40.0
    with torch.no_grad():
        ref = O.pipeline(ws["unet"], ucfg, ws["fusion"], H.oracle_nets(ws, ucfg), ws["vae"], vcfg, lat, pe, ne,
                         [c.repeat(2, 1, 1, 1) for c in pc], num_inference_steps=4, guidance_scale=7.5)
    p_live = H.psnr(img, ref)
    record("full_pipeline4", psnr_vs_golden=p_gold, psnr_vs_live_oracle=p_live, psnr_decode_only=H.psnr(dec, gold["image"].float()),
           latents_rel=H.rel_err(lat_out, gold["latents_out"]))
    assert p_live >= 40.0, p_live


def test_single_controlnet_pipeline_baseline_config0(full):
    """BASELINE configs[0]: UNet + ONE openpose ControlNetModel as `controlnet` (PL:338-351: residuals added without
    fusion blocks), 512x512, 4 DDIM steps, raw pose image; vs the fp32 oracle fixture."""
    from edgestyle_amd.pipeline import StableDiffusionControlNetPipeline
    lat, pe, ne, pose_img = H.single_cn_inputs()
    pipe1 = StableDiffusionControlNetPipeline(vae=full["vae"], unet=full["unet"], controlnet=full["pose"]).to(DEV)
    kw = dict(prompt_embeds=pe, negative_prompt_embeds=ne, image=pose_img, latents=lat, guidance_scale=7.5,
              num_inference_steps=4)
    img = pipe1(output_type="pt", **kw).images.float().cpu()
    lat_out = pipe1(output_type="latent", **kw).images.float().cpu()
    gold = load_file(os.path.join(GOLD, "single_cn_pipeline4.safetensors"))
    assert H.rel_err(lat_out, gold["latents_out"]) <= 3e-2
    p = H.psnr(img, gold["image"].float())
    record("single_cn_pipeline4", psnr_vs_golden=p, latents_rel=H.rel_err(lat_out, gold["latents_out"]))
    assert p >= 40.0, p
    # eager == graph replay, bit for bit
    pipe1.use_graph = False
    img2 = pipe1(output_type="pt", **kw).images.float().cpu()
    assert torch.equal(img, img2)


def test_batch8_graph_matches_batch1_requests(full):
    """BASELINE configs[2] (batch 8, hipGraph-captured step): every image of the batch >= 45 dB against the same request
    served alone at batch 1 (different tile plans / split-K choices, same arithmetic)."""
    pipe, ucfg = full["pipe"], full["ucfg"]
    g = torch.Generator().manual_seed(7)
    s, c0 = ucfg.sample_size, ucfg.block_out_channels[0]
    B = 8
    lat = torch.randn(B, 4, s, s, generator=g)
    pe = (torch.randn(B, 77, ucfg.cross_attention_dim, generator=g) * 0.5).half().float()
    ne = (torch.randn(B, 77, ucfg.cross_attention_dim, generator=g) * 0.5).half().float()
    conds = [(torch.randn(1, c0, s, s, generator=g) * 0.3).half().float() for _ in range(6)]
    img8 = pipe(prompt_embeds=pe, negative_prompt_embeds=ne, image=conds, latents=lat, guidance_scale=7.5,
                num_inference_steps=4, output_type="pt").images.float().cpu()
    assert img8.shape == (B, 3, 512, 512)
    ps = []
    for i in range(B):
        img1 = pipe(prompt_embeds=pe[i:i + 1], negative_prompt_embeds=ne[i:i + 1], image=conds, latents=lat[i:i + 1],
                    guidance_scale=7.5, num_inference_steps=4, output_type="pt").images.float().cpu()
        ps.append(H.psnr(img8[i:i + 1], img1))
    record("batch8_vs_batch1", psnr_min=min(ps), psnr=[round(p, 2) for p in ps])
    assert min(ps) >= 45.0, ps
    # ... and against the ORACLE: the request of the committed 4-step fixture served as element 3 of a batch of 8
    lat4, pe4, ne4, pc4 = H.full_pipeline_inputs()
    lat_b, pe_b, ne_b = lat.clone(), pe.clone(), ne.clone()
    lat_b[3], pe_b[3], ne_b[3] = lat4[0], pe4[0], ne4[0]
    img_b = pipe(prompt_embeds=pe_b, negative_prompt_embeds=ne_b, image=pc4, latents=lat_b, guidance_scale=7.5,
                 num_inference_steps=4, output_type="pt").images.float().cpu()
    gold = load_file(os.path.join(GOLD, "full_pipeline4.safetensors"))
    p_gold = H.psnr(img_b[3:4], gold["image"].float())
    record("batch8_vs_oracle_fixture", psnr_vs_golden=p_gold)
    assert p_gold >= 40.0, p_gold


def test_batch16_operands_beyond_2gib_match_batch1_requests(full):
    """16 try-ons per call at 512 x 512: the batched VAE encode of the conditions (48 samples of 128 ch @ 512 x 512 = 3.2 GB) and the
    level-0 feed-forward input (2.35 GB) outgrow the kernels' 32-bit buffer offsets; es_conv_gemm / es_linear_xs then run such a
    launch as runs of whole samples (tests/test_ops_gpu.py checks the cuts bit for bit at small sizes).  Here: the first and the
    last image of the batch against the same requests served alone (RGB conditions, per-image seeds as in bench.py)."""
    import bench
    pipe, ucfg, vcfg = full["pipe"], full["ucfg"], full["vcfg"]
    for net in pipe.controlnet.nets:
        if getattr(net.config, "uses_vae", False):
            net.set_autoencoder(pipe.vae)
    B = 16
    lat, pe, ne, imgs, cn = bench.make_inputs(ucfg, vcfg, B, torch.device(DEV), seed=42)
    img16 = pipe(prompt_embeds=pe, negative_prompt_embeds=ne, image=imgs, latents=lat, guidance_scale=7.5, num_inference_steps=2,
                 output_type="pt", cond_noise=cn).images.float().cpu()
    assert img16.shape == (B, 3, 512, 512) and bool(torch.isfinite(img16).all())
    ps = []
    for j in (0, B - 1):
        l1, p1, n1, im1, c1 = bench.make_inputs(ucfg, vcfg, 1, torch.device(DEV), seed=42, first_index=j)
        img1 = pipe(prompt_embeds=p1, negative_prompt_embeds=n1, image=im1, latents=l1, guidance_scale=7.5, num_inference_steps=2,
                    output_type="pt", cond_noise=c1).images.float().cpu()
        ps.append(H.psnr(img16[j:j + 1], img1))
    for k in [k for k in pipe._loops if k[0] == B]:
        del pipe._loops[k]                                   # that batch size's graphs and buffers
    torch.cuda.empty_cache()
    record("batch16_vs_batch1", psnr=[round(p, 2) for p in ps])
    assert min(ps) >= 45.0, ps


@pytest.fixture(scope="module")
def full96(full):
    """BASELINE configs[4] geometry: SD1.5 width, 96x96 latents (768x768 images), bf16.  Weights are the `full` fixture's,
    rounded to bf16; the fusion blocks' LayerNorm parameters are [3C, h, w] (MC:14-16) and exist per latent size."""
    import dataclasses
    from edgestyle_amd import weights as W
    from edgestyle_amd.models import StepRunner, AutoencoderKL
    from edgestyle_amd.pipeline import EdgeStyleStableDiffusionControlNetPipeline
    ucfg = dataclasses.replace(full["ucfg"], sample_size=96)
    ws = {k: H.quantize(v, torch.bfloat16) for k, v in full["ws"].items() if k != "fusion"}
    ws["fusion"] = H.quantize(W.random_state_dict(W.fusion_shapes(ucfg), 0, "fusion."), torch.bfloat16)
    runner = StepRunner.from_state_dicts(ws, ucfg, torch.bfloat16, DEV, rank=H.FULL_RANK)
    vae = AutoencoderKL(ws["vae"], full["vcfg"], torch.bfloat16).to(DEV)
    pipe = EdgeStyleStableDiffusionControlNetPipeline(vae=vae, unet=runner.unet, controlnet=runner.controlnet).to(DEV)
    return dict(ucfg=ucfg, vcfg=full["vcfg"], ws=ws, runner=runner, pipe=pipe)


def test_full_width_step_768_bf16_config4(full96):
    """BASELINE configs[4] (768x768, bf16): one full-width 6-cond CFG step at 96x96 latents in bf16 vs the fp32 oracle on
    the same bf16-rounded weights and inputs.  The reference itself is hard-wired to 64x64 (MC:73-102), so the oracle's
    size-generic restatement is the only checker.  bf16 keeps 8 mantissa bits (fp16: 11), so the bar is 8x the fp16 one
    relative to the tensor's max: <= 6e-2 (fp16 above: <= 2e-2; measured values go to gpurun_out/fullsize_parity.jsonl)."""
    from oracle import sd15_oracle as O
    ucfg, ws, runner = full96["ucfg"], full96["ws"], full96["runner"]
    g = torch.Generator().manual_seed(45)
    N, s, c0 = 2, 96, ucfg.block_out_channels[0]
    x = torch.randn(N, 4, s, s, generator=g).bfloat16().float()
    ehs = (torch.randn(N, 77, ucfg.cross_attention_dim, generator=g) * 0.5).bfloat16().float()
    conds = [(torch.randn(N, c0, s, s, generator=g) * 0.3).bfloat16().float() for _ in range(6)]
    assert runner.mode == "grouped"
    out = runner.step_nchw(x.to(DEV), H.FULL_STEP_T, ehs.to(DEV), [c.to(DEV) for c in conds], H.FULL_STEP_SCALES)
    torch.cuda.synchronize()
    out = out.float().cpu()
    with torch.no_grad():
        ref = O.denoise_step(ws["unet"], ucfg, ws["fusion"], H.oracle_nets(ws, ucfg), x, H.FULL_STEP_T, ehs, conds,
                             H.FULL_STEP_SCALES)
    rel = H.rel_err(out, ref)
    record("full_step_768_bf16", noise_max_abs=float((out - ref).abs().max()), noise_rel=rel,
           noise_ref_max=float(ref.abs().max()))
    assert out.shape == (N, 4, s, s) and torch.isfinite(out).all()
    assert rel <= CONFIG4_STEP_REL_CEIL, rel


def test_full_size_pipeline_768_bf16_config4_vs_oracle(full96):
    """BASELINE configs[4] end to end at its real size: 768x768, bf16, 2 DDIM steps, CFG 7.5, graph-replayed, including the
    VAE decode at 768x768 - vs the fp32 oracle pipeline run live (about a minute of CPU).  Bar: PSNR >= 30 dB (the bf16 bar
    of the tiny analogue in test_pipeline_gpu.py; fp16 tests: >= 40 dB); the measured value is recorded."""
    from oracle import sd15_oracle as O
    ucfg, vcfg, ws, pipe = full96["ucfg"], full96["vcfg"], full96["ws"], full96["pipe"]
    g = torch.Generator().manual_seed(46)
    s, c0 = 96, ucfg.block_out_channels[0]
    lat = torch.randn(1, 4, s, s, generator=g)
    pe = (torch.randn(1, 77, ucfg.cross_attention_dim, generator=g) * 0.5).bfloat16().float()
    ne = (torch.randn(1, 77, ucfg.cross_attention_dim, generator=g) * 0.5).bfloat16().float()
    pc = [(torch.randn(1, c0, s, s, generator=g) * 0.3).bfloat16().float() for _ in range(6)]
    img = pipe(prompt_embeds=pe, negative_prompt_embeds=ne, image=pc, latents=lat, guidance_scale=7.5, num_inference_steps=2,
               output_type="pt").images.float().cpu()
    assert img.shape == (1, 3, 768, 768) and torch.isfinite(img).all()
    with torch.no_grad():
        ref = O.pipeline(ws["unet"], ucfg, ws["fusion"], H.oracle_nets(ws, ucfg), ws["vae"], vcfg, lat, pe, ne,
                         [c.repeat(2, 1, 1, 1) for c in pc], num_inference_steps=2, guidance_scale=7.5)
    p_ = H.psnr(img, ref)
    record("full_pipeline2_768_bf16", psnr_vs_live_oracle=p_)
    assert p_ >= CONFIG4_PSNR_FLOOR_2, p_


def test_full_width_chain_modes_agree(full):
    """ES_CHAIN_MODE: the grouped lockstep pass (default, what the bench replays) and the four chains one after the other do
    the same arithmetic with different tile plans / launch groupings: at SD1.5 width they agree to 5e-3 of the tensor's max
    (the tiny-width tests compare each with the oracle)."""
    from edgestyle_amd.models import StepRunner
    pipe = full["pipe"]
    x, ehs, conds = H.full_step_inputs()
    runner = StepRunner(pipe.unet, pipe.controlnet)
    outs = {}
    for mode in ("grouped", "serial"):
        runner.mode = mode
        o = runner.step_nchw(x.to(DEV), H.FULL_STEP_T, ehs.to(DEV), [c.to(DEV) for c in conds], H.FULL_STEP_SCALES)
        torch.cuda.synchronize()
        outs[mode] = o.float().cpu()
    record("full_step_chain_modes", serial_vs_grouped=H.rel_err(outs["serial"], outs["grouped"]))
    assert H.rel_err(outs["serial"], outs["grouped"]) <= 5e-3


def test_full_size_guess_mode_and_unipc_vs_oracle(full):
    """The two optional loop variants at the benchmarked geometry, latents only (no decode): guess_mode (CL:256-264, PL:453-459,
    487-497: un-grouped per-net chains on the conditional half, log-spaced level scales) over 2 DDIM steps, and the UniPC
    scheduler the reference's callers assign (TT:273) over 3 steps (fused predictor-corrector kernel), each vs the oracle."""
    from oracle import sd15_oracle as O
    from edgestyle_amd.schedulers import UniPCMultistepScheduler
    pipe, ws, ucfg, vcfg = full["pipe"], full["ws"], full["ucfg"], full["vcfg"]
    lat, pe, ne, pc = H.full_pipeline_inputs(seed=47)
    nets = H.oracle_nets(ws, ucfg)
    gs = 5.0
    kw = dict(prompt_embeds=pe, negative_prompt_embeds=ne, image=pc, latents=lat, guidance_scale=gs, output_type="latent")
    got_g = pipe(num_inference_steps=2, guess_mode=True, **kw).images.float().cpu()
    with torch.no_grad():
        ref_g = O.pipeline(ws["unet"], ucfg, ws["fusion"], nets, ws["vae"], vcfg, lat, pe, ne, list(pc), num_inference_steps=2,
                           guidance_scale=gs, guess_mode=True, decode=False)
    e_g = H.rel_err(got_g, ref_g)
    # UniPC: the oracle's scheduler restatement around the oracle's step
    steps = 3
    sch = O.UniPC()
    ts = sch.set_timesteps(steps)
    xo = lat.clone()
    ehs = torch.cat([ne, pe])
    oconds = [c.repeat(2, 1, 1, 1) for c in pc]
    with torch.no_grad():
        for t in ts.tolist():
            eps = O.denoise_step(ws["unet"], ucfg, ws["fusion"], nets, torch.cat([xo] * 2), t, ehs, oconds, [1.0] * 6)
            e_u, e_t = eps.chunk(2)
            xo = sch.step(e_u + gs * (e_t - e_u), t, xo)
    old = pipe.scheduler
    pipe.scheduler = UniPCMultistepScheduler.from_config(getattr(old, "config", None))
    try:
        got_u = pipe(num_inference_steps=steps, **kw).images.float().cpu()
    finally:
        pipe.scheduler = old
    e_u_ = H.rel_err(got_u, xo)
    record("full_guess_mode_and_unipc", guess_latents_rel=e_g, unipc_latents_rel=e_u_)
    assert e_g <= 3e-2, e_g
    assert e_u_ <= 3e-2, e_u_


# ----------------------------------------------------------------------------------------------------------------
# round 3: the HEADLINE configuration end to end (50 steps), the RGB-image path at its real size, the native ABI at
# SD1.5 width, configs[4] at its batch size.  Fixtures: tests/golden/make_golden_full50.py (CPU oracle outputs).
# ----------------------------------------------------------------------------------------------------------------
def test_headline_config_50_steps_vs_golden(full):
    """BASELINE configs[1] exactly as bench.py times it: 512x512, 50 DDIM steps, CFG 7.5, batch 1, graph-replayed (the
    loop PL:435-543 as TT:357 drives it) - final latents and decoded image against the committed 50-step oracle fixture.
    north_star's gate: PSNR >= 40 dB on the decoded image.  The latents after steps 1 / 5 / 10 / 25 of the same run are
    compared too (eager run with callback_on_step_end, PL:524-534), so the error growth along the loop is on record."""
    pipe = full["pipe"]
    gold = load_file(os.path.join(GOLD, "full_pipeline50.safetensors"))
    lat, pe, ne, pc = H.full_pipeline_inputs(seed=48)
    kw = dict(prompt_embeds=pe, negative_prompt_embeds=ne, image=pc, latents=lat, guidance_scale=7.5,
              num_inference_steps=50)
    lat_out = pipe(output_type="latent", **kw).images.float().cpu()      # captures the step graph
    img = pipe(output_type="pt", **kw).images.float().cpu()              # second identical call: the whole-loop graph
    assert img.shape == (1, 3, 512, 512) and torch.isfinite(img).all()
    p50 = H.psnr(img, gold["image"].float())
    e50 = H.rel_err(lat_out, gold["latents_out"])
    seen = {}

    def grab(_pipe, i, t, kwargs):
        if i + 1 in (1, 5, 10, 25):
            seen[i + 1] = kwargs["latents"].float().cpu().clone()
        return {}

    lat_eager = pipe(output_type="latent", callback_on_step_end=grab, **kw).images.float().cpu()
    assert torch.equal(lat_eager, lat_out)                               # eager == graph replay, all 50 steps
    growth = {k: H.rel_err(v, gold[f"latents_step{k}"]) for k, v in seen.items()}
    record("headline_pipeline50", psnr_vs_golden=p50, latents_rel=e50, latents_rel_by_step={str(k): round(v, 6) for k, v in growth.items()})
    assert p50 >= 40.0, p50
    assert e50 <= 3e-2, e50
    assert all(v <= 3e-2 for v in growth.values()), growth


def test_rgb_condition_images_full_size_vs_golden(full):
    """The one-time condition embedding at its REAL size (inside bench.py's timed region): six [1,3,512,512] RGB images
    through prepare_image (PL:629-664) - 3 x VAE encode (128 ch @ 512x512) + latent_dist.sample() with the caller's
    noise x 0.18215 -> conv_in (CL:28-42, 289-290) and 3 x the openpose conv stack - against the oracle's
    vae_cond_embedding / cond_embedding, then 2 DDIM steps + decode against the oracle pipeline fed with ITS embeddings."""
    pipe = full["pipe"]
    gold = load_file(os.path.join(GOLD, "full_rgb_pipeline2.safetensors"))
    for net in pipe.controlnet.nets:
        if getattr(net.config, "uses_vae", False):
            net.set_autoencoder(pipe.vae)                                # TT:252-258 (vae= of from_pretrained)
    imgs, noise, lat, pe, ne = H.full_rgb_inputs()
    conds = pipe.prepare_images(imgs, 1, True, noise)                    # NHWC [2,64,64,320] each
    errs = []
    for i, c in enumerate(conds):
        corner = gold[f"cond{i}"]                                        # [2,64,16,16] NCHW
        got = c[:, :16, :16, :64].permute(0, 3, 1, 2).float().cpu()
        st = gold[f"cond{i}_stats"]
        e = float((got - corner).abs().max() / (st[2] + 1e-6))           # relative to the full tensor's max
        errs.append(e)
        assert abs(float(c.float().mean()) - float(st[0])) <= 2e-3 * float(st[2]) + 1e-3, i
        assert abs(float(c.float().abs().mean()) - float(st[1])) <= 1e-2 * float(st[1]) + 1e-3, i
    kw = dict(prompt_embeds=pe, negative_prompt_embeds=ne, image=imgs, latents=lat, guidance_scale=7.5,
              num_inference_steps=2, cond_noise=noise)
    img = pipe(output_type="pt", **kw).images.float().cpu()
    lat_out = pipe(output_type="latent", **kw).images.float().cpu()
    p_ = H.psnr(img, gold["image"].float())
    e_ = H.rel_err(lat_out, gold["latents_out"])
    record("rgb_pipeline2", cond_rel=[round(e, 6) for e in errs], psnr_vs_golden=p_, latents_rel=e_)
    assert max(errs) <= 1e-2, errs
    assert p_ >= 40.0, p_
    assert e_ <= 3e-2, e_


def test_native_abi_full_width_rgb_to_image_equals_pipeline_bitwise(full):
    """The step-level C ABI at SD1.5 width (what tools/ctx_image_fullsize.py checked outside the suite): RGB condition images
    -> es_prepare_conds -> es_denoise_loop -> es_vae_decode with raw device pointers == pipe(image=rgb, cond_noise=...) bit
    for bit, per-plan graphs and the whole-loop graph; and es_denoise_step == StepRunner.step on the pipeline's embeddings."""
    from edgestyle_amd.native import NativeEngine
    pipe = full["pipe"]
    for net in pipe.controlnet.nets:
        if getattr(net.config, "uses_vae", False):
            net.set_autoencoder(pipe.vae)
    imgs, noise, lat, pe, ne = H.full_rgb_inputs(seed=50)
    T, gs = 3, 7.5
    kw = dict(prompt_embeds=pe, negative_prompt_embeds=ne, image=imgs, latents=lat, guidance_scale=gs, num_inference_steps=T,
              cond_noise=noise)
    want_lat = pipe(output_type="latent", **kw).images.clone()
    want_img = pipe(output_type="pt", **kw).images.clone()
    eng = NativeEngine(pipe, batch_size=1, num_inference_steps=T)
    try:
        ehs = torch.cat([ne, pe]).to(DEV, torch.float16).contiguous()
        x = lat.permute(0, 2, 3, 1).contiguous().to(DEV)
        for use_graphs in (True, 2):
            eng.set_options(use_graphs=use_graphs)
            eng.prepare_conds([im.to(DEV) for im in imgs], [None if z is None else z.to(DEV) for z in noise])
            got = eng.denoise_loop(x.clone(), ehs, gs)
            img = eng.vae_decode(got)
            torch.cuda.synchronize()
            assert torch.equal(got.permute(0, 3, 1, 2), want_lat), float((got.permute(0, 3, 1, 2) - want_lat).abs().max())
            assert torch.equal(img, want_img)
    finally:
        eng.close()


@pytest.mark.parametrize("B,T", [(1, 3), (8, 2)])
def test_es_load_weights_full_width_context_equals_pipeline_bitwise(full, B, T, tmp_path):
    """SURVEY 8b's es_load_weights at SD1.5 width: the library builds the context itself from the raw state dicts (1.3 G
    parameters: rank-32 LoRA folds, LayerNorm / proj_out / shortcut folds, packing, arena layout, five launch lists - no model
    walk in Python) and RGB condition images -> es_prepare_conds -> es_denoise_loop -> es_vae_decode through it reproduce
    pipe(image=rgb, cond_noise=...) bit for bit.  The golden fixture of the same call (oracle, 50 steps) is covered through the
    pipeline above; this ties the natively built context to it."""
    import time
    from edgestyle_amd.native import NativeContext
    pipe = full["pipe"]
    for net in pipe.controlnet.nets:
        if getattr(net.config, "uses_vae", False):
            net.set_autoencoder(pipe.vae)
    imgs, noise, lat, pe, ne = H.full_rgb_inputs(seed=52)
    gs = 7.5
    if B > 1:                                  # batch 8 (BASELINE configs[2]): per-image latents / prompts / sampling noise, shared images
        g = torch.Generator().manual_seed(53)
        lat = torch.randn(B, *lat.shape[1:], generator=g)
        pe = (torch.randn(B, *pe.shape[1:], generator=g) * 0.5).half().float()
        ne = (torch.randn(B, *ne.shape[1:], generator=g) * 0.5).half().float()
        noise = [None if z is None else torch.randn(2 * B, *z.shape[1:], generator=g).half().float() for z in noise]
    kw = dict(prompt_embeds=pe, negative_prompt_embeds=ne, image=imgs, latents=lat, guidance_scale=gs, num_inference_steps=T,
              cond_noise=noise)
    want_lat = pipe(output_type="latent", **kw).images.clone()
    want_img = pipe(output_type="pt", **kw).images.clone()
    imgs = [im.repeat_interleave(B, dim=0) for im in imgs]         # the C ABI takes one image per request (PL:647-653 repeats them)
    t0 = time.time()
    nat = NativeContext(full["ws"], full["ucfg"], full["vcfg"], batch_size=B, guidance=True, num_inference_steps=T, device=0)
    build_s = time.time() - t0
    try:
        nat.set_alphas_cumprod(pipe.scheduler.alphas_cumprod)
        ehs = torch.cat([ne, pe]).to(DEV, torch.float16).contiguous()
        x = lat.permute(0, 2, 3, 1).contiguous().to(DEV)
        ts = pipe.scheduler.set_timesteps(T).tolist()
        for use_graphs in (True, 2):
            nat.set_options(use_graphs=use_graphs)
            nat.prepare_conds([im.to(DEV) for im in imgs], [None if z is None else z.to(DEV) for z in noise])
            got = nat.denoise_loop(x.clone(), ehs, gs, ts)
            img = nat.vae_decode(got)
            torch.cuda.synchronize()
            assert torch.equal(got.permute(0, 3, 1, 2), want_lat), float((got.permute(0, 3, 1, 2) - want_lat).abs().max())
            assert torch.equal(img, want_img)
        extra = {}
        if B == 1:
            # ... and as a context image written by the library itself: es_ctx_save -> es_ctx_load -> the same bits
            import ctypes as C
            from edgestyle_amd import lib as L
            lib = L.load()
            path = str(tmp_path / "full.esctx")
            t0 = time.time()
            L.check(lib.es_ctx_save(nat.ctx, path.encode()), "es_ctx_save")
            t_save = time.time() - t0
            ctx2 = C.c_void_p()
            t0 = time.time()
            L.check(lib.es_ctx_load(path.encode(), 0, C.byref(ctx2)), "es_ctx_load")
            t_load = time.time() - t0
            try:
                stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
                ims = [im.to(DEV).contiguous() for im in imgs]
                nzs = [None if z is None else z.to(DEV).contiguous() for z in noise]
                ip = (C.c_void_p * 6)(*[im.data_ptr() for im in ims])
                npz = (C.c_void_p * 6)(*[None if z is None else z.data_ptr() for z in nzs])
                x2 = x.clone()
                tsa = (C.c_float * T)(*[float(t) for t in ts])
                L.check(lib.es_prepare_conds(ctx2, ip, npz, stream), "es_prepare_conds")
                L.check(lib.es_denoise_loop(ctx2, C.c_void_p(x2.data_ptr()), C.c_void_p(ehs.data_ptr()), gs, tsa, T, stream), "es_denoise_loop")
                torch.cuda.synchronize()
                assert torch.equal(x2.permute(0, 3, 1, 2), want_lat)
            finally:
                lib.es_ctx_destroy(ctx2)
            size = os.path.getsize(path)
            assert size <= 4 << 30                          # full-size image <= 4 GiB
            extra = dict(image_gib=round(size / 2 ** 30, 2), save_seconds=round(t_save, 1), load_seconds=round(t_load, 1))
            os.remove(path)
        record("es_load_weights_full_width", batch=B, build_seconds=round(build_s, 1), plan_step_calls=nat.plan_size(2),
               arena_gib=round(nat.lib.es_ctx_arena_bytes(nat.ctx) / 2 ** 30, 2), bitwise_equal_to_pipeline=True, **extra)
    finally:
        nat.close()


def test_config4_batch4_768_bf16_vs_oracle(full96):
    """BASELINE configs[4] at ITS batch size: 768x768, bf16, batch 4, CFG 7.5, 2 DDIM steps, graph-replayed, VAE decode
    included - every image against the fp32 oracle pipeline of the same request (the oracle runs the four requests one by
    one: ~2 minutes of CPU).  north_star's bar is 40 dB for fp16; bf16 keeps 8 mantissa bits against 11, and what this
    configuration measures is recorded next to the asserted floor (DESIGN.md quotes it)."""
    from oracle import sd15_oracle as O
    ucfg, vcfg, ws, pipe = full96["ucfg"], full96["vcfg"], full96["ws"], full96["pipe"]
    g = torch.Generator().manual_seed(51)
    s, c0, B = 96, ucfg.block_out_channels[0], 4
    lat = torch.randn(B, 4, s, s, generator=g)
    pe = (torch.randn(B, 77, ucfg.cross_attention_dim, generator=g) * 0.5).bfloat16().float()
    ne = (torch.randn(B, 77, ucfg.cross_attention_dim, generator=g) * 0.5).bfloat16().float()
    pc = [(torch.randn(1, c0, s, s, generator=g) * 0.3).bfloat16().float() for _ in range(6)]
    img = pipe(prompt_embeds=pe, negative_prompt_embeds=ne, image=pc, latents=lat, guidance_scale=7.5, num_inference_steps=2,
               output_type="pt").images.float().cpu()
    assert img.shape == (B, 3, 768, 768) and torch.isfinite(img).all()
    ps = []
    nets = H.oracle_nets(ws, ucfg)
    with torch.no_grad():
        for b in range(2):                                               # two of the four requests: one CPU minute each
            ref = O.pipeline(ws["unet"], ucfg, ws["fusion"], nets, ws["vae"], vcfg, lat[b:b + 1], pe[b:b + 1], ne[b:b + 1],
                             [c.repeat(2, 1, 1, 1) for c in pc], num_inference_steps=2, guidance_scale=7.5)
            ps.append(H.psnr(img[b:b + 1], ref))
    # all four images against the same requests served alone (batch 1) by the HIP path
    solo = [pipe(prompt_embeds=pe[b:b + 1], negative_prompt_embeds=ne[b:b + 1], image=pc, latents=lat[b:b + 1], guidance_scale=7.5,
                 num_inference_steps=2, output_type="pt").images.float().cpu() for b in range(B)]
    pb = [H.psnr(img[b:b + 1], solo[b]) for b in range(B)]
    record("config4_batch4_768_bf16", psnr_vs_oracle=[round(p, 2) for p in ps], psnr_batch4_vs_batch1=[round(p, 2) for p in pb])
    assert min(ps) >= CONFIG4_PSNR_FLOOR_2, ps
    assert min(pb) >= CONFIG4_PSNR_FLOOR_B4_VS_B1, pb


@pytest.mark.parametrize("steps", [12, 50])
def test_config4_batch4_768_bf16_many_steps_vs_golden(full96, steps):
    """BASELINE configs[4] over MORE than two steps: 768x768, bf16, batch 4, CFG 7.5, graph replayed -
    request 0 against the committed fp32 oracle fixtures (tests/golden/make_golden_768.py - the repo's OWN oracle: for the
    diffusers-owned blocks these gates are parity-unpinned, as oracle/sd15_oracle.py says of itself): 12 DDIM steps (latents after steps 1,
    2, 4, 8, 12) and configs[4]'s OWN 50 steps (latents after 1, 5, 10, 25, 50), each with the decoded image - so the growth of the
    bf16 error along the loop is on record.  north_star's gate is PSNR >= 40 dB on the decoded image: asserted here at both step
    counts (measured 43.9 dB at 12 steps with the wide residual stream, 43.0 without; the 2-step tests above sit at 39 dB because
    the first steps carry the largest error and DDIM contracts it afterwards)."""
    pipe = full96["pipe"]
    ucfg = full96["ucfg"]
    gold = load_file(os.path.join(GOLD, f"full96_pipeline{steps}.safetensors"))
    marks = (1, 2, 4, 8, 12) if steps == 12 else (1, 5, 10, 25, 50)
    g = torch.Generator().manual_seed(51)
    s, c0, B = 96, ucfg.block_out_channels[0], 4
    lat = torch.randn(B, 4, s, s, generator=g)
    pe = (torch.randn(B, 77, ucfg.cross_attention_dim, generator=g) * 0.5).bfloat16().float()
    ne = (torch.randn(B, 77, ucfg.cross_attention_dim, generator=g) * 0.5).bfloat16().float()
    pc = [(torch.randn(1, c0, s, s, generator=g) * 0.3).bfloat16().float() for _ in range(6)]
    kw = dict(prompt_embeds=pe, negative_prompt_embeds=ne, image=pc, latents=lat, guidance_scale=7.5, num_inference_steps=steps)
    seen = {}

    def grab(_pipe, i, t, kwargs):
        if i + 1 in marks:
            seen[i + 1] = kwargs["latents"][:1].float().cpu().clone()
        return {}
    lat_eager = pipe(output_type="latent", callback_on_step_end=grab, **kw).images.float().cpu()
    img = pipe(output_type="pt", **kw).images.float().cpu()
    assert img.shape == (B, 3, 768, 768) and torch.isfinite(img).all()
    growth = {k: H.rel_err(v, gold[f"latents_step{k}"]) for k, v in seen.items()}
    p_ = H.psnr(img[:1], gold["image"].float())
    e_ = H.rel_err(lat_eager[:1], gold["latents_out"])
    # the other three requests of the batch (they were checked for finiteness only): each against the SAME request
    # served alone (batch 1) by the HIP path over the same number of steps - a request's image must not depend on what it was batched
    # with beyond the rounding of differently tiled launches
    solo = [pipe(output_type="pt", **dict(kw, prompt_embeds=pe[b:b + 1], negative_prompt_embeds=ne[b:b + 1], latents=lat[b:b + 1])
                 ).images.float().cpu() for b in range(1, B)]
    pb = [H.psnr(img[b:b + 1], solo[b - 1]) for b in range(1, B)]
    record(f"config4_batch4_768_bf16_{steps}_steps", psnr_vs_golden=p_, latents_rel=e_,
           latents_rel_by_step={str(k): round(v, 6) for k, v in growth.items()}, psnr_requests_1_3_batch4_vs_batch1=[round(p, 2) for p in pb])
    assert p_ >= (CONFIG4_PSNR_FLOOR_12 if steps == 12 else 40.0), p_
    assert e_ <= 2.5e-2, e_                 # (measured 1.2e-2 ... 1.8e-2)
    assert min(pb) >= 40.0, pb


def test_control_guidance_window_at_full_size_vs_oracle_and_its_step_time(full):
    """VERDICT r3 item 7a at SD1.5 width: control_guidance_end = 0.5 over 4 DDIM steps (steps 3 and 4 run the UNet alone + the
    fusion-of-zeros constants) against the oracle pipeline, which runs the six nets with scale 0 like the reference (PL:419-427,
    464-470); and what the windowed half costs: per-image time of a 50-step call with the window closed at 0.5 against the same
    call with it open (recorded; the saving is the six encoder passes of 25 steps)."""
    import time
    from oracle import sd15_oracle as O
    pipe, ws, ucfg, vcfg = full["pipe"], full["ws"], full["ucfg"], full["vcfg"]
    lat, pe, ne, pc = H.full_pipeline_inputs(seed=57)
    kw = dict(prompt_embeds=pe, negative_prompt_embeds=ne, image=pc, latents=lat, guidance_scale=7.5)
    img = pipe(num_inference_steps=4, control_guidance_end=0.5, output_type="pt", **kw).images.float().cpu()
    assert pipe._last_loop.skip == (False, False, True, True)
    with torch.no_grad():
        ref = O.pipeline(ws["unet"], ucfg, ws["fusion"], H.oracle_nets(ws, ucfg), ws["vae"], vcfg, lat, pe, ne,
                         [c.repeat(2, 1, 1, 1) for c in pc], num_inference_steps=4, guidance_scale=7.5, control_guidance_end=0.5)
    p_ = H.psnr(img, ref)
    times = {}
    for name, end in (("open", 1.0), ("closed_at_half", 0.5)):
        for _ in range(3):                                   # step graphs, whole-loop graph, one warm replay
            pipe(num_inference_steps=50, control_guidance_end=end, output_type="pt", **kw)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            pipe(num_inference_steps=50, control_guidance_end=end, output_type="pt", **kw)
        torch.cuda.synchronize()
        times[name] = (time.perf_counter() - t0) / 3 * 1e3
    record("control_guidance_window_full_size", psnr_vs_live_oracle=p_, ms_per_image_window_open=times["open"],
           ms_per_image_window_closed_at_half=times["closed_at_half"])
    assert p_ >= 40.0, p_
    # (the two times are RECORDED, not asserted: a wall-clock comparison inside the parity suite fails on box noise or clock state;
    #  profiles/r0N_fullsize_parity.jsonl carries them - 470.7 vs 366.2 ms per image in round 4)
