"""Block-level parity of the HIP executors (edgestyle_amd/engine.py) against the CPU oracle, tiny config."""
import pytest
import torch

from edgestyle_amd import config as C
from tests.helpers import make_weights, quantize, rel_err, to_nhwc, oracle_nets

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def setup():
    ucfg, vcfg = C.tiny_unet(), C.tiny_vae()
    ws = {k: quantize(v) for k, v in make_weights(ucfg, vcfg).items()}
    return ucfg, vcfg, ws


def _inputs(ucfg, N=2, seed=1):
    g = torch.Generator().manual_seed(seed)
    s = ucfg.sample_size
    x = torch.randn(N, 4, s, s, generator=g).half().float()
    ehs = (torch.randn(N, 77, ucfg.cross_attention_dim, generator=g) * 0.5).half().float()
    return x, ehs


def test_unet_forward_with_residuals(setup):
    from oracle import sd15_oracle as O
    from edgestyle_amd import engine as E
    ucfg, _, ws = setup
    x, ehs = _inputs(ucfg)
    g = torch.Generator().manual_seed(7)
    table = ucfg.residual_table()
    res = [(torch.randn(2, c, s, s, generator=g) * 0.2).half().float() for c, s in table]
    ref = O.unet_forward(ws["unet"], ucfg, x, 981, ehs, res[:-1], res[-1])
    net = E.UNet(ws["unet"], ucfg, torch.float16, DEV)
    t = torch.full((2,), 981.0, device=DEV)
    tproj = net.time_proj(t)
    ctx = net.context(ehs.to(DEV, torch.float16))
    out = net.forward(to_nhwc(x, DEV, cpad=8), tproj, ctx, [to_nhwc(r, DEV) for r in res[:-1]], to_nhwc(res[-1], DEV))
    assert rel_err(out.permute(0, 3, 1, 2), ref) < 1e-2


def test_controlnet_forward_batched_and_lora(setup):
    """3 openpose passes as one batch-3N pass == 3 separate reference passes; fused LoRA == unfused oracle LoRA"""
    from oracle import sd15_oracle as O
    from edgestyle_amd import engine as E
    ucfg, _, ws = setup
    x, ehs = _inputs(ucfg)
    g = torch.Generator().manual_seed(3)
    s, c0 = ucfg.sample_size, ucfg.block_out_channels[0]
    conds = [(torch.randn(2, c0, s, s, generator=g) * 0.3).half().float() for _ in range(3)]
    net = E.ControlNet(ws["openpose"], ucfg, torch.float16, DEV)
    t = torch.full((6,), 500.0, device=DEV)
    tproj = net.time_proj(t)
    ctx = net.context(ehs.repeat(3, 1, 1).to(DEV, torch.float16))
    res = net.forward(to_nhwc(x, DEV, cpad=8), tproj, ctx, [to_nhwc(c, DEV) for c in conds], out_scale=0.5)
    for i, c in enumerate(conds):
        d, m = O.controlnet_forward(ws["openpose"], ucfg, x, 500, ehs, c, 0.5)
        for lvl, r in enumerate(d + [m]):
            got = res[lvl][2 * i:2 * i + 2].permute(0, 3, 1, 2)
            assert rel_err(got, r) < 1e-2, (i, lvl)
    # ControlLoRA: tied encoder + LoRA (unfused in the oracle, folded at load in the HIP path)
    tied = O.tie_weights(ws["lora0"], ws["unet"])
    lnet = E.ControlNet(tied, ucfg, torch.float16, DEV, uses_vae=True)
    tproj = lnet.time_proj(t[:2])
    ctx = lnet.context(ehs.to(DEV, torch.float16))
    res = lnet.forward(to_nhwc(x, DEV, cpad=8), tproj, ctx, [to_nhwc(conds[0], DEV)])
    d, m = O.controlnet_forward(tied, ucfg, x, 500, ehs, conds[0], 1.0)
    for lvl, r in enumerate(d + [m]):
        assert rel_err(res[lvl].permute(0, 3, 1, 2), r) < 1e-2, lvl


def test_cond_embedding_and_vae(setup):
    from oracle import sd15_oracle as O
    from edgestyle_amd import engine as E, ops
    ucfg, vcfg, ws = setup
    g = torch.Generator().manual_seed(5)
    s = ucfg.sample_size
    img = torch.rand(1, 3, s * 8, s * 8, generator=g).half().float()
    net = E.ControlNet(ws["openpose"], ucfg, torch.float16, DEV)
    ce = net.embed_cond(to_nhwc(img, DEV, cpad=8))
    assert rel_err(ce.permute(0, 3, 1, 2), O.cond_embedding(ws["openpose"], ucfg, img)) < 1e-2
    vae = E.VAE(ws["vae"], vcfg, torch.float16, DEV)
    img2 = img * 2 - 1
    mom = vae.encode_moments(to_nhwc(img2, DEV, cpad=8))
    assert rel_err(mom.permute(0, 3, 1, 2), O.vae_encode_moments(ws["vae"], vcfg, img2)) < 1e-2
    z = (torch.randn(1, 4, s, s, generator=g)).half().float()
    dec = vae.decode(to_nhwc(z, DEV, cpad=8))
    assert rel_err(dec.permute(0, 3, 1, 2), O.vae_decode(ws["vae"], vcfg, z)) < 1e-2
    # VAE-latent conditioning of a ControlLoRA net (CL:38-42): sample -> *scaling -> conv_in
    noise = torch.randn(1, 4, s, s, generator=g)
    tied = O.tie_weights(ws["lora0"], ws["unet"])
    lnet = E.ControlNet(tied, ucfg, torch.float16, DEV, uses_vae=True)
    zc = ops.vae_sample(mom, noise.to(DEV), 4, 8, vcfg.scaling_factor)
    emb = lnet.embed_latent(zc)
    ref = O.vae_cond_embedding(tied, ws["vae"], vcfg, img2, noise)
    assert rel_err(emb.permute(0, 3, 1, 2), ref) < 1e-2


def test_full_step_matches_oracle(setup):
    """controlnet(6 nets) -> interleave -> 13 fusion blocks -> unet, one timestep (export_onnx.py:43-74)"""
    from tests.helpers import tiny_step_check
    err, rel = tiny_step_check(DEV)
    assert err < 2e-2 and rel < 1e-2, (err, rel)


def test_hip_step_and_pipeline_vs_committed_golden(setup):
    """HIP path vs tests/golden/*.safetensors (written by tests/golden/make_golden.py from the oracle)."""
    import os
    from safetensors.torch import load_file
    from edgestyle_amd.models import StepRunner
    ucfg, vcfg, ws = setup
    gold = os.path.join(os.path.dirname(__file__), "golden")
    g = load_file(os.path.join(gold, "tiny_step.safetensors"))
    runner = StepRunner.from_state_dicts(ws, ucfg, torch.float16, DEV)
    out = runner.step_nchw(g["x"].to(DEV), 501, g["ehs"].to(DEV), [g[f"cond{i}"].to(DEV) for i in range(6)],
                           [1.0, 0.8, 1.0, 1.0, 0.5, 1.0])
    assert float((out.float().cpu() - g["noise_pred"]).abs().max()) < 2e-2


@pytest.mark.parametrize("mode", ["grouped", "serial"])
def test_full_step_chain_modes_match_oracle(mode):
    """The two execution modes of the four independent encoder chains (grouped lockstep launches / serial) against the oracle, at a latent size (64) where the groups tile in 128-pixel units so `grouped` really
    runs grouped launches (tiny width keeps the CPU oracle fast)."""
    import dataclasses
    from oracle import sd15_oracle as O
    from edgestyle_amd.models import StepRunner
    ucfg = dataclasses.replace(C.tiny_unet(), sample_size=64)
    ws = {k: quantize(v) for k, v in make_weights(ucfg, C.tiny_vae(), seed=5).items()}
    g = torch.Generator().manual_seed(4)
    N, s, c0 = 2, 64, ucfg.block_out_channels[0]
    x = torch.randn(N, 4, s, s, generator=g).half().float()
    ehs = (torch.randn(N, 77, ucfg.cross_attention_dim, generator=g) * 0.5).half().float()
    conds = [(torch.randn(N, c0, s, s, generator=g) * 0.3).half().float() for _ in range(6)]
    scales = [1.0, 0.7, 1.0, 1.0, 1.2, 1.0]
    torch.set_num_threads(16)
    ref = O.denoise_step(ws["unet"], ucfg, ws["fusion"], oracle_nets(ws, ucfg), x, 321, ehs, conds, scales)
    runner = StepRunner.from_state_dicts(ws, ucfg, torch.float16, DEV)
    runner.mode = mode
    out = runner.step_nchw(x.to(DEV), 321, ehs.to(DEV), [c.to(DEV) for c in conds], scales)
    if mode == "grouped":
        assert runner._grouped is not None and runner._grouped.groupable(64)
    err = float((out.float().cpu() - ref).abs().max())
    assert err < 2e-2, err


def _rand_sd(shapes, seed, scale_fn):
    g = torch.Generator().manual_seed(seed)
    return {k: (scale_fn(k, shp) * torch.randn(shp, generator=g) + (1.0 if k.endswith("norm.weight") or ".norm" in k and k.endswith(".weight") else 0.0)).half().float()
            for k, shp in shapes.items()}


@pytest.mark.parametrize("C,heads,H", [(320, 8, 32), (1280, 8, 8)])
def test_full_width_transformer_block_vs_oracle(C, heads, H):
    """One Transformer2DModel at SD1.5 widths (head_dim 40 / 160; LayerNorms folded into their GEMMs, 8-wave and 64x64
    tile launches, ones-column softmax) against the oracle block."""
    from oracle import sd15_oracle as O
    from edgestyle_amd import engine as E
    p, tb = "attentions.0", "attentions.0.transformer_blocks.0"
    D = 768
    shapes = {f"{p}.norm.weight": (C,), f"{p}.norm.bias": (C,), f"{p}.proj_in.weight": (C, C, 1, 1), f"{p}.proj_in.bias": (C,),
              f"{p}.proj_out.weight": (C, C, 1, 1), f"{p}.proj_out.bias": (C,)}
    for i in (1, 2, 3):
        shapes[f"{tb}.norm{i}.weight"] = (C,); shapes[f"{tb}.norm{i}.bias"] = (C,)
    for a, kin in (("attn1", C), ("attn2", D)):
        shapes[f"{tb}.{a}.to_q.weight"] = (C, C); shapes[f"{tb}.{a}.to_k.weight"] = (C, kin); shapes[f"{tb}.{a}.to_v.weight"] = (C, kin)
        shapes[f"{tb}.{a}.to_out.0.weight"] = (C, C); shapes[f"{tb}.{a}.to_out.0.bias"] = (C,)
    shapes[f"{tb}.ff.net.0.proj.weight"] = (8 * C, C); shapes[f"{tb}.ff.net.0.proj.bias"] = (8 * C,)
    shapes[f"{tb}.ff.net.2.weight"] = (C, 4 * C); shapes[f"{tb}.ff.net.2.bias"] = (C,)
    g = torch.Generator().manual_seed(C)
    sd = {}
    for k, shp in shapes.items():
        if "norm" in k and k.endswith(".weight"):
            sd[k] = (1 + 0.1 * torch.randn(shp, generator=g)).half().float()
        elif k.endswith(".bias"):
            sd[k] = (0.05 * torch.randn(shp, generator=g)).half().float()
        else:
            sd[k] = (torch.randn(shp, generator=g) / (shp[1] ** 0.5)).half().float()
    x = (torch.randn(2, C, H, H, generator=g) * 1.5 + 0.3).half().float()
    ehs = (torch.randn(2, 77, D, generator=g) * 0.5).half().float()
    ref = O.transformer(sd, p, x, ehs, heads, 32)
    pk = E._Packer(sd, torch.float16, DEV)
    blk = E.Transformer(pk, p, heads, 32)
    assert blk.ln_fold
    kv = blk.context(ehs.to(DEV, torch.float16))
    out = blk(to_nhwc(x, DEV), kv)
    assert rel_err(out.permute(0, 3, 1, 2), ref) < 1e-2


def test_full_width_resnet_block_vs_oracle():
    """ResnetBlock2D 320 -> 640 with shortcut, time embedding and a concatenated skip input (decoder form)."""
    from oracle import sd15_oracle as O
    from edgestyle_amd import engine as E
    g = torch.Generator().manual_seed(9)
    C1, C2, Cout, H, p = 320, 320, 640, 32, "r"
    sd = {f"{p}.norm1.weight": 1 + 0.1 * torch.randn(C1 + C2, generator=g), f"{p}.norm1.bias": 0.05 * torch.randn(C1 + C2, generator=g),
          f"{p}.conv1.weight": torch.randn(Cout, C1 + C2, 3, 3, generator=g) / (9 * (C1 + C2)) ** 0.5, f"{p}.conv1.bias": 0.05 * torch.randn(Cout, generator=g),
          f"{p}.time_emb_proj.weight": torch.randn(Cout, 1280, generator=g) / 36, f"{p}.time_emb_proj.bias": 0.05 * torch.randn(Cout, generator=g),
          f"{p}.norm2.weight": 1 + 0.1 * torch.randn(Cout, generator=g), f"{p}.norm2.bias": 0.05 * torch.randn(Cout, generator=g),
          f"{p}.conv2.weight": torch.randn(Cout, Cout, 3, 3, generator=g) / (9 * Cout) ** 0.5, f"{p}.conv2.bias": 0.05 * torch.randn(Cout, generator=g),
          f"{p}.conv_shortcut.weight": torch.randn(Cout, C1 + C2, 1, 1, generator=g) / (C1 + C2) ** 0.5, f"{p}.conv_shortcut.bias": 0.05 * torch.randn(Cout, generator=g)}
    sd = {k: v.half().float() for k, v in sd.items()}
    x1 = torch.randn(2, C1, H, H, generator=g).half().float()
    x2 = torch.randn(2, C2, H, H, generator=g).half().float()
    temb = torch.randn(2, 1280, generator=g).half().float()
    ref = O.resnet(sd, p, torch.cat([x1, x2], 1), temb, 32, 1e-5)
    pk = E._Packer(sd, torch.float16, DEV)
    blk = E.Resnet(pk, p, 32, 1e-5, 0)
    from edgestyle_amd import ops
    tproj = ops.linear(torch.nn.functional.silu(temb).to(DEV, torch.float16), pk.conv(f"{p}.time_emb_proj"))
    out = blk(to_nhwc(x1, DEV), tproj, x2=to_nhwc(x2, DEV))
    assert rel_err(out.permute(0, 3, 1, 2), ref) < 1e-2
