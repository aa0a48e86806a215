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
    err = tiny_step_check(DEV)
    assert err < 2e-2, err


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


@pytest.mark.parametrize("mode", ["grouped", "streams", "serial"])
def test_full_step_chain_modes_match_oracle(mode):
    """The three execution modes of the four independent encoder chains (grouped lockstep launches / parallel streams /
    serial) against the oracle, at a latent size (64) where the groups tile in 128-pixel units so `grouped` really
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
