"""CPU suite (-m "not gpu"): the oracle against the committed golden vectors and against independent restatements of
the reference-owned arithmetic; layouts against the published SD1.5 parameter counts."""
import json
import math
import os

import pytest
import torch
import torch.nn as nn
from safetensors.torch import load_file

from edgestyle_amd import config as C, weights as W
from oracle import sd15_oracle as O
from tests.helpers import make_weights, quantize, oracle_nets

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_layouts_match_published_sd15_parameter_counts():
    n = lambda sh: sum(math.prod(s) for s in sh.values())
    assert n(W.unet_shapes(C.sd15_unet())) == 859_520_964            # SD1.5 UNet2DConditionModel
    assert n(W.controlnet_shapes(C.sd15_unet())) == 361_279_120      # SD1.5 ControlNetModel (control_v11p_*)
    assert n(W.vae_shapes(C.sd15_vae())) == 83_653_863               # AutoencoderKL (sd-vae-ft-mse)


def test_residual_table_is_the_reference_hard_coded_table():
    # model/edgestyle_multicontrolnet.py:73-102
    ch = [320, 320, 320, 320, 640, 640, 640, 1280, 1280, 1280, 1280, 1280]
    sz = [64, 64, 64, 32, 32, 32, 16, 16, 16, 8, 8, 8]
    t = C.sd15_unet().residual_table()
    assert [c for c, _ in t[:-1]] == ch and [s for _, s in t[:-1]] == sz and t[-1] == (1280, 8)
    fs = W.fusion_shapes(C.sd15_unet())
    assert fs["multi_controlnet_down_blocks.0.first_normalization.weight"] == (960, 64, 64)     # MC:34-36
    assert fs["multi_controlnet_mid_block.second_normalization.bias"] == (1280, 8, 8)           # MC:44-46
    assert fs["multi_controlnet_down_blocks.4.first_conv.weight"] == (1920, 2, 1, 1)            # MC:28-33


class _RefControlNetBlock(nn.Module):
    """Module-for-module restatement of model/edgestyle_multicontrolnet.py:23-63 with torch.nn layers."""

    def __init__(self, c, size, n):
        super().__init__()
        self.first_conv = nn.Conv2d(c * n, c * n // 2, 1, groups=c * n // 2)
        self.first_normalization = nn.LayerNorm([c * n // 2, *size])
        self.activation = nn.SiLU()
        self.second_conv = nn.Conv2d(c * n // 2, c, 1, groups=c)
        self.second_normalization = nn.LayerNorm([c, *size])
        self.third_conv = nn.Conv2d(c, c, 1, groups=c)

    def forward(self, x):
        x = self.activation(self.first_normalization(self.first_conv(x)))
        x = self.activation(self.second_normalization(self.second_conv(x)))
        return self.third_conv(x)


def test_fusion_block_and_interleave_vs_module_restatement_and_closed_form():
    torch.manual_seed(0)
    c, s, n, b = 8, 4, 6, 2
    m = _RefControlNetBlock(c, (s, s), n)
    for p in m.parameters():
        nn.init.normal_(p, 0.3, 0.5)
    sd = {"blk." + k: v.detach() for k, v in m.state_dict().items()}
    res = [torch.randn(b, c, s, s) for _ in range(n)]
    inter = O.interleave_tensors(res)
    for ci in range(c):                       # MC:494-500: channel = c*6 + net
        for ni in range(n):
            assert torch.equal(inter[:, ci * n + ni], res[ni][:, ci])
    y = O.controlnet_block(sd, "blk", inter)
    assert torch.allclose(y, m(inter), atol=1e-5)
    # closed form of SURVEY §3.3 for one output element
    w1, b1 = sd["blk.first_conv.weight"], sd["blk.first_conv.bias"]
    z = torch.stack([w1[3 * ci + p, 0, 0, 0] * res[2 * p][:, ci] + w1[3 * ci + p, 1, 0, 0] * res[2 * p + 1][:, ci]
                     + b1[3 * ci + p] for ci in range(c) for p in range(3)], dim=1)
    mu, var = z.mean(dim=(1, 2, 3), keepdim=True), z.var(dim=(1, 2, 3), unbiased=False, keepdim=True)
    y1 = torch.nn.functional.silu((z - mu) / (var + 1e-5).sqrt() * sd["blk.first_normalization.weight"]
                                  + sd["blk.first_normalization.bias"])
    w2 = sd["blk.second_conv.weight"]
    u = torch.stack([sum(w2[ci, p, 0, 0] * y1[:, 3 * ci + p] for p in range(3)) + sd["blk.second_conv.bias"][ci]
                     for ci in range(c)], dim=1)
    mu, var = u.mean(dim=(1, 2, 3), keepdim=True), u.var(dim=(1, 2, 3), unbiased=False, keepdim=True)
    v = torch.nn.functional.silu((u - mu) / (var + 1e-5).sqrt() * sd["blk.second_normalization.weight"]
                                 + sd["blk.second_normalization.bias"])
    out = v * sd["blk.third_conv.weight"].view(1, c, 1, 1) + sd["blk.third_conv.bias"].view(1, c, 1, 1)
    assert torch.allclose(y, out, atol=1e-4)


def test_oracle_fusion_equals_the_references_own_code():
    """PINNED: tests/golden/ref_fusion.safetensors holds outputs of the REFERENCE's own `ControlNetBlock`, `interleave_tensors`
    and `interleave_tensors_from_list_of_lists` (model/edgestyle_multicontrolnet.py:23-63, 479-514; executed from its source text
    by tests/golden/make_golden_ref_fusion.py in the build container - the three definitions use torch / torch.nn only).  The
    oracle's restatement must reproduce them for all 13 (channels, size) pairs of MC:73-102 at batch 2: sampled elements to 2e-5
    of the tensor's scale and the full-tensor sums (float64).  This pins the arithmetic the reference owns; the diffusers-owned
    blocks stay unpinned."""
    from safetensors.torch import load_file
    from tests import helpers as H
    gold = load_file(os.path.join(GOLD, "ref_fusion.safetensors"))
    a, b = H.ref_interleave_cases()
    assert torch.equal(O.interleave_tensors(a), gold["interleave_a"])
    assert torch.equal(O.interleave_tensors(b), gold["interleave_lists_1"]) and torch.equal(gold["interleave_lists_0"], gold["interleave_a"])
    for i in range(len(H.REF_FUSION_LEVELS)):
        sd, res = H.ref_fusion_case(i)
        y = O.controlnet_block({"b." + k: v for k, v in sd.items()}, "b", O.interleave_tensors(res)).double()
        want = gold[f"level{i}_sample"].double()
        assert float((H.ref_fusion_sample(y) - want).abs().max()) <= 2e-5 * float(want.abs().max()), i
        sums = gold[f"level{i}_sums"]
        assert abs(float(y.sum()) - float(sums[0])) <= 1e-6 * float(sums[1]) and abs(float(y.abs().sum()) - float(sums[1])) <= 1e-6 * float(sums[1]), i


def test_ddim_known_answers():
    g = json.load(open(os.path.join(GOLD, "ddim.json")))
    s = O.DDIM()
    ts = s.set_timesteps(50)
    assert ts.tolist() == g["timesteps_50"] == list(range(981, 0, -20))       # leading spacing, steps_offset=1
    assert abs(float(s.alphas_cumprod[0]) - 0.99915) < 1e-6                     # 1 - 0.00085
    assert abs(float(s.alphas_cumprod[999]) - 0.00466) < 1e-4                   # SD1.5 terminal alpha_bar
    assert abs(float(s.alphas_cumprod[981]) - g["alphas_cumprod_981"]) < 1e-9
    # eta=0 DDIM with eps == true noise recovers x0 exactly at the last step
    x0 = torch.randn(1, 4, 8, 8)
    eps = torch.randn(1, 4, 8, 8)
    a1 = s.alphas_cumprod[1]
    xt = a1.sqrt() * x0 + (1 - a1).sqrt() * eps
    a0 = s.final_alpha_cumprod
    assert torch.allclose(s.step(eps, 1, xt), a0.sqrt() * x0 + (1 - a0).sqrt() * eps, atol=1e-5)


def test_timestep_sinusoid_layout():
    e = O.timestep_sinusoid(torch.tensor([0.0, 10.0]), 320)
    assert e.shape == (2, 320)
    assert torch.allclose(e[0, :160], torch.ones(160)) and torch.allclose(e[0, 160:], torch.zeros(160))   # [cos|sin]
    assert abs(float(e[1, 160]) - math.sin(10.0)) < 1e-6 and abs(float(e[1, 159]) - math.cos(10.0 * math.exp(-math.log(1e4) * 159 / 160))) < 1e-6


@pytest.fixture(scope="module")
def tiny():
    torch.set_num_threads(8)
    ucfg, vcfg = C.tiny_unet(), C.tiny_vae()
    return ucfg, vcfg, {k: quantize(v) for k, v in make_weights(ucfg, vcfg).items()}


def test_oracle_reproduces_golden_step(tiny):
    ucfg, vcfg, ws = tiny
    g = load_file(os.path.join(GOLD, "tiny_step.safetensors"))
    conds = [g[f"cond{i}"] for i in range(6)]
    nets = oracle_nets(ws, ucfg)
    down, mid = O.multicontrolnet_forward(ws["fusion"], nets, g["x"], 501, g["ehs"], conds, [1.0, 0.8, 1.0, 1.0, 0.5, 1.0])
    assert torch.allclose(mid, g["fused_mid"], atol=2e-4) and torch.allclose(down[0], g["fused_down0"], atol=2e-4)
    noise = O.unet_forward(ws["unet"], ucfg, g["x"], 501, g["ehs"], down, mid)
    assert torch.allclose(noise, g["noise_pred"], atol=5e-4), float((noise - g["noise_pred"]).abs().max())
    # the step function is exactly controlnet -> unet (export_onnx.py:43-74)
    assert torch.allclose(O.denoise_step(ws["unet"], ucfg, ws["fusion"], nets, g["x"], 501, g["ehs"], conds,
                                         [1.0, 0.8, 1.0, 1.0, 0.5, 1.0]), noise, atol=1e-6)


def test_lora_fused_equals_unfused_and_tying_is_aliasing(tiny):
    ucfg, _, ws = tiny
    tied = O.tie_weights(ws["lora0"], ws["unet"])
    assert tied["down_blocks.0.resnets.0.conv1.weight"] is ws["unet"]["down_blocks.0.resnets.0.conv1.weight"]   # CL:45-56
    assert "up_blocks.0.resnets.0.conv1.weight" not in tied
    fused = O.fuse_lora(tied)
    assert not any(".lora_layer." in k for k in fused)
    k = "down_blocks.0.attentions.0.transformer_blocks.0.attn1.to_q"
    assert not torch.equal(fused[k + ".weight"], tied[k + ".weight"])
    assert torch.equal(tied[k + ".weight"], ws["unet"][k + ".weight"])          # the UNet tensor was not mutated
    g = torch.Generator().manual_seed(2)
    s, c0 = ucfg.sample_size, ucfg.block_out_channels[0]
    x = torch.randn(1, 4, s, s, generator=g)
    e = torch.randn(1, 77, ucfg.cross_attention_dim, generator=g)
    c = torch.randn(1, c0, s, s, generator=g)
    d1, m1 = O.controlnet_forward(tied, ucfg, x, 300, e, c, 0.7)
    d2, m2 = O.controlnet_forward(fused, ucfg, x, 300, e, c, 0.7)
    assert max(float((a - b).abs().max()) for a, b in zip(d1 + [m1], d2 + [m2])) < 2e-4
    # conditioning_scale is a plain multiply (CL:266-270) and the cached-cond shortcut skips the embedding (CL:199-203)
    d3, m3 = O.controlnet_forward(tied, ucfg, x, 300, e, c, 1.0)
    assert torch.allclose(m1, m3 * 0.7, atol=1e-5)


def test_cfg_and_guidance_window_semantics(tiny):
    """guidance_scale <= 1 disables CFG (batch B, PL:329-330); control_guidance_end gates the nets per step (PL:419-427)"""
    ucfg, vcfg, ws = tiny
    g = load_file(os.path.join(GOLD, "tiny_pipeline4.safetensors"))
    nets = oracle_nets(ws, ucfg)
    conds2 = [g[f"cond{i}"].repeat(2, 1, 1, 1) for i in range(6)]
    seen = []
    out = O.pipeline(ws["unet"], ucfg, ws["fusion"], nets, ws["vae"], vcfg, g["latents_in"], g["prompt_embeds"],
                     g["negative_prompt_embeds"], conds2, num_inference_steps=4, guidance_scale=7.5, decode=False,
                     on_step=lambda i, t, l, e: seen.append(t))
    assert seen == [751, 501, 251, 1]
    assert torch.allclose(out, g["latents_out"], atol=2e-3), float((out - g["latents_out"]).abs().max())
