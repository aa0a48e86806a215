#!/usr/bin/env python
"""Writes the golden fixtures of tests/golden/ from the CPU oracle (oracle/sd15_oracle.py) on seeded inputs.

The reference itself cannot be imported here (its diffusers dependency is absent and cannot be installed offline,
SURVEY.md §8c) and ships no golden tensors, so these vectors pin the ORACLE RESTATEMENT against regressions and give
the GPU box (which has no /root/reference) fixed expected outputs; they do not pin parity with diffusers itself.
Weights are NOT stored: they are regenerated bit-identically from crc32(key)^seed (edgestyle_amd/weights.py).

    python tests/golden/make_golden.py
"""
import json
import os
import sys

import torch
from safetensors.torch import save_file

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from edgestyle_amd import config as C                     # noqa: E402
from oracle import sd15_oracle as O                       # noqa: E402
from tests.helpers import make_weights, quantize, oracle_nets   # noqa: E402


def main():
    torch.set_num_threads(8)
    ucfg, vcfg = C.tiny_unet(), C.tiny_vae()
    ws = {k: quantize(v) for k, v in make_weights(ucfg, vcfg, seed=0).items()}
    g = torch.Generator().manual_seed(42)
    N, s, c0 = 2, ucfg.sample_size, ucfg.block_out_channels[0]
    x = torch.randn(N, 4, s, s, generator=g).half().float()
    ehs = (torch.randn(N, 77, ucfg.cross_attention_dim, generator=g) * 0.5).half().float()
    conds = [(torch.randn(N, c0, s, s, generator=g) * 0.3).half().float() for _ in range(6)]
    scales = [1.0, 0.8, 1.0, 1.0, 0.5, 1.0]
    nets = oracle_nets(ws, ucfg)
    down, mid = O.multicontrolnet_forward(ws["fusion"], nets, x, 501, ehs, conds, scales)
    noise = O.unet_forward(ws["unet"], ucfg, x, 501, ehs, down, mid)
    out = {"x": x, "ehs": ehs, "noise_pred": noise, "fused_mid": mid, "fused_down0": down[0], "fused_down11": down[11]}
    out.update({f"cond{i}": c for i, c in enumerate(conds)})
    save_file({k: v.contiguous() for k, v in out.items()}, os.path.join(HERE, "tiny_step.safetensors"))

    # 4-step pipeline (BASELINE config 1 analogue at tiny width): final latents + decoded image
    lat = torch.randn(1, 4, s, s, generator=g)
    pe = (torch.randn(1, 77, ucfg.cross_attention_dim, generator=g) * 0.5).half().float()
    ne = (torch.randn(1, 77, ucfg.cross_attention_dim, generator=g) * 0.5).half().float()
    pc = [(torch.randn(1, c0, s, s, generator=g) * 0.3).half().float() for _ in range(6)]
    lat_out = O.pipeline(ws["unet"], ucfg, ws["fusion"], nets, ws["vae"], vcfg, lat, pe, ne,
                         [c.repeat(2, 1, 1, 1) for c in pc], num_inference_steps=4, guidance_scale=7.5, decode=False)
    img = (O.vae_decode(ws["vae"], vcfg, lat_out / vcfg.scaling_factor) / 2 + 0.5).clamp(0, 1)
    p = {"latents_in": lat, "prompt_embeds": pe, "negative_prompt_embeds": ne, "latents_out": lat_out, "image": img}
    p.update({f"cond{i}": c for i, c in enumerate(pc)})
    save_file({k: v.contiguous() for k, v in p.items()}, os.path.join(HERE, "tiny_pipeline4.safetensors"))

    # DDIM schedule known answers (SD1.5 scheduler_config semantics)
    sch = O.DDIM()
    ts = sch.set_timesteps(50)
    json.dump({"timesteps_50": ts.tolist(), "alphas_cumprod_0": float(sch.alphas_cumprod[0]),
               "alphas_cumprod_981": float(sch.alphas_cumprod[981]), "alphas_cumprod_999": float(sch.alphas_cumprod[999])},
              open(os.path.join(HERE, "ddim.json"), "w"), indent=1)
    print("golden written:", os.listdir(HERE))


if __name__ == "__main__":
    main()
