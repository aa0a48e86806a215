#!/usr/bin/env python
"""Round-3 full-size fixtures from the CPU oracle: the HEADLINE configuration end to end, and the RGB-image path.

Same caveat as make_golden_full.py: the reference cannot be imported offline (no diffusers) and holds no vectors, so
these pin the ORACLE RESTATEMENT at the benchmarked size (parity unpinned against diffusers itself).  Inputs and
weights are regenerated from seeds by tests/helpers.py; only the oracle's outputs are stored:

  full_pipeline50.safetensors    BASELINE configs[1]: 512x512, 50 DDIM steps, CFG 7.5, batch 1 (the loop PL:435-543 as
                                 TT:357 drives it): final latents, fp16 image, and the latents after steps 1, 5, 10, 25
                                 (error growth over the loop can be read off the GPU run)
  full_rgb_pipeline2.safetensors the one-time condition embedding at its real size (PL:629-664, CL:28-42, CL:289-290):
                                 six [1,3,512,512] RGB condition images -> 3 x VAE encode + sample (128 ch @ 512x512) and
                                 3 x openpose conv stack on the CFG-duplicated batch -> 2 DDIM steps -> decode: the six
                                 embeddings (a corner each + their full-tensor mean/abs-mean), final latents, fp16 image

    ES_THREADS=4 python tests/golden/make_golden_full50.py [rgb|p50|all]      (p50: ~1 h on 4 cores, 14 GB of RAM)
"""
import os
import sys
import time

import torch
from safetensors.torch import save_file

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import sd15_oracle as O                       # noqa: E402
from tests import helpers as H                            # noqa: E402


def rgb_case(ws, ucfg, vcfg, nets, t0):
    imgs, noise, lat, pe, ne = H.full_rgb_inputs()
    conds = []
    for i, (sd, _) in enumerate(nets):
        x2 = torch.cat([imgs[i]] * 2)                                          # PL:657-658 CFG duplicate, then embed
        if i % 2 == 0:
            conds.append(O.vae_cond_embedding(sd, ws["vae"], vcfg, x2, noise[i]))   # CL:38-42
        else:
            conds.append(O.cond_embedding(sd, ucfg, x2))                        # [D] ControlNetConditioningEmbedding
        print(f"  cond {i} {time.time() - t0:.0f}s", flush=True)
    lat_out = O.pipeline(ws["unet"], ucfg, ws["fusion"], nets, ws["vae"], vcfg, lat, pe, ne, conds,
                         num_inference_steps=2, guidance_scale=7.5, decode=False)
    img = (O.vae_decode(ws["vae"], vcfg, lat_out / vcfg.scaling_factor) / 2 + 0.5).clamp(0, 1)
    out = {"latents_out": lat_out.contiguous(), "image": img.half().contiguous()}
    for i, c in enumerate(conds):
        out[f"cond{i}"] = c[:, :64, :16, :16].contiguous()
        out[f"cond{i}_stats"] = torch.stack([c.mean(), c.abs().mean(), c.abs().max()])
    save_file(out, os.path.join(HERE, "full_rgb_pipeline2.safetensors"))
    print(f"rgb pipeline2 {time.time() - t0:.0f}s", flush=True)


def p50_case(ws, ucfg, vcfg, nets, t0):
    lat, pe, ne, pc = H.full_pipeline_inputs(seed=48)
    keep = {}

    def on_step(i, t, latents, eps):
        if i + 1 in (1, 5, 10, 25):
            keep[f"latents_step{i + 1}"] = latents.clone().contiguous()
        print(f"  step {i + 1}/50 {time.time() - t0:.0f}s", flush=True)

    lat_out = O.pipeline(ws["unet"], ucfg, ws["fusion"], nets, ws["vae"], vcfg, lat, pe, ne,
                         [c.repeat(2, 1, 1, 1) for c in pc], num_inference_steps=50, guidance_scale=7.5, decode=False,
                         on_step=on_step)
    img = (O.vae_decode(ws["vae"], vcfg, lat_out / vcfg.scaling_factor) / 2 + 0.5).clamp(0, 1)
    keep.update({"latents_out": lat_out.contiguous(), "image": img.half().contiguous()})
    save_file(keep, os.path.join(HERE, "full_pipeline50.safetensors"))
    print(f"pipeline50 {time.time() - t0:.0f}s", flush=True)


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    torch.set_num_threads(int(os.environ.get("ES_THREADS", "8")))
    t0 = time.time()
    ucfg, vcfg, ws = H.full_weights()
    print(f"weights {time.time() - t0:.0f}s", flush=True)
    nets = H.oracle_nets(ws, ucfg)
    with torch.no_grad():
        if what in ("rgb", "all"):
            rgb_case(ws, ucfg, vcfg, nets, t0)
        if what in ("p50", "all"):
            p50_case(ws, ucfg, vcfg, nets, t0)


if __name__ == "__main__":
    main()
