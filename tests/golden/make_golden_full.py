#!/usr/bin/env python
"""Full-size golden fixtures (SD1.5 width, 512x512 = the benchmarked configuration) from the CPU oracle.

Same caveat as make_golden.py: the reference cannot be imported offline and holds no vectors, so these pin the
ORACLE RESTATEMENT at the benchmarked size (parity unpinned against diffusers itself).  Inputs and weights are
regenerated from seeds by tests/helpers.py (full_weights / full_*_inputs); only the oracle's outputs are stored:

  full_step.safetensors          one 6-cond CFG step == export_onnx.py:43-74 (noise_pred + a corner of each of the
                                 13 fused residual tensors)
  full_pipeline4.safetensors     BASELINE configs[1] geometry, 4 DDIM steps, CFG 7.5: final latents + decoded image
  single_cn_pipeline4.safetensors  BASELINE configs[0]: UNet + 1 openpose ControlNet, 4 DDIM steps, fp32 oracle

    python tests/golden/make_golden_full.py          (about 10 minutes on 8 cores, 12 GB of RAM)
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


def main():
    torch.set_num_threads(int(os.environ.get("ES_THREADS", "8")))
    t0 = time.time()
    ucfg, vcfg, ws = H.full_weights()
    print(f"weights {time.time() - t0:.0f}s", flush=True)
    nets = H.oracle_nets(ws, ucfg)
    with torch.no_grad():
        x, ehs, conds = H.full_step_inputs()
        down, mid = O.multicontrolnet_forward(ws["fusion"], nets, x, H.FULL_STEP_T, ehs, conds, H.FULL_STEP_SCALES)
        noise = O.unet_forward(ws["unet"], ucfg, x, H.FULL_STEP_T, ehs, down, mid)
        out = {"noise_pred": noise.contiguous()}
        for i, d in enumerate(down + [mid]):
            out[f"fused{i}"] = d[:, :32, :8, :8].contiguous()
        save_file(out, os.path.join(HERE, "full_step.safetensors"))
        print(f"step {time.time() - t0:.0f}s", flush=True)

        lat, pe, ne, pc = H.full_pipeline_inputs()
        lat_out = O.pipeline(ws["unet"], ucfg, ws["fusion"], nets, ws["vae"], vcfg, lat, pe, ne,
                             [c.repeat(2, 1, 1, 1) for c in pc], num_inference_steps=4, guidance_scale=7.5, decode=False)
        img = (O.vae_decode(ws["vae"], vcfg, lat_out / vcfg.scaling_factor) / 2 + 0.5).clamp(0, 1)
        save_file({"latents_out": lat_out.contiguous(), "image": img.half().contiguous()},
                  os.path.join(HERE, "full_pipeline4.safetensors"))
        print(f"pipeline4 {time.time() - t0:.0f}s", flush=True)

        lat, pe, ne, pose = H.single_cn_inputs()
        lat_out = O.pipeline(ws["unet"], ucfg, None, [(ws["openpose"], ucfg)], ws["vae"], vcfg, lat, pe, ne,
                             [torch.cat([pose] * 2)], num_inference_steps=4, guidance_scale=7.5, decode=False)
        img = (O.vae_decode(ws["vae"], vcfg, lat_out / vcfg.scaling_factor) / 2 + 0.5).clamp(0, 1)
        save_file({"latents_out": lat_out.contiguous(), "image": img.half().contiguous()},
                  os.path.join(HERE, "single_cn_pipeline4.safetensors"))
        print(f"single-cn pipeline4 {time.time() - t0:.0f}s", flush=True)


if __name__ == "__main__":
    main()
