#!/usr/bin/env python
"""Fixed per-image cost of a pipeline call (condition embedding + host prep + VAE decode) vs the per-step cost:
time calls at 10 / 30 / 50 steps and fit a line."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    pipe, ws, ucfg, vcfg = bench.build_pipeline(dev, torch.float16, tiny=False, resolution=512)
    lat, pe, ne, imgs, cn = bench.make_inputs(ucfg, vcfg, 1, dev, seed=42)
    res = {}
    for steps in (10, 30, 50, 10, 30, 50):
        def one():
            return pipe(prompt_embeds=pe, negative_prompt_embeds=ne, image=imgs, latents=lat, guidance_scale=7.5,
                        num_inference_steps=steps, output_type="pt", cond_noise=cn).images
        one(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            one()
        torch.cuda.synchronize()
        res[steps] = (time.perf_counter() - t0) / 3 * 1e3
        print(f"steps={steps}: {res[steps]:.1f} ms", flush=True)
    per_step = (res[50] - res[10]) / 40
    print(f"per step {per_step:.2f} ms, fixed per image {res[50] - 50 * per_step:.1f} ms")
    # split the fixed part: conds / decode alone
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        conds = [pipe.prepare_image(img, 1, True, net, noise=cn[i]) for i, (img, net) in enumerate(zip(imgs, pipe.controlnet.nets))]
    torch.cuda.synchronize()
    print(f"condition embedding (6 nets): {(time.perf_counter() - t0) / 5 * 1e3:.1f} ms")
    loop = list(pipe._loops.values())[-1]
    t0 = time.perf_counter()
    for _ in range(5):
        dec = pipe.vae.decode_nhwc(loop.model_in[:1], unscaled_latents=True)
    torch.cuda.synchronize()
    print(f"VAE decode: {(time.perf_counter() - t0) / 5 * 1e3:.1f} ms")


if __name__ == "__main__":
    main()
