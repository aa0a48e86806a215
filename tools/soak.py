#!/usr/bin/env python
"""Soak: alternate batch sizes / step counts / schedulers / guess_mode on one pipeline object and watch device memory."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from edgestyle_amd.schedulers import DDIMScheduler, UniPCMultistepScheduler  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    pipe, ws, ucfg, vcfg = bench.build_pipeline(dev, torch.float16, tiny=False, resolution=512)
    inputs = {B: bench.make_inputs(ucfg, vcfg, B, dev, seed=42) for B in (1, 2)}
    ref = {}
    for it in range(12):
        B = 1 + it % 2
        steps = (10, 20)[(it // 2) % 2]
        unipc = (it // 4) % 2 == 1
        guess = it % 5 == 4
        pipe.scheduler = UniPCMultistepScheduler() if unipc else DDIMScheduler()
        lat, pe, ne, imgs, cn = inputs[B]
        img = pipe(prompt_embeds=pe, negative_prompt_embeds=ne, image=imgs, latents=lat, guidance_scale=7.5,
                   num_inference_steps=steps, output_type="pt", cond_noise=cn, guess_mode=guess).images
        torch.cuda.synchronize()
        assert bool(torch.isfinite(img).all())
        key = (B, steps, unipc, guess)
        if key in ref:
            assert torch.equal(ref[key], img), key          # same inputs -> same image, whatever ran in between
        ref[key] = img.clone()
        print(f"it {it}: B={B} steps={steps} unipc={unipc} guess={guess}  mem allocated {torch.cuda.memory_allocated() / 2**30:.2f} GiB"
              f" reserved {torch.cuda.memory_reserved() / 2**30:.2f} GiB", flush=True)
    for it in range(12, 20):
        B = 1 + it % 2
        lat, pe, ne, imgs, cn = inputs[B]
        pipe.scheduler = DDIMScheduler()
        img = pipe(prompt_embeds=pe, negative_prompt_embeds=ne, image=imgs, latents=lat, guidance_scale=7.5,
                   num_inference_steps=10, output_type="pt", cond_noise=cn).images
        assert torch.equal(ref[(B, 10, False, False)], img)
    print("soak ok; reserved", round(torch.cuda.memory_reserved() / 2**30, 2), "GiB")


if __name__ == "__main__":
    main()
