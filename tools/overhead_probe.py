#!/usr/bin/env python
"""Per-image fixed cost of the batch-1 pipeline: wall time at several step counts (slope = per step, intercept = condition
embedding + host preparation + VAE decode), with and without the VAE decode."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    pipe, ws, ucfg, vcfg = bench.build_pipeline(dev, torch.float16)
    lat, pe, ne, imgs, cn = bench.make_inputs(ucfg, vcfg, 1, dev)

    def run(steps, out):
        best = 1e9
        for _ in range(4):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            pipe(prompt_embeds=pe, negative_prompt_embeds=ne, image=imgs, latents=lat, guidance_scale=7.5,
                 num_inference_steps=steps, output_type=out, cond_noise=cn)
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) * 1e3)
        return best
    for out in ("pt", "latent"):
        t = {s: run(s, out) for s in (10, 30, 50)}
        slope = (t[50] - t[10]) / 40
        print(f"output_type={out}: " + " ".join(f"{s} steps {v:.1f} ms" for s, v in t.items())
              + f" | per step {slope:.3f} ms, intercept {t[10] - 10 * slope:.2f} ms", flush=True)


if __name__ == "__main__":
    main()
