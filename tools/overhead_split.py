#!/usr/bin/env python
"""Per-image time = overhead (condition embedding, preparation, VAE decode) + steps x step time: batch-1 pipeline at 10 / 30 / 50 steps."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

dev = torch.device("cuda", 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
pipe, ws, ucfg, vcfg = bench.build_pipeline(dev, torch.float16)
lat, pe, ne, imgs, cn = bench.make_inputs(ucfg, vcfg, B, dev)
res = {}
for steps in (10, 30, 50, 10, 30, 50):
    ts = []
    for i in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        pipe(prompt_embeds=pe, negative_prompt_embeds=ne, image=imgs, latents=lat, guidance_scale=7.5, num_inference_steps=steps, output_type="pt", cond_noise=cn)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    res[steps] = min(ts[1:])
step = (res[50] - res[10]) / 40
print(f"batch {B}: 10 steps {res[10]:.1f} ms, 30 steps {res[30]:.1f}, 50 steps {res[50]:.1f}: step {step:.3f} ms, per-call overhead {res[10] - 10 * step:.1f} ms", flush=True)
