#!/usr/bin/env python
"""Run-to-run determinism of the whole call at the benchmarked sizes: the same request served repeatedly must give the same bits every time
(no atomics in any data path, fixed-order reductions, counted waits): batch 1 x 8 calls, batch 8 x 5, 768 x 768 bf16 batch 4 x 3."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

dev = torch.device("cuda", 0)
for res, dtype, cases in ((512, torch.float16, ((1, 8), (8, 5))), (768, torch.bfloat16, ((4, 3),))):
    pipe, ws, ucfg, vcfg = bench.build_pipeline(dev, dtype, tiny=False, resolution=res)
    for B, n in cases:
        lat, pe, ne, imgs, cn = bench.make_inputs(ucfg, vcfg, B, dev, seed=42)
        ref = None
        t0 = time.time()
        for i in range(n):
            img = pipe(prompt_embeds=pe, negative_prompt_embeds=ne, image=imgs, latents=lat, guidance_scale=7.5, num_inference_steps=50,
                       output_type="pt", cond_noise=cn).images
            torch.cuda.synchronize()
            assert bool(torch.isfinite(img).all())
            if ref is None:
                ref = img.clone()
            else:
                assert torch.equal(img, ref), f"{res} batch {B}: call {i} differs from call 0 (max abs {float((img - ref).abs().max())})"
        print(f"{res}x{res} {str(dtype)[6:]} batch {B}: {n} calls of 50 steps bitwise equal ({time.time() - t0:.1f} s)", flush=True)
    pipe._loops.clear(); pipe._runner = None
    del pipe
    torch.cuda.empty_cache()
print("determinism soak ok")
