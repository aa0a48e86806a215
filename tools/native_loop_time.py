#!/usr/bin/env python
"""es_denoise_loop + es_vae_decode at the benchmarked size: per-plan graphs (50 step-graph launches) vs the whole loop as one
hipGraph vs the Python pipeline, ms per image."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from edgestyle_amd.models import _as_nhwc  # noqa: E402
from edgestyle_amd.native import NativeEngine  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    pipe, ws, ucfg, vcfg = bench.build_pipeline(dev, torch.float16)
    lat, pe, ne, imgs, cn = bench.make_inputs(ucfg, vcfg, 1, dev)
    kw = dict(prompt_embeds=pe, negative_prompt_embeds=ne, image=imgs, latents=lat, guidance_scale=7.5, num_inference_steps=50,
              output_type="pt", cond_noise=cn)
    want = pipe(**kw).images

    def timeit(fn, n=6):
        fn(); fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3
    print(f"python pipeline (cond embedding + loop + decode): {timeit(lambda: pipe(**kw)):.1f} ms per image", flush=True)
    eng = NativeEngine(pipe, batch_size=1, num_inference_steps=50)
    ehs = torch.cat([ne, pe]).to(dev, torch.float16).contiguous()
    x = lat.permute(0, 2, 3, 1).contiguous().to(dev)
    nz = [None if z is None else z.to(dev) for z in cn]
    out = {}

    def native():
        eng.prepare_conds(imgs, nz)
        out["img"] = eng.vae_decode(eng.denoise_loop(x.clone(), ehs, 7.5))
    for mode in (1, 2):
        eng.set_options(use_graphs=mode)
        ms = timeit(native)
        print(f"native, use_graphs={mode}: {ms:.1f} ms per image; equal to the pipeline bit for bit: {bool(torch.equal(out['img'], want))}", flush=True)
    eng.close()


if __name__ == "__main__":
    main()
