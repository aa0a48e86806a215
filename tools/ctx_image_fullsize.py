#!/usr/bin/env python
"""The context image at the benchmarked size: build the SD1.5-width pipeline (bench.py's), a NativeEngine for batch 1 / 50
steps, save it, and have tests/run_ctx_image.py - a process without torch - load it and run a whole try-on from RGB
condition images; compare with the pipeline here and time the pieces.

    python tools/ctx_image_fullsize.py [--steps 50] [--dir /tmp]
"""
import argparse
import os
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from edgestyle_amd.native import NativeEngine  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--dir", default="/tmp")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    pipe, ws, ucfg, vcfg = bench.build_pipeline(dev, torch.float16)
    lat, pe, ne, imgs, cn = bench.make_inputs(ucfg, vcfg, 1, dev)
    kw = dict(prompt_embeds=pe, negative_prompt_embeds=ne, image=imgs, latents=lat, guidance_scale=7.5,
              num_inference_steps=a.steps, cond_noise=cn)
    want = pipe(output_type="pt", **kw).images.float().cpu()
    eng = NativeEngine(pipe, batch_size=1, num_inference_steps=a.steps)
    path = os.path.join(a.dir, "tryon_b1.esctx")
    t0 = time.time()
    info = eng.save(path)
    t_save = time.time() - t0
    size = os.path.getsize(path)
    print(f"saved {path}: {size / 2**30:.2f} GiB ({info['data_bytes'] / 2**30:.2f} GiB of read-only data in {info['extents']} extents), "
          f"arena {info['arena_bytes'] / 2**30:.2f} GiB in {info['blocks']} segments, "
          f"relocations per plan {info['relocations']}, {t_save:.1f} s", flush=True)
    arrs = dict(n_conds=np.int64(6), latents=lat.permute(0, 2, 3, 1).contiguous().cpu().numpy(),
                ehs=torch.cat([ne, pe]).half().cpu().numpy(), guidance_scale=np.float32(7.5),
                timesteps=pipe.scheduler.set_timesteps(a.steps).float().numpy())
    for i, (im, nz) in enumerate(zip(imgs, cn)):
        arrs[f"img{i}"] = im.float().cpu().numpy()
        if nz is not None:
            arrs[f"noise{i}"] = nz.float().cpu().numpy()
    np.savez(os.path.join(a.dir, "tryon_in.npz"), **arrs)
    eng.close()
    del eng, pipe
    torch.cuda.empty_cache()
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "run_ctx_image.py"), path, os.path.join(a.dir, "tryon_in.npz"),
                        os.path.join(a.dir, "tryon_out.npz")], cwd=ROOT, capture_output=True, text=True, timeout=900)
    print(f"child (load + one try-on, no torch): rc {r.returncode} in {time.time() - t0:.1f} s {r.stdout.strip()} {r.stderr.strip()[-300:]}", flush=True)
    if r.returncode == 0:
        got = torch.from_numpy(np.load(os.path.join(a.dir, "tryon_out.npz"))["image"])
        print("image equal to the pipeline's bit for bit:", bool(torch.equal(got, want)), "max abs diff", float((got - want).abs().max()))
    for f in ("tryon_b1.esctx", "tryon_in.npz", "tryon_out.npz"):
        try:
            os.remove(os.path.join(a.dir, f))
        except OSError:
            pass


if __name__ == "__main__":
    main()
