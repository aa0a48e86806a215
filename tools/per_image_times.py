#!/usr/bin/env python
"""Per-image wall times of the batch-1 pipeline inside ONE process (run-to-run / placement variance probe)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    dev = torch.device("cuda", 0)
    pad_kb = int(os.environ.get("ES_PAD_KB", "0"))        # shifts every later device allocation
    pad = torch.empty(pad_kb * 1024, dtype=torch.uint8, device=dev) if pad_kb else None
    for rebuild in range(int(os.environ.get("ES_REBUILDS", "1"))):
        # (a rebuild re-allocates every weight / activation buffer and re-captures the step graph in the same process)
        pipe, ws, ucfg, vcfg = bench.build_pipeline(dev, torch.float16)
        lat, pe, ne, imgs, cn = bench.make_inputs(ucfg, vcfg, 1, dev)
        ts = []
        for i in range(n + 1):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            pipe(prompt_embeds=pe, negative_prompt_embeds=ne, image=imgs, latents=lat, guidance_scale=7.5,
                 num_inference_steps=50, output_type="pt", cond_noise=cn)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        print("ms per image:", " ".join(f"{t:.1f}" for t in ts[1:]), flush=True)
        del pipe, ws, lat, pe, ne, imgs, cn
        import gc
        from edgestyle_amd import engine, ops
        engine._PACK_CACHE.clear()
        gc.collect()
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
