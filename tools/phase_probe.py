"""Per-image phase times of the pipeline (HIP events): condition embedding / per-call preparation / 50-step loop / decode."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device("cuda", 0)
pipe, ws, ucfg, vcfg = bench.build_pipeline(dev, torch.float16)
lat, pe, ne, imgs, cn = bench.make_inputs(ucfg, vcfg, B, dev)
kw = dict(prompt_embeds=pe, negative_prompt_embeds=ne, image=imgs, latents=lat, guidance_scale=7.5, num_inference_steps=50, output_type="pt", cond_noise=cn)
for _ in range(3):
    pipe(**kw)
pipe.collect_timing = True
acc = {}
n = 5
t0 = time.perf_counter()
for _ in range(n):
    pipe(**kw)
    for k, v in pipe.timing.items():
        acc[k] = acc.get(k, 0.0) + v / n
wall = (time.perf_counter() - t0) / n * 1e3
print(f"batch {B}: wall {wall:.1f} ms per call (with a sync per call); phases (GPU ms): " + ", ".join(f"{k} {v:.2f}" for k, v in acc.items()))
