"""Per-launch GEMM table of one denoising step (in-kernel stamps) at a given batch: which tile each launch got, its time
and TFLOP/s; `--json name` also writes gpurun_out/name."""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--k", type=int, default=0, help="only launches with this kernel size (0: all)")
ap.add_argument("--json", default="")
a = ap.parse_args()
dev = torch.device("cuda", 0)
pipe, ws, ucfg, vcfg = bench.build_pipeline(dev, torch.float16)
lat, pe, ne, imgs, cn = bench.make_inputs(ucfg, vcfg, a.batch, dev)
for _ in range(2):
    pipe(prompt_embeds=pe, negative_prompt_embeds=ne, image=imgs, latents=lat, guidance_scale=7.5, num_inference_steps=6,
         output_type="pt", cond_noise=cn)
res = pipe.profile_one_step()
rows = []
tot = {}
for (fl, k, shp, g), t in res:
    if a.k and k != a.k:
        continue
    key = (k, g.get("bn"))
    tt = tot.setdefault(key, [0.0, 0.0, 0])
    tt[0] += t; tt[1] += fl; tt[2] += 1
    rows.append(f"N{g['N']:4d} H{g['Hout']:3d} Cin{g['C1'] + g['C2']:5d} tail{g.get('ctail', 0):5d} Cout{g['cout']:5d} k{k} s{g['stride']} sk{g['splitk']:2d} "
                f"bn{g.get('bn', 0):3d} {t * 1e6:8.1f} us {fl / t / 1e12:6.0f} TF")
print("\n".join(rows))
for key, (t, fl, n) in sorted(tot.items(), key=lambda kv: -kv[1][0]):
    print(f"k={key[0]} bn={key[1]}: {n} launches {t * 1e3:.3f} ms {fl / t / 1e12:.0f} TF")
if a.json:
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump([dict(geom=m[3], seconds=t) for m, t in res], open(os.path.join("gpurun_out", a.json), "w"))
