#!/usr/bin/env python
"""Where does a batch-1 denoising step spend its time?  Captures the phases of StepRunner.step as separate hipGraphs
(each ControlNet chain alone, all chains concurrently, fusion, UNet decoder) and times their replays with events."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from edgestyle_amd import ops  # noqa: E402


def timeit(fn, reps=10):
    # NB: loop.one_step advances the device step counter; the kernels clamp it to the table, and callers keep
    # reps * (1 + 3) below the 50 rows anyway
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            g.replay()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    return best


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    dev = torch.device("cuda", 0)
    pipe, ws, ucfg, vcfg = bench.build_pipeline(dev, torch.float16)
    lat, pe, ne, imgs, cn = bench.make_inputs(ucfg, vcfg, B, dev)
    pipe(prompt_embeds=pe, negative_prompt_embeds=ne, image=imgs, latents=lat, guidance_scale=7.5,
         num_inference_steps=50, output_type="latent", cond_noise=cn)
    loop = list(pipe._loops.values())[-1]
    r = pipe._runner
    r.state = loop.state
    x, t_rows, conds = loop.model_in, loop.t_rows, loop.conds
    N = x.shape[0]
    ue = r.unet.engine
    res = {}

    def chain(gi):
        net, pos = r.groups[gi]
        eng = net.engine
        tproj = eng.time_proj(t_rows[: len(pos) * N])
        res[gi] = eng.forward(x, tproj, r.state.ctx_nets[gi], [conds[p] for p in pos])

    def unet_enc():
        tproj = ue.time_proj(t_rows[:N])
        res["u"] = (tproj, ue.encode(x, tproj, r.state.ctx_unet))

    for gi, (net, pos) in enumerate(r.groups):
        print(f"chain {gi} ({type(net).__name__} x{len(pos)}): {timeit(lambda gi=gi: chain(gi)):.3f} ms", flush=True)
    print(f"unet encoder: {timeit(unet_enc):.3f} ms", flush=True)

    def fusion():
        rp, bs = [None] * 6, [None] * 6
        for gi, (net, pos) in enumerate(r.groups):
            for j, p in enumerate(pos):
                rp[p] = [q[j * N:] for q in res[gi]]
                bs[p] = [q.stride(0) for q in res[gi]]
        res["f"] = r.controlnet.engine.forward(rp, bs, N, [1.0] * 6, None)

    print(f"fusion (13 levels): {timeit(fusion):.3f} ms", flush=True)

    def decoder():
        tproj, enc = res["u"]
        ue.forward(x, tproj, r.state.ctx_unet, res["f"][:-1], res["f"][-1], encoded=enc)

    print(f"unet decoder: {timeit(decoder):.3f} ms", flush=True)
    for mode in ("grouped", "streams", "serial"):
        r.mode = mode
        loop.step_idx.zero_()
        print(f"whole step ({mode}): {timeit(loop.one_step, reps=8):.3f} ms", flush=True)


if __name__ == "__main__":
    main()
