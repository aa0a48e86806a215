"""The LayerNorm-folded / GEGLU linear layers of the C = 1280 level on the 256 x 256 phase-interleaved tile (bn = 256, round 5) against the
128-wide tile the planner gave them before: same process, alternating graph replays, medians of 7; outputs compared bit for bit."""
import math, os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from edgestyle_amd import ops
DEV = "cuda"
g = torch.Generator().manual_seed(0)


def graph_of(fn, reps=4):
    fn(); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(reps):
            fn()
    gr.replay(); torch.cuda.synchronize()
    return gr


def time_graph(gr, iters=10, reps=4):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        gr.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (iters * reps)


K = 1280
print("M N geglu ln | 128-wide us (TF) | 256 x 256 us (TF) | x | planner's choice with the tile offered")
for M in (28672, 7168, 4096, 3584, 2048, 896, 512):
    for N, geglu, ln in ((10240, True, True), (3840, False, True), (1280, False, True)):
        x = torch.randn(M, K, generator=g).to(DEV, torch.float16)
        pw = ops.pack_weight_ln(torch.randn(N, K, generator=g) / math.sqrt(K), torch.randn(N, generator=g) * 0.1, torch.ones(K), torch.zeros(K), 1e-5,
                                torch.float16, DEV, geglu=geglu)
        outs, gs = {}, {}
        for name in ("small", "big"):
            ops.FORCE_BN, ops.BIG_TILE_256 = (256, True) if name == "big" else (0, False)
            outs[name] = ops.linear(x, pw).clone()
            gs[name] = graph_of(lambda: ops.linear(x, pw))
        ops.FORCE_BN, ops.BIG_TILE_256 = 0, True
        choice = ops.plan_gemm(M, pw.rows_padded, pw.kpad, pw.geglu, bns=(256, 160, 128, 64) if not geglu else (256, 160, 128), allow_split=False)
        ts = {"small": [], "big": []}
        for _ in range(7):
            for name in ("small", "big"):
                ts[name].append(time_graph(gs[name]))
        t0, t1 = statistics.median(ts["small"]), statistics.median(ts["big"])
        fl = 2.0 * M * K * N
        print(f"{M} {N} {int(geglu)} {int(ln)} | {t0:7.1f} ({fl / t0 / 1e6:4.0f}) | {t1:7.1f} ({fl / t1 / 1e6:4.0f}) | x{t0 / t1:.2f} | {choice} | equal={torch.equal(outs['small'], outs['big'])}", flush=True)
