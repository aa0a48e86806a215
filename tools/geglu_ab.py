"""GEGLU launches of a step (tiled K = 1280 and row-stationary K = 320 | 640) with the library in ES_HIP_LIB: medians of 9 graph replays.  One process per library."""
import math, os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from edgestyle_amd import ops
DEV = "cuda"
g = torch.Generator().manual_seed(0)


def bench(fn, iters=10, reps=4):
    fn(); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(reps):
            fn()
    gr.replay(); torch.cuda.synchronize()
    ts = []
    for _ in range(9):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            gr.replay()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / (iters * reps))
    return statistics.median(ts)


out = []
for M, K, N in [(28672, 1280, 10240), (3584, 1280, 10240), (512, 1280, 10240), (57344, 320, 2560), (458752, 320, 2560), (14336, 640, 5120), (2048, 640, 5120)]:
    x = torch.randn(M, K, generator=g).to(DEV, torch.float16)
    w = torch.randn(N, K, generator=g) / math.sqrt(K)
    pw = ops.pack_weight_ln(w, torch.randn(N, generator=g) * 0.1, torch.ones(K), torch.zeros(K), 1e-5, torch.float16, DEV, geglu=True)
    y = ops.linear(x, pw)
    out.append(f"{bench(lambda: ops.linear(x, pw)):.1f} [{float(y.float().abs().sum()):.6e}]")
print(os.path.basename(os.environ.get("ES_HIP_LIB", "product")), " | ".join(out), flush=True)
