"""es_fusion_blocks (the 13 EdgeStyle ControlNetBlocks of a step, three launches) at SD1.5 geometry, random data:
us per call, GB/s against the algorithmic bytes ((6 + 1 + 1) S N e + 8 S e with the skip addend) and against the bytes this
three-pass form moves."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from edgestyle_amd import ops
DEV = "cuda"
g = torch.Generator().manual_seed(0)
TABLE = [(320, 64)] * 3 + [(320, 32)] + [(640, 32)] * 2 + [(640, 16)] + [(1280, 16)] * 2 + [(1280, 8)] * 4


def params(c, s):
    hw = s * s
    r = lambda *sh: torch.randn(*sh, generator=g)
    return {k: v.to(DEV) for k, v in dict(
        w1=r(c, 3, 2) * 0.5, b1=r(c, 3) * 0.1, g1=(1 + 0.1 * r(hw, c, 3)).half(), be1=(0.1 * r(hw, c, 3)).half(),
        w2=r(c, 3) * 0.5, b2=r(c) * 0.1, g2=(1 + 0.1 * r(hw, c)).half(), be2=(0.1 * r(hw, c)).half(), w3=r(c), b3=r(c) * 0.1).items()}


for N in (2, 16):
    blocks, addends = [], []
    S = 0
    for c, s in TABLE:
        res = [torch.randn(N, s * s, c, generator=g).to(DEV, torch.float16) for _ in range(6)]
        blocks.append((res, [r.stride(0) for r in res], params(c, s), s * s, c))
        addends.append(torch.randn(N, s * s, c, generator=g).to(DEV, torch.float16))
        S += c * s * s
    fn = lambda: ops.fusion_blocks(blocks, N, [1.0] * 6, addends=addends)
    fn(); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(4):
            fn()
    best = 1e9
    for _ in range(6):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 4)
    alg = (6 + 1 + 1) * S * N * 2 + 8 * S * 2
    moved = (6 + 6 + 1 + 1 + 1 + 1) * S * N * 2 + 8 * S * 2
    print(f"N={N}: {best * 1e3:.1f} us per step  algorithmic {alg / 1e6:.0f} MB = {alg / best / 1e6:.0f} GB/s  moved {moved / 1e6:.0f} MB = {moved / best / 1e6:.0f} GB/s", flush=True)
