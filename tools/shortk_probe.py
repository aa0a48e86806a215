#!/usr/bin/env python
"""Short-K linear layers (K = 320 / 640, M = 57344 / 14336): the 128-pixel tile against the 256 x 320 tile, which reads
each activation row once per 320 couts instead of once per 160 (L2 -> LDS bytes per FLOP halved)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from edgestyle_amd import ops  # noqa: E402

SHAPES = [(57344, 320, 320), (57344, 320, 960), (57344, 1600, 320), (14336, 640, 640), (14336, 640, 1920), (131072, 320, 320)]


def timed(fn, R):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / R)
    return best * 1e3


def main():
    dev, R = "cuda", 8
    gen = torch.Generator(device=dev).manual_seed(0)
    for M, K, N in SHAPES:
        x = torch.randn(M, 1, 1, K, generator=gen, device=dev).half()
        res = torch.randn(M, 1, 1, N, generator=gen, device=dev).half()
        pws = [ops.pack_weight(torch.randn(N, K, generator=gen, device=dev) * 0.02, torch.randn(N, generator=gen, device=dev), torch.float16, dev)
               for _ in range(R)]
        outs = [torch.empty(M, 1, 1, N, device=dev, dtype=torch.float16) for _ in range(R)]
        row = []
        for bn in (0, 320):
            ops.FORCE_BN = bn
            ops.BIG_TILE = bn == 320
            try:
                t = timed(lambda: [ops.conv_gemm(x, pws[i], residual=res, out=outs[i]) for i in range(R)], R)
            except Exception as e:      # 320 does not divide every N
                t = float("nan")
            row.append(t)
        ops.FORCE_BN = 0
        ops.BIG_TILE = False
        ops.FORCE_BK = 32                     # half-depth stages: three 4-wave workgroups per CU
        ew = ops.EIGHT_WAVES
        ops.EIGHT_WAVES = False
        try:
            row.append(timed(lambda: [ops.conv_gemm(x, pws[i], residual=res, out=outs[i]) for i in range(R)], R))
        finally:
            ops.FORCE_BK = 0
            ops.EIGHT_WAVES = ew
        print(f"M={M} K={K} N={N}: planner {row[0]:.1f} us, 256x320 tile {row[1]:.1f} us, BK=32 x 3 workgroups/CU {row[2]:.1f} us", flush=True)


if __name__ == "__main__":
    main()
