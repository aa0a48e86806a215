#!/usr/bin/env python
"""Weight-streaming-bound 3x3 convolutions of the deep levels at batch 1 (M = 128 ... 3584 output pixels, K = 11520 ... 23040,
N = 1280): the planner's tile against a forced N tile / split-K, grouped like the step's launches.

    python tools/skinny_bench.py [--bn 0,128,160] [--sk 0,4,7]
(this is the harness a 128 x 320 instantiation of conv_gemm_kernel was measured with - and lost.)
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from edgestyle_amd import ops  # noqa: E402

# (group_n or None, N, H, C1, C2, Cout)
SHAPES = [([2, 6, 4, 2], 14, 8, 1280, 0, 1280), (None, 2, 8, 1280, 1280, 1280), (None, 2, 16, 1280, 1280, 1280), (None, 2, 16, 1280, 640, 1280),
          ([2, 6, 4, 2], 14, 16, 1280, 0, 1280), (None, 2, 32, 640, 640, 640)]


def bench(shape, bn, sk, R=6):
    grp, N, H, C1, C2, Cout = shape
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    x = torch.randn(N, H, H, C1, generator=g, device=dev).half()
    x2 = torch.randn(N, H, H, C2, generator=g, device=dev).half() if C2 else None
    ng = len(grp) if grp else 1
    sets = [[ops.pack_weight(torch.randn(Cout, C1 + C2, 3, 3, generator=g, device=dev) * 0.02, torch.randn(Cout, generator=g, device=dev) * 0.1,
                             torch.float16, dev) for _ in range(ng)] for _ in range(R)]
    ops.FORCE_BN = bn
    try:
        kw = dict(x2=x2, group_n=grp, splitk=sk or None)
        outs = [ops.conv_gemm(x, ws if grp else ws[0], **kw) for ws in sets]
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for ws, o in zip(sets, outs):
                ops.conv_gemm(x, ws if grp else ws[0], out=o, **kw)
        best = 1e9
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / R)
    finally:
        ops.FORCE_BN = 0
    M = N * H * H
    return best * 1e3, 2.0 * M * Cout * 9 * (C1 + C2) / (best * 1e-3) / 1e12, outs[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bn", default="0")
    ap.add_argument("--sk", default="0")
    a = ap.parse_args()
    for shp in SHAPES:
        ref = None
        cells = []
        for bn in [int(b) for b in a.bn.split(",")]:
            for sk in [int(s) for s in a.sk.split(",")]:
                try:
                    us, tf, out = bench(shp, bn, sk)
                    if ref is None:
                        ref = out.float()
                    err = float((out.float() - ref).abs().max())
                    cells.append(f"bn{bn}/sk{sk}: {us:6.1f} us {tf:4.0f} TF (d {err:.1e})")
                except Exception as e:
                    cells.append(f"bn{bn}/sk{sk}: n/a {str(e)[:40]}")
        print(shp, " | ".join(cells), flush=True)


if __name__ == "__main__":
    main()
