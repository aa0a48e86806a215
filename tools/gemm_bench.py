#!/usr/bin/env python
"""A/B micro-benchmark of es_conv_gemm on the shapes of the denoising step (random data, hipGraph of R back-to-back
launches timed with events; weights of R different layers are cycled so they stream from HBM like in the real step).

    python tools/gemm_bench.py [--stages 2,3,4] [--shapes b1|b8]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from edgestyle_amd import ops  # noqa: E402

# (N, H, Cin, Cout, ksize)  -> M = N*H*H
SHAPES_B1 = [(2, 64, 320, 320, 3), (6, 64, 320, 320, 3), (2, 32, 640, 640, 3), (6, 32, 640, 640, 3),
             (2, 16, 1280, 1280, 3), (6, 16, 1280, 1280, 3), (2, 8, 1280, 1280, 3), (6, 8, 1280, 1280, 3),
             (2, 16, 2560, 1280, 3), (2, 64, 320, 320, 1), (6, 64, 320, 320, 1), (2, 32, 640, 640, 1),
             (2, 16, 1280, 1280, 1), (2, 64, 320, 2560, 1), (2, 32, 640, 5120, 1), (2, 16, 1280, 10240, 1),
             (2, 64, 1280, 320, 1), (2, 64, 320, 960, 1)]
SHAPES_B8 = [(16, 64, 320, 320, 3), (48, 64, 320, 320, 3), (16, 32, 640, 640, 3), (16, 16, 1280, 1280, 3),
             (16, 8, 1280, 1280, 3), (16, 64, 320, 320, 1), (16, 32, 640, 640, 1), (16, 64, 320, 2560, 1),
             (16, 16, 1280, 10240, 1)]


def bench(shape, stages, R=8, dtype=torch.float16, splitk=None, bn=0):
    N, H, Cin, Cout, k = shape
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    x = torch.randn(N, H, H, Cin, generator=g, device=dev, dtype=torch.float32).to(dtype)
    geglu = Cout == 8 * Cin and k == 1
    pws = [ops.pack_weight(torch.randn(Cout, Cin, k, k, generator=g, device=dev) * 0.02,
                           torch.randn(Cout, generator=g, device=dev) * 0.1, dtype, dev, geglu=geglu) for _ in range(R)]
    outs = [None] * R
    ops.FORCE_STAGES = stages
    ops.FORCE_BN = bn
    try:
        for i in range(R):
            outs[i] = ops.conv_gemm(x, pws[i], splitk=splitk)
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for i in range(R):
                ops.conv_gemm(x, pws[i], out=outs[i], splitk=splitk)
        best = 1e9
        for _ in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / R)
    finally:
        ops.FORCE_STAGES = 0
        ops.FORCE_BN = 0
    M = N * H * H
    flops = 2.0 * M * Cout * k * k * Cin
    sk = splitk or (1 if geglu else ops.choose_splitk(M, pws[0].rows_padded, pws[0].bn, pws[0].kpad))
    return best * 1e3, flops / (best * 1e-3) / 1e12, sk


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--stages", default="2,3,4")
    ap.add_argument("--shapes", default="b1")
    ap.add_argument("--bn", default="0", help="comma list of forced N tiles: 0 = planner, 64 | 128 | 160 | 320")
    ap.add_argument("--xcd", default="-1")
    a = ap.parse_args()
    st = [int(s) for s in a.stages.split(",")]
    shapes = SHAPES_B1 if a.shapes == "b1" else SHAPES_B8
    bms = [int(b) for b in a.bn.split(",")]
    ops.XCD_ORDER = int(a.xcd)
    print("shape(N,H,Cin,Cout,k)  M  splitk | " + " | ".join(f"st{s}/bn{b}: us TF/s" for s in st for b in bms), flush=True)
    for shp in shapes:
        cells = []
        for s in st:
          for b in bms:
            try:
                us, tf, sk = bench(shp, s, bn=b)
                cells.append(f"{us:7.1f} {tf:5.0f}")
            except Exception as e:
                cells.append(f"  n/a ({str(e)[:30]})")
                sk = 0
        print(f"{shp}  M={shp[0]*shp[1]*shp[1]} sk={sk} | " + " | ".join(cells), flush=True)


if __name__ == "__main__":
    main()
