#!/usr/bin/env python
"""Micro-benchmark of es_group_norm (stats + apply) and es_layer_norm on the shapes of one batch-1 denoising step."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from edgestyle_amd import ops  # noqa: E402

GN = [(14, 64, 320), (14, 32, 320), (14, 32, 640), (14, 16, 640), (14, 16, 1280), (14, 8, 1280),
      (2, 64, 320), (2, 64, 640), (2, 64, 960), (2, 32, 640), (2, 32, 1280), (2, 32, 1920), (2, 16, 1280), (2, 16, 2560),
      (2, 8, 2560), (112, 64, 320), (112, 32, 640)]
LN = [(14 * 4096, 320), (14 * 1024, 640), (14 * 256, 1280), (2 * 4096, 320), (2 * 1024, 640), (2 * 256, 1280)]


def timeit(fn, R=10):
    fn()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(R):
            fn()
    best = 1e9
    for _ in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / R)
    return best * 1e3


def main():
    g = torch.Generator(device="cuda").manual_seed(0)
    print("GroupNorm+SiLU (N,H,C): us, GB/s over read+read+write")
    for N, H, C in GN:
        x = torch.randn(N, H, H, C, generator=g, device="cuda").half()
        gam = torch.ones(C, device="cuda"); bet = torch.zeros(C, device="cuda")
        out = torch.empty_like(x)
        us = timeit(lambda: ops.group_norm(x, gam, bet, 32, 1e-5, True))
        print(f"  {(N, H, C)}: {us:7.1f} us  {3 * x.numel() * 2 / us / 1e3:7.0f} GB/s  ({x.numel() * 2 / 1e6:.1f} MB)", flush=True)
    print("LayerNorm (rows, C): us, GB/s over read+write")
    for M, C in LN:
        x = torch.randn(1, M, C, generator=g, device="cuda").half()
        gam = torch.ones(C, device="cuda"); bet = torch.zeros(C, device="cuda")
        us = timeit(lambda: ops.layer_norm(x, gam, bet))
        print(f"  {(M, C)}: {us:7.1f} us  {2 * x.numel() * 2 / us / 1e3:7.0f} GB/s", flush=True)


if __name__ == "__main__":
    main()
