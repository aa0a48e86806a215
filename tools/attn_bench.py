#!/usr/bin/env python
"""Micro-benchmark of es_attention on the self/cross-attention shapes of the SD1.5 UNet (random data)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from edgestyle_amd import ops  # noqa: E402

SHAPES = [(2, 8, 4096, 4096, 40), (6, 8, 4096, 4096, 40), (14, 8, 4096, 4096, 40), (16, 8, 4096, 4096, 40), (112, 8, 4096, 4096, 40), (14, 8, 1024, 1024, 80), (14, 8, 4096, 77, 40), (2, 8, 1024, 1024, 80),
          (16, 8, 1024, 1024, 80), (2, 8, 256, 256, 160), (16, 8, 256, 256, 160), (2, 8, 4096, 77, 40),
          (16, 8, 4096, 77, 40), (16, 8, 1024, 77, 80), (1, 1, 4096, 4096, 512)]


def main():
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    print("N heads Sq Skv d | us  TFLOP/s(unpadded)")
    pick = os.environ.get("ATTN_BENCH_SHAPES")
    shapes = SHAPES if not pick else [SHAPES[int(i)] for i in pick.split(",")]
    for N, h, Sq, Skv, d in shapes:
        C = h * d
        qkv = torch.randn(N, Sq, 3 * C, generator=g, device=dev).half()
        kv = torch.randn(N, Skv, 2 * C, generator=g, device=dev).half()
        if Sq == Skv:
            q, k, v = qkv[:, :, :C], qkv[:, :, C:2 * C], qkv[:, :, 2 * C:]
        else:
            q, k, v = qkv[:, :, :C], kv[:, :, :C], kv[:, :, C:]
        out = ops.attention(q, k, v, h)
        torch.cuda.synchronize()
        R = 5
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for _ in range(R):
                ops.attention(q, k, v, h, out=out)
        best = 1e9
        for _ in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / R)
        fl = 4.0 * N * h * Sq * Skv * d
        print(f"{N} {h} {Sq} {Skv} {d} | {best*1e3:8.1f} {fl/(best*1e-3)/1e12:6.0f}", flush=True)


if __name__ == "__main__":
    main()
