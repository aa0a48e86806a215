#!/usr/bin/env python
"""Headroom probe: es_conv_gemm on the 1x1 / linear shapes of a denoising step next to torch.matmul (hipBLASLt /
rocBLAS) on the same [M,K] x [K,N] problem (random data, R back-to-back launches in a hipGraph, weights cycled).
The library is NOT used by the product path; this only says how far the hand-written kernel is from a tuned GEMM.

    python tools/blas_headroom.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from edgestyle_amd import ops  # noqa: E402

# (M, K, N)
SHAPES = [(57344, 320, 320), (57344, 320, 960), (57344, 1280, 320), (57344, 2880, 320), (14336, 640, 640),
          (14336, 640, 1920), (14336, 2560, 640), (14336, 5760, 640), (3584, 1280, 1280), (3584, 1280, 3840),
          (3584, 5120, 1280), (3584, 11520, 1280), (8192, 320, 320), (8192, 1280, 320), (2048, 640, 640),
          (2048, 2560, 640), (512, 1280, 1280), (512, 5120, 1280),
          (131072, 320, 320), (131072, 2880, 320), (32768, 5760, 640), (8192, 11520, 1280)]


def timed(fn, R):
    fn()
    torch.cuda.synchronize()
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
    print(f"{'M':>7} {'K':>6} {'N':>5} | {'es us':>8} {'TF/s':>6} | {'blas us':>8} {'TF/s':>6} | es/blas")
    for M, K, N in SHAPES:
        x = torch.randn(M, K, generator=gen, device=dev).half()
        ws = [(torch.randn(N, K, generator=gen, device=dev) * 0.02) for _ in range(R)]
        pws = [ops.pack_weight(w, None, torch.float16, dev) for w in ws]
        wts = [w.half().t().contiguous() for w in ws]           # [K,N] row-major, and the transposed view below
        outs = [torch.empty(M, N, device=dev, dtype=torch.float16) for _ in range(R)]
        x4 = x.reshape(M, 1, 1, K)

        def es():
            for i in range(R):
                ops.conv_gemm(x4, pws[i], out=outs[i].reshape(M, 1, 1, N))

        def blas_nn():
            for i in range(R):
                torch.matmul(x, wts[i], out=outs[i])

        whs = [w.half() for w in ws]

        def blas_nt():
            for i in range(R):
                torch.matmul(x, whs[i].t(), out=outs[i])
        te = timed(es, R)
        tb = min(timed(blas_nn, R), timed(blas_nt, R))
        fl = 2.0 * M * K * N
        print(f"{M:7d} {K:6d} {N:5d} | {te:8.1f} {fl / te / 1e6:6.0f} | {tb:8.1f} {fl / tb / 1e6:6.0f} | {te / tb:5.2f}", flush=True)


if __name__ == "__main__":
    main()
