#!/usr/bin/env python
"""Time the LayerNorm-folded linear launches of the C = 1280 levels (es_conv_gemm with ln_colsum: row statistics by Gram MFMAs in
every N tile).  With ES_HIP_LIB pointing at a -DES_LN_NOGRAM=1 tool build the same launches run without those MFMAs (wrong
results): the difference is what statistics handed over by the producer would save.

    python tools/ln_fold_bench.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from edgestyle_amd import ops  # noqa: E402

SHAPES = [(3584, 1280, 10240, True), (3584, 1280, 3840, False), (3584, 1280, 1280, False), (896, 1280, 10240, True), (896, 1280, 3840, False),
          (512, 1280, 10240, True), (512, 1280, 3840, False), (28672, 1280, 10240, True), (28672, 1280, 3840, False), (28672, 1280, 1280, False)]


def main():
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    for M, K, N, geglu in SHAPES:
        R = 6
        x = torch.randn(M, K, generator=g, device=dev).half()
        pws = [ops.pack_weight_ln(torch.randn(N, K, generator=g, device=dev) * 0.02, torch.randn(N, generator=g, device=dev) * 0.1,
                                  1.0 + 0.02 * torch.randn(K, generator=g, device=dev), 0.02 * torch.randn(K, generator=g, device=dev), 1e-5,
                                  torch.float16, dev, geglu=geglu) for _ in range(R)]
        outs = [ops.linear(x, pw) for pw in pws]
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for pw, o in zip(pws, outs):
                ops.linear(x, pw, out=o)
        best = 1e9
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / R)
        print(f"M={M} K={K} N={N} geglu={int(geglu)}: {best * 1e3:7.1f} us  {2.0 * M * N * K / (best * 1e-3) / 1e12:6.0f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()
