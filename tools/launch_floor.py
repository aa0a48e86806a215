#!/usr/bin/env python
"""Per-launch floor inside a hipGraph chain: a trivial elementwise kernel vs the smallest GEMM / GroupNorm / LayerNorm /
attention launches (each launch depends on the previous one, like the denoising step)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from edgestyle_amd import ops  # noqa: E402


def chain(fn, R=50):
    fn()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(R):
            fn()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / R)
    return best * 1e3


def main():
    dev = "cuda"
    a = torch.zeros(1, 8, 8, 64, device=dev, dtype=torch.float16)
    b = torch.zeros_like(a)
    print(f"es_add 4096 elements:            {chain(lambda: ops.add(a, b, out=a)):6.2f} us")
    x = torch.randn(1, 8, 8, 64, device=dev).half()
    pw = ops.pack_weight(torch.randn(64, 64, 1, 1) * 0.1, None, torch.float16, dev)
    o = torch.empty(1, 8, 8, 64, device=dev, dtype=torch.float16)
    print(f"GEMM M=64 K=64 N=64 (1 WG):      {chain(lambda: ops.conv_gemm(x, pw, out=o)):6.2f} us")
    x2 = torch.randn(2, 32, 32, 640, device=dev).half()
    pw2 = ops.pack_weight(torch.randn(640, 640, 1, 1) * 0.02, None, torch.float16, dev)
    o2 = torch.empty(2, 32, 32, 640, device=dev, dtype=torch.float16)
    print(f"GEMM M=2048 K=640 N=640:         {chain(lambda: ops.conv_gemm(x2, pw2, out=o2)):6.2f} us")
    pw3 = ops.pack_weight(torch.randn(640, 640, 3, 3) * 0.02, None, torch.float16, dev)
    for sk in (1, 6):
        print(f"GEMM M=2048 K=5760 N=640 sk={sk}:   {chain(lambda: ops.conv_gemm(x2, pw3, out=o2, splitk=sk)):6.2f} us")
    g1, b1 = torch.ones(640, device=dev), torch.zeros(640, device=dev)
    print(f"GroupNorm (2,32,32,640):         {chain(lambda: ops.group_norm(x2, g1, b1, 32, 1e-5, True)):6.2f} us (2 launches)")
    t = x2.reshape(2, 1024, 640)
    print(f"LayerNorm (2048,640):            {chain(lambda: ops.layer_norm(t, g1, b1)):6.2f} us")
    qkv = torch.randn(2, 1024, 1920, device=dev).half()
    oa = torch.empty(2, 1024, 640, device=dev, dtype=torch.float16)
    print(f"attention (2,8,1024,1024,80):    {chain(lambda: ops.attention(qkv[:, :, :640], qkv[:, :, 640:1280], qkv[:, :, 1280:], 8, out=oa)):6.2f} us")


if __name__ == "__main__":
    main()
