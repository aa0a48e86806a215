#!/usr/bin/env python
"""es_conv_gemm with tile_n-fastest vs tile_m-fastest XCD chunk order, on the shapes of a batch-1 step
(grouped encoder launches use 4 weight sets like the real step)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from edgestyle_amd import ops  # noqa: E402
from tools.gemm_tune import ENC, DEC, B8  # noqa: E402


def time_it(x, pws, outs, group_n, R):
    def run():
        for i in range(R):
            ops.conv_gemm(x, pws[i], out=outs[i], group_n=group_n)
    run()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        run()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / R)
    return best * 1e3


def main():
    dev, dtype, R = "cuda", torch.float16, 3
    g = torch.Generator(device=dev).manual_seed(0)
    print("shape groups | n-fastest us | m-fastest us | auto us")
    for shp in ENC + DEC + B8[:10]:
        N, H, Cin, Cout, k = shp
        grouped = N in (14, 112) and (H * H * 2) % 128 == 0
        group_n = [N // 7, 3 * N // 7, 2 * N // 7, N // 7] if grouped else None
        x = torch.randn(N, H, H, Cin, generator=g, device=dev).to(dtype)

        def mk():
            return ops.pack_weight(torch.randn(Cout, Cin, k, k, generator=g, device=dev) * 0.02,
                                   torch.randn(Cout, generator=g, device=dev) * 0.1, dtype, dev)
        pws = [[mk() for _ in range(4)] if grouped else mk() for _ in range(R)]
        outs = [torch.empty(N, H, H, Cout, device=dev, dtype=dtype) for _ in range(R)]
        res = []
        for order in (0, 1, -1):
            ops.XCD_ORDER = order
            res.append(time_it(x, pws, outs, group_n, R))
        ops.XCD_ORDER = -1
        print(f"{shp} {'4g' if grouped else '1g'} | {res[0]:7.1f} | {res[1]:7.1f} | {res[2]:7.1f}", flush=True)


if __name__ == "__main__":
    main()
