#!/usr/bin/env python
"""Sweep (bn, splitk, stages) of es_conv_gemm over the GEMM shapes of one batch-1 denoising step (grouped encoder:
14 samples; decoder: 2 samples) and compare the launch heuristics of ops.choose_launch_bn / ops.plan_launch with the
best measured configuration.  Output: one line per shape; JSON of all cells with --json.

    python tools/gemm_tune.py [--json gpurun_out/gemm_tune.json] [--set enc|dec|all]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from edgestyle_amd import ops  # noqa: E402

# (N, H, Cin, Cout, ksize)
ENC = [(14, 64, 320, 320, 3), (14, 64, 320, 320, 1), (14, 64, 1280, 320, 1), (14, 64, 320, 960, 1),
       (14, 32, 640, 640, 3), (14, 32, 320, 640, 3), (14, 32, 640, 640, 1), (14, 32, 2560, 640, 1), (14, 32, 640, 1920, 1),
       (14, 16, 1280, 1280, 3), (14, 16, 640, 1280, 3), (14, 16, 1280, 1280, 1), (14, 16, 5120, 1280, 1),
       (14, 16, 1280, 3840, 1), (14, 8, 1280, 1280, 3), (14, 8, 1280, 1280, 1)]
DEC = [(2, 64, 320, 320, 1), (2, 64, 320, 320, 3), (2, 64, 640, 320, 3), (2, 64, 960, 320, 3), (2, 64, 1280, 320, 1),
       (2, 32, 640, 640, 1), (2, 32, 640, 640, 3), (2, 32, 960, 640, 3), (2, 32, 1280, 640, 3), (2, 32, 1920, 640, 3),
       (2, 32, 2560, 640, 1), (2, 16, 1280, 1280, 1), (2, 16, 1280, 1280, 3), (2, 16, 1920, 1280, 3),
       (2, 16, 2560, 1280, 3), (2, 16, 5120, 1280, 1), (2, 8, 1280, 1280, 3), (2, 8, 2560, 1280, 3)]


B8 = [(112, 64, 320, 320, 3), (112, 64, 320, 320, 1), (112, 64, 1280, 320, 1), (112, 64, 320, 960, 1),
      (112, 32, 640, 640, 3), (112, 32, 640, 640, 1), (112, 32, 2560, 640, 1), (112, 16, 1280, 1280, 3),
      (112, 16, 1280, 1280, 1), (112, 8, 1280, 1280, 3), (16, 64, 320, 320, 3), (16, 64, 640, 320, 3),
      (16, 32, 640, 640, 3), (16, 32, 1280, 640, 3), (16, 16, 1280, 1280, 3), (16, 16, 2560, 1280, 3)]


def time_cfg(x, pws, outs, bn, splitk, stages, R, waves=0):
    ops.FORCE_BN = bn
    ops.FORCE_WAVES = waves
    try:
        for i in range(R):
            ops.conv_gemm(x, pws[i], out=outs[i], splitk=splitk, stages=stages)
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for i in range(R):
                ops.conv_gemm(x, pws[i], out=outs[i], splitk=splitk, stages=stages)
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / R)
        return best * 1e3
    finally:
        ops.FORCE_BN = 0
        ops.FORCE_WAVES = 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--json", default=None)
    ap.add_argument("--set", default="all")
    a = ap.parse_args()
    shapes = {"enc": ENC, "dec": DEC, "all": ENC + DEC, "b8": B8, "big": ENC[:10] + B8}[a.set]
    dev, dtype, R = "cuda", torch.float16, 4
    g = torch.Generator(device=dev).manual_seed(0)
    cells = []
    print("shape (N,H,Cin,Cout,k) M | heuristic (bn,sk,st) us | best (bn,sk,st) us | gain", flush=True)
    for shp in shapes:
        N, H, Cin, Cout, k = shp
        M = N * H * H
        x = torch.randn(N, H, H, Cin, generator=g, device=dev).to(dtype)
        pws = [ops.pack_weight(torch.randn(Cout, Cin, k, k, generator=g, device=dev) * 0.02,
                               torch.randn(Cout, generator=g, device=dev) * 0.1, dtype, dev) for _ in range(R)]
        outs = [torch.empty(N, H, H, Cout, device=dev, dtype=dtype) for _ in range(R)]
        pw = pws[0]
        nk = pw.kpad // ops.BK
        hbn = ops.choose_launch_bn(M, pw)
        ops.LANE = 0
        hsk, hst = ops.plan_launch(M, pw, hbn)
        res = {}
        for bn in (64, 128, 160, 320):
            if pw.rows_padded % bn:
                continue
            tiles = -(-M // {320: 256, 64: 64}.get(bn, ops.BM)) * (pw.rows_padded // bn)
            for sk in (1, 2, 3, 4, 5, 6, 8, 10, 12, 16, 20, 24, 32):
                if bn == 64 and (Cin % 64 or M > 16384):
                    continue
                if sk > 1 and (nk // sk < 3 or tiles * sk > 1536):
                    continue
                for st in (2, 4):
                    if st == 4 and (tiles * sk > (1024 if bn == 64 else 256) or bn == 320):
                        continue
                    for wv in (4, 8):
                        if wv == 8 and (bn in (64, 320) or (st == 4 and bn != 128) or Cin % 64):
                            continue
                        try:
                            res[(bn, sk, st, wv)] = time_cfg(x, pws, outs, bn, sk, st, R, wv)
                        except Exception as e:  # noqa: BLE001
                            print("   skip", (bn, sk, st, wv), str(e)[:60], flush=True)
        if (hbn, hsk, hst, 4) not in res:
            res[(hbn, hsk, hst, 4)] = time_cfg(x, pws, outs, hbn, hsk, hst, R, 4)
        h = res[(hbn, hsk, hst, 4)]
        top = sorted(res, key=res.get)[:4]
        bk = top[0]
        print(f"{shp} {M} | {(hbn, hsk, hst)} {h:.1f} | {bk} {res[bk]:.1f} | {h / res[bk]:.2f}x | "
              + " ".join(f"{c}:{res[c]:.1f}" for c in top[1:]), flush=True)
        cells.append(dict(shape=shp, M=M, heuristic=[hbn, hsk, hst], cells={f"{b},{s},{t},{w}": round(v, 2) for (b, s, t, w), v in res.items()}))
    if a.json:
        os.makedirs(os.path.dirname(a.json), exist_ok=True)
        json.dump(cells, open(a.json, "w"))


if __name__ == "__main__":
    main()
