#!/usr/bin/env python
"""Replay the es_conv_gemm launches of ONE denoising step (geometry list dumped by `ES_DUMP_GEMM=1 bench.py` into
gpurun_out/gemm_step_launches.json) as stand-alone eager launches, so that `rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE`
can count the kernel's HBM traffic per launch (PMC collection segfaults inside a hipGraph replay / the full pipeline
on this ROCm build, and needs separate passes per counter).  Every launch gets its own weight tensor, so weights
stream from HBM as in the real step.

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d out -- python3 tools/gemm_step_traffic.py launches.json
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from edgestyle_amd import ops  # noqa: E402


def main(path):
    dev = "cuda"
    launches = json.load(open(path))
    g = torch.Generator(device=dev).manual_seed(0)
    work = []
    for L in launches:
        q = L["geom"]
        x = torch.randn(q["N"], q["H"], q["W"], q["C1"], generator=g, device=dev).half()
        x2 = torch.randn(q["N"], q["H"], q["W"], q["C2"], generator=g, device=dev).half() if q["C2"] else None
        ng = len(q["group_n"]) if q.get("group_n") else 1
        pw = [ops.pack_weight(torch.randn(q["cout"], q["C1"] + q["C2"], q["k"], q["k"], generator=g, device=dev) * 0.02,
                              torch.zeros(q["cout"], device=dev), torch.float16, dev, geglu=q["geglu"]) for _ in range(ng)]
        pw = pw if ng > 1 else pw[0]
        cs = q["cout"] // 2 if q["geglu"] else q["cout"]
        res = torch.randn(q["N"], q["Hout"], q["Wout"], cs, generator=g, device=dev).half() if q["residual"] else None
        work.append((x, x2, pw, res, q))
    torch.cuda.synchronize()
    for x, x2, pw, res, q in work:
        ops.FORCE_BN = q.get("bn", 0)
        ops.conv_gemm(x, pw, x2=x2, stride=q["stride"], pad=q["pad"], upsample=q["upsample"], residual=res,
                      splitk=q["splitk"], stages=q.get("stages", 0), out_hw=(q["Hout"], q["Wout"]),
                      group_n=q.get("group_n"))
    ops.FORCE_BN = 0
    torch.cuda.synchronize()
    print(f"replayed {len(work)} launches")


if __name__ == "__main__":
    main(sys.argv[1])
