#!/usr/bin/env python
"""What bounds the mid-size GEMM launches of a batch-1 step (128 x 128 | 128 x 160 | 64 x 64 tiles, the planner's plan): graph of 8 launches
with 8 different weight sets, best of 4 replays.  Run once per library: the product, and ablation builds of csrc/gemm_conv.hip
(tools/gemm_ablate.sh 1 2 4 6 8: no MFMA / no LDS-DMA in the loop / no fragment reads / neither / no barrier and DMA wait).
    ES_HIP_LIB=edgestyle_amd/lib/ablate/libes_abl4.so python3 tools/mid_gemm_ablate.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gemm_bench import bench  # noqa: E402

SHAPES = [(14, 16, 1280, 1280, 1), (14, 32, 640, 640, 1), (14, 16, 5120, 1280, 1), (14, 32, 2560, 640, 1), (14, 32, 640, 640, 3), (14, 16, 640, 1280, 3),
          (2, 64, 320, 320, 1), (2, 32, 640, 640, 1), (2, 16, 1280, 1280, 1), (2, 32, 640, 640, 3), (2, 16, 1280, 1280, 3), (2, 64, 640, 320, 3)]
print("lib:", os.environ.get("ES_HIP_LIB", "product"), flush=True)
for shp in SHAPES:
    us, tf, sk = bench(shp, 0)
    N, H, Cin, Cout, k = shp
    print(f"  {shp} M={N * H * H} K={Cin * k * k} splitk={sk}: {us:7.1f} us {tf:6.0f} TF", flush=True)
