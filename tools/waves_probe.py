#!/usr/bin/env python
"""3x3 convolutions on the 128-pixel tile: 4 waves x 2 workgroups per CU (2-stage ring) against 8 waves x 1 workgroup per CU with
the 4-deep ring (3 K-steps of DMA in flight), same N tile.   python tools/waves_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from edgestyle_amd import ops  # noqa: E402
from tools.gemm_bench import bench  # noqa: E402

SHAPES = [(16, 64, 320, 320, 3), (16, 32, 640, 640, 3), (16, 16, 1280, 1280, 3), (14, 32, 640, 640, 3), (14, 16, 1280, 1280, 3), (2, 64, 640, 320, 3)]
for shp in SHAPES:
    cells = []
    for waves, stages in ((0, 2), (8, 2), (8, 4), (0, 4)):
        ops.FORCE_WAVES = waves
        try:
            us, tf, sk = bench(shp, stages, bn=128)
            cells.append(f"w{waves or 4}/st{stages}: {us:7.1f} us {tf:5.0f} TF")
        except Exception as e:
            cells.append(f"w{waves or 4}/st{stages}: n/a {str(e)[:50]}")
        finally:
            ops.FORCE_WAVES = 0
    print(shp, " | ".join(cells), flush=True)
