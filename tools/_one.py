import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from edgestyle_amd import ops
from tools.gemm_tune import time_cfg
g = torch.Generator(device="cuda").manual_seed(0)
for shp in [(14, 64, 320, 320, 1), (14, 64, 320, 960, 1), (14, 64, 1280, 320, 1), (14, 32, 640, 640, 1), (14, 32, 640, 1920, 1), (14, 32, 2560, 640, 1),
            (14, 16, 1280, 1280, 1), (14, 16, 1280, 3840, 1), (112, 64, 320, 320, 1), (112, 32, 640, 640, 1)]:
    N, H, Cin, Cout, k = shp
    x = torch.randn(N, H, H, Cin, generator=g, device="cuda").half()
    R = 3
    pws = [ops.pack_weight(torch.randn(Cout, Cin, k, k, generator=g, device="cuda") * 0.02, torch.randn(Cout, generator=g, device="cuda") * 0.1, torch.float16, "cuda") for _ in range(R)]
    outs = [torch.empty(N, H, H, Cout, device="cuda", dtype=torch.float16) for _ in range(R)]
    cells = []
    for bn, st, wv in [(160, 2, 4), (160, 2, 8), (128, 2, 8), (64, 2, 4), (64, 4, 4)]:
        if pws[0].rows_padded % bn: continue
        cells.append(f"bn{bn}/st{st}/w{wv}:{time_cfg(x, pws, outs, bn, 1, st, R, wv):.1f}")
    print(shp, " ".join(cells), flush=True)
