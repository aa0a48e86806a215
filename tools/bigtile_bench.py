"""256-pixel tile (8 waves, staggered half-workgroups) vs the default 128-pixel tile on the large launches of a step."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from edgestyle_amd import ops
DEV = "cuda"
g = torch.Generator().manual_seed(0)


def bench(fn, R=6):
    fn(); torch.cuda.synchronize()
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


# (N, H, Cin, Cout, k, groups)
shapes = [(14, 64, 320, 320, 3, [2, 6, 4, 2]), (14, 32, 640, 640, 3, [2, 6, 4, 2]), (14, 64, 320, 320, 1, [2, 6, 4, 2]),
          (16, 64, 320, 320, 3, None), (16, 64, 640, 320, 3, None), (16, 32, 640, 640, 3, None), (16, 32, 1280, 640, 3, None),
          (16, 16, 1280, 1280, 3, None), (112, 64, 320, 320, 3, [16, 48, 32, 16]), (112, 32, 640, 640, 3, [16, 48, 32, 16]),
          (16, 64, 1600, 320, 1, None), (112, 64, 1600, 320, 1, [16, 48, 32, 16])]
for N, H, Cin, Cout, k, groups in shapes:
    x = torch.randn(N, H, H, Cin, generator=g).to(DEV, torch.float16)
    n = len(groups) if groups else 1
    pws = [ops.pack_weight(torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5, torch.randn(Cout, generator=g) * 0.1,
                           torch.float16, DEV) for _ in range(n)]
    pw = pws if groups else pws[0]
    kw = dict(group_n=groups) if groups else {}
    res, outs = {}, {}
    for bm in (0, 256):
        ops.FORCE_BM = bm
        try:
            outs[bm] = ops.conv_gemm(x, pw, **kw).clone()
            res[bm] = bench(lambda: ops.conv_gemm(x, pw, **kw))
        finally:
            ops.FORCE_BM = 0
    fl = 2.0 * N * H * H * Cout * k * k * Cin
    same = torch.equal(outs[0], outs[256])
    print(f"N={N} H={H} {Cin}->{Cout} k={k} grouped={bool(groups)}: default {res[0]:.1f} us ({fl / res[0] / 1e6:.0f} TF)   "
          f"bm256 {res[256]:.1f} us ({fl / res[256] / 1e6:.0f} TF)   ratio {res[0] / res[256]:.2f}  bitwise_equal={same}", flush=True)
