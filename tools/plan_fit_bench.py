"""Planner check after a kernel change: the planner's choice against each N tile forced (the planner still picks split-K for the forced
tile) on the large launches of a batch-8 / batch-1 step: hipGraph replays, interleaved, medians."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from edgestyle_amd import ops
DEV = "cuda"
g = torch.Generator().manual_seed(0)


def capture(fn, R=4):
    fn(); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(R):
            fn()
    return gr, R


def timed(gr_r):
    gr, R = gr_r
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / R * 1e3


shapes = [(112, 32, 640, 640, 3, [16, 48, 32, 16]), (112, 32, 320, 640, 3, [16, 48, 32, 16]), (112, 32, 3200, 640, 1, [16, 48, 32, 16]),
          (112, 64, 320, 320, 3, [16, 48, 32, 16]), (112, 16, 1280, 1280, 3, [16, 48, 32, 16]), (112, 16, 640, 1280, 3, [16, 48, 32, 16]),
          (112, 8, 1280, 1280, 3, [16, 48, 32, 16]), (16, 64, 640, 320, 3, None), (16, 32, 1280, 640, 3, None), (16, 32, 1920, 640, 3, None),
          (16, 16, 2560, 1280, 3, None), (16, 16, 1280, 1280, 3, None), (16, 32, 640, 640, 3, None), (16, 64, 960, 320, 3, None),
          (14, 32, 640, 640, 3, [2, 6, 4, 2]), (14, 16, 1280, 1280, 3, [2, 6, 4, 2]), (28, 32, 640, 640, 3, [4, 12, 8, 4]), (56, 32, 640, 640, 3, [8, 24, 16, 8])]
for si, (N, H, Cin, Cout, k, groups) in enumerate(shapes):
    x = torch.randn(N, H, H, Cin, generator=g).to(DEV, torch.float16)
    n = len(groups) if groups else 1
    pws = [ops.pack_weight(torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5, torch.randn(Cout, generator=g) * 0.1, torch.float16, DEV)
           for _ in range(n)]
    pw = pws if groups else pws[0]
    kw = dict(group_n=groups) if groups else {}
    M = N * H * H
    graphs, picked = {}, {}
    for bn in (0, 320, 160, 128):
        if bn and pws[0].rows_padded % bn:
            continue
        ops.FORCE_BN = bn
        try:
            picked[bn] = ops.plan_gemm(M, pws[0].rows_padded, pws[0].kpad, False, bns=(bn,) if bn else (320, 160, 128, 64))
            graphs[bn] = capture(lambda: ops.conv_gemm(x, pw, **kw))
        except Exception as e:
            print("  skip", bn, str(e)[:80])
        finally:
            ops.FORCE_BN = 0
    s = {b: [] for b in graphs}
    for _ in range(5):
        for b in graphs:
            s[b].append(timed(graphs[b]))
    fl = 2 * M * Cout * Cin * k * k
    med = {b: statistics.median(v) for b, v in s.items()}
    best = min((b for b in med if b), key=lambda b: med[b])
    print(f"[{si}] N={N} {H}x{H} {Cin}->{Cout} k{k}: " + "  ".join(f"{'auto' if b == 0 else b}{picked[b]} {med[b]:.1f}us/{fl / med[b] / 1e6:.0f}TF" for b in med) +
          f"   best forced {best} ({med[0] / med[best]:.3f}x of auto)", flush=True)
