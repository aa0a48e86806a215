"""The weight-streaming launches of a batch-1 step (8 x 8 and 16 x 16 levels: M = 128 ... 3584, K = 11520 ... 23040): the planner's choice against
forced tiles and split-K factors.  hipGraph replays, interleaved, medians; GB/s = weight bytes / time."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from edgestyle_amd import ops
DEV = "cuda"
g = torch.Generator().manual_seed(0)


def capture(fn, R=6):
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


shapes = [(14, 8, 1280, 1280, 3, [2, 6, 4, 2]), (2, 16, 2560, 1280, 3, None), (2, 16, 1280, 1280, 3, None), (2, 8, 2560, 1280, 3, None), (2, 8, 1280, 1280, 3, None),
          (2, 32, 1280, 640, 3, None), (2, 32, 1920, 640, 3, None)]
for si, (N, H, Cin, Cout, k, groups) in enumerate(shapes):
    x = torch.randn(N, H, H, Cin, generator=g).to(DEV, torch.float16)
    n = len(groups) if groups else 1
    # rotate over 4 weight sets so that the weights of consecutive replays are not the same L2 / MALL lines (a step never repeats a layer)
    sets = [[ops.pack_weight(torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5, torch.randn(Cout, generator=g) * 0.1, torch.float16, DEV)
             for _ in range(n)] for _ in range(6 if not groups else 3)]
    kw = dict(group_n=groups) if groups else {}
    M = N * H * H
    cfgs = {"auto": (0, None)}
    for bn, sks in ((128, (4, 7, 8, 12, 16, 24)), (160, (4, 7, 8, 12)), (320, (4, 8, 12, 15)), (64, (2, 4, 8))):
        if sets[0][0].rows_padded % bn:
            continue
        for sk in sks:
            if sets[0][0].kpad // 64 // sk >= 6:
                cfgs[f"{bn}/sk{sk}"] = (bn, sk)
    graphs = {}
    for name, (bn, sk) in cfgs.items():
        ops.FORCE_BN = bn
        try:
            kk = dict(kw)
            if sk:
                kk["splitk"] = sk
            i = [0]

            def fn():
                i[0] = (i[0] + 1) % len(sets)
                return ops.conv_gemm(x, sets[i[0]] if groups else sets[i[0]][0], **kk)
            graphs[name] = capture(fn)
        except Exception as e:
            pass
        finally:
            ops.FORCE_BN = 0
    s = {k_: [] for k_ in graphs}
    for _ in range(5):
        for k_ in graphs:
            s[k_].append(timed(graphs[k_]))
    wb = n * Cout * Cin * k * k * 2
    med = {k_: statistics.median(v) for k_, v in s.items()}
    best = sorted(med, key=lambda k_: med[k_])[:5]
    print(f"[{si}] N={N} {H}x{H} {Cin}->{Cout} k{k} M={M} weights {wb / 1e6:.0f} MB: auto {med['auto']:.1f}us ({wb / med['auto'] / 1e3:.0f} GB/s)   best: " +
          "  ".join(f"{k_} {med[k_]:.1f}us" for k_ in best), flush=True)
