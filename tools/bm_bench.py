"""The half-height form of the phase-interleaved tile (128 x 320, es_gemm_desc.bm = 128) against the 256 x 320 tile and the planner's
choice on the launches of a batch-1 step that have too few 256-pixel tiles: hipGraph replays, interleaved, medians."""
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


# N, H, Cin, Cout, k, groups
shapes = [(14, 64, 320, 320, 3, [2, 6, 4, 2]), (14, 32, 640, 640, 3, [2, 6, 4, 2]), (14, 32, 320, 640, 3, [2, 6, 4, 2]), (14, 16, 1280, 1280, 3, [2, 6, 4, 2]),
          (14, 16, 640, 1280, 3, [2, 6, 4, 2]), (14, 32, 3200, 640, 1, [2, 6, 4, 2]), (14, 16, 6400, 1280, 1, [2, 6, 4, 2]),
          (2, 64, 640, 320, 3, None), (2, 64, 960, 320, 3, None), (2, 32, 1280, 640, 3, None), (2, 32, 1920, 640, 3, None), (2, 32, 640, 640, 3, None),
          (2, 16, 2560, 1280, 3, None), (2, 16, 1280, 1280, 3, None), (2, 64, 1600, 320, 1, None), (2, 32, 3200, 640, 1, None)]
only = os.environ.get("ES_ONLY")
for si, (N, H, Cin, Cout, k, groups) in enumerate(shapes):
    if only and str(si) not in only.split(","):
        continue
    x = torch.randn(N, H, H, Cin, generator=g).to(DEV, torch.float16)
    n = len(groups) if groups else 1
    pws = [ops.pack_weight(torch.randn(Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5, torch.randn(Cout, generator=g) * 0.1, torch.float16, DEV)
           for _ in range(n)]
    pw = pws if groups else pws[0]
    kw = dict(group_n=groups) if groups else {}
    M = N * H * H
    cfgs = {"auto": (0, 0, None)}
    t256 = -(-M // 256) * (pws[0].rows_padded // 320)
    for bm in (256, 128):
        tiles = -(-M // bm) * (pws[0].rows_padded // 320)
        for sk in (1, 2, 3, 4, 6, 8):
            if tiles * sk <= 300 and (sk == 1 or tiles * (sk - 1) < 256) and pws[0].kpad // 64 // sk >= 8:
                cfgs[f"{bm}/sk{sk}"] = (320, bm, sk)
    graphs, ref = {}, None
    for name, (bn, bm, sk) in cfgs.items():
        ops.FORCE_BN, ops.FORCE_BM = bn, (bm if bm == 128 else 0)
        try:
            kk = dict(kw)
            if sk:
                kk["splitk"] = sk
            y = ops.conv_gemm(x, pw, **kk).clone()
            if ref is None:
                ref = y
            err = float((y.float() - ref.float()).abs().max())
            graphs[name] = (capture(lambda: ops.conv_gemm(x, pw, **kk)), err)
        except Exception as e:
            print("  skip", name, str(e)[:80])
        finally:
            ops.FORCE_BN, ops.FORCE_BM = 0, 0
    s = {k_: [] for k_ in graphs}
    for _ in range(5):
        for k_ in graphs:
            s[k_].append(timed(graphs[k_][0]))
    fl = 2 * M * Cout * Cin * k * k
    print(f"[{si}] N={N} {H}x{H} {Cin}->{Cout} k{k} (256-px tiles {t256}): " +
          "  ".join(f"{k_} {statistics.median(v):.1f}us/{fl / statistics.median(v) / 1e6:.0f}TF(e{graphs[k_][1]:.0e})" for k_, v in s.items()), flush=True)
