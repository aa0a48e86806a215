"""The K = 1280 GEGLU projection (norm3 -> ff.net.0 of the 16 x 16 level: LayerNorm fold + GEGLU epilogue on the 128 x 128 tile) with 4 or 8
waves per tile and a 2- or 4-deep ring: hipGraph replays, interleaved, medians."""
import os, sys, statistics, math
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


for M, C, Cout, groups in [(28672, 1280, 10240, [4096, 12288, 8192, 4096]), (3584, 1280, 10240, [512, 1536, 1024, 512]), (4096, 1280, 10240, None), (512, 1280, 10240, None),
                           (28672, 1280, 3840, [4096, 12288, 8192, 4096])]:
    geglu = Cout == 10240
    x = torch.randn(M, C, generator=g).to(DEV, torch.float16)
    n = len(groups) if groups else 1
    pws = [ops.pack_weight_ln(torch.randn(Cout, C, generator=g) / math.sqrt(C), torch.randn(Cout, generator=g) * 0.1, 1 + 0.1 * torch.randn(C, generator=g),
                              0.1 * torch.randn(C, generator=g), 1e-5, torch.float16, DEV, geglu=geglu) for _ in range(n)]
    pw = pws if groups else pws[0]
    kw = dict(group_n=groups) if groups else {}
    graphs = {}
    for waves in (0, 4, 8):
        for stages in (0, 4):
            ops.FORCE_WAVES, ops.FORCE_STAGES = waves, stages
            try:
                graphs[(waves, stages)] = capture(lambda: ops.linear(x, pw, **kw))
            except Exception as e:
                print("  skip", waves, stages, str(e)[:60])
            finally:
                ops.FORCE_WAVES, ops.FORCE_STAGES = 0, 0
    s = {k: [] for k in graphs}
    for _ in range(5):
        for k in graphs:
            s[k].append(timed(graphs[k]))
    fl = 2 * M * Cout * C
    print(f"M={M} {C}->{Cout} geglu={geglu}: " + "  ".join(f"w{k[0]}/s{k[1]} {statistics.median(v):.1f}us/{fl / statistics.median(v) / 1e6:.0f}TF" for k, v in s.items()), flush=True)
