"""What the epilogue of the 256 x 320 tile costs (csrc/gemm_conv8p.hip): the level-0 / level-1 3x3 convolutions of a step with and
without a residual input, timed as hipGraph replays, interleaved, medians.  Run once per library (ES_HIP_LIB=<ablation build>,
tools/ab8p.sh noload -DES8P_ABL=64 / nostore -DES8P_ABL=128 / noepi -DES8P_ABL=192) and compare the columns."""
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


shapes = [(14, 64, 320, 320, [2, 6, 4, 2]), (112, 64, 320, 320, [16, 48, 32, 16]), (16, 64, 320, 320, None), (128, 64, 320, 320, None),
          (112, 32, 640, 640, [16, 48, 32, 16]), (16, 32, 1280, 640, None), (16, 64, 960, 320, None)]
print("# lib:", os.environ.get("ES_HIP_LIB", "product"))
for N, H, Cin, Cout, groups in shapes:
    x = torch.randn(N, H, H, Cin, generator=g).to(DEV, torch.float16)
    r = (torch.randn(N, H, H, Cout, generator=g) * 0.5).to(DEV, torch.float16)
    n = len(groups) if groups else 1
    pws = [ops.pack_weight(torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5, torch.randn(Cout, generator=g) * 0.1, torch.float16, DEV)
           for _ in range(n)]
    pw = pws if groups else pws[0]
    kw = dict(group_n=groups) if groups else {}
    graphs = {}
    ops.FORCE_BN = 320
    try:
        graphs["plain"] = capture(lambda: ops.conv_gemm(x, pw, **kw))
        graphs["resid"] = capture(lambda: ops.conv_gemm(x, pw, residual=r, **kw))
    finally:
        ops.FORCE_BN = 0
    s = {k: [] for k in graphs}
    for _ in range(7):
        for k in graphs:
            s[k].append(timed(graphs[k]))
    M = N * H * H
    fl = 2 * M * Cout * Cin * 9
    tiles = -(-M // 256) * -(-Cout // 320)
    print(f"N={N} {H}x{H} {Cin}->{Cout} tiles {tiles} ({tiles / 256:.2f} rounds): " +
          "  ".join(f"{k} {statistics.median(v):.1f} us ({fl / statistics.median(v) / 1e6:.0f} TF)" for k, v in s.items()), flush=True)
