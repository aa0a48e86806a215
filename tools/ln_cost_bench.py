import os, sys, statistics, math
sys.path.insert(0, '/root/repo')
import torch
from edgestyle_amd import ops
DEV = "cuda"
g = torch.Generator().manual_seed(0)
def capture(fn, R=4):
    fn(); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(R): fn()
    return gr, R
def timed(gr_r):
    gr, R = gr_r
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / R * 1e3
for M, C, Cout, geglu in [(28672, 1280, 10240, True), (3584, 1280, 10240, True), (28672, 1280, 3840, False), (3584, 1280, 3840, False), (512, 1280, 10240, True)]:
    x = torch.randn(M, C, generator=g).to(DEV, torch.float16)
    w, b = torch.randn(Cout, C, generator=g) / math.sqrt(C), torch.randn(Cout, generator=g) * 0.1
    pln = ops.pack_weight_ln(w, b, 1 + 0.1 * torch.randn(C, generator=g), 0.1 * torch.randn(C, generator=g), 1e-5, torch.float16, DEV, geglu=geglu)
    ppl = ops.pack_weight(w, b, torch.float16, DEV, geglu=geglu)
    graphs = {"ln": capture(lambda: ops.linear(x, pln)), "plain": capture(lambda: ops.linear(x, ppl))}
    if not geglu:
        ops.FORCE_BN = 128
        graphs["plain128"] = capture(lambda: ops.linear(x, ppl))
        ops.FORCE_BN = 0
    s = {k: [] for k in graphs}
    for _ in range(5):
        for k in graphs: s[k].append(timed(graphs[k]))
    fl = 2 * M * Cout * C
    print(f"M={M} {C}->{Cout} geglu={geglu}: " + "  ".join(f"{k} {statistics.median(v):.1f}us/{fl/statistics.median(v)/1e6:.0f}TF" for k, v in s.items()), flush=True)
