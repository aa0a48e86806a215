import os, sys
sys.path.insert(0, '/root/repo')
import torch
from edgestyle_amd import ops
DEV = "cuda"
g = torch.Generator().manual_seed(0)
class Stamps:
    def __init__(self):
        self.buf = torch.zeros(512, dtype=torch.int64, device=DEV); self.descs, self.meta = [], []
    def next(self, meta): return self.buf.data_ptr()
def bench(fn, R=10):
    fn(); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(R): fn()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / R)
    return best * 1e3
for N, H, Cin, Cout, groups in [(14, 8, 1280, 1280, [2, 6, 4, 2]), (2, 16, 2560, 1280, None), (2, 32, 1280, 640, None), (14, 32, 640, 640, [2, 6, 4, 2])]:
    x = torch.randn(N, H, H, Cin, generator=g).to(DEV, torch.float16)
    n = len(groups) if groups else 1
    sets = [[ops.pack_weight(torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5, torch.randn(Cout, generator=g) * 0.1, torch.float16, DEV) for _ in range(n)] for _ in range(3)]
    kw = dict(group_n=groups) if groups else {}
    i = [0]
    def fn():
        i[0] = (i[0] + 1) % 3
        return ops.conv_gemm(x, sets[i[0]] if groups else sets[i[0]][0], **kw)
    t = bench(fn)
    st = Stamps(); ops.PROFILE = st; fn(); ops.PROFILE = None; torch.cuda.synchronize()
    b = st.buf.cpu()[256:272].view(2, 8)
    ph = [[int(b[w, k]) - int(b[w, 0]) for k in range(1, 5)] for w in range(2)]
    M = N * H * H
    print(f"N={N} {H}x{H} {Cin}->{Cout} M={M} plan {ops.plan_gemm(M, sets[0][0].rows_padded, sets[0][0].kpad, False, bns=(320,160,128,64))}: {t:.1f} us per launch incl. reduce; cycles since entry [prologue, first tile landed, K loop done, end]: first wg {ph[0]}, last wg {ph[1]}; the last workgroup entered {int(b[1, 0]) - int(b[0, 0])} cycles after the first", flush=True)
