"""Fixed costs of a small es_conv_gemm launch: s_memtime stamps of the first and last workgroup (wave 0) at kernel entry,
after the prologue (first tiles issued), when the first tile has landed, after the K loop, at the end of the epilogue.
Needs the ES_STAMPS=1 build (tools/gemm_stamps.sh) as ES_HIP_LIB."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from edgestyle_amd import ops
DEV = "cuda"
g = torch.Generator().manual_seed(0)


class Stamps:
    def __init__(self):
        self.buf = torch.zeros(512, dtype=torch.int64, device=DEV)
        self.descs, self.meta = [], []

    def next(self, meta):
        return self.buf.data_ptr()


def bench(fn, R=10):
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


for M, K, N, res in [(8192, 320, 320, True), (2048, 640, 640, True), (512, 1280, 1280, True), (8192, 320, 1600 // 5 * 5, False),
                     (14336, 640, 640, True), (3584, 1280, 1280, True), (57344, 320, 320, True), (3584, 1280, 3840, False)]:
    x = torch.randn(M, K, generator=g).to(DEV, torch.float16)
    r = torch.randn(M, N, generator=g).to(DEV, torch.float16) if res else None
    pws = [ops.pack_weight(torch.randn(N, K, generator=g) / K ** 0.5, torch.randn(N, generator=g) * 0.1, torch.float16, DEV) for _ in range(4)]
    i = [0]

    def fn():
        i[0] = (i[0] + 1) % 4
        return ops.linear(x, pws[i[0]], residual=r)
    for stages in (0, 2, 3, 4):
        ops.FORCE_STAGES = stages
        try:
            t = bench(fn)
            st = Stamps()
            ops.PROFILE = st
            fn()
            ops.PROFILE = None
            torch.cuda.synchronize()
            b = st.buf.cpu()[256:272].view(2, 8)
            ph = [[int(b[w, k]) - int(b[w, 0]) for k in range(1, 5)] for w in range(2)]
            print(f"M={M} K={K} N={N} stages={stages}: {t:.1f} us per launch (back-to-back in a graph); cycles since entry "
                  f"[prologue, first tile landed, K loop done, end]: first wg {ph[0]}, last wg {ph[1]}", flush=True)
        finally:
            ops.FORCE_STAGES = 0
            ops.PROFILE = None
