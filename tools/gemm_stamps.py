"""Where a K-step of conv_gemm_kernel spends its cycles: s_memtime stamps of the 4 waves of two workgroups.
Needs the ES_STAMPS=1 build of gemm_conv.hip as ES_HIP_LIB (tools/gemm_stamps.sh builds it)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from edgestyle_amd import ops

DEV = "cuda"
g = torch.Generator().manual_seed(0)


class Stamps:
    def __init__(self):
        self.buf = torch.zeros(2 * 4 * 8 * 4, dtype=torch.int64, device=DEV)
        self.descs, self.meta = [], []

    def next(self, meta):
        return self.buf.data_ptr()


def run(name, fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    st = Stamps()
    ops.PROFILE = st
    fn()
    ops.PROFILE = None
    torch.cuda.synchronize()
    t = st.buf.cpu().view(2, 4, 8, 4)
    print(name, ": per K-step, cycles from the wave's own loop top: [barrier passed, DMA issued, MFMAs issued] next-top; 4 waves of each of 2 workgroups")
    for b in range(2):
        for w in range(4):
            rows = []
            for k in range(7):
                t0 = int(t[b, w, k, 0])
                rows.append("[%5d %5d %5d] %5d" % (int(t[b, w, k, 1]) - t0, int(t[b, w, k, 2]) - t0, int(t[b, w, k, 3]) - t0, int(t[b, w, k + 1, 0]) - t0))
            print(f"  wg{b} wave{w}: " + " | ".join(rows))


N, H, C = 14, 64, 320
x = torch.randn(N, H, H, C, generator=g).to(DEV, torch.float16)
pws = [ops.pack_weight(torch.randn(320, C, 3, 3, generator=g) / 54, torch.randn(320, generator=g) * 0.1, torch.float16, DEV) for _ in range(4)]
run("grouped conv3x3 320->320 @64x64 x14 (M=57344)", lambda: ops.conv_gemm(x, pws, group_n=[2, 6, 4, 2]))
x2 = torch.randn(14, 32, 32, 640, generator=g).to(DEV, torch.float16)
pw2 = [ops.pack_weight(torch.randn(640, 640, 3, 3, generator=g) / 76, torch.randn(640, generator=g) * 0.1, torch.float16, DEV) for _ in range(4)]
run("grouped conv3x3 640->640 @32x32 x14 (M=14336)", lambda: ops.conv_gemm(x2, pw2, group_n=[2, 6, 4, 2]))
