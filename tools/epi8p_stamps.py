"""Where a workgroup of the 256 x 320 tile spends its time outside the K loop: s_memtime stamps of wave 0 of the first / middle / last
workgroup.  Needs the stamps build: bash tools/ab8p.sh stamps -DES8P_STAMPS=1, ES_HIP_LIB=edgestyle_amd/lib/ablate/libes_8p_stamps.so."""
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


names = ["prologue issued", "first K-tile landed", "K loop done", "pass-0 tile in LDS", "pass-0 stored", "pass-1 tile in LDS", "end"]
for N, H, Cin, Cout, groups in [(14, 64, 320, 320, [2, 6, 4, 2]), (112, 64, 320, 320, [16, 48, 32, 16]), (16, 64, 960, 320, None)]:
    x = torch.randn(N, H, H, Cin, generator=g).to(DEV, torch.float16)
    r = (torch.randn(N, H, H, Cout, generator=g) * 0.5).to(DEV, torch.float16)
    n = len(groups) if groups else 1
    pws = [ops.pack_weight(torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5, torch.randn(Cout, generator=g) * 0.1, torch.float16, DEV)
           for _ in range(n)]
    pw = pws if groups else pws[0]
    kw = dict(group_n=groups) if groups else {}
    ops.FORCE_BN = 320
    try:
        for res in (None, r):
            for _ in range(3):
                ops.conv_gemm(x, pw, residual=res, **kw)
            torch.cuda.synchronize()
            st = Stamps()
            ops.PROFILE = st
            ops.conv_gemm(x, pw, residual=res, **kw)
            ops.PROFILE = None
            torch.cuda.synchronize()
            b = st.buf.cpu()[256:280].view(3, 8)
            for w, nm in enumerate(("first", "last", "middle")):
                d = [int(b[w, k]) - int(b[w, k - 1]) for k in range(1, 8)]
                print(f"N={N} {H}x{H} {Cin}->{Cout} residual={res is not None} {nm} wg: total {int(b[w, 7]) - int(b[w, 0])} cycles; " +
                      ", ".join(f"{a} +{c}" for a, c in zip(names, d)), flush=True)
    finally:
        ops.FORCE_BN = 0
        ops.PROFILE = None
