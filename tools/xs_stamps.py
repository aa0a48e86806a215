"""Where a stage of es_linear_xs spends its cycles: s_memtime stamps of waves 0 (early) and 4 (late) of two workgroups.
Build first:  XS_STAMPS build is made by this script's caller:
  hipcc ... -DXS_STAMPS=1 -c linear_xs.hip -> edgestyle_amd/lib/ablate/libes_xs_stamps.so ; run with ES_HIP_LIB set."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["ES_XS_MIN_M"] = "0"
import torch
from edgestyle_amd import ops

DEV = "cuda"
g = torch.Generator().manual_seed(0)


class Stamps:
    def __init__(self):
        self.buf = torch.zeros(2 * 2 * 8 * 3, dtype=torch.int64, device=DEV)
        self.descs, self.meta = [], []

    def next(self, meta):
        return self.buf.data_ptr()


for M, K, N, geglu in [(57344, 320, 2560, True), (57344, 320, 960, False)]:
    x = torch.randn(M, K, generator=g).to(DEV, torch.float16)
    w = torch.randn(N, K, generator=g) / math.sqrt(K)
    pw = ops.pack_weight_ln(w, torch.randn(N, generator=g) * 0.1, torch.ones(K), torch.zeros(K), 1e-5, torch.float16, DEV, geglu=geglu)
    for _ in range(3):
        ops.linear(x, pw)
    torch.cuda.synchronize()
    st = Stamps()
    ops.PROFILE = st
    ops.linear(x, pw)
    ops.PROFILE = None
    torch.cuda.synchronize()
    t = st.buf.cpu().view(2, 2, 8, 3)
    print(f"M={M} K={K} N={N} geglu={geglu}: cycles relative to the early wave's barrier exit of stage 4")
    for b in range(2):
        t0 = int(t[b, 0, 0, 0])
        for wv, name in ((0, "early: barrier | MFMA done | epilogue done"), (1, "late : barrier | epilogue(prev) done | MFMA done")):
            rows = [" ".join(f"{int(t[b, wv, s, k]) - t0:6d}" for k in range(3)) for s in range(8)]
            print(f"  wg{b} {name}:\n    " + "\n    ".join(rows))
