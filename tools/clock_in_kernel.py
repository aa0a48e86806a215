"""The shader clock a workgroup of the 256 x 320 tile actually runs at: s_memtime (shader cycles) against s_memrealtime (100 MHz) over one
workgroup's life, (a) in a single launch on an idle chip, (b) as the last of 300 back-to-back launches (the state the pipeline's replays run
in).  Needs the stamps build: bash tools/build_variant.sh 8p_stamps gemm_conv8p.hip -DES8P_STAMPS=1; ES_HIP_LIB=edgestyle_amd/lib/ablate/libes_8p_stamps.so."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from edgestyle_amd import ops, lib as L
PROBE = "--probe" in sys.argv
DEV = "cuda"
g = torch.Generator().manual_seed(0)


class Stamps:
    def __init__(self):
        self.buf = torch.zeros(512, dtype=torch.int64, device=DEV)
        self.descs, self.meta = [], []

    def next(self, meta):
        return self.buf.data_ptr()


names = ["prologue issued", "first K-tile landed", "K loop done", "pass-0 tile in LDS", "pass-0 stored", "pass-1 tile in LDS", "end"]
if PROBE:
    pr0 = torch.zeros(2, dtype=torch.int64, device=DEV)
    L.check(L.load().es_clock_probe(pr0.data_ptr(), 50000, None), "es_clock_probe")
    torch.cuda.synchronize()
    print(f"probe wave alone on an idle chip, 50 ms: {int(pr0[0]) / max(int(pr0[1]), 1) * 100:.0f} MHz", flush=True)
for N, H, Cin, Cout in [(112, 64, 320, 320), (16, 64, 960, 320), (112, 32, 640, 640)]:
    x = torch.randn(N, H, H, Cin, generator=g).to(DEV, torch.float16)
    pw = ops.pack_weight(torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5, torch.randn(Cout, generator=g) * 0.1, torch.float16, DEV)
    out = torch.empty(N, H, H, Cout, device=DEV, dtype=torch.float16)
    ops.FORCE_BN = 320
    try:
        for warm in (0, 300):
            torch.cuda.synchronize()
            time.sleep(1.0)
            st = Stamps()
            ops.PROFILE = st
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            pr = torch.zeros(2, dtype=torch.int64, device=DEV)
            side = torch.cuda.Stream()
            if warm and PROBE:
                L.check(L.load().es_clock_probe(pr.data_ptr(), 100000, side.cuda_stream), "es_clock_probe")      # 100 ms beside the launches
            e0.record()
            for _ in range(warm + 1):
                ops.conv_gemm(x, pw, out=out)
            e1.record()
            ops.PROFILE = None
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / (warm + 1)
            if warm and PROBE:
                c, t = [int(v) for v in pr.cpu()]
                print(f"   probe wave on a side stream over the first 100 ms of those launches: {c / max(t, 1) * 100:.0f} MHz", flush=True)
            b = st.buf.cpu()
            fl = 2.0 * N * H * H * Cout * 9 * Cin
            for w, nm in enumerate(("first", "last", "middle")):
                cyc = int(b[256 + w * 8 + 7]) - int(b[256 + w * 8])
                ticks = int(b[288 + w * 2 + 1]) - int(b[288 + w * 2])
                d = [int(b[256 + w * 8 + k]) - int(b[256 + w * 8 + k - 1]) for k in range(1, 8)]
                print(f"N={N} {H}x{H} {Cin}->{Cout} after {warm} launches ({us:.1f} us per launch, {fl / us / 1e6:.0f} TF) {nm} wg: {cyc} cycles in {ticks / 100:.1f} us "
                      f"= {cyc / max(ticks, 1) * 100:.0f} MHz; K loop {d[2]} cycles", flush=True)
    finally:
        ops.FORCE_BN = 0
        ops.PROFILE = None
