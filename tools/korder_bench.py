"""K order of the 3x3 convolutions (es_gemm_desc.korder): tap-major (round 3) vs chunk-major packed weights on the
launches of a step, for the 128-pixel tile and the 256 x 320 tile, interleaved in ONE process (hipGraph replays of R
launches, 3 warm-up rounds, 9 alternating rounds, medians), plus a check of both orders against torch's conv2d on the GPU
(fp32 accumulate of the same fp16 operands: the two orders differ from it, and from each other, by rounding only).

    python tools/korder_bench.py            [ES_ONLY=0,3 to pick shapes]
"""
import os, sys
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


# (N, H, Cin, Cout, stride, upsample, groups, bn list)
shapes = [
    (112, 64, 320, 320, 1, False, [16, 48, 32, 16], (160, 320)),
    (112, 32, 640, 640, 1, False, [16, 48, 32, 16], (160, 320)),
    (112, 16, 1280, 1280, 1, False, [16, 48, 32, 16], (160, 320)),
    (112, 8, 1280, 1280, 1, False, [16, 48, 32, 16], (160, 320)),
    (16, 64, 640, 320, 1, False, None, (160, 320)),
    (16, 32, 1280, 640, 1, False, None, (160, 320)),
    (16, 16, 2560, 1280, 1, False, None, (128, 320)),
    (16, 16, 1280, 1280, 1, True, None, (128, 320)),
    (112, 64, 320, 320, 2, False, [16, 48, 32, 16], (160, 320)),
    (14, 64, 320, 320, 1, False, [2, 6, 4, 2], (160, 320)),
    (14, 32, 640, 640, 1, False, [2, 6, 4, 2], (160, 320)),
    (14, 16, 1280, 1280, 1, False, [2, 6, 4, 2], (160, 0)),
    (14, 8, 1280, 1280, 1, False, [2, 6, 4, 2], (128, 0)),
    (2, 64, 960, 320, 1, False, None, (160, 0)),
    (2, 32, 1920, 640, 1, False, None, (128, 0)),
    (2, 16, 2560, 1280, 1, False, None, (128, 0)),
]
only = os.environ.get("ES_ONLY")
for si, (N, H, Cin, Cout, stride, up, groups, bns) in enumerate(shapes):
    if only and str(si) not in only.split(","):
        continue
    x = (torch.randn(N, H, H, Cin, generator=g)).to(DEV, torch.float16)
    n = len(groups) if groups else 1
    ws = [torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5 for _ in range(n)]
    bs = [torch.randn(Cout, generator=g) * 0.1 for _ in range(n)]
    packed = {}
    for ko in (0, 1):
        ops.CHUNK_MAJOR = bool(ko)
        packed[ko] = [ops.pack_weight(w, b, torch.float16, DEV) for w, b in zip(ws, bs)]
        assert packed[ko][0].korder == ko
    ops.CHUNK_MAJOR = True
    kw = dict(stride=stride, upsample=up)
    if groups:
        kw["group_n"] = groups
    # torch reference (GPU fp32 math on the fp16-rounded operands)
    xin = x.float().permute(0, 3, 1, 2)
    if up:
        xin = torch.nn.functional.interpolate(xin, scale_factor=2.0, mode="nearest")
    refs, n0 = [], 0
    for gi in range(n):
        cnt = groups[gi] if groups else N
        refs.append(torch.nn.functional.conv2d(xin[n0:n0 + cnt], ws[gi].to(DEV).half().float(), bs[gi].to(DEV), stride=stride, padding=1))
        n0 += cnt
    ref = torch.cat(refs).permute(0, 2, 3, 1)
    Hout = ref.shape[1]
    fl = 2.0 * N * Hout * Hout * Cout * 9 * Cin
    line = f"[{si}] N={N} {H}->{Hout} {Cin}->{Cout} s{stride} up{int(up)} grouped={bool(groups)}:"
    for bn in bns:
        ops.FORCE_BN = bn
        try:
            graphs, outs = {}, {}
            for ko in (0, 1):
                pw = packed[ko] if groups else packed[ko][0]
                outs[ko] = ops.conv_gemm(x, pw, **kw).clone()
                graphs[ko] = capture(lambda pw=pw: ops.conv_gemm(x, pw, **kw))
        finally:
            ops.FORCE_BN = 0
        for _ in range(3):
            for ko in (0, 1):
                timed(graphs[ko])
        samples = {0: [], 1: []}
        for _ in range(9):
            for ko in (0, 1):
                samples[ko].append(timed(graphs[ko]))
        med = {ko: sorted(v)[len(v) // 2] for ko, v in samples.items()}
        errs = {ko: float((outs[ko].float() - ref).abs().max() / ref.abs().max()) for ko in (0, 1)}
        line += (f"  | bn{bn or 'auto'}: tap {med[0]:.1f} us ({fl / med[0] / 1e6:.0f} TF) chunk {med[1]:.1f} us ({fl / med[1] / 1e6:.0f} TF) "
                 f"x{med[0] / med[1]:.3f} err {errs[0]:.1e}/{errs[1]:.1e}")
        assert os.environ.get("ES_NO_CHECK") or max(errs.values()) < 3e-3, errs
    print(line, flush=True)
