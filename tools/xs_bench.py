"""es_linear_xs vs the tiled es_conv_gemm on the short-K projections of a batch-1 step (hipGraph replays, events)."""
import math
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("ES_XS_MIN_M", "0")          # no size policy: this tool is what the policy is fitted on
import torch
from edgestyle_amd import ops

DEV = "cuda"
g = torch.Generator().manual_seed(0)


def bench(fn, iters=20):
    fn(); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(4):
            fn()
    gr.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        gr.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (iters * 4)


shapes = [  # M, K, N, geglu, ln, groups
    (57344, 320, 2560, True, True, [8192, 24576, 16384, 8192]),
    (57344, 320, 960, False, True, [8192, 24576, 16384, 8192]),
    (57344, 320, 320, False, True, [8192, 24576, 16384, 8192]),
    (57344, 320, 320, False, False, [8192, 24576, 16384, 8192]),
    (14336, 640, 5120, True, True, [2048, 6144, 4096, 2048]),
    (14336, 640, 1920, False, True, [2048, 6144, 4096, 2048]),
    (14336, 640, 640, False, True, [2048, 6144, 4096, 2048]),
    (8192, 320, 2560, True, True, None),
    (8192, 320, 960, False, True, None),
    (8192, 320, 320, False, True, None),
    (2048, 640, 5120, True, True, None),
    (2048, 640, 1920, False, True, None),
    (2048, 640, 640, False, True, None),
    (8192, 320, 320, False, False, None),
    (3584, 640, 1920, False, True, None),
    (3584, 640, 640, False, True, None),
    (3584, 640, 5120, True, True, None),
    # the UNet decoder's own launches at batch 1 (2 samples) and 2 (4 samples): output projections with a residual, proj_in, cross to_q
    (8192, 320, 320, False, False, None, True),
    (16384, 320, 320, False, False, None, True),
    (16384, 320, 320, False, True, None),
    (16384, 320, 960, False, True, None),
    (32768, 320, 320, False, False, None, True),
    (32768, 320, 320, False, True, None),
]
for M, K, N, geglu, ln, groups, *rest in shapes:
    resid = bool(rest and rest[0])
    x = (torch.randn(M, K, generator=g)).to(DEV, torch.float16)
    n = len(groups) if groups else 1
    pws = []
    for _ in range(n):
        w = torch.randn(N, K, generator=g) / math.sqrt(K)
        b = torch.randn(N, generator=g) * 0.1
        if ln:
            pws.append(ops.pack_weight_ln(w, b, torch.ones(K), torch.zeros(K), 1e-5, torch.float16, DEV, geglu=geglu))
        else:
            pws.append(ops.pack_weight(w, b, torch.float16, DEV, geglu=geglu))
    pw = pws if groups else pws[0]
    kw = dict(group_n=groups) if groups else {}
    if resid:
        kw["residual"] = torch.randn(M, N, generator=g).to(DEV, torch.float16)
    res = {}
    for name, on in (("xs", True), ("tiled", False)):
        ops.XS_ENABLED = on
        res[name] = bench(lambda: ops.linear(x, pw, **kw))
    ops.XS_ENABLED = True
    fl = 2.0 * M * K * N
    print(f"M={M} K={K} N={N} geglu={geglu} ln={ln} res={resid} grouped={bool(groups)}: xs {res['xs']:.1f} us ({fl / res['xs'] / 1e6:.0f} TF)  "
          f"tiled {res['tiled']:.1f} us ({fl / res['tiled'] / 1e6:.0f} TF)  ratio {res['tiled'] / res['xs']:.2f}", flush=True)
