"""es_linear_xs, plain (no GEGLU) launches: the two-barrier ping-pong form against the one-barrier form, same process, alternating
graph replays (medians of 7), outputs compared bit for bit.   python3 tools/xs_pp_bench.py [--big]"""
import math, os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("ES_XS_MIN_M", "0")
import torch
from edgestyle_amd import ops, lib

DEV = "cuda"
g = torch.Generator().manual_seed(0)
L = lib.load()


def graph_of(fn, reps=4):
    fn(); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(reps):
            fn()
    gr.replay(); torch.cuda.synchronize()
    return gr


def time_graph(gr, iters, reps=4):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        gr.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (iters * reps)


big = "--big" in sys.argv
shapes = [  # M, K, N, ln, residual, groups
    (57344, 320, 960, True, False, [8192, 24576, 16384, 8192]),
    (57344, 320, 320, True, False, [8192, 24576, 16384, 8192]),
    (57344, 320, 320, False, True, [8192, 24576, 16384, 8192]),
    (57344, 320, 320, False, False, None),
    (14336, 640, 1920, True, False, [2048, 6144, 4096, 2048]),
    (14336, 640, 640, True, False, [2048, 6144, 4096, 2048]),
    (8192, 320, 960, True, False, None),
    (8192, 320, 320, False, True, None),
    (8000, 320, 320, False, True, None),          # ragged M
    (300, 640, 1920, True, False, None),          # one ragged row block, sliced N
]
if big:
    shapes = [(458752, 320, 960, True, False, None), (458752, 320, 320, False, True, None), (458752, 320, 320, True, False, None),
              (114688, 640, 1920, True, False, None), (114688, 640, 640, True, False, None)]
for M, K, N, ln, res, groups in shapes:
    x = torch.randn(M, K, generator=g).to(DEV, torch.float16)
    r = torch.randn(M, N, generator=g).to(DEV, torch.float16) if res else None
    n = len(groups) if groups else 1
    pws = []
    for _ in range(n):
        w = torch.randn(N, K, generator=g) / math.sqrt(K)
        b = torch.randn(N, generator=g) * 0.1
        pws.append(ops.pack_weight_ln(w, b, 1 + 0.1 * torch.randn(K, generator=g), 0.1 * torch.randn(K, generator=g), 1e-5, torch.float16, DEV) if ln
                   else ops.pack_weight(w, b, torch.float16, DEV))
    pw = pws if groups else pws[0]
    kw = dict(group_n=groups) if groups else {}
    if res:
        kw["residual"] = r
    outs, graphs = {}, {}
    for pp in (0, 1):
        L.es_linear_xs_set_pp(pp)
        outs[pp] = ops.linear(x, pw, **kw).clone()
        graphs[pp] = graph_of(lambda: ops.linear(x, pw, **kw))
    L.es_linear_xs_set_pp(1)
    torch.cuda.synchronize()
    same = torch.equal(outs[0], outs[1])
    ts = {0: [], 1: []}
    for _ in range(7):
        for pp in (0, 1):
            ts[pp].append(time_graph(graphs[pp], 10 if M > 100000 else 20))
    t0, t1 = statistics.median(ts[0]), statistics.median(ts[1])
    fl = 2.0 * M * K * N
    print(f"M={M} K={K} N={N} ln={ln} res={res} grouped={bool(groups)}: one-barrier {t0:.1f} us ({fl / t0 / 1e6:.0f} TF)  ping-pong {t1:.1f} us "
          f"({fl / t1 / 1e6:.0f} TF)  x{t0 / t1:.2f}  bitwise_equal={same}  finite={bool(torch.isfinite(outs[1]).all())}", flush=True)
