"""es_linear_xs: N slices per row block when the row blocks alone are between one and two rounds of 256 workgroups (batch 8, 32 x 32 level:
448 row blocks = 1.75 rounds; 4 slices = 7.0 rounds).  Graph replays, interleaved, medians.   python3 tools/xs_slices_bench.py"""
import math, os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("ES_XS_MIN_M", "0")
import torch
from edgestyle_amd import ops

DEV = "cuda"
g = torch.Generator().manual_seed(0)


def graph_of(fn, reps=3):
    fn(); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(reps):
            fn()
    gr.replay(); torch.cuda.synchronize()
    return gr


def time_graph(gr, iters=8, reps=3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        gr.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (iters * reps)


shapes = [  # M, K, N, geglu, ln
    (114688, 640, 10240, True, True), (114688, 640, 1920, False, True), (114688, 640, 640, False, True), (114688, 640, 640, False, False),
    (129024, 640, 10240, True, True),      # 768 x 768, batch 4: 504 row blocks
    (86016, 640, 10240, True, True),       # batch 6: 336 row blocks
]
for M, K, N, geglu, ln in shapes:
    x = torch.randn(M, K, generator=g).to(DEV, torch.float16)
    w = torch.randn(N, K, generator=g) / math.sqrt(K)
    b = torch.randn(N, generator=g) * 0.1
    pw = (ops.pack_weight_ln(w, b, torch.ones(K), torch.zeros(K), 1e-5, torch.float16, DEV, geglu=geglu) if ln
          else ops.pack_weight(w, b, torch.float16, DEV, geglu=geglu))
    outs, graphs = {}, {}
    for ns in (0, 2, 3, 4):
        ops.XS_FORCE_SLICES = ns
        outs[ns] = ops.linear(x, pw).clone()
        graphs[ns] = graph_of(lambda: ops.linear(x, pw))
    ops.XS_FORCE_SLICES = 0
    ts = {k: [] for k in graphs}
    for _ in range(5):
        for k in graphs:
            ts[k].append(time_graph(graphs[k]))
    fl = 2.0 * M * K * N
    base = statistics.median(ts[0])
    print(f"M={M} ({(M + 255) // 256} row blocks) K={K} N={N} geglu={geglu} ln={ln}: " + "  ".join(
        f"slices {k or 'policy(1)'}: {statistics.median(v):.1f} us ({fl / statistics.median(v) / 1e6:.0f} TF, x{base / statistics.median(v):.2f}, same={torch.equal(outs[0], outs[k])})"
        for k, v in ts.items()), flush=True)
