"""Cross-attention over the 77 text tokens: the K/V-resident kernel against the tiled kernels, same process, alternating graph replays.
   python3 tools/xattn_bench.py"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from edgestyle_amd import ops, lib

L = lib.load()
g = torch.Generator(device="cuda").manual_seed(0)


def graph_of(fn, reps=8):
    fn(); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(reps):
            fn()
    gr.replay(); torch.cuda.synchronize()
    return gr


def time_graph(gr, iters=10, reps=8):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        gr.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (iters * reps)


print("N heads Sq Skv d | tiled us | kv-resident us | x | HBM floor us (Q read + O write at 6 TB/s)")
for N, h, Sq, Skv, d in [(14, 8, 4096, 77, 40), (2, 8, 4096, 77, 40), (14, 8, 1024, 77, 80), (2, 8, 1024, 77, 80),
                         (112, 8, 4096, 77, 40), (16, 8, 4096, 77, 40), (112, 8, 1024, 77, 80), (16, 8, 1024, 77, 80)]:
    C = h * d
    q = torch.randn(N, Sq, C, generator=g, device="cuda").half()
    kv = torch.randn(N, Skv, 2 * C, generator=g, device="cuda").half()
    out = torch.empty_like(q)
    gs = {}
    for on in (0, 1):
        L.es_attention_set_kvres(2 * on)
        gs[on] = graph_of(lambda: ops.attention(q, kv[:, :, :C], kv[:, :, C:], h, out=out))
    L.es_attention_set_kvres(1)
    ts = {0: [], 1: []}
    for _ in range(7):
        for on in (0, 1):
            ts[on].append(time_graph(gs[on]))
    t0, t1 = statistics.median(ts[0]), statistics.median(ts[1])
    print(f"{N} {h} {Sq} {Skv} {d} | {t0:8.1f} | {t1:8.1f} | x{t0 / t1:.2f} | {2 * q.numel() * 2 / 6e12 * 1e6:.1f}", flush=True)
