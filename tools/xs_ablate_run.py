"""Times the ablated es_linear_xs builds of tools/xs_ablate.sh (one process per variant: ES_HIP_LIB is read at import)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = "import os; os.environ[\"ES_XS_MIN_M\"]=\"0\"\n" + r'''
import math, sys, os
sys.path.insert(0, %r)
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
out = []
for M, K, N, geglu in [(57344, 320, 2560, True), (57344, 320, 960, False), (14336, 640, 5120, True), (2048, 640, 1920, False)]:
    x = torch.randn(M, K, generator=g).to(DEV, torch.float16)
    w = torch.randn(N, K, generator=g) / math.sqrt(K)
    pw = ops.pack_weight_ln(w, torch.randn(N, generator=g) * 0.1, torch.ones(K), torch.zeros(K), 1e-5, torch.float16, DEV, geglu=geglu)
    out.append("%%.1f" %% bench(lambda: ops.linear(x, pw)))
print(" ".join(out))
''' % ROOT
print("variant: (57344,320,2560,geglu) (57344,320,960) (14336,640,5120,geglu) (2048,640,1920)  [us]")
for a in os.environ.get("XS_VARIANTS", "0 1 2 4 3 7 39 63").split():
    env = dict(os.environ, ES_HIP_LIB=os.path.join(ROOT, "edgestyle_amd", "lib", "ablate", f"libes_xs_{a}.so"))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    print(f"XS_ABLATE={a}: {r.stdout.strip()} {r.stderr.strip()[-200:] if r.returncode else ''}", flush=True)
