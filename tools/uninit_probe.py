"""Does any kernel of the (tiny) VAE decode read memory it did not write?  Poison the caching allocator's free blocks with
NaN / large values between two eager runs and compare layer by layer."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from edgestyle_amd import config as Cfg, ops, engine as E
from tests.helpers import make_weights, quantize
DEV = "cuda"
ucfg, vcfg = Cfg.tiny_unet(), Cfg.tiny_vae()
ws = {k: quantize(v) for k, v in make_weights(ucfg, vcfg, seed=2).items()}
vae = E.VAE(ws["vae"], vcfg, torch.float16, DEV)
g = torch.Generator().manual_seed(1)
z = torch.randn(2, 64, 64, 8, generator=g).to(DEV, torch.float16)
z[..., 4:] = 0

trace = []
orig = {}
for name in ("conv_gemm", "group_norm", "attention"):
    orig[name] = getattr(ops, name)
    def wrap(fn, name=name):
        def inner(*a, **k):
            y = fn(*a, **k)
            trace.append((name, tuple(y.shape), y.clone()))
            return y
        return inner
    setattr(ops, name, wrap(orig[name]))


def poison(val):
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    blocks = [torch.full((64 << 20,), val, dtype=torch.float16, device=DEV) for _ in range(24)]   # 3 GB
    ws_ = [w for w in ops._workspace.values()]
    for w in ws_:
        w.fill_(float("nan"))
    for p in list(ops._gn_partials.values()):
        p.fill_(float("nan"))
    torch.cuda.synchronize()
    del blocks


def run():
    trace.clear()
    out = vae.decode(z, unscaled_latents=True)
    torch.cuda.synchronize()
    return [(n, s, t) for n, s, t in trace], out.clone()


poison(0.0)
a, oa = run()
poison(float("nan"))
b, ob = run()
poison(777.0)
c, oc = run()
print("final: nan-poison has NaN:", bool(torch.isnan(ob).any()), " diff zero-vs-777 poison:", float((oa.float() - oc.float()).abs().max()))
for i, ((n, s, ta), (_, _, tb), (_, _, tc)) in enumerate(zip(a, b, c)):
    nan = bool(torch.isnan(tb).any())
    d = float((ta.float() - tc.float()).abs().max())
    if nan or d > 0:
        print(f"first divergence at op {i}: {n} out shape {s}: NaN under nan-poison={nan}, |zero-poison - 777-poison|={d}")
        break
else:
    print("no op diverged")
