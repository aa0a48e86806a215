"""Child process of tests/test_load_weights_gpu.py::test_the_profiling_recipe_...: builds a small es_load_weights context and runs
RGB conditions -> es_denoise_loop -> es_vae_decode through the C ABI with per-plan graphs and with the whole-loop graph.  Meant to
be wrapped in `rocprofv3 --kernel-trace -- python3 tests/run_native_loop.py` (no GPU use before this program starts)."""
import dataclasses
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from edgestyle_amd import config as Cfg                     # noqa: E402
from edgestyle_amd.native import NativeContext              # noqa: E402
from tests.helpers import make_weights, quantize            # noqa: E402

B, T = 2, 12
ucfg, vcfg = dataclasses.replace(Cfg.tiny_unet(), sample_size=64), Cfg.tiny_vae()
ws = {k: quantize(v) for k, v in make_weights(ucfg, vcfg, seed=5).items()}
nat = NativeContext(ws, ucfg, vcfg, batch_size=B, guidance=True, num_inference_steps=T, device=0)
g = torch.Generator().manual_seed(1)
s, px = ucfg.sample_size, ucfg.sample_size * vcfg.scale
imgs = [torch.rand(B, 3, px, px, generator=g).cuda() for _ in range(6)]
noise = [torch.randn(2 * B, vcfg.latent_channels, s, s, generator=g).cuda() if i % 2 == 0 else None for i in range(6)]
ehs = (torch.randn(2 * B, 77, ucfg.cross_attention_dim, generator=g) * 0.5).cuda().half()
x0 = torch.randn(B, s, s, 4, generator=g).cuda()
ts = [float(1 + (1000 // T) * i) for i in reversed(range(T))]
outs = []
for graphs in (1, 2, 1, 2):
    nat.set_options(use_graphs=graphs)
    x = x0.clone()
    nat.prepare_conds(imgs, noise)
    nat.denoise_loop(x, ehs, 7.5, ts)
    outs.append(nat.vae_decode(x).clone())
    torch.cuda.synchronize()
assert all(torch.equal(outs[0], o) for o in outs[1:]) and bool(torch.isfinite(outs[0]).all())
nat.close()
print("native loop under the profiler: ok")
