import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dataclasses, torch
from edgestyle_amd import config as Cfg
from tests.helpers import make_weights, quantize
from edgestyle_amd.models import StepRunner, AutoencoderKL, _as_nhwc
from edgestyle_amd.pipeline import EdgeStyleStableDiffusionControlNetPipeline
from edgestyle_amd.native import NativeEngine
DEV = "cuda"
ucfg, vcfg = dataclasses.replace(Cfg.tiny_unet(), sample_size=64), Cfg.tiny_vae()
ws = {k: quantize(v) for k, v in make_weights(ucfg, vcfg, seed=2).items()}
runner = StepRunner.from_state_dicts(ws, ucfg, torch.float16, DEV)
vae = AutoencoderKL(ws["vae"], vcfg).to(DEV)
pipe = EdgeStyleStableDiffusionControlNetPipeline(vae=vae, unet=runner.unet, controlnet=runner.controlnet).to(DEV)
eng = NativeEngine(pipe, batch_size=1, num_inference_steps=6)
g = torch.Generator().manual_seed(23)
s, c0 = ucfg.sample_size, ucfg.block_out_channels[0]
lat = torch.randn(1, 4, s, s, generator=g)
pe = (torch.randn(1, 77, ucfg.cross_attention_dim, generator=g) * 0.5).half().float()
ne = (torch.randn(1, 77, ucfg.cross_attention_dim, generator=g) * 0.5).half().float()
conds = [(torch.randn(1, c0, s, s, generator=g) * 0.3).half().float() for _ in range(6)]
kw = dict(prompt_embeds=pe, negative_prompt_embeds=ne, image=conds, latents=lat, guidance_scale=5.0, num_inference_steps=6)
loop = eng.loop
names = ["coef", "t_table", "scale_table", "ehs", "model_in", "latents"]
def snap():
    d = {n: getattr(loop, n).clone() for n in names}
    d["tproj"] = loop.state.tproj_table.clone(); d["cond_cat"] = loop.state.cond_cat.clone()
    d["ctx0"] = loop.state.ctx_grouped[0].clone(); d["conds0"] = loop.conds[0].clone()
    return d
import ctypes as C
from edgestyle_amd import lib as L
lib = L.load()
per = []
def cb(p, i, t, kw):
    per.append(kw["latents"].clone())
    return {}
pipe.use_graph = False
want = pipe(output_type="latent", callback_on_step_end=cb, **kw).images.clone(); coefA = loop.coef.clone()
snapA = snap()
# native: full loop once (fills tables / prep), then redo step by step
eng.set_conds([_as_nhwc(c.repeat(2, 1, 1, 1), torch.float16, DEV) for c in conds])
ehs = torch.cat([ne, pe]).to(DEV, torch.float16).contiguous()
x = lat.permute(0, 2, 3, 1).contiguous().to(DEV)
eng.set_options(use_graphs=False)
got = eng.denoise_loop(x.clone(), ehs, 5.0)
print("coef diff", float((loop.coef - coefA).abs().max())); print("full loop diff", float((got.permute(0, 3, 1, 2) - want).abs().max()))
# manual: reset state like es_denoise_loop, then single-step
loop.latents.copy_(x); loop.step_idx.zero_(); loop.set_model_in()
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
gs = C.c_float(5.0)
assert lib.es_ctx_launch_plan(eng.ctx, L.PLAN_PREP, None, st) == 0
for i in range(6):
    assert lib.es_ctx_launch_plan(eng.ctx, L.PLAN_STEP, C.byref(gs), st) == 0
    torch.cuda.synchronize()
    print("step", i, "latents diff", float((loop.latents.permute(0, 3, 1, 2) - per[i]).abs().max()), "noise abs", float(loop.noise.float().abs().max()))
# python eager single steps for comparison
loop.latents.copy_(x); loop.step_idx.zero_(); loop.set_model_in()
for i in range(6):
    loop.one_step(); torch.cuda.synchronize()
    print("py step", i, "latents diff", float((loop.latents.permute(0, 3, 1, 2) - per[i]).abs().max()))
