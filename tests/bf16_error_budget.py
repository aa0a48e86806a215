#!/usr/bin/env python
"""configs[4] (768x768, bf16): where do the ~38 dB come from, and what would a WIDE RESIDUAL STREAM buy?
Test infrastructure (imports the oracle): an emulation on the CPU, no GPU involved.

The fp32 oracle is run three times on the request of tests/test_fullsize_gpu.py::test_config4_batch4_768_bf16_vs_oracle
(request 0, 2 DDIM steps, CFG 7.5, VAE decode):
  ref   fp32 everywhere (the checker the GPU tests compare with)
  all   every tensor the HIP path stores in HBM rounded to bf16: conv / linear / norm / activation / attention outputs AND the
        sums of the residual stream (x + h of a ResnetBlock, the three `+ h` of a transformer block, proj_out + res, conv_in +
        cond, skip + fused residual) - what the bf16 pipeline does
  wide  the same, but the residual-stream SUMS stay fp32 (their addends are still bf16-rounded branch outputs) and every reader of
        the stream sees the fp32 value: a pipeline that keeps the tensors every block adds into in a wider type
  acc   the sums exact, every READER of the stream (GroupNorm, conv_shortcut, LayerNorm-folded linears, ffo) sees them rounded to bf16:
        what the hi + lo stream of round 4 (es_gemm_desc.out_lo) implements - only the accumulation of rounding errors goes away
  acc+gn  ... and the GroupNorms read hi + lo

    ES_THREADS=8 python tests/bf16_error_budget.py [steps] [ref,all,wide,acc,acc+gn]        -> profiles/r04_bf16_error_budget.txt (about 10 CPU minutes)
"""
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import sd15_oracle as O                       # noqa: E402
from tests import helpers as H                            # noqa: E402
from tests.golden.make_golden_768 import weights_768, inputs_768   # noqa: E402

MODE = {"round": False, "stream": False, "operand": False, "gn": False}


def r(x):
    return x.bfloat16().float() if MODE["round"] else x


def rs(x):                                                # a residual-stream sum
    return x.bfloat16().float() if (MODE["round"] and MODE["stream"]) else x


def ro(x):                                                # the stream read as a GEMM operand (conv_shortcut tail, LayerNorm-folded linear, ffo's tokens)
    return x.bfloat16().float() if (MODE["round"] and MODE["operand"]) else x


def rg(x):                                                # the stream read by a GroupNorm
    return x.bfloat16().float() if (MODE["round"] and MODE["gn"]) else x


F = O.F
_orig = dict(conv=O.conv, linear=O.linear, group_norm=O.group_norm, layer_norm=O.layer_norm, resnet=O.resnet,
             transformer=O.transformer, attention=O.attention, silu=F.silu, gelu=F.gelu, sdpa=F.scaled_dot_product_attention)


def resnet(sd, p, x, temb, groups, eps):                  # O.resnet with the rounding points of the HIP path
    h = r(F.silu(_orig["group_norm"](sd, p + ".norm1", rg(x), groups, eps)))        # GroupNorm + SiLU: one kernel, one store
    h = _orig["conv"](sd, p + ".conv1", h)
    if temb is not None:
        h = h + r(_orig["linear"](sd, p + ".time_emb_proj", F.silu(temb)))[:, :, None, None]   # temb enters the fp32 accumulator
    h = r(h)
    h = r(F.silu(_orig["group_norm"](sd, p + ".norm2", h, groups, eps)))
    h = _orig["conv"](sd, p + ".conv2", h)
    if (p + ".conv_shortcut.weight") in sd:
        return rs(h + _orig["conv"](sd, p + ".conv_shortcut", ro(x), padding=0))  # folded: one fp32 accumulator, one rounding
    return rs(x + r(h))


def transformer(sd, p, x, ehs, heads, groups):
    b, c, hh, ww = x.shape
    res = x
    h = r(_orig["group_norm"](sd, p + ".norm", rg(x), groups, 1e-6))
    h = rs(_orig["conv"](sd, p + ".proj_in", h, padding=0))
    h = h.permute(0, 2, 3, 1).reshape(b, hh * ww, c)
    tb = p + ".transformer_blocks.0"

    def attn(q_in, ctx, pp):                             # LayerNorm is folded into the projections: no normalised copy is stored
        q = r(_orig["linear"](sd, pp + ".to_q", q_in)); k = r(_orig["linear"](sd, pp + ".to_k", ctx)); v = r(_orig["linear"](sd, pp + ".to_v", ctx))
        d = c // heads
        q, k, v = (t.view(b, -1, heads, d).transpose(1, 2) for t in (q, k, v))
        o = r(F.scaled_dot_product_attention(q, k, v)).transpose(1, 2).reshape(b, -1, c)
        return r(_orig["linear"](sd, pp + ".to_out.0", o))
    n = _orig["layer_norm"](sd, tb + ".norm1", ro(h))
    h = rs(attn(n, n, tb + ".attn1") + h)
    n = _orig["layer_norm"](sd, tb + ".norm2", ro(h))
    h = rs(attn(n, ehs, tb + ".attn2") + h)
    n = _orig["layer_norm"](sd, tb + ".norm3", ro(h))
    g = _orig["linear"](sd, tb + ".ff.net.0.proj", n)
    hidden, gate = g.chunk(2, dim=-1)
    f = r(hidden * F.gelu(gate))                          # GEGLU in the epilogue: one store
    # ff.net.2 + residual + proj_out + residual is ONE launch (linear end to end): one rounding of the result
    h2 = _orig["linear"](sd, tb + ".ff.net.2", f) + ro(h)
    h2 = h2.reshape(b, hh, ww, c).permute(0, 3, 1, 2)
    return rs(_orig["conv"](sd, p + ".proj_out", h2, padding=0) + res)


def conv(sd, p, x, stride=1, padding=1):                  # every other convolution (conv_in, down / upsamplers, zero-convs, VAE): one store
    return r(_orig["conv"](sd, p, x, stride, padding))


def install():
    O.resnet, O.transformer, O.conv = resnet, transformer, conv
    O.linear = lambda sd, p, x: r(_orig["linear"](sd, p, x))
    O.group_norm = lambda sd, p, x, g, e: r(_orig["group_norm"](sd, p, x, g, e))


def restore():
    O.resnet, O.transformer, O.conv, O.linear, O.group_norm = (_orig[k] for k in ("resnet", "transformer", "conv", "linear", "group_norm"))


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    torch.set_num_threads(int(os.environ.get("ES_THREADS", "8")))
    t0 = time.time()
    ucfg, vcfg, ws = weights_768()
    nets = H.oracle_nets(ws, ucfg)
    lat, pe, ne, pc = inputs_768()
    conds = [c.repeat(2, 1, 1, 1) for c in pc]
    out, lines = {}, []
    with torch.no_grad():
        variants = {"ref": (False, False, False, False), "all": (True, True, True, True), "wide": (True, False, False, False),
                    "acc": (True, False, True, True),      # built in round 4: sums exact (hi + lo), every reader sees hi
                    "acc+gn": (True, False, True, False)}  # ... and GroupNorm reads hi + lo
        want = sys.argv[2].split(",") if len(sys.argv) > 2 else ["ref", "all", "wide"]
        for name in want:
            MODE["round"], MODE["stream"], MODE["operand"], MODE["gn"] = variants[name]
            install()
            try:
                lat_out = O.pipeline(ws["unet"], ucfg, ws["fusion"], nets, ws["vae"], vcfg, lat[:1], pe[:1], ne[:1], conds,
                                     num_inference_steps=steps, guidance_scale=7.5, decode=False)
            finally:
                restore()
            MODE["round"] = False
            img = (O.vae_decode(ws["vae"], vcfg, lat_out / vcfg.scaling_factor) / 2 + 0.5).clamp(0, 1)   # decode in fp32: the loop's error only
            out[name] = (lat_out, img)
            print(f"{name}: {time.time() - t0:.0f} s", flush=True)
    for name in [n for n in out if n != "ref"]:
        lr = H.rel_err(out[name][0], out["ref"][0])
        lines.append(f"{name:5s} vs fp32: decoded image {H.psnr(out[name][1], out['ref'][1]):.2f} dB, final latents rel. err {lr:.3e}")
    head = (f"# tests/bf16_error_budget.py {steps}: CPU emulation of bf16 storage in the fp32 oracle, configs[4] request 0 (768x768, {steps} DDIM steps, CFG 7.5; "
            "decode in fp32)\n# all = every stored tensor bf16 (the HIP bf16 pipeline); wide = the same but the residual-stream sums stay fp32\n")
    txt = head + "\n".join(lines) + "\n"
    print(txt)
    tag = "" if len(sys.argv) <= 2 else "_" + sys.argv[2].replace(",", "_").replace("+", "")
    open(os.path.join(os.path.dirname(HERE), "profiles", f"r04_bf16_error_budget_{steps}steps{tag}.txt"), "w").write(txt)


if __name__ == "__main__":
    main()
