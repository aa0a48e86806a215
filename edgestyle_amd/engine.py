"""Host-side executors: SD1.5 UNet / ControlNet / AutoencoderKL / fusion blocks as sequences of C-ABI kernel launches.

Everything here is plumbing: weights are packed once into the layouts the kernels want (NHWC activations,
[Cout][ky,kx,Cin] K-contiguous weights, fused QKV / KV projections, GEGLU-interleaved FF rows, LoRA folded into
private copies, all ResnetBlock time projections concatenated into ONE GEMM) and `forward` is a flat list of
`ops.*` launches on the current HIP stream — no torch arithmetic on the hot path, so a whole denoising step
captures into one hipGraph.

Module trees and forward order follow diffusers==0.26.3 (UNet2DConditionModel, ControlNetModel, AutoencoderKL) as
called from model/controllora.py:150-254 and model/edgestyle_pipeline.py:477-557; key names are the diffusers
state-dict names (edgestyle_amd/weights.py).
"""
import os
import weakref
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import lib as L
from . import ops
from .config import UNetConfig, VAEConfig

SD = Dict[str, torch.Tensor]


def fold_lora(sd: SD) -> SD:
    """W <- W + B.A for every `<linear>.lora_layer.{down,up}.weight` pair, into a NEW dict (the reference's
    fuse_lora mutates the tied UNet parameters in place, CL:728-777 — we never do)."""
    out = {k: v for k, v in sd.items() if ".lora_layer." not in k}
    for k in sd:
        if k.endswith(".lora_layer.down.weight"):
            base = k[: -len(".lora_layer.down.weight")]
            # in double, rounded once: the result does not depend on the summation order of whoever folds (this host or
            # es_load_weights, csrc/builder.hip)
            up = sd[base + ".lora_layer.up.weight"].double()
            out[base + ".weight"] = (sd[base + ".weight"].double() + up @ sd[k].double()).float()
    return out


_PACK_CACHE = {}     # (id(weight), id(bias), dtype, device, layout) -> (weakref(weight), weakref(bias), PackedWeight)


class _Packer:
    """Packs on the target device (torch used for data movement only)."""

    def __init__(self, sd: SD, dtype, device):
        self.sd, self.dtype, self.device = sd, dtype, device

    def t(self, key):
        return self.sd[key].to(self.device, torch.float32)

    def conv(self, p: str, cin_pad: Optional[int] = None, geglu=False, cout_pad=None) -> ops.PackedWeight:
        """Packed once per SOURCE tensor: ControlLoRA nets alias the UNet's tensors for everything they do not adapt
        (tie_weights, CL:623-632: all convs, norms), so their engines share the UNet's packed copies - 0.9 GB less
        HBM, and in the grouped launches the deep-level conv weights are streamed once instead of three times
        (level 3: 60 -> 45 us per launch)."""
        w = self.sd[p + ".weight"]
        b = self.sd.get(p + ".bias")
        key = (id(w), None if b is None else id(b), str(self.dtype), str(self.device), geglu, cin_pad, cout_pad)
        hit = _PACK_CACHE.get(key)
        if hit is not None and hit[0]() is w and (b is None or hit[1]() is b):
            return hit[2]
        pw = ops.pack_weight(self.t(p + ".weight"), None if b is None else b.to(self.device), self.dtype,
                             self.device, geglu=geglu, cin_pad=cin_pad, cout_pad=cout_pad)
        _PACK_CACHE[key] = (weakref.ref(w), None if b is None else weakref.ref(b), pw)
        return pw

    def conv_tail(self, p: str, pt: str) -> ops.PackedWeight:
        """conv `p` with the 1x1 conv `pt` appended along K (ops.pack_weight_tail); cached per source tensors like conv()."""
        w, wt = self.sd[p + ".weight"], self.sd[pt + ".weight"]
        key = (id(w), id(wt), str(self.dtype), str(self.device), "tail")
        hit = _PACK_CACHE.get(key)
        if hit is not None and hit[0]() is w and hit[1]() is wt:
            return hit[2]
        pw = ops.pack_weight_tail(self.t(p + ".weight"), self.t(pt + ".weight"), self.t(p + ".bias") + self.t(pt + ".bias"),
                                  self.dtype, self.device)
        _PACK_CACHE[key] = (weakref.ref(w), weakref.ref(wt), pw)
        return pw

    def cat(self, ps: Sequence[str], bias: bool) -> ops.PackedWeight:
        w = torch.cat([self.t(p + ".weight") for p in ps], 0)
        b = torch.cat([self.t(p + ".bias") for p in ps], 0) if bias else None
        return ops.pack_weight(w, b, self.dtype, self.device)

    def norm(self, p: str) -> Tuple[torch.Tensor, torch.Tensor]:
        return self.t(p + ".weight").contiguous(), self.t(p + ".bias").contiguous()

    def cat_ln(self, ps: Sequence[str], bias: bool, ln: str, geglu: bool = False) -> ops.PackedWeight:
        """Linear layer(s) `ps` with the LayerNorm `ln` in front of them folded in (ops.pack_weight_ln)."""
        w = torch.cat([self.t(p + ".weight") for p in ps], 0)
        b = torch.cat([self.t(p + ".bias") for p in ps], 0) if bias else None
        return ops.pack_weight_ln(w, b, self.t(ln + ".weight"), self.t(ln + ".bias"), 1e-5, self.dtype, self.device,
                                  geglu=geglu)


SHORTCUT_FOLD = os.environ.get("ES_SHORTCUT_FOLD", "1") == "1"   # conv_shortcut folded into conv2 (Resnet.conv2s)
FFO_FOLD = os.environ.get("ES_FFO_FOLD", "1") == "1"   # ff.net.2 + proj_out as one GEMM (Transformer.ffo)
LN_FOLD = os.environ.get("ES_LN_FOLD", "1") in ("1", "qk")
LN_FOLD_FF = os.environ.get("ES_LN_FOLD", "1") == "1"    # "qk": norm1/norm2 only, norm3 -> GEGLU stays a LayerNorm launch     # LayerNorm folded into the Linear it feeds (es_gemm_desc.ln_colsum)


class Resnet:
    def __init__(self, pk: _Packer, p: str, groups: int, eps: float, temb_off: Optional[int]):
        self.n1, self.n2 = pk.norm(p + ".norm1"), pk.norm(p + ".norm2")
        self.conv1, self.conv2 = pk.conv(p + ".conv1"), pk.conv(p + ".conv2")
        self.short = pk.conv(p + ".conv_shortcut") if (p + ".conv_shortcut.weight") in pk.sd else None
        self.groups, self.eps, self.temb_off = groups, eps, temb_off
        self.cout = self.conv1.cout
        # conv2(h) + conv_shortcut(x) is one sum over K: the shortcut's 1x1 weights ride behind the 3x3 taps of conv2 as
        # "tail" channels read from x (| x2) at the output pixel (es_gemm_desc.t1/t2, ops.pack_weight_tail) - one launch
        # and one write + re-read of the shortcut tensor less per block, and the shortcut enters the fp32 accumulator
        # instead of being rounded to 16 bits first.  Needs 64-aligned channels.
        self.conv2s = None
        w2, ws_ = pk.sd[p + ".conv2.weight"], pk.sd.get(p + ".conv_shortcut.weight")
        if SHORTCUT_FOLD and ws_ is not None and w2.shape[1] % 64 == 0 and ws_.shape[1] % 64 == 0:
            self.conv2s = pk.conv_tail(p + ".conv2", p + ".conv_shortcut")

    def __call__(self, x, tproj, x2=None):
        h = ops.group_norm(x, self.n1[0], self.n1[1], self.groups, self.eps, True, x2=x2)
        temb = None if self.temb_off is None else tproj[:, self.temb_off:]
        # (gn_groups: the launch's epilogue hands the GroupNorm statistics of its output over to the GroupNorm that reads it -
        #  norm2 here, the next block's norm1 / Transformer2DModel.norm for the block's output; ops.GN_HANDOVER)
        h = ops.conv_gemm(h, self.conv1, temb=temb, gn_groups=self.groups)
        h = ops.group_norm(h, self.n2[0], self.n2[1], self.groups, self.eps, True)
        if self.conv2s is not None and (x2 is None or (x.shape[3] % 64 == 0 and x2.shape[3] % 64 == 0)):
            return ops.conv_gemm(h, self.conv2s, tail=(x, x2), gn_groups=self.groups)
        xs = ops.conv_gemm(x, self.short, x2=x2) if self.short is not None else x
        return ops.conv_gemm(h, self.conv2, residual=xs, wide=True, gn_groups=self.groups)   # x + h: a sum of the residual stream (ops.WIDE_STREAM)


class Transformer:
    """Transformer2DModel(use_linear_projection=False) + one BasicTransformerBlock."""

    def __init__(self, pk: _Packer, p: str, heads: int, groups: int):
        tb = p + ".transformer_blocks.0"
        self.norm = pk.norm(p + ".norm")
        self.proj_in, self.proj_out = pk.conv(p + ".proj_in"), pk.conv(p + ".proj_out")
        self.ln1, self.ln2, self.ln3 = pk.norm(tb + ".norm1"), pk.norm(tb + ".norm2"), pk.norm(tb + ".norm3")
        self.qkv = pk.cat([tb + ".attn1.to_q", tb + ".attn1.to_k", tb + ".attn1.to_v"], bias=False)
        self.o1 = pk.conv(tb + ".attn1.to_out.0")
        self.q2 = pk.conv(tb + ".attn2.to_q")
        self.kv2 = pk.cat([tb + ".attn2.to_k", tb + ".attn2.to_v"], bias=False)
        self.o2 = pk.conv(tb + ".attn2.to_out.0")
        self.ff1 = pk.conv(tb + ".ff.net.0.proj", geglu=True)
        self.ff2 = pk.conv(tb + ".ff.net.2")
        self.heads, self.groups = heads, groups
        self.c = self.proj_in.cout
        # ff.net.2 -> (+ tokens) -> proj_out is linear end to end (SD1.5 has ONE transformer block per
        # Transformer2DModel): proj_out(ff2(f) + tok) = (Wp Wf) f + Wp tok + (Wp bf + bp), i.e. one GEMM over the
        # channel concat [f | tok] (K = 4C + C) with pre-multiplied weights (fp32 product, rounded once) - one launch
        # and one write + re-read of the tokens less per transformer, same FLOPs.
        self.ffo = None
        if FFO_FOLD and self.c % 64 == 0:
            wp = pk.t(p + ".proj_out.weight").reshape(self.c, self.c).double()
            wf, bf = pk.t(tb + ".ff.net.2.weight").double(), pk.t(tb + ".ff.net.2.bias").double()
            self.ffo = ops.pack_weight(torch.cat([wp @ wf, wp], 1).float(), (wp @ bf + pk.t(p + ".proj_out.bias").double()).float(),
                                       pk.dtype, pk.device)
        # norm1 / norm2 / norm3 each feed exactly one Linear (QKV, to_q, GEGLU projection): folded into it, the
        # LayerNorm launch and the write + re-read of the normalised tokens disappear; needs 64-aligned channels
        # (ops.pack_weight_ln).  Measured per denoising step at batch 1: none 1.869, norm1+2 1.899, all three 1.905
        # images/s (in isolation the GEGLU launch loses more than a LayerNorm launch costs - every one of its 20-80 N
        # tiles redoes the row statistics - but inside the step the fold still wins).
        self.ln_fold = LN_FOLD and self.c % 64 == 0
        if self.ln_fold:
            self.qkv_ln = pk.cat_ln([tb + ".attn1.to_q", tb + ".attn1.to_k", tb + ".attn1.to_v"], False, tb + ".norm1")
            self.q2_ln = pk.cat_ln([tb + ".attn2.to_q"], False, tb + ".norm2")
            self.ff1_ln = pk.cat_ln([tb + ".ff.net.0.proj"], True, tb + ".norm3", geglu=True) if LN_FOLD_FF else None

    def context(self, ehs, out=None, rep: int = 1):
        """K/V projection of the text states [N,77,D] -> [rep*N,77,2C] (rep copies: the nets of a weight-sharing group
        see the same text states); constant over the denoising loop."""
        return ops.linear(ehs, self.kv2, out=out, x_rep=rep)

    def __call__(self, x, kv):
        N, H, W, C = x.shape
        tok = ops.gn_proj_in(x, self.norm[0], self.norm[1], self.groups, 1e-6, self.proj_in).reshape(N, H * W, C)
        fold = self.ln_fold
        qkv = ops.linear(tok, self.qkv_ln) if fold else ops.linear(ops.layer_norm(tok, *self.ln1), self.qkv)
        a = ops.attention(qkv[:, :, :C], qkv[:, :, C:2 * C], qkv[:, :, 2 * C:], self.heads)
        tok = ops.linear(a, self.o1, residual=tok, wide=True)             # the three sums of the token stream and the block's own
        q = ops.linear(tok, self.q2_ln) if fold else ops.linear(ops.layer_norm(tok, *self.ln2), self.q2)
        a = ops.attention(q, kv[:, :, :C], kv[:, :, C:], self.heads)
        tok = ops.linear(a, self.o2, residual=tok, wide=True)
        f = ops.linear(tok, self.ff1_ln) if (fold and self.ff1_ln is not None) else ops.linear(ops.layer_norm(tok, *self.ln3), self.ff1)
        if self.ffo is not None:
            return ops.conv_gemm(f.reshape(N, H, W, 4 * C), self.ffo, x2=tok.reshape(N, H, W, C), residual=x, wide=True, gn_groups=self.groups)
        tok = ops.linear(f, self.ff2, residual=tok, wide=True)
        return ops.conv_gemm(tok.reshape(N, H, W, C), self.proj_out, residual=x, wide=True, gn_groups=self.groups)


class Encoder:
    """conv_in + time_embedding + down_blocks + mid_block: the part a ControlNet shares with the UNet (CL:623-632)."""

    def __init__(self, sd: SD, cfg: UNetConfig, dtype, device, extra_resnets: Sequence[str] = ()):
        sd = fold_lora(sd)
        self.cfg, self.dtype, self.device = cfg, dtype, device
        pk = self.pk = _Packer(sd, dtype, device)
        g, eps, heads = cfg.norm_num_groups, cfg.norm_eps, cfg.num_heads
        self.in_pad = (cfg.in_channels + 7) // 8 * 8
        self.conv_in = pk.conv("conv_in", cin_pad=self.in_pad)
        self.t1, self.t2 = pk.conv("time_embedding.linear_1"), pk.conv("time_embedding.linear_2")
        # every resnet's time_emb_proj concatenated into one GEMM; offsets handed to the resnets
        names = []
        ch = cfg.block_out_channels
        for i in range(len(ch)):
            for j in range(cfg.layers_per_block):
                names.append(f"down_blocks.{i}.resnets.{j}")
        names += ["mid_block.resnets.0", "mid_block.resnets.1"] + list(extra_resnets)
        offs, o = {}, 0
        for nme in names:
            offs[nme] = o
            o += sd[nme + ".time_emb_proj.weight"].shape[0]
        self.tproj = pk.cat([n + ".time_emb_proj" for n in names], bias=True)
        self.temb_offs = offs
        self.down: List[List[Tuple[Resnet, Optional[Transformer]]]] = []
        self.downsample: List[Optional[ops.PackedWeight]] = []
        for i in range(len(ch)):
            blk = []
            for j in range(cfg.layers_per_block):
                r = Resnet(pk, f"down_blocks.{i}.resnets.{j}", g, eps, offs[f"down_blocks.{i}.resnets.{j}"])
                a = Transformer(pk, f"down_blocks.{i}.attentions.{j}", heads, g) if cfg.down_has_attn[i] else None
                blk.append((r, a))
            self.down.append(blk)
            self.downsample.append(pk.conv(f"down_blocks.{i}.downsamplers.0.conv") if i != len(ch) - 1 else None)
        self.mid0 = Resnet(pk, "mid_block.resnets.0", g, eps, offs["mid_block.resnets.0"])
        self.mid_attn = Transformer(pk, "mid_block.attentions.0", heads, g)
        self.mid1 = Resnet(pk, "mid_block.resnets.1", g, eps, offs["mid_block.resnets.1"])

    # ---- pieces -------------------------------------------------------------------------------------------
    def transformers(self) -> List[Transformer]:
        out = [a for blk in self.down for (_, a) in blk if a is not None]
        return out + [self.mid_attn]

    def context(self, ehs, outs: Optional[List[torch.Tensor]] = None) -> List[torch.Tensor]:
        """`outs` (a previous result of matching shape) is overwritten in place: pointers baked into a captured
        graph stay valid across pipeline calls."""
        tr = self.transformers()
        if outs is not None and outs[0].shape[0] == ehs.shape[0]:
            return [t.context(ehs, o) for t, o in zip(tr, outs)]
        return [t.context(ehs) for t in tr]

    def time_proj(self, t_dev: torch.Tensor, step_idx: Optional[torch.Tensor] = None) -> torch.Tensor:
        """CL:150-157 + every ResnetBlock2D.time_emb_proj(silu(emb)) in one shot -> [N, sum(Cout)]."""
        e = ops.timestep_embedding(t_dev, self.cfg.block_out_channels[0], self.dtype)
        e = ops.linear(e, self.t1, act=L.ACT_SILU)
        e = ops.linear(e, self.t2, act=L.ACT_SILU)     # emb is only ever consumed through silu(emb)
        return ops.linear(e, self.tproj)

    @property
    def tproj_width(self) -> int:
        return self.tproj.cout

    def run(self, h, tproj, ctx: List[torch.Tensor]):
        skips = [h]
        ci = 0
        for i, blk in enumerate(self.down):
            for r, a in blk:
                h = r(h, tproj)
                if a is not None:
                    h = a(h, ctx[ci]); ci += 1
                skips.append(h)
            if self.downsample[i] is not None:
                h = ops.conv_gemm(h, self.downsample[i], stride=2, gn_groups=self.cfg.norm_num_groups)
                skips.append(h)
        h = self.mid0(h, tproj)
        h = self.mid_attn(h, ctx[ci])
        h = self.mid1(h, tproj)
        return skips, h


class UNet(Encoder):
    def __init__(self, sd: SD, cfg: UNetConfig, dtype, device):
        n = len(cfg.block_out_channels)
        extra = [f"up_blocks.{i}.resnets.{j}" for i in range(n) for j in range(cfg.layers_per_block + 1)]
        super().__init__(sd, cfg, dtype, device, extra_resnets=extra)
        pk, g, eps, heads = self.pk, cfg.norm_num_groups, cfg.norm_eps, cfg.num_heads
        self.up: List[List[Tuple[Resnet, Optional[Transformer]]]] = []
        self.upsample: List[Optional[ops.PackedWeight]] = []
        for i in range(n):
            blk = []
            for j in range(cfg.layers_per_block + 1):
                nm = f"up_blocks.{i}.resnets.{j}"
                r = Resnet(pk, nm, g, eps, self.temb_offs[nm])
                a = Transformer(pk, f"up_blocks.{i}.attentions.{j}", heads, g) if cfg.up_has_attn[i] else None
                blk.append((r, a))
            self.up.append(blk)
            self.upsample.append(pk.conv(f"up_blocks.{i}.upsamplers.0.conv") if i != n - 1 else None)
        self.norm_out = pk.norm("conv_norm_out")
        self.conv_out = pk.conv("conv_out")
        del self.pk

    def transformers(self):
        return super().transformers() + [a for blk in self.up for (_, a) in blk if a is not None]

    def encode(self, x, tproj, ctx):
        """conv_in + down blocks + mid block: independent of the ControlNet residuals, so it can run concurrently
        with the ControlNet passes on another stream."""
        h = ops.conv_gemm(x, self.conv_in, gn_groups=self.cfg.norm_num_groups)
        return self.run(h, tproj, ctx)

    def forward(self, x, tproj, ctx, down_res: Optional[Sequence] = None, mid_res=None, out=None, encoded=None,
                presummed: bool = False):
        """x: [N,H,W,in_pad] -> noise prediction [N,H,W,out_channels] (PL:500-510).  presummed: down_res / mid_res
        already hold skip + residual (the fusion kernel added the encoder outputs)."""
        skips, h = encoded if encoded is not None else self.encode(x, tproj, ctx)
        skips = list(skips)
        if presummed:
            skips = [r.reshape(s.shape) for s, r in zip(skips, down_res)]
            h = mid_res.reshape(h.shape)
        else:
            if down_res is not None:
                skips = [ops.add(s, r.reshape(s.shape)) for s, r in zip(skips, down_res)]
            if mid_res is not None:
                h = ops.add(h, mid_res.reshape(h.shape))
        ci = len(super().transformers())
        cfg = self.cfg
        for i, blk in enumerate(self.up):
            for r, a in blk:
                h = r(h, tproj, x2=skips.pop())
                if a is not None:
                    h = a(h, ctx[ci]); ci += 1
            if self.upsample[i] is not None:
                h = ops.conv_gemm(h, self.upsample[i], upsample=True)
        h = ops.group_norm(h, self.norm_out[0], self.norm_out[1], cfg.norm_num_groups, cfg.norm_eps, True)
        return ops.conv_gemm(h, self.conv_out, out=out)


class ControlNet(Encoder):
    """ControlNetModel / CachedControlNetModel / (fused) ControlLoRAModel body (CL:58-290)."""

    def __init__(self, sd: SD, cfg: UNetConfig, dtype, device, uses_vae: bool = False):
        super().__init__(sd, cfg, dtype, device)
        pk = self.pk
        table = cfg.residual_table()
        self.zero = [pk.conv(f"controlnet_down_blocks.{i}") for i in range(len(table) - 1)]
        self.zero_mid = pk.conv("controlnet_mid_block")
        self.uses_vae = uses_vae
        self.cond = None
        if not uses_vae and "controlnet_cond_embedding.conv_in.weight" in pk.sd:
            ce = cfg.conditioning_embedding_out_channels
            p = "controlnet_cond_embedding"
            self.cond = [pk.conv(p + ".conv_in", cin_pad=8)]
            self.cond += [pk.conv(f"{p}.blocks.{i}") for i in range(2 * (len(ce) - 1))]
            self.cond.append(pk.conv(p + ".conv_out"))
        del self.pk

    def embed_cond(self, img):
        """ControlNetConditioningEmbedding: img [N,H,W,8(3 real)] -> [N,H/8,W/8,C0]; once per image (PL:660-662)."""
        h = ops.conv_gemm(img, self.cond[0], act=L.ACT_SILU)
        for i, pw in enumerate(self.cond[1:-1]):
            h = ops.conv_gemm(h, pw, stride=2 if i % 2 == 1 else 1, act=L.ACT_SILU)
        return ops.conv_gemm(h, self.cond[-1])

    def embed_latent(self, z, out=None):
        """VAEControlNetConditioningEmbedding tail: conv_vae_out IS conv_in (CL:36,41,595-598). z: [N,h,w,in_pad]"""
        return ops.conv_gemm(z, self.conv_in, out=out)

    def forward(self, x, tproj, ctx, conds: Sequence[torch.Tensor], out_scale: float = 1.0,
                out_scale_dev=None, level_scales: Optional[Sequence[float]] = None):
        """x: [N,H,W,in_pad]; conds: k pre-embedded [N,H,W,C0] tensors -> the pass runs at batch k*N with shared
        weights (tproj/ctx must already be k*N rows).  Returns (13 residual tensors [k*N,HW,C])."""
        k = len(conds)
        N, H, W, _ = x.shape
        c0 = self.conv_in.cout
        h0 = torch.empty((k * N, H, W, c0), dtype=x.dtype, device=x.device)
        for i, c in enumerate(conds):                       # sample = conv_in(sample) + cond   (CL:197-203)
            ops.conv_gemm(x, self.conv_in, residual=c, out=h0[i * N:(i + 1) * N])
        skips, h = self.run(h0, tproj, ctx)
        ls = list(level_scales) if level_scales is not None else [1.0] * (len(self.zero) + 1)   # guess_mode: CL:256-264
        res = [ops.conv_gemm(s, z, out_scale=out_scale * ls[i], out_scale_dev=out_scale_dev)
               for i, (s, z) in enumerate(zip(skips, self.zero))]
        res.append(ops.conv_gemm(h, self.zero_mid, out_scale=out_scale * ls[-1], out_scale_dev=out_scale_dev))
        return res


class Fusion:
    """13 ControlNetBlocks (MC:103-114, 160-169)."""

    def __init__(self, sd: SD, cfg: UNetConfig, dtype, device, sample_size: Optional[int] = None):
        self.table = cfg.residual_table(sample_size)
        self.params = []
        for i in range(len(self.table)):
            p = f"multi_controlnet_down_blocks.{i}" if i < len(self.table) - 1 else "multi_controlnet_mid_block"
            self.params.append(ops.pack_fusion_params(sd, p, dtype, device))

    def forward(self, res_per_net: Sequence[Sequence[torch.Tensor]], bs: Sequence[Sequence[int]], N: int,
                scales: Sequence[float], scales_dev=None, addends=None, first_level: int = 0):
        """res_per_net[i][k]: tensor view whose data_ptr is sample 0 of net i at level first_level + k.  addends: optional
        UNet skip / mid tensors of the same levels; the outputs are then already `skip + residual` (PL:500-510).
        The levels given (all 13 by default) go through one batched call: three launches."""
        nl = len(res_per_net[0])
        table = self.table[first_level:first_level + nl]
        blocks = [([res_per_net[i][k] for i in range(6)], [bs[i][k] for i in range(6)], self.params[first_level + k], s * s, c)
                  for k, (c, s) in enumerate(table)]
        outs = ops.fusion_blocks(blocks, N, scales, scales_dev, addends=addends, first_block=first_level)
        return [o.reshape(N, s, s, c) for o, (c, s) in zip(outs, table)]


class VAE:
    def __init__(self, sd: SD, cfg: VAEConfig, dtype, device, decoder: bool = True, encoder: bool = True):
        self.cfg, self.dtype, self.device = cfg, dtype, device
        pk = _Packer(sd, dtype, device)
        g, eps = cfg.norm_num_groups, cfg.norm_eps
        ch = cfg.block_out_channels
        n = len(ch)
        self.lat_pad = (cfg.latent_channels + 7) // 8 * 8

        def mid(side):
            a = f"{side}.mid_block.attentions.0"
            return dict(r0=Resnet(pk, f"{side}.mid_block.resnets.0", g, eps, None),
                        gn=pk.norm(a + ".group_norm"),
                        qkv=pk.cat([a + ".to_q", a + ".to_k", a + ".to_v"], bias=True),
                        o=pk.conv(a + ".to_out.0"),
                        r1=Resnet(pk, f"{side}.mid_block.resnets.1", g, eps, None))
        if encoder:
            self.e_in = pk.conv("encoder.conv_in", cin_pad=8)
            self.e_down = []
            for i in range(n):
                rs = [Resnet(pk, f"encoder.down_blocks.{i}.resnets.{j}", g, eps, None)
                      for j in range(cfg.layers_per_block)]
                ds = pk.conv(f"encoder.down_blocks.{i}.downsamplers.0.conv") if i != n - 1 else None
                self.e_down.append((rs, ds))
            self.e_mid = mid("encoder")
            self.e_norm = pk.norm("encoder.conv_norm_out")
            self.e_out = pk.conv("encoder.conv_out")
            self.quant = pk.conv("quant_conv")
        if decoder:
            self.post_quant = pk.conv("post_quant_conv", cin_pad=self.lat_pad, cout_pad=8)
            # decode(latents / scaling_factor) (PL:552-557) with the division folded into the 1x1 weights
            b = sd.get("post_quant_conv.bias")
            self.post_quant_scaled = ops.pack_weight(
                (pk.t("post_quant_conv.weight").double() / cfg.scaling_factor).float(), None if b is None else b.to(device), dtype,
                device, cin_pad=self.lat_pad, cout_pad=8)
            self.d_in = pk.conv("decoder.conv_in", cin_pad=8)
            self.d_mid = mid("decoder")
            self.d_up = []
            for i in range(n):
                rs = [Resnet(pk, f"decoder.up_blocks.{i}.resnets.{j}", g, eps, None)
                      for j in range(cfg.layers_per_block + 1)]
                us = pk.conv(f"decoder.up_blocks.{i}.upsamplers.0.conv") if i != n - 1 else None
                self.d_up.append((rs, us))
            self.d_norm = pk.norm("decoder.conv_norm_out")
            self.d_out = pk.conv("decoder.conv_out")

    def _mid(self, m, h):
        cfg = self.cfg
        h = m["r0"](h, None)
        N, H, W, C = h.shape
        n = ops.group_norm(h, m["gn"][0], m["gn"][1], cfg.norm_num_groups, cfg.norm_eps, False).reshape(N, H * W, C)
        qkv = ops.linear(n, m["qkv"])
        a = ops.attention(qkv[:, :, :C], qkv[:, :, C:2 * C], qkv[:, :, 2 * C:], 1)
        h = ops.linear(a, m["o"], residual=h.reshape(N, H * W, C)).reshape(N, H, W, C)
        return m["r1"](h, None)

    def encode_moments(self, img):
        """img: [N,H,W,8 (3 real)] -> moments [N,H/8,W/8,2L]"""
        cfg = self.cfg
        h = ops.conv_gemm(img, self.e_in)
        for rs, ds in self.e_down:
            for r in rs:
                h = r(h, None)
            if ds is not None:     # F.pad(0,1,0,1) + conv s2 p0
                h = ops.conv_gemm(h, ds, stride=2, pad=0, out_hw=(h.shape[1] // 2, h.shape[2] // 2))
        h = self._mid(self.e_mid, h)
        h = ops.group_norm(h, self.e_norm[0], self.e_norm[1], cfg.norm_num_groups, cfg.norm_eps, True)
        h = ops.conv_gemm(h, self.e_out)
        return ops.conv_gemm(h, self.quant)

    def decode(self, z, unscaled_latents: bool = False):
        """z: [N,h,w,lat_pad] -> image [N,8h,8w,3].  unscaled_latents: z is the scheduler's latents and the
        1/scaling_factor of PL:553-554 is applied by the (pre-scaled) post_quant weights."""
        cfg = self.cfg
        h = ops.conv_gemm(z, self.post_quant_scaled if unscaled_latents else self.post_quant)
        h = ops.conv_gemm(h, self.d_in)
        h = self._mid(self.d_mid, h)
        for rs, us in self.d_up:
            for r in rs:
                h = r(h, None)
            if us is not None:
                h = ops.conv_gemm(h, us, upsample=True)
        h = ops.group_norm(h, self.d_norm[0], self.d_norm[1], cfg.norm_num_groups, cfg.norm_eps, True)
        return ops.conv_gemm(h, self.d_out)


class GroupedEncoder:
    """Several encoders of identical architecture but different weights (the 3 batched ControlNet passes + the UNet's
    own conv_in/down/mid path of one denoising step) executed in LOCKSTEP as grouped launches over the
    batch-concatenated activations [sum(n_g), H, W, C]: every conv/linear/GroupNorm/LayerNorm is ONE launch whose
    M tiles / samples / rows select their group's weights, attention is one launch over the whole batch.  At batch 1
    a single net's kernels (M = 2..6 samples) cannot fill 256 CUs; the grouped launch has 14 samples."""

    def __init__(self, encoders: Sequence[Encoder], counts: Sequence[int]):
        self.encs, self.counts = list(encoders), list(counts)
        e0 = self.encs[0]
        self.cfg = e0.cfg
        for e in self.encs[1:]:
            if e.cfg.block_out_channels != e0.cfg.block_out_channels or e.cfg.layers_per_block != e0.cfg.layers_per_block:
                raise L.EdgeStyleHipError("grouped encoders must share the architecture")
        self.ntot = sum(self.counts)
        self.width = max(e.tproj_width for e in self.encs)

    def groupable(self, hw_min: int) -> bool:
        return len(self.encs) <= 4 and all((n * hw_min) % ops.BM == 0 for n in self.counts)

    def time_proj(self, t_rows: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """[ntot, width]: each group's ResnetBlock time projections (offsets of down/mid blocks coincide).  `out`: a
        zero-initialised buffer to fill (columns beyond a group's width are never written)."""
        if out is None:
            out = torch.zeros((self.ntot, self.width), dtype=self.encs[0].dtype, device=t_rows.device)
        es = out.element_size()
        a = 0
        for e, n in zip(self.encs, self.counts):
            proj = e.time_proj(t_rows[:n])
            ops.memcpy2d(out[a].data_ptr(), self.width * es, proj.data_ptr(), e.tproj_width * es, e.tproj_width * es, n)
            a += n
        return out

    def _resnet(self, rs, x, tproj):
        c = self.counts
        r0 = rs[0]
        h = ops.group_norm(x, [r.n1[0] for r in rs], [r.n1[1] for r in rs], r0.groups, r0.eps, True, group_n=c)
        h = ops.conv_gemm(h, [r.conv1 for r in rs], temb=tproj[:, r0.temb_off:], group_n=c, gn_groups=r0.groups)
        h = ops.group_norm(h, [r.n2[0] for r in rs], [r.n2[1] for r in rs], r0.groups, r0.eps, True, group_n=c)
        if all(r.conv2s is not None for r in rs):
            return ops.conv_gemm(h, [r.conv2s for r in rs], tail=(x,), group_n=c, gn_groups=r0.groups)
        xs = ops.conv_gemm(x, [r.short for r in rs], group_n=c) if r0.short is not None else x
        return ops.conv_gemm(h, [r.conv2 for r in rs], residual=xs, group_n=c, wide=True, gn_groups=r0.groups)

    def _transformer(self, ts, x, kv):
        c = self.counts
        t0 = ts[0]
        N, H, W, C = x.shape
        rows = [n * H * W for n in c]
        tok = ops.gn_proj_in(x, [t.norm[0] for t in ts], [t.norm[1] for t in ts], t0.groups, 1e-6, [t.proj_in for t in ts], group_n=c).reshape(N, H * W, C)
        fold = all(t.ln_fold for t in ts)

        def ln_linear(k, plain, folded):
            if fold and folded is not None:
                return ops.linear(tok, [getattr(t, folded) for t in ts], group_n=rows)
            n = ops.layer_norm(tok, [getattr(t, k)[0] for t in ts], [getattr(t, k)[1] for t in ts], group_rows=rows)
            return ops.linear(n, [getattr(t, plain) for t in ts], group_n=rows)
        qkv = ln_linear("ln1", "qkv", "qkv_ln")
        a = ops.attention(qkv[:, :, :C], qkv[:, :, C:2 * C], qkv[:, :, 2 * C:], t0.heads)
        tok = ops.linear(a, [t.o1 for t in ts], residual=tok, group_n=rows, wide=True)
        q = ln_linear("ln2", "q2", "q2_ln")
        a = ops.attention(q, kv[:, :, :C], kv[:, :, C:], t0.heads)
        tok = ops.linear(a, [t.o2 for t in ts], residual=tok, group_n=rows, wide=True)
        f = ln_linear("ln3", "ff1", "ff1_ln" if all(t.ln_fold and t.ff1_ln is not None for t in ts) else None)
        if all(t.ffo is not None for t in ts):
            return ops.conv_gemm(f.reshape(N, H, W, 4 * C), [t.ffo for t in ts], x2=tok.reshape(N, H, W, C), residual=x, group_n=c, wide=True,
                                 gn_groups=t0.groups)
        tok = ops.linear(f, [t.ff2 for t in ts], residual=tok, group_n=rows, wide=True)
        return ops.conv_gemm(tok.reshape(N, H, W, C), [t.proj_out for t in ts], residual=x, group_n=c, wide=True, gn_groups=t0.groups)

    def run(self, h, tproj, ctx: List[torch.Tensor]):
        """h: [ntot,H,W,C0] (each group's conv_in(sample)+cond already applied) -> (skips, mid) over the whole batch."""
        skips = [h]
        ci = 0
        e0 = self.encs[0]
        for i, blk in enumerate(e0.down):
            for j, (r, a) in enumerate(blk):
                h = self._resnet([e.down[i][j][0] for e in self.encs], h, tproj)
                if a is not None:
                    h = self._transformer([e.down[i][j][1] for e in self.encs], h, ctx[ci]); ci += 1
                skips.append(h)
            if e0.downsample[i] is not None:
                h = ops.conv_gemm(h, [e.downsample[i] for e in self.encs], stride=2, group_n=self.counts, gn_groups=e0.cfg.norm_num_groups)
                skips.append(h)
        h = self._resnet([e.mid0 for e in self.encs], h, tproj)
        h = self._transformer([e.mid_attn for e in self.encs], h, ctx[ci])
        h = self._resnet([e.mid1 for e in self.encs], h, tproj)
        return skips, h
