"""CPU suite: host logic of the product package (no GPU, no compute calls): C-ABI surface, weight packing,
on-disk layout round trips, reference error behaviour, scheduler tables, sharding."""
import ctypes
import os
import re

import pytest
import torch

from edgestyle_amd import config as C, weights as W, ops, lib
from edgestyle_amd.models import (UNet2DConditionModel, ControlNetModel, ControlLoRAModel, FusedControlLoRAModel,
                                  AutoencoderKL, EdgeStyleMultiControlNetModel, unet_config_from_json)
from edgestyle_amd.schedulers import DDIMScheduler
from edgestyle_amd.dist import shard_range, shard_seed
from tests.helpers import make_weights

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cabi_library_loads_and_exports_every_declared_symbol():
    assert os.path.exists(lib.LIB_PATH), "build with __graft_entry__.build()"
    header = open(os.path.join(ROOT, "include", "edgestyle_hip.h")).read()
    declared = set(re.findall(r"\b(es_[a-z0-9_]+)\s*\(", header))
    declared = {d for d in declared if not d.endswith("_desc") or d == "es_sizeof_desc"}
    declared -= {"es_ctx_geometry"}
    assert declared == set(lib.SYMBOLS), declared ^ set(lib.SYMBOLS)
    h = ctypes.CDLL(lib.LIB_PATH)
    for name in declared:
        assert getattr(h, name) is not None
    L = lib.load()
    assert L.es_abi_version() == 7
    # struct layouts agree with the C side (sizes are what the kernels index with)
    for i, st in enumerate((lib.GemmDesc, lib.AttnDesc, lib.GnDesc, lib.FusionDesc, lib.LnDesc, lib.XsDesc)):
        assert L.es_sizeof_desc(i) == ctypes.sizeof(st)


def test_plan_records_nothing_and_launches_nothing_without_a_gpu():
    """es_plan object life cycle on the host (no compute calls): create / begin / end / size / destroy; a second plan
    cannot start recording while one records."""
    L = lib.load()
    a, b = ctypes.c_void_p(L.es_plan_create()), ctypes.c_void_p(L.es_plan_create())
    assert L.es_plan_size(a) == 0
    assert L.es_plan_begin_record(a) == 0
    assert L.es_plan_begin_record(b) != 0 and b"recording" in L.es_last_error()
    assert L.es_plan_end_record(b) != 0
    assert L.es_plan_end_record(a) == 0 and L.es_plan_size(a) == 0
    ctx = ctypes.c_void_p()
    assert L.es_ctx_create(0, ctypes.byref(ctx)) == 0
    assert L.es_ctx_plan_size(ctx, lib.PLAN_STEP) == -1
    assert L.es_ctx_set_plan(ctx, lib.PLAN_STEP, a) == 0 and L.es_ctx_plan_size(ctx, lib.PLAN_STEP) == 0
    geo = lib.CtxGeometry(B=1, cfg=1, h=8, w=8, latent_channels=4, latent_pad=8, n_conds=7, n_steps=4, dtype=0)
    assert L.es_ctx_set_geometry(ctx, ctypes.byref(geo)) != 0            # 7 conditions: refused
    L.es_ctx_destroy(ctx)                                               # destroys plan a
    L.es_plan_destroy(b)


def test_plan_image_round_trip_markers_and_context_image_errors(tmp_path):
    """Host-only parts of the context-image machinery: es_plan_import / es_plan_export round trip of a hand-made launch
    list, refusal of truncated or inconsistent images, and es_ctx_load's errors on a missing / foreign file (nothing touches
    a GPU: the file is rejected before any HIP call)."""
    import struct
    L = lib.load()
    # a hand-made image of two recorded calls (csrc/plan.h: 18 = es_memcpy {dst, src, bytes}, 16 = es_incr {ctr}); nothing
    # can RECORD on a CPU-only host (every entry point validates and launches), but import / export are host-only
    blob = struct.pack("<QQQ", 0x1000, 0x2000, 64) + b"\0" * 8 + struct.pack("<Q", 0x3000)
    img = struct.pack("<QQ", 2, len(blob)) + struct.pack("<qQQ", 18, 0, 24) + struct.pack("<qQQ", 16, 32, 8) + blob
    buf = (ctypes.c_char * len(img)).from_buffer_copy(img)
    q = ctypes.c_void_p(L.es_plan_import(buf, len(img)))
    assert q.value and L.es_plan_size(q) == 2 and L.es_plan_count(q, 18) == 1 and L.es_plan_count(q, 16) == 1
    n = L.es_plan_export(q, None, 0)
    assert n == len(img)
    out = (ctypes.c_char * n)()
    assert L.es_plan_export(q, out, n) == n and bytes(out) == img
    assert not L.es_plan_import(buf, len(img) - 8) and b"truncated" in L.es_last_error()
    bad_op = struct.pack("<QQ", 1, 8) + struct.pack("<qQQ", 16, 4, 8) + b"\0" * 8          # record runs past the blob
    assert not L.es_plan_import((ctypes.c_char * len(bad_op)).from_buffer_copy(bad_op), len(bad_op))
    L.es_plan_destroy(q)
    ctx = ctypes.c_void_p()
    assert L.es_ctx_load(str(tmp_path / "missing.esctx").encode(), 0, ctypes.byref(ctx)) != 0 and b"open" in L.es_last_error()
    bad = tmp_path / "bad.esctx"
    bad.write_bytes(b"not a context image at all, just some bytes" * 4)
    assert L.es_ctx_load(str(bad).encode(), 0, ctypes.byref(ctx)) != 0 and b"not a context image" in L.es_last_error()


def test_native_ddim_coefficients_match_the_host_scheduler():
    """es_ddim_coef_table (what es_denoise_loop derives from `timesteps`, host-only code) vs DDIMScheduler.coef_table():
    bit for bit with the scheduler's alphas_cumprod handed over, within 1e-6 with the library's own SD1.5 schedule."""
    import numpy as np
    L = lib.load()
    fp = ctypes.POINTER(ctypes.c_float)
    for T in (4, 20, 50):
        s = DDIMScheduler()
        ts = s.set_timesteps(T).float().numpy().astype(np.float32)
        want = s.coef_table().numpy()
        ac = s.alphas_cumprod.numpy().astype(np.float32)
        out = np.zeros((T, 4), dtype=np.float32)
        assert L.es_ddim_coef_table(ac.ctypes.data_as(fp), len(ac), ts.ctypes.data_as(fp), T, out.ctypes.data_as(fp)) == 0
        assert np.array_equal(out, want)
        assert L.es_ddim_coef_table(None, 0, ts.ctypes.data_as(fp), T, out.ctypes.data_as(fp)) == 0
        assert np.abs(out - want).max() < 1e-6


def test_native_unipc_coefficients_match_the_host_scheduler_bitwise():
    """es_unipc_coef_table (what es_denoise_loop derives from `timesteps` under ES_SCHED_UNIPC; csrc/plan.hip) ==
    UniPCMultistepScheduler.coef_table(), with the library's own double-precision SD1.5 schedule and with the scheduler's."""
    import numpy as np
    from edgestyle_amd.schedulers import UniPCMultistepScheduler
    L = lib.load()
    fp = ctypes.POINTER(ctypes.c_float)
    for T in (2, 3, 4, 20, 50):
        s = UniPCMultistepScheduler()
        ts = s.set_timesteps(T).float().numpy().astype(np.float32)
        want = s.coef_table().numpy()
        out = np.zeros((T, 12), dtype=np.float32)
        assert L.es_unipc_coef_table(None, 0, ts.ctypes.data_as(fp), T, out.ctypes.data_as(fp)) == 0
        assert np.array_equal(out, want)
        ac = s.alphas_cumprod.numpy().astype(np.float64)
        assert L.es_unipc_coef_table(ac.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), len(ac), ts.ctypes.data_as(fp), T, out.ctypes.data_as(fp)) == 0
        assert np.array_equal(out, want)


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(lib, "_lib", None)
    monkeypatch.setattr(lib, "LIB_PATH", "/nonexistent/libedgestyle_hip.so")
    with pytest.raises(lib.EdgeStyleHipError):
        lib.load()


def test_cpu_device_is_refused_not_emulated():
    ucfg = C.tiny_unet()
    unet = UNet2DConditionModel(W.random_state_dict(W.unet_shapes(ucfg)), ucfg)
    with pytest.raises(lib.EdgeStyleHipError):
        unet(torch.zeros(1, 4, 16, 16), 1, torch.zeros(1, 77, ucfg.cross_attention_dim))


def test_pack_weight_layout():
    g = torch.Generator().manual_seed(0)
    w = torch.randn(24, 16, 3, 3, generator=g)
    b = torch.randn(24, generator=g)
    pw = ops.pack_weight(w, b, torch.float32, "cpu")
    assert pw.w.shape == (128, 192) and pw.bn == 128 and pw.cout == 24 and pw.cin == 16 and pw.ksize == 3
    # K index = (ky*3+kx)*Cin + c
    assert float(pw.w[5, (1 * 3 + 2) * 16 + 7]) == float(w[5, 7, 1, 2])
    assert float(pw.w[24:].abs().max()) == 0 and float(pw.w[:, 144:].abs().max()) == 0 and torch.equal(pw.bias[:24], b)
    assert ops.choose_bn(320) == 160 and ops.choose_bn(640) == 128 and ops.choose_bn(960) == 160 and ops.choose_bn(4) == 128
    # Cin 4 -> 8 zero padding (conv_in), Cout 4 -> 8 zero rows (post_quant)
    p2 = ops.pack_weight(torch.ones(4, 4, 1, 1), torch.ones(4), torch.float32, "cpu", cin_pad=8, cout_pad=8)
    assert p2.cin == 8 and p2.cout == 8 and float(p2.w[:4, :4].sum()) == 16 and float(p2.w.sum()) == 16
    # GEGLU: packed rows in blocks of 32 = [16 hidden | 16 gate]
    inner = 64
    wg = torch.arange(2 * inner, dtype=torch.float32)[:, None].repeat(1, 8)
    pg = ops.pack_weight(wg, torch.arange(2 * inner, dtype=torch.float32), torch.float32, "cpu", geglu=True)
    rows = pg.w[: 2 * inner, 0].tolist()
    assert rows[:16] == list(range(0, 16)) and rows[16:32] == list(range(inner, inner + 16))
    assert rows[32:48] == list(range(16, 32)) and rows[48:64] == list(range(inner + 16, inner + 32))
    assert pg.bias[:32].tolist() == rows[:32]


def test_splitk_heuristic_bounds():
    for M, rows, bn, kpad in [(128, 1280, 128, 11520), (8192, 320, 160, 2880), (65536, 1280, 128, 11520), (2, 1280, 128, 320)]:
        s = ops.choose_splitk(M, rows, bn, kpad)
        assert 1 <= s <= kpad // 64


def test_plan_gemm_returns_legal_launches():
    """Every (M, Cout, K) of the SD1.5 path gets a launch the C ABI accepts: bn divides rows_padded, split-K only where
    it is allowed, 4-stage rings only with at most one workgroup per CU, the 64x64 tile only for small launches."""
    for M in (2, 128, 512, 896, 2048, 3584, 8192, 14336, 57344, 458752):
        for rows in (320, 640, 960, 1280, 1920, 2560, 3840):
            for kpad in (320, 640, 1280, 2880, 5760, 11520, 23040):
                for bns in ((160, 128, 64), (320, 160, 128), (128,)):
                    if not any(rows % b == 0 for b in bns):
                        continue
                    bn, sk, st = ops.plan_gemm(M, rows, kpad, bns=bns)
                    assert rows % bn == 0 and bn in bns
                    assert 1 <= sk <= max(1, kpad // 64) and st in (2, 4)
                    if bn == 64:
                        assert M <= ops.PLAN_SMALL_MAX_M or len(bns) == 1
                    if bn == 320:
                        assert st == 2 and M >= ops.PLAN_BIG_MIN_M
                    assert ops.plan_gemm(M, rows, kpad, bns=bns, allow_split=False)[1] == 1


def test_layer_norm_fold_algebra():
    """pack_weight_ln: Linear(LayerNorm(x)) == rstd (x (W gamma)^T - mean colsum) + (W beta + b), with the column sums
    taken from the rounded packed weights (what the kernel's epilogue computes)."""
    g = torch.Generator().manual_seed(5)
    M, C, Cout = 37, 128, 192
    x = torch.randn(M, C, generator=g) * 2 + 1
    gamma, beta = 1 + 0.2 * torch.randn(C, generator=g), 0.1 * torch.randn(C, generator=g)
    w, b = torch.randn(Cout, C, generator=g) / 11, torch.randn(Cout, generator=g) * 0.1
    pw = ops.pack_weight_ln(w, b, gamma, beta, 1e-5, torch.float32, "cpu")
    assert pw.ln_colsum is not None and pw.ln_colsum.shape[0] == pw.rows_padded
    mean = x.mean(1, keepdim=True)
    rstd = (x.var(1, unbiased=False, keepdim=True) + 1e-5).rsqrt()
    y = rstd * (x @ pw.w[:Cout, :C].T - mean * pw.ln_colsum[None, :Cout]) + pw.bias[None, :Cout]
    ref = torch.nn.functional.linear(torch.nn.functional.layer_norm(x, (C,), gamma, beta, 1e-5), w, b)
    assert float((y - ref).abs().max()) < 1e-4


def test_tail_and_ffo_fold_algebra():
    """The two weight folds of this path restated on the CPU from the packed tensors themselves.
    (a) ops.pack_weight_tail: conv3x3(h) + conv1x1(t) == one GEMM over K = (ky,kx,c) taps of h followed by t's channels at
        the output pixel (ResnetBlock2D conv2 + conv_shortcut);
    (b) engine.Transformer.ffo: proj_out(ff2(f) + tok) + x == [Wp Wf | Wp] [f | tok] + (Wp bf + bp) + x."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(9)
    N, C, Ct, Cout, H = 2, 64, 128, 96, 5
    h, t = torch.randn(N, C, H, H, generator=g), torch.randn(N, Ct, H, H, generator=g)
    w3, w1 = torch.randn(Cout, C, 3, 3, generator=g) / 24, torch.randn(Cout, Ct, 1, 1, generator=g) / 11
    b = torch.randn(Cout, generator=g)
    pw = ops.pack_weight_tail(w3, w1, b, torch.float32, "cpu")
    assert pw.ctail == Ct and pw.cin == C and pw.ksize == 3 and pw.kpad == 9 * C + Ct
    cols = F.unfold(h, 3, padding=1)                                  # [N, C*9, HW], index c*9 + tap
    cols = cols.view(N, C, 9, H * H).permute(0, 3, 2, 1).reshape(N, H * H, 9 * C)      # tap-major (ky,kx,c), like the kernel's K
    a = torch.cat([cols, t.permute(0, 2, 3, 1).reshape(N, H * H, Ct)], 2)
    y = (a @ pw.w[:Cout].T + pw.bias[:Cout]).permute(0, 2, 1).reshape(N, Cout, H, H)
    ref = F.conv2d(h, w3, None, padding=1) + F.conv2d(t, w1, None) + b[None, :, None, None]
    assert float((y - ref).abs().max()) < 1e-4
    with pytest.raises(Exception):
        ops.pack_weight_tail(w3[:, :40], w1, b, torch.float32, "cpu")      # 40 input channels: not a multiple of 64

    Cm, M = 64, 11
    f, tok, x = torch.randn(M, 4 * Cm, generator=g), torch.randn(M, Cm, generator=g), torch.randn(M, Cm, generator=g)
    wf, bf = torch.randn(Cm, 4 * Cm, generator=g) / 16, torch.randn(Cm, generator=g) * 0.1
    wp, bp = torch.randn(Cm, Cm, generator=g) / 8, torch.randn(Cm, generator=g) * 0.1
    ref = F.linear(F.linear(f, wf, bf) + tok, wp, bp) + x
    wc = torch.cat([wp.double() @ wf.double(), wp.double()], 1).float()
    bc = (wp.double() @ bf.double() + bp.double()).float()
    pwc = ops.pack_weight(wc, bc, torch.float32, "cpu")
    y = torch.cat([f, tok], 1) @ pwc.w[:Cm, : 5 * Cm].T + pwc.bias[:Cm] + x
    assert float((y - ref).abs().max()) < 1e-4


def test_controllora_state_dict_is_lora_plus_zero_convs_only():
    ucfg = C.tiny_unet()
    ws = make_weights(ucfg, C.tiny_vae())
    unet = UNet2DConditionModel(ws["unet"], ucfg)
    net = ControlLoRAModel(ws["lora0"], ucfg, lora_linear_rank=4, uses_vae=True)
    with pytest.raises(lib.EdgeStyleHipError):
        net.full_state_dict()                                   # tie_weights first (TT:259-261)
    net.tie_weights(unet)
    full = net.full_state_dict()
    assert full["down_blocks.1.resnets.0.conv1.weight"] is ws["unet"]["down_blocks.1.resnets.0.conv1.weight"]
    saved = net.state_dict()
    assert set(saved) == set(W.controllora_saved_shapes(ucfg, 4))          # CL:600-606
    assert all(k.split(".")[0] not in W.SKIP_LAYERS or ".lora_layer." in k for k in saved)
    fused = net.fuse()
    assert isinstance(fused, FusedControlLoRAModel) and not any(".lora_layer." in k for k in fused.state_dict())
    k = "mid_block.attentions.0.transformer_blocks.0.ff.net.2"
    want = ws["unet"][k + ".weight"] + ws["lora0"][k + ".lora_layer.up.weight"] @ ws["lora0"][k + ".lora_layer.down.weight"]
    assert torch.allclose(fused.state_dict()[k + ".weight"], want, atol=1e-6)
    assert torch.equal(ws["unet"][k + ".weight"], unet.state_dict()[k + ".weight"])   # tied UNet tensor untouched


def test_multicontrolnet_directory_layout_round_trip(tmp_path):
    """save_pretrained / from_pretrained (MC:213-282, MC:289-430) incl. load_pattern de-duplication and errors"""
    ucfg, vcfg = C.tiny_unet(), C.tiny_vae()
    ws = make_weights(ucfg, vcfg)
    unet = UNet2DConditionModel(ws["unet"], ucfg)
    vae = AutoencoderKL(ws["vae"], vcfg)
    pose = ControlNetModel(ws["openpose"], ucfg)
    l0 = ControlLoRAModel(ws["lora0"], ucfg, lora_linear_rank=4, uses_vae=True)
    l1 = ControlLoRAModel(ws["lora1"], ucfg, lora_linear_rank=4, uses_vae=True)
    mc = EdgeStyleMultiControlNetModel([l0, pose, l1, pose, l1, pose])
    mc.load_state_dict(ws["fusion"])
    d = str(tmp_path / "controlnet")
    mc.save_pretrained(d, save_pattern=C.CONTROLNET_PATTERN)
    assert sorted(os.listdir(d)) == ["controlnet_0", "controlnet_1", "diffusion_pytorch_model.safetensors"]
    with pytest.raises(ValueError):
        EdgeStyleMultiControlNetModel.from_pretrained(d, controlnet_class=ControlLoRAModel)              # no load_pattern
    with pytest.raises(ValueError):
        EdgeStyleMultiControlNetModel.from_pretrained(d, controlnet_class=ControlLoRAModel,
                                                      load_pattern=C.CONTROLNET_PATTERN,
                                                      static_controlnets=[None, pose, None, pose, None, pose])  # no vae
    with pytest.raises(ValueError):
        EdgeStyleMultiControlNetModel.from_pretrained(d, controlnet_class=ControlLoRAModel, vae=vae,
                                                      load_pattern=C.CONTROLNET_PATTERN)                 # nets missing
    with pytest.raises(ValueError):
        EdgeStyleMultiControlNetModel.from_pretrained(str(tmp_path / "nope"), load_pattern=[0])
    m2 = EdgeStyleMultiControlNetModel.from_pretrained(d, vae=vae, controlnet_class=ControlLoRAModel,
                                                       load_pattern=C.CONTROLNET_PATTERN,
                                                       static_controlnets=[None, pose, None, pose, None, pose])
    assert m2.nets[2] is m2.nets[4] and m2.nets[1] is pose and m2.nets[0] is not m2.nets[2]    # MC:379-398
    assert [len(p) for _, p in m2.groups()] == [1, 3, 2]
    for k, v in ws["fusion"].items():
        assert torch.equal(m2.state_dict()[k], v)
    for k, v in ws["lora1"].items():
        assert torch.equal(m2.nets[2].state_dict()[k], v)
    assert m2.nets[0].config.uses_vae and m2.nets[0].config.lora_linear_rank == 4
    bad = dict(ws["fusion"])
    bad.pop("multi_controlnet_mid_block.third_conv.bias")
    with pytest.raises(RuntimeError):
        m2.load_state_dict(bad)
    # unet / vae round trip + config mapping from a diffusers-style config.json
    unet.save_pretrained(str(tmp_path / "sd" / "unet"))
    u2 = UNet2DConditionModel.from_pretrained(str(tmp_path / "sd"), subfolder="unet", torch_dtype=torch.float16)
    assert u2.cfg == ucfg and torch.equal(u2.state_dict()["conv_in.weight"], ws["unet"]["conv_in.weight"])
    cfg = unet_config_from_json({"block_out_channels": [320, 640, 1280, 1280], "attention_head_dim": 8,
                                 "cross_attention_dim": 768, "down_block_types": ["CrossAttnDownBlock2D"] * 3 + ["DownBlock2D"]})
    assert cfg == C.sd15_unet()


def test_ddim_scheduler_tables_match_oracle():
    from oracle import sd15_oracle as O
    s, o = DDIMScheduler(), O.DDIM()
    assert s.set_timesteps(50).tolist() == o.set_timesteps(50).tolist()
    tab = s.coef_table()
    assert tab.shape == (50, 4)
    x, e = torch.randn(1, 4, 8, 8), torch.randn(1, 4, 8, 8)
    for i in (0, 17, 49):
        c = tab[i]
        mine = c[2] * (x - c[1] * e) / c[0] + c[3] * e
        assert torch.allclose(mine, o.step(e, int(s.timesteps[i]), x), atol=1e-5)
    with pytest.raises(ValueError):
        s.set_timesteps(2000)


def test_shard_ranges_cover_and_seeds_are_world_size_independent():
    for n, w in [(64, 8), (64, 1), (10, 4), (3, 8), (0, 2)]:
        spans = [shard_range(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1
    assert [shard_seed(42, r, 8) for r in range(8)] == [42 + 8 * r for r in range(8)]
    with pytest.raises(ValueError):
        shard_range(4, 4, 4)


def test_unipc_coefficient_table_reproduces_oracle_trajectory():
    """The product scheduler only exports linear-recombination coefficients (the arithmetic is es_cfg_unipc_step);
    applying them in float64 must reproduce the oracle's UniPC restatement step for step."""
    from oracle import sd15_oracle as O
    from edgestyle_amd.schedulers import UniPCMultistepScheduler
    for n in (50, 5, 2):
        s = UniPCMultistepScheduler.from_config({"steps_offset": 1, "timestep_spacing": "leading", "foo": 1})
        ts = s.set_timesteps(n)
        o = O.UniPC()
        assert o.set_timesteps(n).tolist() == ts.tolist()
        tab = s.coef_table().double()
        assert tab.shape == (n, 12) and float(tab[0, 2]) == 0.0 and float(tab[1, 2]) == 1.0
        g = torch.Generator().manual_seed(n)
        xo = torch.randn(1, 4, 8, 8, generator=g)
        xm, last, m0, m1 = xo.double(), torch.zeros(1, 4, 8, 8).double(), torch.zeros(1, 4, 8, 8).double(), torch.zeros(1, 4, 8, 8).double()
        for i, t in enumerate(ts.tolist()):
            eps = torch.randn(1, 4, 8, 8, generator=g)
            xo = o.step(eps, t, xo)
            c = tab[i]
            x0 = (xm - c[1] * eps.double()) / c[0]
            xc = c[3] * last + c[4] * m0 + c[5] * m1 + c[6] * x0 if c[2] != 0 else xm
            xm, m1, m0, last = c[7] * xc + c[8] * x0 + c[9] * m0, m0, x0, xc
            assert float((xm.float() - xo).abs().max() / xo.abs().max()) < 1e-5
    # exactness property of the solver itself: with the true noise as model output it stays on the marginal
    o = O.UniPC()
    ts = o.set_timesteps(20)
    x0t, nz = torch.randn(1, 4, 8, 8), torch.randn(1, 4, 8, 8)
    a, sg = o._alpha_sigma(o.sigmas[0])
    x = (a * x0t + sg * nz).float()
    for t in ts.tolist():
        x = o.step(nz, t, x)
    a, sg = o._alpha_sigma(o.sigmas[-1])
    assert float((x - (a * x0t + sg * nz).float()).abs().max()) < 1e-4


# ---------------------------------------------------------------------------------------------------------------
# serving loop (edgestyle_amd/serve.py): host logic with a stand-in pipeline
class _FakePipe:
    """Per-sample deterministic 'pipeline': image_i depends only on request i's own inputs."""

    def __init__(self, fail_on_steps=None, delay=0.0):
        self.calls = []
        self.fail_on_steps, self.delay = fail_on_steps, delay

    def __call__(self, prompt_embeds, negative_prompt_embeds, image, latents, guidance_scale, num_inference_steps,
                 control_guidance_start, control_guidance_end, output_type):
        import time
        import types
        assert output_type == "pt" and len(image) == 6
        B = latents.shape[0]
        assert all(t.shape[0] == B for t in (prompt_embeds, negative_prompt_embeds, *image))
        self.calls.append((B, num_inference_steps, guidance_scale))
        if self.fail_on_steps == num_inference_steps:
            raise RuntimeError("boom")
        time.sleep(self.delay)
        per = latents.mean(dim=(1, 2, 3)) + prompt_embeds.mean(dim=(1, 2)) - negative_prompt_embeds.mean(dim=(1, 2)) \
            + sum(im.mean(dim=(1, 2, 3)) * (k + 1) for k, im in enumerate(image)) + guidance_scale
        return types.SimpleNamespace(images=per[:, None, None, None].expand(B, 3, 8, 8).clone())


def _request(seed, steps=50, gs=7.5, hw=16):
    from edgestyle_amd.serve import TryOnRequest
    g = torch.Generator().manual_seed(1000 + seed)
    return TryOnRequest([torch.randn(1, 3, hw, hw, generator=g) for _ in range(6)], torch.randn(1, 77, 32, generator=g),
                        torch.randn(1, 77, 32, generator=g), gs, steps, seed)


def test_service_batches_compatible_requests_and_results_do_not_depend_on_the_batch():
    from edgestyle_amd.serve import TryOnService
    solo_pipe = _FakePipe()
    solo = TryOnService(solo_pipe, max_batch=1, max_wait_s=0.0)
    want = [solo.submit(_request(s)).result(timeout=10) for s in range(5)]
    solo.shutdown()
    assert [c[0] for c in solo_pipe.calls] == [1] * 5

    pipe = _FakePipe(delay=0.05)
    svc = TryOnService(pipe, max_batch=8, max_wait_s=0.3, batch_sizes=(1, 2, 4, 8))
    futs = [svc.submit(_request(s)) for s in range(5)]
    got = [f.result(timeout=10) for f in futs]
    svc.shutdown()
    for a, b in zip(got, want):
        assert a.shape == (1, 3, 8, 8) and torch.allclose(a, b, atol=1e-6)
    # five compatible requests: captured batch sizes only -> 4 + 1, oldest first
    assert sorted(c[0] for c in pipe.calls) == [1, 4] and pipe.calls[0][0] == 4
    assert svc.stats["images"] == 5 and svc.stats["calls"] == 2


def test_service_keeps_incompatible_requests_apart_and_survives_a_failing_batch():
    from edgestyle_amd.serve import TryOnService
    pipe = _FakePipe(fail_on_steps=7)
    svc = TryOnService(pipe, max_batch=4, max_wait_s=0.2)
    a = [svc.submit(_request(s, steps=50)) for s in range(2)]
    b = [svc.submit(_request(s, steps=7)) for s in range(2)]          # this batch raises inside the pipeline
    c = [svc.submit(_request(9, steps=50, gs=3.0))]
    for f in a + c:
        assert f.result(timeout=10).shape == (1, 3, 8, 8)
    for f in b:
        with pytest.raises(RuntimeError, match="boom"):
            f.result(timeout=10)
    svc.shutdown()
    assert (2, 50, 7.5) in pipe.calls and (2, 7, 7.5) in pipe.calls and (1, 50, 3.0) in pipe.calls
    with pytest.raises(RuntimeError):
        svc.submit(_request(1))
    with pytest.raises(ValueError):
        TryOnService(pipe, batch_sizes=(2, 4))


def test_service_latents_depend_on_the_request_seed_only():
    from edgestyle_amd.serve import latents_for
    a, b = latents_for(42, 4, 8, 8), latents_for(42, 4, 8, 8)
    assert torch.equal(a, b) and not torch.equal(a, latents_for(43, 4, 8, 8)) and a.shape == (1, 4, 8, 8)


def test_best_embeddings_prompt_picker():
    """BestEmbeddings (model/utils.py:647-684): top-2 colours + top-2 items by CLIP image-text probability, joined into
    "edgestyle, c1, c2, i1, i2"; here with a stand-in model whose logits are known."""
    from types import SimpleNamespace
    from edgestyle_amd.prompts import BestEmbeddings, DEFAULT_COLORS

    class Proc:
        def __call__(self, text, images, return_tensors, padding):
            return {"text": text, "n_images": len(images)}

    class Model:
        device = None

        def __call__(self, text, n_images):
            # image i prefers entries i, i+1, ... (descending logits from index i, cyclic)
            n = len(text)
            logits = torch.stack([torch.roll(torch.arange(n, 0, -1).float(), i) for i in range(n_images)])
            return SimpleNamespace(logits_per_image=logits)

    colors = ["red", "green", "blue", "black"]
    items = ["dress", "shirt", "coat"]
    be = BestEmbeddings(Model(), Proc(), colors=colors, clothing_items=items)
    assert be([object(), object()]) == ["edgestyle, red, green, dress, shirt", "edgestyle, green, blue, shirt, coat"]
    assert BestEmbeddings(Model(), Proc()).colors == DEFAULT_COLORS
    assert be.find_best(colors, [object()], n=3) == [["red", "green", "blue"]]


def test_pointer_field_tables_match_the_descriptor_structs():
    """es_plan_pointer_fields (what NativeEngine.save relocates by, instead of guessing pointers from bit patterns): for every
    descriptor struct mirrored in edgestyle_amd/lib.py, the library's table names exactly the pointer-typed fields of the
    struct - no more, no fewer -, each 8-byte aligned and inside the record, each with a use (1 reads, 2 writes, 3 both)."""
    L = lib.load()

    def table(kind):
        offs, uses = (ctypes.c_int32 * 64)(), (ctypes.c_int32 * 64)()
        elem = ctypes.c_int32(-1)
        n = L.es_plan_pointer_fields(kind, offs, uses, 64, ctypes.byref(elem))
        assert 0 <= n <= 64
        return [(offs[i], uses[i]) for i in range(n)], elem.value

    def struct_ptr_offsets(st):
        out = set()
        for name, tp in st._fields_:
            f = getattr(st, name)
            if tp is ctypes.c_void_p or (isinstance(tp, type) and issubclass(tp, ctypes._Pointer)):
                out.add(f.offset)
            elif isinstance(tp, type) and issubclass(tp, ctypes.Array) and (tp._type_ is ctypes.c_void_p):
                out.update(f.offset + 8 * i for i in range(tp._length_))
        return out

    # op kinds of csrc/plan.h: 1 conv_gemm, 2 linear_xs, 3 attention, 4 group_norm, 6 layer_norm_grouped, 7 / 8 fusion block(s)
    for kind, st in ((1, lib.GemmDesc), (2, lib.XsDesc), (3, lib.AttnDesc), (4, lib.GnDesc), (6, lib.LnDesc), (7, lib.FusionDesc),
                     (8, lib.FusionDesc)):
        fl, elem = table(kind)
        assert elem == (ctypes.sizeof(st) if kind == 8 else 0)
        assert {o for o, _ in fl} == struct_ptr_offsets(st), (kind, sorted(o for o, _ in fl), sorted(struct_ptr_offsets(st)))
        assert len({o for o, _ in fl}) == len(fl) and all(o % 8 == 0 and o + 8 <= ctypes.sizeof(st) and u in (1, 2, 3) for o, u in fl)
    # the outputs are marked as written
    g = dict(table(1)[0])
    assert g[lib.GemmDesc.out.offset] == 2 and g[lib.GemmDesc.w.offset] == 1 and g[lib.GemmDesc.workspace.offset] == 3
    # small argument records (memcpy, incr, ...) and markers / unknown kinds
    assert len(table(18)[0]) == 2 and len(table(16)[0]) == 1 and table(64) == ([], 0) and table(999) == ([], 0)


def test_graph_hazard_guard_sees_a_queue_intercepting_profiler():
    """es_ctx_graph_hazard (csrc/plan.hip): under rocprofv3 (ROCP_TOOL_LIBRARIES / a rocprofiler library in LD_PRELOAD) without
    DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 the HIP runtime faults below hipGraphLaunch (profiles/r04_rocprof_graph_fault.txt); the
    context then replays its plans launch by launch.  One process per environment: the answer is latched at first use."""
    import subprocess
    import sys
    code = ("import ctypes, sys; sys.path.insert(0, %r); from edgestyle_amd import lib; "
            "print(ctypes.CDLL(lib.LIB_PATH).es_ctx_graph_hazard())" % ROOT)

    def ask(**env):
        e = {k: v for k, v in os.environ.items() if k not in ("ROCP_TOOL_LIBRARIES", "LD_PRELOAD", "DEBUG_CLR_GRAPH_PACKET_CAPTURE")}
        e.update(env)
        r = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr[-500:]
        return int(r.stdout.strip().splitlines()[-1]), r.stderr
    assert ask()[0] == 0
    v, err = ask(ROCP_TOOL_LIBRARIES="/opt/rocm/lib/rocprofiler-sdk/librocprofiler-sdk-tool.so")
    assert v == 1 and "launch by launch" in err
    assert ask(ROCP_TOOL_LIBRARIES="/opt/rocm/lib/rocprofiler-sdk/librocprofiler-sdk-tool.so", DEBUG_CLR_GRAPH_PACKET_CAPTURE="0")[0] == 0


def test_big_tile_epilogue_forms():
    """es_conv_gemm8p_form_ok: the epilogue forms the 256 x 320 tile implements for splitk == 1 (both hosts ask before they fix
    bn = 320, so the planned / recorded / reported tile is the tile that runs)."""
    L = lib.load()
    ok = L.es_conv_gemm8p_form_ok
    assert ok(lib.ACT_NONE, 320, 0, 4096, 0) == 1 and ok(lib.ACT_NONE, 320, 0, 4096, 1) == 1
    assert ok(lib.ACT_NONE, 320, 1, 4096, 0) == 1          # one time-embedding row per 128-pixel half
    assert ok(lib.ACT_NONE, 320, 1, 4096, 1) == 0          # ... not beside a residual
    assert ok(lib.ACT_NONE, 320, 1, 64, 0) == 0            # ... nor when a half spans samples
    assert ok(lib.ACT_SILU, 320, 0, 4096, 0) == 0 and ok(lib.ACT_NONE, 4, 0, 4096, 0) == 0
