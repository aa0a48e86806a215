"""es_load_weights without a GPU (include/edgestyle_hip.h; csrc/builder.hip): the native builder's host logic - key lookup and
shape checks, LoRA / LayerNorm / proj_out / shortcut folds, weight packing, the launch planner, the model walk - against the
Python host (edgestyle_amd/engine.py, models.py, ops.py) run on the library's dry recorder: the same calls in the same order
with the same arguments, and the same bytes behind every weight pointer.  (The GPU side of the same comparison - outputs bit for
bit - is tests/test_load_weights_gpu.py.)"""
import ctypes as C
import dataclasses
import itertools

import pytest
import torch

from edgestyle_amd import config as Cfg, lib as L, ops
from edgestyle_amd.native import NativeContext
from tests.helpers import make_weights, quantize, python_dry_context, diff_plans, plan_constants


@pytest.fixture(scope="module")
def tiny():
    ucfg, vcfg = dataclasses.replace(Cfg.tiny_unet(), sample_size=64), Cfg.tiny_vae()
    ws = {k: quantize(v) for k, v in make_weights(ucfg, vcfg, seed=3).items()}
    return ucfg, vcfg, ws


def test_launch_planner_of_the_library_equals_the_python_hosts():
    """es_plan_gemm_choice == ops.plan_gemm over the GEMM shapes of the model (all levels, batch 1..8, every candidate set)."""
    lib = L.load()
    bn, sk, st = C.c_int(), C.c_int(), C.c_int()
    n = 0
    Ms = [77 * 2, 154 * 3, 64, 256, 1024, 2048, 4096, 8192, 12288, 16384, 24576, 32768, 57344, 65536, 131072, 229376, 458752, 1 << 20]
    rows = [128, 256, 320, 384, 640, 1280, 1920, 2560, 3840, 5120, 10240]
    ks = [64, 320, 640, 1280, 2880, 3200, 5760, 6400, 8640, 11520, 17280, 23040]
    cands = [(160, 128), (320, 160, 128, 64), (160, 128, 64), (320, 160, 128)]
    for M, r, k, cand, split in itertools.product(Ms, rows, ks, cands, (True, False)):
        try:
            want = ops.plan_gemm_reference(M, r, k, False, bns=cand, allow_split=split)
            assert ops.plan_gemm(M, r, k, False, bns=cand, allow_split=split) == want          # the Python host now asks the library
        except L.EdgeStyleHipError:
            want = None
        arr = (C.c_int * len(cand))(*cand)
        rc = lib.es_plan_gemm_choice(M, r, k, 0, arr, len(cand), int(split), C.byref(bn), C.byref(sk), C.byref(st))
        got = None if rc else (bn.value, sk.value, st.value)
        assert got == want, (M, r, k, cand, split, got, want)
        n += 1
    assert n > 10000
    # the 256 x 256 tile of the LayerNorm-folded / GEGLU linear layers (round 5): offered first, with and without GEGLU, never split
    for M, r, k, geglu in itertools.product(Ms, [1280, 3840, 10240, 2560, 5120], [320, 640, 1280], (False, True)):
        for cand in ((256, 160, 128), (256, 160, 128, 64), (256,)):
            want = ops.plan_gemm_reference(M, r, k, geglu, bns=cand, allow_split=False)
            assert ops.plan_gemm(M, r, k, geglu, bns=cand, allow_split=False) == want, (M, r, k, geglu, cand)
            assert want[1] == 1 and (not geglu or want[0] in (128, 256))
            if want[0] == 256:
                assert want[2] == 2 and (len(cand) == 1 or (M >= ops.PLAN_BIG_MIN_M and k // 64 >= ops.PLAN_BIG_MIN_NK))
    arr = (C.c_int * 2)(160, 128)
    assert lib.es_plan_gemm_choice(4096, 2560, 320, 1, arr, 2, 1, C.byref(bn), C.byref(sk), C.byref(st)) == 0
    assert (bn.value, sk.value, st.value) == ops.plan_gemm(4096, 2560, 320, True) == ops.plan_gemm_reference(4096, 2560, 320, True)


def test_short_k_kernel_policy_of_the_library_equals_the_python_hosts():
    lib = L.load()
    for M, K, cout, geglu in itertools.product([1024, 8192, 16384, 32768, 57344, 458752], [320, 640, 1280, 768], [320, 640, 960, 1920, 2560, 5120],
                                               [False, True]):
        pw = ops.PackedWeight(w=torch.empty(0, K), bias=None, cout=cout, cin=K, ksize=1, bn=128, geglu=geglu)
        want = ops.xs_shape_reference(M, pw, 8192)
        assert bool(lib.es_linear_xs_eligible(M, 1, K, K, 0, cout, int(geglu))) == want == ops.xs_eligible(M, pw, None, None, 1), (M, K, cout, geglu)


@pytest.mark.parametrize("B,guidance,T", [(1, True, 6), (2, False, 3)])
def test_native_builder_records_the_python_hosts_calls_on_the_same_weight_bytes(tiny, B, guidance, T):
    ucfg, vcfg, ws = tiny
    lib, pctx, keep = python_dry_context(ws, ucfg, vcfg, B, guidance, T)
    nat = NativeContext(ws, ucfg, vcfg, batch_size=B, guidance=guidance, num_inference_steps=T, device=-2)
    try:
        for which in range(L.PLAN_COUNT):
            assert lib.es_ctx_plan_size(pctx, which) == nat.plan_size(which) > 20
            assert diff_plans(lib, pctx, nat.ctx, which) is None
            a, b = plan_constants(lib, pctx, which), plan_constants(lib, nat.ctx, which)
            assert len(a) == len(b) and sum(map(len, a)) > 1 << 20
            assert all(x == y for x, y in zip(a, b)), [i for i, (x, y) in enumerate(zip(a, b)) if x != y][:8]
        assert lib.es_plan_count(lib.es_ctx_plan(nat.ctx, L.PLAN_STEP), 8) == 1          # ONE es_fusion_blocks call per step
    finally:
        nat.close()
        lib.es_ctx_destroy(pctx)


def test_other_net_patterns_two_nets_and_a_lora_net_with_its_own_conv_stack(tiny):
    """Beyond the reference's [lora, pose, lora', pose, lora', pose]: two distinct nets serving three slots each, and a
    ControlLoRA net conditioned through its own conv stack instead of the VAE (uses_vae False, CL:529-598) - the same calls
    and the same weight bytes from both builders."""
    from edgestyle_amd import weights as W
    ucfg, vcfg, ws = tiny
    lib = L.load()
    ws2 = dict(ws, lora_stack=quantize(W.random_state_dict(W.controllora_saved_shapes(ucfg, 4, uses_vae=False), 3, "controlnet_2.")))
    for nets, slots in (((("lora0", 1), ("openpose", 0)), (0, 1, 0, 1, 0, 1)),
                        ((("lora_stack", 2), ("openpose", 0), ("lora1", 1)), (0, 1, 2, 1, 2, 2))):
        _, pctx, keep = python_dry_context(ws2, ucfg, vcfg, 1, True, 4, controlnets=nets, net_of_cond=slots)
        nat = NativeContext(ws2, ucfg, vcfg, num_inference_steps=4, device=-2, controlnets=nets, net_of_cond=slots)
        try:
            for which in range(L.PLAN_COUNT):
                assert diff_plans(lib, pctx, nat.ctx, which) is None
                assert plan_constants(lib, pctx, which) == plan_constants(lib, nat.ctx, which)
        finally:
            nat.close()
            lib.es_ctx_destroy(pctx)


def test_single_controlnet_context(tiny):
    """n_conds = 1 (BASELINE configs[0]: the UNet + ONE openpose ControlNet, PL:338-351): 13 residuals scaled by the
    conditioning scale and added to the UNet's skips straight out of the zero-conv epilogues, no fusion blocks."""
    ucfg, vcfg, ws = tiny
    lib = L.load()
    nets, slots = (("openpose", 0),), (0,)
    _, pctx, keep = python_dry_context(ws, ucfg, vcfg, 1, True, 4, controlnets=nets, net_of_cond=slots)
    nat = NativeContext({k: v for k, v in ws.items() if k != "fusion"}, ucfg, vcfg, num_inference_steps=4, device=-2, controlnets=nets, net_of_cond=slots)
    try:
        for which in range(L.PLAN_COUNT):
            assert diff_plans(lib, pctx, nat.ctx, which) is None
            assert plan_constants(lib, pctx, which) == plan_constants(lib, nat.ctx, which)
        assert lib.es_plan_count(lib.es_ctx_plan(nat.ctx, L.PLAN_STEP), 8) == 0            # no fusion blocks
    finally:
        nat.close()
        lib.es_ctx_destroy(pctx)
    with pytest.raises(L.EdgeStyleHipError, match="exactly one ControlNet"):
        NativeContext(ws, ucfg, vcfg, num_inference_steps=4, device=-1, net_of_cond=(0,))


def test_sources_in_fp16_and_bf16_compute_type(tiny):
    """Checkpoints stored in fp16 (the usual case) describe the same values; a bf16 context builds too."""
    ucfg, vcfg, ws = tiny
    lib = L.load()
    a = NativeContext(ws, ucfg, vcfg, num_inference_steps=4, device=-2)
    ws16 = {k: {kk: vv.half() for kk, vv in v.items()} for k, v in ws.items()}         # exact: the fixture is fp16-rounded
    b = NativeContext(ws16, ucfg, vcfg, num_inference_steps=4, device=-2)
    c = NativeContext(ws, ucfg, vcfg, num_inference_steps=4, device=-2, dtype=torch.bfloat16)
    _, pctx, keep = python_dry_context(ws, ucfg, vcfg, 1, True, 4, dtype=torch.bfloat16)
    try:
        assert plan_constants(lib, a.ctx, L.PLAN_STEP) == plan_constants(lib, b.ctx, L.PLAN_STEP)
        assert c.plan_size(L.PLAN_STEP) == a.plan_size(L.PLAN_STEP)
        for which in range(L.PLAN_COUNT):                                   # bf16: the same calls, the same rounding of every packed value
            assert diff_plans(lib, pctx, c.ctx, which) is None
            assert plan_constants(lib, pctx, which) == plan_constants(lib, c.ctx, which)
        assert plan_constants(lib, c.ctx, L.PLAN_STEP) != plan_constants(lib, a.ctx, L.PLAN_STEP)
    finally:
        for x in (a, b, c):
            x.close()
        lib.es_ctx_destroy(pctx)


def test_missing_keys_wrong_shapes_and_unsupported_requests_are_named(tiny):
    ucfg, vcfg, ws = tiny

    def build(ws_, **kw):
        return NativeContext(ws_, ucfg, vcfg, num_inference_steps=3, device=-1, **kw)

    def without(name, key):
        d = dict(ws)
        d[name] = {k: v for k, v in ws[name].items() if k != key}
        return d
    with pytest.raises(L.EdgeStyleHipError, match="missing key 'mid_block.resnets.1.conv2.weight' in the UNet state dict"):
        build(without("unet", "mid_block.resnets.1.conv2.weight"))
    with pytest.raises(L.EdgeStyleHipError, match="missing key 'controlnet_mid_block.weight' in the ControlNet 2"):
        build(without("lora1", "controlnet_mid_block.weight"))
    with pytest.raises(L.EdgeStyleHipError, match="missing key 'multi_controlnet_mid_block.third_conv.bias' in the fusion"):
        build(without("fusion", "multi_controlnet_mid_block.third_conv.bias"))
    with pytest.raises(L.EdgeStyleHipError, match="missing key 'decoder.conv_out.weight' in the VAE"):
        build(without("vae", "decoder.conv_out.weight"))
    # a LoRA pair with only one half
    with pytest.raises(L.EdgeStyleHipError, match="lora_layer.up.weight"):
        build(without("lora0", "time_embedding.linear_1.lora_layer.up.weight"))
    bad = dict(ws)
    bad["fusion"] = dict(ws["fusion"])
    k = "multi_controlnet_down_blocks.0.first_normalization.weight"
    bad["fusion"][k] = ws["fusion"][k][:, :32]
    with pytest.raises(L.EdgeStyleHipError, match="size mismatch for 'multi_controlnet_down_blocks.0.first_normalization.weight'"):
        build(bad)
    bad = dict(ws)
    bad["unet"] = dict(ws["unet"])
    bad["unet"]["conv_norm_out.weight"] = ws["unet"]["conv_norm_out.weight"][:32]
    with pytest.raises(L.EdgeStyleHipError, match="size mismatch for 'conv_norm_out.weight'"):
        build(bad)
    bad = dict(ws)
    bad["openpose"] = dict(ws["openpose"])
    k = "down_blocks.1.attentions.0.transformer_blocks.0.attn2.to_v.weight"
    bad["openpose"][k] = ws["openpose"][k][:64]                                           # a K/V projection of the wrong width
    with pytest.raises(L.EdgeStyleHipError, match="size mismatch for 'down_blocks.1.attentions.0.transformer_blocks.0.attn2.to_v.weight' in the ControlNet 1"):
        build(bad)
    bad = dict(ws)
    bad["vae"] = dict(ws["vae"])
    bad["vae"]["decoder.conv_out.weight"] = ws["vae"]["decoder.conv_out.weight"][:, :16]
    with pytest.raises(L.EdgeStyleHipError, match="size mismatch for 'decoder.conv_out.weight' in the VAE"):
        build(bad)
    with pytest.raises(L.EdgeStyleHipError, match="net_of_cond"):
        build(ws, net_of_cond=(0, 1, 2, 1, 3, 1))
    # 16x16 latents: the groups of the lockstep pass do not tile in 128-pixel units (the Python host falls back to serial chains)
    with pytest.raises(L.EdgeStyleHipError, match="do not tile"):
        NativeContext(ws, dataclasses.replace(ucfg, sample_size=16), vcfg, num_inference_steps=3, device=-1)
    with pytest.raises(L.EdgeStyleHipError, match="host tensors"):
        build(dict(ws, unet={k: v.double() for k, v in ws["unet"].items()}))


def test_a_dry_context_cannot_be_saved_or_run(tiny, tmp_path):
    """device = -1 / -2 builds are for inspection: no device arena, so es_ctx_save refuses them by name, and so does every
    entry point that would launch their plans (their recorded addresses are arena-relative or host memory: launching them
    was a GPU fault, not an error code - ADVICE r3)."""
    ucfg, vcfg, ws = tiny
    lib = L.load()
    for dev in (-2, -1):
        c = NativeContext(ws, ucfg, vcfg, num_inference_steps=3, device=dev)
        try:
            assert lib.es_ctx_arena_bytes(c.ctx) > 100 << 20
            assert lib.es_ctx_save(c.ctx, str(tmp_path / "x.esctx").encode()) != 0 and b"does not own a device arena" in lib.es_last_error()
            host = torch.zeros(1 << 16)
            hp = C.c_void_p(host.data_ptr())
            six = (C.c_void_p * 6)(*([host.data_ptr()] * 6))
            ts = (C.c_float * 3)(981.0, 641.0, 301.0)
            calls = [
                ("es_ctx_launch_plan", lambda: lib.es_ctx_launch_plan(c.ctx, L.PLAN_STEP, None, None)),
                ("es_vae_decode", lambda: lib.es_vae_decode(c.ctx, hp, hp, None)),
                ("es_denoise_loop", lambda: lib.es_denoise_loop(c.ctx, hp, hp, 7.5, ts, 3, None)),
                ("es_denoise_step", lambda: lib.es_denoise_step(c.ctx, hp, 501.0, hp, six, None, hp, None)),
                ("es_prepare_conds", lambda: lib.es_prepare_conds(c.ctx, six, six, None)),
            ]
            for name, call in calls:
                assert call() != 0, name
                assert b"inspection build" in lib.es_last_error(), (name, lib.es_last_error())
        finally:
            c.close()


def test_dry_recording_validates_but_does_not_launch():
    """es_plan_set_dry: a recording thread's calls are checked and recorded, nothing runs (no GPU here) - and a call the
    kernel would reject is rejected and NOT recorded."""
    lib = L.load()
    plan = C.c_void_p(lib.es_plan_create())
    assert lib.es_plan_begin_record(plan) == 0
    lib.es_plan_set_dry(1)
    try:
        buf = torch.zeros(64)
        assert lib.es_fill_f32(C.c_void_p(buf.data_ptr()), 1.0, 64, None) == 0
        assert lib.es_add(C.c_void_p(buf.data_ptr()), C.c_void_p(buf.data_ptr()), C.c_void_p(buf.data_ptr()), 7, L.ES_F16, None) != 0
        assert lib.es_plan_size(plan) == 1 and float(buf.sum()) == 0.0
        # a head width that passes d % 8 but has no kernel (e.g. block_out_channels / heads = 56) is refused while RECORDING,
        # not at the first launch: es_load_weights therefore rejects such a config like the eager Python builder does
        d = L.AttnDesc()
        q = torch.zeros(4 * 64 * 56 * 2, dtype=torch.float16)
        d.q = d.k = d.v = d.o = q.data_ptr()
        d.N, d.heads, d.Sq, d.Skv, d.d = 1, 2, 64, 64, 56
        d.ldq = d.ldk = d.ldv = d.ldo = 112
        d.bsq = d.bsk = d.bsv = d.bso = 64 * 112
        d.scale, d.dtype = 0.1, L.ES_F16
        assert lib.es_attention(C.byref(d), None) != 0 and b"unsupported head_dim" in lib.es_last_error()
        assert lib.es_plan_size(plan) == 1
        d.d, d.ldq, d.ldk, d.ldv, d.ldo = 40, 80, 80, 80, 80
        assert lib.es_attention(C.byref(d), None) == 0 and lib.es_plan_size(plan) == 2
    finally:
        assert lib.es_plan_set_dry(0) == 1
        lib.es_plan_end_record(plan)
        lib.es_plan_destroy(plan)


@pytest.mark.parametrize("B,guidance,nets,slots", [(1, True, None, None), (2, False, None, None), (1, True, (("openpose", 0),), (0,))])
def test_guess_mode_context(tiny, B, guidance, nets, slots):
    """es_ctx_geometry.guess_mode (CL:256-264; PL:453-459, 487-497): the ControlNets as per-group chains on the conditional half
    only, zero-convs scaled 0.1..1 log-spaced, the fused residuals added into the conditional half of the UNet's skips in place -
    the calls NativeEngine(guess_mode=True) records, argument for argument (the level scales included: float bits)."""
    ucfg, vcfg, ws = tiny
    lib = L.load()
    kw = {} if nets is None else dict(controlnets=nets, net_of_cond=slots)
    _, pctx, keep = python_dry_context(ws, ucfg, vcfg, B, guidance, 4, guess=True, **kw)
    wsn = ws if nets is None else {k: v for k, v in ws.items() if k != "fusion"}
    nat = NativeContext(wsn, ucfg, vcfg, batch_size=B, guidance=guidance, num_inference_steps=4, device=-2, guess_mode=True, **kw)
    try:
        for which in range(L.PLAN_COUNT):
            if which == L.PLAN_STEP_UNET:
                assert nat.plan_size(which) == -1                              # steps outside the window replay ES_PLAN_STEP with scales 0
                continue
            assert lib.es_ctx_plan_size(pctx, which) == nat.plan_size(which) > 5
            assert diff_plans(lib, pctx, nat.ctx, which) is None
            assert plan_constants(lib, pctx, which) == plan_constants(lib, nat.ctx, which)
        n_adds = lib.es_plan_count(lib.es_ctx_plan(nat.ctx, L.PLAN_STEP), 14)          # ES_OP_ADD: 12 skips + the mid tensor
        assert n_adds == 13
    finally:
        nat.close()
        lib.es_ctx_destroy(pctx)
