"""ctypes binding of libedgestyle_hip.so (the C ABI declared in include/edgestyle_hip.h).

The product path has NO fallback: if the shared library is missing or an entry point fails, this raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# ES_HIP_LIB: tools only (ablation / experimental kernel builds); the product always loads the in-tree library
LIB_PATH = os.environ.get("ES_HIP_LIB") or os.path.join(_HERE, "lib", "libedgestyle_hip.so")

ES_F16, ES_BF16, ES_F32 = 0, 1, 2
ABI_VERSION = 7          # include/edgestyle_hip.h ES_ABI_VERSION
ACT_NONE, ACT_SILU, ACT_GEGLU = 0, 1, 2


class GemmDesc(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("x2", C.c_void_p), ("w", C.c_void_p), ("bias", C.c_void_p), ("temb", C.c_void_p),
        ("residual", C.c_void_p), ("out_scale_dev", C.c_void_p), ("out", C.c_void_p), ("workspace", C.c_void_p),
        ("prof", C.c_void_p),
        ("N", C.c_int32), ("Hsrc", C.c_int32), ("Wsrc", C.c_int32), ("C1", C.c_int32), ("C2", C.c_int32),
        ("Hout", C.c_int32), ("Wout", C.c_int32), ("Cout", C.c_int32),
        ("rows_padded", C.c_int32), ("Kpad", C.c_int32),
        ("ksize", C.c_int32), ("stride", C.c_int32), ("pad", C.c_int32),
        ("upsample", C.c_int32), ("temb_stride", C.c_int32), ("act", C.c_int32), ("splitk", C.c_int32),
        ("bn", C.c_int32), ("dtype", C.c_int32),
        ("ngroups", C.c_int32), ("mt_end", C.c_int32 * 4), ("w_g", C.c_void_p * 4), ("bias_g", C.c_void_p * 4),
        ("stages", C.c_int32), ("xcd_m_fastest", C.c_int32), ("waves", C.c_int32),
        ("out_scale", C.c_float),
        ("ln_colsum", C.c_void_p), ("ln_colsum_g", C.c_void_p * 4), ("ln_eps", C.c_float),
        ("t1", C.c_void_p), ("t2", C.c_void_p), ("Ct1", C.c_int32), ("Ct2", C.c_int32),
        ("x_nmod", C.c_int32), ("korder", C.c_int32), ("residual_lo", C.c_void_p), ("out_lo", C.c_void_p),
        ("gn_part", C.c_void_p), ("gn_groups", C.c_int32),
    ]


class XsDesc(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("out", C.c_void_p), ("w", C.c_void_p), ("bias", C.c_void_p),
        ("w_g", C.c_void_p * 4), ("bias_g", C.c_void_p * 4), ("prof", C.c_void_p),
        ("mt_end", C.c_int32 * 4), ("ngroups", C.c_int32),
        ("M", C.c_int32), ("K", C.c_int32), ("Cout", C.c_int32), ("rows_padded", C.c_int32),
        ("ldo", C.c_int32), ("geglu", C.c_int32), ("ln", C.c_int32), ("ln_eps", C.c_float),
        ("nslices", C.c_int32), ("chunks_per_slice", C.c_int32), ("dtype", C.c_int32),
        ("residual", C.c_void_p),
        ("gn_part", C.c_void_p), ("gn_gamma", C.c_void_p), ("gn_beta", C.c_void_p), ("gn_gamma_g", C.c_void_p * 4), ("gn_beta_g", C.c_void_p * 4),
        ("gn_groups", C.c_int32), ("gn_nchunk", C.c_int32), ("gn_hw", C.c_int32), ("gn_eps", C.c_float),
    ]


class CtxGeometry(C.Structure):
    _fields_ = [("B", C.c_int32), ("cfg", C.c_int32), ("h", C.c_int32), ("w", C.c_int32),
                ("latent_channels", C.c_int32), ("latent_pad", C.c_int32), ("n_conds", C.c_int32),
                ("n_steps", C.c_int32), ("dtype", C.c_int32), ("guess_mode", C.c_int32)]


# es_plan / es_ctx enums (include/edgestyle_hip.h)
PLAN_STEP_GENERIC, PLAN_PREP, PLAN_STEP, PLAN_DECODE, PLAN_CONDS, PLAN_STEP_UNET = 0, 1, 2, 3, 4, 5
PLAN_COUNT = 6
(BUF_SAMPLE, BUF_T_ROWS, BUF_EHS, BUF_COND0, BUF_COND1, BUF_COND2, BUF_COND3, BUF_COND4, BUF_COND5, BUF_SCALES, BUF_NOISE,
 BUF_LATENTS, BUF_STEP_IDX, BUF_T_TABLE, BUF_SCALE_TABLE, BUF_COEF, BUF_TIMESTEPS, BUF_IMAGE) = range(18)
BUF_COND_IMG0, BUF_COND_NOISE0 = 18, 24          # + net index
BUF_HIST0, BUF_COUNT = 30, 33                    # UniPC state slots (+ 0..2)
SCHED_DDIM, SCHED_UNIPC = 0, 1
OP_CONV_GEMM, OP_LINEAR_XS, OP_ATTENTION = 1, 2, 3      # csrc/plan.h es_op_kind (es_plan_count)


class AttnDesc(C.Structure):
    _fields_ = [
        ("q", C.c_void_p), ("k", C.c_void_p), ("v", C.c_void_p), ("o", C.c_void_p),
        ("N", C.c_int32), ("heads", C.c_int32), ("Sq", C.c_int32), ("Skv", C.c_int32), ("d", C.c_int32),
        ("ldq", C.c_int32), ("ldk", C.c_int32), ("ldv", C.c_int32), ("ldo", C.c_int32),
        ("bsq", C.c_int64), ("bsk", C.c_int64), ("bsv", C.c_int64), ("bso", C.c_int64),
        ("scale", C.c_float), ("dtype", C.c_int32),
    ]


class GnDesc(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("x2", C.c_void_p), ("out", C.c_void_p),
        ("gamma", C.c_void_p), ("beta", C.c_void_p), ("partials", C.c_void_p),
        ("N", C.c_int32), ("HW", C.c_int32), ("C1", C.c_int32), ("C2", C.c_int32), ("groups", C.c_int32),
        ("eps", C.c_float), ("silu", C.c_int32), ("dtype", C.c_int32),
        ("ngroups", C.c_int32), ("n_end", C.c_int32 * 4), ("gamma_g", C.c_void_p * 4), ("beta_g", C.c_void_p * 4),
        ("ext_chunks", C.c_int32), ("stats_only", C.c_int32),
    ]


class LnDesc(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("out", C.c_void_p), ("gamma_g", C.c_void_p * 4), ("beta_g", C.c_void_p * 4),
        ("row_end", C.c_int32 * 4), ("ngroups", C.c_int32), ("M", C.c_int32), ("C", C.c_int32),
        ("eps", C.c_float), ("dtype", C.c_int32),
    ]


class FusionDesc(C.Structure):
    _fields_ = [
        ("res", C.c_void_p * 6), ("res_bs", C.c_int64 * 6),
        ("w1", C.c_void_p), ("b1", C.c_void_p), ("g1", C.c_void_p), ("be1", C.c_void_p),
        ("w2", C.c_void_p), ("b2", C.c_void_p), ("g2", C.c_void_p), ("be2", C.c_void_p),
        ("w3", C.c_void_p), ("b3", C.c_void_p),
        ("scratch", C.c_void_p), ("u", C.c_void_p), ("out", C.c_void_p),
        ("res_scale_dev", C.c_void_p), ("addend", C.c_void_p),
        ("res_scale", C.c_float * 6),
        ("N", C.c_int32), ("HW", C.c_int32), ("C", C.c_int32),
        ("eps", C.c_float), ("dtype", C.c_int32),
    ]


class Tensor(C.Structure):          # es_tensor
    _fields_ = [("key", C.c_char_p), ("data", C.c_void_p), ("shape", C.c_int64 * 4), ("ndim", C.c_int32), ("dtype", C.c_int32)]


class StateDict(C.Structure):       # es_state_dict
    _fields_ = [("tensors", C.POINTER(Tensor)), ("count", C.c_int32)]


NET_CONTROLNET, NET_CONTROL_LORA_VAE, NET_CONTROL_LORA = 0, 1, 2


class Weights(C.Structure):         # es_weights
    _fields_ = [("unet", StateDict), ("vae", StateDict), ("fusion", StateDict), ("controlnet", StateDict * 6),
                ("controlnet_kind", C.c_int32 * 6), ("n_controlnets", C.c_int32), ("net_of_cond", C.c_int32 * 6)]


class ModelConfig(C.Structure):     # es_model_config
    _fields_ = [("in_channels", C.c_int32), ("out_channels", C.c_int32),
                ("n_blocks", C.c_int32), ("block_out_channels", C.c_int32 * 4), ("down_has_attn", C.c_int32 * 4),
                ("layers_per_block", C.c_int32), ("num_heads", C.c_int32), ("cross_attention_dim", C.c_int32),
                ("norm_num_groups", C.c_int32), ("norm_eps", C.c_float),
                ("n_cond_embed", C.c_int32), ("cond_embed_channels", C.c_int32 * 4), ("conditioning_channels", C.c_int32),
                ("text_tokens", C.c_int32),
                ("vae_n_blocks", C.c_int32), ("vae_block_out_channels", C.c_int32 * 4), ("vae_layers_per_block", C.c_int32),
                ("vae_latent_channels", C.c_int32), ("vae_norm_num_groups", C.c_int32),
                ("vae_norm_eps", C.c_float), ("vae_scaling_factor", C.c_float)]


# every symbol include/edgestyle_hip.h declares: (name, restype, argtypes)
_P, _I, _F, _L = C.c_void_p, C.c_int, C.c_float, C.c_int64
SYMBOLS = {
    "es_abi_version": (C.c_int, []),
    "es_set_operand_limit": (C.c_ulonglong, [C.c_ulonglong]),
    "es_last_error": (C.c_char_p, []),
    "es_sizeof_desc": (C.c_size_t, [_I]),
    "es_conv_gemm": (C.c_int, [C.POINTER(GemmDesc), _P]),
    "es_conv_gemm_workspace_bytes": (C.c_size_t, [C.POINTER(GemmDesc)]),
    "es_linear_xs": (C.c_int, [C.POINTER(XsDesc), _P]),
    "es_linear_xs_set_pp": (C.c_int, [_I]),
    "es_attention": (C.c_int, [C.POINTER(AttnDesc), _P]),
    "es_attention_set_kvres": (C.c_int, [_I]),
    "es_group_norm": (C.c_int, [C.POINTER(GnDesc), _P]),
    "es_group_norm_partials_bytes": (C.c_size_t, [_I, _I]),
    "es_group_norm_is_slab": (C.c_int, [_I, _I, _I]),
    "es_group_norm_chunks": (C.c_int, [_I]),
    "es_layer_norm": (C.c_int, [_P, _P, _P, _P, _I, _I, _F, _I, _P]),
    "es_fusion_block": (C.c_int, [C.POINTER(FusionDesc), _P]),
    "es_fusion_scratch_bytes": (C.c_size_t, [_I]),
    "es_fusion_blocks": (C.c_int, [C.POINTER(FusionDesc), _I, _P]),
    "es_timestep_embedding": (C.c_int, [_P, _P, _I, _I, _I, _P]),
    "es_cfg_ddim_step": (C.c_int, [_P, _P, _P, _P, _P, _F, _I, _I, _I, _I, _I, _I, _I, _P]),
    "es_cfg_unipc_step": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _F, _I, _I, _I, _I, _I, _I, _I, _P]),
    "es_nchw_f32_to_nhwc": (C.c_int, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "es_nhwc_to_nchw_f32": (C.c_int, [_P, _P, _I, _I, _I, _I, _F, _F, _I, _I, _P]),
    "es_add": (C.c_int, [_P, _P, _P, _L, _I, _P]),
    "es_vae_sample": (C.c_int, [_P, _P, _P, _I, _I, _I, _I, _F, _I, _P]),
    "es_incr": (C.c_int, [_P, _P]),
    "es_clock_probe": (C.c_int, [_P, C.c_uint, _P]),
    "es_gather_row": (C.c_int, [_P, _P, _P, _I, _I, _P]),
    "es_layer_norm_grouped": (C.c_int, [C.POINTER(LnDesc), _P]),
    "es_latents_to_input": (C.c_int, [_P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "es_memcpy": (C.c_int, [_P, _P, C.c_size_t, _P]),
    "es_memcpy2d": (C.c_int, [_P, C.c_size_t, _P, C.c_size_t, C.c_size_t, C.c_size_t, _P]),
    "es_fill_f32": (C.c_int, [_P, _F, C.c_size_t, _P]),
    "es_plan_create": (_P, []),
    "es_plan_destroy": (None, [_P]),
    "es_plan_begin_record": (C.c_int, [_P]),
    "es_plan_end_record": (C.c_int, [_P]),
    "es_plan_size": (C.c_int, [_P]),
    "es_plan_count": (C.c_int, [_P, _I]),
    "es_plan_launch": (C.c_int, [_P, _P]),
    "es_ctx_create": (C.c_int, [_I, C.POINTER(_P)]),
    "es_ctx_destroy": (None, [_P]),
    "es_ctx_set_geometry": (C.c_int, [_P, C.POINTER(CtxGeometry)]),
    "es_ctx_set_plan": (C.c_int, [_P, _I, _P]),
    "es_ctx_bind": (C.c_int, [_P, _I, _P, C.c_size_t]),
    "es_ctx_buffer": (_P, [_P, _I, C.POINTER(C.c_size_t)]),
    "es_ctx_arena_bytes": (C.c_size_t, [_P]),
    "es_ctx_set_options": (C.c_int, [_P, C.POINTER(C.c_float), _F, _F, _I]),
    "es_ctx_set_scheduler": (C.c_int, [_P, _I]),
    "es_ctx_set_alphas_cumprod_f64": (C.c_int, [_P, C.POINTER(C.c_double), _I]),
    "es_unipc_coef_table": (C.c_int, [C.POINTER(C.c_double), _I, C.POINTER(C.c_float), _I, C.POINTER(C.c_float)]),
    "es_ctx_set_alphas_cumprod": (C.c_int, [_P, C.POINTER(C.c_float), _I]),
    "es_ctx_plan_size": (C.c_int, [_P, _I]),
    "es_ctx_plan": (_P, [_P, _I]),
    "es_ctx_load": (C.c_int, [C.c_char_p, _I, C.POINTER(_P)]),
    "es_ctx_save": (C.c_int, [_P, C.c_char_p]),
    "es_plan_export": (C.c_size_t, [_P, _P, C.c_size_t]),
    "es_plan_import": (_P, [_P, C.c_size_t]),
    "es_plan_pointer_fields": (C.c_int, [_I, C.POINTER(C.c_int32), C.POINTER(C.c_int32), _I, C.POINTER(C.c_int32)]),
    "es_ctx_launch_plan": (C.c_int, [_P, _I, C.POINTER(C.c_float), _P]),
    "es_ddim_coef_table": (C.c_int, [C.POINTER(C.c_float), _I, C.POINTER(C.c_float), _I, C.POINTER(C.c_float)]),
    "es_denoise_step": (C.c_int, [_P, _P, _F, _P, C.POINTER(_P), C.POINTER(C.c_float), _P, _P]),
    "es_denoise_loop": (C.c_int, [_P, _P, _P, _F, C.POINTER(C.c_float), _I, _P]),
    "es_vae_decode": (C.c_int, [_P, _P, _P, _P]),
    "es_prepare_conds": (C.c_int, [_P, C.POINTER(_P), C.POINTER(_P), _P]),
    "es_load_weights": (C.c_int, [C.POINTER(Weights), C.POINTER(ModelConfig), C.POINTER(CtxGeometry), _I, C.POINTER(_P)]),
    "es_plan_gemm_choice": (C.c_int, [C.c_longlong, _I, _I, _I, C.POINTER(C.c_int), _I, _I, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                      C.POINTER(C.c_int)]),
    "es_linear_xs_eligible": (C.c_int, [C.c_longlong, _I, _I, _I, _I, _I, _I]),
    "es_conv_gemm8p_form_ok": (C.c_int, [_I, _I, _I, C.c_longlong, _I]),
    "es_ctx_graph_hazard": (C.c_int, []),
    "es_plan_set_dry": (C.c_int, [_I]),
}

_lib = None


FUSION_MAX_BATCH = 13     # ES_FUSION_MAX_BATCH


class EdgeStyleHipError(RuntimeError):
    pass


def load():
    """Load the HIP library (once).  Raises EdgeStyleHipError — never falls back to another implementation."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EdgeStyleHipError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback for the product path.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)           # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    if lib.es_abi_version() != ABI_VERSION:
        raise EdgeStyleHipError("libedgestyle_hip.so ABI version mismatch")
    for i, st in enumerate((GemmDesc, AttnDesc, GnDesc, FusionDesc, LnDesc, XsDesc)):
        if lib.es_sizeof_desc(i) != C.sizeof(st):
            raise EdgeStyleHipError(f"descriptor layout mismatch for {st.__name__}: C {lib.es_sizeof_desc(i)} "
                                    f"vs ctypes {C.sizeof(st)}")
    _lib = lib
    return lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().es_last_error()
        raise EdgeStyleHipError(f"{what} failed (rc={rc}): {msg.decode() if msg else ''}")
