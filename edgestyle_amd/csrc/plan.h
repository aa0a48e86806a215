// Launch-list recorder shared by every C-ABI entry point (private to the library; the public side is in
// include/edgestyle_hip.h: es_plan_* / es_ctx_*).  While a plan is recording on the calling thread, each entry point
// appends its arguments to the plan (and still launches: the host flow that builds a plan is an ordinary eager or
// stream-capturing run); es_plan_launch re-issues the list on any stream without the host language in the loop.
#pragma once
#include <stddef.h>
#include <stdint.h>

enum es_op_kind {
  ES_OP_CONV_GEMM = 1, ES_OP_LINEAR_XS, ES_OP_ATTENTION, ES_OP_GROUP_NORM, ES_OP_LAYER_NORM, ES_OP_LAYER_NORM_GROUPED,
  ES_OP_FUSION_BLOCK, ES_OP_FUSION_BLOCKS, ES_OP_TIMESTEP_EMBEDDING, ES_OP_CFG_DDIM, ES_OP_CFG_UNIPC, ES_OP_NCHW_TO_NHWC,
  ES_OP_NHWC_TO_NCHW, ES_OP_ADD, ES_OP_VAE_SAMPLE, ES_OP_INCR, ES_OP_GATHER_ROW, ES_OP_MEMCPY, ES_OP_MEMCPY2D,
  ES_OP_FILL_F32, ES_OP_LATENTS_TO_INPUT,
};

struct es_op_layer_norm { const void* x; void* out; const float* gamma; const float* beta; int M, C; float eps; int dtype; };
struct es_op_timestep { const float* t; void* out; int N, dim, dtype, pad_; };   // pad_: no indeterminate bytes in a record
struct es_op_cfg_ddim { const void* noise; float* latents; void* model_in; const float* coef; const int32_t* step_idx;
                        float guidance_scale; int B, HW, L, Lstride, cfg, nsteps, dtype; };
struct es_op_cfg_unipc { const void* noise; float* latents; float* last_sample; float* m0; float* m1; void* model_in;
                         const float* coef; const int32_t* step_idx; float guidance_scale; int B, HW, L, Lstride, cfg, nsteps, dtype; };
struct es_op_nchw_to_nhwc { const float* in; void* out; int N, C, HW, Cpad, dtype, pad_; };
struct es_op_nhwc_to_nchw { const void* in; float* out; int N, C, HW, Cstride; float scale, shift; int clamp01, dtype; };
struct es_op_add { const void* a; const void* b; void* y; int64_t n; int dtype, pad_; };
struct es_op_vae_sample { const void* moments; const float* noise; void* z; int N, HW, L, Lpad; float scaling; int dtype; };
struct es_op_incr { int32_t* ctr; };
struct es_op_gather_row { const float* table; const int32_t* idx; float* out; int row_len, nrows; };
struct es_op_memcpy { void* dst; const void* src; size_t bytes; };
struct es_op_memcpy2d { void* dst; size_t dpitch; const void* src; size_t spitch; size_t width, height; };
struct es_op_fill_f32 { float* dst; size_t n; float value; int pad_; };
struct es_op_latents_to_input { const float* latents; void* model_in; int B, HW, L, Lstride, cfg, dtype; };

// true while a plan records on this thread; the entry points call es_plan_record(kind, args, bytes) first
extern "C" int es_plan_recording(void);
extern "C" void es_plan_record(int kind, const void* args, size_t bytes);
// es_plan_set_dry(1): a recording thread validates and records every call but launches nothing - the native builder
// (builder.hip) records its plans against addresses of an arena that is only allocated afterwards
extern "C" int es_plan_dry(void);
#define ES_PLAN_RECORD(kind, ptr, bytes) do { if (es_plan_recording()) { es_plan_record((kind), (ptr), (bytes)); if (es_plan_dry()) return 0; } } while (0)
#define ES_PLAN_DRY_RETURN() do { if (es_plan_dry()) return 0; } while (0)

// library-internal (builder.hip): rewrite every non-null pointer field of every recorded call through `map`
struct es_plan;
struct es_ctx;
int es_plan_relocate(es_plan* p, unsigned long long (*map)(unsigned long long addr, int use, void* user), void* user);
void es_ctx_adopt_arena(es_ctx* c, void* arena, size_t bytes, bool on_host);
void es_ctx_add_extent(es_ctx* c, unsigned long long off, unsigned long long bytes);   // persistent data inside the arena (es_ctx_save)
