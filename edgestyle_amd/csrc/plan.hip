// Native execution of the denoising path: recorded launch lists (es_plan) and the context that owns them (es_ctx).
//
// The reference's step function is OnnxUNetAndControlnets.forward (export_onnx.py:43-74) and its loop is
// EdgeStyleStableDiffusionControlNetPipeline.__call__ (model/edgestyle_pipeline.py:435-557); both are Python.  Here the
// host language only BUILDS a context (packs weights, allocates the static buffers and walks the model once while a
// plan records every C-ABI launch of that walk).  After that es_denoise_step / es_denoise_loop / es_vae_decode run with
// plain device pointers from any host: inputs are copied into the context's static buffers, the launch list is re-issued
// on the caller's stream (or replayed as a hipGraph instantiated from it), outputs are copied out.  No interpreter, no
// torch, no allocation in these calls.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <string>
#include <utility>
#include <vector>

#include "../../include/edgestyle_hip.h"
#include "plan.h"

extern "C" void es_set_error(const char* msg);

struct es_plan {
  struct Op { int kind; size_t off, bytes; };
  std::vector<Op> ops;
  std::vector<char> blob;
};

namespace {
thread_local es_plan* g_rec = nullptr;
thread_local bool g_replaying = false;
thread_local bool g_dry = false;

// what a context may change about a recorded list when it re-issues it: the CFG scale of the scheduler call, and the scheduler
// itself (the recorded call is the DDIM update; a context switched to UniPC issues es_cfg_unipc_step on its state slots instead)
struct RunOpts { const float* guidance = nullptr; const es_ctx* unipc = nullptr; };
int unipc_step(const es_ctx* c, const es_op_cfg_ddim* r, float guidance, void* s);

int launch_op(const es_plan* p, const es_plan::Op& op, hipStream_t st, const RunOpts& ro) {
  const float* guidance_override = ro.guidance;
  const char* a = p->blob.data() + op.off;
  void* s = (void*)st;
  switch (op.kind) {
    case ES_OP_CONV_GEMM: return es_conv_gemm((const es_gemm_desc*)a, s);
    case ES_OP_LINEAR_XS: return es_linear_xs((const es_xs_desc*)a, s);
    case ES_OP_ATTENTION: return es_attention((const es_attn_desc*)a, s);
    case ES_OP_GROUP_NORM: return es_group_norm((const es_gn_desc*)a, s);
    case ES_OP_LAYER_NORM: { auto* r = (const es_op_layer_norm*)a; return es_layer_norm(r->x, r->out, r->gamma, r->beta, r->M, r->C, r->eps, r->dtype, s); }
    case ES_OP_LAYER_NORM_GROUPED: return es_layer_norm_grouped((const es_ln_desc*)a, s);
    case ES_OP_FUSION_BLOCK: return es_fusion_block((const es_fusion_desc*)a, s);
    case ES_OP_FUSION_BLOCKS: return es_fusion_blocks((const es_fusion_desc*)a, (int)(op.bytes / sizeof(es_fusion_desc)), s);
    case ES_OP_TIMESTEP_EMBEDDING: { auto* r = (const es_op_timestep*)a; return es_timestep_embedding(r->t, r->out, r->N, r->dim, r->dtype, s); }
    case ES_OP_CFG_DDIM: { auto* r = (const es_op_cfg_ddim*)a;
      if (ro.unipc) return unipc_step(ro.unipc, r, guidance_override ? *guidance_override : r->guidance_scale, s);
      return es_cfg_ddim_step(r->noise, r->latents, r->model_in, r->coef, r->step_idx, guidance_override ? *guidance_override : r->guidance_scale,
                              r->B, r->HW, r->L, r->Lstride, r->cfg, r->nsteps, r->dtype, s); }
    case ES_OP_CFG_UNIPC: { auto* r = (const es_op_cfg_unipc*)a;
      return es_cfg_unipc_step(r->noise, r->latents, r->last_sample, r->m0, r->m1, r->model_in, r->coef, r->step_idx,
                               guidance_override ? *guidance_override : r->guidance_scale, r->B, r->HW, r->L, r->Lstride, r->cfg, r->nsteps, r->dtype, s); }
    case ES_OP_NCHW_TO_NHWC: { auto* r = (const es_op_nchw_to_nhwc*)a; return es_nchw_f32_to_nhwc(r->in, r->out, r->N, r->C, r->HW, r->Cpad, r->dtype, s); }
    case ES_OP_NHWC_TO_NCHW: { auto* r = (const es_op_nhwc_to_nchw*)a; return es_nhwc_to_nchw_f32(r->in, r->out, r->N, r->C, r->HW, r->Cstride, r->scale, r->shift, r->clamp01, r->dtype, s); }
    case ES_OP_ADD: { auto* r = (const es_op_add*)a; return es_add(r->a, r->b, r->y, r->n, r->dtype, s); }
    case ES_OP_VAE_SAMPLE: { auto* r = (const es_op_vae_sample*)a; return es_vae_sample(r->moments, r->noise, r->z, r->N, r->HW, r->L, r->Lpad, r->scaling, r->dtype, s); }
    case ES_OP_INCR: { auto* r = (const es_op_incr*)a; return es_incr(r->ctr, s); }
    case ES_OP_GATHER_ROW: { auto* r = (const es_op_gather_row*)a; return es_gather_row(r->table, r->idx, r->out, r->row_len, r->nrows, s); }
    case ES_OP_MEMCPY: { auto* r = (const es_op_memcpy*)a; return es_memcpy(r->dst, r->src, r->bytes, s); }
    case ES_OP_MEMCPY2D: { auto* r = (const es_op_memcpy2d*)a; return es_memcpy2d(r->dst, r->dpitch, r->src, r->spitch, r->width, r->height, s); }
    case ES_OP_FILL_F32: { auto* r = (const es_op_fill_f32*)a; return es_fill_f32(r->dst, r->value, r->n, s); }
    case ES_OP_LATENTS_TO_INPUT: { auto* r = (const es_op_latents_to_input*)a; return es_latents_to_input(r->latents, r->model_in, r->B, r->HW, r->L, r->Lstride, r->cfg, r->dtype, s); }
    default: es_set_error("es_plan_launch: unknown op"); return -1;
  }
}

int run_plan(const es_plan* p, hipStream_t st, const RunOpts& ro) {
  if (g_rec) { es_set_error("es_plan_launch: a plan is recording on this thread"); return -1; }
  g_replaying = true;
  int rc = 0;
  for (const auto& op : p->ops)
    if ((rc = launch_op(p, op, st, ro)) != 0) break;
  g_replaying = false;
  return rc;
}
}  // namespace

extern "C" int es_plan_recording(void) { return g_rec != nullptr && !g_replaying; }
extern "C" void es_plan_record(int kind, const void* args, size_t bytes) {
  es_plan* p = g_rec;
  const size_t off = (p->blob.size() + 15) & ~(size_t)15;
  p->blob.resize(off + bytes);
  memcpy(p->blob.data() + off, args, bytes);
  p->ops.push_back({kind, off, bytes});
}

extern "C" int es_plan_dry(void) { return g_dry && g_rec != nullptr && !g_replaying; }
extern "C" int es_plan_set_dry(int on) { const int was = g_dry; g_dry = on != 0; return was; }
extern "C" es_plan* es_plan_create(void) { return new es_plan(); }
extern "C" void es_plan_destroy(es_plan* p) { if (g_rec == p) g_rec = nullptr; delete p; }
extern "C" int es_plan_begin_record(es_plan* p) {
  if (!p || g_rec) { es_set_error("es_plan_begin_record: null plan, or another plan is recording on this thread"); return -1; }
  g_rec = p;
  return 0;
}
extern "C" int es_plan_end_record(es_plan* p) {
  if (!p || g_rec != p) { es_set_error("es_plan_end_record: this plan is not recording"); return -1; }
  g_rec = nullptr;
  return 0;
}
extern "C" int es_plan_size(const es_plan* p) { return p ? (int)p->ops.size() : -1; }
extern "C" int es_plan_count(const es_plan* p, int kind) {
  if (!p) return -1;
  int n = 0;
  for (const auto& op : p->ops) n += op.kind == kind;
  return n;
}
extern "C" int es_plan_launch(const es_plan* p, void* stream) {
  if (!p) { es_set_error("es_plan_launch: null plan"); return -1; }
  return run_plan(p, (hipStream_t)stream, RunOpts{});
}

// Pointer fields of a recorded call, by op kind: byte offset inside the argument record and how the op uses the memory
// behind it (1 = reads, 2 = writes, 3 = both).  This is what a relocator needs (edgestyle_amd/native.py save()): exactly
// these 8-byte words are device addresses - nothing else in a record is ever treated as one - and a block of memory that
// every plan only READS holds persistent data (packed weights, parameters, tables) and must travel with a context image,
// while blocks that some call writes are produced at run time and only need their space.  `elem_bytes`: records of the
// array ops (es_fusion_blocks) repeat every elem_bytes; 0 = one record.  Returns the field count (fills at most cap).
#include <stddef.h>
namespace {
struct PtrField { int off; int use; };
#define F_IN(S, m) {(int)offsetof(S, m), 1}
#define F_OUT(S, m) {(int)offsetof(S, m), 2}
#define F_IO(S, m) {(int)offsetof(S, m), 3}
#define F_IN4(S, m) {(int)offsetof(S, m), 1}, {(int)offsetof(S, m) + 8, 1}, {(int)offsetof(S, m) + 16, 1}, {(int)offsetof(S, m) + 24, 1}
const std::vector<PtrField>& ptr_fields(int kind, int& elem) {
  static const std::vector<PtrField> none;
  elem = 0;
  switch (kind) {
    case ES_OP_CONV_GEMM: { static const std::vector<PtrField> f = {
        F_IN(es_gemm_desc, x), F_IN(es_gemm_desc, x2), F_IN(es_gemm_desc, w), F_IN(es_gemm_desc, bias), F_IN(es_gemm_desc, temb),
        F_IN(es_gemm_desc, residual), F_IN(es_gemm_desc, out_scale_dev), F_OUT(es_gemm_desc, out), F_IO(es_gemm_desc, workspace),
        F_OUT(es_gemm_desc, prof), F_IN4(es_gemm_desc, w_g), F_IN4(es_gemm_desc, bias_g), F_IN(es_gemm_desc, ln_colsum),
        F_IN4(es_gemm_desc, ln_colsum_g), F_IN(es_gemm_desc, t1), F_IN(es_gemm_desc, t2), F_IN(es_gemm_desc, residual_lo),
        F_OUT(es_gemm_desc, out_lo), F_OUT(es_gemm_desc, gn_part)}; return f; }
    case ES_OP_LINEAR_XS: { static const std::vector<PtrField> f = {
        F_IN(es_xs_desc, x), F_OUT(es_xs_desc, out), F_IN(es_xs_desc, w), F_IN(es_xs_desc, bias), F_IN4(es_xs_desc, w_g),
        F_IN4(es_xs_desc, bias_g), F_OUT(es_xs_desc, prof), F_IN(es_xs_desc, residual), F_IN(es_xs_desc, gn_part), F_IN(es_xs_desc, gn_gamma),
        F_IN(es_xs_desc, gn_beta), F_IN4(es_xs_desc, gn_gamma_g), F_IN4(es_xs_desc, gn_beta_g)}; return f; }
    case ES_OP_ATTENTION: { static const std::vector<PtrField> f = {
        F_IN(es_attn_desc, q), F_IN(es_attn_desc, k), F_IN(es_attn_desc, v), F_OUT(es_attn_desc, o)}; return f; }
    case ES_OP_GROUP_NORM: { static const std::vector<PtrField> f = {
        F_IN(es_gn_desc, x), F_IN(es_gn_desc, x2), F_OUT(es_gn_desc, out), F_IN(es_gn_desc, gamma), F_IN(es_gn_desc, beta),
        F_IO(es_gn_desc, partials), F_IN4(es_gn_desc, gamma_g), F_IN4(es_gn_desc, beta_g)}; return f; }
    case ES_OP_LAYER_NORM: { static const std::vector<PtrField> f = {
        F_IN(es_op_layer_norm, x), F_OUT(es_op_layer_norm, out), F_IN(es_op_layer_norm, gamma), F_IN(es_op_layer_norm, beta)}; return f; }
    case ES_OP_LAYER_NORM_GROUPED: { static const std::vector<PtrField> f = {
        F_IN(es_ln_desc, x), F_OUT(es_ln_desc, out), F_IN4(es_ln_desc, gamma_g), F_IN4(es_ln_desc, beta_g)}; return f; }
    case ES_OP_FUSION_BLOCKS: elem = (int)sizeof(es_fusion_desc);   // same record layout, repeated
      [[fallthrough]];
    case ES_OP_FUSION_BLOCK: { static const std::vector<PtrField> f = {
        F_IN4(es_fusion_desc, res), {(int)offsetof(es_fusion_desc, res) + 32, 1}, {(int)offsetof(es_fusion_desc, res) + 40, 1},
        F_IN(es_fusion_desc, w1), F_IN(es_fusion_desc, b1), F_IN(es_fusion_desc, g1), F_IN(es_fusion_desc, be1),
        F_IN(es_fusion_desc, w2), F_IN(es_fusion_desc, b2), F_IN(es_fusion_desc, g2), F_IN(es_fusion_desc, be2),
        F_IN(es_fusion_desc, w3), F_IN(es_fusion_desc, b3), F_IO(es_fusion_desc, scratch), F_IO(es_fusion_desc, u),
        F_OUT(es_fusion_desc, out), F_IN(es_fusion_desc, res_scale_dev), F_IN(es_fusion_desc, addend)}; return f; }
    case ES_OP_TIMESTEP_EMBEDDING: { static const std::vector<PtrField> f = {F_IN(es_op_timestep, t), F_OUT(es_op_timestep, out)}; return f; }
    case ES_OP_CFG_DDIM: { static const std::vector<PtrField> f = {
        F_IN(es_op_cfg_ddim, noise), F_IO(es_op_cfg_ddim, latents), F_OUT(es_op_cfg_ddim, model_in), F_IN(es_op_cfg_ddim, coef),
        F_IN(es_op_cfg_ddim, step_idx)}; return f; }
    case ES_OP_CFG_UNIPC: { static const std::vector<PtrField> f = {
        F_IN(es_op_cfg_unipc, noise), F_IO(es_op_cfg_unipc, latents), F_IO(es_op_cfg_unipc, last_sample), F_IO(es_op_cfg_unipc, m0),
        F_IO(es_op_cfg_unipc, m1), F_OUT(es_op_cfg_unipc, model_in), F_IN(es_op_cfg_unipc, coef), F_IN(es_op_cfg_unipc, step_idx)}; return f; }
    case ES_OP_NCHW_TO_NHWC: { static const std::vector<PtrField> f = {F_IN(es_op_nchw_to_nhwc, in), F_OUT(es_op_nchw_to_nhwc, out)}; return f; }
    case ES_OP_NHWC_TO_NCHW: { static const std::vector<PtrField> f = {F_IN(es_op_nhwc_to_nchw, in), F_OUT(es_op_nhwc_to_nchw, out)}; return f; }
    case ES_OP_ADD: { static const std::vector<PtrField> f = {F_IN(es_op_add, a), F_IN(es_op_add, b), F_OUT(es_op_add, y)}; return f; }
    case ES_OP_VAE_SAMPLE: { static const std::vector<PtrField> f = {F_IN(es_op_vae_sample, moments), F_IN(es_op_vae_sample, noise), F_OUT(es_op_vae_sample, z)}; return f; }
    case ES_OP_INCR: { static const std::vector<PtrField> f = {F_IO(es_op_incr, ctr)}; return f; }
    case ES_OP_GATHER_ROW: { static const std::vector<PtrField> f = {F_IN(es_op_gather_row, table), F_IN(es_op_gather_row, idx), F_OUT(es_op_gather_row, out)}; return f; }
    case ES_OP_MEMCPY: { static const std::vector<PtrField> f = {F_OUT(es_op_memcpy, dst), F_IN(es_op_memcpy, src)}; return f; }
    case ES_OP_MEMCPY2D: { static const std::vector<PtrField> f = {F_OUT(es_op_memcpy2d, dst), F_IN(es_op_memcpy2d, src)}; return f; }
    case ES_OP_FILL_F32: { static const std::vector<PtrField> f = {F_OUT(es_op_fill_f32, dst)}; return f; }
    case ES_OP_LATENTS_TO_INPUT: { static const std::vector<PtrField> f = {F_IN(es_op_latents_to_input, latents), F_OUT(es_op_latents_to_input, model_in)}; return f; }
    default: return none;
  }
}
#undef F_IN
#undef F_OUT
#undef F_IO
#undef F_IN4
}  // namespace
extern "C" int es_plan_pointer_fields(int kind, int32_t* offsets, int32_t* uses, int cap, int32_t* elem_bytes) {
  int elem = 0;
  const auto& f = ptr_fields(kind, elem);
  if (elem_bytes) *elem_bytes = elem;
  for (int i = 0; i < (int)f.size() && i < cap; ++i) {
    if (offsets) offsets[i] = f[i].off;
    if (uses) uses[i] = f[i].use;
  }
  return (int)f.size();
}

int es_plan_relocate(es_plan* p, unsigned long long (*map)(unsigned long long, int, void*), void* user) {
  for (const auto& op : p->ops) {
    int elem = 0;
    const auto& f = ptr_fields(op.kind, elem);
    const size_t reps = elem ? op.bytes / (size_t)elem : 1;
    for (size_t r = 0; r < reps; ++r)
      for (const auto& pf : f) {
        const size_t pos = op.off + r * (size_t)elem + (size_t)pf.off;
        if (pos + 8 > op.off + op.bytes) { es_set_error("es_plan_relocate: pointer field outside its record"); return -1; }
        unsigned long long a;
        memcpy(&a, p->blob.data() + pos, 8);
        if (!a) continue;
        a = map(a, pf.use, user);
        memcpy(p->blob.data() + pos, &a, 8);
      }
  }
  return 0;
}
// Flat image of a plan (es_ctx_save / es_ctx_load): u64 n_ops, u64 blob bytes, n_ops x {i64 kind, u64 off, u64 bytes}, blob.
// The blob still holds the RECORDED device addresses: whoever moves it relocates them (edgestyle_amd/native.py save()).
extern "C" size_t es_plan_export(const es_plan* p, void* out, size_t cap) {
  if (!p) return 0;
  const size_t need = 16 + p->ops.size() * 24 + p->blob.size();
  if (!out || cap < need) return need;
  unsigned long long* w = (unsigned long long*)out;
  w[0] = p->ops.size(); w[1] = p->blob.size();
  size_t k = 2;
  for (const auto& op : p->ops) { w[k++] = (unsigned long long)(long long)op.kind; w[k++] = op.off; w[k++] = op.bytes; }
  if (!p->blob.empty()) memcpy(w + k, p->blob.data(), p->blob.size());
  return need;
}
extern "C" es_plan* es_plan_import(const void* data, size_t size) {
  if (!data || size < 16) { es_set_error("es_plan_import: truncated image"); return nullptr; }
  const unsigned long long* w = (const unsigned long long*)data;
  const size_t n = (size_t)w[0], nb = (size_t)w[1];
  if (n > (1u << 24) || size < 16 + n * 24 + nb) { es_set_error("es_plan_import: truncated image"); return nullptr; }
  es_plan* p = new es_plan();
  p->ops.resize(n);
  for (size_t i = 0; i < n; ++i) {
    p->ops[i] = {(int)(long long)w[2 + 3 * i], (size_t)w[3 + 3 * i], (size_t)w[4 + 3 * i]};
    if (p->ops[i].off + p->ops[i].bytes > nb) { delete p; es_set_error("es_plan_import: op outside the blob"); return nullptr; }
  }
  p->blob.assign((const char*)(w + 2 + 3 * n), (const char*)(w + 2 + 3 * n) + nb);
  return p;
}

// ---------------------------------------------------------------------------------------------------------------------
struct es_ctx {
  int device = 0;
  es_plan* plan[ES_PLAN_COUNT] = {};
  hipGraphExec_t exec[ES_PLAN_COUNT] = {};
  float exec_guidance[ES_PLAN_COUNT] = {};
  void* buf[ES_BUF_COUNT] = {};
  size_t bytes[ES_BUF_COUNT] = {};
  es_ctx_geometry g = {};
  float cond_scales[6] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f};
  float control_start = 0.f, control_end = 1.f;
  int use_graphs = 1;
  hipStream_t cap_stream = nullptr; // plans are captured into graphs on a stream of the context's own (the caller's may
                                    // be the legacy default stream, which cannot capture); the graphs launch on the caller's
  float* host = nullptr;            // PINNED staging for the per-call tables (hipHostMalloc): the H2D copies are truly
  size_t host_cap = 0;              // asynchronous, so the buffer is only rewritten after `staged` - recorded behind the
  hipEvent_t staged = nullptr;      // previous call's copies - has completed
  int scheduler = ES_SCHED_DDIM;       // es_ctx_set_scheduler
  std::vector<double> alphas_cumprod_f64;   // UniPC derives its sigmas in double (es_ctx_set_alphas_cumprod_f64; SD1.5 default otherwise)
  std::vector<float> alphas_cumprod;   // the scheduler's schedule (es_ctx_set_alphas_cumprod; SD1.5 default otherwise)
  void* arena = nullptr;               // es_ctx_load / es_load_weights: the one allocation every recorded pointer was relocated into
  size_t arena_bytes = 0;
  bool arena_on_host = false;
  bool launchable = true;              // false: an inspection build (es_load_weights device -1 / -2) whose recorded addresses are not device memory
  std::vector<std::pair<unsigned long long, unsigned long long>> extents;   // (offset, bytes) of the arena's persistent DATA (es_ctx_save)          // es_load_weights(device -2): an inspection build in host memory
  hipGraphExec_t loop_exec = nullptr;  // use_graphs == 2: preparation + all steps of es_denoise_loop as one graph
  int loop_steps = 0;
  float loop_guidance = 0.f;
  float loop_window[2] = {0.f, 1.f};   // the control-guidance window the loop graph was captured for (it decides which steps are UNet-only)
};

namespace {
// pinned staging of at least n floats whose previous contents the device no longer reads
float* staging(es_ctx* c, size_t n) {
  if (c->staged && hipEventSynchronize(c->staged) != hipSuccess) return nullptr;
  if (n > c->host_cap) {
    if (c->host) (void)hipHostFree(c->host);
    c->host = nullptr;
    c->host_cap = 0;
    if (hipHostMalloc((void**)&c->host, n * sizeof(float), hipHostMallocDefault) != hipSuccess) return nullptr;
    c->host_cap = n;
  }
  if (!c->staged && hipEventCreateWithFlags(&c->staged, hipEventDisableTiming) != hipSuccess) return nullptr;
  return c->host;
}
bool capturing(hipStream_t st) {
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  return hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone;
}
// the dry (-1) and host-arena (-2) builds of es_load_weights bind every slot and relocate every plan to addresses that are NOT
// device memory: launching them would fault on the GPU, so every entry point that launches refuses them
int can_launch(const es_ctx* c, const char* who) {
  if (c->launchable) return 0;
  static thread_local char msg[192];
  snprintf(msg, sizeof(msg), "%s: this context is an inspection build (es_load_weights device -1 / -2): its plans hold no device addresses and cannot be launched", who);
  es_set_error(msg);
  return -1;
}
int need(const es_ctx* c, int slot, const char* what) {
  if (!c->buf[slot]) { es_set_error(what); return -1; }
  return 0;
}
int d2d(void* dst, const void* src, size_t n, hipStream_t st) {
  if (dst == src || !n) return 0;
  if (hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToDevice, st) != hipSuccess) { es_set_error("es_ctx: device copy failed"); return -2; }
  return 0;
}
int h2d(void* dst, const void* src, size_t n, hipStream_t st) {
  if (hipMemcpyAsync(dst, src, n, hipMemcpyHostToDevice, st) != hipSuccess) { es_set_error("es_ctx: host-to-device copy failed"); return -2; }
  return 0;
}

// one plan: as a hipGraph captured from the launch list (re-captured when the guidance scale baked into its scheduler
// node changes), or - use_graphs 0, or while the caller's stream is itself capturing - re-issued launch by launch
// ES_CTX_TRACE=1: one line on stderr ahead of every HIP graph call of a context (which call a host-side fault sits in when the
// process runs under a tool that intercepts the HIP / HSA layers)
bool ctx_trace() { static const bool on = [] { const char* e = getenv("ES_CTX_TRACE"); return e && e[0] == '1'; }(); return on; }
#define ES_TRACE(...) do { if (ctx_trace()) { fprintf(stderr, "[es_ctx] " __VA_ARGS__); fputc('\n', stderr); fflush(stderr); } } while (0)

// A profiler that intercepts the HSA queues (rocprofv3 = rocprofiler-sdk) rewrites AQL packets; the HIP runtime's graph packet
// capture block-copies packets it pre-built at instantiate time, and hipGraphLaunch of a context's SECOND kind of graph then
// faults inside the runtime (host SIGSEGV, profiles/r04_rocprof_graph_fault.txt).  DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 makes the
// runtime enqueue graph nodes one by one and is what every profiling script of tools/ exports; a caller who profiles WITHOUT it
// must not reach the faulting path: the context then replays its plans launch by launch (same kernels, same order).
bool graph_hazard() {
  static const bool v = [] {
    const char* cap = getenv("DEBUG_CLR_GRAPH_PACKET_CAPTURE");
    if (cap && cap[0] == '0') return false;
    const char* tl = getenv("ROCP_TOOL_LIBRARIES");
    const char* pre = getenv("LD_PRELOAD");
    bool prof = (tl && *tl) || (pre && strstr(pre, "rocprof"));
    if (!prof && dlsym(RTLD_DEFAULT, "rocprofiler_configure")) prof = true;
    if (prof) fprintf(stderr, "[edgestyle_hip] a queue-intercepting profiler is attached and DEBUG_CLR_GRAPH_PACKET_CAPTURE is not 0: "
                              "the context replays its plans launch by launch instead of as hipGraphs (export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 "
                              "to profile the graph path; profiles/r04_rocprof_graph_fault.txt)\n");
    return prof;
  }();
  return v;
}

int run(es_ctx* c, int which, hipStream_t st, const float* guidance) {
  es_plan* p = c->plan[which];
  if (!p) { es_set_error("es_ctx: plan not set"); return -1; }
  if (can_launch(c, "es_ctx")) return -1;
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  (void)hipStreamIsCapturing(st, &cs);
  RunOpts ro;
  ro.guidance = guidance;
  ro.unipc = c->scheduler == ES_SCHED_UNIPC ? c : nullptr;
  if (!c->use_graphs || cs != hipStreamCaptureStatusNone || graph_hazard()) return run_plan(p, st, ro);
  const float gs = guidance ? *guidance : 0.f;
  if (c->exec[which] && c->exec_guidance[which] != gs) { (void)hipGraphExecDestroy(c->exec[which]); c->exec[which] = nullptr; }
  if (!c->exec[which]) {
    hipGraph_t graph = nullptr;
    if (!c->cap_stream && hipStreamCreateWithFlags(&c->cap_stream, hipStreamNonBlocking) != hipSuccess) { es_set_error("es_ctx: hipStreamCreate failed"); return -2; }
    ES_TRACE("plan %d: hipStreamBeginCapture (%zu calls)", which, p->ops.size());
    if (hipStreamBeginCapture(c->cap_stream, hipStreamCaptureModeThreadLocal) != hipSuccess) { es_set_error("es_ctx: hipStreamBeginCapture failed"); return -2; }
    const int rc = run_plan(p, c->cap_stream, ro);
    ES_TRACE("plan %d: hipStreamEndCapture", which);
    const hipError_t e = hipStreamEndCapture(c->cap_stream, &graph);
    if (rc || e != hipSuccess || !graph) { if (graph) (void)hipGraphDestroy(graph); if (!rc) es_set_error("es_ctx: hipStreamEndCapture failed"); return rc ? rc : -2; }
    ES_TRACE("plan %d: hipGraphInstantiate", which);
    const hipError_t ei = hipGraphInstantiate(&c->exec[which], graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (ei != hipSuccess) { c->exec[which] = nullptr; es_set_error("es_ctx: hipGraphInstantiate failed"); return -2; }
    c->exec_guidance[which] = gs;
  }
  ES_TRACE("plan %d: hipGraphLaunch", which);
  if (hipGraphLaunch(c->exec[which], st) != hipSuccess) { es_set_error("es_ctx: hipGraphLaunch failed"); return -2; }
  ES_TRACE("plan %d: hipGraphLaunch returned", which);
  return 0;
}

// DDIMScheduler of the SD1.5 checkpoints (diffusers scheduler_config: scaled_linear betas 0.00085..0.012, 1000 train
// steps, set_alpha_to_one False, steps_offset 1, eta 0): {sqrt(a_t), sqrt(1-a_t), sqrt(a_prev), sqrt(1-a_prev)} per step,
// a_prev = alphas_cumprod of the NEXT timestep of the list (t - 1000 / n), alphas_cumprod[0] after the last one
void default_alphas(std::vector<float>& ac) {
  // torch.linspace(sqrt(b0), sqrt(b1), 1000, fp32) ** 2 -> cumprod(1 - betas), with torch's fp32 rounding points
  // (linspace fills the second half from the end), so that the default table equals the host scheduler's bit for bit
  ac.resize(1000);
  volatile float b0 = (float)sqrt(0.00085), b1 = (float)sqrt(0.012);
  volatile float step = (b1 - b0) / 999.0f;
  volatile float prod = 1.0f;
  for (int i = 0; i < 1000; ++i) {
    volatile float m = i < 500 ? step * (float)i : step * (float)(999 - i);
    volatile float b = i < 500 ? b0 + m : b1 - m;
    volatile float b2 = b * b;
    volatile float om = 1.0f - b2;
    prod = prod * om;
    ac[i] = prod;
  }
}
void ddim_coef_from(const std::vector<float>& ac, const float* ts, int n, float* out) {
  const int last = (int)ac.size() - 1;
  auto at = [&](float t) { int i = (int)t; i = i < 0 ? 0 : (i > last ? last : i); return ac[i]; };
  for (int i = 0; i < n; ++i) {
    // volatile: every value is rounded to fp32 exactly where the host scheduler (torch fp32 tensors) rounds it
    volatile float a_t = at(ts[i]);
    volatile float a_p = i + 1 < n ? at(ts[i + 1]) : ac[0];
    volatile float b_t = 1.0f - a_t, b_p = 1.0f - a_p;
    out[i * 4 + 0] = sqrtf(a_t); out[i * 4 + 1] = sqrtf(b_t);
    out[i * 4 + 2] = sqrtf(a_p); out[i * 4 + 3] = sqrtf(b_p);
  }
}
void ddim_coef(es_ctx* c, const float* ts, int n, float* out) {
  if (c->alphas_cumprod.empty()) default_alphas(c->alphas_cumprod);
  ddim_coef_from(c->alphas_cumprod, ts, n, out);
}

// UniPCMultistepScheduler (the scheduler the reference's callers assign: TT:273, APP:118) with the SD1.5 scheduler config it
// inherits - bh2, solver_order <= 2, predict_x0, lower_order_final, epsilon prediction.  Every update of the multistep
// predictor / corrector is linear in {last_sample, m0, m1, x0}: per step 12 scalars {alpha_t, sigma_t, use_c, c_last, c_m0, c_m1,
// c_x0, p_x, p_x0, p_m0, 0, 0}, derived here in double exactly as edgestyle_amd/schedulers.py does (same formulas, same order of
// operations; the 2 x 2 system of the second-order corrector is solved in closed form), consumed by es_cfg_unipc_step.
void default_alphas_f64(std::vector<double>& ac) {
  // torch.linspace(sqrt(b0), sqrt(b1), 1000, float64) ** 2 -> cumprod(1 - betas)
  ac.resize(1000);
  const double b0 = sqrt(0.00085), b1 = sqrt(0.012), step = (b1 - b0) / 999.0;
  double prod = 1.0;
  for (int i = 0; i < 1000; ++i) {
    const double b = i < 500 ? b0 + step * (double)i : b1 - step * (double)(999 - i);
    prod *= 1.0 - b * b;
    ac[i] = prod;
  }
}
void unipc_coef_from(const std::vector<double>& ac, const float* ts, int T, int solver_order, float* out) {
  const int n = (int)ac.size();
  auto sig_at = [&](double t) {                       // np.interp(t, arange(n), sqrt((1 - ac) / ac))
    auto sg = [&](int i) { return sqrt((1.0 - ac[i]) / ac[i]); };
    if (t <= 0) return sg(0);
    if (t >= n - 1) return sg(n - 1);
    const int i = (int)t;
    const double f = t - (double)i;
    return f == 0.0 ? sg(i) : sg(i) + (sg(i + 1) - sg(i)) * f;
  };
  std::vector<double> sigmas((size_t)T + 1);
  for (int i = 0; i < T; ++i) sigmas[i] = sig_at((double)ts[i]);
  sigmas[T] = sqrt((1.0 - ac[0]) / ac[0]);
  struct Asl { double a, s, l; };
  auto asl = [&](int idx) { const double s = sigmas[idx]; const double a = 1.0 / sqrt(s * s + 1.0); return Asl{a, s * a, log(a) - log(s * a)}; };
  // rhos of the B(h) = e^h - 1 variant: corrector (order 1: 1/2; order 2: 2 x 2 solve), predictor (order 2: 1/2)
  auto rhos = [&](const double* rks, double hh, int order, bool corrector, double* rho, double& h_phi_1, double& B_h) {
    h_phi_1 = expm1(hh);
    double h_phi_k = h_phi_1 / hh - 1.0;
    B_h = expm1(hh);
    double fact = 1.0, b[2] = {0, 0};
    for (int i = 1; i <= order; ++i) {
      b[i - 1] = h_phi_k * fact / B_h;
      fact *= (double)(i + 1);
      h_phi_k = h_phi_k / hh - 1.0 / fact;
    }
    rho[0] = rho[1] = 0.0;
    if (corrector) {
      if (order == 1) rho[0] = 0.5;
      else { rho[0] = (b[0] - b[1]) / (1.0 - rks[0]); rho[1] = b[0] - rho[0]; }      // [[1, 1], [rk0, 1]] rho = b
    } else if (order == 2) rho[0] = 0.5;
  };
  int lower_order_nums = 0, this_order = 1;
  for (int i = 0; i < T; ++i) {
    const Asl ci = asl(i);
    double row[12] = {ci.a, ci.s, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (i > 0) {                                      // corrector with the order of the previous predictor
      const int order = this_order;
      const Asl c0 = asl(i - 1);
      const double h = ci.l - c0.l;
      double rks[2] = {1.0, 1.0};
      if (order == 2) rks[0] = (asl(i - 2).l - c0.l) / h;
      double rho[2], h_phi_1, B_h;
      rhos(rks, -h, order, true, rho, h_phi_1, B_h);
      const double A = ci.s / c0.s, Hh = ci.a * h_phi_1, Bc = ci.a * B_h;
      row[2] = 1.0; row[3] = A;
      if (order == 1) { row[4] = -Hh + Bc * rho[0]; row[5] = 0.0; row[6] = -Bc * rho[0]; }
      else { row[4] = -Hh + Bc * (rho[0] / rks[0] + rho[1]); row[5] = -Bc * rho[0] / rks[0]; row[6] = -Bc * rho[1]; }
    }
    this_order = std::min(std::min(solver_order, T - i), lower_order_nums + 1);        // lower_order_final + warm-up
    const Asl ct = asl(i + 1);
    const double h = ct.l - ci.l;
    double rks[2] = {1.0, 1.0};
    if (this_order == 2) rks[0] = (asl(i - 1).l - ci.l) / h;
    double rho[2], h_phi_1, B_h;
    rhos(rks, -h, this_order, false, rho, h_phi_1, B_h);
    const double A = ct.s / ci.s, Hh = ct.a * h_phi_1, Bp = ct.a * B_h;
    row[7] = A;
    if (this_order == 1) { row[8] = -Hh; row[9] = 0.0; }
    else { row[8] = -Hh + Bp * rho[0] / rks[0]; row[9] = -Bp * rho[0] / rks[0]; }
    if (lower_order_nums < solver_order) ++lower_order_nums;
    for (int k = 0; k < 12; ++k) out[i * 12 + k] = (float)row[k];
  }
}
void unipc_coef(const es_ctx* c, const float* ts, int T, float* out) {
  std::vector<double> ac = c->alphas_cumprod_f64;
  if (ac.empty()) default_alphas_f64(ac);
  unipc_coef_from(ac, ts, T, 2, out);
}
// the scheduler call of a recorded step, re-issued as the UniPC update on the context's state slots
int unipc_step(const es_ctx* c, const es_op_cfg_ddim* r, float guidance, void* s) {
  return es_cfg_unipc_step(r->noise, r->latents, (float*)c->buf[ES_BUF_HIST0], (float*)c->buf[ES_BUF_HIST1], (float*)c->buf[ES_BUF_HIST2],
                           r->model_in, (const float*)c->buf[ES_BUF_COEF], r->step_idx, guidance, r->B, r->HW, r->L, r->Lstride, r->cfg,
                           r->nsteps, r->dtype, s);
}
}  // namespace

extern "C" int es_ctx_set_scheduler(es_ctx* c, int scheduler) {
  if (!c || (scheduler != ES_SCHED_DDIM && scheduler != ES_SCHED_UNIPC)) { es_set_error("es_ctx_set_scheduler: ES_SCHED_DDIM or ES_SCHED_UNIPC"); return -1; }
  if (scheduler == ES_SCHED_UNIPC) {
    for (int i = 0; i < 3; ++i)
      if (!c->buf[ES_BUF_HIST0 + i] || c->bytes[ES_BUF_HIST0 + i] != c->bytes[ES_BUF_LATENTS]) { es_set_error("es_ctx_set_scheduler: UniPC needs the three ES_BUF_HIST slots bound (fp32, the size of the latents)"); return -1; }
    if (c->bytes[ES_BUF_COEF] < (size_t)c->g.n_steps * 12 * sizeof(float)) { es_set_error("es_ctx_set_scheduler: UniPC needs ES_BUF_COEF of n_steps x 12 floats"); return -1; }
  }
  if (c->scheduler != scheduler) {                     // the scheduler node of every instantiated graph changes
    for (int i = 0; i < ES_PLAN_COUNT; ++i) if (c->exec[i]) { (void)hipGraphExecDestroy(c->exec[i]); c->exec[i] = nullptr; }
    if (c->loop_exec) { (void)hipGraphExecDestroy(c->loop_exec); c->loop_exec = nullptr; }
  }
  c->scheduler = scheduler;
  return 0;
}
/* host-only: the per-step UniPC coefficient rows es_denoise_loop derives from `timesteps` (alphas_cumprod NULL: SD1.5's schedule in double) */
extern "C" int es_unipc_coef_table(const double* alphas_cumprod, int n_alphas, const float* timesteps, int n, float* out) {
  if (!timesteps || !out || n < 1) { es_set_error("es_unipc_coef_table: bad arguments"); return -1; }
  std::vector<double> ac;
  if (alphas_cumprod && n_alphas > 0) ac.assign(alphas_cumprod, alphas_cumprod + n_alphas);
  else default_alphas_f64(ac);
  unipc_coef_from(ac, timesteps, n, 2, out);
  return 0;
}

extern "C" int es_ctx_create(int device, es_ctx** out) {
  if (!out) { es_set_error("es_ctx_create: null out"); return -1; }
  es_ctx* c = new es_ctx();
  c->device = device;
  *out = c;
  return 0;
}
extern "C" void es_ctx_destroy(es_ctx* c) {
  if (!c) return;
  for (int i = 0; i < ES_PLAN_COUNT; ++i) {
    if (c->exec[i]) (void)hipGraphExecDestroy(c->exec[i]);
    if (c->plan[i]) es_plan_destroy(c->plan[i]);
  }
  if (c->loop_exec) (void)hipGraphExecDestroy(c->loop_exec);
  if (c->cap_stream) (void)hipStreamDestroy(c->cap_stream);
  if (c->staged) { (void)hipEventSynchronize(c->staged); (void)hipEventDestroy(c->staged); }
  if (c->host) (void)hipHostFree(c->host);
  if (c->arena) { if (c->arena_on_host) free(c->arena); else (void)hipFree(c->arena); }
  delete c;
}
void es_ctx_adopt_arena(es_ctx* c, void* arena, size_t bytes, bool on_host) {
  c->arena = arena; c->arena_bytes = bytes; c->arena_on_host = on_host;
  c->launchable = arena != nullptr && !on_host;
}
void es_ctx_add_extent(es_ctx* c, unsigned long long off, unsigned long long bytes) { c->extents.emplace_back(off, bytes); }
extern "C" size_t es_ctx_arena_bytes(const es_ctx* c) { return c ? c->arena_bytes : 0; }
extern "C" int es_ctx_set_geometry(es_ctx* c, const es_ctx_geometry* g) {
  if (!c || !g || g->B < 1 || g->h < 1 || g->w < 1 || g->n_steps < 1 || g->n_conds < 1 || g->n_conds > 6) { es_set_error("es_ctx_set_geometry: bad geometry"); return -1; }
  c->g = *g;
  return 0;
}
extern "C" int es_ctx_set_plan(es_ctx* c, int which, es_plan* p) {
  if (!c || which < 0 || which >= ES_PLAN_COUNT || !p) { es_set_error("es_ctx_set_plan: bad arguments"); return -1; }
  if (c->exec[which]) { (void)hipGraphExecDestroy(c->exec[which]); c->exec[which] = nullptr; }
  if (c->loop_exec) { (void)hipGraphExecDestroy(c->loop_exec); c->loop_exec = nullptr; }
  if (c->plan[which] && c->plan[which] != p) es_plan_destroy(c->plan[which]);
  c->plan[which] = p;
  return 0;
}
extern "C" int es_ctx_bind(es_ctx* c, int slot, void* dev, size_t nbytes) {
  if (!c || slot < 0 || slot >= ES_BUF_COUNT || !dev || !nbytes) { es_set_error("es_ctx_bind: bad arguments"); return -1; }
  c->buf[slot] = dev;
  c->bytes[slot] = nbytes;
  return 0;
}
extern "C" void* es_ctx_buffer(const es_ctx* c, int slot, size_t* bytes) {      // borrowed: the slot's device memory
  if (!c || slot < 0 || slot >= ES_BUF_COUNT) return nullptr;
  if (bytes) *bytes = c->bytes[slot];
  return c->buf[slot];
}
extern "C" int es_ctx_graph_hazard(void) { return graph_hazard() ? 1 : 0; }

extern "C" int es_ctx_set_options(es_ctx* c, const float* cond_scales, float control_guidance_start, float control_guidance_end, int use_graphs) {
  if (!c) { es_set_error("es_ctx_set_options: null ctx"); return -1; }
  if (cond_scales) memcpy(c->cond_scales, cond_scales, sizeof(c->cond_scales));
  c->control_start = control_guidance_start;
  c->control_end = control_guidance_end;
  c->use_graphs = use_graphs;
  return 0;
}
extern "C" int es_ctx_set_alphas_cumprod(es_ctx* c, const float* alphas_cumprod, int n) {
  if (!c || !alphas_cumprod || n < 1) { es_set_error("es_ctx_set_alphas_cumprod: bad arguments"); return -1; }
  c->alphas_cumprod.assign(alphas_cumprod, alphas_cumprod + n);
  return 0;
}
extern "C" int es_ctx_set_alphas_cumprod_f64(es_ctx* c, const double* alphas_cumprod, int n) {
  if (!c || !alphas_cumprod || n < 1) { es_set_error("es_ctx_set_alphas_cumprod_f64: bad arguments"); return -1; }
  c->alphas_cumprod_f64.assign(alphas_cumprod, alphas_cumprod + n);
  return 0;
}
/* one plan of the context on `stream` (what es_denoise_loop does n_steps times with ES_PLAN_STEP): lets a host single-step
 * a loop it has prepared, e.g. to look at intermediate latents */
extern "C" int es_ctx_launch_plan(es_ctx* c, int which, const float* guidance_scale, void* stream) {
  if (!c || which < 0 || which >= ES_PLAN_COUNT) { es_set_error("es_ctx_launch_plan: bad arguments"); return -1; }
  if (can_launch(c, "es_ctx_launch_plan")) return -1;
  return run(c, which, (hipStream_t)stream, guidance_scale);
}
/* host-only: the per-step DDIM coefficient rows es_denoise_loop derives from `timesteps` (no GPU involved) */
extern "C" int es_ddim_coef_table(const float* alphas_cumprod, int n_alphas, const float* timesteps, int n, float* out) {
  if (!timesteps || !out || n < 1) { es_set_error("es_ddim_coef_table: bad arguments"); return -1; }
  std::vector<float> ac;
  if (alphas_cumprod && n_alphas > 0) ac.assign(alphas_cumprod, alphas_cumprod + n_alphas);
  else default_alphas(ac);
  ddim_coef_from(ac, timesteps, n, out);
  return 0;
}
extern "C" es_plan* es_ctx_plan(es_ctx* c, int which) {           // borrowed: the context keeps ownership
  return (c && which >= 0 && which < ES_PLAN_COUNT) ? c->plan[which] : nullptr;
}

// A context image written by edgestyle_amd/native.py NativeEngine.save(): geometry, options, the five plans with a
// relocation table each, the bound slots, and the contents of every device block a recorded pointer falls into (packed
// weights, tables, static buffers, activation scratch).  Loading it needs no Python, no torch and no model code: one
// hipMalloc, one pass of copies, pointer relocation - the counterpart of SURVEY 8b's es_load_weights for a host that cannot
// walk the model itself.  Layout (little endian, 8-byte aligned): see save().
extern "C" int es_ctx_load(const char* path, int device, es_ctx** out) {
  if (!path || !out) { es_set_error("es_ctx_load: null argument"); return -1; }
  FILE* f = fopen(path, "rb");
  if (!f) { es_set_error("es_ctx_load: cannot open the file"); return -1; }
  es_ctx* c = nullptr;
  std::vector<char> buf;
  auto fail = [&](const char* msg) { es_set_error(msg); if (c) es_ctx_destroy(c); fclose(f); return -1; };
  auto rd = [&](void* dst, size_t n) { return fread(dst, 1, n, f) == n; };
  struct { char magic[8]; unsigned abi, n_blocks; unsigned long long arena_bytes; } h;
  if (!rd(&h, sizeof(h)) || memcmp(h.magic, "ESCTX\3\0\0", 8) != 0) return fail("es_ctx_load: not a context image (or one of an older format: rebuild it)");
  if (h.abi != ES_ABI_VERSION) return fail("es_ctx_load: the image was written for another ABI version");
  if (h.arena_bytes > (1ull << 40) || h.n_blocks > (1u << 24)) return fail("es_ctx_load: implausible header");
  if (hipSetDevice(device) != hipSuccess) return fail("es_ctx_load: hipSetDevice failed");
  c = new es_ctx();
  c->device = device;
  struct { float cond_scales[6]; float start, end; int use_graphs; unsigned n_alphas; int scheduler; unsigned n_alphas_f64; } o;
  if (!rd(&c->g, sizeof(c->g)) || !rd(&o, sizeof(o))) return fail("es_ctx_load: truncated header");
  if (o.n_alphas > (1u << 20) || o.n_alphas_f64 > (1u << 20)) return fail("es_ctx_load: implausible schedule length");
  if (o.scheduler != ES_SCHED_DDIM && o.scheduler != ES_SCHED_UNIPC) return fail("es_ctx_load: unknown scheduler in the image");
  memcpy(c->cond_scales, o.cond_scales, sizeof(o.cond_scales));
  c->control_start = o.start; c->control_end = o.end; c->use_graphs = o.use_graphs;
  c->alphas_cumprod.resize(o.n_alphas);
  if (o.n_alphas && !rd(c->alphas_cumprod.data(), o.n_alphas * sizeof(float))) return fail("es_ctx_load: truncated schedule");
  if ((o.n_alphas & 1) && fseek(f, 4, SEEK_CUR)) return fail("es_ctx_load: truncated schedule");
  c->alphas_cumprod_f64.resize(o.n_alphas_f64);
  if (o.n_alphas_f64 && !rd(c->alphas_cumprod_f64.data(), o.n_alphas_f64 * sizeof(double))) return fail("es_ctx_load: truncated schedule");
  struct Blk { unsigned long long off, bytes; };
  // (overflow-safe: off <= arena and bytes <= arena - off)
  auto inside = [&](unsigned long long off, unsigned long long bytes) { return off <= h.arena_bytes && bytes <= h.arena_bytes - off; };
  std::vector<Blk> blocks(h.n_blocks);
  if (h.n_blocks && !rd(blocks.data(), h.n_blocks * sizeof(Blk))) return fail("es_ctx_load: truncated block table");
  for (const auto& b : blocks) if (!inside(b.off, b.bytes)) return fail("es_ctx_load: block outside the arena");
  if (hipMalloc(&c->arena, h.arena_bytes ? h.arena_bytes : 256) != hipSuccess) return fail("es_ctx_load: hipMalloc of the arena failed");
  c->arena_bytes = (size_t)h.arena_bytes;
  // run-time-produced memory (activations, split-K slabs, scratch) travels as space only: zero it once
  if (hipMemset(c->arena, 0, h.arena_bytes ? h.arena_bytes : 256) != hipSuccess) return fail("es_ctx_load: hipMemset of the arena failed");
  char* base = (char*)c->arena;
  for (int which = 0; which < ES_PLAN_COUNT; ++which) {
    unsigned long long pb = 0, nrel = 0;
    if (!rd(&pb, 8)) return fail("es_ctx_load: truncated plan table");
    if (!pb) continue;
    if (pb > (1ull << 32)) return fail("es_ctx_load: implausible plan size");
    buf.resize(pb);
    if (!rd(buf.data(), pb) || !rd(&nrel, 8)) return fail("es_ctx_load: truncated plan");
    es_plan* p = es_plan_import(buf.data(), pb);
    if (!p) { if (c) es_ctx_destroy(c); fclose(f); return -1; }
    c->plan[which] = p;
    // The stored relocation list is only trusted as far as the typed pointer-field tables agree with it: an entry must
    // address a pointer field of a recorded call (nothing else in a record - sizes, strides - can be overwritten by a
    // malformed image), and every non-null pointer field must have been relocated (none keeps a foreign address).
    std::vector<size_t> legal;
    for (const auto& op : p->ops) {
      int elem = 0;
      const auto& fl = ptr_fields(op.kind, elem);
      const size_t reps = elem ? op.bytes / (size_t)elem : 1;
      for (size_t r = 0; r < reps; ++r)
        for (const auto& pf : fl) {
          const size_t pos = op.off + r * (size_t)elem + (size_t)pf.off;
          if (pos + 8 > op.off + op.bytes) return fail("es_ctx_load: a recorded call is shorter than its argument record");
          legal.push_back(pos);
        }
    }
    std::sort(legal.begin(), legal.end());
    std::vector<char> done(legal.size(), 0);
    for (unsigned long long i = 0; i < nrel; ++i) {
      unsigned long long r[2];
      if (!rd(r, 16) || r[1] >= h.arena_bytes) return fail("es_ctx_load: bad relocation");
      const auto it = std::lower_bound(legal.begin(), legal.end(), (size_t)r[0]);
      if (it == legal.end() || *it != (size_t)r[0]) return fail("es_ctx_load: a relocation does not address a pointer field of a recorded call");
      const unsigned long long addr = (unsigned long long)(base + r[1]);
      memcpy(p->blob.data() + r[0], &addr, 8);
      done[(size_t)(it - legal.begin())] = 1;
    }
    for (size_t i = 0; i < legal.size(); ++i) {
      unsigned long long a;
      memcpy(&a, p->blob.data() + legal[i], 8);
      if (a && !done[i]) return fail("es_ctx_load: a pointer field of a recorded call has no relocation");
    }
  }
  for (int slot = 0; slot < ES_BUF_COUNT; ++slot) {
    long long r[2];
    if (!rd(r, 16)) return fail("es_ctx_load: truncated slot table");
    if (r[0] >= 0) {
      if (r[1] <= 0 || !inside((unsigned long long)r[0], (unsigned long long)r[1])) return fail("es_ctx_load: bound slot outside the arena");
      c->buf[slot] = base + r[0]; c->bytes[slot] = (size_t)r[1];
    }
  }
  // data extents (memory that every plan only reads: packed weights, parameters, tables), in table order, through a bounded
  // staging buffer
  unsigned long long n_ext = 0;
  if (!rd(&n_ext, 8) || n_ext > (1ull << 24)) return fail("es_ctx_load: truncated extent table");
  std::vector<Blk> ext((size_t)n_ext);
  if (n_ext && !rd(ext.data(), (size_t)n_ext * sizeof(Blk))) return fail("es_ctx_load: truncated extent table");
  for (const auto& b : ext) if (!inside(b.off, b.bytes)) return fail("es_ctx_load: extent outside the arena");
  for (const auto& b : ext) c->extents.emplace_back(b.off, b.bytes);
  buf.resize(64u << 20);
  for (const auto& b : ext) {
    for (unsigned long long done = 0; done < b.bytes;) {
      const size_t n = (size_t)((b.bytes - done) < buf.size() ? (b.bytes - done) : buf.size());
      if (!rd(buf.data(), n)) return fail("es_ctx_load: truncated extent data");
      if (hipMemcpy(base + b.off + done, buf.data(), n, hipMemcpyHostToDevice) != hipSuccess) return fail("es_ctx_load: copy to the device failed");
      done += n;
    }
  }
  fclose(f);
  c->scheduler = o.scheduler;          // (the graphs are instantiated lazily: nothing to rebuild)
  if (o.scheduler == ES_SCHED_UNIPC) {
    const int rc = es_ctx_set_scheduler(c, ES_SCHED_UNIPC);      // checks the state slots the image bound
    if (rc) { es_ctx_destroy(c); return rc; }
  }
  *out = c;
  return 0;
}

// Write a context that owns its arena (es_load_weights, es_ctx_load) as a context image es_ctx_load reads: the same format
// NativeEngine.save (edgestyle_amd/native.py) writes for a Python-built context.  Every pointer field of every recorded call
// (typed tables above) and every bound slot must lie inside the arena; only the persistent DATA extents travel as bytes, the rest
// of the arena (activations, slabs, scratch, input slots) as zero-filled space.
extern "C" int es_ctx_save(const es_ctx* c, const char* path) {
  if (!c || !path) { es_set_error("es_ctx_save: null argument"); return -1; }
  if (!c->arena || c->arena_on_host) { es_set_error("es_ctx_save: the context does not own a device arena (built by a Python host: use NativeEngine.save)"); return -1; }
  const unsigned long long base = (unsigned long long)c->arena, size = c->arena_bytes;
  auto inside = [&](unsigned long long a, unsigned long long n) { return a >= base && a - base <= size && n <= size - (a - base); };
  // written under a temporary name and renamed on success: a failure (a pointer outside the arena, a failed device copy, a full
  // disk) never leaves a truncated image under the final name
  const std::string tmp = std::string(path) + ".tmp";
  FILE* f = fopen(tmp.c_str(), "wb");
  if (!f) { es_set_error("es_ctx_save: cannot open the file"); return -1; }
  auto fail = [&](const char* msg) { es_set_error(msg); fclose(f); (void)remove(tmp.c_str()); return -1; };
  auto wr = [&](const void* p, size_t n) { return fwrite(p, 1, n, f) == n; };
  struct { char magic[8]; unsigned abi, n_blocks; unsigned long long arena_bytes; } h = {{'E', 'S', 'C', 'T', 'X', 3, 0, 0}, ES_ABI_VERSION, 1, size};
  struct { float cond_scales[6]; float start, end; int use_graphs; unsigned n_alphas; int scheduler; unsigned n_alphas_f64; } o;
  memcpy(o.cond_scales, c->cond_scales, sizeof(o.cond_scales));
  o.start = c->control_start; o.end = c->control_end; o.use_graphs = c->use_graphs; o.n_alphas = (unsigned)c->alphas_cumprod.size();
  o.scheduler = c->scheduler; o.n_alphas_f64 = (unsigned)c->alphas_cumprod_f64.size();
  if (!wr(&h, sizeof(h)) || !wr(&c->g, sizeof(c->g)) || !wr(&o, sizeof(o))) return fail("es_ctx_save: write failed");
  if (o.n_alphas && !wr(c->alphas_cumprod.data(), o.n_alphas * sizeof(float))) return fail("es_ctx_save: write failed");
  const unsigned zero4 = 0;
  if ((o.n_alphas & 1) && !wr(&zero4, 4)) return fail("es_ctx_save: write failed");
  if (o.n_alphas_f64 && !wr(c->alphas_cumprod_f64.data(), o.n_alphas_f64 * sizeof(double))) return fail("es_ctx_save: write failed");
  const unsigned long long blk[2] = {0, size};
  if (!wr(blk, 16)) return fail("es_ctx_save: write failed");
  for (int which = 0; which < ES_PLAN_COUNT; ++which) {
    const es_plan* p = c->plan[which];
    unsigned long long pb = p ? es_plan_export(p, nullptr, 0) : 0;
    if (!wr(&pb, 8)) return fail("es_ctx_save: write failed");
    if (!p) continue;
    std::vector<char> img((size_t)pb);
    es_plan_export(p, img.data(), img.size());
    const size_t blob0 = 16 + p->ops.size() * 24;
    std::vector<unsigned long long> rel;
    for (const auto& op : p->ops) {
      int elem = 0;
      const auto& fl = ptr_fields(op.kind, elem);
      const size_t reps = elem ? op.bytes / (size_t)elem : 1;
      for (size_t r = 0; r < reps; ++r)
        for (const auto& pf : fl) {
          const size_t pos = op.off + r * (size_t)elem + (size_t)pf.off;
          unsigned long long a;
          memcpy(&a, p->blob.data() + pos, 8);
          if (!a) continue;
          if (!inside(a, 1)) return fail("es_ctx_save: a recorded pointer lies outside the context's arena");
          rel.push_back(pos); rel.push_back(a - base);
          const unsigned long long off = a - base;          // the image holds no absolute address: the same context saves to the same bytes
          memcpy(img.data() + blob0 + pos, &off, 8);
        }
    }
    if (!wr(img.data(), img.size())) return fail("es_ctx_save: write failed");
    const unsigned long long nrel = rel.size() / 2;
    if (!wr(&nrel, 8) || (nrel && !wr(rel.data(), rel.size() * 8))) return fail("es_ctx_save: write failed");
  }
  for (int slot = 0; slot < ES_BUF_COUNT; ++slot) {
    long long r[2] = {-1, 0};
    if (c->buf[slot]) {
      if (!inside((unsigned long long)c->buf[slot], c->bytes[slot])) return fail("es_ctx_save: a bound slot lies outside the context's arena");
      r[0] = (long long)((unsigned long long)c->buf[slot] - base); r[1] = (long long)c->bytes[slot];
    }
    if (!wr(r, 16)) return fail("es_ctx_save: write failed");
  }
  const unsigned long long n_ext = c->extents.size();
  if (!wr(&n_ext, 8)) return fail("es_ctx_save: write failed");
  for (const auto& e : c->extents) { const unsigned long long r[2] = {e.first, e.second}; if (!wr(r, 16)) return fail("es_ctx_save: write failed"); }
  if (hipSetDevice(c->device) != hipSuccess) return fail("es_ctx_save: hipSetDevice failed");
  std::vector<char> buf(64u << 20);
  for (const auto& e : c->extents)
    for (unsigned long long done = 0; done < e.second;) {
      const size_t n = (size_t)std::min<unsigned long long>(e.second - done, buf.size());
      if (hipMemcpy(buf.data(), (const char*)c->arena + e.first + done, n, hipMemcpyDeviceToHost) != hipSuccess) return fail("es_ctx_save: copy from the device failed");
      if (!wr(buf.data(), n)) return fail("es_ctx_save: write failed");
      done += n;
    }
  if (fclose(f) != 0) { es_set_error("es_ctx_save: write failed"); (void)remove(tmp.c_str()); return -1; }
  if (rename(tmp.c_str(), path) != 0) { es_set_error("es_ctx_save: cannot rename the finished image to its final name"); (void)remove(tmp.c_str()); return -1; }
  return 0;
}

extern "C" int es_ctx_plan_size(const es_ctx* c, int which) {
  return (c && which >= 0 && which < ES_PLAN_COUNT && c->plan[which]) ? es_plan_size(c->plan[which]) : -1;
}

// == OnnxUNetAndControlnets.forward (export_onnx.py:43-74): one ControlNets -> fusion -> UNet evaluation at timestep t
extern "C" int es_denoise_step(es_ctx* c, const void* sample, float t, const void* ehs, const void* const* cond_embeds,
                               const float* scales, void* out_noise, void* stream) {
  if (!c || !sample || !ehs || !cond_embeds || !out_noise) { es_set_error("es_denoise_step: null argument"); return -1; }
  if (can_launch(c, "es_denoise_step")) return -1;
  hipStream_t st = (hipStream_t)stream;
  if (capturing(st)) { es_set_error("es_denoise_step: the stream is capturing; this entry point stages host values (timestep, scales) through pinned memory and cannot be captured - capture es_ctx_launch_plan(ES_PLAN_STEP_GENERIC) on buffers you fill yourself"); return -1; }
  if (!c->plan[ES_PLAN_STEP_GENERIC]) { es_set_error("es_denoise_step: the context has no ES_PLAN_STEP_GENERIC"); return -1; }
  if (need(c, ES_BUF_SAMPLE, "es_denoise_step: ES_BUF_SAMPLE not bound") || need(c, ES_BUF_T_ROWS, "es_denoise_step: ES_BUF_T_ROWS not bound") ||
      need(c, ES_BUF_EHS, "es_denoise_step: ES_BUF_EHS not bound") || need(c, ES_BUF_SCALES, "es_denoise_step: ES_BUF_SCALES not bound") ||
      need(c, ES_BUF_NOISE, "es_denoise_step: ES_BUF_NOISE not bound")) return -1;
  int rc;
  if ((rc = d2d(c->buf[ES_BUF_SAMPLE], sample, c->bytes[ES_BUF_SAMPLE], st))) return rc;
  if ((rc = d2d(c->buf[ES_BUF_EHS], ehs, c->bytes[ES_BUF_EHS], st))) return rc;
  for (int i = 0; i < c->g.n_conds; ++i) {
    if (!cond_embeds[i] || !c->buf[ES_BUF_COND0 + i]) { es_set_error("es_denoise_step: condition embedding missing / not bound"); return -1; }
    if ((rc = d2d(c->buf[ES_BUF_COND0 + i], cond_embeds[i], c->bytes[ES_BUF_COND0 + i], st))) return rc;
  }
  if (hipMemsetD32Async((hipDeviceptr_t)c->buf[ES_BUF_T_ROWS], __builtin_bit_cast(int, t), c->bytes[ES_BUF_T_ROWS] / 4, st) != hipSuccess) { es_set_error("es_denoise_step: fill failed"); return -2; }
  float* hs = staging(c, 8);
  if (!hs) { es_set_error("es_denoise_step: pinned staging allocation failed"); return -2; }
  for (int i = 0; i < 6; ++i) hs[i] = (scales && i < c->g.n_conds) ? scales[i] : 1.f;     // `scales` holds n_conds values
  if ((rc = h2d(c->buf[ES_BUF_SCALES], hs, (size_t)c->g.n_conds * 4, st))) return rc;
  if (hipEventRecord(c->staged, st) != hipSuccess) { es_set_error("es_denoise_step: hipEventRecord failed"); return -2; }
  if ((rc = run(c, ES_PLAN_STEP_GENERIC, st, nullptr))) return rc;
  return d2d(out_noise, c->buf[ES_BUF_NOISE], c->bytes[ES_BUF_NOISE], st);
}

// == the denoising loop of EdgeStyleStableDiffusionControlNetPipeline.__call__ (model/edgestyle_pipeline.py:435-543),
// DDIM eta 0: latents fp32 [B,h,w,L] NHWC in/out, ehs [N,77,D] dtype (negative prompt first under CFG, PL:329-330)
extern "C" int es_denoise_loop(es_ctx* c, float* latents_inout, const void* ehs, float guidance_scale, const float* timesteps,
                               int n_steps, void* stream) {
  if (!c || !latents_inout || !ehs || !timesteps) { es_set_error("es_denoise_loop: null argument"); return -1; }
  if (can_launch(c, "es_denoise_loop")) return -1;
  if (n_steps != c->g.n_steps) { es_set_error("es_denoise_loop: the context was built for another number of steps"); return -1; }
  const int need_slots[] = {ES_BUF_LATENTS, ES_BUF_SAMPLE, ES_BUF_EHS, ES_BUF_STEP_IDX, ES_BUF_T_TABLE, ES_BUF_SCALE_TABLE, ES_BUF_COEF, ES_BUF_TIMESTEPS};
  for (int s : need_slots) if (need(c, s, "es_denoise_loop: a static buffer is not bound")) return -1;
  hipStream_t st = (hipStream_t)stream;
  if (capturing(st)) { es_set_error("es_denoise_loop: the stream is capturing; this entry point stages host tables through pinned memory and cannot be captured (it replays its own hipGraphs: use_graphs 1 | 2)"); return -1; }
  if (!c->plan[ES_PLAN_PREP] || !c->plan[ES_PLAN_STEP]) { es_set_error("es_denoise_loop: the context has no ES_PLAN_PREP / ES_PLAN_STEP"); return -1; }
  const es_ctx_geometry& g = c->g;
  const int T = n_steps, nc = g.n_conds;
  const size_t trow = c->bytes[ES_BUF_T_TABLE] / 4 / T;          // kmax * N timestep copies per step
  const bool unipc = c->scheduler == ES_SCHED_UNIPC;
  const int cw = unipc ? 12 : 4;
  if (unipc && (c->bytes[ES_BUF_COEF] < (size_t)T * 48 || !c->buf[ES_BUF_HIST0] || !c->buf[ES_BUF_HIST1] || !c->buf[ES_BUF_HIST2])) {
    es_set_error("es_denoise_loop: the context's slots do not hold a UniPC state (ES_BUF_HIST*, ES_BUF_COEF of n_steps x 12)"); return -1; }
  float* tt = staging(c, (size_t)T * (trow + nc + cw + 1));
  if (!tt) { es_set_error("es_denoise_loop: pinned staging allocation failed"); return -2; }
  float* sc = tt + (size_t)T * trow;
  float* cf = sc + (size_t)T * nc;
  float* tsd = cf + (size_t)T * cw;
  // steps outside the control-guidance window (PL:419-427: controlnet_keep = 0 for every net) replay ES_PLAN_STEP_UNET when the
  // context has one: no ControlNet pass, no fusion launch (pipeline._Loop.one_step_unet); otherwise ES_PLAN_STEP with scale 0
  std::vector<char> unet_only((size_t)T, 0);
  for (int i = 0; i < T; ++i) {
    for (size_t j = 0; j < trow; ++j) tt[i * trow + j] = timesteps[i];
    const float keep = 1.0f - (float)(((float)i / T < c->control_start) || ((float)(i + 1) / T > c->control_end));   // PL:419-427
    for (int k = 0; k < nc; ++k) sc[i * nc + k] = c->cond_scales[k] * keep;
    unet_only[i] = keep == 0.0f && c->plan[ES_PLAN_STEP_UNET] != nullptr && es_plan_size(c->plan[ES_PLAN_STEP_UNET]) > 0;
    tsd[i] = timesteps[i];
  }
  if (unipc) unipc_coef(c, timesteps, T, cf);
  else ddim_coef(c, timesteps, T, cf);
  int rc;
  if ((rc = h2d(c->buf[ES_BUF_T_TABLE], tt, (size_t)T * trow * 4, st))) return rc;
  if ((rc = h2d(c->buf[ES_BUF_SCALE_TABLE], sc, (size_t)T * nc * 4, st))) return rc;
  if ((rc = h2d(c->buf[ES_BUF_COEF], cf, (size_t)T * cw * 4, st))) return rc;
  if (unipc)                                                            // multistep history: zero before the first step
    for (int i = 0; i < 3; ++i)
      if (hipMemsetAsync(c->buf[ES_BUF_HIST0 + i], 0, c->bytes[ES_BUF_HIST0 + i], st) != hipSuccess) { es_set_error("es_denoise_loop: memset failed"); return -2; }
  if ((rc = h2d(c->buf[ES_BUF_TIMESTEPS], tsd, (size_t)T * 4, st))) return rc;
  if (hipEventRecord(c->staged, st) != hipSuccess) { es_set_error("es_denoise_loop: hipEventRecord failed"); return -2; }
  if (hipMemsetAsync(c->buf[ES_BUF_STEP_IDX], 0, 4, st) != hipSuccess) { es_set_error("es_denoise_loop: memset failed"); return -2; }
  if ((rc = d2d(c->buf[ES_BUF_LATENTS], latents_inout, c->bytes[ES_BUF_LATENTS], st))) return rc;
  if ((rc = d2d(c->buf[ES_BUF_EHS], ehs, c->bytes[ES_BUF_EHS], st))) return rc;
  if ((rc = es_latents_to_input((const float*)c->buf[ES_BUF_LATENTS], c->buf[ES_BUF_SAMPLE], g.B, g.h * g.w, g.latent_channels,
                                g.latent_pad, g.cfg, g.dtype, stream))) return rc;
  if (c->use_graphs == 2 && !graph_hazard()) {
    // the preparation and all n step lists as ONE graph (BASELINE configs[2]: "hipGraph-captured scheduler loop"): every
    // step is the same launch list - the device step counter picks its rows of the tables - so the graph depends on
    // (n_steps, guidance scale) only
    if (c->loop_exec && (c->loop_steps != T || c->loop_guidance != guidance_scale || c->loop_window[0] != c->control_start || c->loop_window[1] != c->control_end)) {
      (void)hipGraphExecDestroy(c->loop_exec); c->loop_exec = nullptr; }
    if (!c->loop_exec) {
      hipGraph_t graph = nullptr;
      if (!c->cap_stream && hipStreamCreateWithFlags(&c->cap_stream, hipStreamNonBlocking) != hipSuccess) { es_set_error("es_ctx: hipStreamCreate failed"); return -2; }
      if (hipStreamBeginCapture(c->cap_stream, hipStreamCaptureModeThreadLocal) != hipSuccess) { es_set_error("es_ctx: hipStreamBeginCapture failed"); return -2; }
      RunOpts ro;
      ro.unipc = c->scheduler == ES_SCHED_UNIPC ? c : nullptr;
      rc = run_plan(c->plan[ES_PLAN_PREP], c->cap_stream, ro);
      ro.guidance = &guidance_scale;
      for (int i = 0; i < T && !rc; ++i) rc = run_plan(c->plan[unet_only[i] ? ES_PLAN_STEP_UNET : ES_PLAN_STEP], c->cap_stream, ro);
      const hipError_t e = hipStreamEndCapture(c->cap_stream, &graph);
      if (rc || e != hipSuccess || !graph) { if (graph) (void)hipGraphDestroy(graph); if (!rc) es_set_error("es_ctx: hipStreamEndCapture failed"); return rc ? rc : -2; }
      const hipError_t ei = hipGraphInstantiate(&c->loop_exec, graph, nullptr, nullptr, 0);
      (void)hipGraphDestroy(graph);
      if (ei != hipSuccess) { c->loop_exec = nullptr; es_set_error("es_ctx: hipGraphInstantiate failed"); return -2; }
      c->loop_steps = T; c->loop_guidance = guidance_scale;
      c->loop_window[0] = c->control_start; c->loop_window[1] = c->control_end;
    }
    if (hipGraphLaunch(c->loop_exec, st) != hipSuccess) { es_set_error("es_ctx: hipGraphLaunch failed"); return -2; }
  } else {
    if ((rc = run(c, ES_PLAN_PREP, st, nullptr))) return rc;      // text K/V projections, condition slots, time-projection table
    for (int i = 0; i < T; ++i)
      if ((rc = run(c, unet_only[i] ? ES_PLAN_STEP_UNET : ES_PLAN_STEP, st, &guidance_scale))) return rc;
  }
  return d2d(latents_inout, c->buf[ES_BUF_LATENTS], c->bytes[ES_BUF_LATENTS], st);
}

// == prepare_image (PL:629-664) + the one-time conditioning embedding (CL:28-42, 289-290) for all nets of the context
extern "C" int es_prepare_conds(es_ctx* c, const float* const* images, const float* const* noise, void* stream) {
  if (!c || !images) { es_set_error("es_prepare_conds: null argument"); return -1; }
  if (can_launch(c, "es_prepare_conds")) return -1;
  if (!c->plan[ES_PLAN_CONDS]) { es_set_error("es_prepare_conds: the context has no ES_PLAN_CONDS (built from pre-embedded conditions)"); return -1; }
  hipStream_t st = (hipStream_t)stream;
  int rc;
  for (int i = 0; i < c->g.n_conds; ++i) {
    if (!images[i] || !c->buf[ES_BUF_COND_IMG0 + i]) { es_set_error("es_prepare_conds: condition image missing / ES_BUF_COND_IMG not bound"); return -1; }
    if ((rc = d2d(c->buf[ES_BUF_COND_IMG0 + i], images[i], c->bytes[ES_BUF_COND_IMG0 + i], st))) return rc;
    if (c->buf[ES_BUF_COND_NOISE0 + i]) {
      if (!noise || !noise[i]) { es_set_error("es_prepare_conds: a VAE-conditioned net needs its latent sampling noise"); return -1; }
      if ((rc = d2d(c->buf[ES_BUF_COND_NOISE0 + i], noise[i], c->bytes[ES_BUF_COND_NOISE0 + i], st))) return rc;
    }
  }
  return run(c, ES_PLAN_CONDS, st, nullptr);
}

// == vae.decode(latents / scaling_factor) + image_processor.postprocess(output_type "pt") (PL:552-572):
// latents fp32 [B,h,w,L] NHWC -> image fp32 [B,3,8h,8w] NCHW in [0,1]
extern "C" int es_vae_decode(es_ctx* c, const float* latents, float* out_img, void* stream) {
  if (!c || !latents || !out_img) { es_set_error("es_vae_decode: null argument"); return -1; }
  if (can_launch(c, "es_vae_decode")) return -1;
  if (need(c, ES_BUF_SAMPLE, "es_vae_decode: ES_BUF_SAMPLE not bound") || need(c, ES_BUF_IMAGE, "es_vae_decode: ES_BUF_IMAGE not bound")) return -1;
  const es_ctx_geometry& g = c->g;
  int rc;
  if ((rc = es_latents_to_input(latents, c->buf[ES_BUF_SAMPLE], g.B, g.h * g.w, g.latent_channels, g.latent_pad, g.cfg, g.dtype, stream))) return rc;
  if ((rc = run(c, ES_PLAN_DECODE, (hipStream_t)stream, nullptr))) return rc;
  return d2d(out_img, c->buf[ES_BUF_IMAGE], c->bytes[ES_BUF_IMAGE], (hipStream_t)stream);
}
