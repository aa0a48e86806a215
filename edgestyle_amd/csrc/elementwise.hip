// Small streaming kernels of the EdgeStyle hot path (gfx950): timestep sinusoid, CFG + DDIM step,
// NCHW<->NHWC boundary conversion, elementwise add, VAE latent sampling, device step counter.
#include "common.h"
#include "../../include/edgestyle_hip.h"
#include "plan.h"
#include <string.h>

namespace {

thread_local char g_err[256] = "";

template <typename T>
__global__ void timestep_kernel(const float* __restrict__ t, T* __restrict__ out, int N, int dim) {
  // diffusers Timesteps(dim, flip_sin_to_cos=True, downscale_freq_shift=0): [cos | sin], f_i = exp(-ln(1e4) i/half)
  const int half = dim / 2;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * half) return;
  const int n = i / half, k = i - n * half;
  const float f = expf(-9.210340371976184f * (float)k / (float)half);
  const float a = t[n] * f;
  out[(size_t)n * dim + k] = from_f32<T>(cosf(a));
  out[(size_t)n * dim + half + k] = from_f32<T>(sinf(a));
}

template <typename T>
__global__ void cfg_ddim_kernel(const T* __restrict__ noise, float* __restrict__ lat, T* __restrict__ model_in,
                                const float* __restrict__ coef, const int* __restrict__ step_idx, float gs, int B,
                                int HW, int L, int Ls, int cfg, int nsteps) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long total = (long long)B * HW * L;
  if (i >= total) return;
  const long long mi = (i / L) * Ls + (i % L);         // model_in is channel-padded to Ls
  const long long mtotal = (long long)B * HW * Ls;
  int sidx = *step_idx;
  sidx = sidx < 0 ? 0 : (sidx >= nsteps ? nsteps - 1 : sidx);
  const float* c = coef + (size_t)sidx * 4;
  float eps;
  if (cfg) {
    const float eu = to_f32(noise[i]), ec = to_f32(noise[total + i]);
    eps = eu + gs * (ec - eu);
  } else {
    eps = to_f32(noise[i]);
  }
  const float x = lat[i];
  const float x0 = (x - c[1] * eps) / c[0];
  float xn = c[2] * x0 + c[3] * eps;
  // the networks' input is the ROUNDED fp32 latent converted once more, exactly what es_latents_to_input derives from the
  // stored latents: without this barrier hipcc fuses the last multiply-add with the conversion (v_fma_mixlo_f16, one
  // rounding instead of two) and one value in ~10^4 lands on the neighbouring fp16
  asm volatile("" : "+v"(xn));
  lat[i] = xn;
  const T xt = from_f32<T>(xn);
  model_in[mi] = xt;
  if (cfg) model_in[mtotal + mi] = xt;
}

// latents (fp32 [B,HW,L]) -> the networks' input: compute dtype, channels zero-padded to Ls, both CFG halves (PL:443-447)
template <typename T>
__global__ void latents_to_input_kernel(const float* __restrict__ lat, T* __restrict__ model_in, int B, int HW, int L,
                                        int Ls, int cfg) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long mtotal = (long long)B * HW * Ls;
  if (i >= mtotal) return;
  const int c = (int)(i % Ls);
  const T v = c < L ? from_f32<T>(lat[(i / Ls) * L + c]) : from_f32<T>(0.f);
  model_in[i] = v;
  if (cfg) model_in[mtotal + i] = v;
}

template <typename T>
__global__ void cfg_unipc_kernel(const T* __restrict__ noise, float* __restrict__ lat, float* __restrict__ last,
                                 float* __restrict__ m0b, float* __restrict__ m1b, T* __restrict__ model_in,
                                 const float* __restrict__ coef, const int* __restrict__ step_idx, float gs, int B,
                                 int HW, int L, int Ls, int cfg, int nsteps) {
  // One UniPC (bh2, order <= 2, predict_x0) step as a linear recombination with host-computed per-step scalars:
  //   x0  = (x - sigma*eps)/alpha                                   convert_model_output
  //   xc  = use_c ? cl*last + cm0*m0 + cm1*m1 + cmt*x0 : x          multistep_uni_c_bh_update
  //   x'  = px*xc + pm0*x0 + pm1*m0                                 multistep_uni_p_bh_update
  //   history: m1 <- m0, m0 <- x0, last <- xc
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long total = (long long)B * HW * L;
  if (i >= total) return;
  const long long mi = (i / L) * Ls + (i % L);
  const long long mtotal = (long long)B * HW * Ls;
  int sidx = *step_idx;
  sidx = sidx < 0 ? 0 : (sidx >= nsteps ? nsteps - 1 : sidx);
  const float* c = coef + (size_t)sidx * 12;
  float eps;
  if (cfg) {
    const float eu = to_f32(noise[i]), ec = to_f32(noise[total + i]);
    eps = eu + gs * (ec - eu);
  } else {
    eps = to_f32(noise[i]);
  }
  const float x = lat[i];
  const float x0 = (x - c[1] * eps) / c[0];
  const float m0 = m0b[i], m1 = m1b[i];
  const float xc = c[2] != 0.f ? c[3] * last[i] + c[4] * m0 + c[5] * m1 + c[6] * x0 : x;
  float xn = c[7] * xc + c[8] * x0 + c[9] * m0;
  asm volatile("" : "+v"(xn));            // as in cfg_ddim_kernel: model_in == fp16(stored latent), bit for bit
  m1b[i] = m0;
  m0b[i] = x0;
  last[i] = xc;
  lat[i] = xn;
  const T xt = from_f32<T>(xn);
  model_in[mi] = xt;
  if (cfg) model_in[mtotal + mi] = xt;
}

template <typename T>
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ in, T* __restrict__ out, int N, int C, int HW,
                                    int Cpad) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long total = (long long)N * HW * Cpad;
  if (i >= total) return;
  const int c = (int)(i % Cpad);
  const long long np = i / Cpad;
  const int px = (int)(np % HW);
  const int n = (int)(np / HW);
  out[i] = c < C ? from_f32<T>(in[((size_t)n * C + c) * HW + px]) : from_f32<T>(0.f);
}

template <typename T>
__global__ void nhwc_to_nchw_kernel(const T* __restrict__ in, float* __restrict__ out, int N, int C, int HW,
                                    int Cs, float scale, float shift, int clamp01) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long total = (long long)N * C * HW;
  if (i >= total) return;
  const int px = (int)(i % HW);
  const long long nc = i / HW;
  const int c = (int)(nc % C);
  const int n = (int)(nc / C);
  float v = to_f32(in[((size_t)n * HW + px) * Cs + c]) * scale + shift;
  if (clamp01) v = fminf(fmaxf(v, 0.f), 1.f);
  out[i] = v;
}

template <typename T>
__global__ void add_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ y, long long n8) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n8;
       i += (long long)gridDim.x * blockDim.x) {
    const auto va = as_vec8<T>(((const u32x4*)a)[i]);
    const auto vb = as_vec8<T>(((const u32x4*)b)[i]);
    typename Traits<T>::vec8 r;
#pragma unroll
    for (int e = 0; e < 8; ++e) r[e] = from_f32<T>(to_f32(va[e]) + to_f32(vb[e]));
    ((typename Traits<T>::vec8*)y)[i] = r;
  }
}

template <typename T>
__global__ void vae_sample_kernel(const T* __restrict__ mom, const float* __restrict__ noise, T* __restrict__ z,
                                  int N, int HW, int L, int Lpad, float scaling) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long total = (long long)N * HW * Lpad;
  if (i >= total) return;
  const int c = (int)(i % Lpad);
  const long long np = i / Lpad;
  const int px = (int)(np % HW);
  const int n = (int)(np / HW);
  float v = 0.f;
  if (c < L) {
    const float mean = to_f32(mom[np * (2 * L) + c]);
    float logvar = to_f32(mom[np * (2 * L) + L + c]);
    logvar = fminf(fmaxf(logvar, -30.f), 20.f);
    v = (mean + expf(0.5f * logvar) * noise[((size_t)n * L + c) * HW + px]) * scaling;
  }
  z[i] = from_f32<T>(v);
}

__global__ void incr_kernel(int* ctr) { *ctr += 1; }

__global__ void gather_row_kernel(const float* __restrict__ table, const int* __restrict__ idx,
                                  float* __restrict__ out, int row_len, int nrows) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  int r = *idx;
  r = r < 0 ? 0 : (r >= nrows ? nrows - 1 : r);          // a counter past the table must never fault the GPU
  if (i < row_len) out[i] = table[(size_t)r * row_len + i];
}

inline unsigned nblk(long long n, int b = 256) { return (unsigned)((n + b - 1) / b); }

}  // namespace

extern "C" void es_set_error(const char* msg) {
  strncpy(g_err, msg, sizeof(g_err) - 1);
  g_err[sizeof(g_err) - 1] = 0;
}
extern "C" const char* es_last_error(void) { return g_err; }
extern "C" int es_abi_version(void) { return ES_ABI_VERSION; }

#define ES_RET(name)                                              \
  do {                                                            \
    if (hipGetLastError() != hipSuccess) {                        \
      es_set_error(name ": launch failed");                       \
      return -2;                                                  \
    }                                                             \
    return 0;                                                     \
  } while (0)

extern "C" int es_timestep_embedding(const float* t, void* out, int N, int dim, int dtype, void* stream) {
  if (!t || !out || dim % 2 || N < 1) { es_set_error("es_timestep_embedding: bad arguments"); return -1; }
  if (es_plan_recording()) { const es_op_timestep a{t, out, N, dim, dtype}; es_plan_record(ES_OP_TIMESTEP_EMBEDDING, &a, sizeof(a)); ES_PLAN_DRY_RETURN(); }
  hipStream_t st = (hipStream_t)stream;
  const long long n = (long long)N * (dim / 2);
  if (dtype == ES_F16) hipLaunchKernelGGL(timestep_kernel<f16>, dim3(nblk(n)), dim3(256), 0, st, t, (f16*)out, N, dim);
  else hipLaunchKernelGGL(timestep_kernel<bf16>, dim3(nblk(n)), dim3(256), 0, st, t, (bf16*)out, N, dim);
  ES_RET("es_timestep_embedding");
}

extern "C" int es_cfg_ddim_step(const void* noise, float* latents, void* model_in, const float* coef,
                                const int32_t* step_idx, float guidance_scale, int B, int HW, int L, int Lstride,
                                int cfg, int nsteps, int dtype, void* stream) {
  if (!noise || !latents || !model_in || !coef || !step_idx || B < 1 || HW < 1 || L < 1 || Lstride < L || nsteps < 1) {
    es_set_error("es_cfg_ddim_step: bad arguments"); return -1;
  }
  if (es_plan_recording()) {
    const es_op_cfg_ddim a{noise, latents, model_in, coef, step_idx, guidance_scale, B, HW, L, Lstride, cfg, nsteps, dtype};
    es_plan_record(ES_OP_CFG_DDIM, &a, sizeof(a)); ES_PLAN_DRY_RETURN();
  }
  hipStream_t st = (hipStream_t)stream;
  const long long n = (long long)B * HW * L;
  if (dtype == ES_F16)
    hipLaunchKernelGGL(cfg_ddim_kernel<f16>, dim3(nblk(n)), dim3(256), 0, st, (const f16*)noise, latents,
                       (f16*)model_in, coef, step_idx, guidance_scale, B, HW, L, Lstride, cfg, nsteps);
  else
    hipLaunchKernelGGL(cfg_ddim_kernel<bf16>, dim3(nblk(n)), dim3(256), 0, st, (const bf16*)noise, latents,
                       (bf16*)model_in, coef, step_idx, guidance_scale, B, HW, L, Lstride, cfg, nsteps);
  ES_RET("es_cfg_ddim_step");
}

extern "C" int es_cfg_unipc_step(const void* noise, float* latents, float* last_sample, float* m0, float* m1,
                                 void* model_in, const float* coef, const int32_t* step_idx, float guidance_scale,
                                 int B, int HW, int L, int Lstride, int cfg, int nsteps, int dtype, void* stream) {
  if (!noise || !latents || !last_sample || !m0 || !m1 || !model_in || !coef || !step_idx || B < 1 || HW < 1 ||
      L < 1 || Lstride < L || nsteps < 1) {
    es_set_error("es_cfg_unipc_step: bad arguments"); return -1;
  }
  if (es_plan_recording()) {
    const es_op_cfg_unipc a{noise, latents, last_sample, m0, m1, model_in, coef, step_idx, guidance_scale, B, HW, L, Lstride, cfg, nsteps, dtype};
    es_plan_record(ES_OP_CFG_UNIPC, &a, sizeof(a)); ES_PLAN_DRY_RETURN();
  }
  hipStream_t st = (hipStream_t)stream;
  const long long n = (long long)B * HW * L;
  if (dtype == ES_F16)
    hipLaunchKernelGGL(cfg_unipc_kernel<f16>, dim3(nblk(n)), dim3(256), 0, st, (const f16*)noise, latents,
                       last_sample, m0, m1, (f16*)model_in, coef, step_idx, guidance_scale, B, HW, L, Lstride, cfg, nsteps);
  else
    hipLaunchKernelGGL(cfg_unipc_kernel<bf16>, dim3(nblk(n)), dim3(256), 0, st, (const bf16*)noise, latents,
                       last_sample, m0, m1, (bf16*)model_in, coef, step_idx, guidance_scale, B, HW, L, Lstride, cfg, nsteps);
  ES_RET("es_cfg_unipc_step");
}

extern "C" int es_nchw_f32_to_nhwc(const float* in, void* out, int N, int C, int HW, int Cpad, int dtype,
                                   void* stream) {
  if (!in || !out || Cpad < C || N < 1) { es_set_error("es_nchw_f32_to_nhwc: bad arguments"); return -1; }
  if (es_plan_recording()) { const es_op_nchw_to_nhwc a{in, out, N, C, HW, Cpad, dtype}; es_plan_record(ES_OP_NCHW_TO_NHWC, &a, sizeof(a)); ES_PLAN_DRY_RETURN(); }
  hipStream_t st = (hipStream_t)stream;
  const long long n = (long long)N * HW * Cpad;
  if (dtype == ES_F16) hipLaunchKernelGGL(nchw_to_nhwc_kernel<f16>, dim3(nblk(n)), dim3(256), 0, st, in, (f16*)out, N, C, HW, Cpad);
  else hipLaunchKernelGGL(nchw_to_nhwc_kernel<bf16>, dim3(nblk(n)), dim3(256), 0, st, in, (bf16*)out, N, C, HW, Cpad);
  ES_RET("es_nchw_f32_to_nhwc");
}

extern "C" int es_nhwc_to_nchw_f32(const void* in, float* out, int N, int C, int HW, int Cstride, float scale,
                                   float shift, int clamp01, int dtype, void* stream) {
  if (!in || !out || Cstride < C || N < 1) { es_set_error("es_nhwc_to_nchw_f32: bad arguments"); return -1; }
  if (es_plan_recording()) { const es_op_nhwc_to_nchw a{in, out, N, C, HW, Cstride, scale, shift, clamp01, dtype}; es_plan_record(ES_OP_NHWC_TO_NCHW, &a, sizeof(a)); ES_PLAN_DRY_RETURN(); }
  hipStream_t st = (hipStream_t)stream;
  const long long n = (long long)N * C * HW;
  if (dtype == ES_F16) hipLaunchKernelGGL(nhwc_to_nchw_kernel<f16>, dim3(nblk(n)), dim3(256), 0, st, (const f16*)in, out, N, C, HW, Cstride, scale, shift, clamp01);
  else hipLaunchKernelGGL(nhwc_to_nchw_kernel<bf16>, dim3(nblk(n)), dim3(256), 0, st, (const bf16*)in, out, N, C, HW, Cstride, scale, shift, clamp01);
  ES_RET("es_nhwc_to_nchw_f32");
}

extern "C" int es_add(const void* a, const void* b, void* y, int64_t n, int dtype, void* stream) {
  if (!a || !b || !y || n < 8 || n % 8) { es_set_error("es_add: n must be a positive multiple of 8"); return -1; }
  if (es_plan_recording()) { const es_op_add rec{a, b, y, n, dtype}; es_plan_record(ES_OP_ADD, &rec, sizeof(rec)); ES_PLAN_DRY_RETURN(); }
  hipStream_t st = (hipStream_t)stream;
  const long long n8 = n / 8;
  unsigned blocks = nblk(n8);
  if (blocks > 2048) blocks = 2048;
  if (dtype == ES_F16) hipLaunchKernelGGL(add_kernel<f16>, dim3(blocks), dim3(256), 0, st, (const f16*)a, (const f16*)b, (f16*)y, n8);
  else hipLaunchKernelGGL(add_kernel<bf16>, dim3(blocks), dim3(256), 0, st, (const bf16*)a, (const bf16*)b, (bf16*)y, n8);
  ES_RET("es_add");
}

extern "C" int es_vae_sample(const void* moments, const float* noise_nchw, void* z, int N, int HW, int L, int Lpad,
                             float scaling, int dtype, void* stream) {
  if (!moments || !noise_nchw || !z || Lpad < L || N < 1) { es_set_error("es_vae_sample: bad arguments"); return -1; }
  if (es_plan_recording()) { const es_op_vae_sample a{moments, noise_nchw, z, N, HW, L, Lpad, scaling, dtype}; es_plan_record(ES_OP_VAE_SAMPLE, &a, sizeof(a)); ES_PLAN_DRY_RETURN(); }
  hipStream_t st = (hipStream_t)stream;
  const long long n = (long long)N * HW * Lpad;
  if (dtype == ES_F16) hipLaunchKernelGGL(vae_sample_kernel<f16>, dim3(nblk(n)), dim3(256), 0, st, (const f16*)moments, noise_nchw, (f16*)z, N, HW, L, Lpad, scaling);
  else hipLaunchKernelGGL(vae_sample_kernel<bf16>, dim3(nblk(n)), dim3(256), 0, st, (const bf16*)moments, noise_nchw, (bf16*)z, N, HW, L, Lpad, scaling);
  ES_RET("es_vae_sample");
}

// One wave that does nothing but watch two counters for `ticks` of the 100 MHz real-time counter: shader cycles (s_memtime) over real time
// = the clock the chip holds while whatever runs beside it (another stream) runs.  It sleeps between looks (s_sleep: no issue slots
// taken from its CU's other waves) and every wave reaches the exit: the real-time counter only moves forward.
__global__ __launch_bounds__(64) void clock_probe_kernel(unsigned long long* out, const unsigned long long ticks) {
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  const unsigned long long c0 = __builtin_amdgcn_s_memtime();
  unsigned long long r = r0;
  while (r - r0 < ticks) {
    __builtin_amdgcn_s_sleep(64);
    r = __builtin_amdgcn_s_memrealtime();
  }
  const unsigned long long c = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) { out[0] = c - c0; out[1] = r - r0; }
}

extern "C" int es_clock_probe(unsigned long long* out2, unsigned int duration_us, void* stream) {
  if (!out2 || duration_us < 1 || duration_us > 30000000u) { es_set_error("es_clock_probe: out2 = device u64[2], 1 us ... 30 s"); return -1; }
  if (es_plan_recording()) { es_set_error("es_clock_probe: a measurement tool, not part of a plan"); return -1; }
  hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, out2, (unsigned long long)duration_us * 100ull);
  ES_RET("es_clock_probe");
}

extern "C" int es_incr(int32_t* ctr, void* stream) {
  if (!ctr) { es_set_error("es_incr: null pointer"); return -1; }
  if (es_plan_recording()) { const es_op_incr a{ctr}; es_plan_record(ES_OP_INCR, &a, sizeof(a)); ES_PLAN_DRY_RETURN(); }
  hipLaunchKernelGGL(incr_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, ctr);
  ES_RET("es_incr");
}

extern "C" int es_gather_row(const float* table, const int32_t* idx, float* out, int row_len, int nrows,
                             void* stream) {
  if (!table || !idx || !out || row_len < 1 || nrows < 1) { es_set_error("es_gather_row: bad arguments"); return -1; }
  if (es_plan_recording()) { const es_op_gather_row a{table, idx, out, row_len, nrows}; es_plan_record(ES_OP_GATHER_ROW, &a, sizeof(a)); ES_PLAN_DRY_RETURN(); }
  hipLaunchKernelGGL(gather_row_kernel, dim3(nblk(row_len)), dim3(256), 0, (hipStream_t)stream, table, idx, out,
                     row_len, nrows);
  ES_RET("es_gather_row");
}

extern "C" int es_latents_to_input(const float* latents, void* model_in, int B, int HW, int L, int Lstride, int cfg,
                                   int dtype, void* stream) {
  if (!latents || !model_in || B < 1 || HW < 1 || L < 1 || Lstride < L) { es_set_error("es_latents_to_input: bad arguments"); return -1; }
  if (es_plan_recording()) { const es_op_latents_to_input a{latents, model_in, B, HW, L, Lstride, cfg, dtype}; es_plan_record(ES_OP_LATENTS_TO_INPUT, &a, sizeof(a)); ES_PLAN_DRY_RETURN(); }
  const long long n = (long long)B * HW * Lstride;
  if (dtype == ES_F16) hipLaunchKernelGGL(latents_to_input_kernel<f16>, dim3(nblk(n)), dim3(256), 0, (hipStream_t)stream, latents, (f16*)model_in, B, HW, L, Lstride, cfg);
  else hipLaunchKernelGGL(latents_to_input_kernel<bf16>, dim3(nblk(n)), dim3(256), 0, (hipStream_t)stream, latents, (bf16*)model_in, B, HW, L, Lstride, cfg);
  ES_RET("es_latents_to_input");
}

/* device-to-device copies and fills as C-ABI calls, so that the data movement of a step is part of a recorded plan */
extern "C" int es_memcpy(void* dst, const void* src, size_t bytes, void* stream) {
  if (!dst || !src || !bytes) { es_set_error("es_memcpy: bad arguments"); return -1; }
  if (es_plan_recording()) { const es_op_memcpy a{dst, src, bytes}; es_plan_record(ES_OP_MEMCPY, &a, sizeof(a)); ES_PLAN_DRY_RETURN(); }
  if (hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess) { es_set_error("es_memcpy: hipMemcpyAsync failed"); return -2; }
  return 0;
}

extern "C" int es_memcpy2d(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t height, void* stream) {
  if (!dst || !src || !width || !height || dpitch < width || spitch < width) { es_set_error("es_memcpy2d: bad arguments"); return -1; }
  if (es_plan_recording()) { const es_op_memcpy2d a{dst, dpitch, src, spitch, width, height}; es_plan_record(ES_OP_MEMCPY2D, &a, sizeof(a)); ES_PLAN_DRY_RETURN(); }
  if (hipMemcpy2DAsync(dst, dpitch, src, spitch, width, height, hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess) { es_set_error("es_memcpy2d: hipMemcpy2DAsync failed"); return -2; }
  return 0;
}

extern "C" int es_fill_f32(float* dst, float value, size_t n, void* stream) {
  if (!dst || !n) { es_set_error("es_fill_f32: bad arguments"); return -1; }
  if (es_plan_recording()) { const es_op_fill_f32 a{dst, n, value}; es_plan_record(ES_OP_FILL_F32, &a, sizeof(a)); ES_PLAN_DRY_RETURN(); }
  if (hipMemsetD32Async((hipDeviceptr_t)dst, __builtin_bit_cast(int, value), n, (hipStream_t)stream) != hipSuccess) { es_set_error("es_fill_f32: hipMemsetD32Async failed"); return -2; }
  return 0;
}

extern "C" size_t es_sizeof_desc(int which) {
  switch (which) {
    case 0: return sizeof(es_gemm_desc);
    case 1: return sizeof(es_attn_desc);
    case 2: return sizeof(es_gn_desc);
    case 3: return sizeof(es_fusion_desc);
    case 4: return sizeof(es_ln_desc);
    case 5: return sizeof(es_xs_desc);
    default: return 0;
  }
}
