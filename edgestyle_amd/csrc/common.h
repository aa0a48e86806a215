// Shared device helpers for the gfx950 (CDNA4) kernels of the EdgeStyle hot path.
// Wave = 64 lanes, MFMA 16x16x32 f16/bf16, fp32 accumulate everywhere.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef _Float16 f16;
typedef __bf16 bf16;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef __fp16 h16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef short s16x4 __attribute__((ext_vector_type(4)));

#define ES_DEVICE __device__ __forceinline__

template <typename T> struct Traits;
template <> struct Traits<f16> {
  typedef f16x8 vec8;
  typedef f16x4 vec4;
  static constexpr bool is_bf16 = false;
};
template <> struct Traits<bf16> {
  typedef bf16x8 vec8;
  typedef bf16x4 vec4;
  static constexpr bool is_bf16 = true;
};

ES_DEVICE f32x4 mfma16(f16x8 a, f16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
ES_DEVICE f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }

ES_DEVICE float to_f32(f16 x) { return (float)x; }
ES_DEVICE float to_f32(bf16 x) { return (float)x; }
template <typename T> ES_DEVICE T from_f32(float x) { return (T)x; }

// a / d for 0 <= a < 2^24, d >= 1, inv = 1.0f / d: the float quotient is off by at most one, fixed by one
// correction step.  (Integer division is ~40 VALU instructions for 32 bits and >100 for 64 bits: index math of the
// streaming kernels goes through this, with the exact division as the fallback for larger ranges.)
ES_DEVICE int fast_div(int a, int d, float inv) {
  int q = (int)((float)a * inv);
  const int r = a - q * d;
  q += (r >= d) - (r < 0);
  return q;
}
ES_DEVICE int div_any(int a, int d, float inv, bool small) { return small ? fast_div(a, d, inv) : a / d; }

ES_DEVICE float silu_f(float x) { return x / (1.0f + __expf(-x)); }
// exact (erf) GELU, as torch.nn.functional.gelu default used by diffusers GEGLU.  erf by Abramowitz-Stegun 7.1.26
// (|abs err| <= 1.5e-7, far below fp16/bf16 resolution): one rcp + one exp + 5 FMA instead of libm erff's ~40 ops.
ES_DEVICE float erf_as(float x) {
  const float ax = fabsf(x);
  const float t = __frcp_rn(1.0f + 0.3275911f * ax);
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float r = 1.0f - poly * __expf(-ax * ax);
  return copysignf(r, x);
}
ES_DEVICE float gelu_f(float x) { return 0.5f * x * (1.0f + erf_as(x * 0.70710678118654752f)); }

// transposed LDS read: 16-lane group reads a 4x16 block of 16-bit elements, lane i gets column i (4 rows)
ES_DEVICE u32x2 lds_read_tr16(const void* lds_ptr) {
  h16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) h16x4*)lds_ptr);
  return __builtin_bit_cast(u32x2, v);
}

template <typename T> ES_DEVICE typename Traits<T>::vec8 as_vec8(u32x4 v) {
  return __builtin_bit_cast(typename Traits<T>::vec8, v);
}

ES_DEVICE float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
ES_DEVICE float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

#define ES_CHECK_LAUNCH() (hipGetLastError() == hipSuccess ? 0 : -2)
