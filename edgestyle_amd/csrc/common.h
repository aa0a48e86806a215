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

// max over the lane pairs (l, l^16) / (l, l^32) with gfx950's v_permlane16/32_swap: swapping two copies of a register
// leaves {own, partner} in the two results in either half, so a symmetric op needs no select and no LDS round trip.
// Inline asm: hipcc folds the builtin on two equal inputs (and the max after it) away; `s_nop 1` covers the two wait
// states between a VALU write of an operand and the swap reading it.
ES_DEVICE float xor16_max(float x) {
  float a = x, b = x;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return fmaxf(a, b);
}
ES_DEVICE float xor32_max(float x) {
  float a = x, b = x;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return fmaxf(a, b);
}

ES_DEVICE float xor16_sum(float x) {
  float a = x, b = x;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return a + b;
}
ES_DEVICE float xor32_sum(float x) {
  float a = x, b = x;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return a + b;
}
// Whole-wave reductions without LDS (`__shfl_xor` compiles to ds_bpermute, ~120 cycles of latency per step): DPP
// inside a 16-lane row (xor 1, xor 2, then the half-row and row mirrors: after the quad steps every lane of a quad holds
// the quad's value, so mirrors pair the right quads), permlane swaps across rows.  Fixed order: deterministic.
template <int CTRL>
ES_DEVICE float dpp_mov(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xF, 0xF, true));
}
ES_DEVICE float wave_sum(float v) {
  v += dpp_mov<0xB1>(v);     // quad_perm [1,0,3,2]
  v += dpp_mov<0x4E>(v);     // quad_perm [2,3,0,1]
  v += dpp_mov<0x141>(v);    // row_half_mirror
  v += dpp_mov<0x140>(v);    // row_mirror
  return xor32_sum(xor16_sum(v));
}
ES_DEVICE float wave_max(float v) {
  v = fmaxf(v, dpp_mov<0xB1>(v));
  v = fmaxf(v, dpp_mov<0x4E>(v));
  v = fmaxf(v, dpp_mov<0x141>(v));
  v = fmaxf(v, dpp_mov<0x140>(v));
  return xor32_max(xor16_max(v));
}

// v_rcp_f32 (1 ulp) instead of an IEEE division (~12 instructions): SiLU sits in the inner loop of GroupNorm-apply,
// the fusion passes and GEMM epilogues, whose outputs are rounded to 16 bits anyway.
ES_DEVICE float silu_f(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
// exact (erf) GELU, as torch.nn.functional.gelu default used by diffusers GEGLU.  erf by Abramowitz-Stegun 7.1.26
// (|abs err| <= 5e-7 against torch's erf GELU over [-12, 12], far below fp16/bf16 resolution), arranged for the GEGLU epilogue:
//   gelu(x) = x * Phi(x),  Phi(x) = x >= 0 ? 1 - h : h,  h = 0.5 * poly(t) * exp(-x^2 / 2),  t = 1 / (1 + p |x| / sqrt 2)
// with the 0.5 folded into the polynomial, the 1/sqrt 2 into p, exp as exp2 of one pre-scaled product: one rcp, one
// exp2, 5 FMA/mul for the polynomial, 5 more ops - 14 issue slots instead of 21 for 0.5 x (1 + erf(x / sqrt 2)).
// (Measured: 505.5 vs 505.6 ms per image - the FF projections are not bound by this epilogue arithmetic.)
// (Round 5 tried this function as single-instruction asm helpers, to keep hipcc's SLP vectoriser from pairing neighbouring evaluations
//  into v_pk_fma_f32 / v_pk_mul_f32: WRONG RESULTS that changed from run to run - hipcc pads the MFMA -> VALU and transcendental -> VALU
//  wait states only for instructions it emits itself, never for the operands of an asm statement, and this function reads matrix-core
//  accumulators and v_rcp / v_exp results.  The packing is switched off per file instead where it costs: Makefile, FLAGS_linear_xs.)
ES_DEVICE float gelu_f(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(ax, 0.3275911f * 0.70710678118654752f, 1.0f));
  const float u = x * 0.84932180028801904f;                       // sqrt(log2(e) / 2): exp(-x^2/2) = exp2(-u^2)
  const float e = __builtin_amdgcn_exp2f(-(u * u));
  float poly = __builtin_fmaf(t, 0.5f * 1.061405429f, 0.5f * -1.453152027f);
  poly = __builtin_fmaf(t, poly, 0.5f * 1.421413741f);
  poly = __builtin_fmaf(t, poly, 0.5f * -0.284496736f);
  poly = __builtin_fmaf(t, poly, 0.5f * 0.254829592f);
  const float h = poly * t * e;                                    // 0.5 * erfc(|x| / sqrt 2)
  // x Phi(x) = x (1 - h) for x >= 0, x h for x < 0  ==  max(x, 0) - |x| h: two instructions instead of compare,
  // subtract, select and multiply (the GEGLU epilogues are bound by their vector-instruction count)
  return __builtin_fmaf(-ax, h, fmaxf(x, 0.f));
}

// transposed LDS read: 16-lane group reads a 4x16 block of 16-bit elements, lane i gets column i (4 rows)
ES_DEVICE u32x2 lds_read_tr16(const void* lds_ptr) {
  h16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) h16x4*)lds_ptr);
  return __builtin_bit_cast(u32x2, v);
}

template <typename T> ES_DEVICE typename Traits<T>::vec8 as_vec8(u32x4 v) {
  return __builtin_bit_cast(typename Traits<T>::vec8, v);
}


// 16-byte global store, optionally write-through (`sc1`): a kernel that ends with dirty lines in its XCD's L2 pays
// their write-back at the kernel boundary (MI355X_MICROARCH.md 'boundary': + bytes / 6 TB/s); write-through stores
// stream out while the kernel still computes.  ES_WT_STORES: 0 = plain, 1 = sc1.
// The `s_nop 1` is REQUIRED: a VMEM store of more than 8 bytes reads its data registers after issue, and the
// instruction after it may overwrite them (hipcc pads that hazard for its own stores, never inside inline asm; the
// version without the nop passed nothing in tests/test_engine_gpu.py).
#ifndef ES_WT_STORES
#define ES_WT_STORES 0
#endif
ES_DEVICE void store16(void* ptr, u32x4 v) {
#if ES_WT_STORES == 2
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(ptr), "v"(v) : "memory");
#elif ES_WT_STORES == 3
  asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" ::"v"(ptr), "v"(v) : "memory");
#elif ES_WT_STORES == 4
  asm volatile("global_store_dwordx4 %0, %1, off sc1 nt\n\ts_nop 1" ::"v"(ptr), "v"(v) : "memory");
#elif ES_WT_STORES
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(ptr), "v"(v) : "memory");
#else
  *(u32x4*)ptr = v;
#endif
}

ES_DEVICE void store8(void* ptr, u32x2 v) {
#if ES_WT_STORES
  asm volatile("global_store_dwordx2 %0, %1, off sc1\n\ts_nop 1" ::"v"(ptr), "v"(v) : "memory");
#else
  *(u32x2*)ptr = v;
#endif
}

// GroupNorm statistics from a GEMM epilogue (es_gemm_desc.gn_part).  `tile` is an LDS image [rows][erow bytes] of the FINAL stored
// values (dtype T) of `rows` consecutive output pixels starting at global pixel m0 and `cols` output channels starting at c0;
// every 64-pixel block and every GroupNorm group that overlaps the tile's channels gets its (sum, sum of squares) written to
//   part[((n * 2 * (HW/64) + 2 * block + slot) * groups + g) * 2 + {0, 1}]
// slot 0 = the tile the group starts in, slot 1 = the next one (zero where the group ends in its first tile).  ONE summation
// order for every caller - 8 sub-blocks of 8 pixels sequentially, the 8 sub-sums sequentially, the group's channels in order - so
// the per-channel sums do not depend on tile shape, wave count or split-K, and a grouped launch equals its per-net launches bit for
// bit (same tile, same planner inputs per layer shape).  ACROSS tile widths the table is equal only up to fp32 rounding: a group
// that straddles an N-tile edge is handed over as slot 0 + slot 1, and where that edge falls depends on the tile width (C = 640,
// 32 groups of 20 channels: a 128-wide tile splits groups 12 + 8, the 320-wide tile and the split-K reduce keep them whole) -
// tests/test_ops_gpu.py::test_group_norm_hand_over_does_not_depend_on_the_tile_or_the_grouping compares the SUM of the two slots
// with a relative tolerance of 2e-6.  (Opt-in feature: ES_GN_HANDOVER; off by default, profiles/r04_gn_handover.txt.)
// `scratch`: LDS floats, (rows / 8 + rows / 64) * cols * 2 of them.  All NT threads call it; it ends behind a barrier-free write.
template <typename T, int NT>
ES_DEVICE void gn_emit_partials(const char* tile, const int erow, const int rows, const int cols, float* scratch, float* part,
                                const int m0, const int M, const int c0, const int Cout, const int HW, const int groups, const int tid) {
  const int oct = cols >> 3;
  float* sub = scratch;                                  // [rows/8][cols][2]
  float* chn = scratch + (rows >> 3) * cols * 2;         // [rows/64][cols][2]
  for (int it = tid; it < (rows >> 3) * oct; it += NT) {
    const int sb = it / oct, o = it - sb * oct;
    float s[8], q[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { s[e] = 0.f; q[e] = 0.f; }
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const auto v = as_vec8<T>(*(const u32x4*)(tile + (sb * 8 + r) * erow + o * 16));
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float f = to_f32(v[e]); s[e] += f; q[e] = __builtin_fmaf(f, f, q[e]); }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) { sub[(sb * cols + o * 8 + e) * 2] = s[e]; sub[(sb * cols + o * 8 + e) * 2 + 1] = q[e]; }
  }
  __syncthreads();
  for (int it = tid; it < (rows >> 6) * cols; it += NT) {
    const int rb = it / cols, c = it - rb * cols;
    float s = 0.f, q = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) { s += sub[((rb * 8 + j) * cols + c) * 2]; q += sub[((rb * 8 + j) * cols + c) * 2 + 1]; }
    chn[it * 2] = s; chn[it * 2 + 1] = q;
  }
  __syncthreads();
  const int cpg = Cout / groups;
  const int g0 = c0 / cpg;                               // first group that overlaps the tile
  int c1 = c0 + cols; if (c1 > Cout) c1 = Cout;
  const int ng = c1 > c0 ? (c1 - 1) / cpg - g0 + 1 : 0;
  const int nch = HW >> 6;
  for (int it = tid; it < (rows >> 6) * ng; it += NT) {
    const int rb = it / ng, g = g0 + (it - rb * ng);
    const int m = m0 + rb * 64;
    if (m >= M) continue;
    const int a = g * cpg > c0 ? g * cpg : c0, b = (g + 1) * cpg < c1 ? (g + 1) * cpg : c1;
    float s = 0.f, q = 0.f;
    for (int c = a; c < b; ++c) { s += chn[(rb * cols + c - c0) * 2]; q += chn[(rb * cols + c - c0) * 2 + 1]; }
    const int n = m / HW, blk = (m - n * HW) >> 6;
    const int slot = g * cpg < c0 ? 1 : 0;
    float* o = part + (((size_t)n * 2 * nch + 2 * blk + slot) * groups + g) * 2;
    o[0] = s; o[1] = q;
    if (!slot && (g + 1) * cpg <= c1) { o[groups * 2] = 0.f; o[groups * 2 + 1] = 0.f; }      // the group ends here: its second entry is zero
  }
}

#define ES_CHECK_LAUNCH() (hipGetLastError() == hipSuccess ? 0 : -2)
