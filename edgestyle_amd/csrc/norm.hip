// GroupNorm(+SiLU) over NHWC and LayerNorm over [M, C] for gfx950 — HBM-bound wavefront-reduction kernels.
//
// GroupNorm: two launches, both with 16-byte vector loads along the contiguous channel dim.
//   1. gn_stats_kernel: grid (nchunk, N).  A block walks its pixel range once; each thread keeps 8 per-channel
//      (sum, sumsq) pairs for its fixed 16-byte channel chunk, then folds them into 32 per-group LDS
//      accumulators -> fp32 partials [N][nchunk][G][2] (deterministic: no global atomics).
//   2. gn_apply_kernel: grid (blocks, N).  Prologue reduces the partials to mean / rstd and builds per-channel
//      scale/shift tables in LDS (gamma*rstd, beta-mean*gamma*rstd), so the streaming loop is one FMA (+SiLU) per
//      element with no integer division.  Two sources (UNet skip concat) are read in place and written as one
//      concatenated, normalised tensor.
// LayerNorm: one wave per row, values held in registers, exact two-pass mean/variance via wave shuffles.
#include <cstdlib>
#include "common.h"
#include "../../include/edgestyle_hip.h"
#include "plan.h"

namespace {

constexpr int GN_MAX_CHUNK = 64;
constexpr int GN_MAX_GROUPS = 64;
#ifndef GN_UNROLL
#define GN_UNROLL 4
#endif
constexpr int GNU = GN_UNROLL;          // independent 16-byte loads in flight per thread in the streaming loops

ES_DEVICE int gn_pixels_per_block(int HW) {
  int ppb = HW / GN_MAX_CHUNK;
  return ppb < 16 ? 16 : ppb;
}

template <typename T>
__global__ __launch_bounds__(256) void gn_stats_kernel(const es_gn_desc p) {
  // LDS: per-(pixel-slot, channel) partial sums, then one thread per group adds them in a FIXED order
  // (bitwise reproducible: no float atomics anywhere).
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int n = blockIdx.y, chunk = blockIdx.x;
  const int C = p.C1 + p.C2, CH8 = C / 8, cpg = C / p.groups;
  const int ppb = gn_pixels_per_block(p.HW);
  const int p0 = chunk * ppb;
  const int p1 = min(p0 + ppb, p.HW);
  const int PS = CH8 <= 256 ? 256 / CH8 : 1;          // pixel slots processed concurrently
  float* csum = (float*)smem;                        // [PS][C]
  float* csq = csum + PS * C;                        // [PS][C]
  for (int q = threadIdx.x; q < PS * CH8; q += 256) {
    const int ps = q / CH8, qq = q - ps * CH8;
    const int c = qq * 8;
    const bool second = c >= p.C1;
    const T* src = second ? (const T*)p.x2 : (const T*)p.x;
    const int cs = second ? p.C2 : p.C1, cc = second ? c - p.C1 : c;
    float s[8], ss[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { s[e] = 0.f; ss[e] = 0.f; }
    // four pixels' loads in flight per thread (a one-load-per-iteration loop pays one memory round trip each);
    // the accumulation order over pixels is unchanged
    int px = p0 + ps;
    for (; px + (GNU - 1) * PS < p1; px += GNU * PS) {
      u32x4 raw[GNU];
#pragma unroll
      for (int u = 0; u < GNU; ++u) raw[u] = *(const u32x4*)(src + ((size_t)n * p.HW + px + u * PS) * cs + cc);
#pragma unroll
      for (int u = 0; u < GNU; ++u) {
        const auto v = as_vec8<T>(raw[u]);
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float f = to_f32(v[e]); s[e] += f; ss[e] += f * f; }
      }
    }
    for (; px < p1; px += PS) {
      const auto v = as_vec8<T>(*(const u32x4*)(src + ((size_t)n * p.HW + px) * cs + cc));
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float f = to_f32(v[e]); s[e] += f; ss[e] += f * f; }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) { csum[ps * C + c + e] = s[e]; csq[ps * C + c + e] = ss[e]; }
  }
  __syncthreads();
  const int nchunk = gridDim.x;
  // L lanes per group walk the group's PS*cpg LDS entries, then a fixed xor tree folds them: no serial tail
  const int L = p.groups <= 32 ? 8 : 4;
  const int gi = threadIdx.x / L, l = threadIdx.x % L;
  float s = 0.f, ss = 0.f;
  if (gi < p.groups) {
    const int items = PS * cpg;
    for (int it = l; it < items; it += L) {
      const int ps = it / cpg, c = gi * cpg + (it - ps * cpg);
      s += csum[ps * C + c];
      ss += csq[ps * C + c];
    }
  }
  // fold the L (4 or 8) lanes of a group: DPP quad steps, then the half-row mirror pairs the two quads (no LDS)
  s += dpp_mov<0xB1>(s); ss += dpp_mov<0xB1>(ss);
  s += dpp_mov<0x4E>(s); ss += dpp_mov<0x4E>(ss);
  if (L == 8) { s += dpp_mov<0x141>(s); ss += dpp_mov<0x141>(ss); }
  if (gi < p.groups && l == 0) {
    float* o = p.partials + ((size_t)n * nchunk + chunk) * p.groups * 2 + gi * 2;
    o[0] = s; o[1] = ss;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void gn_apply_kernel(const es_gn_desc p, const int nchunk) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int n = blockIdx.y;
  const int C = p.C1 + p.C2, CH8 = C / 8, cpg = C / p.groups;
  float* scale = (float*)smem;         // [C]
  float* shift = scale + C;            // [C]
  float* gstat = shift + C;            // [groups*2] mean, rstd
  float* red = gstat + 2 * p.groups;   // [slices][groups*2] partial sums of the partials
  // gamma / beta go into the tables first (raw): their HBM round trip overlaps the partials' one
  {
    const float* gam = p.gamma;
    const float* bet = p.beta;
    if (p.ngroups > 1) {
      const int g = (n >= p.n_end[0]) + (n >= p.n_end[1]) + (n >= p.n_end[2]);
      gam = p.gamma_g[g];
      bet = p.beta_g[g];
    }
    for (int c = threadIdx.x; c < C; c += 256) { scale[c] = gam[c]; shift[c] = bet[c]; }
  }
  {
    // all 256 threads reduce the [nchunk][groups*2] partials: column = (group, stat), rows split into slices;
    // independent loads per thread (no serial latency chain), fixed summation order (bitwise reproducible)
    const int cols = 2 * p.groups;
    const int slices = 256 / cols;
    const int col = threadIdx.x % cols, sl = threadIdx.x / cols;
    if (sl < slices) {
      float acc = 0.f;
      const float* pp = p.partials + (size_t)n * nchunk * cols + col;
      for (int k = sl; k < nchunk; k += slices) acc += pp[(size_t)k * cols];
      red[sl * cols + col] = acc;
    }
    __syncthreads();
    for (int gi = threadIdx.x; gi < p.groups; gi += 256) {
      float s = 0.f, ss = 0.f;
      for (int k = 0; k < slices; ++k) { s += red[k * cols + gi * 2]; ss += red[k * cols + gi * 2 + 1]; }
      const float cnt = (float)cpg * (float)p.HW;
      const float mean = s / cnt;
      float var = ss / cnt - mean * mean;
      var = var < 0.f ? 0.f : var;
      gstat[gi * 2] = mean;
      gstat[gi * 2 + 1] = rsqrtf(var + p.eps);
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {           // same thread wrote scale[c] / shift[c] above
    const int gi = c / cpg;
    const float sc = scale[c] * gstat[gi * 2 + 1];
    scale[c] = sc;
    shift[c] = shift[c] - gstat[gi * 2] * sc;
  }
  __syncthreads();
  const int total = p.HW * CH8;                       // chunks per sample (< 2^31 by the entry-point check)
  const bool small = total < (1 << 24);
  const float inv_ch8 = 1.0f / (float)CH8;
  T* out = (T*)p.out + (size_t)n * p.HW * C;
  // grid-stride loop, four items' loads in flight per thread
  const int gstride = gridDim.x * 256;
  if (gstride % CH8 == 0) {
    // The launcher makes the grid stride a multiple of the chunks per pixel: a thread then stays on ONE 16-byte channel chunk
    // for all of its pixels, so its 8 scale / shift pairs live in registers (read from LDS once) and the pixel index advances
    // by a constant - the loop is load, 8 FMA (+ SiLU), store.  (Per item the general loop below issues four LDS reads and an
    // index division beside one 16-byte global load: instruction-bound at ~2.8 TB/s, like the fusion passes before they were
    // given the same treatment.)  Same arithmetic on the same values: bit-identical results.
    const int i0 = blockIdx.x * 256 + threadIdx.x;
    const int px0 = div_any(i0, CH8, inv_ch8, small);
    const int c = (i0 - px0 * CH8) * 8;
    const int pstep = gstride / CH8;
    const bool second = c >= p.C1;
    const T* src = (second ? (const T*)p.x2 : (const T*)p.x) + (size_t)n * p.HW * (second ? p.C2 : p.C1) + (second ? c - p.C1 : c);
    const int cs = second ? p.C2 : p.C1;
    float sc[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { sc[e] = scale[c + e]; sh[e] = shift[c + e]; }
    T* o = out + c;
    for (int px = px0; px < p.HW; px += GNU * pstep) {
      u32x4 raw[GNU];
#pragma unroll
      for (int u = 0; u < GNU; ++u) {
        const int q = px + u * pstep;
        raw[u] = q < p.HW ? *(const u32x4*)(src + (size_t)q * cs) : u32x4{0u, 0u, 0u, 0u};
      }
#pragma unroll
      for (int u = 0; u < GNU; ++u) {
        const int q = px + u * pstep;
        if (q < p.HW) {
          const auto v = as_vec8<T>(raw[u]);
          typename Traits<T>::vec8 r;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            float f = to_f32(v[e]) * sc[e] + sh[e];
            if (p.silu) f = silu_f(f);
            r[e] = from_f32<T>(f);
          }
          store16(o + (size_t)q * C, __builtin_bit_cast(u32x4, r));
        }
      }
    }
    return;
  }
  for (int i0 = blockIdx.x * 256 + threadIdx.x; i0 < total; i0 += GNU * gstride) {
    u32x4 raw[GNU];
    int pxs[GNU], cs_[GNU];
#pragma unroll
    for (int u = 0; u < GNU; ++u) {
      const int i = i0 + u * gstride;
      raw[u] = u32x4{0u, 0u, 0u, 0u};
      pxs[u] = -1; cs_[u] = 0;
      if (i < total) {
        const int px = div_any(i, CH8, inv_ch8, small);
        const int c = (i - px * CH8) * 8;
        const bool second = c >= p.C1;
        const T* src = second ? (const T*)p.x2 : (const T*)p.x;
        const int cs = second ? p.C2 : p.C1, cc = second ? c - p.C1 : c;
        raw[u] = *(const u32x4*)(src + ((size_t)n * p.HW + px) * cs + cc);
        pxs[u] = px; cs_[u] = c;
      }
    }
#pragma unroll
    for (int u = 0; u < GNU; ++u) {
      if (pxs[u] >= 0) {
        const auto v = as_vec8<T>(raw[u]);
        const int c = cs_[u];
        typename Traits<T>::vec8 r;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float f = to_f32(v[e]) * scale[c + e] + shift[c + e];
          if (p.silu) f = silu_f(f);
          r[e] = from_f32<T>(f);
        }
        store16(out + (size_t)pxs[u] * C + c, __builtin_bit_cast(u32x4, r));
      }
    }
  }
}

// One-launch GroupNorm for slabs that fit a workgroup's registers (the deeper UNet levels): block = (sample, gpb
// consecutive groups); every thread owns one fixed 16-byte channel chunk and up to CPT pixels of it, read ONCE.
// Statistics go through the same deterministic LDS reduction as gn_stats_kernel; the normalised values are written
// straight from the registers.  Saves the second launch and the second read (small tensors are launch-latency bound:
// ~7 us per launch at any size).
template <typename T, int CPT>
__global__ __launch_bounds__(256) void gn_slab_kernel(const es_gn_desc p, const int gpb) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int n = blockIdx.y, gb = blockIdx.x;
  const int C = p.C1 + p.C2, cpg = C / p.groups;
  const int W = gpb * cpg, W8 = W / 8, PS = 256 / W8;
  float* csum = (float*)smem;            // [PS][W]
  float* csq = csum + PS * W;            // [PS][W]
  float* gstat = csq + PS * W;           // [gpb][2] mean, rstd
  const int t = threadIdx.x;
  const int ps = t / W8, cq = t - ps * W8;
  const bool active = ps < PS;
  const int c = gb * W + cq * 8;         // first channel of this thread's chunk
  const bool second = c >= p.C1;
  const T* src = second ? (const T*)p.x2 : (const T*)p.x;
  const int cs = second ? p.C2 : p.C1, cc = second ? c - p.C1 : c;
  const float* gam = p.gamma;
  const float* bet = p.beta;
  if (p.ngroups > 1) {
    const int g = (n >= p.n_end[0]) + (n >= p.n_end[1]) + (n >= p.n_end[2]);
    gam = p.gamma_g[g];
    bet = p.beta_g[g];
  }
  u32x4 raw[CPT];
  f32x4 ga[2], be[2];
  if (active) {
#pragma unroll
    for (int k = 0; k < CPT; ++k) {
      const int px = ps + k * PS;
      raw[k] = u32x4{0u, 0u, 0u, 0u};
      if (px < p.HW) raw[k] = *(const u32x4*)(src + ((size_t)n * p.HW + px) * cs + cc);
    }
    ga[0] = *(const f32x4*)(gam + c); ga[1] = *(const f32x4*)(gam + c + 4);
    be[0] = *(const f32x4*)(bet + c); be[1] = *(const f32x4*)(bet + c + 4);
    float s[8], ss[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { s[e] = 0.f; ss[e] = 0.f; }
#pragma unroll
    for (int k = 0; k < CPT; ++k) {      // pad pixels hold zeros: they add nothing
      const auto v = as_vec8<T>(raw[k]);
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float f = to_f32(v[e]); s[e] += f; ss[e] += f * f; }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) { csum[ps * W + cq * 8 + e] = s[e]; csq[ps * W + cq * 8 + e] = ss[e]; }
  }
  __syncthreads();
  {
    // two fixed-order steps: (1) one thread per channel folds the PS pixel slots (conflict-free LDS columns, no
    // integer division), (2) one wave per local group folds its cpg channel sums with an xor tree
    for (int ch = t; ch < W; ch += 256) {
      float s = 0.f, ss = 0.f;
      for (int q = 0; q < PS; ++q) { s += csum[q * W + ch]; ss += csq[q * W + ch]; }
      csum[ch] = s;                      // row 0 is only read by this same thread above
      csq[ch] = ss;
    }
    __syncthreads();
    const int lg = t >> 6, l = t & 63;
    float s = 0.f, ss = 0.f;
    if (lg < gpb)
      for (int i = l; i < cpg; i += 64) { s += csum[lg * cpg + i]; ss += csq[lg * cpg + i]; }
    s = wave_sum(s);
    ss = wave_sum(ss);
    if (lg < gpb && l == 0) {
      const float cnt = (float)cpg * (float)p.HW;
      const float mean = s / cnt;
      float var = ss / cnt - mean * mean;
      var = var < 0.f ? 0.f : var;
      gstat[lg * 2] = mean;
      gstat[lg * 2 + 1] = rsqrtf(var + p.eps);
    }
  }
  __syncthreads();
  if (active) {
    float sc[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int lg = (cq * 8 + e) / cpg;
      const float g = e < 4 ? ga[0][e & 3] : ga[1][e & 3], b = e < 4 ? be[0][e & 3] : be[1][e & 3];
      sc[e] = g * gstat[lg * 2 + 1];
      sh[e] = b - gstat[lg * 2] * sc[e];
    }
    T* out = (T*)p.out + (size_t)n * p.HW * C + c;
#pragma unroll
    for (int k = 0; k < CPT; ++k) {
      const int px = ps + k * PS;
      if (px < p.HW) {
        const auto v = as_vec8<T>(raw[k]);
        typename Traits<T>::vec8 r;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float f = to_f32(v[e]) * sc[e] + sh[e];
          if (p.silu) f = silu_f(f);
          r[e] = from_f32<T>(f);
        }
        store16(out + (size_t)px * C, __builtin_bit_cast(u32x4, r));
      }
    }
  }
}

// groups per block for the slab kernel (0 = not eligible).  Depends on (HW, C, groups) only, never on N: batched and
// per-net launches of the same layer must take the same path to stay bitwise equal.
int gn_slab_gpb(const es_gn_desc& d, int& cpt) {
  const int C = d.C1 + d.C2, cpg = C / d.groups;
  for (int gpb = 1; gpb <= 4; gpb *= 2) {
    if (d.groups % gpb || (gpb * cpg) % 8) continue;
    const int W8 = gpb * cpg / 8;
    if (W8 > 256) return 0;
    const int PS = 256 / W8;
    cpt = (d.HW + PS - 1) / PS;
    return cpt <= 24 ? gpb : 0;
  }
  return 0;
}

struct LnGroups {
  const float* gamma[4];
  const float* beta[4];
  int row_end[4];
  int ngroups;
};

template <typename T, int VPL /* 16-byte chunks per lane */>
__global__ __launch_bounds__(256) void layer_norm_kernel(const T* __restrict__ x, T* __restrict__ out,
                                                         const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, int M, int C, float eps,
                                                         const LnGroups grp) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  if (grp.ngroups > 1) {
    const int g = (row >= grp.row_end[0]) + (row >= grp.row_end[1]) + (row >= grp.row_end[2]);
    gamma = grp.gamma[g];
    beta = grp.beta[g];
  }
  const int CH8 = C / 8;
  const T* xr = x + (size_t)row * C;
  typename Traits<T>::vec8 v[VPL];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int q = lane + 64 * i;
    if (q < CH8) {
      v[i] = as_vec8<T>(*(const u32x4*)(xr + q * 8));
#pragma unroll
      for (int e = 0; e < 8; ++e) s += to_f32(v[i][e]);
    }
  }
  const float mean = wave_sum(s) / (float)C;
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int q = lane + 64 * i;
    if (q < CH8) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float dlt = to_f32(v[i][e]) - mean; ss += dlt * dlt; }
    }
  }
  const float rstd = rsqrtf(wave_sum(ss) / (float)C + eps);
  T* orow = out + (size_t)row * C;
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int q = lane + 64 * i;
    if (q < CH8) {
      typename Traits<T>::vec8 r;
#pragma unroll
      for (int e = 0; e < 8; ++e)
        r[e] = from_f32<T>((to_f32(v[i][e]) - mean) * rstd * gamma[q * 8 + e] + beta[q * 8 + e]);
      store16(orow + q * 8, __builtin_bit_cast(u32x4, r));
    }
  }
}

template <typename T>
int launch_gn(const es_gn_desc& d, hipStream_t st) {
  const int C = d.C1 + d.C2;
  int cpt = 0;
  static const bool slab_on = !(getenv("ES_GN_SLAB") && getenv("ES_GN_SLAB")[0] == '0');   // tuning switch
  const int gpb = (slab_on && d.ext_chunks <= 0 && !d.stats_only) ? gn_slab_gpb(d, cpt) : 0;
  if (gpb) {
    const int W = gpb * (C / d.groups);
    const size_t lds = (size_t)(2 * (256 / (W / 8)) * W + 2 * gpb) * sizeof(float);
    dim3 grid(d.groups / gpb, d.N);
    if (cpt <= 8) hipLaunchKernelGGL((gn_slab_kernel<T, 8>), grid, dim3(256), lds, st, d, gpb);
    else if (cpt <= 16) hipLaunchKernelGGL((gn_slab_kernel<T, 16>), grid, dim3(256), lds, st, d, gpb);
    else hipLaunchKernelGGL((gn_slab_kernel<T, 24>), grid, dim3(256), lds, st, d, gpb);
    return hipGetLastError() == hipSuccess ? 0 : -2;
  }
  int ppb = d.HW / GN_MAX_CHUNK;
  if (ppb < 16) ppb = 16;
  // ext_chunks > 0: the producing GEMM launch already wrote the per-(sample, group) partial sums (es_gemm_desc.gn_part): the
  // apply pass below reduces them in its prologue exactly as it reduces gn_stats_kernel's - no statistics launch, one read of x
  const int nchunk = d.ext_chunks > 0 ? d.ext_chunks : (d.HW + ppb - 1) / ppb;
  const int CH8 = C / 8;
  const int PS = CH8 <= 256 ? 256 / CH8 : 1;
  const size_t lds_stats = (size_t)2 * PS * C * sizeof(float);
  if (d.ext_chunks <= 0) hipLaunchKernelGGL(gn_stats_kernel<T>, dim3(nchunk, d.N), dim3(256), lds_stats, st, d);
  if (d.stats_only) return hipGetLastError() == hipSuccess ? 0 : -2;      // the consumer normalises (es_xs_desc.gn_part)
  const long long total = (long long)d.HW * (C / 8);
  // workgroups per sample: every workgroup first reduces the partials and builds the scale / shift tables (~2 us), so a thread
  // should stream more than one round of loads behind that prologue - 16 items where that still leaves >= 1024 workgroups for the
  // chip, else 8, else 4 (tools/norm_bench.py, 14 x 64^2 x 320: 32.1 / 27.7 / 28.5 us for 4 / 8 / 16; 112 samples: 215 / 188 / 180)
  static const bool legacy = getenv("ES_GN_LEGACY") && getenv("ES_GN_LEGACY")[0] == '1';     // tool switch: the round-2 geometry
  int ipt = GNU;
  if (!legacy)
    for (int cand : {16, 8}) if (((total + 256 * cand - 1) / (256 * cand)) * d.N >= 1024) { ipt = cand; break; }
  int blocks = (int)((total + 256 * ipt - 1) / (256 * ipt));
  if (blocks < 1) blocks = 1;
  if (blocks > 1024) blocks = 1024;
  if (!legacy) {   // a multiple of q = CH8 / gcd(CH8, 256) workgroups: the grid stride is then a multiple of the chunks per pixel (gn_apply_kernel)
    int a = CH8, b = 256;
    while (b) { const int t = a % b; a = b; b = t; }
    const int q = CH8 / a;
    if (q <= 64) blocks = blocks < q ? q : blocks / q * q;
  }
  const size_t lds = (size_t)(2 * C + 2 * d.groups + 256) * sizeof(float);
  hipLaunchKernelGGL(gn_apply_kernel<T>, dim3(blocks, d.N), dim3(256), lds, st, d, nchunk);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

template <typename T>
int launch_ln(const void* x, void* out, const float* gamma, const float* beta, int M, int C, float eps,
              hipStream_t st, const LnGroups grp = LnGroups{}) {
  const int CH8 = C / 8;
  const int vpl = (CH8 + 63) / 64;
  dim3 grid((M + 3) / 4);
#define ES_LN(V) hipLaunchKernelGGL((layer_norm_kernel<T, V>), grid, dim3(256), 0, st, (const T*)x, (T*)out, gamma, beta, M, C, eps, grp)
  if (vpl <= 1) ES_LN(1);
  else if (vpl <= 2) ES_LN(2);
  else if (vpl <= 3) ES_LN(3);
  else if (vpl <= 4) ES_LN(4);
  else if (vpl <= 8) ES_LN(8);
  else return -3;
#undef ES_LN
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

}  // namespace

extern "C" void es_set_error(const char* msg);

extern "C" size_t es_group_norm_partials_bytes(int N, int groups) {
  return (size_t)N * GN_MAX_CHUNK * groups * 2 * sizeof(float);
}

extern "C" int es_group_norm_chunks(int HW) {          // pixel chunks per sample of the statistics pass: the second extent of `partials`
  if (HW < 1) return 0;
  int ppb = HW / GN_MAX_CHUNK;
  if (ppb < 16) ppb = 16;
  return (HW + ppb - 1) / ppb;
}

extern "C" int es_group_norm_is_slab(int HW, int C, int groups) {
  if (HW < 1 || C < 8 || groups < 1 || C % groups) return 0;
  static const bool slab_on = !(getenv("ES_GN_SLAB") && getenv("ES_GN_SLAB")[0] == '0');
  es_gn_desc d{};
  d.HW = HW; d.C1 = C; d.groups = groups;
  int cpt = 0;
  return slab_on && gn_slab_gpb(d, cpt) ? 1 : 0;
}

extern "C" int es_group_norm(const es_gn_desc* d, void* stream) {
  const int C = d->C1 + d->C2;
  if (d->stats_only) {
    if (!d->x || !d->partials || d->ext_chunks) { es_set_error("es_group_norm: stats_only needs x and partials (and no ext_chunks)"); return -1; }
  } else {
    if (!d->x || !d->out || !d->partials || (d->ngroups <= 1 && (!d->gamma || !d->beta))) { es_set_error("es_group_norm: null pointer"); return -1; }
    for (int g = 0; g < d->ngroups && d->ngroups > 1; ++g)
      if (!d->gamma_g[g] || !d->beta_g[g]) { es_set_error("es_group_norm: null group parameter"); return -1; }
  }
  if (d->ngroups > 4) { es_set_error("es_group_norm: at most 4 groups"); return -1; }
  if (d->ngroups > 1 && d->n_end[d->ngroups - 1] != d->N) { es_set_error("es_group_norm: group table must cover N"); return -1; }
  if (d->C1 % 8 || d->C2 % 8 || (d->C2 && !d->x2)) { es_set_error("es_group_norm: channels must be multiples of 8"); return -1; }
  if (d->groups < 1 || d->groups > GN_MAX_GROUPS || C % d->groups) { es_set_error("es_group_norm: bad group count"); return -1; }
  if (C > 8192) { es_set_error("es_group_norm: C too large for the LDS tables"); return -1; }
  if (d->N < 1 || d->HW < 1) { es_set_error("es_group_norm: empty problem"); return -1; }
  if ((long long)d->HW * (C / 8) >= (1ll << 30)) { es_set_error("es_group_norm: sample too large for 32-bit chunk indices"); return -1; }
  if (d->ext_chunks < 0 || (d->ext_chunks > 0 && (d->x2 || d->C2 || d->HW % 64 || d->ext_chunks != 2 * (d->HW / 64)))) {
    es_set_error("es_group_norm: ext_chunks (statistics from the producer) needs one source, H*W % 64 == 0 and 2 * H*W / 64 entries"); return -1; }
  ES_PLAN_RECORD(ES_OP_GROUP_NORM, d, sizeof(*d));
  hipStream_t st = (hipStream_t)stream;
  es_gn_desc dd = *d;                                  // unused group-table entries must compare false (see es_conv_gemm)
  for (int g = dd.ngroups > 1 ? dd.ngroups : 0; g < 4; ++g) dd.n_end[g] = 0x7FFFFFFF;
  int rc = dd.dtype == ES_F16 ? launch_gn<f16>(dd, st) : launch_gn<bf16>(dd, st);
  if (rc) es_set_error("es_group_norm: launch failed");
  return rc;
}

extern "C" int es_layer_norm(const void* x, void* out, const float* gamma, const float* beta, int M, int C,
                             float eps, int dtype, void* stream) {
  if (!x || !out || !gamma || !beta) { es_set_error("es_layer_norm: null pointer"); return -1; }
  if (C % 8 || M < 1) { es_set_error("es_layer_norm: C must be a multiple of 8"); return -1; }
  if (es_plan_recording()) {
    const es_op_layer_norm a{x, out, gamma, beta, M, C, eps, dtype};
    es_plan_record(ES_OP_LAYER_NORM, &a, sizeof(a));
    ES_PLAN_DRY_RETURN();
  }
  hipStream_t st = (hipStream_t)stream;
  int rc = dtype == ES_F16 ? launch_ln<f16>(x, out, gamma, beta, M, C, eps, st)
                           : launch_ln<bf16>(x, out, gamma, beta, M, C, eps, st);
  if (rc == -3) es_set_error("es_layer_norm: C > 4096 unsupported");
  else if (rc) es_set_error("es_layer_norm: launch failed");
  return rc;
}

extern "C" int es_layer_norm_grouped(const es_ln_desc* d, void* stream) {
  if (!d->x || !d->out || d->C % 8 || d->M < 1 || d->ngroups < 1 || d->ngroups > 4) { es_set_error("es_layer_norm_grouped: bad arguments"); return -1; }
  LnGroups grp;
  grp.ngroups = d->ngroups;
  for (int g = 0; g < 4; ++g) {
    grp.gamma[g] = d->gamma_g[g < d->ngroups ? g : 0];
    grp.beta[g] = d->beta_g[g < d->ngroups ? g : 0];
    grp.row_end[g] = g < d->ngroups ? d->row_end[g] : d->M;
    if (!grp.gamma[g] || !grp.beta[g]) { es_set_error("es_layer_norm_grouped: null parameter"); return -1; }
  }
  if (d->row_end[d->ngroups - 1] != d->M) { es_set_error("es_layer_norm_grouped: group table must cover M"); return -1; }
  ES_PLAN_RECORD(ES_OP_LAYER_NORM_GROUPED, d, sizeof(*d));
  hipStream_t st = (hipStream_t)stream;
  int rc = d->dtype == ES_F16 ? launch_ln<f16>(d->x, d->out, grp.gamma[0], grp.beta[0], d->M, d->C, d->eps, st, grp)
                              : launch_ln<bf16>(d->x, d->out, grp.gamma[0], grp.beta[0], d->M, d->C, d->eps, st, grp);
  if (rc == -3) es_set_error("es_layer_norm_grouped: C > 4096 unsupported");
  else if (rc) es_set_error("es_layer_norm_grouped: launch failed");
  return rc;
}
