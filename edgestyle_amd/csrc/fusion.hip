// EdgeStyle fusion block for gfx950: interleave_tensors (model/edgestyle_multicontrolnet.py:479-501) +
// ControlNetBlock (model/edgestyle_multicontrolnet.py:23-63) with the interleaved [N,6C,H,W] tensor never
// materialised.  With the reference's channel interleave k = c*6 + net and its grouped 1x1 convs, output channel c
// depends only on the SAME channel c of the six nets:
//   z_p = w1[c,p,0]*r[2p] + w1[c,p,1]*r[2p+1] + b1[c,p]                      p = 0..2   (first_conv, groups=3C)
//   y_p = silu( LN1(z)[c,p,h,w] )      LN1 over all (3C,H,W) of a sample, affine planes [3C,H,W]
//   u   = sum_p w2[c,p]*y_p + b2[c]                                          (second_conv, groups=C)
//   v   = silu( LN2(u)[c,h,w] )        LN2 over (C,H,W), affine planes [C,H,W]
//   out = w3[c]*v + b3[c]                                                    (third_conv, depthwise)
// HBM-bound: three streaming passes (stats of z; y,u + stats of u; output), 16-byte loads along C from the six
// NHWC residual tensors in place, fp32 statistics, deterministic two-level reductions (no global atomics).
#include "common.h"
#include "../../include/edgestyle_hip.h"
#include "plan.h"

namespace {

constexpr int FU_MAX_CHUNK = 256;

ES_DEVICE float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// The per-channel parameters of one 8-channel chunk: w1 [8][3][2], b1 [8][3], w2 [8][3], b2 [8] - 104 floats.  Loading them
// per item element by element (26 loads beside the 12 that move data) made the passes instruction-bound: 2.7 TB/s of moved
// bytes against 4.8 with the parameters replaced by constants (tools/fusion_bench.py).  The grids are chosen so that the grid
// stride is a multiple of C / 8: a thread's chunk column then never changes and its parameters are loaded ONCE, ahead of its
// loop (`fixc`); any other geometry reloads them per item, as 16-byte loads.
struct ChunkParams {
  float w1[8][3][2], b1[8][3], w2[8][3], b2[8];
};
ES_DEVICE void load_chunk_params(const es_fusion_desc& p, int c, ChunkParams& cp, bool second) {
  const f32x4* w = (const f32x4*)(p.w1 + (size_t)c * 6);       // c % 8 == 0: 16-byte aligned
#pragma unroll
  for (int k = 0; k < 12; ++k) {
    const f32x4 v = w[k];
#pragma unroll
    for (int r = 0; r < 4; ++r) (&cp.w1[0][0][0])[k * 4 + r] = v[r];
  }
  const f32x4* b = (const f32x4*)(p.b1 + (size_t)c * 3);
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const f32x4 v = b[k];
#pragma unroll
    for (int r = 0; r < 4; ++r) (&cp.b1[0][0])[k * 4 + r] = v[r];
  }
  if (second) {
    const f32x4* w2 = (const f32x4*)(p.w2 + (size_t)c * 3);
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const f32x4 v = w2[k];
#pragma unroll
      for (int r = 0; r < 4; ++r) (&cp.w2[0][0])[k * 4 + r] = v[r];
    }
    const f32x4 b0 = *(const f32x4*)(p.b2 + c), b1v = *(const f32x4*)(p.b2 + c + 4);
#pragma unroll
    for (int r = 0; r < 4; ++r) { cp.b2[r] = b0[r]; cp.b2[4 + r] = b1v[r]; }
  }
}

template <typename T>
ES_DEVICE void load_z(const es_fusion_desc& p, int n, size_t off, const ChunkParams& cp, float z[3][8]) {
  // off = pix*C + c  (element offset inside one sample of one net)
  float r[6][8];
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const auto v = as_vec8<T>(*(const u32x4*)((const T*)p.res[k] + (size_t)n * p.res_bs[k] + off));
    float sc = p.res_scale[k];
    if (p.res_scale_dev) sc *= p.res_scale_dev[k];
#pragma unroll
    for (int e = 0; e < 8; ++e) r[k][e] = to_f32(v[e]) * sc;
  }
#pragma unroll
  for (int e = 0; e < 8; ++e)
#pragma unroll
    for (int q = 0; q < 3; ++q) z[q][e] = cp.w1[e][q][0] * r[2 * q][e] + cp.w1[e][q][1] * r[2 * q + 1][e] + cp.b1[e][q];
}

ES_DEVICE void reduce_partials(const float* part, int nchunk, float cnt, float eps, float& mean, float& rstd,
                               float* red /* LDS [4] */) {
  // wave 0 loads the (<= 64) partial pairs in parallel and reduces them with shuffles (fixed order): no per-thread
  // serial chain of dependent global loads in front of the streaming loop
  if (threadIdx.x < 64) {
    float s = 0.f, ss = 0.f;
    for (int k = threadIdx.x; k < nchunk; k += 64) { s += part[k * 2]; ss += part[k * 2 + 1]; }   // <= 4 per lane
    s = wave_sum(s);
    ss = wave_sum(ss);
    if (threadIdx.x == 0) { red[0] = s; red[1] = ss; }
  }
  __syncthreads();
  const float s = red[0], ss = red[1];
  mean = s / cnt;
  float var = ss / cnt - mean * mean;
  var = var < 0.f ? 0.f : var;
  rstd = rsqrtf(var + eps);
  __syncthreads();
}

// The pass bodies take (block index bx of nb) explicitly so that one launch can cover one fusion block (grid.x = nb)
// or all 13 of a denoising step (FusionBatch: grid.x = sum of the blocks' nb, looked up in a prefix table).
struct FusionBatch {
  es_fusion_desc d[ES_FUSION_MAX_BATCH];
  int ab_end[ES_FUSION_MAX_BATCH];     // prefix sums of the pass A/B grids
  int c_end[ES_FUSION_MAX_BATCH];      // prefix sums of the pass C grids
  int count;
};

ES_DEVICE int batch_lookup(const int* end, int count, int b, int& bx, int& nb) {
  int k = 0;
  while (k + 1 < count && b >= end[k]) ++k;
  const int start = k ? end[k - 1] : 0;
  bx = b - start;
  nb = end[k] - start;
  return k;
}

template <typename T>
ES_DEVICE void fusion_a_body(const es_fusion_desc& p, const int bx, const int nb, const int n) {
  __shared__ float red[4];
  const int CH8 = p.C / 8;
  const int items = p.HW * CH8;
  const bool small = items < (1 << 24);
  const float inv_ch8 = 1.0f / (float)CH8;
  float s = 0.f, ss = 0.f;
  const int i0 = bx * 256 + threadIdx.x;
  const bool fixc = (nb * 256) % CH8 == 0;               // the thread's chunk column is the same for every item
  ChunkParams cp;
  if (fixc && i0 < items) load_chunk_params(p, (i0 - div_any(i0, CH8, inv_ch8, small) * CH8) * 8, cp, false);
  for (int i = i0; i < items; i += nb * 256) {
    if (!fixc) load_chunk_params(p, (i - div_any(i, CH8, inv_ch8, small) * CH8) * 8, cp, false);
    float z[3][8];
    load_z<T>(p, n, (size_t)i * 8, cp, z);
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
      for (int e = 0; e < 8; ++e) { s += z[q][e]; ss += z[q][e] * z[q][e]; }
  }
  s = block_sum(s, red);
  ss = block_sum(ss, red);
  if (threadIdx.x == 0) {
    float* o = p.scratch + (((size_t)n * 2 + 0) * FU_MAX_CHUNK + bx) * 2;
    o[0] = s; o[1] = ss;
  }
}

template <typename T>
ES_DEVICE void fusion_b_body(const es_fusion_desc& p, const int bx, const int nchunk, const int n) {
  __shared__ float red[4];
  const int CH8 = p.C / 8;
  const int items = p.HW * CH8;
  const bool small = items < (1 << 24);
  const float inv_ch8 = 1.0f / (float)CH8;
  float mean1, rstd1;
  reduce_partials(p.scratch + ((size_t)n * 2 + 0) * FU_MAX_CHUNK * 2, nchunk, 3.f * (float)p.C * (float)p.HW,
                  p.eps, mean1, rstd1, red);
  float s = 0.f, ss = 0.f;
  T* U = (T*)p.u + (size_t)n * p.HW * p.C;
  const int i0 = bx * 256 + threadIdx.x;
  const bool fixc = (nchunk * 256) % CH8 == 0;
  ChunkParams cp;
  if (fixc && i0 < items) load_chunk_params(p, (i0 - div_any(i0, CH8, inv_ch8, small) * CH8) * 8, cp, true);
  for (int i = i0; i < items; i += nchunk * 256) {
    if (!fixc) load_chunk_params(p, (i - div_any(i, CH8, inv_ch8, small) * CH8) * 8, cp, true);
    float z[3][8];
    load_z<T>(p, n, (size_t)i * 8, cp, z);
    // affine planes: [(pix*C + c)*3 + q], 24 contiguous values for this thread
    const T* g1 = (const T*)p.g1 + (size_t)i * 24;
    const T* be1 = (const T*)p.be1 + (size_t)i * 24;
    float gv[24], bv[24];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const auto a = as_vec8<T>(*(const u32x4*)(g1 + k * 8));
      const auto b = as_vec8<T>(*(const u32x4*)(be1 + k * 8));
#pragma unroll
      for (int e = 0; e < 8; ++e) { gv[k * 8 + e] = to_f32(a[e]); bv[k * 8 + e] = to_f32(b[e]); }
    }
    typename Traits<T>::vec8 uo;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float u = cp.b2[e];
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const float y = silu_f((z[q][e] - mean1) * rstd1 * gv[e * 3 + q] + bv[e * 3 + q]);
        u += cp.w2[e][q] * y;
      }
      uo[e] = from_f32<T>(u);
      const float ur = to_f32(uo[e]);      // statistics of the value pass C will actually read
      s += ur; ss += ur * ur;
    }
    store16(U + (size_t)i * 8, __builtin_bit_cast(u32x4, uo));
  }
  s = block_sum(s, red);
  ss = block_sum(ss, red);
  if (threadIdx.x == 0) {
    float* o = p.scratch + (((size_t)n * 2 + 1) * FU_MAX_CHUNK + bx) * 2;
    o[0] = s; o[1] = ss;
  }
}

template <typename T>
ES_DEVICE void fusion_c_body(const es_fusion_desc& p, const int bx, const int nb, const int nchunk, const int n) {
  const int CH8 = p.C / 8;
  const int items = p.HW * CH8;
  const bool small = items < (1 << 24);
  const float inv_ch8 = 1.0f / (float)CH8;
  float mean2, rstd2;
  __shared__ float red[4];
  reduce_partials(p.scratch + ((size_t)n * 2 + 1) * FU_MAX_CHUNK * 2, nchunk, (float)p.C * (float)p.HW, p.eps,
                  mean2, rstd2, red);
  const T* U = (const T*)p.u + (size_t)n * p.HW * p.C;
  T* O = (T*)p.out + (size_t)n * p.HW * p.C;
  const int i0 = bx * 256 + threadIdx.x;
  const bool fixc = (nb * 256) % CH8 == 0;
  float w3v[8], b3v[8];
  auto load_w3 = [&](int c) __attribute__((always_inline)) {
    const f32x4 wa = *(const f32x4*)(p.w3 + c), wb = *(const f32x4*)(p.w3 + c + 4);
    const f32x4 ba = *(const f32x4*)(p.b3 + c), bb = *(const f32x4*)(p.b3 + c + 4);
#pragma unroll
    for (int r = 0; r < 4; ++r) { w3v[r] = wa[r]; w3v[4 + r] = wb[r]; b3v[r] = ba[r]; b3v[4 + r] = bb[r]; }
  };
  if (fixc && i0 < items) load_w3((i0 - div_any(i0, CH8, inv_ch8, small) * CH8) * 8);
  for (int i = i0; i < items; i += nb * 256) {
    if (!fixc) load_w3((i - div_any(i, CH8, inv_ch8, small) * CH8) * 8);
    const auto u = as_vec8<T>(*(const u32x4*)(U + (size_t)i * 8));
    const auto g = as_vec8<T>(*(const u32x4*)((const T*)p.g2 + (size_t)i * 8));
    const auto b = as_vec8<T>(*(const u32x4*)((const T*)p.be2 + (size_t)i * 8));
    u32x4 araw = {0u, 0u, 0u, 0u};
    if (p.addend) araw = *(const u32x4*)((const T*)p.addend + ((size_t)n * p.HW * p.C) + (size_t)i * 8);
    const auto a = as_vec8<T>(araw);
    typename Traits<T>::vec8 r;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float v = silu_f((to_f32(u[e]) - mean2) * rstd2 * to_f32(g[e]) + to_f32(b[e]));
      r[e] = from_f32<T>(w3v[e] * v + b3v[e]);
      if (p.addend) r[e] = from_f32<T>(to_f32(r[e]) + to_f32(a[e]));      // == es_add(out, addend)
    }
    store16(O + (size_t)i * 8, __builtin_bit_cast(u32x4, r));
  }
}

template <typename T>
__global__ __launch_bounds__(256) void fusion_pass_a(const es_fusion_desc p) { fusion_a_body<T>(p, blockIdx.x, gridDim.x, blockIdx.y); }
template <typename T>
__global__ __launch_bounds__(256) void fusion_pass_b(const es_fusion_desc p) { fusion_b_body<T>(p, blockIdx.x, gridDim.x, blockIdx.y); }
template <typename T>
__global__ __launch_bounds__(256) void fusion_pass_c(const es_fusion_desc p, const int nchunk) {
  fusion_c_body<T>(p, blockIdx.x, gridDim.x, nchunk, blockIdx.y);
}

template <typename T>
__global__ __launch_bounds__(256) void fusion_batch_a(const FusionBatch b) {
  int bx, nb;
  const int k = batch_lookup(b.ab_end, b.count, blockIdx.x, bx, nb);
  fusion_a_body<T>(b.d[k], bx, nb, blockIdx.y);
}
template <typename T>
__global__ __launch_bounds__(256) void fusion_batch_b(const FusionBatch b) {
  int bx, nb;
  const int k = batch_lookup(b.ab_end, b.count, blockIdx.x, bx, nb);
  fusion_b_body<T>(b.d[k], bx, nb, blockIdx.y);
}
template <typename T>
__global__ __launch_bounds__(256) void fusion_batch_c(const FusionBatch b) {
  int bx, nb;
  const int k = batch_lookup(b.c_end, b.count, blockIdx.x, bx, nb);
  fusion_c_body<T>(b.d[k], bx, nb, b.ab_end[k] - (k ? b.ab_end[k - 1] : 0), blockIdx.y);
}

int gcd_i(int a, int b) { while (b) { const int t = a % b; a = b; b = t; } return a; }

// Workgroups per sample of passes A / B (nchunk: also the number of partial sums) and C (cb).  Both are multiples of
// q = (C / 8) / gcd(C / 8, 256) - 5 for SD1.5's 320 | 640 | 1280 channels - so that the grid stride (256 threads x workgroups)
// is a multiple of C / 8 and every thread keeps ONE chunk column: its per-channel parameters are loaded once.  ~8 items per
// thread in A / B (the 104 parameter floats amortise), ~4 in C.
void fusion_grids(const es_fusion_desc& d, int& nchunk, int& cb) {
  const long long items = (long long)d.HW * (d.C / 8);
  const int q = (d.C / 8) / gcd_i(d.C / 8, 256);
  auto pick = [&](long long want, int cap) {
    int n = (int)(want / q) * q;
    if (n < q) n = q;
    while (n > cap) n -= q;
    return n < 1 ? (int)(want < 1 ? 1 : (want > cap ? cap : want)) : n;      // q > cap: no fixed column (per-item reloads)
  };
  nchunk = pick(items / 2048, FU_MAX_CHUNK);
  cb = pick(items / 1024, 512);
}

template <typename T>
int launch_fusion(const es_fusion_desc& d, hipStream_t st) {
  int nchunk, cb;
  fusion_grids(d, nchunk, cb);
  dim3 grid(nchunk, d.N);
  hipLaunchKernelGGL(fusion_pass_a<T>, grid, dim3(256), 0, st, d);
  hipLaunchKernelGGL(fusion_pass_b<T>, grid, dim3(256), 0, st, d);
  hipLaunchKernelGGL(fusion_pass_c<T>, dim3(cb, d.N), dim3(256), 0, st, d, nchunk);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

template <typename T>
int launch_fusion_batch(const es_fusion_desc* ds, int count, hipStream_t st) {
  FusionBatch b;
  int ab = 0, c = 0;
  for (int k = 0; k < count; ++k) {
    int nchunk, cb;
    fusion_grids(ds[k], nchunk, cb);
    b.d[k] = ds[k];
    b.ab_end[k] = (ab += nchunk);
    b.c_end[k] = (c += cb);
  }
  for (int k = count; k < ES_FUSION_MAX_BATCH; ++k) { b.d[k] = ds[0]; b.ab_end[k] = ab; b.c_end[k] = c; }
  b.count = count;
  hipLaunchKernelGGL(fusion_batch_a<T>, dim3(ab, ds[0].N), dim3(256), 0, st, b);
  hipLaunchKernelGGL(fusion_batch_b<T>, dim3(ab, ds[0].N), dim3(256), 0, st, b);
  hipLaunchKernelGGL(fusion_batch_c<T>, dim3(c, ds[0].N), dim3(256), 0, st, b);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

}  // namespace

extern "C" void es_set_error(const char* msg);

extern "C" size_t es_fusion_scratch_bytes(int N) { return (size_t)N * 2 * FU_MAX_CHUNK * 2 * sizeof(float); }

static int fusion_check(const es_fusion_desc* d) {
  for (int i = 0; i < 6; ++i)
    if (!d->res[i]) { es_set_error("es_fusion_block: null residual pointer"); return -1; }
  if (!d->w1 || !d->b1 || !d->g1 || !d->be1 || !d->w2 || !d->b2 || !d->g2 || !d->be2 || !d->w3 || !d->b3 ||
      !d->scratch || !d->u || !d->out) { es_set_error("es_fusion_block: null pointer"); return -1; }
  if (d->C % 8 || d->N < 1 || d->HW < 1) { es_set_error("es_fusion_block: C must be a multiple of 8"); return -1; }
  if ((long long)d->HW * (d->C / 8) >= (1ll << 30)) { es_set_error("es_fusion_block: sample too large for 32-bit chunk indices"); return -1; }
  return 0;
}

extern "C" int es_fusion_blocks(const es_fusion_desc* ds, int count, void* stream) {
  if (!ds || count < 1 || count > ES_FUSION_MAX_BATCH) { es_set_error("es_fusion_blocks: 1..13 blocks per call"); return -1; }
  for (int k = 0; k < count; ++k) {
    if (fusion_check(ds + k)) return -1;
    if (ds[k].N != ds[0].N || ds[k].dtype != ds[0].dtype) { es_set_error("es_fusion_blocks: blocks must share N and dtype"); return -1; }
    for (int j = 0; j < k; ++j)
      if (ds[j].scratch == ds[k].scratch || ds[j].u == ds[k].u) { es_set_error("es_fusion_blocks: blocks need private scratch and u buffers"); return -1; }
  }
  ES_PLAN_RECORD(ES_OP_FUSION_BLOCKS, ds, sizeof(*ds) * count);
  hipStream_t st = (hipStream_t)stream;
  int rc = ds[0].dtype == ES_F16 ? launch_fusion_batch<f16>(ds, count, st) : launch_fusion_batch<bf16>(ds, count, st);
  if (rc) es_set_error("es_fusion_blocks: launch failed");
  return rc;
}

extern "C" int es_fusion_block(const es_fusion_desc* d, void* stream) {
  if (fusion_check(d)) return -1;
  ES_PLAN_RECORD(ES_OP_FUSION_BLOCK, d, sizeof(*d));
  hipStream_t st = (hipStream_t)stream;
  int rc = d->dtype == ES_F16 ? launch_fusion<f16>(*d, st) : launch_fusion<bf16>(*d, st);
  if (rc) es_set_error("es_fusion_block: launch failed");
  return rc;
}
