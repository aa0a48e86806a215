// Implicit-GEMM convolution / linear for gfx950 on MFMA 16x16x32 (f16|bf16 in, fp32 accumulate).
//
//   D[cout][pixel] = sum_k W[cout][k] * X[pixel][k]      (swapped orientation: A operand = weights,
//                                                         B operand = im2col'd activations)
// so every lane ends up with 4 CONSECUTIVE output channels of one pixel -> 8-byte NHWC stores, and the
// per-channel epilogue terms (bias, time-embedding add) are 4-wide vector loads.
//
// Tile: BM=128 pixels x BN (128|160) couts x BK=64, 256 threads = 4 waves as 2(M) x 2(N); each wave owns
// 64 pixels x BN/2 couts = 4 x (4|5) MFMA fragments.  Operands are staged global -> registers -> LDS
// (16-byte chunks, XOR-swizzled rows of 128 B so ds_read_b128 fragment reads are bank-conflict free),
// double-buffered in LDS with the next tile's global loads in flight behind the current tile's MFMAs;
// one barrier per K-step.  The activation loader does the im2col on the fly: 3x3/1x1, stride 1|2,
// nearest-2x upsample, and channel-concat of two sources (UNet skip connections) are address math only.
#include "common.h"
#include "../../include/edgestyle_hip.h"

namespace {

constexpr int BM = 128;
constexpr int BK = 64;

struct EpiArgs {
  const float* bias;
  const void* temb;
  const void* residual;
  const float* out_scale_dev;
  void* out;
  int M, Cout, Cstore, HWout, temb_stride, act;
  float out_scale;
};

template <typename T>
ES_DEVICE void epilogue_quad(const EpiArgs& e, int m, int c0, float v[4], float scale) {
  // plain / SiLU epilogue for 4 consecutive channels c0..c0+3 of pixel m
  if (m >= e.M || c0 >= e.Cout) return;
  const int n = m / e.HWout;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int c = c0 + r;
    if (c < e.Cout) {
      float x = v[r];
      if (e.bias) x += e.bias[c];
      if (e.temb) x += to_f32(((const T*)e.temb)[(size_t)n * e.temb_stride + c]);
      if (e.act == ES_ACT_SILU) x = silu_f(x);
      x *= scale;
      if (e.residual) x += to_f32(((const T*)e.residual)[(size_t)m * e.Cstore + c]);
      v[r] = x;
    }
  }
  T* o = (T*)e.out + (size_t)m * e.Cstore + c0;
  if (c0 + 3 < e.Cout && (e.Cstore & 3) == 0) {
    typename Traits<T>::vec4 pk;
#pragma unroll
    for (int r = 0; r < 4; ++r) pk[r] = from_f32<T>(v[r]);
    *(typename Traits<T>::vec4*)o = pk;
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (c0 + r < e.Cout) o[r] = from_f32<T>(v[r]);
  }
}

template <typename T>
ES_DEVICE void epilogue_geglu_quad(const EpiArgs& e, int m, int c_hidden, float h[4], float g[4], float scale) {
  // packed weight rows: [.. 16 hidden | 16 gate ..]; c_hidden = packed row of the hidden quad, gate = +16
  if (m >= e.M || c_hidden >= e.Cout) return;
  const int blk = c_hidden >> 5, within = c_hidden & 31;      // within < 16
  const int oc = blk * 16 + within;                          // output column in [0, Cout/2)
  typename Traits<T>::vec4 pk;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float hv = h[r], gv = g[r];
    if (e.bias) { hv += e.bias[c_hidden + r]; gv += e.bias[c_hidden + 16 + r]; }
    float x = hv * gelu_f(gv) * scale;
    if (e.residual) x += to_f32(((const T*)e.residual)[(size_t)m * e.Cstore + oc + r]);
    pk[r] = from_f32<T>(x);
  }
  *(typename Traits<T>::vec4*)((T*)e.out + (size_t)m * e.Cstore + oc) = pk;
}

template <typename T, int BN, bool ALIGNED>
__global__ __launch_bounds__(256, 2) void conv_gemm_kernel(const es_gemm_desc p, const int M, const int nk) {
  constexpr int FM = 4;            // pixel fragments per wave (64 pixels)
  constexpr int FN = BN / 32;      // cout fragments per wave (BN/2 couts)
  constexpr int WROWS = BN / 32;   // weight rows per loader thread
  constexpr int XT = BM * BK * 2;  // bytes
  constexpr int WT = BN * BK * 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave & 1, wn = wave >> 1;
  const int tile_m = blockIdx.x, tile_n = blockIdx.y, z = blockIdx.z;

  // split-K range
  const int ks0 = (int)(((long long)nk * z) / p.splitk);
  const int ks1 = (int)(((long long)nk * (z + 1)) / p.splitk);

  // ---------------- loader state ----------------
  const int kc = tid & 7;
  const int r0 = tid >> 3;
  const int swz = (kc ^ (r0 & 7)) << 4;
  const int Ctot = p.C1 + p.C2;
  const int Ktrue = p.ksize * p.ksize * Ctot;
  const int Hin = p.Hsrc << p.upsample, Win = p.Wsrc << p.upsample;
  const int HWout = p.Hout * p.Wout;

  int iy0[4], ix0[4], nb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = tile_m * BM + r0 + 32 * i;
    if (m < M) {
      const int n = m / HWout;
      const int rem = m - n * HWout;
      const int oy = rem / p.Wout, ox = rem - oy * p.Wout;
      iy0[i] = oy * p.stride - p.pad;
      ix0[i] = ox * p.stride - p.pad;
      nb[i] = n * p.Hsrc * p.Wsrc;
    } else {
      iy0[i] = -(1 << 20); ix0[i] = -(1 << 20); nb[i] = 0;
    }
  }
  const T* wbase = (const T*)p.w + (size_t)(tile_n * BN + r0) * p.Kpad + kc * 8;

  // running (tap, channel) position of this thread's chunk
  int tap, cpos;
  {
    const int kg = ks0 * BK + (ALIGNED ? 0 : kc * 8);
    tap = kg / Ctot;
    cpos = kg - tap * Ctot;
  }

  u32x4 xr[4], wr[WROWS];
  auto load_tile = [&](int ks) {
    // weights
#pragma unroll
    for (int i = 0; i < WROWS; ++i) wr[i] = *(const u32x4*)(wbase + (size_t)(32 * i) * p.Kpad + (size_t)ks * BK);
    // activations (im2col on the fly)
    const int c = ALIGNED ? cpos + kc * 8 : cpos;
    const bool kvalid = ALIGNED ? true : (ks * BK + kc * 8 < Ktrue);
    int ky = 0, kx = 0;
    if (p.ksize == 3) { ky = tap / 3; kx = tap - ky * 3; }
    const bool second = c >= p.C1;
    const T* src = second ? (const T*)p.x2 : (const T*)p.x;
    const int cs = second ? p.C2 : p.C1;
    const int cc = second ? c - p.C1 : c;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int iy = iy0[i] + ky, ix = ix0[i] + kx;
      const bool ok = kvalid && (unsigned)iy < (unsigned)Hin && (unsigned)ix < (unsigned)Win;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (ok) {
        const int pix = nb[i] + (iy >> p.upsample) * p.Wsrc + (ix >> p.upsample);
        v = *(const u32x4*)(src + (size_t)pix * cs + cc);
      }
      xr[i] = v;
    }
    // advance to the next K-step
    cpos += BK;
    while (cpos >= Ctot) { cpos -= Ctot; ++tap; }
  };

  f32x4 acc[FN][FM];
#pragma unroll
  for (int i = 0; i < FN; ++i)
#pragma unroll
    for (int j = 0; j < FM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (ks0 < ks1) load_tile(ks0);

  const int frow = lane & 15, fq = lane >> 4;
  for (int ks = ks0; ks < ks1; ++ks) {
    char* xs = smem + ((ks - ks0) & 1) * (XT + WT);
    char* ws = xs + XT;
#pragma unroll
    for (int i = 0; i < 4; ++i) *(u32x4*)(xs + (r0 + 32 * i) * 128 + swz) = xr[i];
#pragma unroll
    for (int i = 0; i < WROWS; ++i) *(u32x4*)(ws + (r0 + 32 * i) * 128 + swz) = wr[i];
    __syncthreads();
    if (ks + 1 < ks1) load_tile(ks + 1);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      typename Traits<T>::vec8 xa[FM], wa[FN];
#pragma unroll
      for (int j = 0; j < FM; ++j) {
        const int row = wm * 64 + j * 16 + frow;
        xa[j] = as_vec8<T>(*(const u32x4*)(xs + row * 128 + (((4 * s + fq) ^ (row & 7)) << 4)));
      }
#pragma unroll
      for (int i = 0; i < FN; ++i) {
        const int row = wn * (BN / 2) + i * 16 + frow;
        wa[i] = as_vec8<T>(*(const u32x4*)(ws + row * 128 + (((4 * s + fq) ^ (row & 7)) << 4)));
      }
#pragma unroll
      for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int j = 0; j < FM; ++j) acc[i][j] = mfma16(wa[i], xa[j], acc[i][j]);
    }
  }

  // ---------------- epilogue ----------------
  const int mbase = tile_m * BM + wm * 64 + frow;
  const int cbase = tile_n * BN + wn * (BN / 2) + fq * 4;
  if (p.splitk > 1) {
    float* wsp = p.workspace + (size_t)z * M * p.rows_padded;
#pragma unroll
    for (int j = 0; j < FM; ++j) {
      const int m = mbase + j * 16;
      if (m < M) {
#pragma unroll
        for (int i = 0; i < FN; ++i) *(f32x4*)(wsp + (size_t)m * p.rows_padded + cbase + i * 16) = acc[i][j];
      }
    }
    return;
  }
  EpiArgs e;
  e.bias = p.bias; e.temb = p.temb; e.residual = p.residual; e.out_scale_dev = p.out_scale_dev; e.out = p.out;
  e.M = M; e.Cout = p.Cout; e.Cstore = (p.act == ES_ACT_GEGLU) ? p.Cout / 2 : p.Cout; e.HWout = HWout;
  e.temb_stride = p.temb_stride; e.act = p.act; e.out_scale = p.out_scale;
  float scale = p.out_scale;
  if (p.out_scale_dev) scale *= *p.out_scale_dev;
  if (p.act == ES_ACT_GEGLU) {
    if constexpr (FN % 2 == 0) {
#pragma unroll
      for (int j = 0; j < FM; ++j)
#pragma unroll
        for (int i = 0; i < FN; i += 2) {
          float h[4], g[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) { h[r] = acc[i][j][r]; g[r] = acc[i + 1][j][r]; }
          epilogue_geglu_quad<T>(e, mbase + j * 16, cbase + i * 16, h, g, scale);
        }
    }
  } else {
#pragma unroll
    for (int j = 0; j < FM; ++j)
#pragma unroll
      for (int i = 0; i < FN; ++i) {
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = acc[i][j][r];
        epilogue_quad<T>(e, mbase + j * 16, cbase + i * 16, v, scale);
      }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const es_gemm_desc p, const int M) {
  const int quads = p.rows_padded / 4;
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long long)M * quads) return;
  const int m = (int)(idx / quads);
  const int c0 = (int)(idx - (long long)m * quads) * 4;
  if (c0 >= p.Cout) return;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  for (int z = 0; z < p.splitk; ++z)
    s += *(const f32x4*)(p.workspace + ((size_t)z * M + m) * p.rows_padded + c0);
  EpiArgs e;
  e.bias = p.bias; e.temb = p.temb; e.residual = p.residual; e.out_scale_dev = p.out_scale_dev; e.out = p.out;
  e.M = M; e.Cout = p.Cout; e.Cstore = p.Cout; e.HWout = p.Hout * p.Wout;
  e.temb_stride = p.temb_stride; e.act = p.act; e.out_scale = p.out_scale;
  float scale = p.out_scale;
  if (p.out_scale_dev) scale *= *p.out_scale_dev;
  float v[4] = {s[0], s[1], s[2], s[3]};
  epilogue_quad<T>(e, m, c0, v, scale);
}



template <typename T>
int launch(const es_gemm_desc& d, hipStream_t st) {
  const int M = d.N * d.Hout * d.Wout;
  const int nk = d.Kpad / BK;
  const int Ctot = d.C1 + d.C2;
  const bool aligned = (Ctot % BK == 0) && (d.C1 % BK == 0);
  dim3 grid((M + BM - 1) / BM, d.rows_padded / d.bn, d.splitk);
  const size_t lds = 2 * (size_t)(BM + d.bn) * BK * 2;
#define ES_LAUNCH(BNV, AL)                                                                                  \
  do {                                                                                                      \
    auto kfn = conv_gemm_kernel<T, BNV, AL>;                                                                \
    static bool attr_set = false;                                                                           \
    if (!attr_set) {                                                                                        \
      (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);    \
      attr_set = true;                                                                                      \
    }                                                                                                       \
    hipLaunchKernelGGL(kfn, grid, dim3(256), lds, st, d, M, nk);                                            \
  } while (0)
  if (d.bn == 128) { if (aligned) ES_LAUNCH(128, true); else ES_LAUNCH(128, false); }
  else             { if (aligned) ES_LAUNCH(160, true); else ES_LAUNCH(160, false); }
#undef ES_LAUNCH
  if (d.splitk > 1) {
    const long long total = (long long)M * (d.rows_padded / 4);
    hipLaunchKernelGGL(splitk_reduce_kernel<T>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, d, M);
  }
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

}  // namespace

extern "C" void es_set_error(const char* msg);

extern "C" size_t es_conv_gemm_workspace_bytes(const es_gemm_desc* d) {
  if (d->splitk <= 1) return 0;
  return (size_t)d->splitk * d->N * d->Hout * d->Wout * d->rows_padded * sizeof(float);
}

extern "C" int es_conv_gemm(const es_gemm_desc* d, void* stream) {
  const int Ctot = d->C1 + d->C2;
  const int Ktrue = d->ksize * d->ksize * Ctot;
  if (!d->x || !d->w || !d->out) { es_set_error("es_conv_gemm: null pointer"); return -1; }
  if (d->bn != 128 && d->bn != 160) { es_set_error("es_conv_gemm: bn must be 128 or 160"); return -1; }
  if (d->rows_padded % d->bn || d->rows_padded < d->Cout) { es_set_error("es_conv_gemm: bad rows_padded"); return -1; }
  if (d->Kpad % BK || d->Kpad < Ktrue) { es_set_error("es_conv_gemm: bad Kpad"); return -1; }
  if (d->C1 % 8 || d->C2 % 8 || (d->C2 && !d->x2)) { es_set_error("es_conv_gemm: channels must be multiples of 8"); return -1; }
  if (d->ksize != 1 && d->ksize != 3) { es_set_error("es_conv_gemm: ksize must be 1 or 3"); return -1; }
  if (d->splitk < 1 || d->splitk > d->Kpad / BK) { es_set_error("es_conv_gemm: bad splitk"); return -1; }
  if (d->splitk > 1 && (!d->workspace || d->act == ES_ACT_GEGLU)) { es_set_error("es_conv_gemm: splitk needs workspace and no GEGLU"); return -1; }
  if (d->act == ES_ACT_GEGLU && (d->bn != 128 || d->Cout % 32)) { es_set_error("es_conv_gemm: GEGLU needs bn=128, Cout%32==0"); return -1; }
  if (d->N < 1 || d->Hout < 1 || d->Wout < 1) { es_set_error("es_conv_gemm: empty problem"); return -1; }
  hipStream_t st = (hipStream_t)stream;
  int rc = d->dtype == ES_F16 ? launch<f16>(*d, st) : launch<bf16>(*d, st);
  if (rc) es_set_error("es_conv_gemm: launch failed");
  return rc;
}
