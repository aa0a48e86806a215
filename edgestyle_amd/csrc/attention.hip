// Fused attention  O = softmax(Q K^T * scale) V  for gfx950 — one kernel: QK^T, online softmax, PV on MFMA.
//
// Orientation ("swapped QK^T"): each wave computes S^T = K Q^T with mfma(A = K fragment, B = Q^T fragment), so a
// lane owns ONE query column (lane & 15) and its 16 accumulator registers per 64-key tile are keys.  Row max / row
// sum are therefore in-lane reductions plus two cross-lane xor steps (16, 32), and the P^T accumulators are
// already laid out as the B operand of the second product O^T = V^T P^T (no LDS round trip for P).  V is staged
// row-major [key][dv] (coalesced from HBM) and consumed column-wise with ds_read_b64_tr_b16.
//
// Block = 4 waves; each wave owns QF x 16 queries (QF = 2 -> 128 queries per block), K/V tiles of KVT keys are
// shared by the 4 waves through LDS; the next tile's global loads are in flight behind the current tile's MFMAs.
#include <cstdlib>
#include "common.h"
#include "../../include/edgestyle_hip.h"

// Tool-only build (-DES_ATTN_STAMPS): wave 0 of block (1,0,0) accumulates s_memtime deltas per loop phase into
// es_attn_dbg (read back with es_attn_debug_read).  Perturbs the schedule; never compiled into the product library.
#ifdef ES_ATTN_STAMPS
__device__ unsigned long long es_attn_dbg[16];
#define ES_STAMP(i)                                                                    \
  do {                                                                                 \
    const unsigned long long t_now = __builtin_amdgcn_s_memtime();                     \
    dbg_acc[i] += t_now - t_prev;                                                      \
    t_prev = t_now;                                                                    \
  } while (0)
#else
#define ES_STAMP(i)
#endif

namespace {

template <typename T, int KS /* QK k-steps of 32: DPAD = 32*KS */, int DF /* dv fragments of 16 */, int QF, int KVT,
          bool ONES /* head_dim % 16 == 8: V pad column d holds 1.0, so the PV product also yields the row sum */>
__global__ __launch_bounds__(256, (QF == 1 && KS <= 3) ? 3 : ((KS <= 5 && QF <= 2) ? 2 : 1)) void attention_kernel(const es_attn_desc p) {
  constexpr int DPAD = 32 * KS;
  constexpr int DVP = 16 * DF;
  constexpr int KROW = DPAD * 2 + 16;   // bytes per K row in LDS (+16 B pad)
  constexpr int VROW = DVP * 2 + 16;
  constexpr int KF = KVT / 16;          // key fragments per tile
  constexpr int KCH = DPAD / 8;         // 16-byte chunks per K row
  constexpr int VCH = DVP / 8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // two LDS buffers for the K/V tile: tile t+1 is written while the other waves may still read tile t, so the loop
  // needs ONE barrier per tile (d <= 160; the 512-wide VAE head keeps a single buffer and two barriers)
  constexpr bool DBUF = KS <= 5;
  constexpr int TILE_BYTES = KVT * KROW + KVT * VROW;
  char* ks_ = smem;
  char* vs_ = smem + KVT * KROW;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int col = lane & 15, g = lane >> 4;
  const int h = blockIdx.y, n = blockIdx.z;
  const int q0 = blockIdx.x * (64 * QF) + wave * (16 * QF);
  const int d = p.d;
  const int dch = d / 8;                // valid 16-byte chunks per row

  const T* Q = (const T*)p.q + (size_t)n * p.bsq + (size_t)h * d;
  const T* K = (const T*)p.k + (size_t)n * p.bsk + (size_t)h * d;
  const T* V = (const T*)p.v + (size_t)n * p.bsv + (size_t)h * d;
  T* O = (T*)p.o + (size_t)n * p.bso + (size_t)h * d;

  const float sl2 = p.scale * 1.4426950408889634f;
  // Q^T fragments (B operand): lane holds Q[query = col][32*ks + 8*g .. +7]
  typename Traits<T>::vec8 qf[QF][KS];
#pragma unroll
  for (int f = 0; f < QF; ++f) {
    int qi = q0 + f * 16 + col;
    qi = qi < p.Sq ? qi : p.Sq - 1;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int ch = 4 * s + g;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (ch < dch) v = *(const u32x4*)(Q + (size_t)qi * p.ldq + ch * 8);
      auto qv = as_vec8<T>(v);
      // fold softmax scale * log2(e) into Q once: the scores leave the MFMA ready for exp2
#pragma unroll
      for (int e = 0; e < 8; ++e) qv[e] = from_f32<T>(to_f32(qv[e]) * sl2);
      qf[f][s] = qv;
    }
  }

  // zero the K pad chunks once (they multiply the zero Q pad; must not be NaN garbage)
  for (int b = 0; b < (DBUF ? 2 : 1); ++b) {
    for (int i = tid; i < KVT * KCH; i += 256) {
      const int r = i / KCH, c = i - r * KCH;
      if (c >= dch) *(u32x4*)(ks_ + b * TILE_BYTES + r * KROW + c * 16) = u32x4{0u, 0u, 0u, 0u};
    }
    for (int i = tid; i < KVT * VCH; i += 256) {
      const int r = i / VCH, c = i - r * VCH;
      // ONES: element d of every V row is 1.0 (f16 0x3C00 / bf16 0x3F80): row d of O^T accumulates sum_k P[k]
      const unsigned one = (ONES && c == dch) ? (sizeof(T) == 2 && Traits<T>::is_bf16 ? 0x3F80u : 0x3C00u) : 0u;
      if (c >= dch) *(u32x4*)(vs_ + b * TILE_BYTES + r * VROW + c * 16) = u32x4{one, 0u, 0u, 0u};
    }
  }

  // staging: chunks of the K/V tile handled by this thread.  Raw buffer loads with the hardware range check do the
  // predication: a chunk this thread does not own, or a key row >= Skv, is an out-of-range offset and reads as zero —
  // no branches, no selects, no 64-bit address math per tile (one 32-bit add per chunk).
  constexpr int KPT = (KVT * KCH + 255) / 256;   // chunks per thread (upper bound; K and V tiles have equal counts)
  constexpr unsigned OOB = 0xFFFFFF00u;
  u32x4 kr[KPT], vr[KPT];
  const int nch = KVT * dch;                     // valid chunks per tile
  const auto rK = __builtin_amdgcn_make_buffer_rsrc((void*)K, (short)0, (int)(((size_t)(p.Skv - 1) * p.ldk + d) * 2), 0x00020000);
  const auto rV = __builtin_amdgcn_make_buffer_rsrc((void*)V, (short)0, (int)(((size_t)(p.Skv - 1) * p.ldv + d) * 2), 0x00020000);
  unsigned koff[KPT], voffs[KPT];
  int klds[KPT], vlds[KPT];
#pragma unroll
  for (int i = 0; i < KPT; ++i) {
    const int idx = tid + i * 256;
    const bool own = idx < nch;
    const int r = own ? idx / dch : 0;
    const int c = own ? idx - r * dch : 0;
    koff[i] = own ? (unsigned)((r * p.ldk + c * 8) * 2) : OOB;
    voffs[i] = own ? (unsigned)((r * p.ldv + c * 8) * 2) : OOB;
    klds[i] = own ? r * KROW + c * 16 : -1;
    vlds[i] = own ? r * VROW + c * 16 : -1;
  }
  const unsigned kstep = (unsigned)(KVT * p.ldk * 2), vstep = (unsigned)(KVT * p.ldv * 2);
  auto load_kv = [&]() {                          // loads the NEXT tile and advances the offsets
#pragma unroll
    for (int i = 0; i < KPT; ++i) {
      kr[i] = __builtin_amdgcn_raw_buffer_load_b128(rK, (int)koff[i], 0, 0);
      vr[i] = __builtin_amdgcn_raw_buffer_load_b128(rV, (int)voffs[i], 0, 0);
      koff[i] = koff[i] >= OOB ? OOB : koff[i] + kstep;
      voffs[i] = voffs[i] >= OOB ? OOB : voffs[i] + vstep;
    }
  };
  auto store_kv = [&](int buf) {
#pragma unroll
    for (int i = 0; i < KPT; ++i) {
      if (klds[i] >= 0) {
        *(u32x4*)(ks_ + buf * TILE_BYTES + klds[i]) = kr[i];
        *(u32x4*)(vs_ + buf * TILE_BYTES + vlds[i]) = vr[i];
      }
    }
  };

  // Online softmax state per query (= per lane column).  The running reference m is kept NEGATED and splatted as
  // the initial accumulator of the S^T MFMA chain, so S' = S*scale*log2e - m comes out of the matrix core and the
  // softmax costs one v_exp per score (no subtract, no scale).  m is raised lazily: only when some score of the
  // wave exceeds it by more than LAZY (2^8 headroom is harmless in fp32 accumulators and in fp16/bf16 P), and
  // always on the first tile.
  constexpr float LAZY = 8.0f;
  f32x4 o[QF][DF];
  f32x4 negm[QF];
  float lrun[QF];
#pragma unroll
  for (int f = 0; f < QF; ++f) {
    negm[f] = f32x4{0.f, 0.f, 0.f, 0.f}; lrun[f] = 0.f;
#pragma unroll
    for (int j = 0; j < DF; ++j) o[f][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  load_kv();
  if (DBUF) { store_kv(0); if (KVT < p.Skv) load_kv(); }
  int buf = 0;
#ifdef ES_ATTN_STAMPS
  unsigned long long dbg_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long t_prev = __builtin_amdgcn_s_memtime();
#endif
  for (int kv0 = 0; kv0 < p.Skv; kv0 += KVT) {
    ES_STAMP(0);                   // loop overhead / previous iteration's tail
    if (DBUF) {
      __syncthreads();             // tile kv0 (written last iteration / prologue) visible; tile kv0-KVT fully consumed
    } else {
      __syncthreads();             // previous tile fully consumed (and pad zeroing visible on first pass)
      store_kv(0);
      __syncthreads();
      if (kv0 + KVT < p.Skv) load_kv();
    }
    ES_STAMP(1);                   // barrier
    const char* kb = ks_ + buf * TILE_BYTES;
    const char* vb = vs_ + buf * TILE_BYTES;

    // ---- S^T = K Q^T ----  (first K-step starts the chain from a literal zero accumulator)
    f32x4 s[QF][KF];
#pragma unroll
    for (int ksx = 0; ksx < KS; ++ksx) {
#pragma unroll
      for (int kf = 0; kf < KF; ++kf) {
        const auto ka = as_vec8<T>(*(const u32x4*)(kb + (kf * 16 + col) * KROW + (4 * ksx + g) * 16));
#pragma unroll
        for (int f = 0; f < QF; ++f)
          s[f][kf] = mfma16(ka, qf[f][ksx], ksx == 0 ? negm[f] : s[f][kf]);
      }
    }
#ifdef ES_ATTN_STAMPS
    asm volatile("s_nop 0" ::"v"(s[0][0]), "v"(s[QF - 1][KF - 1]));   // wait for the S^T accumulators
#endif
    ES_STAMP(2);                   // K fragment reads + QK MFMAs
    // ---- online softmax (per query = per lane column) ----
    // VALU budget matters more than MFMA at head_dim 40/80 (v_exp issues at half rate): per score one v_exp, half a
    // v_max3 and half a packed convert; key masking only on the ragged last tile.
    const bool ragged = kv0 + KVT > p.Skv;
    typename Traits<T>::vec8 pb[QF][KF / 2];
    if (__builtin_expect(ragged, 0)) {     // one cold block for all fragments: never if-converted into the hot loop
#pragma unroll
      for (int f = 0; f < QF; ++f)
#pragma unroll
        for (int kf = 0; kf < KF; ++kf)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            s[f][kf][r] = (kv0 + kf * 16 + g * 4 + r >= p.Skv) ? -3.0e38f : s[f][kf][r];
    }
#pragma unroll
    for (int f = 0; f < QF; ++f) {
      float mx = s[f][0][0];
#pragma unroll
      for (int kf = 0; kf < KF; ++kf) {
        mx = fmaxf(fmaxf(mx, s[f][kf][0]), s[f][kf][1]);
        mx = fmaxf(fmaxf(mx, s[f][kf][2]), s[f][kf][3]);
      }
      // lanes l, l^16, l^32, l^48 hold the four key groups of one query: two VALU half-swaps (no LDS round trip)
      mx = xor16_max(mx);
      mx = xor32_max(mx);
      if (kv0 == 0 || !__all(mx <= LAZY)) {                 // wave-uniform: raise the reference of this wave's queries
        const float dlt = kv0 == 0 ? mx : fmaxf(mx, 0.f);   // key 0 is always valid, so the first-tile max is finite
        const float nm = negm[f][0] - dlt;
        negm[f] = f32x4{nm, nm, nm, nm};
#pragma unroll
        for (int kf = 0; kf < KF; ++kf)
#pragma unroll
          for (int r = 0; r < 4; ++r) s[f][kf][r] -= dlt;
        if (kv0 != 0) {
          const float alpha = __builtin_amdgcn_exp2f(-dlt);
          lrun[f] *= alpha;
#pragma unroll
          for (int j = 0; j < DF; ++j) o[f][j] *= alpha;
        }
      }
#pragma unroll
      for (int kf = 0; kf < KF; ++kf)
#pragma unroll
        for (int r = 0; r < 4; ++r) s[f][kf][r] = __builtin_amdgcn_exp2f(s[f][kf][r]);
      if constexpr (!ONES) {
        float rs = 0.f;
#pragma unroll
        for (int kf = 0; kf < KF; ++kf)
#pragma unroll
          for (int r = 0; r < 4; ++r) rs += s[f][kf][r];
        lrun[f] += rs;                                      // per-lane partial; the 4 key groups are summed at the end
      }
      // P^T as B operand: k index 8g+j <-> key 32*sx + 16*(j>>2) + 4g + (j&3)
#pragma unroll
      for (int sx = 0; sx < KF / 2; ++sx) {
        typename Traits<T>::vec8 b;
#pragma unroll
        for (int j = 0; j < 8; ++j) b[j] = from_f32<T>(s[f][2 * sx + (j >> 2)][j & 3]);
        pb[f][sx] = b;
      }
    }
#ifdef ES_ATTN_STAMPS
    asm volatile("s_nop 0" ::"v"(pb[0][0]), "v"(pb[QF - 1][KF / 2 - 1]));
#endif
    ES_STAMP(3);                   // softmax
    // ---- O^T += V^T P^T ----
#pragma unroll
    for (int sx = 0; sx < KF / 2; ++sx) {
#pragma unroll
      for (int j = 0; j < DF; ++j) {
        // lane 4q+pp of each 16-lane group supplies row q, columns 4pp..4pp+3 of its 4x16 block
        const int q = col >> 2, pp = col & 3;
        const char* base = vb + (32 * sx + 4 * g + q) * VROW + (j * 16 + 4 * pp) * 2;
        const u32x2 lo = lds_read_tr16(base);
        const u32x2 hi = lds_read_tr16(base + 16 * VROW);
        const u32x4 av = {lo[0], lo[1], hi[0], hi[1]};
        const auto va = as_vec8<T>(av);
#pragma unroll
        for (int f = 0; f < QF; ++f) o[f][j] = mfma16(va, pb[f][sx], o[f][j]);
      }
    }
#ifdef ES_ATTN_STAMPS
    asm volatile("s_nop 0" ::"v"(o[0][0]), "v"(o[QF - 1][DF - 1]));
#endif
    ES_STAMP(4);                   // V reads + PV MFMAs
    if (DBUF) {
      // tile kv0+KVT (in registers since the top of this iteration) -> the other buffer; then fetch kv0+2*KVT
      if (kv0 + KVT < p.Skv) {
        store_kv(buf ^ 1);
        if (kv0 + 2 * KVT < p.Skv) load_kv();
      }
      buf ^= 1;
    }
    ES_STAMP(5);                   // staging: registers -> LDS, next global loads
  }
#ifdef ES_ATTN_STAMPS
  if (tid == 0 && blockIdx.x == 1 && blockIdx.y == 0 && blockIdx.z == 0)
    for (int i = 0; i < 8; ++i) es_attn_dbg[i] = dbg_acc[i];
#endif

  // ---- epilogue: O[query][dv] = O^T / l ----
#pragma unroll
  for (int f = 0; f < QF; ++f) {
    const int qi = q0 + f * 16 + col;
    float l;
    if constexpr (ONES) {
      l = __shfl(o[f][DF - 1][0], 32 + col, 64);            // row d of O^T = fragment DF-1, lane group 2, register 0
    } else {
      l = lrun[f];
      l = xor32_sum(xor16_sum(l));
    }
    const float inv = 1.0f / l;
    if (qi < p.Sq) {
#pragma unroll
      for (int j = 0; j < DF; ++j) {
        const int dv = j * 16 + g * 4;
        if (dv < d) {     // d % 8 == 0 so a quad is entirely valid or entirely pad
          typename Traits<T>::vec4 pk;
#pragma unroll
          for (int r = 0; r < 4; ++r) pk[r] = from_f32<T>(o[f][j][r] * inv);
          *(typename Traits<T>::vec4*)(O + (size_t)qi * p.ldo + dv) = pk;
        }
      }
    }
  }
}

template <typename T, int KS, int DF, int QF, int KVT, bool ONES = false>
int launch_attn(const es_attn_desc& d, hipStream_t st) {
  constexpr int lds = (KS <= 5 ? 2 : 1) * (KVT * (32 * KS * 2 + 16) + KVT * (16 * DF * 2 + 16));
  auto kfn = attention_kernel<T, KS, DF, QF, KVT, ONES>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_set = true;
  }
  dim3 grid((d.Sq + 64 * QF - 1) / (64 * QF), d.heads, d.N);
  hipLaunchKernelGGL(kfn, grid, dim3(256), lds, st, d);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

template <typename T>
int dispatch(const es_attn_desc& d, hipStream_t st) {
  // 32 queries per wave (128 per block) only when that still yields >= 2 blocks per CU; else 16 per wave
  static const long long big_thr = getenv("ES_ATTN_BIG") ? atoll(getenv("ES_ATTN_BIG")) : 512;
  const bool big = (long long)((d.Sq + 127) / 128) * d.heads * d.N >= big_thr;
  // (64 queries per wave was measured slower: 309 registers -> one wave per SIMD)
  switch (d.d) {
    case 8: return launch_attn<T, 1, 1, 2, 64, true>(d, st);
    case 16: return launch_attn<T, 1, 1, 2, 64>(d, st);
    case 24: return launch_attn<T, 1, 2, 2, 64, true>(d, st);
    case 32: return launch_attn<T, 1, 2, 2, 64>(d, st);
    case 40: return big ? launch_attn<T, 2, 3, 2, 64, true>(d, st) : launch_attn<T, 2, 3, 1, 64, true>(d, st);
    case 48: return big ? launch_attn<T, 2, 3, 2, 64>(d, st) : launch_attn<T, 2, 3, 1, 64>(d, st);
    case 64: return launch_attn<T, 2, 4, 2, 64>(d, st);
    case 80: return big ? launch_attn<T, 3, 5, 2, 64>(d, st) : launch_attn<T, 3, 5, 1, 64>(d, st);
    case 128: return launch_attn<T, 4, 8, 2, 64>(d, st);
    case 160: return launch_attn<T, 5, 10, 2, 64>(d, st);
    case 512: return launch_attn<T, 16, 32, 1, 32>(d, st);
    default: return -3;
  }
}

}  // namespace

extern "C" void es_set_error(const char* msg);

#ifdef ES_ATTN_STAMPS
extern "C" int es_attn_debug_read(unsigned long long* host16) {
  return (int)hipMemcpyFromSymbol(host16, HIP_SYMBOL(es_attn_dbg), 16 * sizeof(unsigned long long));
}
#endif

extern "C" int es_attention(const es_attn_desc* d, void* stream) {
  if (!d->q || !d->k || !d->v || !d->o) { es_set_error("es_attention: null pointer"); return -1; }
  if (d->d % 8 || d->ldq % 8 || d->ldk % 8 || d->ldv % 8 || d->ldo % 4) { es_set_error("es_attention: d and strides must be multiples of 8"); return -1; }
  if (d->Sq < 1 || d->Skv < 1 || d->N < 1 || d->heads < 1) { es_set_error("es_attention: empty problem"); return -1; }
  hipStream_t st = (hipStream_t)stream;
  int rc = d->dtype == ES_F16 ? dispatch<f16>(*d, st) : dispatch<bf16>(*d, st);
  if (rc == -3) es_set_error("es_attention: unsupported head_dim (8,16,24,32,40,48,64,80,128,160,512)");
  else if (rc) es_set_error("es_attention: launch failed");
  return rc;
}
