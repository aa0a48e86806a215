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
#include <type_traits>
#include "common.h"
#include "../../include/edgestyle_hip.h"
#include "plan.h"

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
          store8(O + (size_t)qi * p.ldo + dv, __builtin_bit_cast(u32x2, pk));
        }
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------
// 32x32 score tiles for the VALU-bound head dims (40, 80: UNet levels 0 and 1, where the softmax and not the matrix
// core sets the pace).  S^T = K Q^T runs on mfma_f32_32x32x16: K-steps of 16 fit d = 40 / 80 as 48 / 80 (the 16x16x32
// form pads them to 64 / 96), and a 32x32x16 MFMA blocks the SIMD's vector issue for 8 of its 32 cycles instead of 8
// of 16, so the QK^T product costs 6 issue slots per 64 keys x 32 queries instead of 16.  A lane then owns ONE query
// (lane & 31) and 16 keys per tile: the row max is 31 in-lane max + one permlane32 swap for 32 queries (was two
// 16-query groups with two swaps each).  PV stays on 16x16x32 (d_v = 40 pads to 48, not 64): the exp'd scores are
// packed to 16 bits in accumulator order and ONE v_permlane16_swap per register pair re-deals them into the B
// operands of the two 16-query halves (X = registers 0..7, Y = registers 8..15 of a tile; after the swap X holds
// queries 0..15 in all four lane groups, Y queries 16..31).  k index 8g+j of that operand is key
// 16(g&1) + 4(g>>1) + 8(j>>2) + (j&3) of the tile; the V^T fragments are read with the same map.
typedef float f32x16 __attribute__((ext_vector_type(16)));
// Tool-only builds (tools/attn_ablate.sh).  ATTN_ABLATE: time attention32_kernel with one ingredient removed (results wrong by
// construction): 1 = no v_exp, 2 = no PV MFMAs, 4 = no QK^T MFMAs, 8 = no row max.  The product library has ATTN_ABLATE == 0.
#ifndef ATTN_ABLATE
#define ATTN_ABLATE 0
#endif
// ATTN_PRIO: static s_setprio from the wave's hardware slot on its SIMD (HW_ID.WAVE_ID): co-resident waves of different
// workgroups run the same program; at equal priority they share the matrix pipe and the vector issue evenly, stay in phase
// and the SIMD does MFMA phases and softmax phases one after the other.  0 = off, 1 = slot & 1, 2 = slot & 3,
// 3 = by phase: priority 1 while a wave issues its MFMA clusters (8 issue cycles per 16-32 cycles of matrix pipe), 0 in its softmax.
#ifndef ATTN_PRIO
#define ATTN_PRIO 3
#endif
ES_DEVICE void attn_static_prio() {
#if ATTN_PRIO == 1 || ATTN_PRIO == 2
  const unsigned slot = __builtin_amdgcn_s_getreg((3 << 11) | 4);     // HW_REG_HW_ID[3:0] = wave slot on the SIMD
#if ATTN_PRIO == 1
  if (slot & 1) __builtin_amdgcn_s_setprio(1);
#else
  switch (slot & 3) {
    case 1: __builtin_amdgcn_s_setprio(1); break;
    case 2: __builtin_amdgcn_s_setprio(2); break;
    case 3: __builtin_amdgcn_s_setprio(3); break;
    default: break;
  }
#endif
#endif
}
ES_DEVICE f32x16 mfma32(f16x8 a, f16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
ES_DEVICE f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }

template <typename T> ES_DEVICE unsigned pack2(float a, float b) {
  typedef T v2 __attribute__((ext_vector_type(2)));
  v2 r;
  r[0] = from_f32<T>(a); r[1] = from_f32<T>(b);
  return __builtin_bit_cast(unsigned, r);
}

template <typename T, int KS /* QK k-steps of 16: DPAD = 16*KS */, int DF /* dv fragments of 16 */, bool ONES,
          int QB = 1 /* 32-query blocks per wave: every K / V fragment read from LDS serves all of them */>
__global__ __launch_bounds__(256, 2) void attention32_kernel(const es_attn_desc p) {
  constexpr int KVT = 64;
  constexpr int DPAD = 16 * KS;
  constexpr int DVP = 16 * DF;
  constexpr int KROW = DPAD * 2 + 16;   // bytes per K row in LDS (+16 B pad: 16 consecutive rows hit 16 distinct bank quads)
  constexpr int VROW = DVP * 2 + 16;
  constexpr int KCH = DPAD / 8;
  constexpr int VCH = DVP / 8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int TILE_BYTES = KVT * KROW + KVT * VROW;
  char* ks_ = smem;
  char* vs_ = smem + KVT * KROW;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n32 = lane & 31, hi = lane >> 5;      // 32x32 layouts: query column / key half
  const int col = lane & 15, g = lane >> 4;       // 16x16 layouts (PV product, epilogue)
  const int h = blockIdx.y, n = blockIdx.z;
  const int q0 = blockIdx.x * (128 * QB) + wave * (32 * QB);
  const int d = p.d;
  const int dch = d / 8;
  attn_static_prio();

  const T* Q = (const T*)p.q + (size_t)n * p.bsq + (size_t)h * d;
  const T* K = (const T*)p.k + (size_t)n * p.bsk + (size_t)h * d;
  const T* V = (const T*)p.v + (size_t)n * p.bsv + (size_t)h * d;
  T* O = (T*)p.o + (size_t)n * p.bso + (size_t)h * d;

  const float sl2 = p.scale * 1.4426950408889634f;
  // Q^T fragments (B operand of 32x32x16): lane holds Q[query = n32][16*ks + 8*hi .. +7], pre-scaled by scale*log2(e)
  typename Traits<T>::vec8 qf[QB][KS];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    int qi = q0 + qb * 32 + n32;
    qi = qi < p.Sq ? qi : p.Sq - 1;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int ch = 2 * s + hi;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (ch < dch) v = *(const u32x4*)(Q + (size_t)qi * p.ldq + ch * 8);
      auto qv = as_vec8<T>(v);
#pragma unroll
      for (int e = 0; e < 8; ++e) qv[e] = from_f32<T>(to_f32(qv[e]) * sl2);
      qf[qb][s] = qv;
    }
  }

  for (int b = 0; b < 2; ++b) {
    for (int i = tid; i < KVT * KCH; i += 256) {
      const int r = i / KCH, c = i - r * KCH;
      // ONES (head_dim % 16 == 8): element d of every K row is 1.0 and element d of the lane's Q row holds the negated
      // running reference, so S^T = K Q^T leaves the matrix core with the reference already subtracted and the MFMA
      // chains start from the inline constant 0 - no 16-register splat of the reference to copy into each chain
      const unsigned kone = (ONES && c == dch) ? (Traits<T>::is_bf16 ? 0x3F80u : 0x3C00u) : 0u;
      if (c >= dch) *(u32x4*)(ks_ + b * TILE_BYTES + r * KROW + c * 16) = u32x4{kone, 0u, 0u, 0u};
    }
    for (int i = tid; i < KVT * VCH; i += 256) {
      const int r = i / VCH, c = i - r * VCH;
      const unsigned one = (ONES && c == dch) ? (Traits<T>::is_bf16 ? 0x3F80u : 0x3C00u) : 0u;
      if (c >= dch) *(u32x4*)(vs_ + b * TILE_BYTES + r * VROW + c * 16) = u32x4{one, 0u, 0u, 0u};
    }
  }

  constexpr int KPT = (KVT * KCH + 255) / 256;
  constexpr unsigned OOB = 0xFFFFFF00u;
  u32x4 kr[KPT], vr[KPT];
  const int nch = KVT * dch;
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
  auto load_kv = [&]() {
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

  constexpr float LAZY = 8.0f;
  f32x4 o[QB][2][DF];             // O^T of the two 16-query halves of each block, 16x16 layout
  f32x16 negm[QB];                // !ONES: negated running reference of this lane's query, splatted: the S^T chains start from it
  float negm_f[QB];               // ONES: the same reference as a scalar (it lives in the pad element of the Q operand)
  float lrun[QB];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    negm_f[qb] = 0.f; lrun[qb] = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) negm[qb][r] = 0.f;
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int j = 0; j < DF; ++j) o[qb][f][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  load_kv();
  store_kv(0);
  if (KVT < p.Skv) load_kv();
  int buf = 0;
  // V^T fragment rows of this lane for the key map above: lanes 4q..4q+3 of a 16-lane group supply row q of a 4x16 block
  const int vrow0 = 16 * (g & 1) + 4 * (g >> 1) + (col >> 2);
  const int vcol0 = 4 * (col & 3);
  for (int kv0 = 0; kv0 < p.Skv; kv0 += KVT) {
    __syncthreads();
    const char* kb = ks_ + buf * TILE_BYTES;
    const char* vb = vs_ + buf * TILE_BYTES;

    f32x16 s[QB][2];
#if ATTN_PRIO == 3
    __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int ksx = 0; ksx < KS; ++ksx) {
        const auto ka = as_vec8<T>(*(const u32x4*)(kb + (t * 32 + n32) * KROW + (2 * ksx + hi) * 16));
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
#if ATTN_ABLATE & 4
          if (ksx == 0) s[qb][t] = negm[qb];
          asm volatile("" : "+v"(s[qb][t]) : "v"(ka), "v"(qf[qb][ksx]));
#else
          s[qb][t] = mfma32(ka, qf[qb][ksx], ksx == 0 ? (ONES ? f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f} : negm[qb]) : s[qb][t]);
#endif
        }
      }
#if ATTN_PRIO == 3
    __builtin_amdgcn_s_setprio(0);
#endif
    if (__builtin_expect(kv0 + KVT > p.Skv, 0)) {
#pragma unroll
      for (int qb = 0; qb < QB; ++qb)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            s[qb][t][r] = (kv0 + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi >= p.Skv) ? -3.0e38f : s[qb][t][r];
    }
    float mx[QB];
    bool calm = true;
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
      mx[qb] = s[qb][0][0];
#if !(ATTN_ABLATE & 8)
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; r += 2) mx[qb] = fmaxf(fmaxf(mx[qb], s[qb][t][r]), s[qb][t][r + 1]);
      mx[qb] = xor32_max(mx[qb]);
#endif
      calm = calm && mx[qb] <= LAZY;
    }
    if (kv0 == 0 || !__all(calm)) {
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) {
        float dlt = kv0 == 0 ? mx[qb] : fmaxf(mx[qb], 0.f);
        if constexpr (ONES) {
          // the reference must be a value of T (it is an element of the Q operand); what the scores and O are rescaled
          // by is the step it actually made.  Any reference is exact - O / l does not depend on it.
          const T nmt = from_f32<T>(negm_f[qb] - dlt);
          dlt = negm_f[qb] - to_f32(nmt);
          negm_f[qb] = to_f32(nmt);
          if (hi) qf[qb][KS - 1][0] = nmt;
        } else {
          const float nm = negm[qb][0] - dlt;
#pragma unroll
          for (int r = 0; r < 16; ++r) negm[qb][r] = nm;
        }
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) s[qb][t][r] -= dlt;
        if (kv0 != 0) {
          const float alpha = __builtin_amdgcn_exp2f(-dlt);
          lrun[qb] *= alpha;
          // alpha lives on the 32-query layout (query = lane & 31); O^T on the 16x16 one (query = 16 f + (lane & 15))
          float a0 = alpha, a1 = alpha;
          asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a0), "+v"(a1));
#pragma unroll
          for (int j = 0; j < DF; ++j) { o[qb][0][j] *= a0; o[qb][1][j] *= a1; }
        }
      }
    }
    typename Traits<T>::vec8 pb[QB][2][2];     // [block][query half][tile]
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) s[qb][t][r] = (ATTN_ABLATE & 1) ? s[qb][t][r] : __builtin_amdgcn_exp2f(s[qb][t][r]);
      if constexpr (!ONES) {
        float rs = 0.f;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) rs += s[qb][t][r];
        lrun[qb] += rs;
      }
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        unsigned x[4], y[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          x[i] = pack2<T>(s[qb][t][2 * i], s[qb][t][2 * i + 1]);
          y[i] = pack2<T>(s[qb][t][8 + 2 * i], s[qb][t][9 + 2 * i]);
        }
        asm volatile("s_nop 1\n\t"
                     "v_permlane16_swap_b32 %0, %4\n\tv_permlane16_swap_b32 %1, %5\n\t"
                     "v_permlane16_swap_b32 %2, %6\n\tv_permlane16_swap_b32 %3, %7\n\ts_nop 1"
                     : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(y[0]), "+v"(y[1]), "+v"(y[2]), "+v"(y[3]));
        pb[qb][0][t] = as_vec8<T>(u32x4{x[0], x[1], x[2], x[3]});
        pb[qb][1][t] = as_vec8<T>(u32x4{y[0], y[1], y[2], y[3]});
      }
    }
#if ATTN_PRIO == 3
    __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int j = 0; j < DF; ++j) {
        const char* base = vb + (32 * t + vrow0) * VROW + (j * 16 + vcol0) * 2;
        const u32x2 lo = lds_read_tr16(base);
        const u32x2 hi2 = lds_read_tr16(base + 8 * VROW);
        const auto va = as_vec8<T>(u32x4{lo[0], lo[1], hi2[0], hi2[1]});
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
#if ATTN_ABLATE & 2
          asm volatile("" : "+v"(o[qb][0][j]), "+v"(o[qb][1][j]) : "v"(va), "v"(pb[qb][0][t]), "v"(pb[qb][1][t]));
#else
          o[qb][0][j] = mfma16(va, pb[qb][0][t], o[qb][0][j]);
          o[qb][1][j] = mfma16(va, pb[qb][1][t], o[qb][1][j]);
#endif
        }
      }
#if ATTN_PRIO == 3
    __builtin_amdgcn_s_setprio(0);
#endif
    if (kv0 + KVT < p.Skv) {
      store_kv(buf ^ 1);
      if (kv0 + 2 * KVT < p.Skv) load_kv();
    }
    buf ^= 1;
  }

  // ---- epilogue: O[query][dv] = O^T / l ----
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    float l0, l1;
    if constexpr (ONES) {
      l0 = __shfl(o[qb][0][DF - 1][0], 32 + col, 64);
      l1 = __shfl(o[qb][1][DF - 1][0], 32 + col, 64);
    } else {
      const float l = xor32_sum(lrun[qb]);                   // per query, 32-query layout
      l0 = l; l1 = l;
      asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(l0), "+v"(l1));
    }
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      const int qi = q0 + qb * 32 + f * 16 + col;
      const float inv = 1.0f / (f ? l1 : l0);
      if (qi < p.Sq) {
#pragma unroll
        for (int j = 0; j < DF; ++j) {
          const int dv = j * 16 + g * 4;
          if (dv < d) {
            typename Traits<T>::vec4 pk;
#pragma unroll
            for (int r = 0; r < 4; ++r) pk[r] = from_f32<T>(o[qb][f][j][r] * inv);
            store8(O + (size_t)qi * p.ldo + dv, __builtin_bit_cast(u32x2, pk));
          }
        }
      }
    }
  }
}

template <typename T, int KS, int DF, bool ONES, int QB = 1>
int launch_attn32(const es_attn_desc& d, hipStream_t st) {
  constexpr int lds = 2 * (64 * (16 * KS * 2 + 16) + 64 * (16 * DF * 2 + 16));
  auto kfn = attention32_kernel<T, KS, DF, ONES, QB>;
  dim3 grid((d.Sq + 128 * QB - 1) / (128 * QB), d.heads, d.N);
  hipLaunchKernelGGL(kfn, grid, dim3(256), lds, st, d);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

// ---------------------------------------------------------------------------------------------------------------
// head_dim 40 self-attention, PING-PONG between the two waves of a SIMD (the structure of gemm_conv8p.hip applied to
// attention).  attention32_kernel is bound by the vector pipe (per 32 queries x 64 keys: 32 v_exp = 256 issue cycles +
// 16 packed converts + the row max, against 384 cycles of matrix pipe), and its waves run [MFMA cluster | softmax | MFMA
// cluster] one after the other while the co-resident waves do the same program in near lockstep: the two pipes take
// turns (tool builds: no MFMAs at all -27 %, no v_exp -17 %).  Here a workgroup is 8 waves = two groups of four (SIMD
// partners are w and w + 4) that alternate, separated by workgroup barriers and ONE barrier out of phase:
//     M phase:  O += V(t) P(t)  (12 MFMA 16x16x32)   and   S(t+1) = K(t+1) Q  (6 MFMA 32x32x16)     - matrix pipe
//     V phase:  P(t+1) = softmax step of S(t+1): row max, lazy reference, v_exp, packed converts   - vector pipe
// so on every SIMD one wave's MFMA stream runs beside its partner's softmax by construction, not by luck of arbitration.
// K / V tiles of 64 keys go through a 2-deep LDS ring by the register path (the +16 B row pad and the ones / reference
// pad column rule out lane-linear LDS-DMA): in the window of two slots in which K(t+1) and V(t) are read, every thread
// loads its chunk of K(t+2) / V(t+1) at the start of the first slot and stores it at the end of the second one.
// Same arithmetic as attention32_kernel<T, 3, 3, true, QB>: reference in the pad element of Q, row sum from the ones
// column of V.  Requires d == 40 and Skv % 64 == 0 (self-attention).
template <typename T, int QB>
__global__ __launch_bounds__(512, 2) void attention40pp_kernel(const es_attn_desc p) {
  constexpr int KS = 3, DF = 3, KVT = 64;
  constexpr int KROW = 16 * KS * 2 + 16, VROW = 16 * DF * 2 + 16;
  constexpr int KBUF = KVT * KROW, VBUF = KVT * VROW;
  constexpr int d = 40, dch = 5;
  constexpr float LAZY = 8.0f;
  typedef typename Traits<T>::vec8 vec8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* ks_ = smem;                    // K ring: 2 x KBUF
  char* vs_ = smem + 2 * KBUF;         // V ring: 2 x VBUF

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;                      // stagger group
  const int n32 = lane & 31, hi = lane >> 5;      // 32x32 layouts: query column / key half
  const int col = lane & 15, g = lane >> 4;       // 16x16 layouts (PV product, epilogue)
  const int h = blockIdx.y, n = blockIdx.z;
  const int q0 = blockIdx.x * (256 * QB) + wave * (32 * QB);

  const T* Q = (const T*)p.q + (size_t)n * p.bsq + (size_t)h * d;
  const T* K = (const T*)p.k + (size_t)n * p.bsk + (size_t)h * d;
  const T* V = (const T*)p.v + (size_t)n * p.bsv + (size_t)h * d;
  T* O = (T*)p.o + (size_t)n * p.bso + (size_t)h * d;

  const float sl2 = p.scale * 1.4426950408889634f;
  vec8 qf[QB][KS];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    int qi = q0 + qb * 32 + n32;
    qi = qi < p.Sq ? qi : p.Sq - 1;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int ch = 2 * s + hi;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (ch < dch) v = *(const u32x4*)(Q + (size_t)qi * p.ldq + ch * 8);
      auto qv = as_vec8<T>(v);
#pragma unroll
      for (int e = 0; e < 8; ++e) qv[e] = from_f32<T>(to_f32(qv[e]) * sl2);
      qf[qb][s] = qv;
    }
  }
  // pad chunk of every K / V row in both ring buffers: element 40 = 1.0 (reference / row-sum column), the rest zero
  {
    const unsigned one = Traits<T>::is_bf16 ? 0x3F80u : 0x3C00u;
    for (int i = tid; i < 2 * KVT; i += 512) {
      *(u32x4*)(ks_ + (i >> 6) * KBUF + (i & 63) * KROW + dch * 16) = u32x4{one, 0u, 0u, 0u};
      *(u32x4*)(vs_ + (i >> 6) * VBUF + (i & 63) * VROW + dch * 16) = u32x4{one, 0u, 0u, 0u};
    }
  }
  // staging: 320 chunks of 16 B per K (V) tile, one per thread (threads 320.. load out of range: zeros, never stored)
  constexpr unsigned OOB = 0xFFFFFF00u;
  const bool own = tid < KVT * dch;
  const int srow = own ? tid / dch : 0, sch = own ? tid - srow * dch : 0;
  const auto rK = __builtin_amdgcn_make_buffer_rsrc((void*)K, (short)0, (int)(((size_t)(p.Skv - 1) * p.ldk + d) * 2), 0x00020000);
  const auto rV = __builtin_amdgcn_make_buffer_rsrc((void*)V, (short)0, (int)(((size_t)(p.Skv - 1) * p.ldv + d) * 2), 0x00020000);
  unsigned koff = own ? (unsigned)((srow * p.ldk + sch * 8) * 2) : OOB;      // of tile 0
  unsigned voff = own ? (unsigned)((srow * p.ldv + sch * 8) * 2) : OOB;
  const int klds = srow * KROW + sch * 16, vlds = srow * VROW + sch * 16;
  const unsigned kstep = (unsigned)(KVT * p.ldk * 2), vstep = (unsigned)(KVT * p.ldv * 2);
  const int nt = p.Skv / KVT;
  u32x4 kr, vr;
  auto load_k = [&](int t) { kr = __builtin_amdgcn_raw_buffer_load_b128(rK, (int)(own && t < nt ? koff + (unsigned)t * kstep : OOB), 0, 0); };
  auto load_v = [&](int t) { vr = __builtin_amdgcn_raw_buffer_load_b128(rV, (int)(own && t < nt ? voff + (unsigned)t * vstep : OOB), 0, 0); };
  auto store_k = [&](int t) { if (own) *(u32x4*)(ks_ + (t & 1) * KBUF + klds) = kr; };
  auto store_v = [&](int t) { if (own) *(u32x4*)(vs_ + (t & 1) * VBUF + vlds) = vr; };

  f32x4 o[QB][2][DF];
  float negm_f[QB];
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    negm_f[qb] = 0.f;
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int j = 0; j < DF; ++j) o[qb][f][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  // prologue: K(0), V(0), K(1) resident before the first slot
  load_k(0); load_v(0);
  store_k(0); store_v(0);
  load_k(1);
  store_k(1);
  __syncthreads();

  const int vrow0 = 16 * (g & 1) + 4 * (g >> 1) + (col >> 2);
  const int vcol0 = 4 * (col & 3);
  f32x16 s[QB][2];
  vec8 pb[QB][2][2];
  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

  auto qk = [&](int t) __attribute__((always_inline)) {          // S = K(t) Q^T
    const char* kb = ks_ + (t & 1) * KBUF;
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
      for (int ksx = 0; ksx < KS; ++ksx) {
        const auto ka = as_vec8<T>(*(const u32x4*)(kb + (tt * 32 + n32) * KROW + (2 * ksx + hi) * 16));
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) s[qb][tt] = mfma32(ka, qf[qb][ksx], ksx == 0 ? zero16 : s[qb][tt]);
      }
  };
  auto pv = [&](int t) __attribute__((always_inline)) {          // O += V(t) P
    const char* vb = vs_ + (t & 1) * VBUF;
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
      for (int j = 0; j < DF; ++j) {
        const char* base = vb + (32 * tt + vrow0) * VROW + (j * 16 + vcol0) * 2;
        const u32x2 lo = lds_read_tr16(base);
        const u32x2 hi2 = lds_read_tr16(base + 8 * VROW);
        const auto va = as_vec8<T>(u32x4{lo[0], lo[1], hi2[0], hi2[1]});
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
          o[qb][0][j] = mfma16(va, pb[qb][0][tt], o[qb][0][j]);
          o[qb][1][j] = mfma16(va, pb[qb][1][tt], o[qb][1][j]);
        }
      }
  };
  auto softmax = [&](bool first) __attribute__((always_inline)) {      // S -> P (packed B operands), lazy reference
    float mx[QB];
    bool calm = true;
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
      mx[qb] = s[qb][0][0];
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int r = 0; r < 16; r += 2) mx[qb] = fmaxf(fmaxf(mx[qb], s[qb][tt][r]), s[qb][tt][r + 1]);
      mx[qb] = xor32_max(mx[qb]);
      calm = calm && mx[qb] <= LAZY;
    }
    if (first || !__all(calm)) {
#pragma unroll
      for (int qb = 0; qb < QB; ++qb) {
        float dlt = first ? mx[qb] : fmaxf(mx[qb], 0.f);
        const T nmt = from_f32<T>(negm_f[qb] - dlt);     // the reference is an element of the Q operand: a value of T
        dlt = negm_f[qb] - to_f32(nmt);
        negm_f[qb] = to_f32(nmt);
        if (hi) qf[qb][KS - 1][0] = nmt;
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
          for (int r = 0; r < 16; ++r) s[qb][tt][r] -= dlt;
        if (!first) {
          const float alpha = __builtin_amdgcn_exp2f(-dlt);
          float a0 = alpha, a1 = alpha;
          asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a0), "+v"(a1));
#pragma unroll
          for (int j = 0; j < DF; ++j) { o[qb][0][j] *= a0; o[qb][1][j] *= a1; }
        }
      }
    }
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int r = 0; r < 16; ++r) s[qb][tt][r] = __builtin_amdgcn_exp2f(s[qb][tt][r]);
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
        unsigned x[4], y[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          x[i] = pack2<T>(s[qb][tt][2 * i], s[qb][tt][2 * i + 1]);
          y[i] = pack2<T>(s[qb][tt][8 + 2 * i], s[qb][tt][9 + 2 * i]);
        }
        asm volatile("s_nop 1\n\t"
                     "v_permlane16_swap_b32 %0, %4\n\tv_permlane16_swap_b32 %1, %5\n\t"
                     "v_permlane16_swap_b32 %2, %6\n\tv_permlane16_swap_b32 %3, %7\n\ts_nop 1"
                     : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(y[0]), "+v"(y[1]), "+v"(y[2]), "+v"(y[3]));
        pb[qb][0][tt] = as_vec8<T>(u32x4{x[0], x[1], x[2], x[3]});
        pb[qb][1][tt] = as_vec8<T>(u32x4{y[0], y[1], y[2], y[3]});
      }
    }
  };

  // slots (between consecutive workgroup barriers): group 0 runs  QK(0) | V(0) | M(0) | V(1) | M(1) ...,
  // group 1 the same one slot later.  K(t+1) and V(t) are read in the two slots in which the groups run M(t); in that
  // window every thread loads K(t+2) / V(t+1) at the start of the first slot and stores them at the end of the second:
  //   group 0:  first slot = its M(t)   second = its V(t+1)        group 1:  first = its V(t)   second = its M(t)
  if (grp == 1) __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_s_setprio(1);
  qk(0);
  __builtin_amdgcn_s_setprio(0);
  __builtin_amdgcn_s_barrier();
#ifdef ES_ATTN_STAMPS
  // tool build: cycles of waves 0 (group 0) and 4 (group 1) of block (1,0,0) in [V-phase work | barrier wait | M-phase work | barrier wait]
  unsigned long long pp_acc[4] = {0, 0, 0, 0};
  unsigned long long pp_prev = __builtin_amdgcn_s_memtime();
#define PP_STAMP(i) do { asm volatile("s_nop 0" ::"v"(s[0][0][0]), "v"(o[0][0][0][0])); const unsigned long long now_ = __builtin_amdgcn_s_memtime(); pp_acc[i] += now_ - pp_prev; pp_prev = now_; } while (0)
#else
#define PP_STAMP(i)
#endif
  for (int t = 0; t < nt; ++t) {
    // ---- V phase: softmax of S(t) ----
    if (grp == 1) { load_k(t + 2); load_v(t + 1); }
    softmax(t == 0);
    if (grp == 0 && t > 0) { store_k(t + 1); store_v(t); }
    PP_STAMP(0);
    __builtin_amdgcn_s_barrier();
    PP_STAMP(1);
    // ---- M phase: O += V(t) P(t), S(t+1) = K(t+1) Q ----
    if (grp == 0) { load_k(t + 2); load_v(t + 1); }
    __builtin_amdgcn_s_setprio(1);
    pv(t);
    if (t + 1 < nt) qk(t + 1);
    __builtin_amdgcn_s_setprio(0);
    if (grp == 1) { store_k(t + 2); store_v(t + 1); }
    PP_STAMP(2);
    __builtin_amdgcn_s_barrier();
    PP_STAMP(3);
  }
#ifdef ES_ATTN_STAMPS
  if ((tid == 0 || tid == 256) && blockIdx.x == 1 && blockIdx.y == 0 && blockIdx.z == 0)
    for (int i = 0; i < 4; ++i) es_attn_dbg[8 + grp * 4 + i] = pp_acc[i];
#endif
#undef PP_STAMP
  if (grp == 0) __builtin_amdgcn_s_barrier();

  // ---- epilogue: O[query][dv] = O^T / l  (l = the ones column: row 40 of O^T) ----
#pragma unroll
  for (int qb = 0; qb < QB; ++qb) {
    const float l0 = __shfl(o[qb][0][DF - 1][0], 32 + col, 64);
    const float l1 = __shfl(o[qb][1][DF - 1][0], 32 + col, 64);
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      const int qi = q0 + qb * 32 + f * 16 + col;
      const float inv = 1.0f / (f ? l1 : l0);
      if (qi < p.Sq) {
#pragma unroll
        for (int j = 0; j < DF; ++j) {
          const int dv = j * 16 + g * 4;
          if (dv < d) {
            typename Traits<T>::vec4 pk;
#pragma unroll
            for (int r = 0; r < 4; ++r) pk[r] = from_f32<T>(o[qb][f][j][r] * inv);
            store8(O + (size_t)qi * p.ldo + dv, __builtin_bit_cast(u32x2, pk));
          }
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Cross-attention over the text tokens (77 keys; Skv <= 96), head_dim 40 | 80: K AND V RESIDENT IN REGISTERS.
// The tiled kernels above treat the 77 keys as two K / V tiles: every block of 64-256 queries stages them through LDS again (pad
// zeroing, two barriers per tile, a rescale step between the tiles) - 33 us for the 14 x 8 x 4096 launch of a batch-1 step whose
// Q read + O write take ~12 us of HBM time (8.7 % MFMA-busy, profiles/r04_mfma_util_b1.json).  Here a wave owns ONE head: it loads
// that head's K as the A operands of S^T = K Q^T (3 key tiles x KS fragments of mfma 32x32x16) and V^T as the A operands of
// O^T = V^T P^T (3 key chunks x DF fragments of mfma 16x16x32, gathered with the key order the packed P^T operands have) ONCE, then
// walks a strip of queries in blocks of 32: Q block (prefetched one block ahead) -> 9 | 15 MFMAs -> single-pass softmax in registers
// (every key of a query is in the lane pair l, l + 32: no running maximum, no rescale) -> 18 | 30 MFMAs -> O.  No LDS, no barrier:
// the waves of a workgroup (four heads) only share the cache lines of their query rows.
template <typename T, int KS /* QK k-steps of 16 */, int DF /* dv fragments of 16 (d = 40: the pad column 40 is the ones column) */, bool ONES>
__global__ __launch_bounds__(512, 2) void attention_kvres_kernel(const es_attn_desc p, const int qper) {
  // Third version (round 5): the first two read Q and wrote O in fragment shape - a wave instruction touched 32 rows x 32 bytes, 32 cache
  // lines per KB - and were bound by that request rate (0.76-0.94x of the tiled kernels, profiles/r05_xattn_bench.txt).  Now the workgroup
  // (eight waves = eight heads, one per wave) moves 32 WHOLE query rows per block: every thread loads 16-byte chunks of consecutive
  // addresses into an LDS tile [32][heads x d] (a block ahead, through registers), the waves read their head's fragments from it, write their
  // O fragments into a second tile, and the workgroup stores that tile as whole rows.  Two barriers per block.
  constexpr int KT = 3;                             // key tiles of 32
  typedef typename Traits<T>::vec8 vec8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n32 = lane & 31, hi = lane >> 5;        // 32x32 layouts: query column / key half
  const int col = lane & 15, g = lane >> 4;         // 16x16 layouts (PV product, epilogue)
  const int h0 = blockIdx.y * 8, n = blockIdx.z;
  const int nh = p.heads - h0 < 8 ? p.heads - h0 : 8;     // heads of this workgroup
  const int h = h0 + wave;
  const bool live = wave < nh;                      // (a wave without a head still takes part in the copies and the barriers)
  const int d = p.d, dch = d / 8;
  const int CW = nh * d;                            // channels of the workgroup's rows
  const int CH = CW / 8;                            // 16-byte chunks per row
  const int QROW = CW * 2 + 16;                     // LDS row stride (bytes)
  char* qt = smem;                                  // [2][32][QROW]
  char* ot = smem + 2 * 32 * QROW;                  // [32][QROW]
  const T* Qg = (const T*)p.q + (size_t)n * p.bsq + (size_t)h0 * d;
  T* Og = (T*)p.o + (size_t)n * p.bso + (size_t)h0 * d;
  const T* K = (const T*)p.k + (size_t)n * p.bsk + (size_t)(live ? h : h0) * d;
  const T* V = (const T*)p.v + (size_t)n * p.bsv + (size_t)(live ? h : h0) * d;
  const float sl2 = p.scale * 1.4426950408889634f;
  const int q_begin = blockIdx.x * qper;
  int q_end = q_begin + qper;
  q_end = q_end < p.Sq ? q_end : p.Sq;
  const bool short3 = p.Skv <= 80;                  // keys 80..95 (registers 8..15 of the third score tile) are padding: the 77 text tokens

  // cooperative copies: chunk idx = tid + k * 512 of a 32-row tile -> (row, chunk of the row)
  constexpr int CPT = KS == 3 ? 3 : 5;              // ceil(32 * 8 heads * d / 8 / 512): d = 40 -> 2.5, d = 80 -> 5
  const float inv_ch = __builtin_amdgcn_rcpf((float)CH);
  int crow[CPT], cch[CPT];
#pragma unroll
  for (int k = 0; k < CPT; ++k) {
    const int idx = tid + k * 512;
    crow[k] = fast_div(idx, CH, inv_ch);
    cch[k] = idx - crow[k] * CH;
    if (crow[k] >= 32) { crow[k] = -1; cch[k] = 0; }
  }
  u32x4 qreg[CPT];
  auto load_q = [&](int q0) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < CPT; ++k) {
      qreg[k] = u32x4{0u, 0u, 0u, 0u};
      const int qi = q0 + crow[k];
      if (crow[k] >= 0 && qi < p.Sq) qreg[k] = *(const u32x4*)(Qg + (size_t)qi * p.ldq + cch[k] * 8);
    }
  };
  auto store_q = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < CPT; ++k)
      if (crow[k] >= 0) *(u32x4*)(qt + buf * 32 * QROW + crow[k] * QROW + cch[k] * 16) = qreg[k];
  };
  load_q(q_begin);

  // K fragments, pre-scaled by scale * log2(e): lane holds K[key = 32 t + n32][16 s + 8 hi .. + 7]; keys >= Skv and channels >= d are zero
  vec8 kf[KT][KS];
#pragma unroll
  for (int t = 0; t < KT; ++t)
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int key = 32 * t + n32, ch = 2 * s + hi;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (key < p.Skv && ch < dch) v = *(const u32x4*)(K + (size_t)key * p.ldk + ch * 8);
      auto kv = as_vec8<T>(v);
#pragma unroll
      for (int e = 0; e < 8; ++e) kv[e] = from_f32<T>(to_f32(kv[e]) * sl2);
      kf[t][s] = kv;
    }
  // V^T fragments: element e of lane (col, g) is V[key = 32 t + 16 (g & 1) + 4 (g >> 1) + 8 (e >> 2) + (e & 3)][dv = 16 j + col] -
  // the key order of the packed P^T operands below (attention32_kernel reads the same map with ds_read_b64_tr_b16).
  // ONES: column d (the first pad column) holds 1.0 for every real key, so row d of O^T is the softmax denominator.
  vec8 vf[KT][DF];
#pragma unroll
  for (int t = 0; t < KT; ++t)
#pragma unroll
    for (int j = 0; j < DF; ++j) {
      vec8 v;
      const int dv = 16 * j + col;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int key = 32 * t + 16 * (g & 1) + 4 * (g >> 1) + 8 * (e >> 2) + (e & 3);
        T x = from_f32<T>(0.f);
        if (key < p.Skv) {
          if (dv < d) x = V[(size_t)key * p.ldv + dv];
          else if (ONES && dv == d) x = from_f32<T>(1.0f);
        }
        v[e] = x;
      }
      vf[t][j] = v;
    }
  store_q(0);
  if (q_begin + 32 < q_end) load_q(q_begin + 32);
  __syncthreads();

  const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const int hb = wave * d * 2;                      // byte offset of this wave's head inside a tile row
  int buf = 0;
  for (int q0 = q_begin; q0 < q_end; q0 += 32, buf ^= 1) {
    f32x4 o[2][DF];
    float l0 = 1.f, l1 = 1.f;
    if (live) {
      vec8 qf[KS];
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const int ch = 2 * s + hi;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (ch < dch) v = *(const u32x4*)(qt + buf * 32 * QROW + n32 * QROW + hb + ch * 16);
        qf[s] = as_vec8<T>(v);
      }
      f32x16 sc[KT];
#pragma unroll
      for (int t = 0; t < KT; ++t)
#pragma unroll
        for (int s = 0; s < KS; ++s) sc[t] = mfma32(kf[t][s], qf[s], s == 0 ? zero16 : sc[t]);
      // keys beyond Skv (zero K rows: score 0) leave the softmax
#pragma unroll
      for (int t = 0; t < KT; ++t)
        if (32 * t + 32 > p.Skv) {                    // wave-uniform
#pragma unroll
          for (int r = 0; r < 16; ++r) sc[t][r] = (32 * t + (r & 3) + 8 * (r >> 2) + 4 * hi >= p.Skv) ? -3.0e38f : sc[t][r];
        }
      float mx = sc[0][0];
#pragma unroll
      for (int t = 0; t < KT; ++t)
#pragma unroll
        for (int r = 0; r < 16; r += 2)
          if (t < 2 || r < 8 || !short3) mx = fmaxf(fmaxf(mx, sc[t][r]), sc[t][r + 1]);
      mx = xor32_max(mx);
      float rs = 0.f;
#pragma unroll
      for (int t = 0; t < KT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          if (t < 2 || r < 8 || !short3) { sc[t][r] = __builtin_amdgcn_exp2f(sc[t][r] - mx); if (!ONES) rs += sc[t][r]; }
          else sc[t][r] = 0.f;
        }
      vec8 pb[2][KT];                                 // [query half][key chunk]: B operands of the PV product
#pragma unroll
      for (int t = 0; t < KT; ++t) {
        unsigned x[4], y[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          x[i] = pack2<T>(sc[t][2 * i], sc[t][2 * i + 1]);
          y[i] = (t == 2 && short3) ? 0u : pack2<T>(sc[t][8 + 2 * i], sc[t][9 + 2 * i]);
        }
        asm volatile("s_nop 1\n\t"
                     "v_permlane16_swap_b32 %0, %4\n\tv_permlane16_swap_b32 %1, %5\n\t"
                     "v_permlane16_swap_b32 %2, %6\n\tv_permlane16_swap_b32 %3, %7\n\ts_nop 1"
                     : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(y[0]), "+v"(y[1]), "+v"(y[2]), "+v"(y[3]));
        pb[0][t] = as_vec8<T>(u32x4{x[0], x[1], x[2], x[3]});
        pb[1][t] = as_vec8<T>(u32x4{y[0], y[1], y[2], y[3]});
      }
#pragma unroll
      for (int j = 0; j < DF; ++j)
#pragma unroll
        for (int t = 0; t < KT; ++t) {
          o[0][j] = mfma16(vf[t][j], pb[0][t], t == 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : o[0][j]);
          o[1][j] = mfma16(vf[t][j], pb[1][t], t == 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : o[1][j]);
        }
      if constexpr (ONES) {
        // row d of O^T = fragment d / 16, lane group (d % 16) / 4, register d % 4: d = 40 -> fragment 2, lanes 32..47, register 0
        l0 = __shfl(o[0][DF - 1][0], 32 + col, 64);
        l1 = __shfl(o[1][DF - 1][0], 32 + col, 64);
      } else {
        const float l = xor32_sum(rs);                // per query, 32-query layout -> the two 16-query halves of the 16x16 layout
        l0 = l; l1 = l;
        asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(l0), "+v"(l1));
      }
    }
    __syncthreads();                                // the previous block's O tile has been stored; every wave is done with Q tile `buf ^ 1`
    if (live) {
#pragma unroll
      for (int f = 0; f < 2; ++f) {
        const float inv = __builtin_amdgcn_rcpf(f ? l1 : l0);
#pragma unroll
        for (int j = 0; j < DF; ++j) {
          const int dv = j * 16 + g * 4;
          if (dv < d) {
            typename Traits<T>::vec4 pk;
#pragma unroll
            for (int r = 0; r < 4; ++r) pk[r] = from_f32<T>(o[f][j][r] * inv);
            *(typename Traits<T>::vec4*)(ot + (f * 16 + col) * QROW + hb + dv * 2) = pk;
          }
        }
      }
    }
    if (q0 + 32 < q_end) {                          // the next block's rows (in registers since the last block) -> the other Q tile
      store_q(buf ^ 1);
      if (q0 + 64 < q_end) load_q(q0 + 64);
    }
    __syncthreads();                                // O tile complete, next Q tile visible
#pragma unroll
    for (int k = 0; k < CPT; ++k) {
      const int qi = q0 + crow[k];
      if (crow[k] >= 0 && qi < p.Sq)
        store16(Og + (size_t)qi * p.ldo + cch[k] * 8, *(const u32x4*)(ot + crow[k] * QROW + cch[k] * 16));
    }
  }
}

template <typename T, int KS, int DF, bool ONES>
int launch_attn_kvres(const es_attn_desc& d, hipStream_t st) {
  // strips of queries, whole 32-query blocks: ONE round of workgroups (eight heads each, one wave per head; one workgroup per CU) - at
  // most 256 of them; a part-filled second round (308 workgroups in the first grid rule) cost a third of the launch
  const int per_strip = (d.heads + 7) / 8 * d.N;
  int strips = 256 / per_strip;
  const int blocks32 = (d.Sq + 31) / 32;
  strips = strips < 1 ? 1 : (strips > blocks32 ? blocks32 : strips);
  const int qper = ((blocks32 + strips - 1) / strips) * 32;
  strips = (d.Sq + qper - 1) / qper;
  const int nh = d.heads < 8 ? d.heads : 8;
  const size_t lds = (size_t)3 * 32 * (nh * d.d * 2 + 16);
  auto kfn = attention_kvres_kernel<T, KS, DF, ONES>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 32 * (8 * 80 * 2 + 16));
    attr_set = true;
  }
  dim3 grid(strips, (d.heads + 7) / 8, d.N);
  hipLaunchKernelGGL(kfn, grid, dim3(512), lds, st, d, qper);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

template <typename T, int QB>
int launch_attn40pp(const es_attn_desc& d, hipStream_t st) {
  constexpr int lds = 2 * (64 * (16 * 3 * 2 + 16) + 64 * (16 * 3 * 2 + 16));
  dim3 grid((d.Sq + 256 * QB - 1) / (256 * QB), d.heads, d.N);
  hipLaunchKernelGGL((attention40pp_kernel<T, QB>), grid, dim3(512), lds, st, d);
  return hipGetLastError() == hipSuccess ? 0 : -2;
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

// 1 (default; ES_ATTN_KVRES=0 turns it off, 2 forces it for every eligible shape): launches of enough samples to fill one round of its
// workgroups - where it wins (profiles/r05_xattn_bench.txt: 1.13-1.45x at head_dim 40 from 14 samples up, 1.21x at head_dim 80 with 112
// samples; it LOSES on the two-sample decoder launches and at head_dim 80 with 14-16 samples: 0.5-0.9x, those stay on the tiled kernels).
// Two earlier versions (Q / O in fragment-shaped accesses; a grid of 1.2 rounds) lost everywhere: what matters for this kernel is whole-row
// Q / O tiles AND exactly one round of workgroups.  It remains bound by its eight waves moving in lockstep (2.5-3 us per 32-query block
// against ~1.2 us of vector issue).
int attn_kvres = [] { const char* e = getenv("ES_ATTN_KVRES"); return e ? atoi(e) : 1; }();

template <typename T>
int dispatch(const es_attn_desc& d, hipStream_t st) {
  // 32 queries per wave (128 per block) only when that still yields >= 2 blocks per CU; else 16 per wave
  static const long long big_thr = getenv("ES_ATTN_BIG") ? atoll(getenv("ES_ATTN_BIG")) : 512;
  const bool big = (long long)((d.Sq + 127) / 128) * d.heads * d.N >= big_thr;
  // the text-token cross-attention (77 keys) of the 64 x 64 and 32 x 32 levels: K / V resident in registers
  if (attn_kvres && d.Skv <= 96 && d.Sq >= 64 && (d.d == 40 || d.d == 80)) {
    const long long groups = (long long)((d.heads + 7) / 8) * d.N;           // workgroups per strip of queries
    if (attn_kvres == 2 || groups >= (d.d == 40 ? 12 : 64))
      return d.d == 40 ? launch_attn_kvres<T, 3, 3, true>(d, st) : launch_attn_kvres<T, 5, 5, false>(d, st);
  }
  static const bool tile32 = !(getenv("ES_ATTN32") && atoi(getenv("ES_ATTN32")) == 0);   // A/B switch (tools)
  if (big && tile32) {
    // measured (tools/attn_bench.py, 14 x 8 x 4096^2): head_dim 40 476 -> 449 us; head_dim 80 is no faster (189 VGPRs,
    // two waves per SIMD instead of three) and stays on the 16x16 kernel unless asked for (ES_ATTN32=2, tests)
    static const bool tile32_80 = getenv("ES_ATTN32") && atoi(getenv("ES_ATTN32")) == 2;
    // 64 queries per wave where that still leaves two 256-query blocks per CU: the kernel is bound by LDS reads of the K / V
    // fragments (tool build with MFMAs, v_exp and row max removed: 265 of 448 us), and a fragment then serves two blocks
    static const int qb2 = getenv("ES_ATTN_QB") ? atoi(getenv("ES_ATTN_QB")) : 2;
    // the ping-pong kernel (8 waves, two groups one barrier out of phase): ES_ATTN_PP = 0 off, 1 = 32 queries per wave,
    // 2 = 64 queries per wave, unset (-1) = 32 queries per wave from 256 blocks of 256 queries on.
    // (Rounds 3-4 ran 64 queries per wave from 4096 blocks on: per tile it is the faster form in isolation (tools/attn_bench.py; stamps: 1730
    //  cycles per 64-query tile against 2 x 1050) - inside the pipeline it is the slower one, -0.6 % per batch-8 call and -1.1 % per 768 x 768
    //  call with 32: profiles/r05_pipeline_ab.txt.  The step runs at the package power limit there; the denser form buys clock it then loses.)
    static const int pp = getenv("ES_ATTN_PP") ? atoi(getenv("ES_ATTN_PP")) : -1;
    if (d.d == 40 && pp != 0 && d.Skv % 64 == 0 && d.Skv >= 128) {
      const long long wg256 = (long long)((d.Sq + 255) / 256) * d.heads * d.N;
      if (pp == 2) return launch_attn40pp<T, 2>(d, st);
      if (pp == 1 || (pp == -1 && wg256 >= 256)) return launch_attn40pp<T, 1>(d, st);
    }
    if (d.d == 40 && qb2 == 2 && (long long)((d.Sq + 255) / 256) * d.heads * d.N >= big_thr) {
      return launch_attn32<T, 3, 3, true, 2>(d, st);
    }
    if (d.d == 40) return launch_attn32<T, 3, 3, true>(d, st);
    if (d.d == 80 && tile32_80) return launch_attn32<T, 5, 5, false>(d, st);
  }
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

extern "C" int es_attention_set_kvres(int on) { const int prev = attn_kvres; attn_kvres = on; return prev; }

extern "C" int es_attention(const es_attn_desc* d, void* stream) {
  if (!d->q || !d->k || !d->v || !d->o) { es_set_error("es_attention: null pointer"); return -1; }
  if (d->d % 8 || d->ldq % 8 || d->ldk % 8 || d->ldv % 8 || d->ldo % 4) { es_set_error("es_attention: d and strides must be multiples of 8"); return -1; }
  if (d->Sq < 1 || d->Skv < 1 || d->N < 1 || d->heads < 1) { es_set_error("es_attention: empty problem"); return -1; }
  {   // the head widths dispatch() has a kernel for - checked HERE so that a dry recording (es_load_weights) rejects what a launch would
    static const int ok_d[] = {8, 16, 24, 32, 40, 48, 64, 80, 128, 160, 512};
    bool ok = false;
    for (int v : ok_d) ok = ok || d->d == v;
    if (!ok) { es_set_error("es_attention: unsupported head_dim (8,16,24,32,40,48,64,80,128,160,512)"); return -3; }
  }
  ES_PLAN_RECORD(ES_OP_ATTENTION, d, sizeof(*d));       // after validation: a rejected call never enters a recording plan
  hipStream_t st = (hipStream_t)stream;
  int rc = d->dtype == ES_F16 ? dispatch<f16>(*d, st) : dispatch<bf16>(*d, st);
  if (rc == -3) es_set_error("es_attention: unsupported head_dim (8,16,24,32,40,48,64,80,128,160,512)");
  else if (rc) es_set_error("es_attention: launch failed");
  return rc;
}
