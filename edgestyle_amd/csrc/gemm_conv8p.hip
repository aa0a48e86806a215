// Implicit-GEMM convolution for gfx950, large launches: 256 pixels x 320 couts per workgroup, 8 waves, ONE workgroup
// per CU, phase-interleaved mainloop (the structure of the CDNA4 guide's 256^2 8-phase GEMM template, re-derived for
// this tile and for an im2col loader).
//
//   D[cout][pixel] = sum_k W[cout][k] * X[pixel][k]      (A operand = weights, B operand = im2col'd activations)
//
// Why a second kernel beside conv_gemm_kernel (gemm_conv.hip): that loop - 128 x 160 tile, one barrier per K-step,
// two independent workgroups per CU - tops out at 1.0-1.08 PFLOP/s on the big 3x3 launches; its K-step is bound by
// L2->LDS staging bytes per FLOP and by two workgroups issuing DMAs, fragment reads and MFMAs in lockstep.  Here:
//   * tile 256 x 320 x 64: half the staged bytes per FLOP (0.0070 B/FLOP against 0.0141);
//   * waves as 2 (pixel halves) x 4 (80 couts): 128 px x 80 couts = 8 x 5 MFMA 16x16x32 fragments = 160 accumulator
//     registers per lane; the two waves of a SIMD are w and w + 4 = the two pixel halves ("groups");
//   * a K-tile (64 deep) is FOUR phases (pixel quarter h, 32-deep K half kk) of 20 MFMAs each; a phase is
//        fragment reads (4 or 9 ds_read_b128) | LDS-DMA issue for the NEXT K-tile | s_barrier | lgkmcnt(0) |
//        20 MFMAs at s_setprio 1 | s_barrier
//     and group 1 runs ONE BARRIER behind group 0, so on every SIMD one wave issues its MFMAs while its partner
//     issues reads and DMAs: the matrix pipe sees a continuous MFMA stream from alternating waves;
//   * LDS ring of two K-tiles (2 x 72 KB); the 9 DMA pieces (1 KB each) a wave issues per K-tile are spread over the
//     four phases 2 | 3 | 2 | 2 in the order they are needed (weights, first pixel quarter, second pixel quarter) and
//     retired by COUNTED s_waitcnt vmcnt: vmcnt(2) in phase 3 (weights + first quarter of the next tile landed, the
//     second quarter stays in flight across the tile boundary), vmcnt(2) in phase 0 (second quarter landed, the two
//     pieces just issued stay in flight).  Never vmcnt(0) inside the loop.  (Tried and measured within +-2 % of this:
//     3 | 2 | 2 | 2, everything in phases 0-1, weights first / pixels second, no s_setprio, lgkmcnt(0) before the barrier.
//     A chunk-major K order - the nine taps of a 64-channel chunk back to back, for L2 hits on the activation re-reads -
//     measured 5-12 % SLOWER: consecutive K-tiles then read weight lines Cin * 2 bytes apart.)
// Hazards (slots = intervals between consecutive workgroup barriers; group 0 reads in even slots and computes in
// odd ones, group 1 the other way round):
//   RAW  a wave's counted wait sits BEFORE the first barrier of its phase; the reads of the data it retires sit at
//        least one full phase later for both groups (phase 3 wait -> phase 0 reads, phase 0 wait -> phase 1 reads):
//        one barrier more than the unstaggered minimum, as the guide prescribes for staggered groups.
//   WAR  a sub-buffer of tile t - 1 is re-staged (for tile t + 1) no earlier than phase 0 of tile t for the weights
//        (last read: phase 2 of tile t - 1, retired by lgkmcnt(0) + two barriers before any wave reaches that issue) and
//        phase 2 / 3 for the pixel quarters.
// The im2col loader, grouped weights, tail sources, split-K and the epilogue semantics are those of conv_gemm_kernel
// (same rounding points: results are bit-identical to the 128-pixel tile for splitk == 1).
#ifndef ES_WT_STORES
#define ES_WT_STORES 1
#endif
#include "common.h"
#include "../../include/edgestyle_hip.h"

// Tool-only builds (tools/ab8p.sh): ES8P_ABL removes one ingredient (results are wrong by construction).  ABL bits: 1 = every DMA out of range (issued, zero-filled, no memory traffic),
// 2 = no MFMAs, 4 = no barrier stagger, 8 = activation DMAs out of range only, 16 = weight DMAs out of range only, 32 = chunk-major order with the
// activation tile staged on tap 0 of every 64-channel chunk ONLY (taps 1-8 multiply whatever the buffer holds: the DMA count and L2->LDS bytes of an
// LDS-resident halo patch, on non-zero data - what that design could gain at most), 64 = no residual loads in the epilogue, 128 = no output stores.  The product library is built with ES8P_ABL == 0.
#ifndef ES8P_ABL
#define ES8P_ABL 0
#endif
// Tool-only diagnostic build (tools/epi8p_stamps.py): ES8P_STAMPS=1 writes s_memtime stamps of wave 0 of the first and the last workgroup
// (entry | prologue issued | first K-tile landed | K loop done | pass-0 tile written | pass-0 stored | pass-1 tile written | end) into prof[256 ...]
#ifndef ES8P_STAMPS
#define ES8P_STAMPS 0
#endif


namespace {

typedef __attribute__((address_space(3))) void* lptr_t;

// Pixel rows per workgroup are a template parameter of the kernel; only BM = 256 is instantiated.  Round 4 measured the same loop on a
// 128 x 320 tile (10 MFMAs per phase, for launches with fewer 256-pixel tiles than CUs): never ahead of the 128 x 160 tile with its two
// workgroups per CU (profiles/r04_bm128_ablation.txt: 844 against 1025 TFLOP/s on the level-1 grouped convolution of a batch-1 step).
constexpr int RB = 128;
constexpr unsigned OOB = 0xFFFFFF00u;
// N tile: 320 couts (FN = 5 fragments of 16 per wave, the convolutions) or - round 5 - 256 couts (FN = 4), the form the LayerNorm-folded
// linear layers and the GEGLU projections of the C = 1280 level run on: every N of those layers is a multiple of 256 (1280, 3840, 10240),
// the [16 hidden | 16 gate] row interleave of the GEGLU weights pairs up inside a wave's 64 couts, and the 32 accumulator registers
// less leave room for the row statistics of the fold.

template <int XI>
ES_DEVICE void row_offsets4(unsigned (&voff)[XI], const int (&iy0)[XI], const int (&ix0)[XI], const int (&nb)[XI],
                            const int tp, const int cs, const int chan, const int ksize, const int KK, const int pad,
                            const int Hin, const int Win, const int upsample, const int Wsrc) {
  int ky = 0, kx = 0;
  if (ksize == 3) { ky = (tp * 11) >> 5; kx = tp - ky * 3; }
  if (tp >= KK) { ky = pad; kx = pad; }                              // tail tap: the output pixel
#pragma unroll
  for (int i = 0; i < XI; ++i) {
    const int iy = iy0[i] + ky, ix = ix0[i] + kx;
    const bool ok = (unsigned)iy < (unsigned)Hin && (unsigned)ix < (unsigned)Win;
    const int pix = nb[i] + (iy >> upsample) * Wsrc + (ix >> upsample);
    voff[i] = ok ? (unsigned)(((size_t)pix * cs + chan) * 2) : OOB;
  }
}

template <typename T, bool KO /* chunk-major K order, es_gemm_desc.korder == 1 */, int BM /* 256 | 128 pixel rows */,
          int FN = 5 /* cout fragments per wave: N tile = 64 FN */, bool LN = false /* LayerNorm fold (es_gemm_desc.ln_colsum) */,
          bool GEGLU = false /* ES_ACT_GEGLU epilogue */>
__global__ __launch_bounds__(512, 2) void conv_gemm8p_kernel(const es_gemm_desc p, const int M, const int nk,
                                                             const void* const tail1, const void* const tail2,
                                                             const int tailC1, const int tailC2) {
  constexpr int BN = 64 * FN;                           // 320 | 256 couts per workgroup
  constexpr int WT = BN * RB;                           // 40 | 32 KB of weights per K-tile
  constexpr int WI = FN;                                // weight DMA pieces per wave per K-tile
  constexpr int EROW = BN * 2 + 16;                     // epilogue tile row stride (bytes): BN couts, 128 pixel rows per pass
  static_assert(FN == 5 || FN == 4, "N tile 320 or 256");
  static_assert(!(LN || GEGLU) || FN == 4, "the LayerNorm fold and the GEGLU epilogue live on the 256-wide tile");
  constexpr int XT = BM * RB, BUF = XT + WT;            // 32 | 16 KB of pixels + 40 | 32 KB of weights per K-tile
  constexpr int GP = BM / 2;                            // pixel rows per wave group
  constexpr int FM = GP / 16, FMH = FM / 2;             // pixel fragments per wave, per phase
  constexpr int XI = BM / 64, XQ = XI / 2;              // pixel DMA pieces per wave per K-tile, per pixel quarter
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp8 = wave >> 2;          // pixel half of the tile = stagger group (SIMD partners are w and w + 4)
  const int wn = wave & 3;             // 80-cout column of the tile

  int tile_m, tile_n, z;
  {
    const int tm = (M + BM - 1) / BM, tn = p.rows_padded / BN;
    const int nwg = gridDim.x, b = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = b & 7;
    const int w = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);      // bijective for any nwg
    const float inv_tm = __builtin_amdgcn_rcpf((float)tm), inv_tn = __builtin_amdgcn_rcpf((float)tn);
    if (p.xcd_m_fastest) { const int t = fast_div(w, tm, inv_tm); tile_m = w - t * tm; z = fast_div(t, tn, inv_tn); tile_n = t - z * tn; }
    else                 { const int t = fast_div(w, tn, inv_tn); tile_n = w - t * tn; z = fast_div(t, tm, inv_tm); tile_m = t - z * tm; }
  }
#if ES8P_STAMPS
  const int sblk = blockIdx.x == 0 ? 0 : (blockIdx.x == gridDim.x - 1 ? 1 : (blockIdx.x == gridDim.x / 2 ? 2 : -1));
  auto stamp = [&](int k) __attribute__((always_inline)) {
    if (p.prof && sblk >= 0 && tid == 0) {
      p.prof[256 + sblk * 8 + k] = __builtin_amdgcn_s_memtime();
      if (k == 0 || k == 7) p.prof[288 + sblk * 2 + (k ? 1 : 0)] = __builtin_amdgcn_s_memrealtime();      // 100 MHz: cycles / ticks = the clock the workgroup ran at
    }
  };
  stamp(0);
#else
  if (p.prof && tid == 0) atomicMin(p.prof, (unsigned long long)__builtin_amdgcn_s_memrealtime());
#endif

  int ks0 = 0, ks1 = nk;
  if (p.splitk > 1) {
    const float inv_sk = __builtin_amdgcn_rcpf((float)p.splitk);
    ks0 = fast_div(nk * z, p.splitk, inv_sk);
    ks1 = fast_div(nk * (z + 1), p.splitk, inv_sk);
  }

  // ---------------- loader state (as conv_gemm_kernel; ALIGNED form only: C1, C2, tails multiples of 64) ----------------
  const int lrow = lane >> 3;                      // row inside an 8-row DMA piece
  const int lslot = lane & 7;                      // LDS slot the DMA writes for this lane
  const int kc = lslot ^ (lrow & 7);               // global chunk that lands there (XOR swizzle on the SOURCE side)
  const int Ctot = p.C1 + p.C2;
  const int KK = p.ksize * p.ksize;
  const int pC1 = p.C1, pC2 = p.C2, pCt1 = tailC1, pCt2 = tailC2;
  const bool has_tail = tail1 != nullptr;
  const int Hin = p.Hsrc << p.upsample, Win = p.Wsrc << p.upsample;
  const int HWout = p.Hout * p.Wout;
  const bool small_m = M < (1 << 24);
  const float inv_hw = __builtin_amdgcn_rcpf((float)HWout), inv_w = __builtin_amdgcn_rcpf((float)p.Wout);
  // X pieces of this wave: i = XQ * h + e -> tile rows GP * grp8 + GP/2 * h + 8 * (XQ * wn + e) + lrow (its own pixel half)
  int iy0[XI], ix0[XI], nb[XI];
#pragma unroll
  for (int i = 0; i < XI; ++i) {
    const int m = tile_m * BM + grp8 * GP + (i / XQ) * (GP / 2) + (wn * XQ + (i % XQ)) * 8 + lrow;
    iy0[i] = -(1 << 20); ix0[i] = -(1 << 20); nb[i] = 0;
    if (m < M) {
      int n, oy, ox;
      if (HWout == 1) {
        n = m; oy = 0; ox = 0;
      } else if (small_m) {
        n = fast_div(m, HWout, inv_hw);
        const int rem = m - n * HWout;
        oy = fast_div(rem, p.Wout, inv_w);
        ox = rem - oy * p.Wout;
      } else {
        n = m / HWout;
        const int rem = m - n * HWout;
        oy = rem / p.Wout; ox = rem - oy * p.Wout;
      }
      iy0[i] = oy * p.stride - p.pad;
      ix0[i] = ox * p.stride - p.pad;
      if (p.x_nmod) n -= (n / p.x_nmod) * p.x_nmod;
      nb[i] = n * p.Hsrc * p.Wsrc;
    }
  }
  // chunk-major K order: per-row state = source pixel of tap (0,0) + tap-validity mask (see conv_gemm_kernel)
  int pix0[KO ? XI : 1];
  unsigned tmask[KO ? XI : 1];
  if constexpr (KO) {
#pragma unroll
    for (int i = 0; i < XI; ++i) {
      const int ay = iy0[i], ax = ix0[i];
      unsigned vy = 0, vx = 0;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        vy |= ((unsigned)(ay + k) < (unsigned)Hin ? 1u : 0u) << k;
        vx |= ((unsigned)(ax + k) < (unsigned)Win ? 1u : 0u) << k;
      }
      unsigned mk = ((vy & 1) ? vx : 0u) | ((vy & 2) ? vx << 3 : 0u) | ((vy & 4) ? vx << 6 : 0u);
      mk |= ay > -(1 << 19) ? 1u << 9 : 0u;
      mk |= (unsigned)(ay & p.upsample) << 10 | (unsigned)(ax & p.upsample) << 11;
      tmask[i] = mk;
      pix0[i] = nb[i] + (ay >> p.upsample) * p.Wsrc + (ax >> p.upsample);
    }
  }
  int grp = 0;
  if (p.ngroups > 1) {
    const int t128 = (tile_m * BM) / 128;               // group table is in 128-pixel units
    grp = (t128 >= p.mt_end[0]) + (t128 >= p.mt_end[1]) + (t128 >= p.mt_end[2]);
  }
  const void* wsel = p.ngroups > 1 ? p.w_g[grp] : p.w;
  const float* bsel = p.ngroups > 1 ? p.bias_g[grp] : p.bias;
  const auto rW = __builtin_amdgcn_make_buffer_rsrc((void*)wsel, (short)0, (int)((size_t)p.rows_padded * p.Kpad * 2), 0x00020000);
  const int Nsrc = p.x_nmod ? p.x_nmod : p.N;
  const void* const pt1 = tail1 ? tail1 : p.x;
  const void* const pt2 = tail2 ? tail2 : p.x;
  const void* const px = p.x;
  const void* const px2 = p.x2 ? p.x2 : p.x;
  const int nX1 = (int)((size_t)Nsrc * p.Hsrc * p.Wsrc * pC1 * 2);
  const int nX2 = (int)((size_t)Nsrc * p.Hsrc * p.Wsrc * (p.x2 ? pC2 : pC1) * 2);
  const int nT1 = has_tail ? (int)((size_t)p.N * p.Hout * p.Wout * pCt1 * 2) : 0;
  const int nT2 = tail2 ? (int)((size_t)p.N * p.Hout * p.Wout * pCt2 * 2) : 0;
  unsigned woff[WI];
#pragma unroll
  for (int i = 0; i < WI; ++i)
    woff[i] = (unsigned)(((size_t)(tile_n * BN + 8 * (wave * WI + i) + lrow) * p.Kpad + kc * 8) * 2);

  int tap, cpos;                                        // of the NEXT K-tile to stage
  {
    const int kg = ks0 * 64;
    tap = kg == 0 ? 0 : fast_div(kg, Ctot, __builtin_amdgcn_rcpf((float)Ctot));
    cpos = kg - tap * Ctot;
    if (tap >= KK) { tap = KK; cpos = kg - KK * Ctot; }
    if constexpr (KO) {                                 // K-tile = (64-channel chunk, tap), then the tail
      const int kt = (KK * Ctot) / 64;
      if (ks0 >= kt) { tap = KK; cpos = (ks0 - kt) * 64; }
      else { const int ch = fast_div(ks0, 9, 1.0f / 9.0f); tap = ks0 - ch * 9; cpos = ch * 64; }
    }
  }
  unsigned voff[XI];
#pragma unroll
  for (int i = 0; i < XI; ++i) voff[i] = OOB;
  const int ks_tail = (KK * Ctot) / 64;
  const int pk_ksize = p.ksize, pk_pad = p.pad, pk_up = p.upsample, pk_wsrc = p.Wsrc;

  // staging of one K-tile in three calls: weights [i0, i1), pixel quarter 0, pixel quarter 1
  auto issue_w = [&](int ks, int boff, int i0, int i1) __attribute__((always_inline)) {
    const int soff_w = ks * RB;
#pragma unroll
    for (int i = 0; i < WI; ++i)
      if (i >= i0 && i < i1)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, (lptr_t)(smem + boff + XT + (wave * WI + i) * 1024), 16,
                                                 (ES8P_ABL & (1 | 16)) ? (int)OOB : (int)woff[i], soff_w, 0, 0);
  };
  // source of the K-tile `ks` (wave-uniform): base pointer, records, channel stride, scalar channel offset
  const void* xbase = px;
  int xrec = 0, soff_x = 0;
  bool x_skip = false;                                  // ES8P_ABL & 32 only
  // (tap-major order) K-tiles of the current (tap, source) segment that are still to be prepared: inside a segment the next tile is
  // the same rows 64 channels further on - ONE scalar add; everything else (tap -> row offsets, which of the four sources, its
  // descriptor) changes only between segments.  Round 3 re-derived all of it every K-tile: ~75 scalar instructions (select chains
  // over the four sources, the tap / channel bookkeeping) in the segment of phase 1 that runs beside the partner wave's 20 MFMAs -
  // longer than those MFMAs, so it paced the slot (the chunk-major experiment of round 4, which ADDS ~25 instructions there, lost 7 %).
  int seg_left = 0;
  auto select_x = [&](int ks) __attribute__((always_inline)) {
    bool fast = false;
    if constexpr (!KO) fast = seg_left > 0;
    if (fast) {
      soff_x += 128; --seg_left;
    } else {
    const bool tail = KO ? ks >= ks_tail : tap >= KK;
    if constexpr (KO && (ES8P_ABL & 32) != 0) x_skip = !tail && tap != 0;
    int q0 = pC1, q1 = pC2, q2 = pCt1, q3 = pCt2;
    asm("" : "+s"(q0), "+s"(q1), "+s"(q2), "+s"(q3));
    const int c1 = tail ? q2 : q0;
    const int second = cpos >= c1 ? 1 : 0;              // a K-tile never straddles taps or sources
    const int cs = second ? (tail ? q3 : q1) : c1;
    const int cc = second ? cpos - c1 : cpos;
    if constexpr (KO) {
      int ky = (tap * 11) >> 5, kx = tap - ky * 3;
      if (tail) { ky = pk_pad; kx = pk_pad; }
      const unsigned tb = 1u << tap, cs2 = (unsigned)cs * 2u;
      const int dsc = ky * pk_wsrc + kx;
#pragma unroll
      for (int i = 0; i < XI; ++i) {
        int pix = pix0[i] + dsc;
        if (pk_up) pix = pix0[i] + ((int)(((tmask[i] >> 10) & 1) + ky) >> 1) * pk_wsrc + ((int)(((tmask[i] >> 11) & 1) + kx) >> 1);
        voff[i] = (tmask[i] & tb) ? __umul24((unsigned)pix, cs2) + (unsigned)(kc * 16) : OOB;
      }
    } else {
      row_offsets4<XI>(voff, iy0, ix0, nb, tap, cs, kc * 8, pk_ksize, KK, pk_pad, Hin, Win, pk_up, pk_wsrc);
    }
    const void *b0 = px, *b1 = px2, *b2 = pt1, *b3 = pt2;
    int n0 = nX1, n1 = nX2, n2 = nT1, n3 = nT2;
    asm("" : "+s"(b0), "+s"(b1), "+s"(b2), "+s"(b3), "+s"(n0), "+s"(n1), "+s"(n2), "+s"(n3));
    const void* const base = tail ? (second ? b3 : b2) : (second ? b1 : b0);
    const int nrec = tail ? (second ? n3 : n2) : (second ? n1 : n0);
    xbase = base; xrec = nrec;
    soff_x = cc * 2;
    if constexpr (KO) {
      if (tap < KK) {
        if (++tap == KK) { cpos += 64; if (cpos >= Ctot) cpos = 0; else tap = 0; }       // next chunk, or on to the tail
      } else {
        cpos += 64;
      }
    } else {
      // the whole segment at once: (tap, cpos) move to the first tile of the NEXT segment
      const int seg_end = second ? (tail ? q2 + q3 : Ctot) : c1;
      seg_left = __builtin_amdgcn_readfirstlane(((seg_end - cpos) >> 6) - 1);
      cpos = seg_end;
      if (tap < KK && cpos >= Ctot) { cpos = 0; ++tap; }
      cpos = __builtin_amdgcn_readfirstlane(cpos);
      tap = __builtin_amdgcn_readfirstlane(tap);
    }
    }
  };
  auto issue_x = [&](int boff, int h) __attribute__((always_inline)) {
    if constexpr (KO && (ES8P_ABL & 32) != 0) { if (x_skip) return; }
    const auto rS = __builtin_amdgcn_make_buffer_rsrc((void*)xbase, (short)0, xrec, 0x00020000);
#pragma unroll
    for (int i = 0; i < XI; ++i)
      if (i / XQ == h)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(
            rS, (lptr_t)(smem + boff + (grp8 * (GP / 8) + (i / XQ) * (GP / 16) + wn * XQ + (i % XQ)) * 1024), 16,
            (ES8P_ABL & (1 | 8)) ? (int)OOB : (int)voff[i], soff_x, 0, 0);
  };

  f32x4 acc[FN][FM];
#pragma unroll
  for (int i = 0; i < FN; ++i)
#pragma unroll
    for (int j = 0; j < FM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---------------- prologue: K-tile ks0 entirely, then the stagger ----------------
  const int nkt = ks1 - ks0;
  if (nkt > 0) {
    issue_w(ks0, 0, 0, WI);
    select_x(ks0);
    issue_x(0, 0);
    issue_x(0, 1);
  }
  const int frow = lane & 15, fq = lane >> 4;
  // per-lane fragment read offsets inside a K-tile buffer: row r of a sub-tile sits at r * 128, its 16-byte chunk c at
  // slot c ^ (r & 7); every fragment row of this lane has r & 7 == frow & 7
  const int xo0 = ((0 + fq) ^ (frow & 7)) << 4, xo1 = ((4 + fq) ^ (frow & 7)) << 4;
  const int xrow = (grp8 * GP + frow) * RB;
  const int wrow = XT + (wn * (16 * FN) + frow) * RB;
#if ES8P_STAMPS
  stamp(1);
#endif
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
#if ES8P_STAMPS
  stamp(2);
#endif
  if (!(ES8P_ABL & 4) && grp8 == 1) __builtin_amdgcn_s_barrier();            // group 1 runs one barrier behind group 0

  typename Traits<T>::vec8 xa[FMH], wa[FN];
  // LayerNorm fold: the row statistics on the matrix core, as conv_gemm_kernel computes them (mfma(X, X) = the Gram block whose
  // diagonal is sum x^2 of each row, mfma(ones, X) = sum x) and in the same order per row, so that the two tiles stay bit-identical:
  // the four waves of a pixel half read the same pixel fragments; wave wn takes fragment wn of each quarter (one more ds_read_b128 per
  // phase: its own copy, no run-time index into xa) - two more MFMAs per phase of 16.
  f32x4 ln_gram[LN ? 2 : 1], ln_sum[LN ? 2 : 1];
  typename Traits<T>::vec8 ln_ones, xl;
  if constexpr (LN) {
#pragma unroll
    for (int h = 0; h < 2; ++h) { ln_gram[h] = f32x4{0.f, 0.f, 0.f, 0.f}; ln_sum[h] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int e = 0; e < 8; ++e) ln_ones[e] = from_f32<T>(1.0f);
  }
  const int xlrow = xrow + wn * 16 * RB;                  // + h * (GP / 2) * RB: this wave's statistics fragment of quarter h
  int boff = 0;
  for (int t = 0; t < nkt; ++t) {
    const bool nxt = t + 1 < nkt;                         // wave-uniform
    const int ksn = ks0 + t + 1;
    const int nboff = boff ^ BUF;
    const char* xs = smem + boff;
#define ES_RD(off) as_vec8<T>(*(const u32x4*)(xs + (off)))
#define ES_MFMA(H)                                                                    \
    __builtin_amdgcn_s_barrier();                                                     \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                \
    __builtin_amdgcn_sched_barrier(0);                                                \
    __builtin_amdgcn_s_setprio(1);                                                    \
    if constexpr ((ES8P_ABL & 2) != 0) {                                              \
      _Pragma("unroll") for (int j = 0; j < FMH; ++j) asm volatile("" ::"v"(xa[j]));  \
      _Pragma("unroll") for (int i = 0; i < FN; ++i) asm volatile("" ::"v"(wa[i]));   \
    } else {                                                                          \
    if constexpr (LN) {                                                               \
      ln_gram[H] = mfma16(xl, xl, ln_gram[H]);                                        \
      ln_sum[H] = mfma16(ln_ones, xl, ln_sum[H]);                                     \
    }                                                                                 \
    _Pragma("unroll") for (int i = 0; i < FN; ++i)                                    \
      _Pragma("unroll") for (int j = 0; j < FMH; ++j)                                 \
        acc[i][(H) * FMH + j] = mfma16(wa[i], xa[j], acc[i][(H) * FMH + j]);          \
    }                                                                                 \
    __builtin_amdgcn_s_setprio(0);                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                \
    __builtin_amdgcn_s_barrier();
    // ---- phase 0: (h 0, kk 0) ----
#pragma unroll
    for (int j = 0; j < FMH; ++j) xa[j] = ES_RD(xrow + xo0 + (0 * (GP / 2) + j * 16) * RB);
    if constexpr (LN) xl = ES_RD(xlrow + xo0 + 0 * (GP / 2) * RB);
#pragma unroll
    for (int i = 0; i < FN; ++i) wa[i] = ES_RD(wrow + xo0 + i * 16 * RB);
    if (nxt) {
      issue_w(ksn, nboff, 0, 2);
      asm volatile("s_waitcnt vmcnt(2)" ::: "memory");    // the second pixel quarter of THIS tile has landed
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    ES_MFMA(0)
    // ---- phase 1: (h 1, kk 0) ----
#pragma unroll
    for (int j = 0; j < FMH; ++j) xa[j] = ES_RD(xrow + xo0 + (1 * (GP / 2) + j * 16) * RB);
    if constexpr (LN) xl = ES_RD(xlrow + xo0 + 1 * (GP / 2) * RB);
    if (nxt) { issue_w(ksn, nboff, 2, WI); select_x(ksn); }
    ES_MFMA(1)
    // ---- phase 2: (h 0, kk 1) ----
#pragma unroll
    for (int j = 0; j < FMH; ++j) xa[j] = ES_RD(xrow + xo1 + (0 * (GP / 2) + j * 16) * RB);
    if constexpr (LN) xl = ES_RD(xlrow + xo1 + 0 * (GP / 2) * RB);
#pragma unroll
    for (int i = 0; i < FN; ++i) wa[i] = ES_RD(wrow + xo1 + i * 16 * RB);
    if (nxt) issue_x(nboff, 0);
    ES_MFMA(0)
    // ---- phase 3: (h 1, kk 1) ----
#pragma unroll
    for (int j = 0; j < FMH; ++j) xa[j] = ES_RD(xrow + xo1 + (1 * (GP / 2) + j * 16) * RB);
    if constexpr (LN) xl = ES_RD(xlrow + xo1 + 1 * (GP / 2) * RB);
    if (nxt) {
      issue_x(nboff, 1);
      if (KO && (ES8P_ABL & 32) != 0 && x_skip) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(XQ) : "memory");   // weights + first pixel quarter of the next tile have landed
    }
    ES_MFMA(1)
#undef ES_MFMA
#undef ES_RD
    boff = nboff;
  }
  if (!(ES8P_ABL & 4) && grp8 == 0) __builtin_amdgcn_s_barrier();            // balance the stagger: every wave is past its last LDS read
#if ES8P_STAMPS
  stamp(3);
#endif

  const int prow = grp8 * GP + frow;                      // + j * 16 : pixel row inside the tile
  const int pcol = wn * (16 * FN) + fq * 4;               // + i * 16 : cout column inside the tile
  // ---------------- split-K: raw fp32 partials ----------------
  if (p.splitk > 1) {
    float* wsp = p.workspace + (size_t)z * M * p.rows_padded;
#pragma unroll
    for (int j = 0; j < FM; ++j) {
      const int m = tile_m * BM + prow + j * 16;
      if (m < M) {
#pragma unroll
        for (int i = 0; i < FN; ++i)
          store16(wsp + (size_t)m * p.rows_padded + tile_n * BN + pcol + i * 16, __builtin_bit_cast(u32x4, acc[i][j]));
      }
    }
    if (p.prof && tid == 0) atomicMax(p.prof + 1, (unsigned long long)__builtin_amdgcn_s_memrealtime());
    return;
  }

  {
  // ---------------- epilogue: two passes of 128 pixels (one wave group's half each) through an LDS tile [pixel][320 couts] ----------------
  // Round 3 ran the passes over cout halves with every run-time option (time embedding, activation, scale, the kinds of residual) tested
  // per VALUE inside fully unrolled loops: 15 k instructions, 111 k cycles per workgroup against 150 k for the whole K loop of a level-0
  // convolution (tools/epi8p_stamps.py).  Now the options are decided once per pass and the common forms are straight-line code:
  //   write phase (the four waves that own the pixel half, one per SIMD): (acc + bias) + temb, [SiLU], * scale -> LDS, the time-embedding
  //     row fetched once per wave when all 128 pixels belong to one sample (H*W % 128 == 0);
  //   store phase (all eight waves): 16-byte chunks of whole 640-byte rows, [+ residual | + residual pair -> value pair], coalesced stores.
  // Same arithmetic in the same order as conv_gemm_kernel: bit-identical outputs.
  const int Cstore = GEGLU ? p.Cout / 2 : p.Cout;
  float scale = p.out_scale;
  if (p.out_scale_dev) scale *= *p.out_scale_dev;
  // lane coordinates re-derived from the thread index (opaque to the compiler: nothing of the epilogue is kept in registers across the K loop)
  int tid_e = threadIdx.x;
  asm volatile("" : "+v"(tid_e));
  const int tid = tid_e, frow = tid_e & 15, fq = (tid_e >> 4) & 3;
  const int pcol = wn * (16 * FN) + fq * 4;
  char* et = smem;
  // LayerNorm fold: (mean, rstd) of the BM pixel rows, behind the epilogue tile (conv_gemm_kernel's arithmetic, value for value)
  float* rowstat = (float*)(smem + 2 * BUF - BM * 8);
  const float* lnsel = nullptr;
  if constexpr (LN) {
    lnsel = p.ngroups > 1 ? p.ln_colsum_g[grp] : p.ln_colsum;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const float dq = fq == (frow >> 2) ? ln_gram[h][frow & 3] : 0.f;
      const float q_ = xor32_sum(xor16_sum(dq));
      if (fq == 0) {
        const int row = grp8 * GP + h * (GP / 2) + wn * 16 + frow;
        const float inv_c = 1.0f / (float)p.C1;
        const float mean = ln_sum[h][0] * inv_c;
        float var = q_ * inv_c - mean * mean;
        var = var < 0.f ? 0.f : var;
        rowstat[row * 2] = mean;
        rowstat[row * 2 + 1] = rsqrtf(var + p.ln_eps);
      }
    }
    __syncthreads();
  }
  T* outp = (T*)p.out;
  const T* resp = (const T*)p.residual;
  constexpr int CH = (GEGLU ? BN / 2 : BN) / 8;            // 16-byte chunks per tile row (40 | 32; GEGLU stores half the columns: 16)
  constexpr int RPF = GP * CH / 512;                       // chunks per thread and pass
  const int cbase = tile_n * (GEGLU ? BN / 2 : BN);        // first STORED channel of the tile
  const bool tuni = p.temb != nullptr;                     // (es_conv_gemm8p_takes: then H*W % GP == 0 - one sample per pixel half - and no residual)
  // WMODE: 0 = bias only, 1 = bias + one time-embedding row per wave        SMODE: 0 = no residual, 1 = residual, 2 = value pair out, 3 = value pair in and out
  // (an activation, per-pixel time-embedding rows, Cout % 8 != 0: es_conv_gemm runs those on the 128 x 160 tile, es_conv_gemm8p_takes)
  auto passes = [&](auto wmode_c, auto smode_c) __attribute__((always_inline)) {
    constexpr int WMODE = decltype(wmode_c)::value, SMODE = decltype(smode_c)::value;
    constexpr bool RES = SMODE != 0;
    constexpr int PF = !RES ? 0 : (SMODE >= 2 ? RPF / 2 : RPF);   // residual chunks prefetched (value pairs: half of them, registers)
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      const int m_tile = tile_m * BM + pass * GP;
      u32x4 rpre[RES ? PF : 1];
      // the residual chunks this thread will add: requested by the group that does NOT write this pass while it waits for the tile, by the
      // writing group after its accumulators are in LDS (never live beside them: 160 + 40 registers would not fit)
      auto load_res = [&]() __attribute__((always_inline)) {
        if constexpr (RES) {
#pragma unroll
          for (int k = 0; k < PF; ++k) {
            const int idx = tid + k * 512;
            const int row = CH == 40 ? ((idx >> 3) * 13108) >> 16 : (CH == 32 ? idx >> 5 : idx >> 4), ch = idx - row * CH;   // idx / 40 for idx < 5120 | idx / 32 | idx / 16
            const int m = m_tile + row, c = cbase + ch * 8;
            rpre[k] = u32x4{0u, 0u, 0u, 0u};
            if (!(ES8P_ABL & 64) && resp && m < M && c < Cstore) rpre[k] = *(const u32x4*)(resp + (size_t)m * Cstore + c);
          }
        }
      };
      if (pass) __syncthreads();                           // previous pass's tile no longer read
      if (grp8 == pass) {
        char* wbase = et + frow * EROW + (wn * (16 * FN) + fq * 4) * 2;
        const int gbase = tile_n * BN;                       // first GEMM column (weight row) of the tile
        f32x4 bias[FN];
#pragma unroll
        for (int i = 0; i < FN; ++i)
          bias[i] = bsel ? *(const f32x4*)(bsel + gbase + pcol + i * 16) : f32x4{0.f, 0.f, 0.f, 0.f};
        f32x4 lncs[LN ? FN : 1];
        if constexpr (LN) {
#pragma unroll
          for (int i = 0; i < FN; ++i) lncs[i] = *(const f32x4*)(lnsel + gbase + pcol + i * 16);
        }
        {
          f32x4 tvv[WMODE == 1 ? FN : 1];
          if constexpr (WMODE == 1) {
            const int m0 = m_tile < M ? m_tile : M - 1;
            const int n = m0 / HWout;                      // (wave-uniform: the whole pixel half lies in sample n)
            const T* trow = (const T*)p.temb + (size_t)n * p.temb_stride;
#pragma unroll
            for (int i = 0; i < FN; ++i) {
              const int c = gbase + pcol + i * 16;
              tvv[i] = f32x4{0.f, 0.f, 0.f, 0.f};
              if (c + 3 < p.Cout) {
                const auto t4 = *(const typename Traits<T>::vec4*)(trow + c);
#pragma unroll
                for (int r = 0; r < 4; ++r) tvv[i][r] = to_f32(t4[r]);
              } else {
                for (int r = 0; r < 4 && c + r < p.Cout; ++r) tvv[i][r] = to_f32(trow[c + r]);
              }
            }
          }
#pragma unroll
          for (int j = 0; j < FM; ++j) {
            float ln_mean = 0.f, ln_rstd = 1.f;
            if constexpr (LN) { ln_mean = rowstat[(pass * GP + j * 16 + frow) * 2]; ln_rstd = rowstat[(pass * GP + j * 16 + frow) * 2 + 1]; }
            if constexpr (GEGLU) {
              // [16 hidden | 16 gate] rows: fragments (i, i + 1) of this lane are hidden and gate of the same 4 output channels
#pragma unroll
              for (int i = 0; i < FN; i += 2) {
                typename Traits<T>::vec4 pk;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                  float ah = acc[i][j][r], ag = acc[i + 1][j][r];
                  if constexpr (LN) {
                    ah = ln_rstd * (ah - ln_mean * lncs[i][r]);
                    ag = ln_rstd * (ag - ln_mean * lncs[i + 1][r]);
                  }
                  const float hv = ah + bias[i][r], gv = ag + bias[i + 1][r];
                  pk[r] = from_f32<T>(hv * gelu_f(gv) * scale);
                }
                *(typename Traits<T>::vec4*)(et + (j * 16 + frow) * EROW + ((wn * (16 * FN) + i * 16) / 2 + fq * 4) * 2) = pk;
              }
            } else {
#pragma unroll
            for (int i = 0; i < FN; ++i) {
              typename Traits<T>::vec4 pk;
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                float a = acc[i][j][r];
                if constexpr (LN) a = ln_rstd * (a - ln_mean * lncs[i][r]);
                float x = a + bias[i][r];
                if constexpr (WMODE == 1) x += tvv[i][r]; else x += 0.f;      // (+ 0: as the general form rounds a -0)
                pk[r] = from_f32<T>(x * scale);
              }
              *(typename Traits<T>::vec4*)(wbase + j * 16 * EROW + i * 32) = pk;
            }
            }
          }
        }
        load_res();
      } else {
        load_res();
      }
      __syncthreads();
#if ES8P_STAMPS
      stamp(4 + pass * 2);
#endif
      {
#pragma unroll
        for (int k = 0; k < RPF; ++k) {
          const int idx = tid + k * 512;
          const int row = CH == 40 ? ((idx >> 3) * 13108) >> 16 : (CH == 32 ? idx >> 5 : idx >> 4), ch = idx - row * CH;
          const int m = m_tile + row, c = cbase + ch * 8;
          if (m < M && c < Cstore) {
            auto v = as_vec8<T>(*(const u32x4*)(et + row * EROW + ch * 16));
            const size_t o = (size_t)m * Cstore + c;
            constexpr bool wide = SMODE >= 2, wide_in = SMODE == 3;
            if constexpr (wide) {                                    // wide residual stream: as in conv_gemm_kernel
              const auto rv = as_vec8<T>(k < PF ? rpre[k < PF ? k : 0] : *(const u32x4*)(resp + o));
              typename Traits<T>::vec8 lv, lo;
              if constexpr (wide_in) lv = as_vec8<T>(*(const u32x4*)((const T*)p.residual_lo + o));
#pragma unroll
              for (int e = 0; e < 8; ++e) {
                float sum = to_f32(v[e]) + to_f32(rv[e]);
                if constexpr (wide_in) sum += to_f32(lv[e]);
                v[e] = from_f32<T>(sum);
                lo[e] = from_f32<T>(sum - to_f32(v[e]));
              }
              store16((T*)p.out_lo + o, __builtin_bit_cast(u32x4, lo));
            } else if constexpr (SMODE == 1) {
              const auto rv = as_vec8<T>(rpre[k < PF ? k : 0]);
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] = from_f32<T>(to_f32(v[e]) + to_f32(rv[e]));
            }
            if (!(ES8P_ABL & 128)) store16(outp + o, __builtin_bit_cast(u32x4, v));
            else asm volatile("" ::"v"(v));
            if (p.gn_part) *(u32x4*)(et + row * EROW + ch * 16) = __builtin_bit_cast(u32x4, v);   // the FINAL value back into the tile
          }
          if constexpr (SMODE >= 2) __builtin_amdgcn_sched_barrier(0);   // (value pairs: chunk by chunk, or the tile reads of all ten are hoisted and spill)
        }
        if (p.gn_part) {                                   // GroupNorm statistics for the consumer (es_gemm_desc.gn_part)
          __syncthreads();
          gn_emit_partials<T, 512>(et, EROW, GP, BN, (float*)(et + GP * EROW), p.gn_part, m_tile, M, cbase, Cstore, HWout, p.gn_groups, tid);
        }
      }
#if ES8P_STAMPS
      stamp(5 + pass * 2);
#endif
    }
  };
  using std::integral_constant;
  if (tuni) passes(integral_constant<int, 1>{}, integral_constant<int, 0>{});
  else if (p.out_lo) { if (p.residual_lo) passes(integral_constant<int, 0>{}, integral_constant<int, 3>{}); else passes(integral_constant<int, 0>{}, integral_constant<int, 2>{}); }
  else if (resp) passes(integral_constant<int, 0>{}, integral_constant<int, 1>{});
  else passes(integral_constant<int, 0>{}, integral_constant<int, 0>{});
  }
#if !ES8P_STAMPS
  if (p.prof && tid == 0) atomicMax(p.prof + 1, (unsigned long long)__builtin_amdgcn_s_memrealtime());
#endif
}

}  // namespace

// the epilogue forms this tile implements (split-K launches write raw partials: every form)
extern "C" int es_conv_gemm8p_form_ok(int act, int cout, int has_temb, long long out_hw, int has_residual) {
  if (act != ES_ACT_NONE || (cout & 7)) return 0;
  if (has_temb && ((out_hw & 127) || has_residual)) return 0;
  return 1;
}
bool es_conv_gemm8p_takes(const es_gemm_desc& d) {
  if (d.bn == 256) {
    // the 256-wide form: plain, LayerNorm-folded and GEGLU epilogues (no time embedding; a residual only without GEGLU); split-K: plain only
    if (d.temb || (d.Cout & 7) || (d.act != ES_ACT_NONE && d.act != ES_ACT_GEGLU)) return false;
    if (d.act == ES_ACT_GEGLU && (d.residual || d.out_lo || d.gn_part || d.splitk > 1)) return false;
    if (d.ln_colsum && (d.splitk > 1 || d.gn_part)) return false;
    return true;
  }
  if (d.splitk > 1) return true;
  return es_conv_gemm8p_form_ok(d.act, d.Cout, d.temb != nullptr, (long long)d.Hout * d.Wout, d.residual != nullptr) != 0;
}

// called by es_conv_gemm (gemm_conv.hip) after validation, for d.bn == 320 | 256
int es_conv_gemm8p_launch(const es_gemm_desc& d, hipStream_t st) {
  if (!es_conv_gemm8p_takes(d)) return -2;
  const int M = d.N * d.Hout * d.Wout;
  const int nk = d.Kpad / 64;
  const int tn = d.rows_padded / d.bn;
  constexpr int bm = 256;
  dim3 grid(((M + bm - 1) / bm) * tn * d.splitk);
#define ES8P_LAUNCH(TT, KOV, FNV, LNV, GGV)                                                                              \
  do {                                                                                                                    \
    auto kfn = conv_gemm8p_kernel<TT, KOV, 256, FNV, LNV, GGV>;                                                           \
    constexpr size_t lds = 2 * ((size_t)256 * RB + (size_t)64 * FNV * RB);                                                \
    static bool attr_set = false;                                                                                         \
    if (!attr_set) {                                                                                                      \
      (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                  \
      attr_set = true;                                                                                                    \
    }                                                                                                                     \
    hipLaunchKernelGGL(kfn, grid, dim3(512), lds, st, d, M, nk, d.t1, d.t2, d.Ct1, d.Ct2);                                \
  } while (0)
#define ES8P_LAUNCH_T(TT)                                                                                                \
  do {                                                                                                                    \
    if (d.bn == 256) {                                                                                                    \
      const bool ln = d.ln_colsum != nullptr, gg = d.act == ES_ACT_GEGLU;                                                 \
      if (ln && gg) ES8P_LAUNCH(TT, false, 4, true, true);                                                                \
      else if (ln) ES8P_LAUNCH(TT, false, 4, true, false);                                                                \
      else if (gg) ES8P_LAUNCH(TT, false, 4, false, true);                                                                \
      else ES8P_LAUNCH(TT, false, 4, false, false);                                                                       \
    } else if (d.korder) ES8P_LAUNCH(TT, true, 5, false, false);                                                          \
    else ES8P_LAUNCH(TT, false, 5, false, false);                                                                         \
  } while (0)
  if (d.dtype == ES_F16) ES8P_LAUNCH_T(f16); else ES8P_LAUNCH_T(bf16);
#undef ES8P_LAUNCH_T
#undef ES8P_LAUNCH
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
