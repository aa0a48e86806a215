// Implicit-GEMM convolution / linear for gfx950 on MFMA 16x16x32 (f16|bf16 in, fp32 accumulate).
//
//   D[cout][pixel] = sum_k W[cout][k] * X[pixel][k]      (A operand = weights, B operand = im2col'd activations)
//
// Tile: BM=128 pixels x BN (128|160) couts x BK=64, 256 threads = 4 waves as 2(M) x 2(N); each wave owns
// 64 pixels x BN/2 couts = 4 x (4|5) MFMA fragments.  Large launches (>= one round of 256 workgroups, K >= 1024) go to the
// 256 x 320 phase-interleaved tile of gemm_conv8p.hip (bn = 320), which moves half the L2->LDS bytes per FLOP.
// Tiny launches (a few dozen 128x128 tiles: the deep UNet levels at batch 1) use BM=64 x BN=64: they are bound by
// weight bytes in flight x HBM latency, and four times as many workgroups keep four times as many DMA rings busy.
//
// Staging is LDS-DMA (`buffer_load_dwordx4 ... lds`): every wave-instruction copies 8 tile rows x 128 B straight from
// global memory into LDS — no VGPR round trip and no ds_write traffic (ds_write_b128 tops out at ~79 B/clk/CU, which
// capped the register-staged version of this kernel).  The DMA destination is lane-linear, so the XOR swizzle that
// makes the ds_read_b128 fragment reads bank-conflict free is applied to the per-lane SOURCE address (LDS slot p of
// row r holds global chunk p ^ (r & 7)) and again on the read.  The im2col is address math on that source pointer:
// 3x3/1x1, stride 1|2, nearest-2x upsample and the channel concat of two tensors (UNet skip connections) select
// the row offset; padded taps are out-of-range offsets that the buffer descriptor's range check zero-fills.  LDS ring of 2 stages (2 workgroups per CU) or 4 stages
// (1 workgroup per CU, for grids too small to double up): the DMAs of the next STAGES-1 K-steps are in flight behind
// the MFMAs of the current one (counted s_waitcnt vmcnt + raw s_barrier), one barrier per K-step.
//
// Epilogue: bias / per-sample time-embedding add / SiLU / GEGLU / conditioning scale are applied in registers
// (fp32), the tile is transposed through LDS, and the residual add + store run as full-line 16-byte accesses along
// the NHWC channel dim (a lane-owns-4-channels direct store serialised on partial-line write round trips).
// Output and split-K slab stores of THIS file are write-through (`sc1`, common.h store16): a GEMM that ends with its
// output dirty in the XCDs' L2s pays the write-back at the kernel boundary; streamed out behind the remaining tiles it
// is free.  Measured on the batch-1 pipeline (tools/per_image_times.py, 5 processes each, same box): plain 508.8 ms per
// image, sc1 504.4 ms (+0.9 %).  (A first version of the asm store lacked the wait state after it: wrong results, caught
// by tests/test_engine_gpu.py, and "speed-ups" of 1-6 % that were the corrupted data's lower power draw.)
#ifndef ES_WT_STORES
#define ES_WT_STORES 1
#endif
#include "common.h"
#include "../../include/edgestyle_hip.h"
#include "plan.h"

// Tool-only: cache policy of the weight-tile DMAs (aux bits of buffer_load ... lds: 2 = nt).  Product builds use 0.
#ifndef ES_W_AUX
#define ES_W_AUX 0
#endif

int es_conv_gemm8p_launch(const es_gemm_desc& d, hipStream_t st);   // gemm_conv8p.hip: the 256 x 320 | 256 x 256 phase-interleaved tile
bool es_conv_gemm8p_takes(const es_gemm_desc& d);                    // ... and whether its epilogue has the form this launch needs

unsigned long long es_operand_limit_v = 0x7FFFFFFFull;   // bytes one activation operand of a launch may span (32-bit buffer offsets)

namespace {

constexpr int BK = 64;

// Tool-only ablation builds (tools/gemm_ablate.sh): time the K loop with one ingredient removed.  Results are wrong
// by construction; the product library is always built with ES_ABLATE == 0.
//   1: no MFMA   2: no LDS-DMA inside the loop   4: no fragment reads inside the loop   8: no barrier / DMA wait
//   16: no activation DMAs   32: no weight DMAs   64: every DMA out of range (issued, zero-filled, no memory traffic)
#ifndef ES_ABLATE
#define ES_ABLATE 0
#endif
// Tool-only diagnostic build (tools/gemm_stamps.py): ES_STAMPS=1 writes s_memtime stamps of the four waves of two
// workgroups (loop top / after the barrier / after the DMA issue / after the MFMAs of K-steps 8..15) into `prof`.
#ifndef ES_STAMPS
#define ES_STAMPS 0
#endif

typedef __attribute__((address_space(3))) void* lptr_t;

template <typename T>
ES_DEVICE void store_elems(T* o, const float* v, int n) {
  for (int r = 0; r < n; ++r) o[r] = from_f32<T>(v[r]);
}

// byte offset of (row i, tap tp, channel chan) of an activation source, or OOB (zero-filled by the buffer range check)
template <int XI>
ES_DEVICE void row_offsets_fn(unsigned (&voff)[XI], const int (&iy0)[XI], const int (&ix0)[XI], const int (&nb)[XI],
                              const int tp, const int cs, const int chan, const int ksize, const int KK, const int pad,
                              const int Hin, const int Win, const int upsample, const int Wsrc) {
  int ky = 0, kx = 0;
  if (ksize == 3) { ky = (tp * 11) >> 5; kx = tp - ky * 3; }         // tp in [0,9): tp/3 without a divide
  if (tp >= KK) { ky = pad; kx = pad; }                              // tail tap: the output pixel (stride 1, no upsample)
#pragma unroll
  for (int i = 0; i < XI; ++i) {
    const int iy = iy0[i] + ky, ix = ix0[i] + kx;
    const bool ok = (unsigned)iy < (unsigned)Hin && (unsigned)ix < (unsigned)Win;
    const int pix = nb[i] + (iy >> upsample) * Wsrc + (ix >> upsample);
    voff[i] = ok ? (unsigned)(((size_t)pix * cs + chan) * 2) : 0xFFFFFF00u;
  }
}

template <typename T, int BM, int BN, bool ALIGNED, int STAGES, int FM = 4 /* pixel fragments per wave */,
          bool LN = false /* LayerNorm folded into this linear layer */,
          bool KO = false /* chunk-major K order (es_gemm_desc.korder == 1): 3x3, 64-aligned channels */>
// (second launch bound = minimum waves per SIMD: 8-wave workgroups need 4 to keep two workgroups on a CU)
__global__ __launch_bounds__(BM * 8 / FM, (STAGES == 2 && BM == 128) ? (FM == 2 ? 4 : 2) : 1) void conv_gemm_kernel(
    const es_gemm_desc p, const int M, const int nk, const void* const tail1, const void* const tail2, const int tailC1,
    const int tailC2) {
  // (the 1x1 tail sources travel as plain kernel arguments, not through `p`: any read of the descriptor's t1/t2/Ct1/Ct2
  //  fields inside the K loop made hipcc keep the whole by-value descriptor in scratch memory - 3.4x slower kernels)
  constexpr int BKT = BK;            // K depth of a stage
  constexpr int RB = BKT * 2;        // bytes per tile row
  constexpr int CPR = BKT / 8;       // 16-byte chunks per row
  constexpr int RPP = 1024 / RB;     // rows per 1 KB LDS-DMA piece
  // waves: WM along pixels x 2 along couts; each owns 16*FM px x BN/2 couts.  FM = 4: 4 waves per 128-pixel tile.
  // FM = 2: the same tile on 8 waves (two per SIMD) for launches that leave a workgroup alone on its CU, where one
  // wave per SIMD serialises DMA issue, fragment reads and MFMAs (measured 0.7 us per K-step vs 0.21 us of MFMA).
  constexpr int WM = BM / (16 * FM);
  constexpr int NW = WM * 2;
  constexpr int NT = NW * 64;
  constexpr int XI = (BM / RPP) / NW; // activation DMA pieces (1 KB each) per wave per K-step
  constexpr int FN = BN / 32;        // cout fragments per wave (BN/2 couts)
  constexpr int WP = BN / RPP;       // weight DMA pieces per K-step (1 KB each), dealt block-wise to waves
  constexpr int WI = (WP + NW - 1) / NW;
  constexpr int XT = BM * RB;        // bytes per stage
  constexpr int WT = BN * RB;
  static_assert((size_t)BM * (BN * 2 + 16) <= (size_t)STAGES * (BM + BN) * RB, "the epilogue tile must fit the stage ring");
  static_assert(LN || (size_t)BM * (BN * 2 + 16) + (size_t)(BM / 8 + BM / 64) * BN * 8 <= (size_t)STAGES * (BM + BN) * RB,
                "the GroupNorm-statistics scratch (gn_emit_partials) must fit behind the epilogue tile");
  constexpr int BNP = BN;            // couts staged by the epilogue
  constexpr int EROW = BNP * 2 + 16; // epilogue tile row stride (bytes), padded against bank conflicts
  constexpr int NI = XI + WI;        // LDS-DMA instructions per wave per K-step
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave % WM, wn = wave / WM;
  // XCD-aware tile mapping.  Workgroups are dealt round-robin to the 8 XCDs (private L2 each), so linear block id b
  // runs on XCD b % 8.  Give every XCD one CONTIGUOUS chunk of the tile list, ordered so that consecutive tiles share
  // the larger operand: weights larger than activations (small M, deep layers) -> tile_m fastest, an XCD owns whole
  // (tile_n, k-slice) columns and each weight byte crosses the fabric once instead of 8 times; otherwise tile_n fastest.
  int tile_m, tile_n, z;
  {
    const int tm = (M + BM - 1) / BM, tn = p.rows_padded / BN;
    const int nwg = gridDim.x, b = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = b & 7;
    const int w = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);      // bijective for any nwg
    // (float-reciprocal divisions, exact below 2^24: an integer division is ~35 instructions, and the prologue of every
    //  workgroup - ~950 instructions, 1.3 us before its first DMA, tools/gemm_phase_stamps.py - is on the critical path of
    //  the ~100 single-round launches of a batch-1 step)
    const float inv_tm = __builtin_amdgcn_rcpf((float)tm), inv_tn = __builtin_amdgcn_rcpf((float)tn);
    if (p.xcd_m_fastest) { const int t = fast_div(w, tm, inv_tm); tile_m = w - t * tm; z = fast_div(t, tn, inv_tn); tile_n = t - z * tn; }
    else                 { const int t = fast_div(w, tn, inv_tn); tile_n = w - t * tn; z = fast_div(t, tm, inv_tm); tile_m = t - z * tm; }
  }
#if ES_STAMPS
  const int sblk = blockIdx.x == 40 ? 0 : (blockIdx.x == 333 ? 1 : -1);
  auto stamp = [&](int ks, int what) __attribute__((always_inline)) {
    if (p.prof && sblk >= 0 && lane == 0 && wave < 4 && ks >= 8 && ks < 16)
      p.prof[((sblk * 4 + wave) * 8 + (ks - 8)) * 4 + what] = __builtin_amdgcn_s_memtime();
  };
  auto pstamp = [&](int what) __attribute__((always_inline)) {
    if (p.prof && tid == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1))
      p.prof[256 + (blockIdx.x == 0 ? 0 : 1) * 8 + what] = __builtin_amdgcn_s_memtime();
  };
  pstamp(0);
#else
  if (p.prof && tid == 0) atomicMin(p.prof, (unsigned long long)__builtin_amdgcn_s_memrealtime());
#endif

  int ks0 = 0, ks1 = nk;
  if (p.splitk > 1) {                                   // nk * (z + 1) <= 360 * 32: exact in the float-reciprocal form
    const float inv_sk = __builtin_amdgcn_rcpf((float)p.splitk);
    ks0 = fast_div(nk * z, p.splitk, inv_sk);
    ks1 = fast_div(nk * (z + 1), p.splitk, inv_sk);
  }

  // ---------------- loader state ----------------
  const int lrow = lane / CPR;                     // row inside an RPP-row DMA piece
  const int lslot = lane % CPR;                    // LDS slot the DMA writes for this lane
  // global chunk that lands in that slot: the XOR swizzle is applied on the SOURCE side (lane-linear destination)
  const int kc = lslot ^ (lrow & 7);
  const int Ctot = p.C1 + p.C2;
  const int KK = p.ksize * p.ksize;
  // (struct fields that meet in a select are copied to locals first: `c ? p.a : p.b` otherwise becomes a load through a
  // selected ADDRESS, which pins the whole by-value descriptor in scratch memory - 700+ bytes per lane, 3.4x slower)
  const int pC1 = p.C1, pC2 = p.C2, pCt1 = tailC1, pCt2 = tailC2;
  const bool has_tail = tail1 != nullptr;
  const int Ctail = has_tail ? pCt1 + pCt2 : 0;        // 1x1 tail sources (conv_shortcut folded into conv2): tap index KK
  const int Ktrue = KK * Ctot + Ctail;
  const int Hin = p.Hsrc << p.upsample, Win = p.Wsrc << p.upsample;
  const int HWout = p.Hout * p.Wout;

  // Per-row im2col state: output pixel -> top-left source coordinate.  The loader uses raw buffer loads to LDS
  // (`buffer_load_dwordx4 ... offen lds`): the per-lane byte offset goes in a VGPR, the per-K-step uniform part
  // (channel offset inside the pixel, K offset of the weight row) in the scalar offset, and a padded / out-of-range
  // tap is simply an offset beyond the descriptor's num_records — the hardware range check zero-fills it, so there
  // is no zero page, no select on pointers and no 64-bit address arithmetic in the K loop.  Offsets per (tap, source)
  // are recomputed only when the tap or the concat source changes (every Cin/64 K-steps), not every K-step.
  constexpr unsigned OOB = 0xFFFFFF00u;
  // (integer division costs ~40 VALU instructions and sits on every launch's critical path: linear layers skip it,
  // convs use a float reciprocal + one correction step, exact below 2^24)
  int iy0[XI], ix0[XI], nb[XI];
  const bool small_m = M < (1 << 24);
  const float inv_hw = __builtin_amdgcn_rcpf((float)HWout), inv_w = __builtin_amdgcn_rcpf((float)p.Wout);
#pragma unroll
  for (int i = 0; i < XI; ++i) {
    const int m = tile_m * BM + RPP * (wave * XI + i) + lrow;
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
      if (p.x_nmod) n -= (n / p.x_nmod) * p.x_nmod;      // several groups read the same source samples
      nb[i] = n * p.Hsrc * p.Wsrc;
    }
  }
  // Chunk-major K order: the tap changes with EVERY K-step, so the im2col state of a row is kept as the source pixel
  // of tap (0,0) plus a tap-validity mask (bits 0-8: taps, bit 9: the row exists = the tail tap, bits 10/11: the
  // parities that place a nearest-2x upsampled tap) - a K-step's four offsets cost ~5 vector instructions each
  // instead of the ~12 of row_offsets_fn (that difference, issued beside the partner wave's MFMAs every K-step, is
  // what a first chunk-major attempt in round 3 lost 5-12 % to).
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
  const float* lnsel = LN ? (p.ngroups > 1 ? p.ln_colsum_g[grp] : p.ln_colsum) : nullptr;   // LayerNorm fold
  const auto rW = __builtin_amdgcn_make_buffer_rsrc(
      (void*)wsel, (short)0, (int)((size_t)p.rows_padded * p.Kpad * 2), 0x00020000);
  const int Nsrc = p.x_nmod ? p.x_nmod : p.N;          // samples the source tensors hold
  const auto rX1 = __builtin_amdgcn_make_buffer_rsrc(
      (void*)p.x, (short)0, (int)((size_t)Nsrc * p.Hsrc * p.Wsrc * p.C1 * 2), 0x00020000);
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
  for (int i = 0; i < WI; ++i) {
    const int piece = wave * WI + i;               // 8 weight rows; pieces beyond the tile read out of range (zeros)
    woff[i] = piece < WP ? (unsigned)(((size_t)(tile_n * BN + RPP * piece + lrow) * p.Kpad + kc * 8) * 2) : OOB;
  }

  int tap, cpos;
  {
    const int kg = ks0 * BKT + (ALIGNED ? 0 : kc * 8);
    tap = kg == 0 ? 0 : fast_div(kg, Ctot, __builtin_amdgcn_rcpf((float)Ctot));      // kg < Kpad < 2^24
    cpos = kg - tap * Ctot;
    if (ALIGNED && tap >= KK) { tap = KK; cpos = kg - KK * Ctot; }      // a split-K slice that starts inside the tail
    if constexpr (KO) {                                                 // K-step = (64-channel chunk, tap), then the tail
      const int kt = (KK * Ctot) / BKT;
      if (ks0 >= kt) { tap = KK; cpos = (ks0 - kt) * BKT; }
      else { const int ch = fast_div(ks0, 9, 1.0f / 9.0f); tap = ks0 - ch * 9; cpos = ch * BKT; }
    }
  }
  unsigned voff[XI];
#pragma unroll
  for (int i = 0; i < XI; ++i) voff[i] = OOB;
  const void* xbase = p.x;                                // (aligned launches) the current segment's source, its size, the scalar channel offset
  int xrec = 0, soff_x = 0, seg_left = 0;
  // (a free function, not a second lambda: with a closure nested inside issue_tile's closure hipcc stopped scalarising
  //  the captures once the tail sources were added, and kept them - and the by-value descriptor - in scratch memory)
  const int ks_tail = (KK * Ctot) / BKT;               // first K-step of the tail sources (aligned launches only)
  const int pk_ksize = p.ksize, pk_pad = p.pad, pk_up = p.upsample, pk_wsrc = p.Wsrc;
#define row_offsets(tp, cs, chan) \
  row_offsets_fn<XI>(voff, iy0, ix0, nb, (tp), (cs), (chan), pk_ksize, KK, pk_pad, Hin, Win, pk_up, pk_wsrc)
  auto issue_tile = [&](int ks, int stage) __attribute__((always_inline)) {
    char* xs = smem + stage * (XT + WT);
    char* ws = xs + XT;
    const int soff_w = ks * RB;
#pragma unroll
    for (int i = 0; i < WI; ++i)
      if (!(ES_ABLATE & 32) && wave * WI + i < WP)    // wave-uniform
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, (lptr_t)(ws + (wave * WI + i) * 1024), 16,
                                                 (ES_ABLATE & 64) ? (int)OOB : (int)woff[i], soff_w, 0, ES_W_AUX);
    if constexpr ((ES_ABLATE & 16) != 0) {
      // ablation: no activation DMAs
    } else if constexpr (ALIGNED) {
      // Tap-major order: inside a (tap, source) segment the next K-step reads the same rows 64 channels further on - ONE scalar add.
      // Round 3 re-derived the tap, the source (four-way select chains over base pointers and sizes) and the channel bookkeeping at
      // every K-step: ~75 scalar instructions per wave and K-step ahead of the MFMAs.  Now only at segment boundaries (every
      // C / 64 K-steps); measured on the 256 x 320 tile, which shares the scheme: +2.5 ... +7.5 % (profiles/r04_gemm_segments.txt).
      bool fast = false;
      if constexpr (!KO) fast = seg_left > 0;
      if (fast) {
        soff_x += 128; --seg_left;
      } else {
      const bool tail = KO ? ks >= ks_tail : tap >= KK;         // wave-uniform, like everything below
      int q0 = pC1, q1 = pC2, q2 = pCt1, q3 = pCt2;
      asm("" : "+s"(q0), "+s"(q1), "+s"(q2), "+s"(q3));
      const int c1 = tail ? q2 : q0;
      const int second = cpos >= c1 ? 1 : 0;              // a K-step never straddles taps or sources
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
          voff[i] = (tmask[i] & tb) ? __umul24((unsigned)pix, cs2) + (unsigned)(kc * 16) : 0xFFFFFF00u;
        }
      } else {
        row_offsets(tap, cs, kc * 8);
      }
#if ES_ABLATE & 64
#pragma unroll
      for (int i = 0; i < XI; ++i) voff[i] = OOB;
#endif
      // the source of this segment: wave-uniform selects of (base, bytes) - a four-way branch over four ready-made descriptors made
      // hipcc keep the by-value kernel descriptor in scratch memory (3.4x slower kernels)
      // (the empty asm makes the eight candidates opaque values: hipcc otherwise rewrites "select of loads from the
      //  closure" into "load from a selected closure ADDRESS", which pins the closure and every variable it captures in
      //  scratch memory - 300-800 bytes per lane and 3.4x slower kernels)
      const void *b0 = px, *b1 = px2, *b2 = pt1, *b3 = pt2;
      int n0 = nX1, n1 = nX2, n2 = nT1, n3 = nT2;
      asm("" : "+s"(b0), "+s"(b1), "+s"(b2), "+s"(b3), "+s"(n0), "+s"(n1), "+s"(n2), "+s"(n3));
      xbase = tail ? (second ? b3 : b2) : (second ? b1 : b0);
      xrec = tail ? (second ? n3 : n2) : (second ? n1 : n0);
      soff_x = cc * 2;
      if constexpr (KO) {
        if (tap < KK) {
          if (++tap == KK) { cpos += BKT; if (cpos >= Ctot) cpos = 0; else tap = 0; }     // next chunk, or on to the tail
        } else {
          cpos += BKT;
        }
      } else {
        // the whole segment at once: (tap, cpos) move to the first K-step of the NEXT segment
        const int seg_end = second ? (tail ? q2 + q3 : Ctot) : c1;
        seg_left = __builtin_amdgcn_readfirstlane(((seg_end - cpos) >> 6) - 1);
        cpos = seg_end;
        if (tap < KK && cpos >= Ctot) { cpos = 0; ++tap; }
        cpos = __builtin_amdgcn_readfirstlane(cpos);
        tap = __builtin_amdgcn_readfirstlane(tap);
      }
      }
      const auto rS = __builtin_amdgcn_make_buffer_rsrc((void*)xbase, (short)0, xrec, 0x00020000);
#pragma unroll
      for (int i = 0; i < XI; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rS, (lptr_t)(xs + (wave * XI + i) * 1024), 16, (int)voff[i], soff_x, 0, 0);
    } else {
      // small-Cin layers (conv_in, cond embedding): tap and channel differ per lane, single source
      if (ks * BKT + kc * 8 < Ktrue) {
        row_offsets(tap, p.C1, cpos);
      } else {
#pragma unroll
        for (int i = 0; i < XI; ++i) voff[i] = OOB;
      }
#pragma unroll
      for (int i = 0; i < XI; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rX1, (lptr_t)(xs + (wave * XI + i) * 1024), 16, (int)voff[i], 0, 0, 0);
    }
    if constexpr (!ALIGNED) {
      cpos += BKT;
      if (tap < KK) { while (cpos >= Ctot) { cpos -= Ctot; ++tap; } }      // (the tail is the last tap: cpos just runs on)
    }
  };

#undef row_offsets
  f32x4 acc[FN][FM];
#pragma unroll
  for (int i = 0; i < FN; ++i)
#pragma unroll
    for (int j = 0; j < FM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // prologue: STAGES-1 tiles in flight
#pragma unroll
  for (int s = 0; s < STAGES - 1; ++s)
    if (ks0 + s < ks1) issue_tile(ks0 + s, s);

#if ES_STAMPS
  pstamp(1);
#endif
  const int frow = lane & 15, fq = lane >> 4;
  // the epilogue's bias is fetched here, behind the first tile's DMA: loading it after the K loop put one more HBM
  // round trip (~1 us) on the critical path of every launch
  f32x4 bias[FN];
#pragma unroll
  for (int i = 0; i < FN; ++i)
    bias[i] = (bsel && p.splitk == 1) ? *(const f32x4*)(bsel + tile_n * BN + wn * (BN / 2) + fq * 4 + i * 16)
                                      : f32x4{0.f, 0.f, 0.f, 0.f};
  // LayerNorm fold: column sums of the (gamma-folded) weights for this lane's couts, and this wave's share of the row
  // statistics.  The X fragments a wave loads for its MFMAs cover its 16*FM rows x the whole K-step, and the two waves
  // of an N pair load the same ones: each takes HALF of them (fragments j < FM/2 or j >= FM/2) and lets the matrix
  // core do the sums: mfma(X, X) is the 16x16 Gram block whose DIAGONAL is sum_k x^2 of each row, mfma(ones, X) has
  // sum_k x of row p in every entry of column p.  Two MFMAs per fragment instead of ~16 v_dot2 (10-cycle issue each),
  // no extra LDS traffic, and the K reduction needs no cross-lane step.
  constexpr int LNH = LN ? FM / 2 : 1;
  f32x4 lncs[LN ? FN : 1];
  f32x4 ln_gram[LNH], ln_sum[LNH];
  typename Traits<T>::vec8 ln_ones;
  if constexpr (LN) {
#pragma unroll
    for (int jj = 0; jj < LNH; ++jj) { ln_gram[jj] = f32x4{0.f, 0.f, 0.f, 0.f}; ln_sum[jj] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int e = 0; e < 8; ++e) ln_ones[e] = from_f32<T>(1.0f);
  }
  int stage = 0, istage = STAGES - 1;
  // operand fragments of a K-step (read from LDS each K-step)
  typename Traits<T>::vec8 xa0[FM], wa0[FN], xa1[FM], wa1[FN];
  auto do_mfmas = [&]() __attribute__((always_inline)) {
#if ES_ABLATE & 1
#pragma unroll
    for (int j = 0; j < FM; ++j) { asm volatile("" ::"v"(xa0[j]), "v"(xa1[j])); }
#pragma unroll
    for (int i = 0; i < FN; ++i) { asm volatile("" ::"v"(wa0[i]), "v"(wa1[i])); }
#else
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
      for (int j = 0; j < FM; ++j) acc[i][j] = mfma16(wa0[i], xa0[j], acc[i][j]);
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
      for (int j = 0; j < FM; ++j) acc[i][j] = mfma16(wa1[i], xa1[j], acc[i][j]);
#endif
  };
  for (int ks = ks0; ks < ks1; ++ks) {
#if ES_STAMPS
    stamp(ks - ks0, 0);
#endif
    // tile ks must have landed; up to STAGES-2 younger tiles (NI DMA instructions each) may stay in flight
    if (!(ES_ABLATE & 8)) {
      if (STAGES > 2 && ks + STAGES - 2 < ks1) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((STAGES - 2) * NI) : "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();                    // everyone's DMA of tile ks landed; everyone finished tile ks-1
    }
#if ES_STAMPS
    stamp(ks - ks0, 1);
    if (ks == ks0) pstamp(2);
#endif
    const char* xs = smem + stage * (XT + WT);
    const char* ws = xs + XT;
    // Fragment reads are software-pipelined against the MFMAs: the reads of the second 32-deep half are in flight
    // behind the first half's MFMAs, and the next tile's DMA issue (address math + 9 LDS-DMA instructions) sits
    // between the two read groups where it covers the first group's LDS latency.
    if (!(ES_ABLATE & 4) || ks == ks0) {
#pragma unroll
    for (int j = 0; j < FM; ++j) {
      const int row = wm * (16 * FM) + j * 16 + frow;
      xa0[j] = as_vec8<T>(*(const u32x4*)(xs + row * 128 + (((0 + fq) ^ (row & 7)) << 4)));
    }
#pragma unroll
    for (int i = 0; i < FN; ++i) {
      const int row = wn * (BN / 2) + i * 16 + frow;
      wa0[i] = as_vec8<T>(*(const u32x4*)(ws + row * 128 + (((0 + fq) ^ (row & 7)) << 4)));
    }
    }
    if (!(ES_ABLATE & 2) && ks + STAGES - 1 < ks1) issue_tile(ks + STAGES - 1, istage);
#if ES_STAMPS
    stamp(ks - ks0, 2);
#endif
    if (!(ES_ABLATE & 4) || ks == ks0) {
#pragma unroll
    for (int j = 0; j < FM; ++j) {
      const int row = wm * (16 * FM) + j * 16 + frow;
      xa1[j] = as_vec8<T>(*(const u32x4*)(xs + row * 128 + (((4 + fq) ^ (row & 7)) << 4)));
    }
#pragma unroll
    for (int i = 0; i < FN; ++i) {
      const int row = wn * (BN / 2) + i * 16 + frow;
      wa1[i] = as_vec8<T>(*(const u32x4*)(ws + row * 128 + (((4 + fq) ^ (row & 7)) << 4)));
    }
    }
#ifndef ES_LN_NOGRAM            // tool-only ablation: the row statistics' MFMAs removed (results are wrong by construction)
#define ES_LN_NOGRAM 0
#endif
    if constexpr (LN && !ES_LN_NOGRAM) {
#pragma unroll
      for (int jj = 0; jj < LNH; ++jj) {
        const auto f0 = wn ? xa0[LNH + jj] : xa0[jj];
        const auto f1 = wn ? xa1[LNH + jj] : xa1[jj];
        ln_gram[jj] = mfma16(f0, f0, ln_gram[jj]);
        ln_gram[jj] = mfma16(f1, f1, ln_gram[jj]);
        ln_sum[jj] = mfma16(ln_ones, f0, ln_sum[jj]);
        ln_sum[jj] = mfma16(ln_ones, f1, ln_sum[jj]);
      }
    }
    do_mfmas();
#if ES_STAMPS
    stamp(ks - ks0, 3);
#endif
    istage = istage + 1 == STAGES ? 0 : istage + 1;
    stage = stage + 1 == STAGES ? 0 : stage + 1;
  }

#if ES_STAMPS
  pstamp(3);
#endif
  // ---------------- split-K: raw fp32 partials (16 B per lane) ----------------
  const int prow = wm * (16 * FM) + frow;                 // + j*16 : pixel row inside the tile
  const int pcol = wn * (BN / 2) + fq * 4;                // + i*16 : cout column inside the tile
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
#if !ES_STAMPS
    if (p.prof && tid == 0) atomicMax(p.prof + 1, (unsigned long long)__builtin_amdgcn_s_memrealtime());
#endif
    return;
  }

  // ---------------- epilogue phase A: registers -> LDS tile [pixel][cout] in T ----------------
  const bool geglu = p.act == ES_ACT_GEGLU;
  const int Cstore = geglu ? p.Cout / 2 : p.Cout;
  float scale = p.out_scale;
  if (p.out_scale_dev) scale *= *p.out_scale_dev;
  char* et = smem;
  T* outp = (T*)p.out;
  const T* resp = (const T*)p.residual;
  // residual rows are requested before the LDS transposition, so their round trip overlaps phase A
  constexpr int RPF = (BM * (BN / 8) + NT - 1) / NT;     // 16-byte residual chunks per thread
  u32x4 rpre[RPF];
  const bool vec_store = (Cstore & 7) == 0;
  {
    if (resp && vec_store) {
      const int CH = (geglu ? BN / 2 : BN) / 8;
      const float inv_ch = __builtin_amdgcn_rcpf((float)CH);
      const int c_tile = tile_n * (geglu ? BN / 2 : BN);
#pragma unroll
      for (int k = 0; k < RPF; ++k) {
        const int idx = tid + k * NT;
        const int row = fast_div(idx, CH, inv_ch), ch = idx - row * CH;
        const int m = tile_m * BM + row, c = c_tile + ch * 8;
        rpre[k] = u32x4{0u, 0u, 0u, 0u};
        if (idx < BM * CH && m < M && c < Cstore) rpre[k] = *(const u32x4*)(resp + (size_t)m * Cstore + c);
      }
    }
  }
  float* rowstat = (float*)(smem + (size_t)STAGES * (XT + WT) - BM * 8);    // [BM][2] mean, rstd: beyond the epilogue tile
  {
    __syncthreads();                                      // stage buffers no longer read
    if constexpr (LN) {
      // (the column sums are fetched here, not before the K loop: 4-5 more live quads push the 8-wave variants past
      // 128 VGPRs = one workgroup per CU, which costs far more than this one L2 round trip behind the barrier)
#pragma unroll
      for (int i = 0; i < FN; ++i) lncs[i] = *(const f32x4*)(lnsel + tile_n * BN + wn * (BN / 2) + fq * 4 + i * 16);
      // row p of fragment jj: sum x is in every register of the lanes with frow == p; sum x^2 is the Gram diagonal,
      // register p % 4 of the lane group p / 4 - moved to group 0 by two half-swaps
#pragma unroll
      for (int jj = 0; jj < LNH; ++jj) {
        const float dq = fq == (frow >> 2) ? ln_gram[jj][frow & 3] : 0.f;
        const float q_ = xor32_sum(xor16_sum(dq));
        if (fq == 0) {
          const int row = wm * (16 * FM) + ((wn ? LNH : 0) + jj) * 16 + frow;
          const float inv_c = 1.0f / (float)p.C1;
          const float mean = ln_sum[jj][0] * inv_c;
          float var = q_ * inv_c - mean * mean;
          var = var < 0.f ? 0.f : var;
          rowstat[row * 2] = mean;
          rowstat[row * 2 + 1] = rsqrtf(var + p.ln_eps);
        }
      }
      __syncthreads();
    }
    // The common forms as straight-line code (round 4: the options used to be tested per VALUE inside the unrolled loops - ~35 instructions
    // per value; tools/epi8p_stamps.py measured the same code at 40 % of a level-0 convolution's time on the 256 x 320 tile): no
    // activation, and the time embedding - if any - as ONE row per workgroup (all BM pixels of the tile in one sample: H*W % BM == 0).
    // Same arithmetic in the same order as the general form below: (acc + bias) + temb, * scale.
    const bool tuni = !LN && p.temb && (HWout % BM) == 0;  // (LayerNorm-folded launches are linear layers: no time embedding, no registers for one)
    if (!geglu && p.act == ES_ACT_NONE && (!p.temb || tuni)) {
      f32x4 tvv[LN ? 1 : FN];
#pragma unroll
      for (int i = 0; i < (LN ? 1 : FN); ++i) tvv[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (tuni) {
        const int m0 = tile_m * BM < M ? tile_m * BM : M - 1;
        const T* trow = (const T*)p.temb + (size_t)(m0 / HWout) * p.temb_stride;
#pragma unroll
        for (int i = 0; i < (LN ? 0 : FN); ++i) {
          const int c = tile_n * BN + pcol + i * 16;
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
        if constexpr (LN) { ln_mean = rowstat[(prow + j * 16) * 2]; ln_rstd = rowstat[(prow + j * 16) * 2 + 1]; }
#pragma unroll
        for (int i = 0; i < FN; ++i) {
          typename Traits<T>::vec4 pk;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float a = acc[i][j][r];
            if constexpr (LN) a = ln_rstd * (a - ln_mean * lncs[i][r]);
            pk[r] = from_f32<T>((a + bias[i][r] + tvv[LN ? 0 : i][r]) * scale);
          }
          *(typename Traits<T>::vec4*)(et + (prow + j * 16) * EROW + (pcol + i * 16) * 2) = pk;
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < FM; ++j) {
        const int m = tile_m * BM + prow + j * 16;
        const int mc = m < M ? m : M - 1;
        const int n = small_m ? fast_div(mc, HWout, inv_hw) : mc / HWout;
        float ln_mean = 0.f, ln_rstd = 1.f;
        if constexpr (LN) { ln_mean = rowstat[(prow + j * 16) * 2]; ln_rstd = rowstat[(prow + j * 16) * 2 + 1]; }
        if (geglu) {
          if constexpr (FN % 2 == 0) {
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
              *(typename Traits<T>::vec4*)(et + (prow + j * 16) * EROW + ((wn * (BN / 2) + i * 16) / 2 + fq * 4) * 2) = pk;
            }
          }
        } else {
#pragma unroll
          for (int i = 0; i < FN; ++i) {
            const int c = tile_n * BN + pcol + i * 16;
            float tv[4] = {0.f, 0.f, 0.f, 0.f};
            if (p.temb && c < p.Cout) {
              const T* tp = (const T*)p.temb + (size_t)n * p.temb_stride + c;
              if (c + 3 < p.Cout) {
                const auto t4 = *(const typename Traits<T>::vec4*)tp;
#pragma unroll
                for (int r = 0; r < 4; ++r) tv[r] = to_f32(t4[r]);
              } else {
                for (int r = 0; r < 4 && c + r < p.Cout; ++r) tv[r] = to_f32(tp[r]);
              }
            }
            typename Traits<T>::vec4 pk;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              float a = acc[i][j][r];
              if constexpr (LN) a = ln_rstd * (a - ln_mean * lncs[i][r]);
              float x = a + bias[i][r] + tv[r];
              if (p.act == ES_ACT_SILU) x = silu_f(x);
              pk[r] = from_f32<T>(x * scale);
            }
            *(typename Traits<T>::vec4*)(et + (prow + j * 16) * EROW + (pcol + i * 16) * 2) = pk;
          }
        }
      }
    }
    __syncthreads();

    // ---------------- epilogue phase B: coalesced residual add + store along the channel dim ----------------
    const int BNo = geglu ? BN / 2 : BNP;                 // tile width in stored channels (this pass)
    const int c_tile = tile_n * (geglu ? BN / 2 : BN);
    if (vec_store) {
      const int CH = BNo / 8;
      const float inv_ch = __builtin_amdgcn_rcpf((float)CH);
      // one form per kind of residual (0 none, 1 plain, 2 value pair out, 3 value pair in and out), chosen once
      auto store_rows = [&](auto mode_c) __attribute__((always_inline)) {
        constexpr int MODE = decltype(mode_c)::value;
#pragma unroll
        for (int k = 0; k < RPF; ++k) {
          const int idx = tid + k * NT;
          const int row = fast_div(idx, CH, inv_ch), ch = idx - row * CH;
          const int m = tile_m * BM + row, c = c_tile + ch * 8;
          if (idx < BM * CH && m < M && c < Cstore) {
            auto v = as_vec8<T>(*(const u32x4*)(et + row * EROW + ch * 16));
            const size_t o = (size_t)m * Cstore + c;
            if constexpr (MODE >= 2) {
              // wide residual stream (es_gemm_desc.out_lo): the sum in fp32 over residual hi + lo, written back as hi + lo.  The lo
              // chunk is requested here, not ahead of the K loop: its registers would push the 8-wave tiles over their budget
              const auto rv = as_vec8<T>(rpre[k]);
              typename Traits<T>::vec8 lv, lo;
              if constexpr (MODE == 3) lv = as_vec8<T>(*(const u32x4*)((const T*)p.residual_lo + o));
#pragma unroll
              for (int e = 0; e < 8; ++e) {
                float sum = to_f32(v[e]) + to_f32(rv[e]);
                if constexpr (MODE == 3) sum += to_f32(lv[e]);
                v[e] = from_f32<T>(sum);
                lo[e] = from_f32<T>(sum - to_f32(v[e]));
              }
              store16((T*)p.out_lo + o, __builtin_bit_cast(u32x4, lo));
            } else if constexpr (MODE == 1) {
              const auto rv = as_vec8<T>(rpre[k]);
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] = from_f32<T>(to_f32(v[e]) + to_f32(rv[e]));
            }
            store16(outp + o, __builtin_bit_cast(u32x4, v));
            if (p.gn_part) *(u32x4*)(et + row * EROW + ch * 16) = __builtin_bit_cast(u32x4, v);   // the FINAL value back into the tile
          }
        }
      };
      if (p.out_lo) { if (p.residual_lo) store_rows(std::integral_constant<int, 3>{}); else store_rows(std::integral_constant<int, 2>{}); }
      else if (resp) store_rows(std::integral_constant<int, 1>{});
      else store_rows(std::integral_constant<int, 0>{});
      if (p.gn_part) {                                     // GroupNorm statistics of this tile for the consumer (es_gemm_desc.gn_part)
        __syncthreads();
        gn_emit_partials<T, NT>(et, EROW, BM, BNo, (float*)(et + BM * EROW), p.gn_part, tile_m * BM, M, c_tile, Cstore, HWout, p.gn_groups, tid);
      }
    } else {
      // narrow outputs (conv_out: 4 or 3 channels): scalar tail path
      const float inv_bno = __builtin_amdgcn_rcpf((float)BNo);
      for (int idx = tid; idx < BM * BNo; idx += NT) {
        const int row = fast_div(idx, BNo, inv_bno), cc = idx - row * BNo;
        const int m = tile_m * BM + row, c = c_tile + cc;
        if (m < M && c < Cstore) {
          float x = to_f32(*(const T*)(et + row * EROW + cc * 2));
          if (resp) x += to_f32(resp[(size_t)m * Cstore + c]);
          outp[(size_t)m * Cstore + c] = from_f32<T>(x);
        }
      }
    }
  }
#if !ES_STAMPS
  if (p.prof && tid == 0) atomicMax(p.prof + 1, (unsigned long long)__builtin_amdgcn_s_memrealtime());
#else
  pstamp(4);
#endif
}

template <typename T>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const es_gemm_desc p, const int M) {
  // one thread = 8 consecutive channels of one pixel: sum the fp32 partials, then the same epilogue.  Every global
  // load of the epilogue (bias, time embedding, residual) is issued before the slab loop: one round trip, not four.
  const int oct = p.rows_padded / 8;
  const int idx = blockIdx.x * 256 + threadIdx.x;         // M * oct < 2^31 (entry-point check)
  if (idx >= M * oct) return;
  const bool small = M * oct < (1 << 24);
  const int m = div_any(idx, oct, 1.0f / (float)oct, small);
  const int c0 = (idx - m * oct) * 8;
  if (c0 >= p.Cout) return;
  const int nv = p.Cout - c0 < 8 ? p.Cout - c0 : 8;
  const bool full = nv == 8 && (p.Cout & 7) == 0;        // 16-byte aligned rows of 8 valid channels
  const int n = div_any(m, p.Hout * p.Wout, 1.0f / (float)(p.Hout * p.Wout), small);
  const float* bsel = p.bias;
  if (p.ngroups > 1) {
    const int tm = m >> 7;
    bsel = p.bias_g[(tm >= p.mt_end[0]) + (tm >= p.mt_end[1]) + (tm >= p.mt_end[2])];
  }
  float bv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, tv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  float rv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (bsel) {                                             // bias is padded to rows_padded: always 8 readable values
    const f32x4 b0 = *(const f32x4*)(bsel + c0), b1 = *(const f32x4*)(bsel + c0 + 4);
#pragma unroll
    for (int r = 0; r < 4; ++r) { bv[r] = b0[r]; bv[4 + r] = b1[r]; }
  }
  if (p.temb) {
    const T* tp = (const T*)p.temb + (size_t)n * p.temb_stride + c0;
    if (full && (p.temb_stride & 7) == 0 && (((size_t)p.temb) & 15) == 0) {
      const auto t8 = as_vec8<T>(*(const u32x4*)tp);
#pragma unroll
      for (int r = 0; r < 8; ++r) tv[r] = to_f32(t8[r]);
    } else {
      for (int r = 0; r < nv; ++r) tv[r] = to_f32(tp[r]);
    }
  }
  if (p.residual) {
    const T* rp = (const T*)p.residual + (size_t)m * p.Cout + c0;
    if (full) {
      const auto r8 = as_vec8<T>(*(const u32x4*)rp);
#pragma unroll
      for (int r = 0; r < 8; ++r) rv[r] = to_f32(r8[r]);
    } else {
      for (int r = 0; r < nv; ++r) rv[r] = to_f32(rp[r]);
    }
  }
  float scale = p.out_scale;
  if (p.out_scale_dev) scale *= *p.out_scale_dev;
  // slabs are summed in slice order (deterministic), four slices' loads in flight at a time
  f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
  const size_t zstride = (size_t)M * p.rows_padded;
  const float* w = p.workspace + (size_t)m * p.rows_padded + c0;
  int z = 0;
  for (; z + 4 <= p.splitk; z += 4) {
    f32x4 a[4], b[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { a[u] = *(const f32x4*)(w + (z + u) * zstride); b[u] = *(const f32x4*)(w + (z + u) * zstride + 4); }
#pragma unroll
    for (int u = 0; u < 4; ++u) { s0 += a[u]; s1 += b[u]; }
  }
  for (; z < p.splitk; ++z) {
    s0 += *(const f32x4*)(w + z * zstride);
    s1 += *(const f32x4*)(w + z * zstride + 4);
  }
  float v[8] = {s0[0], s0[1], s0[2], s0[3], s1[0], s1[1], s1[2], s1[3]};
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    float x = v[r] + bv[r] + tv[r];
    if (p.act == ES_ACT_SILU) x = silu_f(x);
    x = to_f32(from_f32<T>(x * scale));                   // same rounding point as the fused epilogue
    v[r] = x + rv[r];
  }
  T* o = (T*)p.out + (size_t)m * p.Cout + c0;
  if (full && p.out_lo) {                                  // wide residual stream: see the fused epilogue
    typename Traits<T>::vec8 pk, lo;
    if (p.residual_lo) {
      const auto l8 = as_vec8<T>(*(const u32x4*)((const T*)p.residual_lo + (size_t)m * p.Cout + c0));
#pragma unroll
      for (int r = 0; r < 8; ++r) v[r] += to_f32(l8[r]);
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) { pk[r] = from_f32<T>(v[r]); lo[r] = from_f32<T>(v[r] - to_f32(pk[r])); }
    store16(o, __builtin_bit_cast(u32x4, pk));
    store16((T*)p.out_lo + (size_t)m * p.Cout + c0, __builtin_bit_cast(u32x4, lo));
  } else if (full) {
    typename Traits<T>::vec8 pk;
#pragma unroll
    for (int r = 0; r < 8; ++r) pk[r] = from_f32<T>(v[r]);
    store16(o, __builtin_bit_cast(u32x4, pk));
  } else {
    store_elems<T>(o, v, nv);
  }
}

// The same reduction for outputs whose GroupNorm statistics are handed over (es_gemm_desc.gn_part): one workgroup owns 64 pixels x
// CB channels (CB = a divisor of rows_padded that holds whole GroupNorm groups like an N tile does), so that the final values pass
// through an LDS tile and gn_emit_partials sums them in the order the fused epilogues use.  Same arithmetic per element as
// splitk_reduce_kernel (the two produce identical outputs).
template <typename T, int CB>
__global__ __launch_bounds__(64 * (CB / 8)) void splitk_reduce_gn_kernel(const es_gemm_desc p, const int M) {
  // one (pixel, 8 channels) item per thread, like splitk_reduce_kernel: every thread's slab loads are in flight at once (a first
  // version with 256 threads and 4-5 items per thread took 14 us where the plain reduce takes 8: its round trips ran one after another)
  constexpr int OCT = CB / 8, NTH = 64 * OCT, ITEMS = 1, EROW = CB * 2 + 16;
  __shared__ __attribute__((aligned(16))) char smem[64 * EROW + (8 + 1) * CB * 8];
  const int tid = threadIdx.x;
  const int nct = p.rows_padded / CB;
  const int mb = blockIdx.x / nct, ct = blockIdx.x - mb * nct;
  const int HW = p.Hout * p.Wout;
  float scale = p.out_scale;
  if (p.out_scale_dev) scale *= *p.out_scale_dev;
  const size_t zstride = (size_t)M * p.rows_padded;
#pragma unroll
  for (int it = 0; it < ITEMS; ++it) {
    const int idx = tid + it * NTH;
    const int row = idx / OCT, o = idx - row * OCT;
    const int m = mb * 64 + row, c0 = ct * CB + o * 8;
    u32x4 res = {0u, 0u, 0u, 0u};
    if (c0 < p.Cout) {                                     // (Cout % 8 == 0: whole octets)
      const int n = m / HW;
      const float* bsel = p.bias;
      if (p.ngroups > 1) {
        const int tm = m >> 7;
        bsel = p.bias_g[(tm >= p.mt_end[0]) + (tm >= p.mt_end[1]) + (tm >= p.mt_end[2])];
      }
      float bv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, tv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      float rv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (bsel) {
        const f32x4 b0 = *(const f32x4*)(bsel + c0), b1 = *(const f32x4*)(bsel + c0 + 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) { bv[r] = b0[r]; bv[4 + r] = b1[r]; }
      }
      if (p.temb) {
        const T* tp = (const T*)p.temb + (size_t)n * p.temb_stride + c0;
        if ((p.temb_stride & 7) == 0 && (((size_t)p.temb) & 15) == 0) {
          const auto t8 = as_vec8<T>(*(const u32x4*)tp);
#pragma unroll
          for (int r = 0; r < 8; ++r) tv[r] = to_f32(t8[r]);
        } else {
          for (int r = 0; r < 8; ++r) tv[r] = to_f32(tp[r]);
        }
      }
      if (p.residual) {
        const auto r8 = as_vec8<T>(*(const u32x4*)((const T*)p.residual + (size_t)m * p.Cout + c0));
#pragma unroll
        for (int r = 0; r < 8; ++r) rv[r] = to_f32(r8[r]);
      }
      f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
      const float* w = p.workspace + (size_t)m * p.rows_padded + c0;
      int z = 0;
      for (; z + 4 <= p.splitk; z += 4) {
        f32x4 a[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { a[u] = *(const f32x4*)(w + (z + u) * zstride); b[u] = *(const f32x4*)(w + (z + u) * zstride + 4); }
#pragma unroll
        for (int u = 0; u < 4; ++u) { s0 += a[u]; s1 += b[u]; }
      }
      for (; z < p.splitk; ++z) {
        s0 += *(const f32x4*)(w + z * zstride);
        s1 += *(const f32x4*)(w + z * zstride + 4);
      }
      float v[8] = {s0[0], s0[1], s0[2], s0[3], s1[0], s1[1], s1[2], s1[3]};
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        float x = v[r] + bv[r] + tv[r];
        if (p.act == ES_ACT_SILU) x = silu_f(x);
        x = to_f32(from_f32<T>(x * scale));
        v[r] = x + rv[r];
      }
      typename Traits<T>::vec8 pk, lo;
      if (p.out_lo) {
        if (p.residual_lo) {
          const auto l8 = as_vec8<T>(*(const u32x4*)((const T*)p.residual_lo + (size_t)m * p.Cout + c0));
#pragma unroll
          for (int r = 0; r < 8; ++r) v[r] += to_f32(l8[r]);
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) { pk[r] = from_f32<T>(v[r]); lo[r] = from_f32<T>(v[r] - to_f32(pk[r])); }
        store16((T*)p.out_lo + (size_t)m * p.Cout + c0, __builtin_bit_cast(u32x4, lo));
      } else {
#pragma unroll
        for (int r = 0; r < 8; ++r) pk[r] = from_f32<T>(v[r]);
      }
      res = __builtin_bit_cast(u32x4, pk);
      store16((T*)p.out + (size_t)m * p.Cout + c0, res);
    }
    *(u32x4*)(smem + row * EROW + o * 16) = res;
  }
  __syncthreads();
  gn_emit_partials<T, NTH>(smem, EROW, 64, CB, (float*)(smem + 64 * EROW), p.gn_part, mb * 64, M, ct * CB, p.Cout, HW, p.gn_groups, tid);
}

template <typename T>
int launch(const es_gemm_desc& d0, hipStream_t st) {
  es_gemm_desc d = d0;
  // the 256 x 320 tile keeps the common epilogue forms only (gemm_conv8p.hip); an activation, time-embedding rows that differ inside a
  // 128-pixel half or meet a residual, or a Cout that is no multiple of 8 run on the 128 x 160 tile: the same results
  if (d.bn == 320 && !es_conv_gemm8p_takes(d)) { d.bn = 160; d.stages = 2; }
  if (d.bn == 256 && !es_conv_gemm8p_takes(d)) { d.bn = 128; d.stages = 2; }      // (rows_padded % 256 == 0: the 128-wide tile fits too)
  const int M = d.N * d.Hout * d.Wout;
  const int nk = d.Kpad / BK;
  const int Ctot = d.C1 + d.C2;
  const bool aligned = (Ctot % BK == 0) && (d.C1 % BK == 0);
  // Pixel tile: 128 rows (4 or 8 waves) by default, 64 rows with bn = 64; bn = 320 is the 256-pixel tile of gemm_conv8p.hip
  const int tn = d.rows_padded / d.bn;
  const int bm = d.bn == 64 ? 64 : 128;
  dim3 grid(((M + bm - 1) / bm) * tn * d.splitk);
  int stages = d.stages;
  if (stages == 0) stages = 2;   // measured (tools/gemm_bench.py): 2 stages x 2 workgroups/CU beats a 3-4 deep ring at 1/CU
#define ES_LAUNCH_F(BMV, BNV, AL, ST, FMV)                                                                  \
  do {                                                                                                      \
    if (AL && d.korder) ES_LAUNCH_K(BMV, BNV, AL, ST, FMV, AL);                                             \
    else ES_LAUNCH_K(BMV, BNV, AL, ST, FMV, false);                                                         \
  } while (0)
#define ES_LAUNCH_K(BMV, BNV, AL, ST, FMV, KOV)                                                             \
  do {                                                                                                      \
    auto kfn = conv_gemm_kernel<T, BMV, BNV, AL, ST, FMV, false, KOV>;                                      \
    const size_t lds = (size_t)ST * (BMV + BNV) * BK * 2;                                                   \
    static bool attr_set = false;                                                                           \
    if (!attr_set) {                                                                                        \
      (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);    \
      attr_set = true;                                                                                      \
    }                                                                                                       \
    hipLaunchKernelGGL(kfn, grid, dim3(BMV * 8 / FMV), lds, st, d, M, nk, d.t1, d.t2, d.Ct1, d.Ct2);                                  \
  } while (0)
#define ES_LAUNCH_LN(BMV, BNV, ST, FMV)                                                                      \
  do {                                                                                                      \
    auto kfn = conv_gemm_kernel<T, BMV, BNV, true, ST, FMV, true>;                                          \
    const size_t lds = (size_t)ST * (BMV + BNV) * BK * 2;                                                   \
    static bool attr_set = false;                                                                           \
    if (!attr_set) {                                                                                        \
      (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);    \
      attr_set = true;                                                                                      \
    }                                                                                                       \
    hipLaunchKernelGGL(kfn, grid, dim3(BMV * 8 / FMV), lds, st, d, M, nk, d.t1, d.t2, d.Ct1, d.Ct2);                                  \
  } while (0)
#define ES_LAUNCH(BMV, BNV, AL, ST) ES_LAUNCH_F(BMV, BNV, AL, ST, 4)
#define ES_LAUNCH_ST(BNV)                                                                                   \
  do {                                                                                                      \
    if (d.waves == 8 && stages == 4 && BNV == 128) ES_LAUNCH_F(128, 128, true, 4, 2);                       \
    else if (d.waves == 8) ES_LAUNCH_F(128, BNV, true, 2, 2);                                               \
    else if (stages == 2) ES_LAUNCH(128, BNV, true, 2);                                                     \
    else if (stages == 3) ES_LAUNCH(128, BNV, true, 3);                                                     \
    else ES_LAUNCH(128, BNV, true, 4);                                                                      \
  } while (0)
  if (d.bn == 256) { if (es_conv_gemm8p_launch(d, st)) return -2; }
  else if (d.ln_colsum) {
    // LayerNorm-folded linear layers: the instantiations the planner can pick for a 1x1 launch
    const bool w8 = d.waves == 8;
    if (d.bn == 64)            { if (stages == 4) ES_LAUNCH_LN(64, 64, 4, 2); else ES_LAUNCH_LN(64, 64, 2, 2); }
    else if (d.bn == 128 && w8) { if (stages == 4) ES_LAUNCH_LN(128, 128, 4, 2); else ES_LAUNCH_LN(128, 128, 2, 2); }
    else if (d.bn == 128)       { if (stages == 4) ES_LAUNCH_LN(128, 128, 4, 4); else ES_LAUNCH_LN(128, 128, 2, 4); }
    else if (w8)                { ES_LAUNCH_LN(128, 160, 2, 2); }
    else                        { if (stages == 4) ES_LAUNCH_LN(128, 160, 4, 4); else ES_LAUNCH_LN(128, 160, 2, 4); }
  }
  else if (d.bn == 64)  { if (stages == 4) ES_LAUNCH_F(64, 64, true, 4, 2); else ES_LAUNCH_F(64, 64, true, 2, 2); }
  else if (d.bn == 320) { if (es_conv_gemm8p_launch(d, st)) return -2; }
  else if (d.bn == 128) { if (aligned) ES_LAUNCH_ST(128); else ES_LAUNCH(128, 128, false, 2); }
  else                  { if (aligned) ES_LAUNCH_ST(160); else ES_LAUNCH(128, 160, false, 2); }
#undef ES_LAUNCH_ST
#undef ES_LAUNCH
#undef ES_LAUNCH_F
#undef ES_LAUNCH_K
#undef ES_LAUNCH_LN
  if (d.splitk > 1 && d.gn_part) {
    // statistics hand-over: 64 pixels x CB channels per workgroup, CB like the N tile the statistics' two-entry scheme assumes
    // (CB channels per workgroup: 64 x CB / 8 <= 1024 threads, and CB must hold a GroupNorm group: es_conv_gemm checks cpg <= 64 here)
    const int cb = (d.bn == 320 || d.bn == 160) ? 80 : (d.bn == 128 ? 128 : 64);
    dim3 grid((unsigned)((M / 64) * (d.rows_padded / cb)));
    if (cb == 80) hipLaunchKernelGGL((splitk_reduce_gn_kernel<T, 80>), grid, dim3(64 * 10), 0, st, d, M);
    else if (cb == 128) hipLaunchKernelGGL((splitk_reduce_gn_kernel<T, 128>), grid, dim3(64 * 16), 0, st, d, M);
    else hipLaunchKernelGGL((splitk_reduce_gn_kernel<T, 64>), grid, dim3(64 * 8), 0, st, d, M);
  } else if (d.splitk > 1) {
    const long long total = (long long)M * (d.rows_padded / 8);
    hipLaunchKernelGGL(splitk_reduce_kernel<T>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, d, M);
  }
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

// ---- launches whose operands outgrow the kernels' 32-bit buffer offsets -------------------------------------------------------
// The kernels address activations, tail sources and outputs through buffer resources with 32-bit byte offsets (2 GiB each).  At 512 x 512
// that is 11 try-ons per launch (the batched VAE encode of the conditions first, then the level-0 feed-forward); an MI355X holds far more.
// Such a launch is run as several ordinary launches over runs of whole samples - per weight group first, then as many samples as fit -
// each with its operand pointers moved to its first sample: every tile of a sub-launch computes exactly what it would have computed in
// the whole launch (the tiles of a launch never interact), so nothing about the results depends on where the cuts fall.
#define ES_OPERAND_LIMIT es_operand_limit_v            /* 0x7FFFFFFF; tests lower it (es_set_operand_limit) to cut small launches */

unsigned long long sample_bytes(const es_gemm_desc& d) {          // of the widest activation operand, per sample
  const unsigned long long cs = d.act == ES_ACT_GEGLU ? d.Cout / 2 : d.Cout;
  unsigned long long b = d.x_nmod ? 0ull : (unsigned long long)d.Hsrc * d.Wsrc * (d.C1 > d.C2 ? d.C1 : d.C2) * 2;
  const unsigned long long o = (unsigned long long)d.Hout * d.Wout * cs * 2;
  b = b > o ? b : o;
  if (d.t1) { const unsigned long long t = (unsigned long long)d.Hout * d.Wout * (d.Ct1 > d.Ct2 ? d.Ct1 : d.Ct2) * 2; b = b > t ? b : t; }
  return b;
}

bool oversize(const es_gemm_desc& d) {
  if (sample_bytes(d) * (unsigned long long)d.N >= ES_OPERAND_LIMIT) return true;
  if (d.x_nmod && (unsigned long long)d.x_nmod * d.Hsrc * d.Wsrc * d.C1 * 2 >= ES_OPERAND_LIMIT) return true;
  return d.splitk > 1 && (long long)d.N * d.Hout * d.Wout * (d.rows_padded / 8) >= (1ll << 31);
}

int launch_in_chunks(const es_gemm_desc& d0, hipStream_t st, const char** why, bool dry) {      // dry: the checks alone
  const long long hw = (long long)d0.Hout * d0.Wout;
  const unsigned long long per = sample_bytes(d0);
  if (d0.gn_part) { *why = "es_conv_gemm: a launch larger than 2 GiB per operand cannot hand GroupNorm statistics over (gn_part)"; return -1; }
  if (d0.x_nmod && (unsigned long long)d0.x_nmod * d0.Hsrc * d0.Wsrc * d0.C1 * 2 >= ES_OPERAND_LIMIT) { *why = "es_conv_gemm: source larger than 2 GiB (32-bit buffer offsets)"; return -1; }
  long long ns = (long long)((ES_OPERAND_LIMIT - 1) / (per ? per : 1));
  if (d0.splitk > 1) { const long long cap = ((1ll << 31) - 1) / (hw * (d0.rows_padded / 8)); ns = ns < cap ? ns : cap; }
  if (hw == 1 && ns >= 256) ns -= ns % 256;                          // linear layers: rows are the samples; cut between 256-row tiles
  if (d0.x_nmod) ns -= ns % d0.x_nmod;
  if (ns < 1) { *why = "es_conv_gemm: one sample's operands exceed 2 GiB (32-bit buffer offsets)"; return -1; }
  const int ng = d0.ngroups > 1 ? d0.ngroups : 1;
  const long long cs = d0.act == ES_ACT_GEGLU ? d0.Cout / 2 : d0.Cout;
  long long g0 = 0;
  for (int g = 0; g < ng; ++g) {
    long long g1 = d0.N;
    if (d0.ngroups > 1) {
      const long long px = (long long)d0.mt_end[g] * 128;
      if (px % hw) { *why = "es_conv_gemm: a grouped launch larger than 2 GiB per operand needs groups of whole samples"; return -1; }
      g1 = px / hw;
    }
    if (d0.x_nmod && g0 % d0.x_nmod) { *why = "es_conv_gemm: a launch larger than 2 GiB per operand needs groups that start on a multiple of x_nmod"; return -1; }
    for (long long n0 = g0; n0 < g1; n0 += ns) {
      es_gemm_desc s = d0;
      s.N = (int)((g1 - n0) < ns ? (g1 - n0) : ns);
      if (d0.ngroups > 1) {
        s.w = d0.w_g[g]; s.bias = d0.bias_g[g];
        if (d0.ln_colsum) s.ln_colsum = d0.ln_colsum_g[g];
      }
      s.ngroups = 0;
      for (int k = 0; k < 4; ++k) s.mt_end[k] = 0x7FFFFFFF;
      auto adv = [](const void* p, long long bytes) -> const void* { return p ? (const void*)((const char*)p + bytes) : nullptr; };
      const long long src_px = n0 * d0.Hsrc * d0.Wsrc, out_px = n0 * hw;
      if (!d0.x_nmod) s.x = adv(d0.x, src_px * d0.C1 * 2);
      s.x2 = adv(d0.x2, src_px * d0.C2 * 2);
      s.t1 = adv(d0.t1, out_px * d0.Ct1 * 2);
      s.t2 = adv(d0.t2, out_px * d0.Ct2 * 2);
      s.temb = adv(d0.temb, n0 * d0.temb_stride * 2);
      s.residual = adv(d0.residual, out_px * cs * 2);
      s.residual_lo = adv(d0.residual_lo, out_px * cs * 2);
      s.out = (void*)adv(d0.out, out_px * cs * 2);
      s.out_lo = (void*)adv(d0.out_lo, out_px * cs * 2);
      if (dry) continue;
      const int rc = s.dtype == ES_F16 ? launch<f16>(s, st) : launch<bf16>(s, st);
      if (rc) { *why = "es_conv_gemm: launch failed"; return rc; }
    }
    g0 = g1;
  }
  return 0;
}

}  // namespace

extern "C" void es_set_error(const char* msg);

extern "C" unsigned long long es_set_operand_limit(unsigned long long bytes) {
  const unsigned long long prev = es_operand_limit_v;
  es_operand_limit_v = bytes && bytes < 0x7FFFFFFFull ? bytes : 0x7FFFFFFFull;
  return prev;
}

extern "C" size_t es_conv_gemm_workspace_bytes(const es_gemm_desc* d) {
  if (d->splitk <= 1) return 0;
  return (size_t)d->splitk * d->N * d->Hout * d->Wout * d->rows_padded * sizeof(float);
}

extern "C" int es_conv_gemm(const es_gemm_desc* d, void* stream) {
  const int Ctot = d->C1 + d->C2;
  const int Ktrue = d->ksize * d->ksize * Ctot + (d->t1 ? d->Ct1 + d->Ct2 : 0);
  if (!d->x || !d->out || (d->ngroups <= 1 && !d->w)) { es_set_error("es_conv_gemm: null pointer"); return -1; }
  if (d->ngroups > 4) { es_set_error("es_conv_gemm: at most 4 groups"); return -1; }
  if (d->ngroups > 1) {
    const int tm = (d->N * d->Hout * d->Wout + 127) / 128;
    for (int g = 0; g < d->ngroups; ++g)
      if (!d->w_g[g] || d->mt_end[g] <= (g ? d->mt_end[g - 1] : 0)) { es_set_error("es_conv_gemm: bad group table"); return -1; }
    if (d->mt_end[d->ngroups - 1] != tm || (d->N * d->Hout * d->Wout) % 128) { es_set_error("es_conv_gemm: groups must tile M in whole 128-pixel tiles"); return -1; }
    if (d->bn == 320 || d->bn == 256)
      for (int g = 0; g < d->ngroups; ++g)
        if (d->mt_end[g] & 1) { es_set_error("es_conv_gemm: 256-pixel tiles need groups of whole 256-pixel tiles"); return -1; }
  }
  if (d->bn != 64 && d->bn != 128 && d->bn != 160 && d->bn != 320 && d->bn != 256) { es_set_error("es_conv_gemm: bn must be 64, 128, 160, 256 or 320"); return -1; }
  if (d->bn == 256 && (d->C1 % BK || d->C2 % BK || (d->stages != 0 && d->stages != 2) || d->waves == 8 || d->korder || d->temb)) {
    es_set_error("es_conv_gemm: bn=256 is the 256-pixel phase-interleaved tile of the LayerNorm-folded / GEGLU linear layers: 64-aligned channels, 2 stages, tap-major K, no time embedding"); return -1; }
  if (d->bn == 64 && (d->C1 % BK || d->C2 % BK || d->stages == 3 || d->waves == 8 || d->act == ES_ACT_GEGLU)) {
    es_set_error("es_conv_gemm: bn=64 is the 64x64 tile: 64-aligned channels, 2 or 4 stages, no GEGLU"); return -1; }
  if (d->bn == 320 && (d->C1 % BK || d->C2 % BK || (d->stages != 0 && d->stages != 2) || d->act == ES_ACT_GEGLU || d->waves == 8)) {
    es_set_error("es_conv_gemm: bn=320 is the 256-pixel phase-interleaved tile: 64-aligned channels, 2 stages, no GEGLU"); return -1; }
  if (d->rows_padded % d->bn || d->rows_padded < d->Cout) { es_set_error("es_conv_gemm: bad rows_padded"); return -1; }
  if (d->Kpad % BK || d->Kpad < Ktrue) { es_set_error("es_conv_gemm: bad Kpad"); return -1; }
  if (d->C1 % 8 || d->C2 % 8 || (d->C2 && !d->x2)) { es_set_error("es_conv_gemm: channels must be multiples of 8"); return -1; }
  if (d->C2 && (d->C1 % BK || d->C2 % BK)) { es_set_error("es_conv_gemm: concatenated sources need C1, C2 multiples of 64"); return -1; }
  if ((size_t)d->rows_padded * d->Kpad * 2 >= 0x7FFFFFFFull) { es_set_error("es_conv_gemm: weights larger than 2 GiB (32-bit buffer offsets)"); return -1; }
  const bool chunked = oversize(*d);       // activations beyond the 32-bit buffer offsets: runs of whole samples, one launch each (launch_in_chunks)
  if (chunked) {
    const char* why = nullptr;
    if (launch_in_chunks(*d, nullptr, &why, true)) { es_set_error(why); return -1; }
  }
  if (d->ksize != 1 && d->ksize != 3) { es_set_error("es_conv_gemm: ksize must be 1 or 3"); return -1; }
  if (d->t1 && (d->stride != 1 || d->upsample || d->C1 % BK || d->C2 % BK || d->Ct1 % BK || d->Ct2 % BK || d->Ct1 < BK ||
                (d->Ct2 && !d->t2) || d->ln_colsum || d->Hout != d->Hsrc || d->Wout != d->Wsrc ||
                false)) {
    es_set_error("es_conv_gemm: 1x1 tail sources need stride 1, no upsample, same-size output, 64-aligned channels"); return -1; }
  if (!d->t1 && (d->t2 || d->Ct1 || d->Ct2)) { es_set_error("es_conv_gemm: tail fields set without t1"); return -1; }
  if (d->x_nmod < 0 || (d->x_nmod && (d->x2 || d->t1 || d->x_nmod > d->N))) { es_set_error("es_conv_gemm: x_nmod needs a single source and 0 < x_nmod <= N"); return -1; }
  if (d->out_lo && (!d->residual || (d->Cout & 7) || d->act == ES_ACT_GEGLU)) {
    es_set_error("es_conv_gemm: out_lo (wide residual stream) needs a residual and an output width that is a multiple of 8"); return -1; }
  if (d->residual_lo && !d->out_lo) { es_set_error("es_conv_gemm: residual_lo without out_lo"); return -1; }
  if (d->gn_part) {
    const int hw = d->Hout * d->Wout;
    if (d->gn_groups < 1 || d->Cout % d->gn_groups || (d->Cout & 7) || hw % 64 || d->act == ES_ACT_GEGLU || d->ln_colsum ||
        d->Cout / d->gn_groups > (d->bn == 320 ? 160 : d->bn) || (d->splitk > 1 && d->Cout / d->gn_groups > 64)) {
      es_set_error("es_conv_gemm: gn_part needs H*W % 64 == 0, Cout % 8 == 0 and whole GroupNorm groups no wider than the N tile"); return -1; }
  }
  if (d->korder != 0 && (d->korder != 1 || d->ksize != 3 || d->C1 % BK || d->C2 % BK || d->ln_colsum)) {
    es_set_error("es_conv_gemm: korder 1 (chunk-major K) needs ksize 3 and 64-aligned C1, C2"); return -1; }
  if (d->splitk < 1 || d->splitk > d->Kpad / BK) { es_set_error("es_conv_gemm: bad splitk"); return -1; }
  if (d->splitk > 1 && (!d->workspace || d->act == ES_ACT_GEGLU)) { es_set_error("es_conv_gemm: splitk needs workspace and no GEGLU"); return -1; }
  if (d->act == ES_ACT_GEGLU && ((d->bn != 128 && d->bn != 256) || d->Cout % 32)) { es_set_error("es_conv_gemm: GEGLU needs bn=128 | 256, Cout%32==0"); return -1; }
  if (d->N < 1 || d->Hout < 1 || d->Wout < 1) { es_set_error("es_conv_gemm: empty problem"); return -1; }
  if (d->stages != 0 && (d->stages < 2 || d->stages > 4)) { es_set_error("es_conv_gemm: stages must be 0 (auto), 2, 3 or 4"); return -1; }
  if (d->waves != 0 && d->waves != 4 && d->waves != 8) { es_set_error("es_conv_gemm: waves must be 0 (auto), 4 or 8"); return -1; }
  if (d->ln_colsum && (d->ksize != 1 || d->stride != 1 || d->C2 || d->C1 % BK || d->Kpad != d->C1 || d->splitk != 1 ||
                       d->bn == 320 || d->upsample || d->temb || d->stages == 3)) {
    es_set_error("es_conv_gemm: LayerNorm fold needs a plain linear layer: ksize 1, one source, K = C1 (multiple of 64), splitk 1, bn 64|128|160"); return -1; }
  if (d->ln_colsum && d->ngroups > 1)
    for (int g = 0; g < d->ngroups; ++g)
      if (!d->ln_colsum_g[g]) { es_set_error("es_conv_gemm: grouped LayerNorm fold needs ln_colsum_g for every group"); return -1; }
  if (d->waves == 8 && (d->bn == 320 || d->C1 % BK || d->C2 % BK || d->stages == 3 ||
                        (d->stages == 4 && d->bn != 128))) {
    es_set_error("es_conv_gemm: waves=8 is the 128-pixel tile on 8 waves: 64-aligned channels, 2 stages (4 with bn=128)"); return -1; }
  ES_PLAN_RECORD(ES_OP_CONV_GEMM, d, sizeof(*d));       // after validation: a rejected call never enters a recording plan
  hipStream_t st = (hipStream_t)stream;
  // the kernels find a tile's group as (t >= end[0]) + (t >= end[1]) + (t >= end[2]): unused entries must compare false
  // (a caller's zero-initialised table sent every tile of a TWO-group launch to groups 1..3: null weights, GPU fault)
  es_gemm_desc dd = *d;
  for (int g = dd.ngroups > 1 ? dd.ngroups : 0; g < 4; ++g) dd.mt_end[g] = 0x7FFFFFFF;
  if (chunked) {
    const char* why = "es_conv_gemm: launch failed";
    const int rc = launch_in_chunks(*d, st, &why, false);
    if (rc) es_set_error(why);
    return rc;
  }
  int rc = dd.dtype == ES_F16 ? launch<f16>(dd, st) : launch<bf16>(dd, st);
  if (rc) es_set_error("es_conv_gemm: launch failed");
  return rc;
}
