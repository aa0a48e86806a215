// Row-stationary short-K linear layer for gfx950:  out[M, N] = epilogue( LN?(X[M, K]) W[N, K]^T + bias ),  K = 320 | 640.
//
// The transformer projections of the two shallow UNet levels (to_q|k|v: K = C -> 3C, GEGLU: K = C -> 8C, C = 320 | 640)
// have few K-steps and many output columns.  The tiled kernel (gemm_conv.hip) re-stages the activation tile once per
// N tile and pays its pipeline fill + epilogue drain per 5-10 K-steps: 300-500 TFLOP/s.  Here one workgroup of 8 waves
// owns 256 rows for ALL of its output columns:
//   * a wave keeps its 32 rows x K of activations IN REGISTERS as MFMA B operands (80 | 160 VGPRs), loaded once,
//     LayerNorm-ed in registers (fp32 statistics; gamma / beta are folded into W / bias by ops.pack_weight_ln);
//   * the weights stream through LDS in 40 KB stages (64 | 32 output columns x K) filled by LDS-DMA
//     (`buffer_load_dwordx4 ... lds`, 5 per wave per stage, XOR swizzle on the source address) in a 3-deep ring: two
//     stages are always in flight behind the MFMAs of the current one, one counted `s_waitcnt vmcnt` + one raw barrier
//     per stage, and the stream never drains between "tiles" because there are none;
//   * per stage a wave issues NF x K/32 x 2 MFMAs (16x16x32) against NF x K/32 ds_read_b128 weight fragments - the
//     same LDS bytes per MFMA as the tiled kernel, with no activation reads at all;
//   * epilogue per stage in registers (bias, exact-erf GEGLU), staged through a wave-private LDS tile so that global
//     stores are full 128-byte lines (16 B per lane), write-through like the GEMM's.
// Small M (the batch-1 UNet decoder): the N range is split over `nslices` workgroups per row block, each re-reading
// its rows (cheap: they stay in L2) so the launch still fills the chip.
//
// Replaces (as called from model/controllora.py:205-238 through diffusers BasicTransformerBlock): norm1 -> attn1.to_q|k|v,
// norm3 -> ff.net.0 (GEGLU) at the 64x64 and 32x32 levels.  Same math as es_conv_gemm with ln_colsum / ES_ACT_GEGLU;
// results differ from it only by fp16 rounding of the normalised activations (tests/test_ops_gpu.py).
#ifndef ES_WT_STORES
#define ES_WT_STORES 1
#endif
#include "common.h"
#include "../../include/edgestyle_hip.h"
#include "plan.h"

// Tool-only ablation builds (tools/xs_ablate.sh): time the kernel with one ingredient removed.  Results are wrong by
// construction; the product library is always built with XS_ABLATE == 0.
//   1: no MFMA   2: no GELU (hidden * gate)   4: no global stores   8: no DMA wait / barrier   16: no DMA inside the loop
//   32: no weight fragment reads inside the K loop
#ifndef XS_ABLATE
#define XS_ABLATE 0
#endif
// Tool-only diagnostic build (tools/xs_stamps.py): XS_STAMPS=1 writes s_memtime stamps of waves 0 and 4 of two workgroups
// (after the barrier / after the MFMAs / after the epilogue of stages 4..11) into the buffer passed as `prof`.
#ifndef XS_STAMPS
#define XS_STAMPS 0
#endif

namespace {

typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int XS_ROWS = 256;          // rows per workgroup (8 waves x 32)
constexpr int XS_STAGES = 3;
constexpr int XS_STAGE_W = 40960;     // weight bytes per stage
constexpr int XS_STAGE = XS_STAGE_W + 256;   // + the stage's bias values (fp32)
constexpr int XS_OROW = 144;          // wave-private output staging: 32 rows x (128 + 16 pad) bytes
constexpr int XS_LDS = XS_STAGES * XS_STAGE + 8 * 32 * XS_OROW;

template <int N> ES_DEVICE void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <typename T, int KC /* K / 32 */, bool GEGLU, bool LN, bool RES = false, bool PP = false /* two-barrier ping-pong (below) */,
          bool GN = false /* GroupNorm in front: es_xs_desc.gn_part */>
__global__ __launch_bounds__(512, 2) void linear_xs_kernel(const es_xs_desc p) {
  constexpr int K = KC * 32;
  constexpr int NF = KC == 10 ? 4 : 2;          // 16-column fragments per stage
  constexpr int CH = NF * 16;                    // output columns (GEMM N) per stage
  constexpr int SUB = CH * 128;                  // bytes of one 64-deep sub-tile [CH rows][128 B]
  constexpr int PPS = CH / 8;                    // 1 KB DMA pieces per sub-tile
  constexpr int OUTB = GEGLU ? CH : CH * 2;      // stored bytes per row per stage
  constexpr int P = 128 / OUTB;                  // stages per full 128-byte output line
  static_assert(KC == 10 || KC == 20, "K = 320 or 640");
  static_assert(!RES || (P == 1 && !GEGLU && !LN), "residual: K = 320, a full output line per stage (the attention / proj output layers)");
  static_assert(NF * 16 * K * 2 == XS_STAGE_W, "stage geometry");
  static_assert(!GN || (!GEGLU && !LN && !RES), "GroupNorm in front: the plain projection (Transformer2DModel.norm -> proj_in)");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int frow = lane & 15, fq = lane >> 4;
  const int slice = blockIdx.x % p.nslices, rb = blockIdx.x / p.nslices;
#if XS_STAMPS
  const int sblk = blockIdx.x == 3 ? 0 : (blockIdx.x == 100 ? 1 : -1);
  auto stamp = [&](int ci, int what) __attribute__((always_inline)) {
    if (p.prof && sblk >= 0 && (wave & 3) == 0 && lane == 0 && ci >= 4 && ci < 12)
      p.prof[((sblk * 2 + (wave >> 2)) * 8 + (ci - 4)) * 3 + what] = __builtin_amdgcn_s_memtime();
  };
#else
  if (p.prof && tid == 0) atomicMin(p.prof, (unsigned long long)__builtin_amdgcn_s_memrealtime());
#endif

  // this workgroup's stages: [c0, c1), whole groups of P
  const int total = (p.Cout + CH - 1) / CH;
  const int c0 = slice * p.chunks_per_slice;
  int c1 = c0 + p.chunks_per_slice;
  c1 = c1 < total ? c1 : total;
  const int nch = c1 - c0;

  int grp = 0;
  if (p.ngroups > 1) {
    const int t128 = rb * (XS_ROWS / 128);
    grp = (t128 >= p.mt_end[0]) + (t128 >= p.mt_end[1]) + (t128 >= p.mt_end[2]);
  }
  const void* wsel = p.ngroups > 1 ? p.w_g[grp] : p.w;
  const float* bsel = p.ngroups > 1 ? p.bias_g[grp] : p.bias;
  const auto rW = __builtin_amdgcn_make_buffer_rsrc((void*)wsel, (short)0, (int)((size_t)p.rows_padded * K * 2), 0x00020000);
  const auto rB = __builtin_amdgcn_make_buffer_rsrc((void*)bsel, (short)0, (int)((size_t)p.rows_padded * 4), 0x00020000);
  const auto rX = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, (short)0, (int)((size_t)p.M * K * 2), 0x00020000);

  // ---------------- weight stream: DMA addressing ----------------
  const int lrow = lane >> 3;                    // row inside an 8-row piece
  const int kc8 = (lane & 7) ^ lrow;             // global 16-byte chunk landing in LDS slot (lane & 7): XOR swizzle on the source
  const unsigned wvoff = (unsigned)((lrow * K + kc8 * 8) * 2);
  auto issue = [&](int c, int stage) __attribute__((always_inline)) {
    char* sb = smem + stage * XS_STAGE;
    const int n0 = c * CH;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int piece = wave * 5 + i;            // 40 pieces per stage
      const int st = piece / PPS, g = piece - st * PPS;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, (lptr_t)(sb + st * SUB + g * 1024), 16, (int)wvoff,
                                               ((n0 + g * 8) * K + st * 64) * 2, 0, 0);
    }
    // the stage's bias values (every wave writes the same 256 bytes: keeps the per-wave DMA count uniform)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, (lptr_t)(sb + XS_STAGE_W), 4, lane * 4, n0 * 4, 0, 0);
  };
  constexpr int NDMA = 6;                        // VMEM instructions per wave per stage

  // ---------------- activations: 32 rows x K per wave, straight into MFMA B-operand registers ----------------
  const int r0 = rb * XS_ROWS + wave * 32;
  typename Traits<T>::vec8 xr[2][KC];
#pragma unroll
  for (int rf = 0; rf < 2; ++rf)
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) {
      const int row = r0 + rf * 16 + frow;
      const unsigned off = row < p.M ? (unsigned)(((size_t)row * K + kc * 32 + fq * 8) * 2) : 0xFFFFFF00u;
      xr[rf][kc] = as_vec8<T>(__builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rX, (int)off, 0, 0)));
    }
  if constexpr (GN) {
    // GroupNorm in front of the projection (Transformer2DModel.norm -> proj_in, eps 1e-6, no activation): x is the RAW tensor [N, hw, K]
    // and es_group_norm(stats_only) has left the per-(sample, pixel chunk, group) sums of x and x^2 in gn_part.  A row block lies inside ONE
    // sample (hw % 256 == 0): 16 threads per group fold the sample's chunks in a fixed order into (rstd, -mean * rstd), kept in the output
    // staging area (nobody stages an output before the first barrier of the main loop); then every lane normalises the values it holds:
    //   x' = x * (rstd_g * gamma_c) + (beta_c - mean_g * rstd_g * gamma_c)        rounded once to the compute dtype, like es_group_norm's output.
    // One read of x instead of three passes over it (statistics, apply: read + write, projection: read).
    float* gst = (float*)(smem + XS_STAGES * XS_STAGE);
    const int n = (int)(((long long)rb * XS_ROWS) / p.gn_hw);
    const int G = p.gn_groups, cpg = K / G;
    const float inv_cnt = 1.0f / ((float)cpg * (float)p.gn_hw);
    for (int g0 = 0; g0 < G; g0 += 32) {
      const int g = g0 + (tid >> 4), l = tid & 15;
      float s = 0.f, ss = 0.f;
      if (g < G)
        for (int ch = l; ch < p.gn_nchunk; ch += 16) {
          const float* q = p.gn_part + (((size_t)n * p.gn_nchunk + ch) * G + g) * 2;
          s += q[0]; ss += q[1];
        }
      s += dpp_mov<0xB1>(s); ss += dpp_mov<0xB1>(ss);          // the 16 lanes of a group are one DPP row: quad steps, then the mirrors
      s += dpp_mov<0x4E>(s); ss += dpp_mov<0x4E>(ss);
      s += dpp_mov<0x141>(s); ss += dpp_mov<0x141>(ss);
      s += dpp_mov<0x140>(s); ss += dpp_mov<0x140>(ss);
      if (g < G && l == 0) {
        const float mean = s * inv_cnt;
        float var = ss * inv_cnt - mean * mean;
        var = var < 0.f ? 0.f : var;
        const float rstd = rsqrtf(var + p.gn_eps);
        gst[g * 2] = rstd;
        gst[g * 2 + 1] = -mean * rstd;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  if (nch > 0) issue(c0, 0);
  if (nch > 1) issue(c0 + 1, 1);

  if constexpr (GN) {
    const float* gsel = p.ngroups > 1 ? p.gn_gamma_g[grp] : p.gn_gamma;
    const float* bsel2 = p.ngroups > 1 ? p.gn_beta_g[grp] : p.gn_beta;
    const float* gst = (const float*)(smem + XS_STAGES * XS_STAGE);
    const int cpg = K / p.gn_groups;
    const unsigned inv = (65536u + (unsigned)cpg - 1u) / (unsigned)cpg;          // c / cpg == (c * inv) >> 16 for c < 2^14
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) {
      const int cb = kc * 32 + fq * 8;
      const f32x4 g0v = *(const f32x4*)(gsel + cb), g1v = *(const f32x4*)(gsel + cb + 4);
      const f32x4 b0v = *(const f32x4*)(bsel2 + cb), b1v = *(const f32x4*)(bsel2 + cb + 4);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int g = (int)(((unsigned)(cb + e) * inv) >> 16);
        const float gam = e < 4 ? g0v[e & 3] : g1v[e & 3], bet = e < 4 ? b0v[e & 3] : b1v[e & 3];
        const float a = gst[g * 2] * gam, b = __builtin_fmaf(gst[g * 2 + 1], gam, bet);
#pragma unroll
        for (int rf = 0; rf < 2; ++rf) xr[rf][kc][e] = from_f32<T>(__builtin_fmaf(to_f32(xr[rf][kc][e]), a, b));
      }
      asm volatile("" : "+v"(xr[0][kc]), "+v"(xr[1][kc]));
    }
  }

  if constexpr (LN) {
    // LayerNorm statistics of each row from the registers (a row's K values sit in the four lanes frow, frow+16,
    // frow+32, frow+48), fp32 sums, then the normalised values replace the raw ones.  One fragment at a time and an
    // opaque barrier on it in between: hipcc otherwise keeps fp32 copies of all 160 | 320 values alive across the two
    // passes (K = 640: 750 bytes of scratch per lane).
#pragma unroll
    for (int rf = 0; rf < 2; ++rf) {
      float s = 0.f, ss = 0.f;
#pragma unroll
      for (int kc = 0; kc < KC; ++kc) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float v = to_f32(xr[rf][kc][e]); s += v; ss = __builtin_fmaf(v, v, ss); }
        asm volatile("" : "+v"(xr[rf][kc]));
      }
      s = xor32_sum(xor16_sum(s));
      ss = xor32_sum(xor16_sum(ss));
      const float mean = s * (1.0f / (float)K);
      float var = ss * (1.0f / (float)K) - mean * mean;
      var = var < 0.f ? 0.f : var;
      const float rstd = rsqrtf(var + p.ln_eps);
      const float shift = -mean * rstd;
#pragma unroll
      for (int kc = 0; kc < KC; ++kc) {
#pragma unroll
        for (int e = 0; e < 8; ++e) xr[rf][kc][e] = from_f32<T>(__builtin_fmaf(to_f32(xr[rf][kc][e]), rstd, shift));
        asm volatile("" : "+v"(xr[rf][kc]));
      }
    }
  }

  char* ostage = smem + XS_STAGES * XS_STAGE + wave * (32 * XS_OROW);
  const auto rO = __builtin_amdgcn_make_buffer_rsrc(p.out, (short)0, (int)((size_t)p.M * p.ldo * 2), 0x00020000);
  // residual rows (es_xs_desc.residual: out = x W^T + bias + residual, the attention / proj output layers): the 16-byte chunks a lane
  // stores are requested at the TOP of the stage that produces them - four range-checked loads, issued unconditionally like the stores so
  // that the counted waits see the same queue in every wave - and consumed in its epilogue (the late waves': one stage later, two sets)
  const auto rR = __builtin_amdgcn_make_buffer_rsrc((void*)(RES ? p.residual : p.out), (short)0, (int)((size_t)p.M * p.ldo * 2), 0x00020000);
  auto load_res = [&](int ci, u32x4 (&rr)[4]) __attribute__((always_inline)) {
    if constexpr (RES) {
      const int line = c0 + ci;                      // P == 1: one 64-channel line per stage
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = i * 8 + (lane >> 3), col = lane & 7;
        const unsigned off = (unsigned)(((size_t)(r0 + row) * p.ldo + line * 64 + col * 8) * 2);
        rr[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rR, (int)off, 0, 0));
      }
    }
  };

  // The two waves that share a SIMD (w and w + 4) run the same program between the same barriers: left alone they do
  // their MFMAs together and their epilogue arithmetic together, and the matrix pipe idles through every epilogue
  // (measured: removing the MFMAs from this kernel saved exactly their stand-alone time).  Waves 4-7 therefore run
  // half a stage out of phase: they defer the epilogue of stage c to the start of stage c + 1 (accumulators and bias
  // kept in registers), so inside one barrier interval a SIMD sees [MFMA | epilogue] from one wave beside
  // [epilogue | MFMA] from the other.
  const bool late = wave >= 4;
  // ... and they are the younger half of the workgroup: at equal priority the SIMD's arbiter serves the older wave first
  // (MI355X_MICROARCH.md, two waves per SIMD, items 2 and 4), so a late wave's epilogue crawled behind its partner's
  // MFMAs AND epilogue (in-kernel stamps: 3700 cycles against 1300 for the same code in the older wave, which then sat
  // idle at the barrier for half the stage).  One static s_setprio for the younger half, no per-phase flips.
#ifndef XS_PRIO
#define XS_PRIO 1
#endif
#ifndef XS_RDMIX        // ping-pong form: fragment reads interleaved with the MFMA pairs (1) or in bursts of NF (0)
#define XS_RDMIX 1
#endif
#ifndef XS_WPF          // ping-pong form: K-chunks of weight fragments read ahead of their MFMAs
#define XS_WPF 2
#endif
#ifndef XS_PRIO_PP      // ping-pong form: 0 none, 1 static for the younger half, (XS_PRIO == 2: around the MFMA slot)
#define XS_PRIO_PP 0
#endif
  if (XS_PRIO == 1 && late && !PP) __builtin_amdgcn_s_setprio(1);
  if (PP && XS_PRIO_PP == 1 && late) __builtin_amdgcn_s_setprio(1);

  // top of iteration ci: stage ci landed?  VMEM operations of this wave younger than its DMAs of stage ci (issued at the
  // top of iteration ci-2): the next stage's NDMA, plus the 4 output stores of every finished line group in between.
  // Early waves store the line of stage c at the end of iteration c, late waves at the start of iteration c + 1; the
  // first iterations (no stores yet in the window) use the conservative count.
  // (with a residual: four more loads per stage in the window, already retired by the epilogue that consumed them)
  auto top = [&](int ci, u32x4 (&rr)[4]) __attribute__((always_inline)) {
    const int cio = ci - (late ? 1 : 0);
    if (XS_ABLATE & 8) {
    } else if (ci + 1 >= nch) {
      wait_vm<0>();
    } else if (ci < (late ? 3 : 2)) {
      wait_vm<NDMA>();
    } else if constexpr (P == 1) {
      wait_vm<NDMA + 8 + (RES ? 4 : 0)>();
    } else if constexpr (P == 2) {
      wait_vm<NDMA + 4>();
    } else {
      if ((cio & 3) < 2) wait_vm<NDMA + 4>(); else wait_vm<NDMA>();
    }
    if (!(XS_ABLATE & 8)) __builtin_amdgcn_s_barrier();   // everyone's pieces of stage ci landed; everyone is done with stage ci-1
    load_res(ci, rr);
    if (!(XS_ABLATE & 16) && ci + 2 < nch) issue(c0 + ci + 2, (ci + 2) % XS_STAGES);
  };

  // MFMAs of stage ci into acc (+ the stage's bias values, which live in the LDS stage that is recycled two barriers on)
  auto compute = [&](int ci, f32x4 (&acc)[NF][2], f32x4 (&bias)[NF]) __attribute__((always_inline)) {
    const char* sb = smem + (ci % XS_STAGES) * XS_STAGE;
    // weight fragments are read XS_WPF K-chunks (NF x 16 bytes per lane each) ahead of the MFMAs that consume them.  One chunk ahead covers
    // 8 MFMAs = 128 cycles, less than an LDS round trip while four waves read and the other four stage their outputs: in the ping-pong
    // form (one MFMA stream per SIMD, nothing to hide behind) stamps showed the MFMA slot paced by it - 80 MFMAs in 2000-2350 cycles
    constexpr int WPF = PP ? XS_WPF : 1;
    typename Traits<T>::vec8 wf[WPF + 1][NF];
    auto wread = [&](int kc, typename Traits<T>::vec8 (&dst)[NF]) __attribute__((always_inline)) {
      const int st = kc >> 1, h = kc & 1;
#pragma unroll
      for (int nf = 0; nf < NF; ++nf) {
        const int row = nf * 16 + frow;
        dst[nf] = as_vec8<T>(*(const u32x4*)(sb + st * SUB + row * 128 + (((4 * h + fq) ^ (row & 7)) << 4)));
      }
    };
#pragma unroll
    for (int a = 0; a < WPF && a < KC; ++a) wread(a, wf[a]);
    // the bias enters as the C operand of each accumulator's first MFMA: no add in the epilogue
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) bias[nf] = *(const f32x4*)(sb + XS_STAGE_W + (nf * 16 + fq * 4) * 4);
    // (C = the bias registers, D = the accumulator: no copy of the bias into two accumulators per fragment - 32 v_mov per stage)
#define XS_C(kc_, a_) ((kc_) == 0 ? bias[nf] : (a_))
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) {
      if constexpr (PP && XS_RDMIX) {
        // ping-pong form: ONE fragment read ahead of every MFMA pair instead of NF reads in a burst ahead of 2 NF MFMAs - a burst
        // of four ds_read_b128 from each of the four MFMA waves at once took 45-85 cycles to issue where an MFMA covers 16
        // (stamps: 80 MFMAs in 1650 cycles without the reads, 1950-2350 with them, whatever the prefetch depth)
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) {
          if (!(XS_ABLATE & 32) && kc + WPF < KC) {
            const int k2 = kc + WPF, st = k2 >> 1, h = k2 & 1, row = nf * 16 + frow;
            wf[k2 % (WPF + 1)][nf] = as_vec8<T>(*(const u32x4*)(sb + st * SUB + row * 128 + (((4 * h + fq) ^ (row & 7)) << 4)));
          }
          __builtin_amdgcn_sched_barrier(0);
          acc[nf][0] = mfma16(wf[(XS_ABLATE & 32) ? 0 : (kc % (WPF + 1))][nf], xr[0][kc], XS_C(kc, acc[nf][0]));
          acc[nf][1] = mfma16(wf[(XS_ABLATE & 32) ? 0 : (kc % (WPF + 1))][nf], xr[1][kc], XS_C(kc, acc[nf][1]));
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
      if (!(XS_ABLATE & 32) && kc + WPF < KC) wread(kc + WPF, wf[(kc + WPF) % (WPF + 1)]);
      __builtin_amdgcn_sched_barrier(0);          // (hipcc otherwise sinks every read to just before its first MFMA)
#pragma unroll
      for (int nf = 0; nf < NF; ++nf) {
#if XS_ABLATE & 1
        if (kc == 0) { acc[nf][0] = bias[nf]; acc[nf][1] = bias[nf]; }
        asm volatile("" ::"v"(wf[(XS_ABLATE & 32) ? 0 : (kc % (WPF + 1))][nf]), "v"(xr[0][kc]), "v"(xr[1][kc]));
#else
        acc[nf][0] = mfma16(wf[(XS_ABLATE & 32) ? 0 : (kc % (WPF + 1))][nf], xr[0][kc], XS_C(kc, acc[nf][0]));
        acc[nf][1] = mfma16(wf[(XS_ABLATE & 32) ? 0 : (kc % (WPF + 1))][nf], xr[1][kc], XS_C(kc, acc[nf][1]));
#endif
      }
      __builtin_amdgcn_sched_barrier(0);
      }
    }
  };

#undef XS_C
  // epilogue of stage ci: registers -> wave-private LDS tile -> (every P stages) full-line global stores
  auto epilogue = [&](int ci, const f32x4 (&acc)[NF][2], const f32x4 (&bias)[NF], const u32x4 (&rr)[4]) __attribute__((always_inline)) {
    const int sub = ci % P;                       // position of this stage inside its 128-byte output line
    if constexpr (RES) {
      // the residual chunks of this stage have landed?  Younger operations of this wave: an early wave has issued the DMAs of stage
      // ci + 2 since; a late wave (it runs this one stage later) those, the stores of stage ci - 1, the next four residual loads and
      // the DMAs of stage ci + 3.  Near the ends, where some of these were not issued, wait for everything.
      // (ping-pong form: both groups issue the DMAs of stage ci + 2 right before this epilogue - nothing else is younger)
      if (PP || !late) { if (ci + 2 < nch) wait_vm<NDMA>(); else wait_vm<0>(); }
      else { if (ci >= 1 && ci + 3 < nch) wait_vm<NDMA + 4 + 4 + NDMA>(); else wait_vm<0>(); }
    }
#pragma unroll
    for (int nf = 0; nf < NF; nf += (GEGLU ? 2 : 1)) {
#pragma unroll
      for (int rf = 0; rf < 2; ++rf) {
        typename Traits<T>::vec4 pk;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if constexpr (GEGLU) {
            const float gate = acc[nf + 1][rf][r];
            pk[r] = from_f32<T>(acc[nf][rf][r] * ((XS_ABLATE & 2) ? gate : gelu_f(gate)));
          } else {
            pk[r] = from_f32<T>(acc[nf][rf][r]);
          }
        }
        const int chan = (GEGLU ? nf / 2 : nf) * 16 + fq * 4;
        *(typename Traits<T>::vec4*)(ostage + (rf * 16 + frow) * XS_OROW + sub * OUTB + chan * 2) = pk;
      }
    }
    if (sub == P - 1) {
      // a full 128-byte line per row is staged: 4 coalesced 16-byte-per-lane stores (8 rows each)
      const int line = (c0 + ci) / P;             // 64-channel group of the output row
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = i * 8 + (lane >> 3), col = lane & 7;
        u32x4 v = *(const u32x4*)(ostage + row * XS_OROW + col * 16);
        if constexpr (RES) {
          auto a = as_vec8<T>(v);
          const auto b = as_vec8<T>(rr[i]);
#pragma unroll
          for (int e = 0; e < 8; ++e) a[e] = from_f32<T>(to_f32(a[e]) + to_f32(b[e]));
          v = __builtin_bit_cast(u32x4, a);
        }
        // range-checked buffer store (rows >= M fall beyond num_records and are dropped by the hardware): the store
        // is issued unconditionally, so every wave's VMEM count per iteration is the same - the counted waits above
        // depend on it.  aux 16 = sc1 (write-through, like the GEMM's output stores)
        const unsigned off = (unsigned)(((size_t)(r0 + row) * p.ldo + line * 64 + col * 8) * 2);
        if (!(XS_ABLATE & 4)) __builtin_amdgcn_raw_buffer_store_b128(v, rO, (int)off, 0, ES_WT_STORES ? 16 : 0);
        else asm volatile("" ::"v"(v), "v"(off));
      }
    }
  };

  f32x4 accA[NF][2], accB[NF][2], bA[NF], bB[NF];
  u32x4 rrA[4], rrB[4];
  if constexpr (PP) {
    // Two-barrier ping-pong (round 5).  In-kernel stamps of the one-barrier form on the plain projections (tools/xs_stamps.py, 57344 x 320
    // -> 960): the "half a stage out of phase" of waves 4-7 is only as long as an epilogue, and a plain epilogue is short - both waves of a
    // SIMD then run their 80 MFMAs at the same time (the early wave's MFMA phase stretched from 1300 to 3600 cycles), and the ~1500 cycles of
    // barrier + six DMA issues + fragment-read latency + epilogue at the stage boundary see no MFMA at all (29 % of the stage).  Here the two
    // groups alternate by construction, as in gemm_conv8p.hip: a stage is TWO slots between workgroup barriers; in a slot one group only
    // issues MFMAs (stage t) while the other does everything else (DMAs of stage t + 2, epilogue of stage t, output stores); group 1 runs
    // one slot behind group 0.  Ring of three stages as before; every wave issues the DMAs of stage t + 2 right after its MFMAs of stage t
    // (the buffer's last reader, group 1's MFMAs of stage t - 1, finished two barriers earlier) and retires its pieces of stage t + 1 by a
    // COUNTED wait before the barrier that opens group 0's MFMA slot of that stage: group 0 at the end of its own "other" slot, group 1 at
    // the end of its MFMA slot.  One accumulator set per wave (the epilogue of a stage ends before the next MFMAs start).
    constexpr int ST = 4;                          // output stores per finished 128-byte line
    constexpr int RS = RES ? 4 : 0;                // residual loads per stage
    auto line_done = [&](int t) { return t >= 0 && (t % P) == P - 1; };
    // every wave: its pieces of stage 0 have landed (stage 1's stay in flight)
    if (nch > 1) wait_vm<NDMA>(); else wait_vm<0>();
    if (late) __builtin_amdgcn_s_barrier();
    for (int t = 0; t < nch; ++t) {
      __builtin_amdgcn_s_barrier();                // stage t complete and visible (group 0: barrier 2t, group 1: barrier 2t + 1)
      load_res(t, rrA);                            // (the residual chunks arrive behind the MFMAs)
#if XS_STAMPS
      stamp(t, 0);
#endif
      if (XS_PRIO == 2) __builtin_amdgcn_s_setprio(1);
      compute(t, accA, bA);
      if (XS_PRIO == 2) __builtin_amdgcn_s_setprio(0);
#if XS_STAMPS
      stamp(t, 1);
#endif
      if (late && t + 1 < nch) {
        // group 1: its pieces of stage t + 1 (issued one slot ago, ahead of the stores of stage t - 1 and this stage's residual loads)
        if (t + 2 >= nch && t == 0) wait_vm<RS>();
        else if (line_done(t - 1)) wait_vm<ST + RS>(); else wait_vm<RS>();
      }
      if (!late || t + 1 < nch) __builtin_amdgcn_s_barrier();
      if (!(XS_ABLATE & 16) && t + 2 < nch) issue(c0 + t + 2, (t + 2) % XS_STAGES);
      epilogue(t, accA, bA, rrA);
      if (!late && t + 1 < nch) {
        // group 0: its pieces of stage t + 1 (younger: the DMAs of stage t + 2 just issued, this stage's stores)
        if (t + 2 < nch) { if (line_done(t)) wait_vm<NDMA + ST>(); else wait_vm<NDMA>(); }
        else { if (line_done(t)) wait_vm<ST>(); else wait_vm<0>(); }
      }
#if XS_STAMPS
      stamp(t, 2);
#endif
    }
  } else
  if (!late) {
    for (int ci = 0; ci < nch; ++ci) {
      top(ci, rrA);
#if XS_STAMPS
      stamp(ci, 0);
#endif
      compute(ci, accA, bA);
#if XS_STAMPS
      stamp(ci, 1);
#endif
      epilogue(ci, accA, bA, rrA);
#if XS_STAMPS
      stamp(ci, 2);
#endif
    }
  } else {
    // unrolled by two so that the two accumulator sets keep static names
    int ci = 0;
    for (; ci + 1 < nch; ci += 2) {
      top(ci, rrA);
#if XS_STAMPS
      stamp(ci, 0);
#endif
      if (ci > 0) epilogue(ci - 1, accB, bB, rrB);
#if XS_STAMPS
      stamp(ci, 1);
#endif
      compute(ci, accA, bA);
#if XS_STAMPS
      stamp(ci, 2);
#endif
      top(ci + 1, rrB);
#if XS_STAMPS
      stamp(ci + 1, 0);
#endif
      epilogue(ci, accA, bA, rrA);
#if XS_STAMPS
      stamp(ci + 1, 1);
#endif
      compute(ci + 1, accB, bB);
#if XS_STAMPS
      stamp(ci + 1, 2);
#endif
    }
    if (ci < nch) {
      top(ci, rrA);
      if (ci > 0) epilogue(ci - 1, accB, bB, rrB);
      compute(ci, accA, bA);
      epilogue(ci, accA, bA, rrA);
    } else if (nch > 0) {
      epilogue(nch - 1, accB, bB, rrB);
    }
  }
#if !XS_STAMPS
  if (p.prof && tid == 0) atomicMax(p.prof + 1, (unsigned long long)__builtin_amdgcn_s_memrealtime());
#endif
}

template <typename T, int KC, bool GEGLU, bool LN, bool RES = false, bool PP = false, bool GN = false>
int launch_one(const es_xs_desc& d, hipStream_t st) {
  const int rbs = (d.M + XS_ROWS - 1) / XS_ROWS;
  auto kfn = linear_xs_kernel<T, KC, GEGLU, LN, RES, PP, GN>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, XS_LDS);
    attr_set = true;
  }
  hipLaunchKernelGGL(kfn, dim3(rbs * d.nslices), dim3(512), XS_LDS, st, d);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

// form of the plain (no GEGLU) launches: 1 = two-barrier ping-pong (default), 0 = the one-barrier form (ES_XS_PP=0; es_linear_xs_set_pp for
// A/B runs inside one process).  Same arithmetic in the same order either way: bit-identical outputs.
int xs_pp = [] { const char* e = getenv("ES_XS_PP"); return e ? atoi(e) : 1; }();

template <typename T>
int launch(const es_xs_desc& d, hipStream_t st) {
  const bool g = d.geglu != 0, ln = d.ln != 0;
  if (d.gn_part)       // GroupNorm in front (checked by es_linear_xs: plain projection); the ping-pong form whatever es_linear_xs_set_pp says
    return d.K == 320 ? launch_one<T, 10, false, false, false, true, true>(d, st) : launch_one<T, 20, false, false, false, true, true>(d, st);
  if (xs_pp && !g) {
    if (d.residual) return launch_one<T, 10, false, false, true, true>(d, st);
    if (d.K == 320) return ln ? launch_one<T, 10, false, true, false, true>(d, st) : launch_one<T, 10, false, false, false, true>(d, st);
    return ln ? launch_one<T, 20, false, true, false, true>(d, st) : launch_one<T, 20, false, false, false, true>(d, st);
  }
  if (d.residual) return launch_one<T, 10, false, false, true>(d, st);     // (K = 320, no GEGLU, no LayerNorm: checked by es_linear_xs)
  if (d.K == 320) {
    if (g) return ln ? launch_one<T, 10, true, true>(d, st) : launch_one<T, 10, true, false>(d, st);
    return ln ? launch_one<T, 10, false, true>(d, st) : launch_one<T, 10, false, false>(d, st);
  }
  if (g) return ln ? launch_one<T, 20, true, true>(d, st) : launch_one<T, 20, true, false>(d, st);
  return ln ? launch_one<T, 20, false, true>(d, st) : launch_one<T, 20, false, false>(d, st);
}

}  // namespace

extern "C" void es_set_error(const char* msg);

extern unsigned long long es_operand_limit_v;          // gemm_conv.hip: 0x7FFFFFFF unless a test lowered it

extern "C" int es_linear_xs_set_pp(int on) { const int prev = xs_pp; xs_pp = on; return prev; }

extern "C" int es_linear_xs(const es_xs_desc* d, void* stream) {
  if (!d->x || !d->out || (d->ngroups <= 1 && (!d->w || !d->bias))) { es_set_error("es_linear_xs: null pointer (bias is required: pass zeros)"); return -1; }
  if (d->K != 320 && d->K != 640) { es_set_error("es_linear_xs: K must be 320 or 640"); return -1; }
  const int CH = d->K == 320 ? 64 : 32;
  const int P = 128 / (d->geglu ? CH : CH * 2);
  if (d->M < 1 || d->Cout < CH || d->Cout % (CH * P) || d->rows_padded < d->Cout) {
    es_set_error("es_linear_xs: Cout must be a multiple of the 128-byte output line (64 stored channels)"); return -1; }
  const int total = d->Cout / CH;
  if (d->nslices < 1 || d->chunks_per_slice < P || d->chunks_per_slice % P ||
      (long long)d->nslices * d->chunks_per_slice < total || (long long)(d->nslices - 1) * d->chunks_per_slice >= total) {
    es_set_error("es_linear_xs: slices must cover Cout in whole output lines, none empty"); return -1; }
  const int cstore = d->geglu ? d->Cout / 2 : d->Cout;
  if (d->ldo < cstore || d->ldo % 8) { es_set_error("es_linear_xs: bad output pitch"); return -1; }
  if ((size_t)d->rows_padded * d->K * 2 >= 0x7FFFFFFFull) { es_set_error("es_linear_xs: weights larger than 2 GiB (32-bit buffer offsets)"); return -1; }
  // activations / outputs beyond the kernel's 32-bit buffer offsets: runs of whole 256-row blocks, one launch each (below)
  const bool chunked = (size_t)d->M * d->K * 2 >= es_operand_limit_v || ((size_t)d->M + 256) * d->ldo * 2 >= es_operand_limit_v;
  if (d->ngroups > 4) { es_set_error("es_linear_xs: at most 4 groups"); return -1; }
  if (d->gn_part) {
    if (d->geglu || d->ln || d->residual) { es_set_error("es_linear_xs: GroupNorm in front (gn_part) needs the plain projection: no GEGLU, no LayerNorm fold, no residual"); return -1; }
    if (d->gn_groups < 1 || d->gn_groups > 32 || d->K % d->gn_groups || d->gn_hw < 256 || d->gn_hw % 256 || d->M % d->gn_hw || d->gn_nchunk < 1 || d->gn_nchunk > 64) {
      es_set_error("es_linear_xs: gn_part needs 1..32 groups dividing K, samples of a whole number of 256-row blocks covering M, 1..64 chunks"); return -1; }
    if (d->ngroups <= 1 && (!d->gn_gamma || !d->gn_beta)) { es_set_error("es_linear_xs: gn_part without gn_gamma / gn_beta"); return -1; }
    for (int g = 0; g < d->ngroups && d->ngroups > 1; ++g)
      if (!d->gn_gamma_g[g] || !d->gn_beta_g[g]) { es_set_error("es_linear_xs: gn_part without gn_gamma_g / gn_beta_g for every group"); return -1; }
  }
  if (d->residual && (d->K != 320 || d->geglu || d->ln)) { es_set_error("es_linear_xs: a residual needs K = 320, no GEGLU, no LayerNorm fold (rows of ldo elements, like out)"); return -1; }
  es_xs_desc dd = *d;
  if (d->ngroups > 1) {
    const int tm = (d->M + 127) / 128;
    for (int g = 0; g < d->ngroups; ++g)
      if (!d->w_g[g] || !d->bias_g[g] || d->mt_end[g] <= (g ? d->mt_end[g - 1] : 0) || (d->mt_end[g] & 1)) {
        es_set_error("es_linear_xs: groups must be non-empty runs of whole 256-row blocks"); return -1; }
    if (d->mt_end[d->ngroups - 1] != tm) { es_set_error("es_linear_xs: groups must cover M"); return -1; }
  }
  for (int g = dd.ngroups > 1 ? dd.ngroups : 0; g < 4; ++g) dd.mt_end[g] = 0x7FFFFFFF;
  ES_PLAN_RECORD(ES_OP_LINEAR_XS, d, sizeof(*d));
  int rc = 0;
  if (chunked) {
    // (like es_conv_gemm's launch_in_chunks: the row blocks of a launch never interact, so the cuts change nothing in the results)
    const size_t wide = (size_t)(d->K > d->ldo ? d->K : d->ldo) * 2;
    long long rows = es_operand_limit_v > 512 * wide ? (long long)((es_operand_limit_v - 256 * wide - 1) / wide) : 256;
    rows -= rows % 256;
    if (rows < 256) rows = 256;
    if (d->gn_part) {                      // whole samples per cut: the statistics are per sample
      if (rows < d->gn_hw) { es_set_error("es_linear_xs: one sample exceeds 2 GiB (32-bit buffer offsets)"); return -1; }
      rows -= rows % d->gn_hw;
    }
    const int ng = d->ngroups > 1 ? d->ngroups : 1;
    long long r0 = 0;
    for (int g = 0; g < ng && !rc; ++g) {
      const long long r1 = d->ngroups > 1 && g + 1 < ng ? (long long)d->mt_end[g] * 128 : d->M;
      for (long long a = r0; a < r1 && !rc; a += rows) {
        es_xs_desc s = *d;
        s.M = (int)(r1 - a < rows ? r1 - a : rows);
        if (d->ngroups > 1) { s.w = d->w_g[g]; s.bias = d->bias_g[g]; }
        s.ngroups = 0;
        for (int k = 0; k < 4; ++k) s.mt_end[k] = 0x7FFFFFFF;
        s.x = (const char*)d->x + (size_t)a * d->K * 2;
        s.out = (char*)d->out + (size_t)a * d->ldo * 2;
        if (d->residual) s.residual = (const char*)d->residual + (size_t)a * d->ldo * 2;
        if (d->gn_part) {
          s.gn_part = d->gn_part + (size_t)(a / d->gn_hw) * d->gn_nchunk * d->gn_groups * 2;
          if (d->ngroups > 1) { s.gn_gamma = d->gn_gamma_g[g]; s.gn_beta = d->gn_beta_g[g]; }
        }
        rc = s.dtype == ES_F16 ? launch<f16>(s, (hipStream_t)stream) : launch<bf16>(s, (hipStream_t)stream);
      }
      r0 = r1;
    }
  } else {
    rc = dd.dtype == ES_F16 ? launch<f16>(dd, (hipStream_t)stream) : launch<bf16>(dd, (hipStream_t)stream);
  }
  if (rc) es_set_error("es_linear_xs: launch failed");
  return rc;
}
