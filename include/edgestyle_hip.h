/*
 * edgestyle_hip.h — C ABI of libedgestyle_hip.so: the MI355X (gfx950) kernels behind EdgeStyle's
 * multi-ControlNet SD1.5 denoising hot path.
 *
 * The reference (andrei-ace/EdgeStyle) has no FFI: every FLOP of this path is a PyTorch/cuDNN op launched by
 * diffusers modules.  Each entry point below names the reference/diffusers op it replaces so a
 * maintainer can bind it (ctypes stub shown in INTEGRATION.md).  Conventions:
 *   - every pointer is a DEVICE pointer owned by the caller (a torch tensor's data_ptr()); it must outlive the
 *     stream operation; nothing is allocated, freed or synchronised inside (hipGraph-capture safe);
 *   - activations are NHWC ([N,H,W,C], C contiguous) in `dtype` (ES_F16 / ES_BF16); accumulation is fp32;
 *   - `stream` is a hipStream_t passed as void*; calls are asynchronous w.r.t. it;
 *   - return 0 on success, <0 on error (es_last_error() gives the text); never throws across the boundary.
 */
#ifndef EDGESTYLE_HIP_H
#define EDGESTYLE_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { ES_F16 = 0, ES_BF16 = 1, ES_F32 = 2 /* es_tensor sources only: the kernels compute in ES_F16 / ES_BF16 */ };
enum { ES_ACT_NONE = 0, ES_ACT_SILU = 1, ES_ACT_GEGLU = 2 };

/* 7 (round 5): es_gemm_desc.bn = 256; es_xs_desc.gn_* and es_gn_desc.stats_only (both structs grew: GroupNorm in front of a row-stationary
 * projection); es_conv_gemm8p_form_ok, es_ctx_graph_hazard, es_linear_xs_set_pp, es_attention_set_kvres, es_set_operand_limit,
 * es_group_norm_chunks, es_clock_probe.  Context images carry the version and are rebuilt across it. */
#define ES_ABI_VERSION 7
int es_abi_version(void);
/* sizeof the descriptor structs as compiled (0 gemm, 1 attn, 2 gn, 3 fusion, 4 ln, 5 xs): lets a binding verify its mirror */
size_t es_sizeof_desc(int which);
const char* es_last_error(void);

/* ---------------------------------------------------------------------------------------------------------
 * es_conv_gemm — implicit-GEMM convolution / linear on MFMA (16x16x32 f16|bf16, fp32 accumulate).
 * Replaces: torch conv2d 3x3 s1/s2 + 1x1 and nn.Linear as invoked by diffusers ResnetBlock2D.conv1/conv2,
 * conv_shortcut, Downsample2D, Upsample2D (nearest-2x fused into the loader), Transformer2DModel.proj_in/out,
 * Attention.to_q/k/v/to_out, GEGLU/FeedForward, TimestepEmbedding, ControlNet zero-convs
 * (model/controllora.py:197-254) and the skip-concat `torch.cat([h, res], 1)` of the UNet up blocks (x2).
 *   out[m, co] = act( sum_k A[m,k] * W[co,k] + bias[co] + temb[n(m), co] ) * out_scale (+ residual[m, co])
 * with m = (n, oy, ox), k = (ky, kx, c) tap-major over the channel-concatenated sources (x | x2).
 * W is pre-packed (ops.pack_weight / es_load_weights): [rows_padded][Kpad] K-contiguous, dtype; K order: `korder`.
 * --------------------------------------------------------------------------------------------------------- */
typedef struct {
  const void* x;          /* [N, Hsrc, Wsrc, C1] */
  const void* x2;         /* optional second source [N, Hsrc, Wsrc, C2] (channel concat), or NULL */
  const void* w;          /* packed weights [rows_padded][Kpad] */
  const float* bias;      /* [rows_padded] fp32 or NULL */
  const void* temb;       /* optional per-sample per-channel add, dtype, [N, temb_stride] (already offset) */
  const void* residual;   /* optional [M, Cout_store] dtype, added after act*scale */
  const float* out_scale_dev; /* optional device scalar multiplied into the result (conditioning_scale) */
  void* out;              /* [N, Hout, Wout, Cout_store] dtype */
  float* workspace;       /* split-K partials [splitk][M][rows_padded] fp32 (NULL if splitk == 1) */
  unsigned long long* prof; /* optional device u64[2]: every workgroup atomically mins its start / maxes its end
                               s_memrealtime stamp (100 MHz) -> launch duration as executed inside a hipGraph */
  int32_t N, Hsrc, Wsrc, C1, C2;
  int32_t Hout, Wout, Cout;   /* Cout = true GEMM N (before GEGLU halving) */
  int32_t rows_padded, Kpad;  /* packed weight geometry */
  int32_t ksize, stride, pad; /* ksize 1|3 ; input coord = o*stride + k - pad */
  int32_t upsample;           /* 1: sources are nearest-2x upsampled on the fly (Hin = 2*Hsrc) */
  int32_t temb_stride;
  int32_t act;                /* ES_ACT_* ; GEGLU halves the stored width (weights packed interleaved) */
  int32_t splitk;
  int32_t bn;                 /* N tile: 128 | 160 (128-pixel tile), 64 (64 x 64 tile, tiny launches), 320 (256 x 320 phase-interleaved
                               * tile, csrc/gemm_conv8p.hip: large launches with 64-aligned channels), 256 (its 256 x 256 form: the
                               * LayerNorm-folded / GEGLU linear layers, ln_colsum / ES_ACT_GEGLU allowed, no temb); rows_padded % bn == 0 */
  int32_t dtype;
  /* Grouped launch: ngroups (<= 4) problems of identical geometry but different weights run as ONE launch over the
   * batch-concatenated activations (the 3 batched ControlNet passes + the UNet encoder of a denoising step): M tiles
   * [mt_end[g-1], mt_end[g]) (units of 128 pixels) use w_g[g] / bias_g[g].  ngroups <= 1: plain launch (w, bias). */
  int32_t ngroups;
  int32_t mt_end[4];
  const void* w_g[4];
  const float* bias_g[4];
  int32_t stages;             /* LDS ring depth: 0 = auto, 2 (2 workgroups/CU), 3 or 4 (1 workgroup/CU) */
  int32_t xcd_m_fastest;      /* tile order inside an XCD's chunk: 1 = tile_m fastest (weights are the larger operand) */
  int32_t waves;              /* waves per 128-pixel workgroup: 0 / 4 = default, 8 = two per SIMD (launches that leave
                               * a workgroup alone on its CU: one wave per SIMD cannot overlap DMA issue, LDS reads and
                               * MFMAs with itself) */
  float out_scale;
  /* LayerNorm folded into a linear layer (the three LayerNorms of a BasicTransformerBlock feed exactly one Linear
   * each): x is the RAW residual stream, w holds W * gamma (per input channel), bias holds W beta + b, and
   * ln_colsum[j] = sum_k w[j][k] (fp32, of the rounded packed weights).  The kernel accumulates every row's sum and
   * sum of squares from the activation tiles it stages anyway and applies
   *   out[m][j] = rstd_m * (acc[m][j] - mean_m * ln_colsum[j]) + bias[j]
   * in the epilogue.  Needs ksize 1, one source, K == C1 (a multiple of 64), splitk 1, bn 64|128|160. */
  const float* ln_colsum;     /* NULL: plain launch */
  const float* ln_colsum_g[4];
  float ln_eps;
  /* 1x1 tail sources (ResnetBlock2D: conv2(h) + conv_shortcut(x) as ONE launch): after the ksize*ksize taps over
   * (x | x2) the K axis continues with ONE centre tap over the channel concat (t1 | t2), tensors of the OUTPUT's
   * spatial size [N, Hout, Wout, Ct1|Ct2]; the packed weights are [W_conv | W_shortcut] along K, the bias is the sum.
   * Needs stride 1, no upsample, 64-aligned C1, C2, Ct1, Ct2.  t1 == NULL: no tail. */
  const void* t1;
  const void* t2;
  int32_t Ct1, Ct2;
  /* source batch modulo: sample n of the launch reads source sample n % x_nmod of x (x then holds x_nmod samples): the
   * same tensor feeds several groups of a grouped launch (conv_in of all ControlNets + the UNet on ONE sample tensor,
   * CL:197-203; text states shared by the nets of a weight-sharing group) without a replicated copy.  0 = off.
   * Needs one source (no x2), no tail sources. */
  int32_t x_nmod;
  /* K order of the packed weights (and of the loader): 0 = tap-major, k = (ky, kx, c) over the channel concat;
   * 1 = chunk-major, k = (c / 64, ky, kx, c % 64): the nine taps of one 64-channel chunk back to back, so that eight of
   * the nine activation tiles a workgroup stages per chunk re-read lines it fetched one K-tile earlier (L2 hits instead
   * of Infinity-Cache round trips: with tap-major order the 32 workgroups of an XCD sweep 5-10 MB of activations between
   * two reads of the same line).  The 1x1 tail sources follow the k*k*Ctot main part in either order.
   * 1 needs ksize 3 and 64-aligned C1, C2; ops.pack_weight / es_load_weights pack every such convolution this way. */
  int32_t korder;
  /* WIDE RESIDUAL STREAM (bf16 pipelines, BASELINE configs[4]): the tensors every ResnetBlock / transformer block adds into
   * travel as TWO tensors of the compute dtype, hi + lo with hi = round(x), lo = round(x - hi) - 16 significant bits in bf16.
   * MFMA operands, GroupNorm and everything else read `hi` (the ordinary tensor); only the residual add sees both:
   *   sum = round(act(...) * scale) + residual + residual_lo   (fp32);   out = round(sum);   out_lo = round(sum - out)
   * so the rounding of the stream's sums no longer accumulates from block to block (it was 5 dB of the bf16 error budget:
   * profiles/r04_bf16_error_budget_2steps.txt).  out_lo needs `residual` and an output width that is a multiple of 8;
   * residual_lo needs out_lo.  Both NULL: the single-tensor stream of every fp16 pipeline. */
  const void* residual_lo;
  void* out_lo;
  /* GroupNorm statistics handed over by the producer: the epilogue (fused or split-K reduce) also writes, for every 64-pixel block
   * of its output and every GroupNorm group, the block's sum and sum of squares - fp32 [N][2 * HW/64][gn_groups][2]: entry
   * 2 * block + 0 holds the part of a group inside the N tile the group STARTS in, entry 2 * block + 1 the part inside the next
   * N tile (zero when the group ends in its first tile) - summed in ONE fixed order whatever tile, wave count or split-K the launch
   * uses (64 pixels as 8 sequential sub-blocks of 8, then the group's channels in order).  es_group_norm with
   * es_gn_desc.ext_chunks = 2 * HW/64 then normalises in a single pass: no statistics launch, no second read.  Needs the
   * output's H*W % 64 == 0, Cout % 8 == 0 and Cout / gn_groups <= the N tile.  NULL: no statistics. */
  float* gn_part;
  int32_t gn_groups;
} es_gemm_desc;
int es_conv_gemm(const es_gemm_desc* d, void* stream);
/* The kernels address each activation operand (sources, tail sources, output) through 32-bit buffer offsets: 2 GiB.  A launch whose
 * operands are larger is run by es_conv_gemm / es_linear_xs as several launches over runs of whole samples (per weight group, then as
 * many samples as fit), each with its operand pointers moved to its first sample - the tiles of a launch never interact, so the results
 * do not depend on the cuts.  Test knob: lower the limit so that small launches are cut (0 restores 2 GiB); returns the previous value. */
unsigned long long es_set_operand_limit(unsigned long long bytes);
size_t es_conv_gemm_workspace_bytes(const es_gemm_desc* d);

/* ---------------------------------------------------------------------------------------------------------
 * es_linear_xs — row-stationary short-K linear layer (K = 320 | 640): out = epilogue(LayerNorm?(x) W^T + bias).
 * Replaces, at the 64x64 and 32x32 UNet / ControlNet levels: BasicTransformerBlock.norm1 -> attn1.to_q|k|v and
 * norm3 -> ff.net.0 (GEGLU) (diffusers, called under model/controllora.py:205-238) - the same math as es_conv_gemm with
 * ln_colsum / ES_ACT_GEGLU, organised for few K-steps and many output columns (csrc/linear_xs.hip).
 * x [M, K] dtype, raw residual stream when ln != 0 (gamma / beta are folded into w / bias: ops.pack_weight_ln);
 * w packed [rows_padded][K] (GEGLU: rows interleaved 16 hidden | 16 gate, as for es_conv_gemm); bias fp32
 * [rows_padded] (required); out [M, ldo] dtype, Cout (or Cout / 2 for GEGLU) columns written.
 * Workgroup = 256 rows x one slice of the output columns: grid = ceil(M / 256) * nslices, slice s covers the
 * `chunks_per_slice` column chunks (64 columns at K = 320, 32 at K = 640) starting at s * chunks_per_slice.
 * Grouped launch like es_conv_gemm: rows [mt_end[g-1], mt_end[g]) * 128 use w_g[g] / bias_g[g] (even mt_end only).
 * --------------------------------------------------------------------------------------------------------- */
typedef struct {
  const void* x; void* out;
  const void* w; const float* bias;
  const void* w_g[4]; const float* bias_g[4];
  unsigned long long* prof;   /* as es_gemm_desc.prof */
  int32_t mt_end[4];
  int32_t ngroups;
  int32_t M, K, Cout, rows_padded;
  int32_t ldo;                /* output row pitch in elements */
  int32_t geglu, ln;
  float ln_eps;
  int32_t nslices, chunks_per_slice;
  int32_t dtype;
  /* optional: out = x W^T + bias + residual - rows of ldo elements, like out (Attention.to_out + residual, BasicTransformerBlock).
   * K = 320 only, no GEGLU, no LayerNorm fold */
  const void* residual;
  /* optional: GroupNorm in front (Transformer2DModel.norm -> proj_in: eps 1e-6, no activation).  x is then the RAW tensor [N, gn_hw, K];
   * gn_part holds the sums of x and x^2 per (sample, pixel chunk, GroupNorm group) - fp32 [N][gn_nchunk][gn_groups][2], as
   * es_group_norm(stats_only = 1) writes them (gn_nchunk = es_group_norm_chunks(gn_hw)) - and gn_gamma / gn_beta (gn_gamma_g / gn_beta_g
   * per weight group of a grouped launch) the affine parameters, fp32 [K].  The kernel normalises the rows it holds in registers:
   * one read of x instead of three passes over it.  Needs the plain projection (no GEGLU, LayerNorm fold or residual), gn_hw % 256 == 0,
   * M == N * gn_hw, at most 32 groups.  gn_part == NULL: off. */
  const float* gn_part;
  const float* gn_gamma; const float* gn_beta;
  const float* gn_gamma_g[4]; const float* gn_beta_g[4];
  int32_t gn_groups, gn_nchunk, gn_hw;
  float gn_eps;
} es_xs_desc;
int es_linear_xs(const es_xs_desc* d, void* stream);
/* tool / test knob: the form of the plain (no GEGLU) launches - 1 = two-barrier ping-pong between the wave groups (default; ES_XS_PP=0
 * in the environment turns it off), 0 = the one-barrier form.  Bit-identical outputs.  Returns the previous setting. */
int es_linear_xs_set_pp(int on);

/* Fused attention softmax(Q K^T * scale) V (flash-style, online softmax, MFMA).
 * Replaces torch.nn.functional.scaled_dot_product_attention under diffusers Attention (attn1/attn2/VAE attn).
 * Q [N,Sq,ldq], K [N,Skv,ldk], V [N,Skv,ldv], O [N,Sq,ldo]; head h lives at column h*d of every row. */
typedef struct {
  const void* q; const void* k; const void* v; void* o;
  int32_t N, heads, Sq, Skv, d;
  int32_t ldq, ldk, ldv, ldo;         /* row strides in elements */
  int64_t bsq, bsk, bsv, bso;         /* batch strides in elements */
  float scale;
  int32_t dtype;
} es_attn_desc;
int es_attention(const es_attn_desc* d, void* stream);
/* tool / test knob: launches with Skv <= 96 and head_dim 40 | 80 (the text-token cross-attention of the two shallow UNet levels) on the
 * K/V-resident kernel - 0 never, 1 (default; ES_ATTN_KVRES in the environment) where it wins (>= 12 samples at head_dim 40, >= 64 at 80:
 * profiles/r05_xattn_bench.txt), 2 every eligible launch.  Returns the previous setting. */
int es_attention_set_kvres(int on);

/* GroupNorm (+SiLU) over NHWC with optional channel-concat of two sources.
 * Replaces torch group_norm + silu of ResnetBlock2D.norm1/norm2, conv_norm_out, Transformer2DModel.norm.
 * partials: fp32 scratch [N][nchunk<=32][groups][2]. */
typedef struct {
  const void* x; const void* x2; void* out;
  const float* gamma; const float* beta; float* partials;
  int32_t N, HW, C1, C2, groups;
  float eps;
  int32_t silu, dtype;
  /* grouped launch: samples [n_end[g-1], n_end[g]) use gamma_g[g] / beta_g[g] (ngroups <= 1: gamma, beta) */
  int32_t ngroups;
  int32_t n_end[4];
  const float* gamma_g[4];
  const float* beta_g[4];
  /* > 0: `partials` already HOLDS the statistics of x as ext_chunks partial sums per (sample, group) - written by the producing
   * es_conv_gemm launch (es_gemm_desc.gn_part, ext_chunks = 2 * HW/64): no statistics pass, one read of x.  One source only. */
  int32_t ext_chunks;
  /* 1: the statistics pass alone - `partials` receives the sums of x and x^2 per (sample, pixel chunk, group), fp32
   * [N][es_group_norm_chunks(HW)][groups][2]; nothing is normalised, out / gamma / beta are not read.  The consumer applies the
   * normalisation itself (es_xs_desc.gn_part: Transformer2DModel.norm -> proj_in as one read of x). */
  int32_t stats_only;
} es_gn_desc;
int es_group_norm(const es_gn_desc* d, void* stream);
size_t es_group_norm_partials_bytes(int N, int groups);
int es_group_norm_chunks(int HW);                       /* pixel chunks per sample of the statistics pass */
int es_group_norm_is_slab(int HW, int C, int groups);   /* 1: this geometry runs as ONE launch (slab form) */

/* LayerNorm over the last dim of [M, C] (BasicTransformerBlock.norm1/2/3), eps 1e-5. */
int es_layer_norm(const void* x, void* out, const float* gamma, const float* beta, int M, int C, float eps,
                  int dtype, void* stream);
/* grouped LayerNorm: rows [row_end[g-1], row_end[g]) use gamma_g[g] / beta_g[g] */
typedef struct {
  const void* x; void* out;
  const float* gamma_g[4]; const float* beta_g[4];
  int32_t row_end[4];
  int32_t ngroups, M, C;
  float eps;
  int32_t dtype;
} es_ln_desc;
int es_layer_norm_grouped(const es_ln_desc* d, void* stream);

/* EdgeStyle fusion block: interleave (model/edgestyle_multicontrolnet.py:479-501) + ControlNetBlock
 * (model/edgestyle_multicontrolnet.py:23-63) without materialising the interleaved tensor.
 * res[i]: residual of net i, [N, HW, C] dtype (batch stride res_bs[i] elements so nets batched together can be
 * addressed in place).  Params are repacked pixel-major: w1[C][3][2], b1[C][3], g1/be1[HW][C][3] (dtype),
 * w2[C][3], b2[C], g2/be2[HW][C] (dtype), w3[C], b3[C] (fp32 unless noted).
 * scratch: fp32 [N][2][nchunk<=256][2] partial sums; u: dtype [N,HW,C] intermediate; out: [N,HW,C] dtype. */
typedef struct {
  const void* res[6];
  int64_t res_bs[6];
  const float* w1; const float* b1; const void* g1; const void* be1;
  const float* w2; const float* b2; const void* g2; const void* be2;
  const float* w3; const float* b3;
  float* scratch; void* u; void* out;
  const float* res_scale_dev;  /* optional device float[6]: per-net conditioning_scale (CL:266-270), graph-updatable */
  const void* addend;          /* optional [N,HW,C] dtype: the UNet skip / mid tensor this residual is added to
                                * (PL:500-510): out = round(block) + addend, rounded like the separate add */
  float res_scale[6];          /* per-net conditioning_scale applied to res[i] on load (multiplied with the above) */
  int32_t N, HW, C;
  float eps;
  int32_t dtype;
} es_fusion_desc;
int es_fusion_block(const es_fusion_desc* d, void* stream);
size_t es_fusion_scratch_bytes(int N);
/* All fusion blocks of one denoising step (MC:103-114, 160-169: 12 down + 1 mid) in three launches instead of 3 per
 * block: same arithmetic per block as es_fusion_block; every block needs its own scratch and u buffer; all share N and
 * dtype. */
#define ES_FUSION_MAX_BATCH 13
int es_fusion_blocks(const es_fusion_desc* descs, int count, void* stream);

/* Sinusoidal timestep features (diffusers Timesteps(dim, flip_sin_to_cos=True, freq_shift=0); CL:150-155):
 * out[n, :] = [cos(t_n f_i) | sin(t_n f_i)], dtype.  t: fp32 [N] device. */
int es_timestep_embedding(const float* t, void* out, int N, int dim, int dtype, void* stream);

/* CFG combine + DDIM(eta=0) step (model/edgestyle_pipeline.py:513-522):
 *   eps = e_u + g (e_c - e_u);  x0 = (x - sqrt(1-a_t) eps)/sqrt(a_t);  x <- sqrt(a_prev) x0 + sqrt(1-a_prev) eps
 * noise: [2B or B, HW, L] dtype NHWC (uncond first); latents fp32 [B,HW,L] in/out; model_in: dtype [2B or B,HW,Lstride]
 * rewritten for the next step (CFG duplicate, PL:443-447).  coef: device fp32 table [steps][4] =
 * {sqrt(a_t), sqrt(1-a_t), sqrt(a_prev), sqrt(1-a_prev)}, row selected by *step_idx (device int32). */
int es_cfg_ddim_step(const void* noise, float* latents, void* model_in, const float* coef, const int32_t* step_idx,
                     float guidance_scale, int B, int HW, int L, int Lstride, int cfg, int nsteps, int dtype,
                     void* stream);   /* *step_idx is clamped to [0, nsteps) */

/* CFG combine + one UniPCMultistepScheduler step (the scheduler the reference's callers swap in, TT:273 / APP:118):
 * bh2, solver_order <= 2, predict_x0, epsilon prediction.  The corrector/predictor updates are linear in
 * {last_sample, m0, m1, x0}; coef: device fp32 [steps][12] = {alpha_t, sigma_t, use_corrector, c_last, c_m0, c_m1,
 * c_x0, p_x, p_x0, p_m0, 0, 0} computed on the host (edgestyle_amd/schedulers.py).  last_sample/m0/m1: fp32 state
 * [B,HW,L], zero before the first step. */
int es_cfg_unipc_step(const void* noise, float* latents, float* last_sample, float* m0, float* m1, void* model_in,
                      const float* coef, const int32_t* step_idx, float guidance_scale, int B, int HW, int L,
                      int Lstride, int cfg, int nsteps, int dtype, void* stream);

/* Layout / dtype conversion at the drop-in boundary (callers hand NCHW fp32, TT:328-359). */
int es_nchw_f32_to_nhwc(const float* in, void* out, int N, int C, int HW, int Cpad, int dtype, void* stream);
int es_nhwc_to_nchw_f32(const void* in, float* out, int N, int C, int HW, int Cstride, float scale, float shift,
                        int clamp01, int dtype, void* stream);
/* y = a + b (elementwise, dtype), n elements (n % 8 == 0) */
int es_add(const void* a, const void* b, void* y, int64_t n, int dtype, void* stream);
/* DiagonalGaussianDistribution.sample()*scaling_factor (CL:39-40): moments [N,HW,2L] -> z [N,HW,Lpad] */
int es_vae_sample(const void* moments, const float* noise_nchw, void* z, int N, int HW, int L, int Lpad,
                  float scaling, int dtype, void* stream);
/* dst (int32 device) += 1  — advances the step counter inside a captured graph */
int es_incr(int32_t* ctr, void* stream);
/* Measurement tool (tools/clock_in_kernel.py; no reference counterpart): ONE wave on `stream` watches the shader-cycle counter against the
 * 100 MHz real-time counter for duration_us and writes {shader cycles, 100 MHz ticks} to out2 (device u64[2]) - launched on a side
 * stream it reports the clock the chip holds while the work on the other streams runs.  Not free: a resident wave keeps a workgroup of
 * the one-per-CU kernels (256 x 320 tile, linear_xs) off its CU, so a launch of exactly 256 such workgroups takes two rounds beside it -
 * which is why bench.py samples rocm-smi instead (same reading).  Never part of a plan. */
int es_clock_probe(unsigned long long* out2, unsigned int duration_us, void* stream);
/* out[0..row_len) = table[clamp(*idx, 0, nrows-1)][0..row_len) — selects this step's timestep / conditioning-scale row inside a
 * captured graph (PL:435, PL:464-470) so a replay needs no host-side scalar update */
int es_gather_row(const float* table, const int32_t* idx, float* out, int row_len, int nrows, void* stream);

/* DiagonalGaussian / scheduler glue used by the native loop: latents fp32 [B,HW,L] -> the networks' input (dtype,
 * channels zero-padded to Lstride, duplicated for CFG; PL:443-447) */
int es_latents_to_input(const float* latents, void* model_in, int B, int HW, int L, int Lstride, int cfg, int dtype,
                        void* stream);
/* device-to-device copy / strided copy / fp32 fill on `stream`, as C-ABI calls so that they are part of a recorded plan */
int es_memcpy(void* dst, const void* src, size_t bytes, void* stream);
int es_memcpy2d(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t height, void* stream);
int es_fill_f32(float* dst, float value, size_t n, void* stream);

/* =========================================================================================================
 * Step-level ABI: a denoising step, the denoising loop and the VAE decode behind plain device pointers.
 *
 * es_plan — a recorded launch list.  Between es_plan_begin_record and es_plan_end_record every C-ABI call made on the
 * calling thread is appended to the plan (and still executed / stream-captured as usual); es_plan_launch re-issues the
 * list on any stream.  The pointers inside are the ones recorded: the builder keeps those buffers alive.
 * es_ctx  — owns up to ES_PLAN_COUNT plans plus the static device buffers they read and write (bound by slot), and
 * runs them: as hipGraphs instantiated from the launch lists (default) or launch by launch.
 * The reference has no FFI; what these entry points replace is Python:
 *   es_denoise_step  == OnnxUNetAndControlnets.forward                 export_onnx.py:43-74
 *   es_denoise_loop  == the loop of EdgeStyleStableDiffusionControlNetPipeline.__call__   model/edgestyle_pipeline.py:435-543
 *   es_vae_decode    == vae.decode(latents / scaling_factor) + postprocess               model/edgestyle_pipeline.py:552-572
 *   es_prepare_conds == prepare_image + CachedControlNetModel.preprocess_image            model/edgestyle_pipeline.py:629-664, controllora.py:289-290
 * A context is BUILT either by es_load_weights below (natively, from raw state-dict tensors) or by a host that walks the model
 * itself (edgestyle_amd/native.py NativeEngine: packs the weights, allocates the static buffers, records the plans); after that
 * no interpreter is involved in these calls.
 * ========================================================================================================= */
typedef struct es_plan es_plan;
typedef struct es_ctx es_ctx;
es_plan* es_plan_create(void);
void es_plan_destroy(es_plan* p);
int es_plan_begin_record(es_plan* p);
int es_plan_end_record(es_plan* p);
int es_plan_size(const es_plan* p);                    /* number of recorded calls */
int es_plan_count(const es_plan* p, int kind);         /* ... of one kind (csrc/plan.h: 1 = es_conv_gemm, 2 = es_linear_xs, ...) */
int es_plan_launch(const es_plan* p, void* stream);
/* flat image of a plan: returns the bytes needed; writes them when cap suffices.  The recorded device addresses inside
 * are NOT relocated (edgestyle_amd/native.py save() builds the relocation tables es_ctx_load applies) */
size_t es_plan_export(const es_plan* p, void* out, size_t cap);
/* The pointer fields of a recorded call, by op kind (csrc/plan.h es_op_kind): byte offsets inside the argument record and how
 * the op uses the memory behind each (1 = reads, 2 = writes, 3 = both); *elem_bytes != 0: the record is an array of elements
 * of that size (es_fusion_blocks).  Returns the field count and fills at most `cap` entries.  This is what relocates a plan:
 * exactly these words are device addresses (NativeEngine.save), nothing is guessed from bit patterns. */
int es_plan_pointer_fields(int kind, int32_t* offsets, int32_t* uses, int cap, int32_t* elem_bytes);
es_plan* es_plan_import(const void* data, size_t size);

enum { ES_PLAN_STEP_GENERIC = 0,   /* es_denoise_step: text K/V projections + condition slots + time embedding + step */
       ES_PLAN_PREP = 1,           /* es_denoise_loop, once: text K/V projections, condition slots, time-projection table */
       ES_PLAN_STEP = 2,           /* es_denoise_loop, per step: table-driven step + CFG + scheduler + counter */
       ES_PLAN_DECODE = 3,         /* es_vae_decode */
       ES_PLAN_CONDS = 4,          /* es_prepare_conds: RGB condition images -> the condition embeddings in ES_BUF_COND* */
       ES_PLAN_STEP_UNET = 5,      /* es_denoise_loop, per step OUTSIDE the control-guidance window (PL:419-427: every controlnet_keep is 0): the
                                    * UNet alone plus the constant residuals the fusion blocks return for zero inputs - the six ControlNet passes
                                    * and the fusion launches are skipped, not multiplied by zero.  Optional: without it such a step runs
                                    * ES_PLAN_STEP with scale 0, like the reference. */
       ES_PLAN_COUNT = 6 };
enum { ES_BUF_SAMPLE = 0,          /* dtype [N,h,w,latent_pad]: the networks' input (both CFG halves) */
       ES_BUF_T_ROWS, ES_BUF_EHS,  /* fp32 [kmax*N] timestep copies; dtype [N,77,D] text states */
       ES_BUF_COND0, ES_BUF_COND1, ES_BUF_COND2, ES_BUF_COND3, ES_BUF_COND4, ES_BUF_COND5,   /* dtype [N,h,w,C0] each */
       ES_BUF_SCALES,              /* fp32 [n_conds] conditioning scales of the current step */
       ES_BUF_NOISE,               /* dtype [N,h,w,out_channels] noise prediction */
       ES_BUF_LATENTS,             /* fp32 [B,h,w,L] */
       ES_BUF_STEP_IDX,            /* int32 device step counter */
       ES_BUF_T_TABLE, ES_BUF_SCALE_TABLE, ES_BUF_COEF, ES_BUF_TIMESTEPS,   /* fp32 [T,kmax*N], [T,n_conds], [T,4], [T] */
       ES_BUF_IMAGE,               /* fp32 [B,3,8h,8w] NCHW decoded image in [0,1] */
       /* es_prepare_conds inputs: the RGB condition image of net i, fp32 NCHW [B,3,8h,8w] (VAE-conditioned nets: in [-1,1],
        * pose nets: in [0,1], as the reference's callers pass them, TT:29-48), and - VAE-conditioned nets only - the
        * latent_dist.sample() noise of CL:39 for the CFG-duplicated batch, fp32 [N,latent_channels,h,w] */
       ES_BUF_COND_IMG0, ES_BUF_COND_IMG1, ES_BUF_COND_IMG2, ES_BUF_COND_IMG3, ES_BUF_COND_IMG4, ES_BUF_COND_IMG5,
       ES_BUF_COND_NOISE0, ES_BUF_COND_NOISE1, ES_BUF_COND_NOISE2, ES_BUF_COND_NOISE3, ES_BUF_COND_NOISE4, ES_BUF_COND_NOISE5,
       ES_BUF_HIST0, ES_BUF_HIST1, ES_BUF_HIST2,   /* UniPC multistep state: last_sample, m0, m1 - fp32 [B,h,w,L] each (es_ctx_set_scheduler) */
       ES_BUF_COUNT };
enum { ES_SCHED_DDIM = 0, ES_SCHED_UNIPC = 1 };
typedef struct {
  int32_t B, cfg;                  /* images per call; 1 = classifier-free guidance (N = 2B) */
  int32_t h, w;                    /* latent size */
  int32_t latent_channels, latent_pad;
  int32_t n_conds;                 /* 6 (the fused multi-ControlNet) or 1 (a single ControlNet) */
  int32_t n_steps;                 /* the ES_PLAN_PREP time table is built for this many steps */
  int32_t dtype;
  int32_t guess_mode;              /* 1: the recorded step is guess_mode's (CL:256-264, PL:453-459, 487-497): per-net chains with log-spaced
                                    * level scales; under CFG the ControlNets see the conditional half only - the condition slots then hold
                                    * B samples - and the fused residuals are added to that half */
} es_ctx_geometry;
int es_ctx_create(int device, es_ctx** out);
void es_ctx_destroy(es_ctx* c);                         /* destroys its plans too */
int es_ctx_set_geometry(es_ctx* c, const es_ctx_geometry* g);
int es_ctx_set_plan(es_ctx* c, int which, es_plan* p);  /* the context takes ownership of the plan */
int es_ctx_bind(es_ctx* c, int slot, void* dev, size_t bytes);
void* es_ctx_buffer(const es_ctx* c, int slot, size_t* bytes);
size_t es_ctx_arena_bytes(const es_ctx* c);   /* size of the context's own arena (es_load_weights / es_ctx_load; a dry build reports what it would be) */   /* the memory bound to a slot (borrowed; NULL if unbound) */
/* cond_scales: float[6] or NULL (keep); control guidance window (PL:419-427); use_graphs: 0 = re-issue launch by launch,
 * 1 = one hipGraph per plan (es_denoise_loop launches the step graph n times), 2 = additionally the preparation and all n
 * steps of es_denoise_loop as ONE graph (instantiated on first use per (n_steps, guidance scale)) */
int es_ctx_set_options(es_ctx* c, const float* cond_scales, float control_guidance_start, float control_guidance_end,
                       int use_graphs);
int es_ctx_set_alphas_cumprod(es_ctx* c, const float* alphas_cumprod, int n);   /* scheduler schedule (default: SD1.5's) */
/* The update es_denoise_loop applies after each step's noise prediction: ES_SCHED_DDIM (eta 0; the BASELINE metric's scheduler,
 * model/edgestyle_pipeline.py:520-522 with the pipeline's default) or ES_SCHED_UNIPC - UniPCMultistepScheduler with the SD1.5
 * scheduler config, bh2, order 2, the scheduler the reference's callers assign (test_text2image_pretrained_openpose.py:273,
 * app.py:118); its timesteps come from the caller like DDIM's (UniPC spaces them over n + 1 intervals: 951, 901, ... for 20
 * steps).  The recorded step list is the same: the context re-issues its scheduler call as es_cfg_unipc_step on the
 * ES_BUF_HIST* slots with a n_steps x 12 coefficient table in ES_BUF_COEF (contexts of es_load_weights and NativeEngine bind both). */
int es_ctx_set_scheduler(es_ctx* c, int scheduler);
/* UniPC works from a double-precision schedule like the Python scheduler (default: SD1.5's scaled_linear betas, derived in double;
 * the fp32 table of es_ctx_set_alphas_cumprod is the DDIM update's) */
int es_ctx_set_alphas_cumprod_f64(es_ctx* c, const double* alphas_cumprod, int n);
/* host-only: out[n][12], the UniPC coefficient rows of a timestep list (alphas_cumprod NULL = SD1.5's schedule, in double) */
int es_unipc_coef_table(const double* alphas_cumprod, int n_alphas, const float* timesteps, int n, float* out);
int es_ctx_plan_size(const es_ctx* c, int which);
es_plan* es_ctx_plan(es_ctx* c, int which);             /* borrowed */
/* Load a context image written by NativeEngine.save(path) (edgestyle_amd/native.py): packed weights, tables, static buffers,
 * the five launch lists and their pointer relocations.  No Python, torch or model code is needed to load or run it: build once
 * (es_load_weights or a Python host), ship the image, start in under a second. */
int es_ctx_load(const char* path, int device, es_ctx** out);
/* Write a context that owns its arena - one built by es_load_weights, or loaded by es_ctx_load - as such an image (same format):
 * build from checkpoints once (seconds), start from the image afterwards (one hipMalloc + copies).  Contexts of a Python host
 * hold pointers into the host's allocator pools instead of one arena: NativeEngine.save writes those. */
int es_ctx_save(const es_ctx* c, const char* path);
/* ---------------------------------------------------------------------------------------------------------
 * es_load_weights - build a context from the reference's state dicts, natively.
 * Replaces, on the loading side: EdgeStyleMultiControlNetModel.from_pretrained + load_state_dict
 * (model/edgestyle_multicontrolnet.py:173-211, 289-430), ControlLoRAModel.tie_weights / load_state_dict / fuse
 * (model/controllora.py:600-632, 728-777) and the module construction of the diffusers UNet / ControlNet / AutoencoderKL
 * the reference instantiates (test_text2image_pretrained_openpose.py:224-261): everything between "tensors of a
 * checkpoint in host memory" and "a context es_denoise_loop can run".
 * Tensors are described, not copied: `key` is the state-dict key exactly as the reference's checkpoints spell it
 * (diffusers UNet2DConditionModel / ControlNetModel / AutoencoderKL names; `<linear>.lora_layer.{down,up}.weight` and
 * `controlnet_down_blocks.*` / `controlnet_mid_block.*` for a ControlLoRA net, CL:600-606; `multi_controlnet_down_blocks.{i}.*`
 * / `multi_controlnet_mid_block.*` for the fusion blocks, MC:173-193), `data` a pointer (host memory, or device memory of
 * `device`: read back first) to a C-contiguous tensor of `dtype` (ES_F32 / ES_F16 / ES_BF16) that stays valid during the call;
 * nothing is kept: the caller keeps ownership.  A safetensors file maps onto this directly.
 * The builder folds W + B.A into private copies (the UNet's tensors are never modified), folds every LayerNorm into the Linear
 * it feeds, ff.net.2 into proj_out and conv_shortcut behind conv2, packs everything into the kernels' layouts, lays out ONE
 * device arena (weights, static buffers, activations with lifetime-based reuse, split-K workspace), records the five launch
 * lists and binds the slots.  A missing key or a tensor of the wrong shape is an error naming the key.
 * Scope: the reference's configurations - six condition slots over 1..3 distinct ControlNets + the UNet through the fusion blocks
 * (TT:252-258), or n_conds = 1: ONE ControlNet whose 13 residuals go to the UNet directly (PL:338-351; `fusion` is not read) -
 * with 64-aligned channel widths; DDIM or UniPC (es_ctx_set_scheduler).  device -1: dry build - everything but the device allocation and the upload (plans keep
 * arena-relative addresses; for inspection with es_ctx_plan / es_plan_export on a host without a GPU); device -2: the same
 * with the arena in host memory, contents included (what tests read the packed weights from).  Neither can launch: every entry point that would
 * run one of their plans returns an error that says so.
 * --------------------------------------------------------------------------------------------------------- */
typedef struct {
  const char* key;
  const void* data;                /* host (or device) memory, C-contiguous; borrowed for the duration of the call */
  int64_t shape[4];
  int32_t ndim;                    /* 1..4 */
  int32_t dtype;                   /* ES_F32 | ES_F16 | ES_BF16 */
} es_tensor;
typedef struct { const es_tensor* tensors; int32_t count; } es_state_dict;
enum { ES_NET_CONTROLNET = 0,        /* ControlNetModel: its own encoder + conv-stack conditioning embedding (the openpose net, TT:247-250) */
       ES_NET_CONTROL_LORA_VAE = 1,  /* ControlLoRAModel(uses_vae): LoRA + zero-convs, encoder tied to the UNet, conditioned through the VAE (CL:28-42) */
       ES_NET_CONTROL_LORA = 2 };    /* ControlLoRAModel with its own conv-stack conditioning embedding */
typedef struct {
  es_state_dict unet, vae, fusion;
  es_state_dict controlnet[6];     /* the distinct nets */
  int32_t controlnet_kind[6];      /* ES_NET_* */
  int32_t n_controlnets;
  int32_t net_of_cond[6];          /* which net serves condition slot i (TT:252-258: {0, 1, 2, 1, 2, 1}); n_conds entries are read */
} es_weights;
typedef struct {                   /* config.json of the SD1.5 checkpoints (README.md:131-135; MC:73-102 hard-codes their consequences) */
  int32_t in_channels, out_channels;
  int32_t n_blocks, block_out_channels[4], down_has_attn[4];
  int32_t layers_per_block, num_heads, cross_attention_dim, norm_num_groups;
  float norm_eps;
  int32_t n_cond_embed, cond_embed_channels[4], conditioning_channels;
  int32_t text_tokens;             /* 77 */
  int32_t vae_n_blocks, vae_block_out_channels[4], vae_layers_per_block, vae_latent_channels, vae_norm_num_groups;
  float vae_norm_eps, vae_scaling_factor;
} es_model_config;
/* g: B, cfg, h, w, n_conds (6 | 1), n_steps, dtype are read; latent_channels / latent_pad are derived from the config */
int es_load_weights(const es_weights* w, const es_model_config* cfg, const es_ctx_geometry* g, int device, es_ctx** out);
/* The launch planner the builders share (host-only): the N tile (bn), split-K factor and LDS ring depth es_load_weights picks for
 * an es_conv_gemm launch of M output pixels among the candidate tiles `bns` (edgestyle_amd/ops.py plan_gemm makes the same
 * choice for the Python host; tests hold the two against each other), and whether a plain linear layer takes es_linear_xs. */
int es_plan_gemm_choice(long long M, int rows_padded, int kpad, int geglu, const int* bns, int n_bns, int allow_split,
                        int* bn, int* splitk, int* stages);
int es_linear_xs_eligible(long long M, int ksize, int kpad, int cin, int ctail, int cout, int geglu);
/* Does the 256 x 320 tile (bn = 320) implement the epilogue of a splitk == 1 launch with this form?  (It keeps the common
 * forms only - bias [+ one time-embedding row per 128-pixel half] [+ residual]; split-K launches write raw partials: every
 * form.)  1 = yes.  Both hosts ask BEFORE they fix bn, so that the tile a launch is planned, recorded and reported with is the
 * tile that runs (es_conv_gemm itself still falls back to the 128 x 160 tile for a caller that did not ask). */
int es_conv_gemm8p_form_ok(int act, int cout, int has_temb, long long out_hw, int has_residual);
/* 1 when this process runs under a profiler that intercepts the HSA queues (rocprofv3: ROCP_TOOL_LIBRARIES / a rocprofiler
 * library in LD_PRELOAD / a loaded rocprofiler_configure) while the HIP runtime still block-copies pre-built graph packets
 * (DEBUG_CLR_GRAPH_PACKET_CAPTURE unset or not 0): hipGraphLaunch of a context's graphs then faults inside the runtime
 * (profiles/r04_rocprof_graph_fault.txt).  The context replays its plans launch by launch in that case. */
int es_ctx_graph_hazard(void);
/* es_plan_set_dry(1): while a plan records on this thread, calls are validated and recorded but nothing is launched (the
 * pointers need not exist yet).  Returns the previous setting. */
int es_plan_set_dry(int on);

/* run ONE plan of the context on `stream` (guidance_scale: pointer to the CFG scale for the scheduler call of
 * ES_PLAN_STEP, or NULL = as recorded): single-stepping a prepared loop */
int es_ctx_launch_plan(es_ctx* c, int which, const float* guidance_scale, void* stream);
/* host-only helper: out[n][4] = {sqrt(a_t), sqrt(1-a_t), sqrt(a_prev), sqrt(1-a_prev)} for the DDIM (eta 0) update of
 * each timestep of the list, a_prev = alphas_cumprod of the next timestep (alphas_cumprod[0] after the last one: the
 * SD1.5 scheduler_config's set_alpha_to_one False); alphas_cumprod NULL = SD1.5's scaled_linear schedule */
int es_ddim_coef_table(const float* alphas_cumprod, int n_alphas, const float* timesteps, int n, float* out);
/* sample dtype [N,h,w,latent_pad]; ehs dtype [N,77,D]; cond_embeds[n_conds] dtype [N,h,w,C0]; scales float[n_conds] (host)
 * or NULL (ones); out_noise dtype [N,h,w,out_channels].  All device pointers except `scales`.
 * Stream capture: es_denoise_step and es_denoise_loop stage host values (timestep, scales, per-step tables) through pinned
 * memory and REFUSE a capturing stream (-1); they replay hipGraphs of their own (es_ctx_set_options use_graphs).  To place a
 * step inside a graph of yours, fill the bound buffers yourself and capture es_ctx_launch_plan, which es_vae_decode and
 * es_prepare_conds (device-to-device copies + one plan) also support. */
int es_denoise_step(es_ctx* c, const void* sample, float t, const void* ehs, const void* const* cond_embeds,
                    const float* scales, void* out_noise, void* stream);
/* latents fp32 [B,h,w,L] (device, in/out); ehs as above; timesteps: HOST float[n_steps] (981, 961, ... for 50 steps);
 * the condition embeddings are the ones last copied into ES_BUF_COND* (es_denoise_step does, or the builder). */
int es_denoise_loop(es_ctx* c, float* latents_inout, const void* ehs, float guidance_scale, const float* timesteps,
                    int n_steps, void* stream);
int es_vae_decode(es_ctx* c, const float* latents, float* out_img, void* stream);
/* == prepare_image + the one-time conditioning embedding (model/edgestyle_pipeline.py:352-377, 629-664;
 * model/controllora.py:28-42, 289-290): images[n_conds] device fp32 NCHW [B,3,8h,8w]; noise[n_conds] device fp32
 * [N,latent_channels,h,w] for the nets whose ES_BUF_COND_NOISE slot is bound (the VAE-conditioned ones: the library has no
 * RNG, the host draws what the reference's global generator would), ignored (may be NULL) for the others.  Each shared
 * encoder (the VAE of the LoRA nets, the conv stack of the pose net) runs once over the un-duplicated images of all its
 * nets; results land in ES_BUF_COND0.. as [N,h,w,C0], ready for es_denoise_loop / es_ctx_launch_plan. */
int es_prepare_conds(es_ctx* c, const float* const* images, const float* const* noise, void* stream);

#ifdef __cplusplus
}
#endif
#endif
