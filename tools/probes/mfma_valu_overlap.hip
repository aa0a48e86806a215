// Probe: do MFMA work of one wave and transcendental / plain VALU work of ANOTHER wave on the same SIMD overlap on gfx950?
// And inside ONE wave (VALU issued in the shadow of its own MFMAs)?  Build: hipcc --offload-arch=gfx950 -O3 -o overlap_probe ...
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// mode bits: 1 = waves 0-3 run MFMA32 chains, 2 = waves 4-7 run v_exp streams, 4 = waves 4-7 run v_fma streams instead of exp,
// 8 = single role: every wave runs MFMAs with `fill` v_exp after each MFMA (in-wave interleave), 16 = MFMA 16x16x32 instead
template <int MODE, int FILL>
__global__ __launch_bounds__(512) void probe(float* out, int iters) {
  const int wave = threadIdx.x >> 6;
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
  float acc = 0.f;
  if constexpr (MODE & 8) {
    f32x16 c0 = {}, c1 = {};
    float e[8];
    for (int i = 0; i < 8; ++i) e[i] = -0.001f * (threadIdx.x + i);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
#pragma unroll
        for (int f = 0; f < FILL; ++f) e[f] = __builtin_amdgcn_exp2f(e[f]);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
#pragma unroll
        for (int f = 0; f < FILL; ++f) e[(f + 4) & 7] = __builtin_amdgcn_exp2f(e[(f + 4) & 7]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    for (int i = 0; i < 16; ++i) acc += c0[i] + c1[i];
    for (int i = 0; i < 8; ++i) acc += e[i];
  } else {
    const bool mf = wave < 4;
    if (mf) {
      if (MODE & 1) {
        if constexpr (MODE & 16) {
          f32x4 c0 = {}, c1 = {}, c2 = {}, c3 = {};
          for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c0, 0, 0, 0);
              c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c1, 0, 0, 0);
              c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c2, 0, 0, 0);
              c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c3, 0, 0, 0);
            }
          }
          for (int i = 0; i < 4; ++i) acc += c0[i] + c1[i] + c2[i] + c3[i];
        } else {
          f32x16 c0 = {}, c1 = {};
          for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
              c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
            }
          }
          for (int i = 0; i < 16; ++i) acc += c0[i] + c1[i];
        }
      }
    } else {
      if (MODE & 2) {
        float e[8];
        for (int i = 0; i < 8; ++i) e[i] = -0.001f * (threadIdx.x + i);
        for (int it = 0; it < iters; ++it) {
#pragma unroll
          for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
              if (MODE & 4) e[i] = __builtin_fmaf(e[i], 0.999f, 0.001f);
              else e[i] = __builtin_amdgcn_exp2f(e[i]);
            }
        }
        for (int i = 0; i < 8; ++i) acc += e[i];
      }
    }
  }
  if (acc == 12345.678f) out[threadIdx.x] = acc;
}

template <int MODE, int FILL>
float run(float* out, int iters, const char* name, double mfma_per_wave, double valu_per_wave) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  probe<MODE, FILL><<<256 * 4, 512>>>(out, iters);
  hipDeviceSynchronize();
  float best = 1e9;
  for (int r = 0; r < 5; ++r) {
    hipEventRecord(e0);
    probe<MODE, FILL><<<256, 512>>>(out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    best = ms < best ? ms : best;
  }
  printf("%-58s %8.1f us   (%.0f MFMA/wave, %.0f VALU/wave)\n", name, best * 1e3, mfma_per_wave, valu_per_wave);
  return best;
}

int main() {
  float* out; hipMalloc(&out, 4096);
  const int it = 4000;
  printf("one 512-thread workgroup per CU (2 waves per SIMD), %d iterations\n", it);
  run<1, 0>(out, it, "waves 0-3: MFMA 32x32x16 (2 chains)   | waves 4-7: idle", it * 8.0, 0);
  run<2, 0>(out, it, "waves 0-3: idle                      | waves 4-7: v_exp", 0, it * 32.0);
  run<3, 0>(out, it, "waves 0-3: MFMA 32x32x16             | waves 4-7: v_exp", it * 8.0, it * 32.0);
  run<6, 0>(out, it, "waves 0-3: idle                      | waves 4-7: v_fma", 0, it * 32.0);
  run<7, 0>(out, it, "waves 0-3: MFMA 32x32x16             | waves 4-7: v_fma", it * 8.0, it * 32.0);
  run<17, 0>(out, it, "waves 0-3: MFMA 16x16x32 (4 chains)   | waves 4-7: idle", it * 16.0, 0);
  run<19, 0>(out, it, "waves 0-3: MFMA 16x16x32             | waves 4-7: v_exp", it * 16.0, it * 32.0);
  run<8, 0>(out, it, "all 8 waves: MFMA 32x32x16 only (in-wave, fill 0)", it * 8.0, 0);
  run<8, 1>(out, it, "all 8 waves: MFMA 32x32x16 + 1 v_exp per MFMA", it * 8.0, it * 8.0);
  run<8, 2>(out, it, "all 8 waves: MFMA 32x32x16 + 2 v_exp per MFMA", it * 8.0, it * 16.0);
  run<8, 3>(out, it, "all 8 waves: MFMA 32x32x16 + 3 v_exp per MFMA", it * 8.0, it * 24.0);
  run<8, 4>(out, it, "all 8 waves: MFMA 32x32x16 + 4 v_exp per MFMA", it * 8.0, it * 32.0);
  return 0;
}
