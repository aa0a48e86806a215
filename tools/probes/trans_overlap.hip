// Probe: do transcendental instructions (v_exp_f32: 8 issue cycles alone) overlap with ordinary VALU instructions on gfx950 - inside one wave's
// stream (alternating independent chains), and between the two waves of a SIMD (one wave all v_exp_f32, its partner all v_fma_f32)?
// If they do, the V phase of the head_dim-40 attention (half of it v_exp_f32) has slack to win by interleaving; if not, it is at its floor.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/trans_overlap.hip -o gpurun_out/trans_overlap && gpurun_out/trans_overlap
#include <hip/hip_runtime.h>
#include <cstdio>

// MODE 0: 32 v_exp_f32      1: 32 v_fma_f32      2: 16 exp + 16 fma alternating      3: 16 exp + 16 v_cvt_pk_f16_f32 alternating
//      4: 16 exp + 16 v_max3 alternating          5: waves 0-3 all exp (32), waves 4-7 all fma (32)   6: 16 exp then 16 fma (blocks, not alternating)
//      7: 8 exp + 24 fma alternating 1:3
template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, int iters, int nwaves) {
  const int wave = threadIdx.x >> 6;
  if (wave >= nwaves) return;
  float e[16], f[16];
  for (int i = 0; i < 16; ++i) { e[i] = -0.001f * (threadIdx.x + i) - 0.5f; f[i] = 0.5f + 0.001f * i; }
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0 || (MODE == 5 && wave < 4)) {
#pragma unroll
      for (int i = 0; i < 16; ++i) { asm volatile("v_exp_f32 %0, %0" : "+v"(e[i])); }
#pragma unroll
      for (int i = 0; i < 16; ++i) { asm volatile("v_exp_f32 %0, %0" : "+v"(f[i])); }
    } else if (MODE == 1 || (MODE == 5 && wave >= 4)) {
#pragma unroll
      for (int i = 0; i < 16; ++i) { asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(e[i])); }
#pragma unroll
      for (int i = 0; i < 16; ++i) { asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(f[i])); }
    } else if (MODE == 2) {
#pragma unroll
      for (int i = 0; i < 16; ++i) { asm volatile("v_exp_f32 %0, %0\n v_fma_f32 %1, %1, %1, %1" : "+v"(e[i]), "+v"(f[i])); }
    } else if (MODE == 3) {
#pragma unroll
      for (int i = 0; i < 16; ++i) { asm volatile("v_exp_f32 %0, %0\n v_cvt_pk_f16_f32 %1, %1, %1" : "+v"(e[i]), "+v"(f[i])); }
    } else if (MODE == 4) {
#pragma unroll
      for (int i = 0; i < 16; ++i) { asm volatile("v_exp_f32 %0, %0\n v_max3_f32 %1, %1, %1, %1" : "+v"(e[i]), "+v"(f[i])); }
    } else if (MODE == 6) {
#pragma unroll
      for (int i = 0; i < 16; ++i) { asm volatile("v_exp_f32 %0, %0" : "+v"(e[i])); }
#pragma unroll
      for (int i = 0; i < 16; ++i) { asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(f[i])); }
    } else if (MODE == 7) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        asm volatile("v_exp_f32 %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3"
                     : "+v"(e[i]), "+v"(f[i]), "+v"(f[8 + i]), "+v"(e[8 + i]));
      }
    }
  }
  float acc = 0.f;
  for (int i = 0; i < 16; ++i) acc += e[i] + f[i];
  if (acc == 12345.678f) out[threadIdx.x] = acc;
}

template <int MODE>
void run(float* out, const char* name, double alone_cycles) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int it = 4000;
  for (int nw = 4; nw <= 8; nw += 4) {
    if (MODE == 5 && nw == 4) continue;
    k<MODE><<<256, 512>>>(out, it, nw);
    hipDeviceSynchronize();
    float best = 1e9;
    for (int r = 0; r < 5; ++r) {
      hipEventRecord(e0);
      k<MODE><<<256, 512>>>(out, it, nw);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
      best = ms < best ? ms : best;
    }
    const double per_iter_ns = best * 1e6 / it;                  // one loop iteration = 32 instructions per wave
    printf("%-56s %d wave(s)/SIMD: %8.1f us  %6.1f ns per 32-instruction iteration (%.0f cycles at 2.4 GHz; issue costs added up: %.0f per wave)\n",
           name, nw / 4, best * 1e3, per_iter_ns, per_iter_ns * 2.4, alone_cycles);
  }
}

int main() {
  float* out; (void)hipMalloc(&out, 4096);
  run<0>(out, "32 v_exp_f32", 256);
  run<1>(out, "32 v_fma_f32", 128);
  run<2>(out, "16 v_exp_f32 + 16 v_fma_f32, alternating", 192);
  run<6>(out, "16 v_exp_f32 then 16 v_fma_f32", 192);
  run<7>(out, "8 v_exp_f32 + 24 v_fma_f32, 1:3", 160);
  run<3>(out, "16 v_exp_f32 + 16 v_cvt_pk_f16_f32, alternating", 192);
  run<4>(out, "16 v_exp_f32 + 16 v_max3_f32, alternating", 192);
  run<5>(out, "waves 0-3: 32 v_exp_f32 | waves 4-7: 32 v_fma_f32", 384);
  return 0;
}
