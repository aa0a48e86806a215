// Probe: issue cost of a few VALU instructions on gfx950 (one and two waves per SIMD, 16 independent chains per wave).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h2 __attribute__((ext_vector_type(2)));

template <int OP>
__global__ __launch_bounds__(512) void k(float* out, int iters, int nwaves) {
  if ((int)(threadIdx.x >> 6) >= nwaves) return;
  float e[16];
  for (int i = 0; i < 16; ++i) e[i] = -0.001f * (threadIdx.x + i) - 0.5f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (OP == 0) e[i] = __builtin_amdgcn_exp2f(e[i]);
      if (OP == 1) { _Float16 h = (_Float16)e[i]; asm volatile("v_exp_f16 %0, %1" : "=v"(h) : "v"(h)); e[i] = (float)h; }
      if (OP == 2) asm volatile("v_exp_f16 %0, %0" : "+v"(e[i]));          // raw: low 16 bits as f16
      if (OP == 3) asm volatile("v_exp_f32 %0, %0" : "+v"(e[i]));
      if (OP == 4) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(e[i]));
      if (OP == 5) asm volatile("v_pk_mul_f16 %0, %0, %0" : "+v"(e[i]));
      if (OP == 6) asm volatile("v_cvt_pk_f16_f32 %0, %0, %0" : "+v"(e[i]));
      if (OP == 7) asm volatile("v_max3_f32 %0, %0, %0, %0" : "+v"(e[i]));
      if (OP == 8) asm volatile("v_rcp_f32 %0, %0" : "+v"(e[i]));
      if (OP == 9) asm volatile("v_ldexp_f32 %0, %0, %0" : "+v"(e[i]));
      if (OP == 10) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(*(double*)&e[i & 14]));
    }
  }
  float acc = 0.f;
  for (int i = 0; i < 16; ++i) acc += e[i];
  if (acc == 12345.678f) out[threadIdx.x] = acc;
}

template <int OP>
void run(float* out, const char* name) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int it = 4000;
  for (int nw = 4; nw <= 8; nw += 4) {
    k<OP><<<256, 512>>>(out, it, nw);
    hipDeviceSynchronize();
    float best = 1e9;
    for (int r = 0; r < 5; ++r) {
      hipEventRecord(e0);
      k<OP><<<256, 512>>>(out, it, nw);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
      best = ms < best ? ms : best;
    }
    const double per = best * 1e6 / (double)(it * 16) / (nw / 4);   // ns per wave-instruction per SIMD
    printf("%-28s %d wave(s)/SIMD: %7.1f us  %.2f ns per instruction per SIMD (%.1f cycles at 2.4 GHz)\n", name, nw / 4, best * 1e3, per, per * 2.4);
  }
}

int main() {
  float* out; (void)hipMalloc(&out, 4096);
  run<0>(out, "exp2f builtin (f32)");
  run<3>(out, "v_exp_f32 asm");
  run<2>(out, "v_exp_f16 asm");
  run<1>(out, "cvt + v_exp_f16 + cvt");
  run<8>(out, "v_rcp_f32");
  run<4>(out, "v_fma_f32");
  run<10>(out, "v_pk_fma_f32");
  run<5>(out, "v_pk_mul_f16");
  run<6>(out, "v_cvt_pk_f16_f32");
  run<7>(out, "v_max3_f32");
  run<9>(out, "v_ldexp_f32");
  return 0;
}
