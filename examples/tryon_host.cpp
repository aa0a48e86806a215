// A complete host for the try-on path in plain C++ against the C ABI (include/edgestyle_hip.h) and the HIP runtime: no
// Python, no torch, no model code.  It loads a context image (edgestyle_amd/native.py NativeEngine.save), feeds RGB
// condition images, runs the 6-ControlNet denoising loop and the VAE decode, and writes the latents and the image.
//
//   hipcc -O2 -Iinclude examples/tryon_host.cpp -Ledgestyle_amd/lib -ledgestyle_hip -Wl,-rpath,$PWD/edgestyle_amd/lib -o tryon_host
//   ./tryon_host ctx.esctx inputs.bin outputs.bin
//
// inputs.bin (little endian, written by tests/test_native_gpu.py):
//   int32 B, h, w, L, D, n_conds, n_steps, has_noise[n_conds]; float guidance_scale; float timesteps[n_steps];
//   float latents[B*h*w*L] (NHWC); uint16 ehs[2B*77*D] (fp16 bits, negative prompt rows first);
//   per net: float image[B*3*8h*8w] (NCHW), then - if has_noise - float noise[2B*L*h*w]
// outputs.bin: float latents[B*h*w*L]; float image[B*3*8h*8w]
//
// What each call replaces in the reference: es_prepare_conds = prepare_image + preprocess_image (PL:629-664, CL:289-290),
// es_denoise_loop = the loop of EdgeStyleStableDiffusionControlNetPipeline.__call__ (PL:435-543), es_vae_decode = PL:552-572.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

#include "edgestyle_hip.h"

#define HIP_OK(x) do { if ((x) != hipSuccess) { fprintf(stderr, "HIP error at %s:%d\n", __FILE__, __LINE__); return 2; } } while (0)
#define ES_OK(x) do { if ((x) != 0) { fprintf(stderr, "%s: %s\n", #x, es_last_error()); return 3; } } while (0)

template <typename T>
static bool rd(FILE* f, std::vector<T>& v, size_t n) { v.resize(n); return fread(v.data(), sizeof(T), n, f) == n; }

int main(int argc, char** argv) {
  if (argc != 4) { fprintf(stderr, "usage: %s ctx.esctx inputs.bin outputs.bin\n", argv[0]); return 1; }
  FILE* f = fopen(argv[2], "rb");
  if (!f) { perror(argv[2]); return 1; }
  int32_t hd[7];
  if (fread(hd, 4, 7, f) != 7) return 1;
  const int B = hd[0], h = hd[1], w = hd[2], L = hd[3], D = hd[4], nc = hd[5], T = hd[6];
  std::vector<int32_t> has_noise;
  float gs;
  std::vector<float> ts, lat;
  std::vector<uint16_t> ehs;
  if (!rd(f, has_noise, nc) || fread(&gs, 4, 1, f) != 1 || !rd(f, ts, T) || !rd(f, lat, (size_t)B * h * w * L) ||
      !rd(f, ehs, (size_t)2 * B * 77 * D)) return 1;
  const size_t img_n = (size_t)B * 3 * 8 * h * 8 * w, noise_n = (size_t)2 * B * L * h * w;

  es_ctx* ctx = nullptr;
  ES_OK(es_ctx_load(argv[1], 0, &ctx));                       // one hipMalloc + copies + pointer relocation

  std::vector<const float*> d_img(nc, nullptr), d_noise(nc, nullptr);
  std::vector<float> tmp;
  for (int i = 0; i < nc; ++i) {
    float* p = nullptr;
    if (!rd(f, tmp, img_n)) return 1;
    HIP_OK(hipMalloc(&p, img_n * 4));
    HIP_OK(hipMemcpy(p, tmp.data(), img_n * 4, hipMemcpyHostToDevice));
    d_img[i] = p;
    if (has_noise[i]) {
      if (!rd(f, tmp, noise_n)) return 1;
      HIP_OK(hipMalloc(&p, noise_n * 4));
      HIP_OK(hipMemcpy(p, tmp.data(), noise_n * 4, hipMemcpyHostToDevice));
      d_noise[i] = p;
    }
  }
  fclose(f);
  float *d_lat = nullptr, *d_out = nullptr;
  void* d_ehs = nullptr;
  HIP_OK(hipMalloc(&d_lat, lat.size() * 4));
  HIP_OK(hipMemcpy(d_lat, lat.data(), lat.size() * 4, hipMemcpyHostToDevice));
  HIP_OK(hipMalloc(&d_ehs, ehs.size() * 2));
  HIP_OK(hipMemcpy(d_ehs, ehs.data(), ehs.size() * 2, hipMemcpyHostToDevice));
  HIP_OK(hipMalloc(&d_out, img_n * 4));

  hipStream_t st;
  HIP_OK(hipStreamCreate(&st));
  ES_OK(es_prepare_conds(ctx, d_img.data(), d_noise.data(), st));
  ES_OK(es_denoise_loop(ctx, d_lat, d_ehs, gs, ts.data(), T, st));
  ES_OK(es_vae_decode(ctx, d_lat, d_out, st));
  HIP_OK(hipStreamSynchronize(st));

  std::vector<float> out_img(img_n);
  HIP_OK(hipMemcpy(lat.data(), d_lat, lat.size() * 4, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(out_img.data(), d_out, img_n * 4, hipMemcpyDeviceToHost));
  FILE* o = fopen(argv[3], "wb");
  if (!o) { perror(argv[3]); return 1; }
  fwrite(lat.data(), 4, lat.size(), o);
  fwrite(out_img.data(), 4, img_n, o);
  fclose(o);
  es_ctx_destroy(ctx);
  printf("ok\n");
  return 0;
}
