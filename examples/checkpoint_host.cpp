// From checkpoint directories to a try-on image in plain C++: no Python, no torch, no model code of this repository.
// The host memory-maps the safetensors files of the reference's on-disk layout, describes their tensors to es_load_weights
// (include/edgestyle_hip.h) - which folds, packs, lays out the arena and records the launch lists natively - and then drives
// es_prepare_conds / es_denoise_loop / es_vae_decode exactly like examples/tryon_host.cpp does for a context image.
//
//   hipcc -O2 -Iinclude examples/checkpoint_host.cpp -Ledgestyle_amd/lib -ledgestyle_hip -Wl,-rpath,$PWD/edgestyle_amd/lib -o checkpoint_host
//   ./checkpoint_host <unet_dir> <vae_dir> <multi_controlnet_dir> <openpose_dir> inputs.bin outputs.bin [context.esctx]
// (the optional last argument: also write the built context as an image - es_ctx_save - that examples/tryon_host.cpp and
//  es_ctx_load start from in under a second next time)
//
// Directory layout (what the reference reads and writes):
//   <unet_dir>/, <vae_dir>/, <openpose_dir>/      config.json + diffusion_pytorch_model.safetensors (diffusers save_pretrained)
//   <multi_controlnet_dir>/diffusion_pytorch_model.safetensors   the 13 fusion blocks (model/edgestyle_multicontrolnet.py:213-282)
//   <multi_controlnet_dir>/controlnet_0/, controlnet_1/          the two ControlLoRA nets: LoRA matrices + zero-convs only
//                                                                (MC:380-398 with load_pattern [0, None, 1, None, 1, None];
//                                                                model/controllora.py:600-606)
// The six condition slots are [controlnet_0, openpose, controlnet_1, openpose, controlnet_1, openpose]
// (test_text2image_pretrained_openpose.py:252-258).
// inputs.bin / outputs.bin: as examples/tryon_host.cpp; optionally followed by int32 n, float alphas_cumprod[n] (a scheduler
// table to use instead of the library's SD1.5 default).
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "edgestyle_hip.h"

#define HIP_OK(x) do { if ((x) != hipSuccess) { fprintf(stderr, "HIP error at %s:%d\n", __FILE__, __LINE__); return 2; } } while (0)
#define ES_OK(x) do { if ((x) != 0) { fprintf(stderr, "%s: %s\n", #x, es_last_error()); return 3; } } while (0)

// ---- a JSON reader just large enough for safetensors headers and config.json ---------------------------------------------
struct Json {
  enum Kind { Null, Bool, Num, Str, Arr, Obj } kind = Null;
  double num = 0; bool b = false; std::string str;
  std::vector<Json> arr;
  std::vector<std::pair<std::string, Json>> obj;
  const Json* get(const std::string& k) const { for (auto& kv : obj) if (kv.first == k) return &kv.second; return nullptr; }
};
struct JsonParser {
  const char* p; const char* e;
  [[noreturn]] void bad(const char* m) { fprintf(stderr, "json: %s\n", m); exit(4); }
  void ws() { while (p < e && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) ++p; }
  std::string string() {
    std::string s;
    if (*p != '"') bad("string expected");
    for (++p; p < e && *p != '"'; ++p) {
      if (*p == '\\' && p + 1 < e) { ++p; s += (*p == 'n' ? '\n' : *p == 't' ? '\t' : *p); } else s += *p;
    }
    if (p >= e) bad("unterminated string");
    ++p;
    return s;
  }
  Json value() {
    ws();
    if (p >= e) bad("unexpected end");
    Json j;
    if (*p == '{') {
      j.kind = Json::Obj; ++p; ws();
      if (*p == '}') { ++p; return j; }
      for (;;) {
        ws(); std::string k = string(); ws();
        if (*p != ':') bad("':' expected");
        ++p;
        j.obj.emplace_back(k, value()); ws();
        if (*p == ',') { ++p; continue; }
        if (*p == '}') { ++p; return j; }
        bad("',' or '}' expected");
      }
    }
    if (*p == '[') {
      j.kind = Json::Arr; ++p; ws();
      if (*p == ']') { ++p; return j; }
      for (;;) {
        j.arr.push_back(value()); ws();
        if (*p == ',') { ++p; continue; }
        if (*p == ']') { ++p; return j; }
        bad("',' or ']' expected");
      }
    }
    if (*p == '"') { j.kind = Json::Str; j.str = string(); return j; }
    if (!strncmp(p, "true", 4)) { j.kind = Json::Bool; j.b = true; p += 4; return j; }
    if (!strncmp(p, "false", 5)) { j.kind = Json::Bool; p += 5; return j; }
    if (!strncmp(p, "null", 4)) { p += 4; return j; }
    char* end = nullptr;
    j.kind = Json::Num; j.num = strtod(p, &end);
    if (end == p) bad("value expected");
    p = end;
    return j;
  }
};
static Json parse_json(const char* data, size_t n) { JsonParser q{data, data + n}; return q.value(); }

// ---- safetensors: u64 header length, JSON header {key: {dtype, shape, data_offsets}}, raw little-endian data -------------------
struct SafeTensors {
  void* map = nullptr; size_t bytes = 0;
  std::vector<std::string> keys;            // owns the key strings the descriptors point at
  std::vector<es_tensor> tensors;
  bool open(const std::string& path) {
    const int fd = ::open(path.c_str(), O_RDONLY);
    if (fd < 0) { perror(path.c_str()); return false; }
    struct stat st;
    if (fstat(fd, &st) != 0) { close(fd); return false; }
    bytes = (size_t)st.st_size;
    map = mmap(nullptr, bytes, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (map == MAP_FAILED || bytes < 8) { fprintf(stderr, "%s: cannot map\n", path.c_str()); return false; }
    uint64_t hl;
    memcpy(&hl, map, 8);
    if (hl > bytes - 8) { fprintf(stderr, "%s: bad header length\n", path.c_str()); return false; }
    const Json h = parse_json((const char*)map + 8, (size_t)hl);
    const char* data = (const char*)map + 8 + hl;
    keys.reserve(h.obj.size());
    for (const auto& kv : h.obj) {
      if (kv.first == "__metadata__") continue;
      const Json *dt = kv.second.get("dtype"), *sh = kv.second.get("shape"), *of = kv.second.get("data_offsets");
      if (!dt || !sh || !of || of->arr.size() != 2 || sh->arr.size() > 4 || sh->arr.empty()) { fprintf(stderr, "%s: entry '%s' not understood\n", path.c_str(), kv.first.c_str()); return false; }
      es_tensor t;
      memset(&t, 0, sizeof(t));
      t.dtype = dt->str == "F32" ? ES_F32 : dt->str == "F16" ? ES_F16 : dt->str == "BF16" ? ES_BF16 : -1;
      if (t.dtype < 0) { fprintf(stderr, "%s: '%s' has dtype %s (F32 / F16 / BF16 only)\n", path.c_str(), kv.first.c_str(), dt->str.c_str()); return false; }
      t.ndim = (int32_t)sh->arr.size();
      size_t n = t.dtype == ES_F32 ? 4 : 2;
      for (int i = 0; i < t.ndim; ++i) { t.shape[i] = (int64_t)sh->arr[i].num; n *= (size_t)t.shape[i]; }
      const size_t a = (size_t)of->arr[0].num, b = (size_t)of->arr[1].num;
      if (b - a != n || 8 + hl + b > bytes) { fprintf(stderr, "%s: '%s' lies outside the file\n", path.c_str(), kv.first.c_str()); return false; }
      t.data = data + a;
      keys.push_back(kv.first);
      tensors.push_back(t);
    }
    for (size_t i = 0; i < tensors.size(); ++i) tensors[i].key = keys[i].c_str();
    return true;
  }
  es_state_dict dict() const { return es_state_dict{tensors.data(), (int32_t)tensors.size()}; }
};

static bool read_file(const std::string& path, std::string& out) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) { perror(path.c_str()); return false; }
  char buf[4096]; size_t n;
  while ((n = fread(buf, 1, sizeof(buf), f)) > 0) out.append(buf, n);
  fclose(f);
  return true;
}
static int ints(const Json* j, int32_t* dst, int cap) {
  int n = 0;
  if (j) for (const auto& v : j->arr) if (n < cap) dst[n++] = v.kind == Json::Bool ? (int)v.b : (int)v.num;
  return n;
}
static double num(const Json& j, const char* k, double dflt) { const Json* v = j.get(k); return v && v->kind == Json::Num ? v->num : dflt; }

template <typename T>
static bool rd(FILE* f, std::vector<T>& v, size_t n) { v.resize(n); return fread(v.data(), sizeof(T), n, f) == n; }

int main(int argc, char** argv) {
  if (argc != 7 && argc != 8) { fprintf(stderr, "usage: %s unet_dir vae_dir multi_controlnet_dir openpose_dir inputs.bin outputs.bin [context.esctx]\n", argv[0]); return 1; }
  const std::string W = "/diffusion_pytorch_model.safetensors";
  SafeTensors unet, vae, fusion, lora0, lora1, pose;
  if (!unet.open(argv[1] + W) || !vae.open(argv[2] + W) || !fusion.open(argv[3] + W) || !lora0.open(std::string(argv[3]) + "/controlnet_0" + W) ||
      !lora1.open(std::string(argv[3]) + "/controlnet_1" + W) || !pose.open(argv[4] + W)) return 1;

  // config.json of the UNet and the VAE (diffusers names; this repository's own writer uses the same ones for these fields)
  std::string ucs, vcs;
  if (!read_file(std::string(argv[1]) + "/config.json", ucs) || !read_file(std::string(argv[2]) + "/config.json", vcs)) return 1;
  const Json uc = parse_json(ucs.data(), ucs.size()), vc = parse_json(vcs.data(), vcs.size());
  es_model_config mc;
  memset(&mc, 0, sizeof(mc));
  mc.in_channels = (int)num(uc, "in_channels", 4); mc.out_channels = (int)num(uc, "out_channels", 4);
  mc.n_blocks = ints(uc.get("block_out_channels"), mc.block_out_channels, 4);
  if (const Json* t = uc.get("down_block_types")) { int i = 0; for (const auto& s : t->arr) if (i < 4) mc.down_has_attn[i++] = s.str.rfind("CrossAttn", 0) == 0; }
  else ints(uc.get("down_has_attn"), mc.down_has_attn, 4);
  mc.layers_per_block = (int)num(uc, "layers_per_block", 2);
  mc.num_heads = (int)num(uc, "num_heads", num(uc, "attention_head_dim", 8));      // diffusers' SD1.5 config calls the head COUNT attention_head_dim
  mc.cross_attention_dim = (int)num(uc, "cross_attention_dim", 768);
  mc.norm_num_groups = (int)num(uc, "norm_num_groups", 32); mc.norm_eps = (float)num(uc, "norm_eps", 1e-5);
  mc.n_cond_embed = ints(uc.get("conditioning_embedding_out_channels"), mc.cond_embed_channels, 4);
  if (!mc.n_cond_embed) { const int32_t d[4] = {16, 32, 96, 256}; memcpy(mc.cond_embed_channels, d, sizeof(d)); mc.n_cond_embed = 4; }
  mc.conditioning_channels = 3; mc.text_tokens = 77;
  mc.vae_n_blocks = ints(vc.get("block_out_channels"), mc.vae_block_out_channels, 4);
  mc.vae_layers_per_block = (int)num(vc, "layers_per_block", 2); mc.vae_latent_channels = (int)num(vc, "latent_channels", 4);
  mc.vae_norm_num_groups = (int)num(vc, "norm_num_groups", 32); mc.vae_norm_eps = (float)num(vc, "norm_eps", 1e-6);
  mc.vae_scaling_factor = (float)num(vc, "scaling_factor", 0.18215);

  FILE* f = fopen(argv[5], "rb");
  if (!f) { perror(argv[5]); return 1; }
  int32_t hd[7];
  if (fread(hd, 4, 7, f) != 7) return 1;
  const int B = hd[0], h = hd[1], w = hd[2], L = hd[3], D = hd[4], nc = hd[5], T = hd[6];
  if (nc != 6 || D != mc.cross_attention_dim || L != mc.in_channels) { fprintf(stderr, "inputs.bin does not fit the checkpoints\n"); return 1; }
  std::vector<int32_t> has_noise;
  float gs;
  std::vector<float> ts, lat;
  std::vector<uint16_t> ehs;
  if (!rd(f, has_noise, nc) || fread(&gs, 4, 1, f) != 1 || !rd(f, ts, T) || !rd(f, lat, (size_t)B * h * w * L) || !rd(f, ehs, (size_t)2 * B * 77 * D)) return 1;
  const size_t img_n = (size_t)B * 3 * 8 * h * 8 * w, noise_n = (size_t)2 * B * L * h * w;

  // ---- the context, straight from the mapped checkpoints
  es_weights wts;
  memset(&wts, 0, sizeof(wts));
  wts.unet = unet.dict(); wts.vae = vae.dict(); wts.fusion = fusion.dict();
  wts.controlnet[0] = lora0.dict(); wts.controlnet_kind[0] = ES_NET_CONTROL_LORA_VAE;
  wts.controlnet[1] = pose.dict(); wts.controlnet_kind[1] = ES_NET_CONTROLNET;
  wts.controlnet[2] = lora1.dict(); wts.controlnet_kind[2] = ES_NET_CONTROL_LORA_VAE;
  wts.n_controlnets = 3;
  const int32_t slots[6] = {0, 1, 2, 1, 2, 1};
  memcpy(wts.net_of_cond, slots, sizeof(slots));
  es_ctx_geometry geo;
  memset(&geo, 0, sizeof(geo));
  geo.B = B; geo.cfg = 1; geo.h = h; geo.w = w; geo.n_conds = 6; geo.n_steps = T; geo.dtype = ES_F16;
  es_ctx* ctx = nullptr;
  ES_OK(es_load_weights(&wts, &mc, &geo, 0, &ctx));
  printf("context built: %d calls per denoising step\n", es_ctx_plan_size(ctx, ES_PLAN_STEP));

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
  int32_t n_alphas = 0;
  if (fread(&n_alphas, 4, 1, f) == 1 && n_alphas > 0) {
    std::vector<float> ac;
    if (!rd(f, ac, (size_t)n_alphas)) return 1;
    ES_OK(es_ctx_set_alphas_cumprod(ctx, ac.data(), n_alphas));
  }
  fclose(f);
  if (argc == 8) ES_OK(es_ctx_save(ctx, argv[7]));
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
  FILE* o = fopen(argv[6], "wb");
  if (!o) { perror(argv[6]); return 1; }
  fwrite(lat.data(), 4, lat.size(), o);
  fwrite(out_img.data(), 4, img_n, o);
  fclose(o);
  es_ctx_destroy(ctx);
  printf("ok\n");
  return 0;
}
