// es_load_weights — the native context builder.
//
// Takes the reference's state dicts as plain {key, dtype, shape, host pointer} descriptors - the UNet, the VAE, the ControlNets
// (diffusers keys; ControlLoRA nets hold their LoRA matrices and zero-convs only and are tied to the UNet,
// model/controllora.py:600-632) and the fusion blocks (model/edgestyle_multicontrolnet.py:173-193) - and builds a complete
// es_ctx without any interpreter: weights are folded (W + B.A, LayerNorm into the Linear it feeds, ff.net.2 into proj_out,
// conv_shortcut behind conv2) and packed into the kernels' layouts on the host, the model is walked once per plan while a
// DRY recorder (plan.h) validates and records every C-ABI call against addresses of an arena that does not exist yet, then the
// arena is allocated in one piece, the plans are relocated through the typed pointer-field tables and the packed weights are
// uploaded.  The walk is the one edgestyle_amd/engine.py + models.py (StepRunner, grouped lockstep mode) + native.py perform
// through Python; tests/test_load_weights_*.py hold the two builders against each other call by call and bit by bit.
//
// Scope: the reference's configurations - the fused six-slot multi-ControlNet model (TT:252-258) or ONE ControlNet whose 13
// residuals go to the UNet without fusion blocks (PL:338-351, BASELINE configs[0]) - with 64-aligned channel widths, grouped
// lockstep execution or guess_mode's per-group chains (es_ctx_geometry.guess_mode), DDIM or UniPC (es_ctx_set_scheduler).  What
// the Python builder offers beyond that (latent sizes whose groups do not tile, outside guess_mode) is refused here with an error,
// never approximated.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <deque>
#include <functional>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/edgestyle_hip.h"
#include "plan.h"

extern "C" void es_set_error(const char* msg);
extern "C" int es_plan_set_dry(int on);

namespace {

[[noreturn]] void fail(const std::string& m) { throw std::runtime_error(m); }
inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }
inline int round_up(int a, int b) { return (a + b - 1) / b * b; }
constexpr int BK = 64, BM = 128;
constexpr unsigned long long FAKE_HEAP = 1ull << 44, FAKE_WS = 1ull << 45;

void parallel_for(long long n, const std::function<void(long long, long long)>& fn) {
  unsigned nt = std::thread::hardware_concurrency();
  nt = nt < 1 ? 1 : (nt > 16 ? 16 : nt);
  if (n < 64 || nt == 1) { fn(0, n); return; }
  std::vector<std::thread> th;
  const long long step = (n + nt - 1) / nt;
  for (long long a = 0; a < n; a += step) th.emplace_back(fn, a, std::min(n, a + step));
  for (auto& t : th) t.join();
}

// ---- 16-bit storage types, round to nearest even like torch's .to(dtype) ------------------------------------------------
inline uint16_t f32_to_f16(float f) { _Float16 h = (_Float16)f; uint16_t u; memcpy(&u, &h, 2); return u; }
inline float f16_to_f32(uint16_t u) { _Float16 h; memcpy(&h, &u, 2); return (float)h; }
inline uint16_t f32_to_bf16(float f) {
  uint32_t u; memcpy(&u, &f, 4);
  if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (uint16_t)((u >> 16) | 0x40);      // NaN stays NaN
  return (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}
inline float bf16_to_f32(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }

// ---- state dicts ----------------------------------------------------------------------------------------------------------
struct Dict {
  std::unordered_map<std::string, const es_tensor*> m;
  std::string name;
  void init(const es_state_dict& sd, const char* nm) {
    name = nm;
    if (sd.count < 0 || (sd.count > 0 && !sd.tensors)) fail(std::string("es_load_weights: the ") + nm + " state dict has a count but no tensors");
    for (int i = 0; i < sd.count; ++i) {
      const es_tensor& t = sd.tensors[i];
      if (!t.key || !t.data || t.ndim < 1 || t.ndim > 4 || t.dtype < 0 || t.dtype > ES_F32) fail("es_load_weights: malformed tensor descriptor #" + std::to_string(i) + " in the " + name + " state dict");
      long long n = 1;
      for (int d = 0; d < t.ndim; ++d) { if (t.shape[d] < 1 || t.shape[d] > (1ll << 31)) n = -1; else if (n > 0) n *= t.shape[d]; if (n > (1ll << 33)) n = -1; }
      if (n < 0) fail(std::string("es_load_weights: implausible shape of '") + t.key + "' in the " + name + " state dict");
      m[t.key] = &t;
    }
  }
  const es_tensor* find(const std::string& k) const { auto it = m.find(k); return it == m.end() ? nullptr : it->second; }
};
long long numel(const es_tensor* t) { long long n = 1; for (int i = 0; i < t->ndim; ++i) n *= t->shape[i]; return n; }
// the source tensors may live on a device as well (a host that already holds the checkpoint there): copied to the host first
bool on_device(const void* p) {
  hipPointerAttribute_t a;
  const bool dev = hipPointerGetAttributes(&a, p) == hipSuccess && a.type == hipMemoryTypeDevice;
  (void)hipGetLastError();
  return dev;
}
std::vector<float> to_f32(const es_tensor* t) {
  const long long n = numel(t);
  std::vector<float> v((size_t)n);
  const void* src = t->data;
  std::vector<char> tmp;
  if (on_device(src)) {
    tmp.resize((size_t)n * (t->dtype == ES_F32 ? 4 : 2));
    if (hipMemcpy(tmp.data(), src, tmp.size(), hipMemcpyDeviceToHost) != hipSuccess) fail(std::string("es_load_weights: cannot read '") + t->key + "' from the device");
    src = tmp.data();
  }
  if (t->dtype == ES_F32) memcpy(v.data(), src, (size_t)n * 4);
  else if (t->dtype == ES_F16) { const uint16_t* s = (const uint16_t*)src; for (long long i = 0; i < n; ++i) v[i] = f16_to_f32(s[i]); }
  else { const uint16_t* s = (const uint16_t*)src; for (long long i = 0; i < n; ++i) v[i] = bf16_to_f32(s[i]); }
  return v;
}

// A net's view of its weights: its own dict first, then - ControlLoRA nets, tie_weights CL:623-632 - the UNet's encoder keys;
// `<linear>.lora_layer.{down,up}.weight` pairs are folded into the Linear they adapt (W + B.A, CL:728-777, into a private copy).
struct Weights {
  const Dict* own = nullptr;
  const Dict* tied = nullptr;      // the UNet (ControlLoRA nets) or null
  std::string who;
  static bool encoder_key(const std::string& k) {
    const std::string head = k.substr(0, k.find('.'));
    return head == "conv_in" || head == "time_embedding" || head == "down_blocks" || head == "mid_block";
  }
  const es_tensor* find(const std::string& k) const {
    if (const es_tensor* t = own->find(k)) return t;
    if (tied && encoder_key(k)) return tied->find(k);
    return nullptr;
  }
  bool has(const std::string& k) const { return find(k) != nullptr; }
  const es_tensor* get(const std::string& k) const {
    const es_tensor* t = find(k);
    if (!t) fail("es_load_weights: missing key '" + k + "' in the " + who + " state dict");
    return t;
  }
  const es_tensor* shaped(const std::string& k, std::initializer_list<long long> want) const {
    const es_tensor* t = get(k);
    bool ok = t->ndim == (int)want.size();
    int i = 0;
    for (long long w : want) { if (ok && w >= 0 && t->shape[i] != w) ok = false; ++i; }
    if (!ok) {
      std::string s = "es_load_weights: size mismatch for '" + k + "' in the " + who + " state dict: (";
      for (int j = 0; j < t->ndim; ++j) s += std::to_string(t->shape[j]) + (j + 1 < t->ndim ? "," : "");
      s += ") vs (";
      i = 0;
      for (long long w : want) { s += (w < 0 ? std::string("*") : std::to_string(w)) + (++i < (int)want.size() ? "," : ""); }
      fail(s + ")");
    }
    return t;
  }
  bool lora(const std::string& base) const { return own->find(base + ".lora_layer.down.weight") != nullptr; }
  // [rows][cols...] fp32, LoRA folded; *identity: the source tensor when the values are its own (pack cache key), else null
  std::vector<float> matrix(const std::string& base, const es_tensor** identity = nullptr) const {
    const es_tensor* w = get(base + ".weight");
    std::vector<float> v = to_f32(w);
    if (identity) *identity = w;
    if (lora(base)) {
      if (w->ndim != 2) fail("es_load_weights: LoRA on a non-Linear layer '" + base + "' (conv LoRA is never enabled by the reference, TR:278-283)");
      const es_tensor* dn = get(base + ".lora_layer.down.weight");
      const es_tensor* up = get(base + ".lora_layer.up.weight");
      const long long rows = w->shape[0], cols = w->shape[1], r = dn->shape[0];
      if (dn->ndim != 2 || up->ndim != 2 || dn->shape[1] != cols || up->shape[0] != rows || up->shape[1] != r)
        fail("es_load_weights: LoRA shapes of '" + base + "' do not fit the layer");
      const std::vector<float> d = to_f32(dn), u = to_f32(up);
      parallel_for(rows, [&](long long a, long long b) {
        std::vector<double> acc((size_t)cols);
        for (long long i = a; i < b; ++i) {
          for (long long j = 0; j < cols; ++j) acc[j] = (double)v[i * cols + j];
          for (long long k = 0; k < r; ++k) {
            const double uk = (double)u[i * r + k];
            const float* dr = d.data() + k * cols;
            for (long long j = 0; j < cols; ++j) acc[j] += uk * (double)dr[j];
          }
          for (long long j = 0; j < cols; ++j) v[i * cols + j] = (float)acc[j];
        }
      });
      if (identity) *identity = nullptr;
    }
    return v;
  }
  std::vector<float> vec(const std::string& key, long long n, const es_tensor** identity = nullptr) const {
    const es_tensor* t = shaped(key, {n});
    if (identity) *identity = t;
    return to_f32(t);
  }
};

// ---- memory: one arena laid out before it exists --------------------------------------------------------------------------
struct Heap {
  unsigned long long top = 0;
  std::map<unsigned long long, unsigned long long> free_;     // offset -> bytes
  unsigned long long alloc(unsigned long long n) {
    n = (n + 255) & ~255ull;
    auto best = free_.end();
    for (auto it = free_.begin(); it != free_.end(); ++it)
      if (it->second >= n && (best == free_.end() || it->second < best->second)) best = it;
    if (best != free_.end()) {
      const unsigned long long off = best->first, sz = best->second;
      free_.erase(best);
      if (sz > n) free_[off + n] = sz - n;
      return off;
    }
    if (!free_.empty()) {                                     // grow the block that touches the top
      auto last = std::prev(free_.end());
      if (last->first + last->second == top) {
        const unsigned long long off = last->first;
        top = off + n;
        free_.erase(last);
        return off;
      }
    }
    const unsigned long long off = top;
    top += n;
    return off;
  }
  void release(unsigned long long off, unsigned long long n) {
    n = (n + 255) & ~255ull;
    auto it = free_.emplace(off, n).first;
    auto nx = std::next(it);
    if (nx != free_.end() && it->first + it->second == nx->first) { it->second += nx->second; free_.erase(nx); }
    if (it != free_.begin()) {
      auto pv = std::prev(it);
      if (pv->first + pv->second == it->first) { pv->second += it->second; free_.erase(it); }
    }
  }
};
struct Blk {
  Heap* heap; unsigned long long off, bytes;
  ~Blk() { if (heap) heap->release(off, bytes); }
};

// a view of device memory: NHWC (tokens: h = tokens, w = 1), `ld` elements from one pixel to the next
struct T {
  unsigned long long p = 0;
  int n = 0, h = 0, w = 0, c = 0;
  long long ld = 0;
  int es = 2;
  std::shared_ptr<Blk> b;
  // wide residual stream (ops.WIDE_STREAM): the low part of a stream tensor (same shape, same views) rides on the tensor
  unsigned long long lo = 0;
  std::shared_ptr<Blk> lo_b;
  // GroupNorm statistics handed over by the producing GEMM (ops.GN_HANDOVER): fp32 [n][2 * hw/64][gnp_groups][2] of the WHOLE tensor
  unsigned long long gnp = 0;
  std::shared_ptr<Blk> gnp_b;
  int gnp_groups = 0;
  explicit operator bool() const { return p != 0; }
  long long hw() const { return (long long)h * w; }
  long long numel() const { return (long long)n * h * w * c; }
  bool contig() const { return ld == c; }
  long long bstride() const { return hw() * ld; }
  size_t bytes() const { return (size_t)numel() * es; }
  T batch(int a, int cnt = -1) const {
    T t = *this;
    t.p += (unsigned long long)a * (unsigned long long)bstride() * es;
    if (t.lo) t.lo += (unsigned long long)a * (unsigned long long)bstride() * es;
    t.gnp = 0; t.gnp_b.reset(); t.gnp_groups = 0;         // (a batch slice is read as an addend / operand, never normalised as it stands)
    t.n = cnt < 0 ? n - a : cnt;
    if (a < 0 || t.n < 0 || a + t.n > n) fail("builder: batch slice out of range");
    return t;
  }
  T chan(int c0, int cn) const {
    T t = *this;
    if (c0 < 0 || c0 + cn > c) fail("builder: channel slice out of range");
    t.p += (unsigned long long)c0 * es;
    t.lo = 0; t.lo_b.reset();                            // (a channel slice of a stream tensor is an operand, never a residual)
    t.gnp = 0; t.gnp_b.reset(); t.gnp_groups = 0;
    t.c = cn;
    return t;
  }
  T view(int n2, int h2, int w2, int c2) const {
    if (!contig() || (long long)n2 * h2 * w2 * c2 != numel()) fail("builder: view of a non-contiguous or differently sized tensor");
    T t = *this;
    t.n = n2; t.h = h2; t.w = w2; t.c = c2; t.ld = c2;
    return t;
  }
  void* ptr() const { return (void*)p; }
};

struct PW {                       // ops.PackedWeight
  unsigned long long w = 0, bias = 0, ln_colsum = 0;
  int cout = 0, cin = 0, ksize = 1, bn = 128, rows_padded = 0, kpad = 0, ctail = 0, korder = 0;
  bool geglu = false;
  float ln_eps = 1e-5f;
  const uint16_t* host_w = nullptr;      // the staged copy, until the upload
  long long w_numel() const { return (long long)rows_padded * kpad; }
};
using PWs = std::vector<const PW*>;
struct Norm { unsigned long long g = 0, b = 0; };

struct CA {                       // keyword arguments of ops.conv_gemm
  int stride = 1, pad = -1;
  bool upsample = false;
  T x2, temb, residual, out, out_scale_dev;
  long long temb_stride = 0;
  int act = ES_ACT_NONE;
  float out_scale = 1.f;
  int out_h = 0, out_w = 0;
  std::vector<int> group_n;
  std::vector<T> tails;
  int x_rep = 1;
  bool wide = false;              // this launch adds into the residual stream (ops.conv_gemm(wide=True))
  int gn_groups = 0;              // > 0: the output is read by a GroupNorm over that many groups - hand the statistics over
};

// ---- the launch planner of edgestyle_amd/ops.py (plan_gemm, xs_eligible): identical decisions are what makes a natively built
// context bit-identical to a Python-built one; tests/test_host_cpu.py sweeps both over the shapes of the model ---------------
#pragma clang fp contract(off)
constexpr double PLAN_T160 = 1.25, PLAN_ALONE = 0.9, PLAN_TFIX = 2.0, PLAN_RED_FIX = 12.0, PLAN_SLAB_BYTES_PER_UNIT = 4.0e6;
constexpr double PLAN_T320 = 2.3, PLAN_T320_FIX = 3.0;
// the 256 x 256 tile of the LayerNorm-folded / GEGLU linear layers (tools/ln256_bench.py, round 5)
constexpr double PLAN_T256 = 2.6, PLAN_T256_FIX = 6.0, PLAN_T256_GEGLU = 7.0, PLAN_LN_TK = 1.55, PLAN_T256_MARGIN = 0.93;
constexpr int PLAN_BIG_MIN_M = 2048, PLAN_BIG_MIN_NK = 16;
constexpr double PLAN_T64_ALONE = 0.5, PLAN_T64 = 0.16, PLAN_T64_LONG = 1.5;
constexpr int PLAN_SMALL_MAX_M = 16384, PLAN_MIN_SLICE = 12, PLAN_NK_NOSPLIT = 10, PLAN_RESIDENT = 512;

// Tool knobs (tools/ab_env_bench.sh: A/Bs of the planner's constants INSIDE the pipeline - the constants were fitted on launches timed
// in isolation, and the step runs at the package power limit where the ranking can differ): ES_PLAN_T320 (cost of a K-step of the 256 x 320
// tile, default 2.3), ES_PLAN_BIG_MIN_NK (fewest K-steps it is offered for, 16), ES_PLAN_T256_MARGIN (0.93).  Unset: the fitted values.
double plan_env(const char* name, double dflt) { const char* e = getenv(name); return e && *e ? atof(e) : dflt; }
bool plan_gemm(long long M, int rows_padded, int kpad, bool geglu, const int* bns, int nb, bool allow_split, int* o_bn, int* o_sk, int* o_st) {
  static const double PLAN_T320 = plan_env("ES_PLAN_T320", ::PLAN_T320), PLAN_T256_MARGIN = plan_env("ES_PLAN_T256_MARGIN", ::PLAN_T256_MARGIN);
  static const int PLAN_BIG_MIN_NK = (int)plan_env("ES_PLAN_BIG_MIN_NK", ::PLAN_BIG_MIN_NK);
  // The 256 x 256 phase-interleaved tile (round 5) is offered by the callers for the LayerNorm-folded / GEGLU linear layers whose N is a
  // multiple of 256.  The choice among the OTHER tiles is made first, exactly as before; the 256-wide tile then replaces it where a
  // model fitted on those layers (tools/ln256_bench.py, profiles/r05_ln256_bench.txt: a round of 512 workgroups of the 128-wide tile
  // takes nk x 1.55 + 6 units with the fold's statistics, a round of 256 workgroups of the 256 x 256 tile nk x 2.6 + 6, + 7 with the
  // GEGLU epilogue; WHOLE rounds - its last part-filled round costs a full tile time) says it is at least 7 % faster.
  int others[8], no = 0;
  bool has256 = false;
  for (int bi = 0; bi < nb; ++bi) { if (bns[bi] == 256) has256 = true; else if (no < 8) others[no++] = bns[bi]; }
  if (has256) {
    const int nk = kpad / BK;
    int obn = 0, osk = 1, ost = 2;
    const bool have_other = no > 0 && plan_gemm(M, rows_padded, kpad, geglu, others, no, allow_split, &obn, &osk, &ost);
    if (rows_padded % 256) { if (!have_other) return false; *o_bn = obn; *o_sk = osk; *o_st = ost; return true; }
    const long long t256 = ((M + 255) / 256) * (rows_padded / 256);
    const double c256 = (double)((t256 + 255) / 256) * ((double)nk * PLAN_T256 + PLAN_T256_FIX + (geglu ? PLAN_T256_GEGLU : 0.0));
    bool take = !have_other;
    if (have_other && M >= PLAN_BIG_MIN_M && nk >= PLAN_BIG_MIN_NK && osk == 1 && obn != 64) {
      const long long to = ((M + BM - 1) / BM) * (rows_padded / obn);
      const double co = (double)((to + PLAN_RESIDENT - 1) / PLAN_RESIDENT) * ((double)nk * PLAN_LN_TK * (obn == 160 ? PLAN_T160 : 1.0) + PLAN_T256_FIX);
      take = c256 < PLAN_T256_MARGIN * co;
    }
    if (take) { *o_bn = 256; *o_sk = 1; *o_st = 2; } else { *o_bn = obn; *o_sk = osk; *o_st = ost; }
    return true;
  }
  if (geglu) { *o_bn = 128; *o_sk = 1; *o_st = 2; return true; }
  const int nk = kpad / BK;
  bool have = false;
  double bt = 0; int bbn = 0, bsk = 0; long long bwgs = 0;
  for (int bi = 0; bi < nb; ++bi) {
    const int bn = bns[bi];
    if (rows_padded % bn) continue;
    if (bn == 320 && (M < PLAN_BIG_MIN_M || nk < PLAN_BIG_MIN_NK) && nb > 1) continue;
    if (bn == 64 && M > PLAN_SMALL_MAX_M && nb > 1) continue;
    const int bm = bn == 320 ? 256 : (bn == 64 ? 64 : BM);
    const int resident = bn == 320 ? PLAN_RESIDENT / 2 : (bn == 64 ? 2 * PLAN_RESIDENT : PLAN_RESIDENT);
    const long long tiles = ((M + bm - 1) / bm) * (rows_padded / bn);
    int last = 1;
    if (allow_split && tiles < resident / 2 && nk > PLAN_NK_NOSPLIT) last = std::max(1, std::min(nk / PLAN_MIN_SLICE, 32));
    for (int sk = 1; sk <= last; ++sk) {
      const long long wgs = tiles * sk;
      double tk;
      if (bn == 320) tk = PLAN_T320;
      else if (bn == 64) {
        if (sk > 1 && wgs > PLAN_RESIDENT) continue;
        const double w = (double)wgs / (double)(PLAN_RESIDENT / 2);
        const double ex = std::max(0.0, w - 1.0);
        tk = (PLAN_T64_ALONE + PLAN_T64 * (ex * ex)) * (nk <= 40 ? 1.0 : PLAN_T64_LONG);
      } else tk = (bn == 160 ? PLAN_T160 : 1.0) * (wgs <= PLAN_RESIDENT / 2 ? PLAN_ALONE : 1.0);
      // (the 256 x 320 tile costs its fractional number of rounds, no less than 0.9: ops.plan_gemm_reference)
      const double rounds = bn == 320 ? std::max((double)wgs / 256.0, 0.9) : (double)((wgs + resident - 1) / resident);
      double t = rounds * (((double)nk / (double)sk) * tk + (bn == 320 ? PLAN_T320_FIX : PLAN_TFIX));
      if (sk > 1) t += PLAN_RED_FIX + (double)sk * (double)M * (double)rows_padded * 8.0 / PLAN_SLAB_BYTES_PER_UNIT;
      if (!have || t < bt) { have = true; bt = t; bbn = bn; bsk = sk; bwgs = wgs; }
    }
  }
  if (!have) return false;
  *o_bn = bbn; *o_sk = bsk;
  if (bbn == 64) *o_st = (bwgs <= PLAN_RESIDENT && nk / bsk >= 6) ? 4 : 2;
  else *o_st = (bbn != 320 && bwgs <= PLAN_RESIDENT / 2 && nk / bsk >= 6) ? 4 : 2;
  return true;
}

constexpr int XS_TARGET_WGS = 256, XS_MIN_M = 8192;
bool xs_shape_fits(int ksize, int kpad, int cin, int ctail, int cout, bool geglu) {      // what the kernel can run at all
  if (ksize != 1 || (kpad != 320 && kpad != 640) || cin != kpad || ctail) return false;
  const int ch = kpad == 320 ? 64 : 32;
  const int line = ch * (128 / (geglu ? ch : 2 * ch));
  return !(cout % line || cout < 4 * line);
}
bool xs_shape_ok(long long M, int ksize, int kpad, int cin, int ctail, int cout, bool geglu) {       // ... and where it wins
  if (!xs_shape_fits(ksize, kpad, cin, ctail, cout, geglu)) return false;
  if (M < XS_MIN_M || (M < 4 * XS_MIN_M && cout < (kpad == 320 ? 960 : 1920))) return false;
  return true;
}
bool gn_fold_on() { static const bool on = [] { const char* e = getenv("ES_GN_FOLD"); return !e || std::string(e) == "1"; }(); return on; }
int choose_bn(int cout) { return (cout % 160 == 0 && cout % 128 != 0) ? 160 : 128; }
bool big_tile_256() { static const bool on = [] { const char* e = getenv("ES_BIG_TILE_256"); return !e || e[0] != '0'; }(); return on; }

// ---- the builder ------------------------------------------------------------------------------------------------------------
struct Builder {
  Heap heap;
  unsigned long long ws_bytes = 0;
  int dt = ES_F16;
  std::vector<std::pair<unsigned long long, std::vector<char>>> uploads;     // (fake address, bytes) of everything persistent with contents
  std::deque<PW> pws;
  std::map<std::string, const PW*> pack_cache;
  std::map<const void*, unsigned long long> vec_cache;
  std::map<std::pair<int, int>, unsigned long long> gn_partials, fusion_scratch;
  std::map<int, unsigned long long> zero_bias;

  // -- allocation
  T empty(int n, int h, int w, int c, int es = 2) {
    T t; t.n = n; t.h = h; t.w = w; t.c = c; t.ld = c; t.es = es;
    auto blk = std::make_shared<Blk>();
    blk->heap = &heap; blk->bytes = std::max<size_t>(t.bytes(), 1); blk->off = heap.alloc(blk->bytes);
    t.b = blk; t.p = FAKE_HEAP + blk->off;
    return t;
  }
  unsigned long long persistent(size_t bytes) { return FAKE_HEAP + heap.alloc(std::max<size_t>(bytes, 1)); }   // zero-filled, never freed
  T persistent_t(int n, int h, int w, int c, int es = 2) {
    T t; t.n = n; t.h = h; t.w = w; t.c = c; t.ld = c; t.es = es;
    t.p = persistent(t.bytes());
    return t;
  }
  char* staged(unsigned long long addr, size_t bytes) {
    uploads.emplace_back(addr, std::vector<char>(bytes, 0));
    return uploads.back().second.data();
  }
  unsigned long long upload_f32(const std::vector<float>& v) {
    const unsigned long long a = persistent(v.size() * 4);
    memcpy(staged(a, v.size() * 4), v.data(), v.size() * 4);
    return a;
  }
  unsigned long long workspace(unsigned long long bytes) { ws_bytes = std::max(ws_bytes, bytes); return FAKE_WS; }
  void add_into(const T& a, const T& b, const T& out) {     // ops.add(a, b, out=out)
    if (!a.contig() || !b.contig() || !out.contig() || a.numel() != b.numel() || a.numel() != out.numel()) fail("builder: add of non-contiguous or differently sized tensors");
    ok(es_add(a.ptr(), b.ptr(), out.ptr(), a.numel(), dt, nullptr), "es_add");
  }
  T add(const T& a, const T& b) {                          // ops.add
    if (!a.contig() || !b.contig() || a.numel() != b.numel()) fail("builder: add of non-contiguous or differently sized tensors");
    T out = empty(a.n, a.h, a.w, a.c);
    ok(es_add(a.ptr(), b.ptr(), out.ptr(), a.numel(), dt, nullptr), "es_add");
    return out;
  }
  // ops.gn_handover: a rule of the shape alone - where the stand-alone GroupNorm is the two-launch form
  static bool gn_handover(long long hw, int c, int groups) {
    static const std::string mode = [] { const char* e = getenv("ES_GN_HANDOVER"); return std::string(e ? e : "0"); }();      // off by default: measured a net loss (ops.py)
    if ((mode != "1" && mode != "all") || c % 8 || hw % 64 || c % groups || c / groups > 64) return false;
    return mode == "all" || !es_group_norm_is_slab((int)hw, c, groups);
  }
  static bool gn_handover_on() { const char* e = getenv("ES_GN_HANDOVER"); const std::string m(e ? e : "0"); return m == "1" || m == "all"; }
  // ops.wide_stream: "auto" (default) = bf16 pipelines only
  bool wide_stream() const {
    static const std::string mode = [] { const char* e = getenv("ES_WIDE_STREAM"); return std::string(e ? e : "auto"); }();
    return mode == "1" || (mode == "auto" && dt == ES_BF16);
  }
  uint16_t enc(float f) const { return dt == ES_F16 ? f32_to_f16(f) : f32_to_bf16(f); }
  float dec(uint16_t u) const { return dt == ES_F16 ? f16_to_f32(u) : bf16_to_f32(u); }

  // -- weight packing (ops.pack_weight*): [rows_padded][Kpad] K-contiguous, K = (ky, kx, c) tap-major or (c / 64, ky, kx, c % 64) chunk-major
  // w: [cout][cin][k][k] fp32 (linear: k = 1)
  const PW* pack(const std::vector<float>& w, int cout, int cin, int k, const std::vector<float>* bias, bool geglu = false, int cin_pad = 0, int cout_pad = 0,
                 const std::vector<float>* tail = nullptr, int ctail = 0) {
    if ((long long)w.size() != (long long)cout * cin * k * k) fail("builder: weight size does not match its shape");
    const int cp = cin_pad ? cin_pad : round_up(cin, 8);
    if (cp < cin || cp % 8 || (cout_pad && cout_pad < cout)) fail("builder: channel padding smaller than the tensor");
    const int ktrue = k * k * cp + ctail;
    const int cout_eff = cout_pad ? cout_pad : cout;
    const int bn = geglu ? 128 : choose_bn(cout_eff);
    const int rows = round_up(cout_eff, bn);
    const int kpad = round_up(ktrue, BK);
    const int inner = cout / 2;
    if (geglu && inner % 16) fail("builder: GEGLU inner width must be a multiple of 16");
    pws.emplace_back();
    PW& p = pws.back();
    p.cout = cout_eff; p.cin = cp; p.ksize = k; p.bn = bn; p.rows_padded = rows; p.kpad = kpad; p.geglu = geglu; p.ctail = ctail;
    // K order (ops.choose_korder): with ES_CHUNK_MAJOR=1 every 3x3 convolution over 64-aligned channels is packed chunk-major,
    // k = (c / 64, tap, c % 64); the default is tap-major (the chunk-major order measured 5-10 % slower, ops.py)
    static const bool chunk_major_on = [] { const char* e = getenv("ES_CHUNK_MAJOR"); return e && e[0] == '1' && !e[1]; }();
    const bool chunk_major = chunk_major_on && k == 3 && cp % BK == 0;
    p.korder = chunk_major ? 1 : 0;
    p.w = persistent((size_t)rows * kpad * 2);
    uint16_t* dst = (uint16_t*)staged(p.w, (size_t)rows * kpad * 2);
    p.host_w = dst;
    auto src_row = [&](int r) { if (!geglu) return r; const int blk = r / 32, within = r % 32; return within < 16 ? blk * 16 + within : inner + blk * 16 + (within - 16); };
    parallel_for(cout, [&](long long a, long long b) {
      for (long long r = a; r < b; ++r) {
        const float* s = w.data() + (size_t)src_row((int)r) * cin * k * k;
        uint16_t* d = dst + (size_t)r * kpad;
        for (int t = 0; t < k * k; ++t)
          for (int ci = 0; ci < cin; ++ci)
            d[chunk_major ? ((ci / BK) * k * k + t) * BK + ci % BK : t * cp + ci] = enc(s[(size_t)ci * k * k + t]);
        if (tail) for (int j = 0; j < ctail; ++j) d[k * k * cp + j] = enc((*tail)[(size_t)r * ctail + j]);
      }
    });
    if (bias) {
      if ((int)bias->size() != cout) fail("builder: bias size does not match the weight");
      p.bias = persistent((size_t)rows * 4);
      float* bd = (float*)staged(p.bias, (size_t)rows * 4);
      for (int r = 0; r < cout; ++r) bd[r] = (*bias)[src_row(r)];
    }
    return &p;
  }
  // Linear(LayerNorm(x)) as one launch on the raw x (ops.pack_weight_ln): w [rows][C]
  const PW* pack_ln(const std::vector<float>& w, int rows, int C, const std::vector<float>* bias, const std::vector<float>& gamma, const std::vector<float>& beta,
                    float eps, bool geglu) {
    std::vector<float> wg((size_t)rows * C), fb((size_t)rows);
    parallel_for(rows, [&](long long a, long long b) {
      for (long long r = a; r < b; ++r) {
        double acc = 0.0;
        for (int j = 0; j < C; ++j) { wg[r * C + j] = w[r * C + j] * gamma[j]; acc += (double)w[r * C + j] * (double)beta[j]; }
        if (bias) acc += (double)(*bias)[r];
        fb[r] = (float)acc;
      }
    });
    PW* p = const_cast<PW*>(pack(wg, rows, C, 1, &fb, geglu));
    // column sums of the ROUNDED weights the MFMAs multiply, in the packed row order
    const uint16_t* pk = p->host_w;
    std::vector<float> cs((size_t)p->rows_padded);
    for (int r = 0; r < p->rows_padded; ++r) {
      double acc = 0.0;
      for (int j = 0; j < p->kpad; ++j) acc += (double)dec(pk[(size_t)r * p->kpad + j]);
      cs[r] = (float)acc;
    }
    p->ln_colsum = upload_f32(cs);
    p->ln_eps = eps;
    return p;
  }
  unsigned long long norm_vec(const Weights& W, const std::string& key, long long n) {
    const es_tensor* id = nullptr;
    std::vector<float> v = W.vec(key, n, &id);
    auto it = vec_cache.find(id->data);
    if (it != vec_cache.end()) return it->second;
    return vec_cache[id->data] = upload_f32(v);
  }
  Norm norm(const Weights& W, const std::string& p, long long n) { return {norm_vec(W, p + ".weight", n), norm_vec(W, p + ".bias", n)}; }

  // conv / Linear `p` (.weight [+ .bias]); packed once per SOURCE tensor (ControlLoRA nets alias the UNet's convolutions)
  const PW* conv(const Weights& W, const std::string& p, int cin_pad = 0, bool geglu = false, int cout_pad = 0) {
    const es_tensor* wt = W.get(p + ".weight");
    if (wt->ndim != 2 && wt->ndim != 4) fail("es_load_weights: '" + p + ".weight' is neither a Linear nor a Conv2d weight");
    const int cout = (int)wt->shape[0], cin = (int)wt->shape[1], k = wt->ndim == 4 ? (int)wt->shape[2] : 1;
    if (wt->ndim == 4 && (wt->shape[3] != k || (k != 1 && k != 3))) fail("es_load_weights: '" + p + ".weight': only 1x1 and 3x3 kernels");
    const es_tensor* bt = W.find(p + ".bias");
    std::string key;
    if (!W.lora(p)) {
      char buf[128];
      snprintf(buf, sizeof(buf), "%p|%p|%d|%d|%d", wt->data, bt ? bt->data : nullptr, (int)geglu, cin_pad, cout_pad);
      key = buf;
      auto it = pack_cache.find(key);
      if (it != pack_cache.end()) return it->second;
    }
    std::vector<float> w = W.matrix(p), b;
    if (bt) b = W.vec(p + ".bias", cout);
    const PW* pw = pack(w, cout, cin, k, bt ? &b : nullptr, geglu, cin_pad, cout_pad);
    if (!key.empty()) pack_cache[key] = pw;
    return pw;
  }
  // several Linear layers stacked along the output dimension (QKV, KV, all time_emb_proj of a net)
  void stack(const Weights& W, const std::vector<std::string>& ps, bool bias, std::vector<float>& w, std::vector<float>& b, int& rows, int& cols) {
    rows = 0; cols = -1;
    for (const auto& p : ps) {
      const es_tensor* t = W.get(p + ".weight");
      if (t->ndim != 2 || (cols >= 0 && t->shape[1] != cols)) fail("es_load_weights: '" + p + ".weight' does not stack with its neighbours");
      cols = (int)t->shape[1];
      std::vector<float> m = W.matrix(p);
      w.insert(w.end(), m.begin(), m.end());
      if (bias) { std::vector<float> v = W.vec(p + ".bias", t->shape[0]); b.insert(b.end(), v.begin(), v.end()); }
      rows += (int)t->shape[0];
    }
  }
  const PW* cat(const Weights& W, const std::vector<std::string>& ps, bool bias) {
    std::vector<float> w, b; int rows, cols;
    stack(W, ps, bias, w, b, rows, cols);
    return pack(w, rows, cols, 1, bias ? &b : nullptr);
  }
  const PW* cat_ln(const Weights& W, const std::vector<std::string>& ps, bool bias, const std::string& ln, bool geglu = false) {
    std::vector<float> w, b; int rows, cols;
    stack(W, ps, bias, w, b, rows, cols);
    return pack_ln(w, rows, cols, bias ? &b : nullptr, W.vec(ln + ".weight", cols), W.vec(ln + ".bias", cols), 1e-5f, geglu);
  }
  // conv `p` with the 1x1 conv `pt` appended along K (ops.pack_weight_tail)
  const PW* conv_tail(const Weights& W, const std::string& p, const std::string& pt) {
    const es_tensor* wt = W.get(p + ".weight");
    const es_tensor* tt = W.get(pt + ".weight");
    char buf[96];
    snprintf(buf, sizeof(buf), "%p|%p|tail", wt->data, tt->data);
    auto it = pack_cache.find(buf);
    if (it != pack_cache.end()) return it->second;
    const int cout = (int)wt->shape[0], cin = (int)wt->shape[1], k = (int)wt->shape[2], ct = (int)tt->shape[1];
    if (wt->ndim != 4 || cin % BK || ct % BK || tt->shape[0] != cout) fail("es_load_weights: conv2 + conv_shortcut fold needs 64-aligned channels ('" + p + "')");
    std::vector<float> w = W.matrix(p), tw = W.matrix(pt), b = W.vec(p + ".bias", cout), b2 = W.vec(pt + ".bias", cout);
    for (int i = 0; i < cout; ++i) b[i] = b[i] + b2[i];
    const PW* pw = pack(w, cout, cin, k, &b, false, 0, 0, &tw, ct);
    pack_cache[buf] = pw;
    return pw;
  }

  // -- ops (edgestyle_amd/ops.py): descriptors handed to the C ABI, which validates and records them --------------------------
  static void ok(int rc, const char* what) { if (rc) fail(std::string(what) + ": " + es_last_error()); }

  struct GnFront { unsigned long long part = 0; std::vector<Norm> nl; int groups = 0, nchunk = 0, hw = 0; float eps = 0.f; };
  T linear_xs(const T& x, const PWs& pl, long long M, const T& out, const std::vector<long long>& group_rows, const T& residual = T(),
              const GnFront* gn = nullptr) {
    const PW* pw = pl[0];
    const int K = pw->kpad, ch = K == 320 ? 64 : 32;
    const int pline = 128 / (pw->geglu ? ch : 2 * ch);
    const int total = pw->cout / ch, lines = total / pline;
    const long long rbs = (M + 255) / 256;
    const int want = (int)std::max<long long>(1, std::min<long long>(lines, XS_TARGET_WGS / rbs));
    const int lps = cdiv(lines, want), nslices = cdiv(lines, lps);
    es_xs_desc d;
    memset(&d, 0, sizeof(d));
    auto bias_of = [&](const PW* q) -> unsigned long long {
      if (q->bias) return q->bias;
      auto it = zero_bias.find(q->rows_padded);
      if (it != zero_bias.end()) return it->second;
      return zero_bias[q->rows_padded] = persistent((size_t)q->rows_padded * 4);
    };
    d.x = x.ptr(); d.out = out.ptr(); d.w = (const void*)pw->w; d.bias = (const float*)bias_of(pw);
    d.M = (int)M; d.K = K; d.Cout = pw->cout; d.rows_padded = pw->rows_padded; d.ldo = out.c;
    d.geglu = pw->geglu; d.ln = pw->ln_colsum != 0; d.ln_eps = pw->ln_eps;
    d.nslices = nslices; d.chunks_per_slice = lps * pline; d.dtype = dt;
    if (residual) d.residual = residual.ptr();
    if (gn) {
      d.gn_part = (const float*)gn->part; d.gn_groups = gn->groups; d.gn_nchunk = gn->nchunk; d.gn_hw = gn->hw; d.gn_eps = gn->eps;
      if (pl.size() > 1) {
        if (gn->nl.size() != pl.size()) fail("grouped linear_xs: one GroupNorm parameter set per weight group");
        for (size_t g = 0; g < pl.size(); ++g) { d.gn_gamma_g[g] = (const float*)gn->nl[g].g; d.gn_beta_g[g] = (const float*)gn->nl[g].b; }
      } else { d.gn_gamma = (const float*)gn->nl[0].g; d.gn_beta = (const float*)gn->nl[0].b; }
    }
    if (pl.size() > 1) {
      d.ngroups = (int)pl.size();
      long long acc = 0;
      for (size_t g = 0; g < pl.size(); ++g) {
        const PW* q = pl[g];
        if (q->rows_padded != pw->rows_padded || q->kpad != pw->kpad || q->cout != pw->cout || q->geglu != pw->geglu) fail("grouped linear_xs: weight geometry differs between groups");
        acc += group_rows[g] / 128;
        d.mt_end[g] = (int)acc; d.w_g[g] = (const void*)q->w; d.bias_g[g] = (const float*)bias_of(q);
      }
    }
    ok(es_linear_xs(&d, nullptr), "es_linear_xs");
    return out;
  }

  T conv_gemm(const T& x, const PWs& pl, const CA& a = CA()) {
    const PW* pw = pl[0];
    const bool grouped = pl.size() > 1;
    int N = x.n;
    const int H = x.h, Wd = x.w, C1 = x.c, nsrc = x.n;
    if (!x.contig()) fail("conv_gemm: the source must be contiguous");
    if (a.x_rep > 1) { if (a.x2 || !a.tails.empty()) fail("conv_gemm: x_rep needs a single source"); N *= a.x_rep; }
    const int C2 = a.x2 ? a.x2.c : 0;
    if (C1 + C2 != pw->cin) fail("conv_gemm: input channels " + std::to_string(C1) + "+" + std::to_string(C2) + " != packed " + std::to_string(pw->cin));
    int tsum = 0;
    for (const auto& t : a.tails) tsum += t.c;
    if (tsum != pw->ctail || a.tails.size() > 2) fail("conv_gemm: tail channels do not match the packed weights");
    const int k = pw->ksize, pad = a.pad < 0 ? (k == 3 ? 1 : 0) : a.pad;
    const int Hin = a.upsample ? H * 2 : H, Win = a.upsample ? Wd * 2 : Wd;
    const int Hout = a.out_h ? a.out_h : (Hin + 2 * pad - k) / a.stride + 1, Wout = a.out_w ? a.out_w : (Win + 2 * pad - k) / a.stride + 1;
    const int act_i = pw->geglu ? ES_ACT_GEGLU : a.act;
    const int cstore = pw->geglu ? pw->cout / 2 : pw->cout;
    T out = a.out ? a.out : empty(N, Hout, Wout, cstore);
    if (out.numel() != (long long)N * Hout * Wout * cstore || !out.contig()) fail("conv_gemm: output buffer of another size");
    const long long M = (long long)N * Hout * Wout, hw = (long long)Hout * Wout;
    // (a residual rides on es_linear_xs at K = 320, unless the launch carries the two-word residual stream: ops.conv_gemm)
    static const bool xs_residual = [] { const char* e = getenv("ES_XS_RESIDUAL"); return !e || std::string(e) == "1"; }();
    const bool xs_res = !a.residual || (xs_residual && pw->kpad == 320 && !pw->geglu && !pw->ln_colsum && a.residual.contig() &&
                                        a.residual.numel() == (long long)N * Hout * Wout * cstore && !(a.wide && wide_stream()));
    const bool plain = k == 1 && a.stride == 1 && !a.upsample && !a.x2 && !a.temb && xs_res && a.tails.empty() && a.x_rep == 1 &&
                       a.act == ES_ACT_NONE && a.out_scale == 1.f && !a.out_scale_dev;
    if (plain && xs_shape_ok(M, pw->ksize, pw->kpad, pw->cin, pw->ctail, pw->cout, pw->geglu)) {
      bool e = true;
      if (grouped) {
        if (pl.size() > 4) e = false;
        for (int n : a.group_n) if ((n * hw) % 256) e = false;
        for (const PW* q : pl) if ((q->ln_colsum == 0) != (pw->ln_colsum == 0)) e = false;
      }
      if (e) {
        std::vector<long long> rows;
        for (int n : a.group_n) rows.push_back(n * hw);
        T xo = out;
        linear_xs(x, grouped ? pl : PWs{pw}, M, xo, rows, a.residual);
        return out;
      }
    }
    bool big_ok = C1 % BK == 0 && C2 % BK == 0 && !pw->geglu && !pw->ln_colsum;
    if (big_ok && grouped) for (int n : a.group_n) if ((n * hw) % 256) big_ok = false;
    // the 256 x 256 phase-interleaved tile: LayerNorm-folded and GEGLU linear layers whose N is a multiple of 256 (ops.conv_gemm: the same rule)
    bool ln256_ok = big_tile_256() && (pw->geglu || pw->ln_colsum) && k == 1 && a.stride == 1 && !a.upsample && !a.x2 && a.tails.empty() && !a.temb &&
                    a.x_rep == 1 && C1 % BK == 0 && pw->rows_padded % 256 == 0 && pw->cout % 8 == 0 && !(pw->geglu && a.residual) && !a.wide && a.gn_groups == 0;
    if (ln256_ok && grouped) for (int n : a.group_n) if ((n * hw) % 256) ln256_ok = false;
    const bool small_ok = C1 % BK == 0 && C2 % BK == 0 && !pw->geglu;
    int cand[5], nc = 0;
    if (ln256_ok) cand[nc++] = 256;
    if (big_ok) cand[nc++] = 320;
    cand[nc++] = 160; cand[nc++] = 128;
    if (small_ok) cand[nc++] = 64;
    int bn, splitk, stages;
    if (!plan_gemm(M, pw->rows_padded, pw->kpad, pw->geglu, cand, nc, pw->ln_colsum == 0, &bn, &splitk, &stages)) fail("plan_gemm: rows_padded fits no N tile");
    if (bn == 320 && splitk == 1 && !es_conv_gemm8p_form_ok(act_i, pw->cout, a.temb ? 1 : 0, hw, a.residual ? 1 : 0)) {
      // the 256 x 320 tile does not implement this epilogue form: plan again without it (ops.conv_gemm does the same)
      if (!plan_gemm(M, pw->rows_padded, pw->kpad, pw->geglu, cand + 1, nc - 1, pw->ln_colsum == 0, &bn, &splitk, &stages)) fail("plan_gemm: rows_padded fits no N tile");
    }
    es_gemm_desc d;
    memset(&d, 0, sizeof(d));
    d.x = x.ptr(); d.x2 = a.x2 ? a.x2.ptr() : nullptr; d.w = (const void*)pw->w; d.bias = (const float*)pw->bias;
    d.temb = a.temb ? a.temb.ptr() : nullptr; d.residual = a.residual ? a.residual.ptr() : nullptr;
    d.out_scale_dev = a.out_scale_dev ? (const float*)a.out_scale_dev.ptr() : nullptr; d.out = out.ptr();
    d.N = N; d.Hsrc = H; d.Wsrc = Wd; d.C1 = C1; d.C2 = C2; d.Hout = Hout; d.Wout = Wout; d.Cout = pw->cout;
    d.rows_padded = pw->rows_padded; d.Kpad = pw->kpad; d.ksize = k; d.stride = a.stride; d.pad = pad; d.upsample = a.upsample;
    d.temb_stride = a.temb ? (int)a.temb_stride : 0;
    d.act = act_i; d.splitk = splitk; d.bn = bn; d.dtype = dt; d.out_scale = a.out_scale; d.stages = stages;
    const long long src_numel = x.numel() * a.x_rep + (a.x2 ? a.x2.numel() : 0);
    d.xcd_m_fastest = (!grouped && splitk == 1 && M <= 2048 && pw->w_numel() > src_numel) ? 1 : 0;
    d.x_nmod = a.x_rep > 1 ? nsrc : 0;
    d.korder = pw->korder;
    if (a.gn_groups > 0 && gn_handover(hw, cstore, a.gn_groups) && !pw->geglu && !pw->ln_colsum && pw->cout / a.gn_groups <= (bn == 320 ? 160 : bn)) {
      T part = empty(N, (int)(2 * (hw / 64)), 1, a.gn_groups * 2, 4);
      d.gn_part = (float*)part.ptr(); d.gn_groups = a.gn_groups;
      out.gnp = part.p; out.gnp_b = part.b; out.gnp_groups = a.gn_groups;
    }
    if (a.wide && a.residual && wide_stream() && cstore % 8 == 0 && !pw->geglu) {
      // the sum over (residual hi + lo) in fp32, stored as hi + lo (es_gemm_desc.out_lo)
      if (a.residual.lo) d.residual_lo = (const void*)a.residual.lo;
      T lo = empty(N, Hout, Wout, cstore);
      d.out_lo = lo.ptr();
      out.lo = lo.p; out.lo_b = lo.b;
    }
    if (k == 1 && M <= 65536 && C1 % BK == 0 && C2 % BK == 0 && bn != 64 && bn != 320 && bn != 256 && !(stages == 4 && bn != 128) && stages != 3) d.waves = 8;
    if (splitk > 1) d.workspace = (float*)workspace((unsigned long long)splitk * M * pw->rows_padded * 4);
    if (pw->ln_colsum) {
      for (const PW* q : pl) if (!q->ln_colsum) fail("LayerNorm-folded weights need a plain linear launch (all groups folded)");
      if (a.x2 || k != 1) fail("LayerNorm-folded weights need a plain linear launch");
      d.ln_colsum = (const float*)pw->ln_colsum; d.ln_eps = pw->ln_eps;
    }
    if (!a.tails.empty()) {
      for (const auto& t : a.tails) if (t.n != N || t.h != Hout || t.w != Wout || !t.contig()) fail("conv_gemm: tail sources must be contiguous [N,Hout,Wout,C]");
      d.t1 = a.tails[0].ptr(); d.Ct1 = a.tails[0].c;
      if (a.tails.size() == 2) { d.t2 = a.tails[1].ptr(); d.Ct2 = a.tails[1].c; }
    }
    if (grouped) {
      const int gran = bn == 320 ? 256 : BM;
      int sum = 0;
      for (int n : a.group_n) { sum += n; if ((n * hw) % gran) fail("grouped conv_gemm: groups must cover N in whole tiles"); }
      if (pl.size() > 4 || a.group_n.size() != pl.size() || sum != N) fail("grouped conv_gemm: bad group table");
      d.ngroups = (int)pl.size();
      long long acc = 0;
      for (size_t g = 0; g < pl.size(); ++g) {
        const PW* q = pl[g];
        if (q->rows_padded != pw->rows_padded || q->kpad != pw->kpad || q->cout != pw->cout || q->cin != pw->cin || q->ksize != pw->ksize ||
            q->geglu != pw->geglu || q->ctail != pw->ctail || q->korder != pw->korder) fail("grouped conv_gemm: weight geometry differs between groups");
        acc += a.group_n[g] * hw / BM;
        d.mt_end[g] = (int)acc; d.w_g[g] = (const void*)q->w; d.bias_g[g] = (const float*)q->bias; d.ln_colsum_g[g] = (const float*)q->ln_colsum;
      }
    }
    ok(es_conv_gemm(&d, nullptr), "es_conv_gemm");
    return out;
  }
  T conv_gemm(const T& x, const PW* pw, const CA& a = CA()) { return conv_gemm(x, PWs{pw}, a); }

  // x: [..., K] contiguous -> [..., cstore]: the same kernel on a 1x1 "image" of M pixels (ops.linear)
  T linear(const T& x, const PWs& pl, CA a = CA()) {
    const PW* p0 = pl[0];
    const long long M = x.numel() / x.c;
    const int cstore = p0->geglu ? p0->cout / 2 : p0->cout, rep = a.x_rep;
    if (a.residual) a.residual = a.residual.view((int)(M * rep), 1, 1, cstore);      // (views keep the low part)
    T keep = a.out;
    if (a.out) a.out = a.out.view((int)(M * rep), 1, 1, cstore);
    T y = conv_gemm(x.view((int)M, 1, 1, x.c), pl, a);
    T r = y;
    r.n = x.n * rep; r.h = x.h; r.w = x.w; r.c = cstore; r.ld = cstore;
    return r;
  }
  T linear(const T& x, const PW* pw, const CA& a = CA()) { return linear(x, PWs{pw}, a); }

  T attention(const T& q, const T& k, const T& v, int heads) {
    const int N = q.n, Sq = (int)q.hw(), Cq = q.c, Skv = (int)k.hw(), dh = Cq / heads;
    T out = empty(N, q.h, q.w, Cq);
    es_attn_desc d;
    memset(&d, 0, sizeof(d));
    d.q = q.ptr(); d.k = k.ptr(); d.v = v.ptr(); d.o = out.ptr();
    d.N = N; d.heads = heads; d.Sq = Sq; d.Skv = Skv; d.d = dh;
    d.ldq = (int)q.ld; d.ldk = (int)k.ld; d.ldv = (int)v.ld; d.ldo = (int)out.ld;
    d.bsq = q.bstride(); d.bsk = k.bstride(); d.bsv = v.bstride(); d.bso = out.bstride();
    d.scale = (float)(1.0 / sqrt((double)dh));
    d.dtype = dt;
    ok(es_attention(&d, nullptr), "es_attention");
    return out;
  }

  T group_norm(const T& x, const std::vector<Norm>& nl, int groups, float eps, bool silu, const T& x2 = T(), const std::vector<int>& group_n = {}) {
    const int N = x.n, C1 = x.c, C2 = x2 ? x2.c : 0;
    T out = empty(N, x.h, x.w, C1 + C2);
    auto key = std::make_pair(N, groups);
    auto it = gn_partials.find(key);
    if (it == gn_partials.end()) it = gn_partials.emplace(key, persistent(es_group_norm_partials_bytes(N, groups))).first;
    es_gn_desc d;
    memset(&d, 0, sizeof(d));
    d.x = x.ptr(); d.x2 = x2 ? x2.ptr() : nullptr; d.out = out.ptr(); d.partials = (float*)it->second;
    if (nl.size() > 1) {
      int sum = 0;
      for (int n : group_n) sum += n;
      if (nl.size() > 4 || group_n.size() != nl.size() || sum != N) fail("grouped group_norm: bad group table");
      d.ngroups = (int)nl.size();
      int acc = 0;
      for (size_t g = 0; g < nl.size(); ++g) { acc += group_n[g]; d.n_end[g] = acc; d.gamma_g[g] = (const float*)nl[g].g; d.beta_g[g] = (const float*)nl[g].b; }
    } else { d.gamma = (const float*)nl[0].g; d.beta = (const float*)nl[0].b; }
    d.N = N; d.HW = (int)x.hw(); d.C1 = C1; d.C2 = C2; d.groups = groups; d.eps = eps; d.silu = silu; d.dtype = dt;
    if (x.gnp && !x2 && gn_handover(x.hw(), C1, groups) && x.gnp_groups == groups) {      // the producer's statistics: one streaming pass
      d.partials = (float*)x.gnp; d.ext_chunks = (int)(2 * (x.hw() / 64));
    }
    ok(es_group_norm(&d, nullptr), "es_group_norm");
    return out;
  }
  T group_norm(const T& x, const Norm& n, int groups, float eps, bool silu, const T& x2 = T()) { return group_norm(x, std::vector<Norm>{n}, groups, eps, silu, x2); }

  // Transformer2DModel.norm -> proj_in (ops.gn_proj_in: the same rule, the same two launches): a statistics pass and the projection on
  // es_linear_xs with the GroupNorm applied to the rows it holds in registers; elsewhere es_group_norm + es_conv_gemm as before
  bool gn_fold_ok(long long M, long long hw, int groups, const PWs& pl, const std::vector<int>& group_n) const {
    const PW* pw = pl[0];
    if (!gn_fold_on() || gn_handover_on() || hw % 256 || groups > 32) return false;
    if (pw->geglu || pw->ln_colsum || pw->kpad % groups || !xs_shape_fits(pw->ksize, pw->kpad, pw->cin, pw->ctail, pw->cout, pw->geglu)) return false;
    if (M < (pw->kpad == 320 ? 8192 : 32768)) return false;
    if (pl.size() > 1) {
      if (pl.size() > 4) return false;
      for (int n : group_n) if ((n * hw) % 256) return false;
      for (const PW* q : pl)
        if (q->rows_padded != pw->rows_padded || q->kpad != pw->kpad || q->cout != pw->cout || q->geglu != pw->geglu || q->ln_colsum) return false;
    }
    return true;
  }
  T gn_proj_in(const T& x, const std::vector<Norm>& nl, int groups, float eps, const PWs& pl, const std::vector<int>& group_n = {}) {
    const int N = x.n;
    const long long hw = x.hw(), M = (long long)N * hw;
    const bool grouped = pl.size() > 1;
    if (!(x.contig() && gn_fold_ok(M, hw, groups, pl, group_n))) {
      CA g; g.group_n = group_n;
      return conv_gemm(group_norm(x, nl, groups, eps, false, T(), group_n), pl, g);
    }
    auto key = std::make_pair(N, groups);
    auto it = gn_partials.find(key);
    if (it == gn_partials.end()) it = gn_partials.emplace(key, persistent(es_group_norm_partials_bytes(N, groups))).first;
    es_gn_desc d;
    memset(&d, 0, sizeof(d));
    d.x = x.ptr(); d.partials = (float*)it->second;
    d.N = N; d.HW = (int)hw; d.C1 = x.c; d.C2 = 0; d.groups = groups; d.eps = eps; d.silu = 0; d.dtype = dt; d.stats_only = 1;
    ok(es_group_norm(&d, nullptr), "es_group_norm");
    T out = empty(N, x.h, x.w, pl[0]->cout);
    GnFront gn;
    gn.part = it->second; gn.nl = nl; gn.groups = groups; gn.nchunk = es_group_norm_chunks((int)hw); gn.hw = (int)hw; gn.eps = eps;
    std::vector<long long> rows;
    if (grouped) for (int n : group_n) rows.push_back(n * hw);
    linear_xs(x.view((int)M, 1, 1, x.c), pl, M, out.view((int)M, 1, 1, pl[0]->cout), rows, T(), &gn);
    return out;
  }

  T timestep_embedding(const T& t, int n, int dim) {
    T out = empty(n, 1, 1, dim);
    ok(es_timestep_embedding((const float*)t.ptr(), out.ptr(), n, dim, dt, nullptr), "es_timestep_embedding");
    return out;
  }
  void memcpy_t(const T& dst, const T& src) {
    if (!dst.contig() || !src.contig() || dst.bytes() != src.bytes()) fail("memcpy: contiguous tensors of equal byte size");
    ok(es_memcpy(dst.ptr(), src.ptr(), src.bytes(), nullptr), "es_memcpy");
  }
  void memcpy2d(unsigned long long dst, size_t dpitch, unsigned long long src, size_t spitch, size_t width, size_t height) {
    ok(es_memcpy2d((void*)dst, dpitch, (const void*)src, spitch, width, height, nullptr), "es_memcpy2d");
  }
  void gather_row(const T& table, int nrows, const T& idx, const T& out, int row_len) {
    ok(es_gather_row((const float*)table.ptr(), (const int32_t*)idx.ptr(), (float*)out.ptr(), row_len, nrows, nullptr), "es_gather_row");
  }
};

// ---- the model (edgestyle_amd/engine.py) --------------------------------------------------------------------------------------
struct UCfg {
  int in_ch, out_ch, nb, ch[4], has_attn[4], lpb, heads, cross, groups; float eps;
  int nce, ce[4], cond_ch;
};
struct VCfg { int nb, ch[4], lpb, latent, groups; float eps, scaling; int scale() const { return 1 << (nb - 1); } };

struct Resnet {
  Norm n1, n2;
  const PW *conv1 = nullptr, *conv2 = nullptr, *shortc = nullptr, *conv2s = nullptr;
  int groups = 32; float eps = 1e-5f; int temb_off = -1;
  Resnet() {}
  Resnet(Builder& B, const Weights& W, const std::string& p, int groups_, float eps_, int temb_off_) : groups(groups_), eps(eps_), temb_off(temb_off_) {
    const es_tensor* w1 = W.get(p + ".conv1.weight");
    const int cin = (int)w1->shape[1], cout = (int)w1->shape[0];
    W.shaped(p + ".conv1.weight", {cout, cin, 3, 3}); W.shaped(p + ".conv2.weight", {cout, cout, 3, 3});
    if (W.has(p + ".conv_shortcut.weight")) W.shaped(p + ".conv_shortcut.weight", {cout, cin, 1, 1});
    else if (cin != cout) fail("es_load_weights: '" + p + "' changes the channel count but has no conv_shortcut");
    if (temb_off_ >= 0) W.shaped(p + ".time_emb_proj.weight", {cout, -1});
    n1 = B.norm(W, p + ".norm1", cin); n2 = B.norm(W, p + ".norm2", cout);
    conv1 = B.conv(W, p + ".conv1");
    const bool has_short = W.has(p + ".conv_shortcut.weight");
    if (has_short && cout % 64 == 0 && cin % 64 == 0) conv2s = B.conv_tail(W, p + ".conv2", p + ".conv_shortcut");
    else { conv2 = B.conv(W, p + ".conv2"); if (has_short) shortc = B.conv(W, p + ".conv_shortcut"); }
  }
  // tproj: [rows, width] table of all time projections (null: the VAE's resnets)
  T run(Builder& B, const T& x, const T& tproj, const T& x2 = T()) const {
    T h = B.group_norm(x, n1, groups, eps, true, x2);
    CA a;
    if (temb_off >= 0) { a.temb = tproj.chan(temb_off, tproj.c - temb_off); a.temb_stride = tproj.ld; }
    a.gn_groups = groups;
    h = B.conv_gemm(h, conv1, a);
    h = B.group_norm(h, n2, groups, eps, true);
    if (conv2s) {
      if (x2 && (x.c % 64 || x2.c % 64)) fail("builder: skip concat with channels that are not multiples of 64");
      CA t; t.tails.push_back(x); if (x2) t.tails.push_back(x2); t.gn_groups = groups;
      return B.conv_gemm(h, conv2s, t);
    }
    T xs = x;
    if (shortc) { CA s; s.x2 = x2; xs = B.conv_gemm(x, shortc, s); }
    CA r; r.residual = xs; r.wide = true; r.gn_groups = groups;
    return B.conv_gemm(h, conv2, r);
  }
};

struct Transformer {               // Transformer2DModel(use_linear_projection False) + one BasicTransformerBlock, all folds on
  Norm norm;
  const PW *proj_in, *qkv_ln, *o1, *q2_ln, *kv2, *o2, *ff1_ln, *ffo;
  int heads = 8, groups = 32, c = 0;
  Transformer() {}
  Transformer(Builder& B, const Weights& W, const std::string& p, int heads_, int groups_) : heads(heads_), groups(groups_) {
    const std::string tb = p + ".transformer_blocks.0";
    const es_tensor* pi = W.get(p + ".proj_in.weight");
    c = (int)pi->shape[0];
    if (c % 64) fail("es_load_weights: transformer width " + std::to_string(c) + " ('" + p + "') is not a multiple of 64: the LayerNorm / proj_out folds of this builder need it");
    const int cross = (int)W.get(tb + ".attn2.to_k.weight")->shape[1];
    W.shaped(p + ".proj_in.weight", {c, c, 1, 1}); W.shaped(p + ".proj_out.weight", {c, c, 1, 1});
    for (const char* q : {".attn1.to_q", ".attn1.to_k", ".attn1.to_v", ".attn1.to_out.0", ".attn2.to_q", ".attn2.to_out.0"}) W.shaped(tb + q + ".weight", {c, c});
    W.shaped(tb + ".attn2.to_k.weight", {c, cross}); W.shaped(tb + ".attn2.to_v.weight", {c, cross});
    W.shaped(tb + ".ff.net.0.proj.weight", {8 * c, c}); W.shaped(tb + ".ff.net.2.weight", {c, 4 * c});
    if (c % heads_) fail("es_load_weights: transformer width of '" + p + "' is not a multiple of the head count");
    norm = B.norm(W, p + ".norm", c);
    proj_in = B.conv(W, p + ".proj_in");
    qkv_ln = B.cat_ln(W, {tb + ".attn1.to_q", tb + ".attn1.to_k", tb + ".attn1.to_v"}, false, tb + ".norm1");
    o1 = B.conv(W, tb + ".attn1.to_out.0");
    q2_ln = B.cat_ln(W, {tb + ".attn2.to_q"}, false, tb + ".norm2");
    kv2 = B.cat(W, {tb + ".attn2.to_k", tb + ".attn2.to_v"}, false);
    o2 = B.conv(W, tb + ".attn2.to_out.0");
    ff1_ln = B.cat_ln(W, {tb + ".ff.net.0.proj"}, true, tb + ".norm3", true);
    // proj_out(ff.net.2(f) + tok) = (Wp Wf) f + Wp tok + (Wp bf + bp): one GEMM over the channel concat [f | tok]
    const std::vector<float> wp = W.matrix(p + ".proj_out"), wf = W.matrix(tb + ".ff.net.2"), bf = W.vec(tb + ".ff.net.2.bias", c), bp = W.vec(p + ".proj_out.bias", c);
    const int K = 4 * c;
    if ((long long)wp.size() != (long long)c * c || (long long)wf.size() != (long long)c * K) fail("es_load_weights: proj_out / ff.net.2 shapes of '" + p + "'");
    std::vector<float> w((size_t)c * (K + c)), b((size_t)c);
    parallel_for(c, [&](long long a, long long e) {
      std::vector<double> acc((size_t)K);
      for (long long i = a; i < e; ++i) {
        std::fill(acc.begin(), acc.end(), 0.0);
        double bacc = 0.0;
        for (int m = 0; m < c; ++m) {
          const double wim = (double)wp[i * c + m];
          const float* fr = wf.data() + (size_t)m * K;
          for (int j = 0; j < K; ++j) acc[j] += wim * (double)fr[j];
          bacc += wim * (double)bf[m];
        }
        for (int j = 0; j < K; ++j) w[i * (K + c) + j] = (float)acc[j];
        for (int j = 0; j < c; ++j) w[i * (K + c) + K + j] = wp[i * c + j];
        b[i] = (float)(bacc + (double)bp[i]);
      }
    });
    ffo = B.pack(w, c, K + c, 1, &b);
  }
  T context(Builder& B, const T& ehs, const T& out, int rep) const { CA a; a.out = out; a.x_rep = rep; return B.linear(ehs, kv2, a); }
};

struct Encoder {                   // conv_in + time_embedding + down_blocks + mid_block (CL:623-632)
  UCfg cfg; int in_pad = 8;
  const PW *conv_in, *t1, *t2, *tproj;
  std::map<std::string, int> temb_offs;
  std::vector<std::vector<std::pair<Resnet, int>>> down;     // (resnet, index into tr or -1)
  std::vector<Transformer> tr;                               // transformers in forward order
  std::vector<const PW*> downsample;
  Resnet mid0, mid1; int mid_attn = -1;
  int n_enc_tr = 0;                                          // transformers of the encoder part (UNet appends its decoder's)
  void init(Builder& B, const Weights& W, const UCfg& c, const std::vector<std::string>& extra_resnets) {
    cfg = c;
    in_pad = round_up(c.in_ch, 8);
    W.shaped("conv_in.weight", {c.ch[0], c.in_ch, 3, 3});
    W.shaped("time_embedding.linear_1.weight", {4 * c.ch[0], c.ch[0]}); W.shaped("time_embedding.linear_2.weight", {4 * c.ch[0], 4 * c.ch[0]});
    conv_in = B.conv(W, "conv_in", in_pad);
    t1 = B.conv(W, "time_embedding.linear_1"); t2 = B.conv(W, "time_embedding.linear_2");
    std::vector<std::string> names;
    for (int i = 0; i < c.nb; ++i) for (int j = 0; j < c.lpb; ++j) names.push_back("down_blocks." + std::to_string(i) + ".resnets." + std::to_string(j));
    names.push_back("mid_block.resnets.0"); names.push_back("mid_block.resnets.1");
    names.insert(names.end(), extra_resnets.begin(), extra_resnets.end());
    int o = 0;
    std::vector<std::string> tp;
    for (const auto& nm : names) { temb_offs[nm] = o; o += (int)W.shaped(nm + ".time_emb_proj.weight", {-1, 4 * c.ch[0]})->shape[0]; tp.push_back(nm + ".time_emb_proj"); }
    tproj = B.cat(W, tp, true);
    for (int i = 0; i < c.nb; ++i) {
      down.emplace_back();
      for (int j = 0; j < c.lpb; ++j) {
        const std::string r = "down_blocks." + std::to_string(i) + ".resnets." + std::to_string(j);
        int ti = -1;
        Resnet rs(B, W, r, c.groups, c.eps, temb_offs[r]);
        if (c.has_attn[i]) { tr.emplace_back(B, W, "down_blocks." + std::to_string(i) + ".attentions." + std::to_string(j), c.heads, c.groups); ti = (int)tr.size() - 1; }
        down.back().emplace_back(rs, ti);
      }
      downsample.push_back(i != c.nb - 1 ? B.conv(W, "down_blocks." + std::to_string(i) + ".downsamplers.0.conv") : nullptr);
    }
    mid0 = Resnet(B, W, "mid_block.resnets.0", c.groups, c.eps, temb_offs["mid_block.resnets.0"]);
    tr.emplace_back(B, W, "mid_block.attentions.0", c.heads, c.groups); mid_attn = (int)tr.size() - 1;
    mid1 = Resnet(B, W, "mid_block.resnets.1", c.groups, c.eps, temb_offs["mid_block.resnets.1"]);
    n_enc_tr = (int)tr.size();
  }
  int tproj_width() const { return tproj->cout; }
  // CL:150-157 + every ResnetBlock2D.time_emb_proj(silu(emb)) in one shot: t fp32 [n] -> [n, sum(Cout)]
  T time_proj(Builder& B, const T& t, int n) const {
    T e = B.timestep_embedding(t, n, cfg.ch[0]);
    CA s; s.act = ES_ACT_SILU;
    e = B.linear(e, t1, s);
    e = B.linear(e, t2, s);
    return B.linear(e, tproj);
  }
};

struct UNet : Encoder {
  std::vector<std::vector<std::pair<Resnet, int>>> up;
  std::vector<const PW*> upsample;
  Norm norm_out; const PW* conv_out;
  void init(Builder& B, const Weights& W, const UCfg& c) {
    std::vector<std::string> extra;
    for (int i = 0; i < c.nb; ++i) for (int j = 0; j < c.lpb + 1; ++j) extra.push_back("up_blocks." + std::to_string(i) + ".resnets." + std::to_string(j));
    Encoder::init(B, W, c, extra);
    for (int i = 0; i < c.nb; ++i) {
      up.emplace_back();
      for (int j = 0; j < c.lpb + 1; ++j) {
        const std::string r = "up_blocks." + std::to_string(i) + ".resnets." + std::to_string(j);
        int ti = -1;
        Resnet rs(B, W, r, c.groups, c.eps, temb_offs[r]);
        if (c.has_attn[c.nb - 1 - i]) { tr.emplace_back(B, W, "up_blocks." + std::to_string(i) + ".attentions." + std::to_string(j), c.heads, c.groups); ti = (int)tr.size() - 1; }
        up.back().emplace_back(rs, ti);
      }
      upsample.push_back(i != c.nb - 1 ? B.conv(W, "up_blocks." + std::to_string(i) + ".upsamplers.0.conv") : nullptr);
    }
    W.shaped("conv_out.weight", {c.out_ch, c.ch[0], 3, 3});
    norm_out = B.norm(W, "conv_norm_out", c.ch[0]);
    conv_out = B.conv(W, "conv_out");
  }
};

struct ControlNet : Encoder {
  std::vector<const PW*> zero; const PW* zero_mid = nullptr;
  std::vector<const PW*> cond;     // conv-stack conditioning embedding (nets without a VAE)
  bool uses_vae = false;
  void init(Builder& B, const Weights& W, const UCfg& c, bool uses_vae_) {
    Encoder::init(B, W, c, {});
    uses_vae = uses_vae_;
    std::vector<int> rc{c.ch[0]};                      // channels of the 12 down residuals (MC:73-102)
    for (int i = 0; i < c.nb; ++i) { for (int j = 0; j < c.lpb; ++j) rc.push_back(c.ch[i]); if (i != c.nb - 1) rc.push_back(c.ch[i]); }
    for (size_t i = 0; i < rc.size(); ++i) {
      W.shaped("controlnet_down_blocks." + std::to_string(i) + ".weight", {rc[i], rc[i], 1, 1});
      zero.push_back(B.conv(W, "controlnet_down_blocks." + std::to_string(i)));
    }
    W.shaped("controlnet_mid_block.weight", {c.ch[c.nb - 1], c.ch[c.nb - 1], 1, 1});
    zero_mid = B.conv(W, "controlnet_mid_block");
    if (!uses_vae) {
      const std::string p = "controlnet_cond_embedding";
      W.shaped(p + ".conv_in.weight", {c.ce[0], c.cond_ch, 3, 3});
      W.shaped(p + ".conv_out.weight", {c.ch[0], c.ce[c.nce - 1], 3, 3});
      cond.push_back(B.conv(W, p + ".conv_in", 8));
      for (int i = 0; i < 2 * (c.nce - 1); ++i) cond.push_back(B.conv(W, p + ".blocks." + std::to_string(i)));
      cond.push_back(B.conv(W, p + ".conv_out"));
    }
  }
  T embed_cond(Builder& B, const T& img) const {      // ControlNetConditioningEmbedding: [N,H,W,8] -> [N,H/8,W/8,C0]
    CA s; s.act = ES_ACT_SILU;
    T h = B.conv_gemm(img, cond[0], s);
    for (size_t i = 1; i + 1 < cond.size(); ++i) { CA a; a.act = ES_ACT_SILU; a.stride = ((i - 1) % 2 == 1) ? 2 : 1; h = B.conv_gemm(h, cond[i], a); }
    return B.conv_gemm(h, cond.back());
  }
};

struct FusionParams { unsigned long long w1, b1, g1, be1, w2, b2, g2, be2, w3, b3; int c, s; };

struct VAE {
  VCfg cfg; int lat_pad = 8;
  struct Mid { Resnet r0, r1; Norm gn; const PW *qkv, *o; };
  const PW *e_in, *e_out, *quant, *post_quant_scaled, *d_in, *d_out;
  std::vector<std::pair<std::vector<Resnet>, const PW*>> e_down, d_up;
  Mid e_mid, d_mid; Norm e_norm, d_norm;
  Mid mid(Builder& B, const Weights& W, const std::string& side) {
    const std::string a = side + ".mid_block.attentions.0";
    const int c = cfg.ch[cfg.nb - 1];
    Mid m;
    m.r0 = Resnet(B, W, side + ".mid_block.resnets.0", cfg.groups, cfg.eps, -1);
    m.gn = B.norm(W, a + ".group_norm", c);
    m.qkv = B.cat(W, {a + ".to_q", a + ".to_k", a + ".to_v"}, true);
    m.o = B.conv(W, a + ".to_out.0");
    m.r1 = Resnet(B, W, side + ".mid_block.resnets.1", cfg.groups, cfg.eps, -1);
    return m;
  }
  void init(Builder& B, const Weights& W, const VCfg& c) {
    cfg = c;
    lat_pad = round_up(c.latent, 8);
    const int n = c.nb;
    W.shaped("encoder.conv_in.weight", {c.ch[0], 3, 3, 3}); W.shaped("encoder.conv_out.weight", {2 * c.latent, c.ch[n - 1], 3, 3});
    W.shaped("quant_conv.weight", {2 * c.latent, 2 * c.latent, 1, 1}); W.shaped("post_quant_conv.weight", {c.latent, c.latent, 1, 1});
    W.shaped("decoder.conv_in.weight", {c.ch[n - 1], c.latent, 3, 3}); W.shaped("decoder.conv_out.weight", {3, c.ch[0], 3, 3});
    for (const char* side : {"encoder", "decoder"})
      for (const char* q : {".to_q", ".to_k", ".to_v", ".to_out.0"}) W.shaped(std::string(side) + ".mid_block.attentions.0" + q + ".weight", {c.ch[n - 1], c.ch[n - 1]});
    e_in = B.conv(W, "encoder.conv_in", 8);
    for (int i = 0; i < n; ++i) {
      std::vector<Resnet> rs;
      for (int j = 0; j < c.lpb; ++j) rs.emplace_back(B, W, "encoder.down_blocks." + std::to_string(i) + ".resnets." + std::to_string(j), c.groups, c.eps, -1);
      e_down.emplace_back(rs, i != n - 1 ? B.conv(W, "encoder.down_blocks." + std::to_string(i) + ".downsamplers.0.conv") : nullptr);
    }
    e_mid = mid(B, W, "encoder");
    e_norm = B.norm(W, "encoder.conv_norm_out", c.ch[n - 1]);
    e_out = B.conv(W, "encoder.conv_out");
    quant = B.conv(W, "quant_conv");
    // decode(latents / scaling_factor) (PL:552-557) with the division folded into the 1x1 weights
    {
      const es_tensor* wt = W.get("post_quant_conv.weight");
      std::vector<float> w = W.matrix("post_quant_conv"), b;
      for (auto& v : w) v = (float)((double)v / (double)c.scaling);
      const bool hb = W.has("post_quant_conv.bias");
      if (hb) b = W.vec("post_quant_conv.bias", wt->shape[0]);
      post_quant_scaled = B.pack(w, (int)wt->shape[0], (int)wt->shape[1], 1, hb ? &b : nullptr, false, lat_pad, 8);
    }
    d_in = B.conv(W, "decoder.conv_in", 8);
    d_mid = mid(B, W, "decoder");
    for (int i = 0; i < n; ++i) {
      std::vector<Resnet> rs;
      for (int j = 0; j < c.lpb + 1; ++j) rs.emplace_back(B, W, "decoder.up_blocks." + std::to_string(i) + ".resnets." + std::to_string(j), c.groups, c.eps, -1);
      d_up.emplace_back(rs, i != n - 1 ? B.conv(W, "decoder.up_blocks." + std::to_string(i) + ".upsamplers.0.conv") : nullptr);
    }
    d_norm = B.norm(W, "decoder.conv_norm_out", c.ch[0]);
    d_out = B.conv(W, "decoder.conv_out");
  }
  T run_mid(Builder& B, const Mid& m, T h) const {
    h = m.r0.run(B, h, T());
    const int N = h.n, H = h.h, Wd = h.w, C = h.c;
    T n = B.group_norm(h, m.gn, cfg.groups, cfg.eps, false).view(N, H * Wd, 1, C);
    T qkv = B.linear(n, m.qkv);
    T a = B.attention(qkv.chan(0, C), qkv.chan(C, C), qkv.chan(2 * C, C), 1);
    CA r; r.residual = h.view(N, H * Wd, 1, C);
    h = B.linear(a, m.o, r).view(N, H, Wd, C);
    return m.r1.run(B, h, T());
  }
  T encode_moments(Builder& B, const T& img) const {       // [N,H,W,8 (3 real)] -> moments [N,H/8,W/8,2L]
    T h = B.conv_gemm(img, e_in);
    for (const auto& blk : e_down) {
      for (const auto& r : blk.first) h = r.run(B, h, T());
      if (blk.second) { CA a; a.stride = 2; a.pad = 0; a.out_h = h.h / 2; a.out_w = h.w / 2; h = B.conv_gemm(h, blk.second, a); }   // F.pad(0,1,0,1) + conv s2 p0
    }
    h = run_mid(B, e_mid, h);
    h = B.group_norm(h, e_norm, cfg.groups, cfg.eps, true);
    h = B.conv_gemm(h, e_out);
    return B.conv_gemm(h, quant);
  }
  T decode(Builder& B, const T& z) const {                 // scheduler latents [N,h,w,lat_pad] -> image [N,8h,8w,3]
    T h = B.conv_gemm(z, post_quant_scaled);
    h = B.conv_gemm(h, d_in);
    h = run_mid(B, d_mid, h);
    for (const auto& blk : d_up) {
      for (const auto& r : blk.first) h = r.run(B, h, T());
      if (blk.second) { CA a; a.upsample = true; h = B.conv_gemm(h, blk.second, a); }
    }
    h = B.group_norm(h, d_norm, cfg.groups, cfg.eps, true);
    return B.conv_gemm(h, d_out);
  }
};

// ---- GroupedEncoder (engine.py) + StepRunner (models.py), grouped lockstep mode ------------------------------------------------
struct Model {
  Builder B;
  UCfg ucfg; VCfg vcfg;
  es_ctx_geometry geo;
  Dict d_unet, d_vae, d_fusion, d_cn[6];
  UNet unet;
  VAE vae;
  std::vector<std::unique_ptr<ControlNet>> nets;          // distinct nets
  std::vector<int> net_of_cond;
  std::vector<std::pair<int, std::vector<int>>> groups;   // (net, positions): nets sharing weights run as one batched chain
  std::vector<FusionParams> fusion;
  std::vector<std::pair<int, int>> table;                 // (channels, size) of the 13 residual levels
  // grouped encoder
  std::vector<const Encoder*> encs; std::vector<int> counts; int ntot = 0, width = 0, ncn = 0;
  // static buffers (pipeline._Loop + StepState + NativeEngine)
  int B_ = 1, N = 2, T_ = 50, h = 64, w = 64, nn = 6, kmax = 3;
  T latents, model_in, noise, ehs, step_idx, t_rows, scales_cur, t_table, scale_table, coef, ts_dev, image;
  std::vector<T> conds, cond_img, cond_noise, gbufs, hist;
  std::vector<T> ctx_grouped, ctx_unet; std::vector<std::vector<T>> ctx_nets;
  T cond_cat, tproj_table, tproj_cur, tproj_gen;
  // StepRunner.prepare_fused_zero: what the fusion blocks return for zero residuals (biases and LayerNorm planes only) - constants,
  // filled by ONE real es_fusion_blocks launch at the end of es_load_weights; `fz_zero` / `fz_u` are that launch's zero inputs and scratch
  std::vector<T> fused_zero, fz_zero, fz_u;
  // guess_mode (es_ctx_geometry.guess_mode; StepRunner._step_guess): per-group K/V projections of the ControlNets' text states
  bool guess = false;
  int Bc = 2;                                              // samples the ControlNets run on: the conditional half under CFG
  std::vector<std::vector<T>> ctx_guess;

  std::vector<Norm> norms(const std::vector<const Resnet*>& rs, int which) const { std::vector<Norm> v; for (auto r : rs) v.push_back(which == 1 ? r->n1 : r->n2); return v; }

  T g_resnet(const std::vector<const Resnet*>& rs, const T& x, const T& tproj) {
    const Resnet* r0 = rs[0];
    T hh = B.group_norm(x, norms(rs, 1), r0->groups, r0->eps, true, T(), counts);
    PWs c1, c2, c2s, sh; bool all_s = true;
    for (auto r : rs) { c1.push_back(r->conv1); c2.push_back(r->conv2); c2s.push_back(r->conv2s); sh.push_back(r->shortc); if (!r->conv2s) all_s = false; }
    CA a; a.temb = tproj.chan(r0->temb_off, tproj.c - r0->temb_off); a.temb_stride = tproj.ld; a.group_n = counts; a.gn_groups = r0->groups;
    hh = B.conv_gemm(hh, c1, a);
    hh = B.group_norm(hh, norms(rs, 2), r0->groups, r0->eps, true, T(), counts);
    if (all_s) { CA t; t.tails.push_back(x); t.group_n = counts; t.gn_groups = r0->groups; return B.conv_gemm(hh, c2s, t); }
    for (auto r : rs) if (r->conv2s) fail("builder: grouped resnets must fold conv_shortcut all alike");
    T xs = x;
    if (r0->shortc) { CA s; s.group_n = counts; xs = B.conv_gemm(x, sh, s); }
    CA r; r.residual = xs; r.group_n = counts; r.wide = true; r.gn_groups = r0->groups;
    return B.conv_gemm(hh, c2, r);
  }
  T g_transformer(const std::vector<const Transformer*>& ts, const T& x, const T& kv) {
    const Transformer* t0 = ts[0];
    const int Nn = x.n, H = x.h, Wd = x.w, C = x.c;
    std::vector<int> rows;
    for (int n : counts) rows.push_back(n * H * Wd);
    std::vector<Norm> nl; PWs pin, qkv, o1, q2, o2, ff1, ffo;
    for (auto t : ts) { nl.push_back(t->norm); pin.push_back(t->proj_in); qkv.push_back(t->qkv_ln); o1.push_back(t->o1); q2.push_back(t->q2_ln); o2.push_back(t->o2); ff1.push_back(t->ff1_ln); ffo.push_back(t->ffo); }
    T tok = B.gn_proj_in(x, nl, t0->groups, 1e-6f, pin, counts).view(Nn, H * Wd, 1, C);
    CA gr; gr.group_n = rows;
    T q = B.linear(tok, qkv, gr);
    T a = B.attention(q.chan(0, C), q.chan(C, C), q.chan(2 * C, C), t0->heads);
    CA r1 = gr; r1.residual = tok; r1.wide = true;
    tok = B.linear(a, o1, r1);
    T qq = B.linear(tok, q2, gr);
    a = B.attention(qq, kv.chan(0, C), kv.chan(C, C), t0->heads);
    CA r2 = gr; r2.residual = tok; r2.wide = true;
    tok = B.linear(a, o2, r2);
    T f = B.linear(tok, ff1, gr);
    CA fo; fo.x2 = tok.view(Nn, H, Wd, C); fo.residual = x; fo.group_n = counts; fo.wide = true; fo.gn_groups = t0->groups;
    return B.conv_gemm(f.view(Nn, H, Wd, 4 * C), ffo, fo);
  }
  // h: [ntot,H,W,C0] -> (skips, mid) over the whole batch (GroupedEncoder.run)
  void g_run(T hh, const T& tproj, std::vector<T>& skips, T& mid) {
    const Encoder* e0 = encs[0];
    skips.push_back(hh);
    int ci = 0;
    auto rs_of = [&](const std::function<const Resnet*(const Encoder*)>& f) { std::vector<const Resnet*> v; for (auto e : encs) v.push_back(f(e)); return v; };
    auto ts_of = [&](int ti) { std::vector<const Transformer*> v; for (auto e : encs) v.push_back(&e->tr[ti]); return v; };
    for (size_t i = 0; i < e0->down.size(); ++i) {
      for (size_t j = 0; j < e0->down[i].size(); ++j) {
        hh = g_resnet(rs_of([&](const Encoder* e) { return &e->down[i][j].first; }), hh, tproj);
        if (e0->down[i][j].second >= 0) { hh = g_transformer(ts_of(e0->down[i][j].second), hh, ctx_grouped[ci]); ++ci; }
        skips.push_back(hh);
      }
      if (e0->downsample[i]) {
        PWs ds; for (auto e : encs) ds.push_back(e->downsample[i]);
        CA a; a.stride = 2; a.group_n = counts; a.gn_groups = ucfg.groups;
        hh = B.conv_gemm(hh, ds, a);
        skips.push_back(hh);
      }
    }
    hh = g_resnet(rs_of([](const Encoder* e) { return &e->mid0; }), hh, tproj);
    hh = g_transformer(ts_of(e0->mid_attn), hh, ctx_grouped[ci]);
    mid = g_resnet(rs_of([](const Encoder* e) { return &e->mid1; }), hh, tproj);
  }
  // one transformer of the UNet decoder (engine.Transformer.__call__, folded paths)
  T transformer(const Transformer& t, const T& x, const T& kv) {
    const int Nn = x.n, H = x.h, Wd = x.w, C = x.c;
    T tok = B.gn_proj_in(x, std::vector<Norm>{t.norm}, t.groups, 1e-6f, PWs{t.proj_in}).view(Nn, H * Wd, 1, C);
    T q = B.linear(tok, t.qkv_ln);
    T a = B.attention(q.chan(0, C), q.chan(C, C), q.chan(2 * C, C), t.heads);
    CA r1; r1.residual = tok; r1.wide = true;
    tok = B.linear(a, t.o1, r1);
    T qq = B.linear(tok, t.q2_ln);
    a = B.attention(qq, kv.chan(0, C), kv.chan(C, C), t.heads);
    CA r2; r2.residual = tok; r2.wide = true;
    tok = B.linear(a, t.o2, r2);
    T f = B.linear(tok, t.ff1_ln);
    CA fo; fo.x2 = tok.view(Nn, H, Wd, C); fo.residual = x; fo.wide = true; fo.gn_groups = t.groups;
    return B.conv_gemm(f.view(Nn, H, Wd, 4 * C), t.ffo, fo);
  }

  // StepRunner.set_context: every cross-attention K/V projection of the text states, once per call
  void set_context() {
    size_t gi = 0;
    for (const auto& g : groups) {
      const ControlNet& net = *nets[g.first];
      for (int ti = 0; ti < net.n_enc_tr; ++ti) net.tr[ti].context(B, ehs, ctx_nets[gi][ti], (int)g.second.size());
      ++gi;
    }
    for (size_t ti = 0; ti < unet.tr.size(); ++ti) unet.tr[ti].context(B, ehs, ctx_unet[ti], 1);
  }
  void set_conds() {
    int a = 0;
    for (const auto& g : groups) for (int p : g.second) { B.memcpy_t(cond_cat.batch(a, N), conds[p]); a += N; }
  }
  // StepRunner.set_time_table: tproj_table[s] = the time projections of step s for every row of the grouped batch
  void set_time_table() {
    int a = 0;
    const size_t es = 2;
    for (size_t e = 0; e < encs.size(); ++e) {
      T proj = encs[e]->time_proj(B, ts_dev, T_);
      const size_t wb = (size_t)encs[e]->tproj_width() * es;
      for (int j = 0; j < counts[e]; ++j)
        B.memcpy2d(tproj_table.p + (unsigned long long)(a + j) * width * es, (size_t)ntot * width * es, proj.p, wb, wb, (size_t)T_);
      a += counts[e];
    }
  }
  T time_proj_generic() {                                   // GroupedEncoder.time_proj(t_rows, tproj_gen)
    const size_t es = 2;
    int a = 0;
    for (size_t e = 0; e < encs.size(); ++e) {
      T proj = encs[e]->time_proj(B, t_rows, counts[e]);
      const size_t wb = (size_t)encs[e]->tproj_width() * es;
      B.memcpy2d(tproj_gen.p + (unsigned long long)a * width * es, (size_t)width * es, proj.p, wb, wb, (size_t)counts[e]);
      a += counts[e];
    }
    return tproj_gen;
  }
  // EdgeStyleMultiControlNetModel's engine.forward (MC:151-169): rpn[net][level] = that net's residual (first sample; the batch
  // stride is the tensor's), Nf samples per net, optional addends (the UNet's own skip / mid tensors: outputs are then the sums)
  std::vector<T> fuse(const std::vector<std::vector<T>>& rpn, int Nf, const std::vector<T>* addends) {
    const size_t nl = rpn[0].size();
    std::vector<es_fusion_desc> fd(nl);
    std::vector<T> us, outs;
    for (size_t k = 0; k < nl; ++k) {
      const FusionParams& fp = fusion[k];
      const int HW = fp.s * fp.s, Cc = fp.c;
      auto key = std::make_pair(Nf, (int)k);
      auto it = B.fusion_scratch.find(key);
      if (it == B.fusion_scratch.end()) it = B.fusion_scratch.emplace(key, B.persistent(es_fusion_scratch_bytes(Nf))).first;
      T u = B.empty(Nf, HW, 1, Cc), out = B.empty(Nf, HW, 1, Cc);
      es_fusion_desc& d = fd[k];
      memset(&d, 0, sizeof(d));
      for (int i = 0; i < 6; ++i) { d.res[i] = rpn[i][k].ptr(); d.res_bs[i] = rpn[i][k].bstride(); d.res_scale[i] = 1.f; }
      d.res_scale_dev = (const float*)scales_cur.ptr();
      d.w1 = (const float*)fp.w1; d.b1 = (const float*)fp.b1; d.g1 = (const void*)fp.g1; d.be1 = (const void*)fp.be1;
      d.w2 = (const float*)fp.w2; d.b2 = (const float*)fp.b2; d.g2 = (const void*)fp.g2; d.be2 = (const void*)fp.be2;
      d.w3 = (const float*)fp.w3; d.b3 = (const float*)fp.b3;
      d.scratch = (float*)it->second; d.u = u.ptr(); d.out = out.ptr();
      d.N = Nf; d.HW = HW; d.C = Cc; d.eps = 1e-5f; d.dtype = B.dt;
      if (addends) {
        const T& ad = (*addends)[k];
        if (ad.numel() != (long long)Nf * HW * Cc || !ad.contig()) fail("builder: fusion addend of another size");
        d.addend = ad.ptr();
      }
      us.push_back(u);
      outs.push_back(out.view(Nf, fp.s, fp.s, Cc));
    }
    Builder::ok(es_fusion_blocks(fd.data(), (int)fd.size(), nullptr), "es_fusion_blocks");
    return outs;
  }
  // StepRunner._step_grouped + UNet.forward(presummed)
  void step(bool table_driven) {
    const T& x = model_in;
    const int c0 = unet.conv_in->cout;
    T h0 = B.empty(ntot, x.h, x.w, c0);
    {
      PWs ci; for (auto e : encs) ci.push_back(e->conv_in);
      CA a; a.residual = cond_cat; a.group_n = counts; a.out = h0; a.x_rep = ntot / N; a.wide = true; a.gn_groups = ucfg.groups;
      h0 = B.conv_gemm(x, ci, a);                          // (the same buffer; with a wide stream it now carries its low part)                               // sample = conv_in(sample) + cond for every net, and the UNet's conv_in (CL:197-203)
    }
    T tproj;
    if (table_driven) {
      B.gather_row(tproj_table, T_, step_idx, tproj_cur, (int)((long long)ntot * width * 2 / 4));
      tproj = tproj_cur;
    } else tproj = time_proj_generic();
    std::vector<T> skips; T mid;
    g_run(h0, tproj, skips, mid);
    h0 = T();
    std::vector<T> srcs = skips; srcs.push_back(mid);
    std::vector<T> enc;
    for (const auto& s : srcs) enc.push_back(s.batch(ncn));
    std::vector<T> fused;
    if (nn == 1) {
      // one ControlNet: skip + scale * zero_conv(cn_skip) straight out of the zero-conv epilogue (PL:500-510; CL:266-270)
      const ControlNet* cn = (const ControlNet*)encs[0];
      for (size_t lvl = 0; lvl < srcs.size(); ++lvl) {
        CA a; a.residual = enc[lvl]; a.out_scale = 1.f; a.out_scale_dev = scales_cur.chan(0, 1);
        fused.push_back(B.conv_gemm(srcs[lvl].batch(0, ncn), lvl + 1 < srcs.size() ? cn->zero[lvl] : cn->zero_mid, a));
      }
    } else {
    // zero-convs of all levels (grouped over the ControlNets) -> the fusion blocks, which also add the UNet's own tensors
    std::vector<int> cn_counts(counts.begin(), counts.end() - 1);
    std::vector<T> res;
    for (size_t lvl = 0; lvl < srcs.size(); ++lvl) {
      PWs z; for (size_t e = 0; e + 1 < encs.size(); ++e) { const ControlNet* cn = (const ControlNet*)encs[e]; z.push_back(lvl + 1 < srcs.size() ? cn->zero[lvl] : cn->zero_mid); }
      CA a; a.group_n = cn_counts;
      res.push_back(B.conv_gemm(srcs[lvl].batch(0, ncn), z, a));
    }
    std::vector<int> first(nn, 0);
    { int a = 0; for (const auto& g : groups) for (int p : g.second) { first[p] = a; a += N; } }
    std::vector<std::vector<T>> rpn(nn);
    for (int i = 0; i < nn; ++i) for (size_t k = 0; k < srcs.size(); ++k) rpn[i].push_back(res[k].batch(first[i]));
    fused = fuse(rpn, N, &enc);
    }
    srcs.clear(); skips.clear(); enc.clear(); mid = T();
    unet_decoder(fused, tproj.batch(ncn));                  // skip + residual already summed: PL:500-510
  }
  // UNet.forward from the (summed) skip / mid tensors on: `fused` = 12 skips + the mid tensor at the back
  void unet_decoder(std::vector<T>& fused, const T& tp) {
    T hh = fused.back();
    fused.pop_back();
    int ci = unet.n_enc_tr;
    for (size_t i = 0; i < unet.up.size(); ++i) {
      for (const auto& ra : unet.up[i]) {
        T skip = fused.back();
        fused.pop_back();
        hh = ra.first.run(B, hh, tp, skip);
        if (ra.second >= 0) { hh = transformer(unet.tr[ra.second], hh, ctx_unet[ci]); ++ci; }
      }
      if (unet.upsample[i]) { CA a; a.upsample = true; hh = B.conv_gemm(hh, unet.upsample[i], a); }
    }
    hh = B.group_norm(hh, unet.norm_out, ucfg.groups, ucfg.eps, true);
    CA o; o.out = noise;
    B.conv_gemm(hh, unet.conv_out, o);
  }
  // Encoder.run of ONE encoder (engine.Encoder.run): h -> (skips, mid)
  void enc_run(const Encoder& e, T hh, const T& tp, const std::vector<T>& ctx, std::vector<T>& skips, T& mid) {
    skips.push_back(hh);
    int ci = 0;
    for (size_t i = 0; i < e.down.size(); ++i) {
      for (size_t j = 0; j < e.down[i].size(); ++j) {
        hh = e.down[i][j].first.run(B, hh, tp);
        if (e.down[i][j].second >= 0) { hh = transformer(e.tr[e.down[i][j].second], hh, ctx[ci]); ++ci; }
        skips.push_back(hh);
      }
      if (e.downsample[i]) { CA a; a.stride = 2; a.gn_groups = ucfg.groups; hh = B.conv_gemm(hh, e.downsample[i], a); skips.push_back(hh); }
    }
    hh = e.mid0.run(B, hh, tp);
    hh = transformer(e.tr[e.mid_attn], hh, ctx[ci]);
    mid = e.mid1.run(B, hh, tp);
  }
  // ---- guess_mode (StepRunner.set_context(guess_mode=True) / _step_guess; CL:256-264, PL:453-459, 487-497) -------------------------
  void set_context_guess() {
    for (size_t ti = 0; ti < unet.tr.size(); ++ti) unet.tr[ti].context(B, ehs, ctx_unet[ti], 1);
    const T ehs_c = ehs.batch(N - Bc);                      // the ControlNets see the last Bc rows: the conditional half under CFG
    size_t gi = 0;
    for (const auto& g : groups) {
      const ControlNet& net = *nets[g.first];
      for (int ti = 0; ti < net.n_enc_tr; ++ti) net.tr[ti].context(B, ehs_c, ctx_guess[gi][ti], (int)g.second.size());
      ++gi;
    }
  }
  // torch.logspace(-1, 0, n) of dtype float (ATen's CPU kernel fills the second half from the end), as models._guess_level_scales hands it over
  static std::vector<float> guess_level_scales(int n) {
    std::vector<float> ls((size_t)n);
    const double step = (0.0 - (-1.0)) / (double)(n - 1);  // (the exponent and the power in double, the result rounded to float)
    for (int i = 0; i < n; ++i) ls[i] = (float)(i < n / 2 ? pow(10.0, -1.0 + step * (double)i) : pow(10.0, 0.0 - step * (double)(n - i - 1)));
    return ls;
  }
  void step_guess() {
    const T& x = model_in;
    const int Bh = N - Bc;                                  // first row of the ControlNet batch inside x (0 without CFG)
    const T xc = x.batch(Bh);
    const std::vector<float> ls = guess_level_scales((int)table.size());
    std::vector<std::vector<T>> rpn(nn);
    size_t gi = 0;
    for (const auto& g : groups) {
      const ControlNet& cn = *nets[g.first];
      const int k = (int)g.second.size();
      T tp = cn.time_proj(B, t_rows, k * Bc);
      // ControlNet.forward: sample = conv_in(sample) + cond per net (CL:197-203), one batched pass, the zero-convs scaled per level
      const int c0 = cn.conv_in->cout;
      T h0 = B.empty(k * Bc, x.h, x.w, c0);
      for (int i = 0; i < k; ++i) { CA a; a.residual = conds[g.second[i]]; a.out = h0.batch(i * Bc, Bc); B.conv_gemm(xc, cn.conv_in, a); }
      std::vector<T> skips; T mid;
      enc_run(cn, h0, tp, ctx_guess[gi], skips, mid);
      std::vector<T> res;
      // (one plain ControlNet: its conditioning scale rides on the same epilogue, StepRunner._cn_scale)
      for (size_t lvl = 0; lvl <= skips.size(); ++lvl) {
        CA a; a.out_scale = ls[lvl];
        if (nn == 1) a.out_scale_dev = scales_cur.chan(0, 1);
        res.push_back(lvl < skips.size() ? B.conv_gemm(skips[lvl], cn.zero[lvl], a) : B.conv_gemm(mid, cn.zero_mid, a));
      }
      for (int j = 0; j < k; ++j) for (const auto& r : res) rpn[g.second[j]].push_back(r.batch(j * Bc));
      ++gi;
    }
    std::vector<T> fused = nn == 1 ? rpn[0] : fuse(rpn, Bc, nullptr);
    T tpu = unet.time_proj(B, t_rows, N);
    CA ci_; ci_.gn_groups = ucfg.groups;
    T hh = B.conv_gemm(x, unet.conv_in, ci_);
    std::vector<T> skips;
    enc_run(unet, hh, tpu, ctx_unet, skips, hh);
    // torch.cat([zeros, d]) + skip == add into the conditional half (PL:487-497), in place
    for (size_t k2 = 0; k2 < skips.size(); ++k2) { T v = skips[k2].batch(Bh); B.add_into(v, fused[k2].view(v.n, v.h, v.w, v.c), v); }
    { T v = hh.batch(Bh); B.add_into(v, fused.back().view(v.n, v.h, v.w, v.c), v); }
    skips.push_back(hh);
    unet_decoder(skips, tpu);
  }
  // pipeline._Loop.one_step_unet / StepRunner.step_unet_only: a step outside every control-guidance window (PL:419-427) - the UNet
  // alone, its skip / mid tensors plus the constant fusion-of-zeros residuals (a single ControlNet: plus nothing)
  void one_step_unet() {
    B.gather_row(t_table, T_, step_idx, t_rows, kmax * N);
    B.gather_row(tproj_table, T_, step_idx, tproj_cur, (int)((long long)ntot * width * 2 / 4));
    T tp = tproj_cur.batch(ncn);
    CA ci_; ci_.gn_groups = ucfg.groups;
    T hh = B.conv_gemm(model_in, unet.conv_in, ci_);
    std::vector<T> skips;
    enc_run(unet, hh, tp, ctx_unet, skips, hh);
    if (nn != 1) {
      if (fused_zero.size() != skips.size() + 1) fail("builder: fusion-of-zeros constants do not match the residual levels");
      for (size_t k = 0; k < skips.size(); ++k) skips[k] = B.add(skips[k], fused_zero[k].view(skips[k].n, skips[k].h, skips[k].w, skips[k].c));
      hh = B.add(hh, fused_zero.back().view(hh.n, hh.h, hh.w, hh.c));
    }
    skips.push_back(hh);
    unet_decoder(skips, tp);
    Builder::ok(es_cfg_ddim_step(noise.ptr(), (float*)latents.ptr(), model_in.ptr(), (const float*)coef.ptr(), (const int32_t*)step_idx.ptr(), geo.cfg ? 7.5f : 1.0f,
                                 B_, h * w, ucfg.in_ch, model_in.c, geo.cfg, T_, B.dt, nullptr), "es_cfg_ddim_step");
    Builder::ok(es_incr((int32_t*)step_idx.ptr(), nullptr), "es_incr");
  }
  void one_step() {                                         // pipeline._Loop.one_step: PL:435-522 for the step the device counter selects
    B.gather_row(t_table, T_, step_idx, t_rows, kmax * N);
    B.gather_row(scale_table, T_, step_idx, scales_cur, nn);
    if (guess) step_guess(); else step(true);
    Builder::ok(es_cfg_ddim_step(noise.ptr(), (float*)latents.ptr(), model_in.ptr(), (const float*)coef.ptr(), (const int32_t*)step_idx.ptr(), geo.cfg ? 7.5f : 1.0f,
                                 B_, h * w, ucfg.in_ch, model_in.c, geo.cfg, T_, B.dt, nullptr), "es_cfg_ddim_step");
    Builder::ok(es_incr((int32_t*)step_idx.ptr(), nullptr), "es_incr");
  }
  void decode() {
    T dec = vae.decode(B, model_in.batch(0, B_));
    Builder::ok(es_nhwc_to_nchw_f32(dec.ptr(), (float*)image.ptr(), dec.n, 3, (int)dec.hw(), dec.c, 0.5f, 0.5f, 1, B.dt, nullptr), "es_nhwc_to_nchw_f32");
  }
  // NativeEngine._conds_fn: each shared encoder once over the un-duplicated images of all its nets
  struct CondGroup { bool vae; int net; std::vector<int> idx; T gb; };
  std::vector<CondGroup> cgroups;
  void embed_conds() {
    const int rep = Bc / B_, N = Bc;                        // (guess_mode under CFG: the ControlNets only see the conditional half)
    for (const auto& g : cgroups) {
      T x8 = B.empty(g.gb.n, g.gb.h, g.gb.w, 8);
      Builder::ok(es_nchw_f32_to_nhwc((const float*)g.gb.ptr(), x8.ptr(), g.gb.n, 3, (int)g.gb.hw(), 8, B.dt, nullptr), "es_nchw_f32_to_nhwc");
      if (g.vae) {
        T mom = vae.encode_moments(B, x8);
        for (size_t k = 0; k < g.idx.size(); ++k) {
          const int i = g.idx[k];
          T mk = mom.batch((int)k * B_, B_), mn = mk;
          if (rep > 1) { mn = B.empty(N, mk.h, mk.w, mk.c); B.memcpy_t(mn.batch(0, B_), mk); B.memcpy_t(mn.batch(B_, B_), mk); }
          const ControlNet& net = *nets[net_of_cond[i]];
          T z = B.empty(N, mn.h, mn.w, net.in_pad);
          Builder::ok(es_vae_sample(mn.ptr(), (const float*)cond_noise[i].ptr(), z.ptr(), N, (int)mn.hw(), vcfg.latent, net.in_pad, vcfg.scaling, B.dt, nullptr), "es_vae_sample");
          B.memcpy_t(conds[i], B.conv_gemm(z, net.conv_in));        // conv_vae_out IS conv_in (CL:36,41,595-598)
        }
      } else {
        T emb = nets[g.net]->embed_cond(B, x8);
        for (size_t k = 0; k < g.idx.size(); ++k) {
          const int i = g.idx[k];
          T e = emb.batch((int)k * B_, B_);
          B.memcpy_t(conds[i].batch(0, B_), e);
          if (rep > 1) B.memcpy_t(conds[i].batch(B_, B_), e);
        }
      }
    }
  }

  void pack_fusion(const Weights& W) {
    for (size_t i = 0; i < table.size(); ++i) {
      const std::string p = i + 1 < table.size() ? "multi_controlnet_down_blocks." + std::to_string(i) : "multi_controlnet_mid_block";
      const int c = table[i].first, s = table[i].second, hw = s * s;
      FusionParams f; f.c = c; f.s = s;
      f.w1 = B.upload_f32(to_f32(W.shaped(p + ".first_conv.weight", {3 * c, 2, 1, 1})));                 // [3C,2,1,1], channel j = 3c + p -> [C][3][2]
      f.b1 = B.upload_f32(to_f32(W.shaped(p + ".first_conv.bias", {3 * c})));
      f.w2 = B.upload_f32(to_f32(W.shaped(p + ".second_conv.weight", {c, 3, 1, 1})));
      f.b2 = B.upload_f32(to_f32(W.shaped(p + ".second_conv.bias", {c})));
      f.w3 = B.upload_f32(to_f32(W.shaped(p + ".third_conv.weight", {c, 1, 1, 1})));
      f.b3 = B.upload_f32(to_f32(W.shaped(p + ".third_conv.bias", {c})));
      auto plane = [&](const std::string& key, int per) {    // [per*C, H, W] -> [HW][C][per] (pixel-major), compute dtype
        const std::vector<float> t = to_f32(W.shaped(key, {(long long)per * c, s, s}));
        const unsigned long long a = B.persistent((size_t)hw * c * per * 2);
        uint16_t* d = (uint16_t*)B.staged(a, (size_t)hw * c * per * 2);
        parallel_for(hw, [&](long long x0, long long x1) {
          for (long long px = x0; px < x1; ++px)
            for (int cc = 0; cc < c * per; ++cc) d[(size_t)px * c * per + cc] = B.enc(t[(size_t)cc * hw + px]);
        });
        return a;
      };
      f.g1 = plane(p + ".first_normalization.weight", 3); f.be1 = plane(p + ".first_normalization.bias", 3);
      f.g2 = plane(p + ".second_normalization.weight", 1); f.be2 = plane(p + ".second_normalization.bias", 1);
      fusion.push_back(f);
    }
  }
};

struct Recording {
  es_plan* p;
  explicit Recording(es_plan* pl) : p(pl) { if (es_plan_begin_record(p)) fail(std::string("es_load_weights: ") + es_last_error()); es_plan_set_dry(1); }
  ~Recording() { es_plan_set_dry(0); es_plan_end_record(p); }
};

}  // namespace

extern "C" int es_plan_gemm_choice(long long M, int rows_padded, int kpad, int geglu, const int* bns, int n_bns, int allow_split, int* bn, int* splitk, int* stages) {
  if (!bns || n_bns < 1 || !bn || !splitk || !stages) { es_set_error("es_plan_gemm_choice: null argument"); return -1; }
  if (!plan_gemm(M, rows_padded, kpad, geglu != 0, bns, n_bns, allow_split != 0, bn, splitk, stages)) { es_set_error("es_plan_gemm_choice: rows_padded fits no N tile"); return -1; }
  return 0;
}
extern "C" int es_linear_xs_eligible(long long M, int ksize, int kpad, int cin, int ctail, int cout, int geglu) {
  return xs_shape_ok(M, ksize, kpad, cin, ctail, cout, geglu != 0) ? 1 : 0;
}

extern "C" int es_load_weights(const es_weights* wts, const es_model_config* mc, const es_ctx_geometry* g, int device, es_ctx** out) {
  if (!wts || !mc || !g || !out) { es_set_error("es_load_weights: null argument"); return -1; }
  es_ctx* ctx = nullptr;
  es_plan* plans[ES_PLAN_COUNT] = {};
  void* arena = nullptr;
  try {
    auto m = std::make_unique<Model>();
    Model& M = *m;
    Builder& B = M.B;
    // ---- configuration
    if (mc->n_blocks < 2 || mc->n_blocks > 4 || mc->vae_n_blocks < 2 || mc->vae_n_blocks > 4) fail("es_load_weights: 2..4 resolution levels");
    if (g->n_conds != 6 && g->n_conds != 1) fail("es_load_weights: n_conds is 6 (the reference's fused multi-ControlNet model, MC:66-114) or 1 (a single ControlNet, PL:338-351)");
    if (g->n_conds == 1 && wts->n_controlnets != 1) fail("es_load_weights: a single-ControlNet context takes exactly one ControlNet");
    if (g->dtype != ES_F16 && g->dtype != ES_BF16) fail("es_load_weights: dtype must be ES_F16 or ES_BF16");
    if (g->B < 1 || g->h < 1 || g->w < 1 || g->n_steps < 1) fail("es_load_weights: bad geometry");
    if (wts->n_controlnets < 1 || wts->n_controlnets > 6) fail("es_load_weights: 1..6 distinct ControlNets");
    UCfg& u = M.ucfg;
    u.in_ch = mc->in_channels; u.out_ch = mc->out_channels; u.nb = mc->n_blocks; u.lpb = mc->layers_per_block; u.heads = mc->num_heads;
    u.cross = mc->cross_attention_dim; u.groups = mc->norm_num_groups; u.eps = mc->norm_eps; u.nce = mc->n_cond_embed; u.cond_ch = mc->conditioning_channels;
    for (int i = 0; i < 4; ++i) { u.ch[i] = mc->block_out_channels[i]; u.has_attn[i] = mc->down_has_attn[i]; u.ce[i] = mc->cond_embed_channels[i]; }
    VCfg& v = M.vcfg;
    v.nb = mc->vae_n_blocks; v.lpb = mc->vae_layers_per_block; v.latent = mc->vae_latent_channels; v.groups = mc->vae_norm_num_groups;
    v.eps = mc->vae_norm_eps; v.scaling = mc->vae_scaling_factor;
    for (int i = 0; i < 4; ++i) v.ch[i] = mc->vae_block_out_channels[i];
    if (u.in_ch != v.latent) fail("es_load_weights: the UNet's in_channels must equal the VAE's latent_channels");
    if (u.cond_ch != 3 || u.nce < 2 || u.nce > 4) fail("es_load_weights: conditioning embedding of 3 input channels, 2..4 widths");
    M.geo = *g;
    B.dt = g->dtype;
    M.B_ = g->B; M.N = g->cfg ? 2 * g->B : g->B; M.T_ = g->n_steps; M.h = g->h; M.w = g->w; M.nn = g->n_conds;
    M.guess = g->guess_mode != 0;
    M.Bc = M.guess ? g->B : M.N;
    const int N = M.N, TS = M.T_, h = g->h, w = g->w, Bn = g->B, NN = g->n_conds, Nc = M.Bc;
    if (!M.guess) {   // the groups of the lockstep pass: nets that share weights run as one batched chain; every group must tile
      int cnt[6] = {};
      for (int p = 0; p < NN; ++p) {
        const int ni = wts->net_of_cond[p];
        if (ni < 0 || ni >= wts->n_controlnets) fail("es_load_weights: net_of_cond names a ControlNet that was not given");
        ++cnt[ni];
      }
      const int hw_min = (h >> (u.nb - 1)) * (w >> (u.nb - 1));
      for (int i = 0; i <= wts->n_controlnets; ++i) {
        const long long n = (i < wts->n_controlnets ? (long long)cnt[i] : 1) * N;
        if (n && (n * hw_min) % BM) fail("es_load_weights: the groups of the lockstep pass do not tile in 128-pixel units at this latent size (the Python host falls back to serial chains; this builder does not)");
      }
    }
    // ---- state dicts
    M.d_unet.init(wts->unet, "UNet"); M.d_vae.init(wts->vae, "VAE"); M.d_fusion.init(wts->fusion, "fusion");
    for (int i = 0; i < wts->n_controlnets; ++i) M.d_cn[i].init(wts->controlnet[i], ("ControlNet " + std::to_string(i)).c_str());
    Weights Wu; Wu.own = &M.d_unet; Wu.who = "UNet";
    Weights Wv; Wv.own = &M.d_vae; Wv.who = "VAE";
    Weights Wf; Wf.own = &M.d_fusion; Wf.who = "fusion";
    // ---- weights
    M.unet.init(B, Wu, u);
    M.vae.init(B, Wv, v);
    for (int i = 0; i < wts->n_controlnets; ++i) {
      Weights Wc; Wc.own = &M.d_cn[i]; Wc.who = "ControlNet " + std::to_string(i);
      const int kind = wts->controlnet_kind[i];
      if (kind != ES_NET_CONTROLNET && kind != ES_NET_CONTROL_LORA_VAE && kind != ES_NET_CONTROL_LORA) fail("es_load_weights: unknown controlnet_kind");
      if (kind != ES_NET_CONTROLNET) Wc.tied = &M.d_unet;
      M.nets.emplace_back(new ControlNet());
      M.nets.back()->init(B, Wc, u, kind == ES_NET_CONTROL_LORA_VAE);
    }
    for (int p = 0; p < NN; ++p) {
      const int ni = wts->net_of_cond[p];
      if (ni < 0 || ni >= wts->n_controlnets) fail("es_load_weights: net_of_cond names a ControlNet that was not given");
      M.net_of_cond.push_back(ni);
      bool found = false;
      for (auto& gp : M.groups) if (gp.first == ni) { gp.second.push_back(p); found = true; }
      if (!found) M.groups.push_back({ni, {p}});
    }
    M.kmax = 0;
    for (const auto& gp : M.groups) M.kmax = std::max<int>(M.kmax, (int)gp.second.size());
    if (M.groups.size() > 3) fail("es_load_weights: at most 3 distinct ControlNets run in lockstep with the UNet (4 groups per launch)");
    // residual table (MC:73-102): (channels, size) of the 12 down + 1 mid residuals
    {
      if (h != w) fail("es_load_weights: the fusion blocks' LayerNorm planes are square (MC:34-36): h must equal w");
      int s = h;
      M.table.push_back({u.ch[0], s});
      for (int i = 0; i < u.nb; ++i) {
        for (int j = 0; j < u.lpb; ++j) M.table.push_back({u.ch[i], s});
        if (i != u.nb - 1) { s /= 2; M.table.push_back({u.ch[i], s}); }
      }
      M.table.push_back({u.ch[u.nb - 1], s});
      if (M.table.size() > ES_FUSION_MAX_BATCH) fail("es_load_weights: more residual levels than es_fusion_blocks takes");
    }
    if (NN > 1) M.pack_fusion(Wf);
    // ---- the grouped encoder
    for (const auto& gp : M.groups) { M.encs.push_back(M.nets[gp.first].get()); M.counts.push_back((int)gp.second.size() * N); }
    M.encs.push_back(&M.unet); M.counts.push_back(N);
    M.ntot = 0; M.width = 0;
    for (size_t e = 0; e < M.encs.size(); ++e) { M.ntot += M.counts[e]; M.width = std::max(M.width, M.encs[e]->tproj_width()); }
    M.ncn = M.ntot - N;
    // ---- static buffers
    const int Lc = u.in_ch, Lp = M.unet.in_pad, c0 = u.ch[0], sc = v.scale();
    M.latents = B.persistent_t(Bn, h, w, Lc, 4);
    M.model_in = B.persistent_t(N, h, w, Lp);
    M.noise = B.persistent_t(N, h, w, u.out_ch);
    for (int i = 0; i < NN; ++i) M.conds.push_back(B.persistent_t(Nc, h, w, c0));
    M.ehs = B.persistent_t(N, mc->text_tokens > 0 ? mc->text_tokens : 77, 1, u.cross);
    M.step_idx = B.persistent_t(1, 1, 1, 1, 4);
    M.t_rows = B.persistent_t(1, 1, 1, M.kmax * N, 4);
    M.scales_cur = B.persistent_t(1, 1, 1, NN, 4);
    M.t_table = B.persistent_t(TS, 1, 1, M.kmax * N, 4);
    M.scale_table = B.persistent_t(TS, 1, 1, NN, 4);
    M.coef = B.persistent_t(TS, 1, 1, 12, 4);                       // T x 4 (DDIM) or T x 12 (UniPC) rows
    for (int i = 0; i < 3; ++i) M.hist.push_back(B.persistent_t(Bn, h, w, Lc, 4));   // UniPC multistep state
    M.ts_dev = B.persistent_t(1, 1, 1, TS, 4);
    M.image = B.persistent_t(Bn, h * sc, w * sc, 3, 4);            // NCHW fp32 (the T fields only carry the size)
    {   // text K/V projections, batch-concatenated in group order [net groups..., UNet] (StepRunner.set_context)
      const int n_enc = M.unet.n_enc_tr, T77 = M.ehs.h;
      for (int ti = 0; ti < n_enc; ++ti) M.ctx_grouped.push_back(B.persistent_t(M.ntot, T77, 1, M.unet.tr[ti].kv2->cout));
      int a = 0;
      for (size_t gi = 0; gi < M.groups.size(); ++gi) {
        const int kN = M.counts[gi];
        M.ctx_nets.emplace_back();
        for (int ti = 0; ti < n_enc; ++ti) M.ctx_nets.back().push_back(M.ctx_grouped[ti].batch(a, kN));
        a += kN;
      }
      for (int ti = 0; ti < n_enc; ++ti) M.ctx_unet.push_back(M.ctx_grouped[ti].batch(a, N));
      for (size_t ti = n_enc; ti < M.unet.tr.size(); ++ti) M.ctx_unet.push_back(B.persistent_t(N, T77, 1, M.unet.tr[ti].kv2->cout));
      if (M.guess)                                                 // StepRunner.set_context(guess_mode=True): each group's own K/V rows
        for (const auto& gp : M.groups) {
          M.ctx_guess.emplace_back();
          for (int ti = 0; ti < n_enc; ++ti) M.ctx_guess.back().push_back(B.persistent_t((int)gp.second.size() * Nc, T77, 1, M.nets[gp.first]->tr[ti].kv2->cout));
        }
    }
    M.cond_cat = B.persistent_t(M.ntot, h, w, c0);                 // the UNet's slot stays zero
    M.tproj_table = B.persistent_t(TS, M.ntot, 1, M.width);         // columns beyond a group's width are never written: zero
    M.tproj_cur = B.persistent_t(M.ntot, 1, 1, M.width);
    M.tproj_gen = B.persistent_t(M.ntot, 1, 1, M.width);
    if (NN != 1 && !M.guess)                                       // StepRunner.prepare_fused_zero (constants, filled below)
      for (const auto& fp : M.fusion) M.fused_zero.push_back(B.persistent_t(N, fp.s * fp.s, 1, fp.c));
    // es_prepare_conds inputs: one image batch per shared encoder (the VAE of the LoRA nets first, NativeEngine._conds_fn)
    M.cond_img.resize(NN); M.cond_noise.resize(NN);
    {
      Model::CondGroup vg; vg.vae = true; vg.net = -1;
      for (int i = 0; i < NN; ++i) if (M.nets[M.net_of_cond[i]]->uses_vae) vg.idx.push_back(i);
      std::vector<Model::CondGroup> order;
      bool vae_placed = false;
      for (int i = 0; i < NN; ++i) {
        const int ni = M.net_of_cond[i];
        if (M.nets[ni]->uses_vae) { if (!vae_placed) { order.push_back(vg); vae_placed = true; } continue; }
        bool found = false;
        for (auto& cg : order) if (!cg.vae && cg.net == ni) { cg.idx.push_back(i); found = true; }
        if (!found) { Model::CondGroup cg; cg.vae = false; cg.net = ni; cg.idx.push_back(i); order.push_back(cg); }
      }
      for (auto& cg : order) {
        cg.gb = B.persistent_t((int)cg.idx.size() * Bn, h * sc, w * sc, 3, 4);     // NCHW fp32 [n,3,H,W]
        for (size_t k = 0; k < cg.idx.size(); ++k) {
          T im = cg.gb.batch((int)k * Bn, Bn);
          M.cond_img[cg.idx[k]] = im;
          if (cg.vae) M.cond_noise[cg.idx[k]] = B.persistent_t(Nc, h, w, v.latent, 4);  // NCHW fp32 [N,L,h,w]
        }
        M.cgroups.push_back(cg);
      }
    }
    // ---- the five launch lists
    for (int i = 0; i < ES_PLAN_COUNT; ++i) plans[i] = es_plan_create();
    if (M.guess) {
      // guess_mode: no lockstep tables - the ControlNets run on another batch than the UNet (NativeEngine with guess_mode=True).
      // ES_PLAN_STEP_UNET stays empty: steps outside the control-guidance window run ES_PLAN_STEP with all scales 0
      { Recording r(plans[ES_PLAN_PREP]); M.set_context_guess(); }
      { Recording r(plans[ES_PLAN_STEP]); M.one_step(); }
      { Recording r(plans[ES_PLAN_STEP_GENERIC]); M.set_context_guess(); M.step_guess(); }
    } else {
      { Recording r(plans[ES_PLAN_PREP]); M.set_context(); M.set_conds(); M.set_time_table(); }
      { Recording r(plans[ES_PLAN_STEP]); M.one_step(); }
      { Recording r(plans[ES_PLAN_STEP_GENERIC]); M.set_context(); M.set_conds(); M.step(false); }
      { Recording r(plans[ES_PLAN_STEP_UNET]); M.one_step_unet(); }
    }
    { Recording r(plans[ES_PLAN_DECODE]); M.decode(); }
    { Recording r(plans[ES_PLAN_CONDS]); M.embed_conds(); }
    // zero inputs and scratch of the ONE real launch that fills the fusion-of-zeros constants once the arena exists (the arena is
    // zero-filled and nothing has run in it by then: any block of the activation heap reads as zeros)
    if (NN != 1 && !M.guess)
      for (const auto& fp : M.fusion) { M.fz_zero.push_back(B.empty(N, fp.s * fp.s, 1, fp.c)); M.fz_u.push_back(B.empty(N, fp.s * fp.s, 1, fp.c)); }
    // ---- the arena: allocate, relocate, upload
    const unsigned long long heap_bytes = (B.heap.top + 255) & ~255ull, total = heap_bytes + ((B.ws_bytes + 255) & ~255ull);
    struct Map { unsigned long long heap_base, ws_base, heap_bytes, ws_bytes; bool bad; } mp{0, 0, heap_bytes, B.ws_bytes, false};
    if (device >= 0) {
      if (hipSetDevice(device) != hipSuccess) fail("es_load_weights: hipSetDevice failed");
      if (hipMalloc(&arena, total ? total : 256) != hipSuccess) fail("es_load_weights: hipMalloc of the arena (" + std::to_string(total >> 20) + " MiB) failed");
      if (hipMemset(arena, 0, total ? total : 256) != hipSuccess) fail("es_load_weights: hipMemset of the arena failed");
      mp.heap_base = (unsigned long long)arena;
    } else if (device == -2) {                // inspection build: the arena in (lazily committed) host memory, contents included
      arena = calloc(total ? total : 256, 1);
      if (!arena) fail("es_load_weights: host arena allocation failed");
      mp.heap_base = (unsigned long long)arena;
    } else mp.heap_base = FAKE_HEAP;          // dry build (no device): the plans keep their arena-relative addresses
    mp.ws_base = mp.heap_base + heap_bytes;
    auto reloc = [](unsigned long long a, int, void* user) -> unsigned long long {
      Map* q = (Map*)user;
      if (a >= FAKE_WS) { if (a - FAKE_WS > q->ws_bytes) q->bad = true; return q->ws_base + (a - FAKE_WS); }
      if (a < FAKE_HEAP || a - FAKE_HEAP >= q->heap_bytes) { q->bad = true; return a; }
      return q->heap_base + (a - FAKE_HEAP);
    };
    for (int i = 0; i < ES_PLAN_COUNT; ++i) if (es_plan_relocate(plans[i], reloc, &mp) || mp.bad) fail("es_load_weights: a recorded pointer lies outside the arena");
    auto real = [&](const T& t) { return (void*)(mp.heap_base + (t.p - FAKE_HEAP)); };
    std::vector<size_t> up_sizes;
    for (const auto& up : B.uploads) up_sizes.push_back(up.second.size());
    if (device >= 0)
      for (auto& up : B.uploads) {
        if (hipMemcpy((void*)(mp.heap_base + (up.first - FAKE_HEAP)), up.second.data(), up.second.size(), hipMemcpyHostToDevice) != hipSuccess) fail("es_load_weights: copy to the device failed");
        std::vector<char>().swap(up.second);
      }
    if (device == -2)
      for (auto& up : B.uploads) memcpy((void*)(mp.heap_base + (up.first - FAKE_HEAP)), up.second.data(), up.second.size());
    if (device >= 0 && NN != 1 && !M.guess) {
      // the constants ES_PLAN_STEP_UNET adds: ControlNetBlock(interleave(zeros)) of every level (MC:151-169 with all scales 0)
      std::vector<es_fusion_desc> fd(M.fusion.size());
      for (size_t k = 0; k < fd.size(); ++k) {
        const FusionParams& fp = M.fusion[k];
        auto it = B.fusion_scratch.find(std::make_pair(N, (int)k));
        if (it == B.fusion_scratch.end()) fail("es_load_weights: no fusion scratch for the fusion-of-zeros launch");
        auto rp = [&](unsigned long long a) { return (void*)(mp.heap_base + (a - FAKE_HEAP)); };
        es_fusion_desc& d = fd[k];
        memset(&d, 0, sizeof(d));
        for (int i = 0; i < 6; ++i) { d.res[i] = real(M.fz_zero[k]); d.res_bs[i] = M.fz_zero[k].bstride(); d.res_scale[i] = 0.f; }
        d.w1 = (const float*)rp(fp.w1); d.b1 = (const float*)rp(fp.b1); d.g1 = rp(fp.g1); d.be1 = rp(fp.be1);
        d.w2 = (const float*)rp(fp.w2); d.b2 = (const float*)rp(fp.b2); d.g2 = rp(fp.g2); d.be2 = rp(fp.be2);
        d.w3 = (const float*)rp(fp.w3); d.b3 = (const float*)rp(fp.b3);
        d.scratch = (float*)rp(it->second); d.u = real(M.fz_u[k]); d.out = real(M.fused_zero[k]);
        d.N = N; d.HW = fp.s * fp.s; d.C = fp.c; d.eps = 1e-5f; d.dtype = B.dt;
      }
      if (es_fusion_blocks(fd.data(), (int)fd.size(), nullptr) || hipDeviceSynchronize() != hipSuccess) fail(std::string("es_load_weights: the fusion-of-zeros launch failed: ") + es_last_error());
      // the scratch and the intermediate of that launch are activation space again: back to zeros, as a fresh arena has them
      for (size_t k = 0; k < fd.size(); ++k)
        if (hipMemset(real(M.fz_u[k]), 0, M.fz_u[k].bytes()) != hipSuccess) fail("es_load_weights: hipMemset failed");
    }
    if (es_ctx_create(device < 0 ? 0 : device, &ctx)) fail("es_load_weights: es_ctx_create failed");
    es_ctx_adopt_arena(ctx, arena, (size_t)total, device == -2);
    arena = nullptr;
    for (const auto& up : B.uploads) es_ctx_add_extent(ctx, up.first - FAKE_HEAP, up.second.size() ? up.second.size() : up_sizes[&up - &B.uploads[0]]);
    for (const auto& fz : M.fused_zero) es_ctx_add_extent(ctx, fz.p - FAKE_HEAP, fz.bytes());       // constants computed on the device: they travel with an image
    es_ctx_geometry geo = *g;
    geo.latent_channels = Lc; geo.latent_pad = Lp;
    if (es_ctx_set_geometry(ctx, &geo)) fail(std::string("es_load_weights: ") + es_last_error());
    for (int i = 0; i < ES_PLAN_COUNT; ++i) {
      if (M.guess && i == ES_PLAN_STEP_UNET) continue;       // (not recorded in guess_mode: destroyed with the other leftovers below)
      if (es_ctx_set_plan(ctx, i, plans[i])) fail(std::string("es_load_weights: ") + es_last_error());
      plans[i] = nullptr;
    }
    auto bind = [&](int slot, const T& t) { if (es_ctx_bind(ctx, slot, real(t), t.bytes())) fail(std::string("es_load_weights: ") + es_last_error()); };
    bind(ES_BUF_SAMPLE, M.model_in); bind(ES_BUF_T_ROWS, M.t_rows); bind(ES_BUF_EHS, M.ehs); bind(ES_BUF_SCALES, M.scales_cur);
    bind(ES_BUF_NOISE, M.noise); bind(ES_BUF_LATENTS, M.latents); bind(ES_BUF_STEP_IDX, M.step_idx); bind(ES_BUF_T_TABLE, M.t_table);
    bind(ES_BUF_SCALE_TABLE, M.scale_table); bind(ES_BUF_COEF, M.coef); bind(ES_BUF_TIMESTEPS, M.ts_dev); bind(ES_BUF_IMAGE, M.image);
    for (int i = 0; i < 3; ++i) bind(ES_BUF_HIST0 + i, M.hist[i]);
    for (int i = 0; i < NN; ++i) {
      bind(ES_BUF_COND0 + i, M.conds[i]);
      bind(ES_BUF_COND_IMG0 + i, M.cond_img[i]);
      if (M.cond_noise[i]) bind(ES_BUF_COND_NOISE0 + i, M.cond_noise[i]);
    }
    for (auto*& p : plans) if (p) { es_plan_destroy(p); p = nullptr; }
    *out = ctx;
    return 0;
  } catch (const std::exception& e) {
    es_plan_set_dry(0);
    for (auto* p : plans) if (p) es_plan_destroy(p);
    if (ctx) es_ctx_destroy(ctx);
    if (arena) { if (device == -2) free(arena); else (void)hipFree(arena); }
    es_set_error(e.what());
    return -1;
  }
}
