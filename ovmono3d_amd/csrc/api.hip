// C ABI of libovm3d (see include/ovm3d.h): handle, weight packing, forward orchestration.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/ovm3d.h"
#include "kernels.hpp"
#include "det2d.hpp"
#include "gdino.hpp"

using namespace ovm;

static_assert(sizeof(OvmDet3D) == kRecFloats * 4, "record layout");

namespace {

struct Split {                 // device fp16 split tensor
  half_t* hi = nullptr; half_t* lo = nullptr;
};

struct PackedLinear { Split w; float* bias = nullptr; int N = 0, K = 0; };

struct Layer {
  float *ln1g, *ln1b, *ln2g, *ln2b, *ls1, *ls2;
  PackedLinear qkv, proj, fc1, fc2;
  // SAM tower: window side of the block (0 = global attention) and its relative-position tables [2 s - 1][64] (s = window side,
  // or the canvas grid for a global block: resized at create when the checkpoint's table has another length)
  int ws = 0; float *relh = nullptr, *relw = nullptr;
};

struct SfpStage {              // 1x1 conv + LN, 3x3 conv + LN
  PackedLinear c1, c3; float *n1g, *n1b, *n3g, *n3b;
};

struct FpnLevel {              // one output level of the simple feature pyramid, finest first ("p2", "p3", ...)
  int side = 0; float stride = 0.f;
  SfpStage st;
  float* T1 = nullptr;         // fp32 conv output (1x1, then reused by the 3x3)
  Split pad;                   // zero-bordered LN(1x1) image: the implicit-GEMM input of the 3x3
  float* p = nullptr;          // the level's feature map, NHWC fp32
  Split rpad;                  // zero-bordered fp16 copy of p (RPN conv input), when the checkpoint has an RPN
};

}  // namespace

struct OvmHandle {
  OvmConfig cfg;
  int device = 0;
  std::string err;
  std::vector<void*> allocs;
  int G = 0, G2 = 0, T = 0, Tpad = 0, D = 0, C = 0, Kpe = 640, npass = 1;
  int patch = 14, nlev = 3;          // by tower: 14 / 3 levels (DINOv2, scales 2 1 0.5) or 16 / 4 levels (CLIP, scales 4 2 1 0.5)
  float ln_eps = 1e-6f;             // LayerNorm eps of the ViT blocks (1e-6 dinov2, 1e-5 open_clip)
  int mlp_act = 0;                  // fc1 activation: 0 erf-GELU, 3 QuickGELU
  int roiK = 0;
  // weights
  PackedLinear pe; float *cls = nullptr, *pos = nullptr;
  float *lnpre_g = nullptr, *lnpre_b = nullptr;                 // open_clip ln_pre
  // SAM tower (windowed blocks run on window-partitioned rows like the Swin backbone of the detector)
  bool sam = false; int sam_ws = 0, sam_nw = 0, sam_rows = 0;   // window side, windows per image, rows per image of the partitioned layout
  int* sam_map = nullptr;                                       // [max_batch][sam_rows]: token row of X, -1 = padding
  Split XW, CTX; float *QKVF = nullptr, *RELH = nullptr, *RELW = nullptr; int ldrel = 0;
  std::vector<Layer> layers;
  PackedLinear dfuse; bool has_dfuse = false;
  PackedLinear convt;                                           // ConvT D -> D/2 (first layer of the scale-2 and scale-4 stages... per stage)
  PackedLinear convt4a, convt4b; float *up_ln_g = nullptr, *up_ln_b = nullptr;   // scale-4 stage: ConvT D -> D/2, LN, GELU, ConvT D/2 -> D/4
  FpnLevel lv[kMaxLevels];
  PackedLinear cube_fc1, cube_fc2, cube_out;
  PackedLinear box_fc1, box_fc2, box_out; bool has_box = false;
  PackedLinear rpn_conv, rpn_out; bool has_rpn = false;
  // workspace
  float* X = nullptr;
  Split PA, HN, AO, F1, Q, Kx, Vt, DT, DT4, DF, CT, CT4a, CT4b;
  float *dtok = nullptr, *FUS = nullptr;
  Split RF, H1, H2; float* HO = nullptr; int lastN = 0;
  float* attn_tail_ws = nullptr; int* attn_tail_cnt = nullptr;     // attention's leftover-query partials / arrival counters (attn_tail.hpp)
  float* splitk_ws = nullptr; size_t splitk_cap = 0;               // split-K partials of this handle's thin GEMMs (two handles on two streams
                                                                   // must not share the launcher's process-global, re-sizable workspace)
  float* rec = nullptr; int* keep = nullptr;
  int *d_bidx = nullptr;
  ImageDesc* d_imgs = nullptr; ImageMeta* d_meta = nullptr;
  ImageDesc* h_imgs = nullptr; ImageMeta* h_meta = nullptr;     // pinned staging
  Det2dWorkspace det;
  int lastB = 0;
  // optional per-kernel-category timing with HIP events on the caller's stream
  bool prof = false;
  unsigned prof_mask = ~0u;          // categories that are bracketed while prof is on
  bool corun = false;               // ovm_set_corun
  std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_ev[OVM_PROF_NCAT];
  size_t prof_used[OVM_PROF_NCAT] = {0};
  // ovm_infer: side stream of the text-prompted detector + 2D detection buffers (capacity = the detector's query count)
  hipStream_t side = nullptr; hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  float *inf_boxes = nullptr, *inf_scores = nullptr; int *inf_classes = nullptr, *inf_n = nullptr, *inf_idx = nullptr, *inf_counts = nullptr;
  int inf_cap = 0;
  void* inf_ws = nullptr; size_t inf_ws_bytes = 0;      // scratch of the GroundingDINO output glue (launch_gdino_post_ws)
  int* inf_host = nullptr;                              // pinned: kept 2D count, record count
};

namespace {

#define HCHECK(h, call)                                                                    \
  do {                                                                                     \
    hipError_t e_ = (call);                                                                \
    if (e_ != hipSuccess) {                                                                \
      (h)->err = std::string(#call) + ": " + hipGetErrorString(e_);                        \
      return OVM_ERR_HIP;                                                                  \
    }                                                                                      \
  } while (0)

#define KCHECK(h, call)                                                                    \
  do {                                                                                     \
    int r_ = (call);                                                                       \
    if (r_ != OVM_OK) {                                                                    \
      if ((h)->err.empty() || r_ != OVM_ERR_HIP) (h)->err = std::string(#call) + " failed (" + std::to_string(r_) + ")"; \
      return r_;                                                                           \
    }                                                                                      \
  } while (0)

template <typename Tp>
int dalloc(OvmHandle* h, Tp** p, size_t count, bool zero = false) {
  void* q = nullptr;
  size_t bytes = count * sizeof(Tp);
  if (bytes == 0) bytes = 16;
  HCHECK(h, hipMalloc(&q, bytes));
  h->allocs.push_back(q);
  if (zero) HCHECK(h, hipMemset(q, 0, bytes));
  *p = (Tp*)q;
  return OVM_OK;
}

// interleaved split image [rows][K/32][hi 32 | lo 32] (f16x3 mode; plain fp16 rows in one-pass mode): the A-operand layout of the
// 256 x 256 GEMM (one 128-byte LDS-DMA line per row and k-group holds both parts)
int salloc_il(OvmHandle* h, Split* s, size_t count) {
  if (h->npass != 3) { s->lo = nullptr; return dalloc(h, &s->hi, count); }
  int r = dalloc(h, &s->hi, 2 * count);
  if (r) return r;
  s->lo = s->hi + 32;
  return OVM_OK;
}

int salloc(OvmHandle* h, Split* s, size_t count, bool zero = false) {
  int r = dalloc(h, &s->hi, count, zero);
  if (r) return r;
  if (h->npass == 3) return dalloc(h, &s->lo, count, zero);
  s->lo = nullptr;
  return OVM_OK;
}

struct WeightMap {
  std::map<std::string, const OvmTensor*> m;
  const OvmTensor* get(const std::string& k) const {
    auto it = m.find(k);
    return it == m.end() ? nullptr : it->second;
  }
};

int64_t numel(const OvmTensor* t) { int64_t n = 1; for (int i = 0; i < t->ndim; ++i) n *= t->shape[i]; return n; }

int upload_f32(OvmHandle* h, const WeightMap& wm, const std::string& key, int64_t expect, float** out) {
  const OvmTensor* t = wm.get(key);
  if (!t) { h->err = "missing weight: " + key; return OVM_ERR_MISSING_WEIGHT; }
  if (numel(t) != expect) { h->err = "bad shape for " + key; return OVM_ERR_SHAPE; }
  int r = dalloc(h, out, (size_t)expect);
  if (r) return r;
  HCHECK(h, hipMemcpy(*out, t->data, (size_t)expect * 4, hipMemcpyHostToDevice));
  return OVM_OK;
}

int upload_vec(OvmHandle* h, const std::vector<float>& v, float** out) {
  int r = dalloc(h, out, v.size());
  if (r) return r;
  HCHECK(h, hipMemcpy(*out, v.data(), v.size() * 4, hipMemcpyHostToDevice));
  return OVM_OK;
}

// Pack a host [N][K] fp32 matrix (already in GEMM k-order) into device fp16: one-pass mode [Npad][Kpad]; split mode the
// interleaved image [Npad][Kpad/32][hi 32 | lo 32] the GEMM kernels stream (gemm.hpp), w.lo = w.hi + 32.
int upload_packed(OvmHandle* h, const std::vector<float>& w, int N, int K, int Kpad, PackedLinear* out) {
  const int Npad = (N + 127) / 128 * 128;
  const bool il = h->npass == 3;
  if (il && Kpad % 32 != 0) { h->err = "packed K must be a multiple of 32"; return OVM_ERR_SHAPE; }
  const size_t ld = il ? (size_t)2 * Kpad : (size_t)Kpad;
  std::vector<half_t> buf((size_t)Npad * ld, (half_t)0.f);
  for (int n = 0; n < N; ++n)
    for (int k = 0; k < K; ++k) {
      const float x = w[(size_t)n * K + k];
      const half_t hh = (half_t)x;
      if (il) {
        const size_t o = (size_t)n * ld + (size_t)(k >> 5) * 64 + (k & 31);
        buf[o] = hh;
        buf[o + 32] = (half_t)((x - (float)hh) * kLoScale);
      } else {
        buf[(size_t)n * ld + k] = hh;
      }
    }
  int r = dalloc(h, &out->w.hi, buf.size());
  if (r) return r;
  HCHECK(h, hipMemcpy(out->w.hi, buf.data(), buf.size() * 2, hipMemcpyHostToDevice));
  out->w.lo = il ? out->w.hi + 32 : nullptr;
  out->N = N; out->K = Kpad;
  return OVM_OK;
}

int get_host(OvmHandle* h, const WeightMap& wm, const std::string& key, int64_t expect, const float** p) {
  const OvmTensor* t = wm.get(key);
  if (!t) { h->err = "missing weight: " + key; return OVM_ERR_MISSING_WEIGHT; }
  if (numel(t) != expect) { h->err = "bad shape for " + key + " (expected " + std::to_string(expect) + ")"; return OVM_ERR_SHAPE; }
  *p = t->data;
  return OVM_OK;
}

// nn.Linear weight [N][K] (+ optional bias) -> packed
int pack_linear(OvmHandle* h, const WeightMap& wm, const std::string& prefix, int N, int K, PackedLinear* out,
                bool bias = true, int Kpad = -1) {
  const float* w; int r = get_host(h, wm, prefix + ".weight", (int64_t)N * K, &w);
  if (r) return r;
  std::vector<float> v(w, w + (size_t)N * K);
  r = upload_packed(h, v, N, K, Kpad < 0 ? K : Kpad, out);
  if (r) return r;
  if (bias) return upload_f32(h, wm, prefix + ".bias", N, &out->bias);
  out->bias = nullptr;
  return OVM_OK;
}

// conv weight [Cout][Cin][k][k] -> [Cout][(ky*k+kx)*Cin + c]
int pack_conv(OvmHandle* h, const WeightMap& wm, const std::string& prefix, int Cout, int Cin, int k, PackedLinear* out,
              bool bias) {
  const float* w; int r = get_host(h, wm, prefix + ".weight", (int64_t)Cout * Cin * k * k, &w);
  if (r) return r;
  std::vector<float> v((size_t)Cout * Cin * k * k);
  for (int o = 0; o < Cout; ++o)
    for (int c = 0; c < Cin; ++c)
      for (int t = 0; t < k * k; ++t) v[((size_t)o * k * k + t) * Cin + c] = w[((size_t)o * Cin + c) * k * k + t];
  r = upload_packed(h, v, Cout, Cin * k * k, Cin * k * k, out);
  if (r) return r;
  if (bias) return upload_f32(h, wm, prefix + ".bias", Cout, &out->bias);
  out->bias = nullptr;
  return OVM_OK;
}

// FC over pooled RoI features: reference flatten order is (c, ph, pw); ROIAlign here emits (ph, pw, c)
int pack_roi_fc(OvmHandle* h, const WeightMap& wm, const std::string& prefix, int N, int C, int res, PackedLinear* out) {
  const int K = C * res * res;
  const float* w; int r = get_host(h, wm, prefix + ".weight", (int64_t)N * K, &w);
  if (r) return r;
  std::vector<float> v((size_t)N * K);
  for (int n = 0; n < N; ++n)
    for (int c = 0; c < C; ++c)
      for (int s = 0; s < res * res; ++s) v[(size_t)n * K + (size_t)s * C + c] = w[(size_t)n * K + (size_t)c * res * res + s];
  r = upload_packed(h, v, N, K, K, out);
  if (r) return r;
  return upload_f32(h, wm, prefix + ".bias", N, &out->bias);
}

// several small nn.Linear heads sharing one input, concatenated along N
int pack_concat(OvmHandle* h, const WeightMap& wm, const std::vector<std::pair<std::string, int>>& parts, int K,
                PackedLinear* out) {
  int N = 0; for (auto& p : parts) N += p.second;
  std::vector<float> v((size_t)N * K), b((size_t)N);
  int n0 = 0;
  for (auto& p : parts) {
    const float *w, *bb;
    int r = get_host(h, wm, p.first + ".weight", (int64_t)p.second * K, &w); if (r) return r;
    r = get_host(h, wm, p.first + ".bias", p.second, &bb); if (r) return r;
    memcpy(&v[(size_t)n0 * K], w, (size_t)p.second * K * 4);
    memcpy(&b[n0], bb, (size_t)p.second * 4);
    n0 += p.second;
  }
  int r = upload_packed(h, v, N, K, K, out);
  if (r) return r;
  return upload_vec(h, b, &out->bias);
}

// nn.Linear stored as bare parameters (nn.MultiheadAttention in_proj_weight / in_proj_bias)
int pack_linear_named(OvmHandle* h, const WeightMap& wm, const std::string& wkey, const std::string& bkey, int N, int K, PackedLinear* out) {
  const float* w; int r = get_host(h, wm, wkey, (int64_t)N * K, &w);
  if (r) return r;
  std::vector<float> v(w, w + (size_t)N * K);
  r = upload_packed(h, v, N, K, K, out);
  if (r) return r;
  return upload_f32(h, wm, bkey, N, &out->bias);
}

// ConvTranspose2d k2 s2 weight [Cin][Cout][2][2] -> GEMM rows [(a*2+b)*Cout + co][ci]; bias [Cout] (applied per co in EPI_CONVT)
int pack_convt(OvmHandle* h, const WeightMap& wm, const std::string& prefix, int Cin, int Cout, PackedLinear* out) {
  const float* w; int r = get_host(h, wm, prefix + ".weight", (int64_t)Cin * Cout * 4, &w);
  if (r) return r;
  std::vector<float> v((size_t)4 * Cout * Cin);
  for (int ci = 0; ci < Cin; ++ci)
    for (int co = 0; co < Cout; ++co)
      for (int q = 0; q < 4; ++q) v[((size_t)q * Cout + co) * Cin + ci] = w[((size_t)ci * Cout + co) * 4 + q];
  r = upload_packed(h, v, 4 * Cout, Cin, Cin, out);
  if (r) return r;
  return upload_f32(h, wm, prefix + ".bias", Cout, &out->bias);
}

int pack_sfp_stage(OvmHandle* h, const WeightMap& wm, const std::string& p1, const std::string& p3, int Cin, SfpStage* s) {
  const int C = h->C;
  int r = pack_conv(h, wm, p1, C, Cin, 1, &s->c1, false); if (r) return r;
  r = upload_f32(h, wm, p1 + ".norm.weight", C, &s->n1g); if (r) return r;
  r = upload_f32(h, wm, p1 + ".norm.bias", C, &s->n1b); if (r) return r;
  r = pack_conv(h, wm, p3, C, C, 3, &s->c3, false); if (r) return r;
  r = upload_f32(h, wm, p3 + ".norm.weight", C, &s->n3g); if (r) return r;
  return upload_f32(h, wm, p3 + ".norm.bias", C, &s->n3b);
}

struct ProfScope {
  OvmHandle* h; int cat; hipStream_t s; hipEvent_t stop = nullptr;
  ProfScope(OvmHandle* h_, int cat_, hipStream_t s_) : h(h_), cat(cat_), s(s_) {
    if (!h->prof || cat < 0 || !((h->prof_mask >> cat) & 1u)) return;
    auto& pool = h->prof_ev[cat];
    if (h->prof_used[cat] == pool.size()) {
      hipEvent_t a, b;
      if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
      pool.push_back({a, b});
    }
    auto& pr = pool[h->prof_used[cat]++];
    (void)hipEventRecord(pr.first, s);
    stop = pr.second;
  }
  ~ProfScope() { if (stop) (void)hipEventRecord(stop, s); }
};

int g_use_gemm256 = 1;     // ovm_tune_set("gemm256", 0 | 1)
int g_gemm256_ksplit = 0;  // ovm_tune_set("gemm256_ksplit", n)

int gemm(OvmHandle* h, const GemmParams& p_in, int epi, int amode, hipStream_t s, int cat = -1) {
  ProfScope ps(h, cat, s);
  GemmParams p = p_in;
  if (!p.part_ws && h->splitk_ws) { p.part_ws = h->splitk_ws; p.part_cap = h->splitk_cap; }
  // the large block contractions (qkv, fc1: >= 2048 rows and >= 3072 columns -> at least 192 tiles of 256 x 256) go to the
  // two-wave-group 256 x 256 kernel; everything else keeps the 128 x 128 kernels
  // (at batch >= 4 the N = D contractions - proj, fc2 - reach that tile count too: 128 x 128 tiles fetch twice the operand bytes
  // per MFMA from L2, which is what bounds them at batch 1, where only 128-wide tiles fill the chip)
  if (g_use_gemm256 && amode == A_ROWMAJOR && gemm256_supported(p, h->npass) &&
      (long)((p.M + 255) / 256) * (p.N / 256) >= 192 && (epi == EPI_STORE || epi == EPI_RESID || epi == EPI_GELU || epi == EPI_QKV))
    return launch_gemm256(p, epi, 1, s);
  // experiment knob (ovm_tune_set "gemm256_ksplit"): the long-K contractions with too few 256-wide tiles (fc2 at batch 1: 64 tiles,
  // K = 4096) as k-slices of 256 x 256 tiles + a reduce pass - half the operand fetch of 128 x 128 tiles
  if (g_gemm256_ksplit > 1 && amode == A_ROWMAJOR && gemm256_supported(p, h->npass) && epi == EPI_RESID && p.K >= 2048 && !p.row_map)
    return launch_gemm256(p, epi, g_gemm256_ksplit, s);
  return launch_gemm(p, h->npass, epi, amode, s);
}

GemmParams gp_base(const Split& A, int lda, const PackedLinear& W, int M) {
  GemmParams p; memset(&p, 0, sizeof(p));
  p.Ahi = A.hi; p.Alo = A.lo; p.lda = lda;
  p.Whi = W.w.hi; p.Wlo = W.w.lo;
  p.M = M; p.N = W.N; p.K = W.K; p.bias = W.bias;
  return p;
}

void fill_meta(OvmHandle* h, const OvmImage* images, int B) {
  for (int b = 0; b < B; ++b) {
    const OvmImage& im = images[b];
    h->h_imgs[b] = ImageDesc{im.data, im.height, im.width, im.stride_c, im.stride_h, im.stride_w};
    ImageMeta m;
    for (int i = 0; i < 9; ++i) m.K[i] = im.K[i];
    m.ratio = (float)((double)im.orig_height / (double)im.height);     // rcnn3d.py:92 (python float -> fp32 tensor)
    m.net_h = im.height; m.net_w = im.width; m.orig_h = im.orig_height; m.orig_w = im.orig_width;
    h->h_meta[b] = m;
  }
}

}  // namespace

namespace ovm { void set_use_gemm256(int v) { g_use_gemm256 = v; } void set_gemm256_ksplit(int v) { g_gemm256_ksplit = v; } }

static int backbone_launches(OvmHandle* h, int B, const float* prompt_depth, int depth_h, int depth_w, hipStream_t s);

extern "C" {

const char* ovm_version(void) { return "libovm3d 0.2 (gfx950)"; }

int ovm_abi_sizeof(const char* name) {
  if (!name) return -1;
  const std::string n(name);
  if (n == "OvmConfig") return (int)sizeof(OvmConfig);
  if (n == "OvmTensor") return (int)sizeof(OvmTensor);
  if (n == "OvmImage") return (int)sizeof(OvmImage);
  if (n == "OvmDet3D") return (int)sizeof(OvmDet3D);
  if (n == "OvmGdinoConfig") return (int)sizeof(OvmGdinoConfig);
  if (n == "OvmJpegInfo") return (int)sizeof(OvmJpegInfo);
  return -1;
}

const char* ovm_last_error(const OvmHandle* h) { return h ? h->err.c_str() : "null handle"; }

int ovm_destroy(OvmHandle* h) {
  if (!h) return OVM_OK;
  hipSetDevice(h->device);
  for (void* p : h->allocs) hipFree(p);
  if (h->h_imgs) hipHostFree(h->h_imgs);
  if (h->h_meta) hipHostFree(h->h_meta);
  if (h->inf_host) hipHostFree(h->inf_host);
  for (int c = 0; c < OVM_PROF_NCAT; ++c)
    for (auto& pr : h->prof_ev[c]) { hipEventDestroy(pr.first); hipEventDestroy(pr.second); }
  if (h->ev_fork) hipEventDestroy(h->ev_fork);
  if (h->ev_join) hipEventDestroy(h->ev_join);
  if (h->side) hipStreamDestroy(h->side);
  delete h;
  return OVM_OK;
}

int ovm_host_shard_range(int64_t n, int32_t rank, int32_t world, int64_t* begin, int64_t* end) {
  if (world <= 0 || rank < 0 || rank >= world || n < 0) return OVM_ERR_INVALID;
  // InferenceSampler._get_local_indices: shard sizes n//W + (r < n%W), contiguous
  const int64_t q = n / world, r = n % world;
  const int64_t b = rank * q + (rank < r ? rank : r);
  *begin = b;
  *end = b + q + (rank < r ? 1 : 0);
  return OVM_OK;
}

// PyTorch upsample_bicubic2d, align_corners=False, scale_factor given (so the source scale is
// 1/scale_factor, which is what dinov2's +0.1 offset relies on), A = -0.75, border indices clamped.
int ovm_host_interp_pos_embed(const float* pos, int32_t M, int32_t D, int32_t G, float* out) {
  if (M <= 0 || D <= 0 || G <= 0) return OVM_ERR_INVALID;
  memcpy(out, pos, (size_t)D * 4);
  if (G == M) { memcpy(out + D, pos + D, (size_t)M * M * D * 4); return OVM_OK; }
  const double sf = ((double)G + 0.1) / (double)M;                  // python: float(w0 + 0.1) / M (double)
  const float scale = (float)(1.0 / sf);
  auto coef = [](float t, float* w) {
    const float A = -0.75f;
    auto c1 = [&](float x) { return ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f; };
    auto c2 = [&](float x) { return ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A; };
    w[0] = c2(t + 1.f); w[1] = c1(t); w[2] = c1(1.f - t); w[3] = c2(2.f - t);
  };
  const float* src = pos + D;
  float* dst = out + D;
  for (int oy = 0; oy < G; ++oy) {
    const float ry = scale * ((float)oy + 0.5f) - 0.5f;
    const int iy = (int)floorf(ry);
    float wy[4]; coef(ry - (float)iy, wy);
    for (int ox = 0; ox < G; ++ox) {
      const float rx = scale * ((float)ox + 0.5f) - 0.5f;
      const int ix = (int)floorf(rx);
      float wx[4]; coef(rx - (float)ix, wx);
      float* o = dst + ((size_t)oy * G + ox) * D;
      for (int d = 0; d < D; ++d) o[d] = 0.f;
      for (int a = 0; a < 4; ++a) {
        int yy = iy - 1 + a; yy = yy < 0 ? 0 : (yy > M - 1 ? M - 1 : yy);
        float rowacc_w[4];
        for (int b = 0; b < 4; ++b) rowacc_w[b] = wx[b];
        // PyTorch evaluates cubic_interp1d along x first for each of the 4 rows, then along y
        for (int d = 0; d < D; ++d) {
          float acc = 0.f;
          for (int b = 0; b < 4; ++b) {
            int xx = ix - 1 + b; xx = xx < 0 ? 0 : (xx > M - 1 ? M - 1 : xx);
            acc += src[((size_t)yy * M + xx) * D + d] * rowacc_w[b];
          }
          o[d] += acc * wy[a];
        }
      }
    }
  }
  return OVM_OK;
}

// F.interpolate(pos[1,D,M,M], size=(G,G), mode="bicubic", align_corners=False, antialias=True) of the patch part of an
// open_clip positional embedding [1 + M*M][D], class row kept (reference clip.py:98-133). PyTorch's antialiased path is a
// separable, NORMALISED filter (not the clamped 4-tap one above): per output index, taps j in [xmin, xmin + xsize) with
// xmin = max(int(center - support + 0.5), 0), xsize = min(int(center + support + 0.5), M) - xmin, center = scale (i + 0.5),
// scale = M / G, support = 2 max(scale, 1), weight = cubic_{a = -0.5}((j - center + 0.5) / max(scale, 1)) / sum; all in fp32,
// width pass first, then height (ATen UpSampleKernel.cpp, _compute_indices_min_size_weights_aa). Returns pos unchanged when
// G == M (:117-118).
int ovm_host_resize_pos_embed_aa(const float* pos, int32_t M, int32_t D, int32_t G, float* out) {
  if (M <= 0 || D <= 0 || G <= 0) return OVM_ERR_INVALID;
  memcpy(out, pos, (size_t)D * 4);
  if (G == M) { memcpy(out + D, pos + D, (size_t)M * M * D * 4); return OVM_OK; }
  const float scale = (float)M / (float)G;
  const float support = scale >= 1.f ? 2.f * scale : 2.f;
  const float invscale = scale >= 1.f ? 1.f / scale : 1.f;
  const int max_taps = (int)ceilf(support) * 2 + 1;
  auto filt = [](float x) {
    const float a = -0.5f;
    x = fabsf(x);
    if (x < 1.f) return ((a + 2.f) * x - (a + 3.f)) * x * x + 1.f;
    if (x < 2.f) return ((a * x - 5.f * a) * x + 8.f * a) * x - 4.f * a;
    return 0.f;
  };
  std::vector<int> xmin(G), xsize(G);
  std::vector<float> wt((size_t)G * max_taps, 0.f);
  for (int i = 0; i < G; ++i) {
    const float center = scale * ((float)i + 0.5f);
    int lo = (int)(center - support + 0.5f); if (lo < 0) lo = 0;
    int hi = (int)(center + support + 0.5f); if (hi > M) hi = M;
    int n = hi - lo; if (n < 0) n = 0; if (n > max_taps) n = max_taps;
    xmin[i] = lo; xsize[i] = n;
    float total = 0.f;
    for (int j = 0; j < n; ++j) { const float w = filt(((float)(j + lo) - center + 0.5f) * invscale); wt[(size_t)i * max_taps + j] = w; total += w; }
    const float inv = total != 0.f ? 1.f / total : 0.f;
    for (int j = 0; j < n; ++j) wt[(size_t)i * max_taps + j] *= inv;
  }
  const float* src = pos + D;                              // [M][M][D]
  std::vector<float> tmp((size_t)M * G * D);               // width pass: [M][G][D]
  for (int y = 0; y < M; ++y)
    for (int ox = 0; ox < G; ++ox) {
      float* o = &tmp[((size_t)y * G + ox) * D];
      const float* w = &wt[(size_t)ox * max_taps];
      for (int d = 0; d < D; ++d) {
        float t = xsize[ox] > 0 ? src[((size_t)y * M + xmin[ox]) * D + d] * w[0] : 0.f;
        for (int j = 1; j < xsize[ox]; ++j) t += src[((size_t)y * M + xmin[ox] + j) * D + d] * w[j];
        o[d] = t;
      }
    }
  float* dst = out + D;
  for (int oy = 0; oy < G; ++oy) {
    const float* w = &wt[(size_t)oy * max_taps];
    for (int ox = 0; ox < G; ++ox) {
      float* o = dst + ((size_t)oy * G + ox) * D;
      for (int d = 0; d < D; ++d) {
        float t = xsize[oy] > 0 ? tmp[((size_t)xmin[oy] * G + ox) * D + d] * w[0] : 0.f;
        for (int j = 1; j < xsize[oy]; ++j) t += tmp[((size_t)(xmin[oy] + j) * G + ox) * D + d] * w[j];
        o[d] = t;
      }
    }
  }
  return OVM_OK;
}

}  // extern "C" (host helpers with internal linkage follow)

// F.interpolate(src[1,D,M,M], size=(G,G), mode="bicubic", align_corners=False) on a channels-last table [M*M][D] -> [G*G][D]
// (A = -0.75, border indices clamped, x pass then y as upsample_bicubic2d evaluates it); scale = the source step per output pixel
static int host_bicubic_grid(const float* src, int M, int D, int G, float scale, float* dst) {
  if (M <= 0 || D <= 0 || G <= 0) return OVM_ERR_INVALID;
  if (G == M) { memcpy(dst, src, (size_t)M * M * D * 4); return OVM_OK; }
  auto coef = [](float t, float* w) {
    const float A = -0.75f;
    auto c1 = [&](float x) { return ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f; };
    auto c2 = [&](float x) { return ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A; };
    w[0] = c2(t + 1.f); w[1] = c1(t); w[2] = c1(1.f - t); w[3] = c2(2.f - t);
  };
  for (int oy = 0; oy < G; ++oy) {
    const float ry = scale * ((float)oy + 0.5f) - 0.5f;
    const int iy = (int)floorf(ry);
    float wy[4]; coef(ry - (float)iy, wy);
    for (int ox = 0; ox < G; ++ox) {
      const float rx = scale * ((float)ox + 0.5f) - 0.5f;
      const int ix = (int)floorf(rx);
      float wx[4]; coef(rx - (float)ix, wx);
      float* o = dst + ((size_t)oy * G + ox) * D;
      for (int d = 0; d < D; ++d) o[d] = 0.f;
      for (int a = 0; a < 4; ++a) {
        int yy = iy - 1 + a; yy = yy < 0 ? 0 : (yy > M - 1 ? M - 1 : yy);
        for (int d = 0; d < D; ++d) {
          float acc = 0.f;
          for (int b = 0; b < 4; ++b) {
            int xx = ix - 1 + b; xx = xx < 0 ? 0 : (xx > M - 1 ? M - 1 : xx);
            acc += src[((size_t)yy * M + xx) * D + d] * wx[b];
          }
          o[d] += acc * wy[a];
        }
      }
    }
  }
  return OVM_OK;
}

// F.interpolate(src[1,C,L], size=Lo, mode="linear", align_corners=False) on rows [L][C] -> [Lo][C]
static void host_linear_rows(const float* src, int L, int C, int Lo, float* dst) {
  if (L == Lo) { memcpy(dst, src, (size_t)L * C * 4); return; }
  const float scale = (float)L / (float)Lo;
  for (int o = 0; o < Lo; ++o) {
    float f = scale * ((float)o + 0.5f) - 0.5f; if (f < 0.f) f = 0.f;
    int i0 = (int)f; if (i0 > L - 1) i0 = L - 1;
    const int i1 = i0 + (i0 < L - 1 ? 1 : 0);
    const float l1 = f - (float)i0, l0 = 1.f - l1;
    for (int c = 0; c < C; ++c) dst[(size_t)o * C + c] = l0 * src[(size_t)i0 * C + c] + l1 * src[(size_t)i1 * C + c];
  }
}

extern "C" {

int ovm_host_sincos_pos_embed(int32_t D, int32_t G, float* out) {
  if (D <= 0 || D % 4 != 0 || G <= 0) return OVM_ERR_INVALID;
  const int Q = D / 4;                                       // frequencies per (coordinate, sin / cos)
  std::vector<double> omega(Q);
  for (int i = 0; i < Q; ++i) omega[i] = 1.0 / pow(10000.0, (double)i / (double)Q);
  for (int d = 0; d < D; ++d) out[d] = 0.f;
  for (int y = 0; y < G; ++y)
    for (int x = 0; x < G; ++x) {
      float* o = out + (size_t)(1 + y * G + x) * D;
      // meshgrid(grid_w, grid_h): "grid[0]" is the x coordinate and feeds the FIRST half (named emb_h upstream)
      for (int i = 0; i < Q; ++i) {
        const double ax = (double)x * omega[i], ay = (double)y * omega[i];
        o[i] = (float)sin(ax); o[Q + i] = (float)cos(ax);
        o[2 * Q + i] = (float)sin(ay); o[3 * Q + i] = (float)cos(ay);
      }
    }
  return OVM_OK;
}

int ovm_create(const OvmConfig* cfg, const OvmTensor* weights, int32_t n_weights, int32_t device, OvmHandle** out) {
  if (!cfg || !out) return OVM_ERR_INVALID;
  OvmHandle* h = new OvmHandle();
  *out = h;
  h->cfg = *cfg; h->device = device;
  const OvmConfig& c = h->cfg;
  const bool clip = c.tower == OVM_TOWER_CLIP, mae = c.tower == OVM_TOWER_MAE, midas = c.tower == OVM_TOWER_MIDAS, sam = c.tower == OVM_TOWER_SAM;
  const bool p16 = clip || mae || midas || sam;           // patch-16 towers behind the 4-level pyramid
  h->sam = sam;
  if (sam && (c.sam_window < 1 || c.depth > 32 || c.pos_grid < 1)) { h->err = "invalid config (sam_window, depth <= 32, pos_grid)"; return OVM_ERR_INVALID; }
  if (c.tower != OVM_TOWER_DINOV2 && !p16) { h->err = "invalid config (tower)"; return OVM_ERR_INVALID; }
  h->patch = p16 ? 16 : 14; h->nlev = p16 ? 4 : 3;
  h->ln_eps = clip ? 1e-5f : (mae ? 1e-12f : 1e-6f); h->mlp_act = clip ? 3 : 0;
  h->Kpe = p16 ? 768 : 640;
  // the scale-4 stage needs D/4 channels in 64-wide k-steps
  if (c.canvas % h->patch != 0 || c.embed_dim % 128 != 0 || c.embed_dim != c.heads * 64 || (c.precision != 1 && c.precision != 3) ||
      c.fpn_channels % 64 != 0 || c.max_batch < 1 || c.max_rois < 1 || (p16 && c.embed_dim % 256 != 0)) {
    h->err = "invalid config (canvas % patch, embed_dim = heads*64 and %128 (%256 for 4-level towers), precision in {1,3}, fpn_channels %64)";
    return OVM_ERR_INVALID;
  }
  HCHECK(h, hipSetDevice(device));
  h->npass = c.precision;
  h->D = c.embed_dim; h->C = c.fpn_channels;
  h->G = c.canvas / h->patch; h->G2 = h->G * h->G; h->T = h->G2 + (sam ? 0 : 1); h->Tpad = (h->T + 63) / 64 * 64;
  const int D = h->D, C = h->C, G = h->G, G2 = h->G2, T = h->T, L = c.depth, B = c.max_batch, R = c.max_rois;
  {
    // The GEMM kernels address operands with 32-bit element offsets (gemm.hip: gemm_offsets_fit): refuse a max_batch / max_rois
    // whose largest activation image would not fit, here, before anything is allocated - not at the first oversized launch.
    const uint64_t il = h->npass == 3 ? 2 : 1;                       // interleaved split rows are 2K halves long
    const uint64_t side0 = (uint64_t)(p16 ? 4 : 2) * G + 2;          // finest pyramid level, zero-bordered
    const uint64_t worst[] = {(uint64_t)B * T * 4 * D * il,          // fc2's input (GELU output), the longest activation rows
                              (uint64_t)B * G2 * h->Kpe,             // patch rows
                              (uint64_t)B * side0 * side0 * C * 2,   // 3x3 implicit-GEMM image (int offsets, doubled: limit 2^31 elements)
                              (uint64_t)B * R * C * c.pooler_res * c.pooler_res};   // RoI features, fc1's input
    for (uint64_t w : worst)
      if (w > (1ull << 32)) {
        h->err = "max_batch / max_rois too large: an activation image would exceed the GEMM kernels' 32-bit element offsets";
        return OVM_ERR_CAPACITY;
      }
  }
  WeightMap wm;
  for (int i = 0; i < n_weights; ++i) wm.m[weights[i].name] = &weights[i];
  int r;
  const int P = h->patch, PP = P * P;
  const std::string V = clip ? "backbone.net.visual." : "backbone.net.vit.";
  const std::string PEW = clip ? "conv1.weight" : (mae ? "embeddings.patch_embeddings.projection.weight" : "patch_embed.proj.weight");
  // ---- patch embed: [D][3][P][P] -> [D][(py*P+px)*3 + c]; P = 14: K padded 588 -> 640
  {
    const float* w; r = get_host(h, wm, V + PEW, (int64_t)D * 3 * PP, &w); if (r) return r;
    std::vector<float> v((size_t)D * 3 * PP);
    for (int o = 0; o < D; ++o)
      for (int ch = 0; ch < 3; ++ch)
        for (int t = 0; t < PP; ++t) v[(size_t)o * 3 * PP + t * 3 + ch] = w[((size_t)o * 3 + ch) * PP + t];
    r = upload_packed(h, v, D, 3 * PP, h->Kpe, &h->pe); if (r) return r;
    const float* pos;
    std::vector<float> pi((size_t)T * D);
    if (clip) {                                            // conv1 has no bias (open_clip VisionTransformer)
      h->pe.bias = nullptr;
      r = upload_f32(h, wm, V + "class_embedding", D, &h->cls); if (r) return r;
      r = get_host(h, wm, V + "positional_embedding", (int64_t)(1 + c.pos_grid * c.pos_grid) * D, &pos); if (r) return r;
      r = ovm_host_resize_pos_embed_aa(pos, c.pos_grid, D, G, pi.data()); if (r) return r;
      r = upload_f32(h, wm, V + "ln_pre.weight", D, &h->lnpre_g); if (r) return r;
      r = upload_f32(h, wm, V + "ln_pre.bias", D, &h->lnpre_b); if (r) return r;
    } else if (sam) {                                      // segment_anything: no class token, table [grid][grid][D], plain bicubic resize (sam.py:73-86)
      r = upload_f32(h, wm, V + "patch_embed.proj.bias", D, &h->pe.bias); if (r) return r;
      r = get_host(h, wm, V + "pos_embed", (int64_t)c.pos_grid * c.pos_grid * D, &pos); if (r) return r;
      r = host_bicubic_grid(pos, c.pos_grid, D, G, (float)c.pos_grid / (float)G, pi.data()); if (r) return r;
    } else if (mae) {                                      // HF ViTMAE embeddings, position table rebuilt for this grid (mae.py:62-78)
      r = upload_f32(h, wm, V + "embeddings.patch_embeddings.projection.bias", D, &h->pe.bias); if (r) return r;
      r = upload_f32(h, wm, V + "embeddings.cls_token", D, &h->cls); if (r) return r;
      r = ovm_host_sincos_pos_embed(D, G, pi.data()); if (r) return r;
    } else {
      r = upload_f32(h, wm, V + "patch_embed.proj.bias", D, &h->pe.bias); if (r) return r;
      r = upload_f32(h, wm, V + "cls_token", D, &h->cls); if (r) return r;
      r = get_host(h, wm, V + "pos_embed", (int64_t)(1 + c.pos_grid * c.pos_grid) * D, &pos); if (r) return r;
      // timm ViT of MiDaS: the reference resizes with the CLIP tower's antialiased bicubic (midas_final.py:64-66); DINOv2: hub rule
      r = midas ? ovm_host_resize_pos_embed_aa(pos, c.pos_grid, D, G, pi.data()) : ovm_host_interp_pos_embed(pos, c.pos_grid, D, G, pi.data());
      if (r) return r;
    }
    r = upload_vec(h, pi, &h->pos); if (r) return r;
  }
  h->layers.resize(L);
  for (int l = 0; l < L; ++l) {
    Layer& y = h->layers[l];
    if (clip) {                                            // open_clip ResidualAttentionBlock: ln_1, attn (nn.MultiheadAttention), ln_2, mlp
      const std::string Pq = V + "transformer.resblocks." + std::to_string(l) + ".";
      if ((r = upload_f32(h, wm, Pq + "ln_1.weight", D, &y.ln1g))) return r;
      if ((r = upload_f32(h, wm, Pq + "ln_1.bias", D, &y.ln1b))) return r;
      if ((r = upload_f32(h, wm, Pq + "ln_2.weight", D, &y.ln2g))) return r;
      if ((r = upload_f32(h, wm, Pq + "ln_2.bias", D, &y.ln2b))) return r;
      y.ls1 = y.ls2 = nullptr;                             // no LayerScale
      if ((r = pack_linear_named(h, wm, Pq + "attn.in_proj_weight", Pq + "attn.in_proj_bias", 3 * D, D, &y.qkv))) return r;
      if ((r = pack_linear(h, wm, Pq + "attn.out_proj", D, D, &y.proj))) return r;
      if ((r = pack_linear(h, wm, Pq + "mlp.c_fc", 4 * D, D, &y.fc1))) return r;
      if ((r = pack_linear(h, wm, Pq + "mlp.c_proj", D, 4 * D, &y.fc2))) return r;
      continue;
    }
    if (sam) {                                             // segment_anything Block: norm1, attn (qkv, proj, rel_pos_h / _w), norm2, mlp (lin1, lin2)
      const std::string Pq = V + "blocks." + std::to_string(l) + ".";
      if ((r = upload_f32(h, wm, Pq + "norm1.weight", D, &y.ln1g))) return r;
      if ((r = upload_f32(h, wm, Pq + "norm1.bias", D, &y.ln1b))) return r;
      if ((r = upload_f32(h, wm, Pq + "norm2.weight", D, &y.ln2g))) return r;
      if ((r = upload_f32(h, wm, Pq + "norm2.bias", D, &y.ln2b))) return r;
      y.ls1 = y.ls2 = nullptr;
      if ((r = pack_linear(h, wm, Pq + "attn.qkv", 3 * D, D, &y.qkv))) return r;
      if ((r = pack_linear(h, wm, Pq + "attn.proj", D, D, &y.proj))) return r;
      if ((r = pack_linear(h, wm, Pq + "mlp.lin1", 4 * D, D, &y.fc1))) return r;
      if ((r = pack_linear(h, wm, Pq + "mlp.lin2", D, 4 * D, &y.fc2))) return r;
      y.ws = ((c.sam_global_mask >> l) & 1u) ? 0 : c.sam_window;
      const int side = y.ws ? y.ws : G;                    // the attention grid of this block
      for (int hw = 0; hw < 2; ++hw) {
        const std::string key = Pq + (hw ? "attn.rel_pos_w" : "attn.rel_pos_h");
        const OvmTensor* t = wm.get(key);
        if (!t || t->ndim != 2 || t->shape[1] != 64) { h->err = "missing or mis-shaped weight: " + key; return OVM_ERR_MISSING_WEIGHT; }
        std::vector<float> tab((size_t)(2 * side - 1) * 64);
        host_linear_rows(t->data, (int)t->shape[0], 64, 2 * side - 1, tab.data());     // get_rel_pos: F.interpolate(mode="linear") when lengths differ
        if ((r = upload_vec(h, tab, hw ? &y.relw : &y.relh))) return r;
      }
      continue;
    }
    if (mae) {                                             // HF ViTLayer: layernorm_before, attention (q / k / v / output.dense), layernorm_after, MLP
      const std::string Pq = V + "encoder.layer." + std::to_string(l) + ".";
      if ((r = upload_f32(h, wm, Pq + "layernorm_before.weight", D, &y.ln1g))) return r;
      if ((r = upload_f32(h, wm, Pq + "layernorm_before.bias", D, &y.ln1b))) return r;
      if ((r = upload_f32(h, wm, Pq + "layernorm_after.weight", D, &y.ln2g))) return r;
      if ((r = upload_f32(h, wm, Pq + "layernorm_after.bias", D, &y.ln2b))) return r;
      y.ls1 = y.ls2 = nullptr;
      const std::string A = Pq + "attention.attention.";
      if ((r = pack_concat(h, wm, {{A + "query", D}, {A + "key", D}, {A + "value", D}}, D, &y.qkv))) return r;
      if ((r = pack_linear(h, wm, Pq + "attention.output.dense", D, D, &y.proj))) return r;
      if ((r = pack_linear(h, wm, Pq + "intermediate.dense", 4 * D, D, &y.fc1))) return r;
      if ((r = pack_linear(h, wm, Pq + "output.dense", D, 4 * D, &y.fc2))) return r;
      continue;
    }
    const std::string P = V + "blocks." + std::to_string(l) + ".";
    if ((r = upload_f32(h, wm, P + "norm1.weight", D, &y.ln1g))) return r;
    if ((r = upload_f32(h, wm, P + "norm1.bias", D, &y.ln1b))) return r;
    if ((r = upload_f32(h, wm, P + "norm2.weight", D, &y.ln2g))) return r;
    if ((r = upload_f32(h, wm, P + "norm2.bias", D, &y.ln2b))) return r;
    if (midas) { y.ls1 = y.ls2 = nullptr; }                // timm Block with init_values=None: LayerScale is Identity
    else {
      if ((r = upload_f32(h, wm, P + "ls1.gamma", D, &y.ls1))) return r;
      if ((r = upload_f32(h, wm, P + "ls2.gamma", D, &y.ls2))) return r;
    }
    if ((r = pack_linear(h, wm, P + "attn.qkv", 3 * D, D, &y.qkv))) return r;
    if ((r = pack_linear(h, wm, P + "attn.proj", D, D, &y.proj))) return r;
    if ((r = pack_linear(h, wm, P + "mlp.fc1", 4 * D, D, &y.fc1))) return r;
    if ((r = pack_linear(h, wm, P + "mlp.fc2", D, 4 * D, &y.fc2))) return r;
  }
  h->has_dfuse = !p16 && c.use_depth_fusion && wm.get("backbone.net.depth_fusion.weight");
  if (h->has_dfuse) {
    if ((r = pack_linear(h, wm, "backbone.net.depth_fusion", D, D + 1, &h->dfuse, true, D + 64))) return r;
  }
  // ---- SFP (detectron2 SimpleFeaturePyramid; module names simfp_{log2 stride}.{index in the stage's Sequential}):
  //   scale 4   : ConvT(D, D/2) . LN . GELU . ConvT(D/2, D/4) . conv1x1+LN . conv3x3+LN        (4-level towers only)
  //   scale 2   : ConvT(D, D/2) . conv1x1+LN . conv3x3+LN
  //   scale 1   : conv1x1+LN . conv3x3+LN
  //   scale 0.5 : MaxPool2 . conv1x1+LN . conv3x3+LN
  {
    const int first = 2;                                                // int(log2(7)) = int(log2(4)) = 2
    int li = 0;
    auto sname = [&](int lvl, int idx) { return "backbone.simfp_" + std::to_string(first + lvl) + "." + std::to_string(idx); };
    if (p16) {
      if ((r = pack_convt(h, wm, sname(li, 0), D, D / 2, &h->convt4a))) return r;
      if ((r = upload_f32(h, wm, sname(li, 1) + ".weight", D / 2, &h->up_ln_g))) return r;
      if ((r = upload_f32(h, wm, sname(li, 1) + ".bias", D / 2, &h->up_ln_b))) return r;
      if ((r = pack_convt(h, wm, sname(li, 3), D / 2, D / 4, &h->convt4b))) return r;
      if ((r = pack_sfp_stage(h, wm, sname(li, 4), sname(li, 5), D / 4, &h->lv[li].st))) return r;
      h->lv[li].side = 4 * G; h->lv[li].stride = (float)P / 4.f; ++li;
    }
    if ((r = pack_convt(h, wm, sname(li, 0), D, D / 2, &h->convt))) return r;
    if ((r = pack_sfp_stage(h, wm, sname(li, 1), sname(li, 2), D / 2, &h->lv[li].st))) return r;
    h->lv[li].side = 2 * G; h->lv[li].stride = (float)P / 2.f; ++li;
    if ((r = pack_sfp_stage(h, wm, sname(li, 0), sname(li, 1), D, &h->lv[li].st))) return r;
    h->lv[li].side = G; h->lv[li].stride = (float)P; ++li;
    if ((r = pack_sfp_stage(h, wm, sname(li, 1), sname(li, 2), D, &h->lv[li].st))) return r;
    h->lv[li].side = G / 2; h->lv[li].stride = (float)P * 2.f; ++li;
  }
  // ---- heads
  const int res = c.pooler_res, F = c.fc_dim;
  h->roiK = C * res * res;
  const std::string Q = "roi_heads.cube_head.";
  if ((r = pack_roi_fc(h, wm, Q + "feature_generator.fc1", F, C, res, &h->cube_fc1))) return r;
  if ((r = pack_linear(h, wm, Q + "feature_generator.fc2", F, F, &h->cube_fc2))) return r;
  if ((r = pack_concat(h, wm, {{Q + "bbox_3D_center_deltas", 2}, {Q + "bbox_3D_dims", 3}, {Q + "bbox_3D_pose", 6},
                               {Q + "bbox_3D_center_depth", 1}, {Q + "bbox_3D_uncertainty", 1}}, F, &h->cube_out))) return r;
  h->has_box = wm.get("roi_heads.box_head.fc1.weight") && wm.get("roi_heads.box_predictor.cls_score.weight");
  if (h->has_box) {
    if ((r = pack_roi_fc(h, wm, "roi_heads.box_head.fc1", F, C, res, &h->box_fc1))) return r;
    if ((r = pack_linear(h, wm, "roi_heads.box_head.fc2", F, F, &h->box_fc2))) return r;
    if ((r = pack_concat(h, wm, {{"roi_heads.box_predictor.cls_score", c.num_classes + 1},
                                 {"roi_heads.box_predictor.bbox_pred", c.num_classes * 4}}, F, &h->box_out))) return r;
  }
  h->has_rpn = wm.get("proposal_generator.rpn_head.conv.weight") != nullptr;
  if (h->has_rpn) {
    const std::string P = "proposal_generator.rpn_head.";
    if ((r = pack_conv(h, wm, P + "conv", C, C, 3, &h->rpn_conv, true))) return r;
    // 1x1 convs: objectness [A][C] then deltas [4A][C]
    if ((r = pack_concat(h, wm, {{P + "objectness_logits", 3}, {P + "anchor_deltas", 12}}, C, &h->rpn_out))) return r;
  }
  // ---- workspace
  const size_t MT = (size_t)B * T, MP = (size_t)B * G2;
  if ((r = dalloc(h, &h->X, MT * D))) return r;
  if ((r = salloc(h, &h->PA, MP * h->Kpe))) return r;
  if ((r = salloc_il(h, &h->HN, MT * D))) return r;
  if ((r = salloc_il(h, &h->AO, MT * D))) return r;
  if ((r = salloc_il(h, &h->F1, MT * 4 * D))) return r;
  if ((r = salloc(h, &h->Q, MT * D))) return r;
  if ((r = salloc(h, &h->Kx, MT * D))) return r;
  if ((r = salloc(h, &h->Vt, (size_t)B * D * h->Tpad, true))) return r;
  h->splitk_cap = (size_t)112 << 20;     // split-K is only taken for <= 96 tiles of 128 x 128 with <= 16 slices: <= 100.7 MB of fp32 partials
  { char* q = nullptr; if ((r = dalloc(h, &q, h->splitk_cap))) return r; h->splitk_ws = (float*)q; }
  if ((r = dalloc(h, &h->attn_tail_ws, attn_tail_ws_floats(B, c.heads)))) return r;
  if ((r = dalloc(h, &h->attn_tail_cnt, (size_t)B * c.heads * 8, true))) return r;
  if ((r = salloc(h, &h->DT, MP * D))) return r;
  if ((r = salloc(h, &h->DT4, (size_t)B * (G / 2) * (G / 2) * D))) return r;
  if (h->has_dfuse) {
    if ((r = salloc(h, &h->DF, MP * (D + 64)))) return r;
    if ((r = dalloc(h, &h->dtok, MP))) return r;
    if ((r = dalloc(h, &h->FUS, MP * D))) return r;
  }
  if (sam) {
    const int ws = c.sam_window, gp = (G + ws - 1) / ws * ws, nw1 = gp / ws;
    h->sam_ws = ws; h->sam_nw = nw1 * nw1; h->sam_rows = h->sam_nw * ws * ws;
    std::vector<int> map((size_t)B * h->sam_rows);
    for (int b = 0; b < B; ++b)
      for (int wy = 0; wy < nw1; ++wy)
        for (int wx = 0; wx < nw1; ++wx)
          for (int iy = 0; iy < ws; ++iy)
            for (int ix = 0; ix < ws; ++ix) {
              const int y = wy * ws + iy, x = wx * ws + ix;          // window_partition pads bottom / right (segment_anything)
              map[(size_t)b * h->sam_rows + ((size_t)(wy * nw1 + wx) * ws + iy) * ws + ix] = (y < G && x < G) ? b * T + y * G + x : -1;
            }
    if ((r = dalloc(h, &h->sam_map, map.size()))) return r;
    HCHECK(h, hipMemcpy(h->sam_map, map.data(), map.size() * sizeof(int), hipMemcpyHostToDevice));
    const size_t MW = (size_t)B * (h->sam_rows > T ? h->sam_rows : T);
    h->ldrel = ((ws > G ? ws : G) + 3) / 4 * 4;
    if ((r = salloc(h, &h->XW, MW * D))) return r;
    if ((r = salloc(h, &h->CTX, MW * D))) return r;
    if ((r = dalloc(h, &h->QKVF, MW * 3 * D))) return r;
    if ((r = dalloc(h, &h->RELH, MW * c.heads * h->ldrel))) return r;
    if ((r = dalloc(h, &h->RELW, MW * c.heads * h->ldrel))) return r;
  }
  const int G2x = 2 * G, G4 = G / 2;
  if ((r = salloc(h, &h->CT, (size_t)B * G2x * G2x * (D / 2)))) return r;
  if (p16) {
    if ((r = salloc(h, &h->CT4a, (size_t)B * G2x * G2x * (D / 2)))) return r;
    if ((r = salloc(h, &h->CT4b, (size_t)B * 4 * G2x * G2x * (D / 4)))) return r;
  }
  for (int l = 0; l < h->nlev; ++l) {
    FpnLevel& f = h->lv[l];
    const size_t px = (size_t)B * f.side * f.side, pxp = (size_t)B * (f.side + 2) * (f.side + 2);
    if ((r = dalloc(h, &f.T1, px * C))) return r;
    if ((r = salloc(h, &f.pad, pxp * C, true))) return r;
    if ((r = dalloc(h, &f.p, px * C))) return r;
    if (h->has_rpn && (r = salloc(h, &f.rpad, pxp * C, true))) return r;
  }
  const size_t RR = (size_t)R * B;
  if ((r = salloc(h, &h->RF, RR * h->roiK))) return r;
  if ((r = salloc(h, &h->H1, RR * F))) return r;
  if ((r = salloc(h, &h->H2, RR * F))) return r;
  if ((r = dalloc(h, &h->HO, RR * 256))) return r;
  if ((r = dalloc(h, &h->rec, RR * kRecFloats))) return r;
  if ((r = dalloc(h, &h->keep, RR))) return r;
  if ((r = dalloc(h, &h->d_bidx, RR))) return r;
  if ((r = dalloc(h, &h->d_imgs, (size_t)B))) return r;
  if ((r = dalloc(h, &h->d_meta, (size_t)B))) return r;
  HCHECK(h, hipHostMalloc((void**)&h->h_imgs, sizeof(ImageDesc) * B));
  HCHECK(h, hipHostMalloc((void**)&h->h_meta, sizeof(ImageMeta) * B));
  if (h->has_rpn && h->has_box) {
    std::vector<void*> extra;
    int sides[kMaxLevels];
    for (int l = 0; l < h->nlev; ++l) sides[l] = h->lv[l].side;
    r = det2d_alloc(&h->det, B, h->nlev, sides, C, c.num_classes, R, c.rpn_pre_topk, c.rpn_post_topk, c.detections_per_image, &extra);
    for (void* p : extra) h->allocs.push_back(p);
    if (r) { h->err = "det2d workspace allocation failed"; return r; }
  }
  HCHECK(h, hipDeviceSynchronize());
  return OVM_OK;
}

static int sfp_branch(OvmHandle* h, const Split& in, int lda, int Bn, const FpnLevel& f, hipStream_t s) {
  const int C = h->C, Hs = f.side, M = Bn * Hs * Hs;
  const SfpStage& st = f.st;
  GemmParams p = gp_base(in, lda, st.c1, M);
  p.C = f.T1; p.ldc = C;
  KCHECK(h, gemm(h, p, EPI_STORE, A_ROWMAJOR, s));
  LnOut o; memset(&o, 0, sizeof(o));
  o.hi = f.pad.hi; o.lo = f.pad.lo; o.ld = C; o.padH = Hs; o.padW = Hs;
  KCHECK(h, launch_ln_rows(f.T1, C, M, C, st.n1g, st.n1b, 1e-6f, o, s));
  GemmParams q; memset(&q, 0, sizeof(q));
  q.Ahi = f.pad.hi; q.Alo = f.pad.lo; q.Whi = st.c3.w.hi; q.Wlo = st.c3.w.lo;
  q.M = M; q.N = C; q.K = 9 * C; q.cH = Hs; q.cW = Hs; q.cC = C;
  q.C = f.T1; q.ldc = C;                                 // 1x1 output already consumed by the LN above
  KCHECK(h, gemm(h, q, EPI_STORE, A_CONV3X3, s));
  LnOut o2; memset(&o2, 0, sizeof(o2));
  o2.f32 = f.p; o2.ldf = C;
  if (f.rpad.hi) { o2.hi = f.rpad.hi; o2.lo = f.rpad.lo; o2.ld = C; o2.padH = Hs; o2.padW = Hs; }
  KCHECK(h, launch_ln_rows(f.T1, C, M, C, st.n3g, st.n3b, 1e-6f, o2, s));
  return OVM_OK;
}

// ConvTranspose2d k2 s2 as a GEMM over the source pixels (EPI_CONVT scatters the 2x2 outputs): [Bn][Gs][Gs][Cin] -> [Bn][2Gs][2Gs][Cout]
static int convt_up(OvmHandle* h, const Split& in, int Cin, int Bn, int Gs, const PackedLinear& w, const Split& out, hipStream_t s) {
  GemmParams p = gp_base(in, Cin, w, Bn * Gs * Gs);
  p.Ohi = out.hi; p.Olo = out.lo; p.G = Gs; p.Cout = w.N / 4;
  KCHECK(h, gemm(h, p, EPI_CONVT, A_ROWMAJOR, s));
  return OVM_OK;
}

int ovm_backbone_forward(OvmHandle* h, const OvmImage* images, int32_t B, const float* prompt_depth, int32_t depth_h,
                         int32_t depth_w, float* p2, float* p3, float* p4, ovm_stream_t stream) {
  if (!h) return OVM_ERR_INVALID;
  h->err.clear();
  hipStream_t s = (hipStream_t)stream;
  const OvmConfig& c = h->cfg;
  if (B < 1 || B > c.max_batch) { h->err = "batch exceeds max_batch"; return OVM_ERR_CAPACITY; }
  for (int b = 0; b < B; ++b)
    if (images[b].height > c.canvas || images[b].width > c.canvas || images[b].height < 1 || images[b].width < 1) {
      h->err = "image larger than SQUARE_PAD canvas"; return OVM_ERR_SHAPE;
    }
  HCHECK(h, hipSetDevice(h->device));
  fill_meta(h, images, B);
  HCHECK(h, hipMemcpyAsync(h->d_imgs, h->h_imgs, sizeof(ImageDesc) * B, hipMemcpyHostToDevice, s));
  HCHECK(h, hipMemcpyAsync(h->d_meta, h->h_meta, sizeof(ImageMeta) * B, hipMemcpyHostToDevice, s));
  h->lastB = B;
  if (prompt_depth && c.tower != OVM_TOWER_DINOV2) {
    // detectron2's SimpleFeaturePyramid.forward(x) takes no depth; the fork's RCNN3D passes one to every backbone and
    // would raise a TypeError here: refuse rather than silently drop it
    h->err = "prompt_depth is only defined for the DINOv2 tower (depth_fusion, dino.py:91-105)"; return OVM_ERR_INVALID;
  }
  if (prompt_depth && !h->has_dfuse) { h->err = "prompt_depth given but depth_fusion weights absent / disabled"; return OVM_ERR_INVALID; }
  const int rr = backbone_launches(h, B, prompt_depth, depth_h, depth_w, s);
  if (rr) return rr;
  const int C = h->C;
  float* outs[3] = {p2, p3, p4};
  for (int l = 0; l < 3; ++l)
    if (outs[l]) HCHECK(h, hipMemcpyAsync(outs[l], h->lv[l].p, (size_t)B * h->lv[l].side * h->lv[l].side * C * 4, hipMemcpyDeviceToDevice, s));
  return OVM_OK;
}

// every kernel launch of the backbone (patch embed .. pyramid), on stream s, shapes fixed by (B, canvas)
static int backbone_launches(OvmHandle* h, int B, const float* prompt_depth, int depth_h, int depth_w, hipStream_t s) {
  const OvmConfig& c = h->cfg;
  const int D = h->D, G = h->G, G2 = h->G2, T = h->T, L = c.depth;
  // ---- patch embed (+ preprocess) ----
  KCHECK(h, launch_patch_gather(h->d_imgs, B, G, h->patch, h->Kpe, c.pixel_mean, c.pixel_std, h->PA.hi, h->PA.lo, s));
  if (!h->sam) KCHECK(h, launch_cls_init(h->X, h->cls, h->pos, B, T, D, s));       // SAM: no class token (T = G^2)
  {
    GemmParams p = gp_base(h->PA, h->Kpe, h->pe, B * G2);
    p.X = h->X; p.ldx = D; p.pos = h->pos; p.G2 = G2; p.T = T;
    KCHECK(h, gemm(h, p, EPI_PATCH, A_ROWMAJOR, s));
  }
  const int M = B * T;
  const float eps = h->ln_eps;
  if (h->lnpre_g) {                                        // open_clip: x = ln_pre(x + pos) (reference clip.py:78-79), in place
    LnOut o; memset(&o, 0, sizeof(o)); o.f32 = h->X; o.ldf = D;
    ProfScope ps(h, OVM_PROF_LN, s); KCHECK(h, launch_ln_rows(h->X, D, M, D, h->lnpre_g, h->lnpre_b, eps, o, s));
  }
  for (int l = 0; l < L; ++l) {
    const Layer& y = h->layers[l];
    const int il = h->npass == 3 ? 1 : 0, am = il ? 2 : 1;    // activations of the blocks: interleaved split images in f16x3 mode
    LnOut o; memset(&o, 0, sizeof(o)); o.hi = h->HN.hi; o.lo = h->HN.lo; o.ld = am * D; o.il = il;
    if (h->sam) {
      // segment_anything Block (reference sam.py:100-106 runs vit.blocks as they are): norm1 -> [zero-padded 14 x 14 windows] ->
      // attention with the decomposed relative-position bias -> [un-partition] -> + shortcut. Same organisation as a Swin block of
      // the detector: the norm writes window-partitioned rows (padding rows zero AFTER the norm), the projection's epilogue
      // scatters back through the same map and adds the shortcut. Scores are exact fp32 products on the matrix cores (attn_f32).
      const int ws = y.ws, rows = ws ? h->sam_rows : T, Mw = B * rows, side = ws ? ws : G, Tq = side * side, nseq = Mw / Tq;
      {
        RowOpParams rp; memset(&rp, 0, sizeof(rp));
        rp.x = h->X; rp.ldx = D; rp.gamma = y.ln1g; rp.beta = y.ln1b; rp.eps = eps; rp.M = Mw; rp.D = D;
        if (ws) { rp.idx = h->sam_map; rp.nidx = 1; rp.seg = D; rp.zero_masked = 1; }
        rp.hi = h->XW.hi; rp.lo = h->XW.lo; rp.ldh = D;
        ProfScope ps(h, OVM_PROF_LN, s); KCHECK(h, launch_rowop(rp, s));
      }
      {
        GemmParams p = gp_base(h->XW, D, y.qkv, Mw);
        p.C = h->QKVF; p.ldc = 3 * D;
        KCHECK(h, gemm(h, p, EPI_STORE, A_ROWMAJOR, s, OVM_PROF_QKV));
      }
      {
        ProfScope ps(h, OVM_PROF_ATTN, s);
        // the bias tables use the UNSCALED query (add_decomposed_rel_pos is given q, the scores use q * scale)
        KCHECK(h, launch_relpos_tables(h->QKVF, 3 * D, Mw, c.heads, 64, side, side, y.relh, y.relw, h->RELH, h->RELW, h->ldrel, s));
        AttnF32Params a; memset(&a, 0, sizeof(a));
        a.q = h->QKVF; a.k = h->QKVF + D; a.v = h->QKVF + 2 * D; a.ldq = a.ldk = a.ldv = 3 * D;
        a.sq1 = a.sk1 = a.sv1 = (long)Tq * 3 * D; a.sq2 = a.sk2 = a.sv2 = 64;
        a.ohi = h->CTX.hi; a.olo = h->CTX.lo; a.ldoh = D; a.soh1 = (long)Tq * D; a.soh2 = 64;
        a.nb1 = nseq; a.nb2 = c.heads; a.Tq = Tq; a.Tk = Tq; a.DH = 64; a.scale = 0.125f;
        a.rel_h = h->RELH; a.rel_w = h->RELW; a.rel_gw = side; a.ldrel = h->ldrel;
        KCHECK(h, launch_attn_f32(a, s));
      }
      {
        GemmParams p = gp_base(h->CTX, D, y.proj, Mw);
        p.X = h->X; p.ldx = D; p.row_map = ws ? h->sam_map : nullptr;
        KCHECK(h, gemm(h, p, EPI_RESID, A_ROWMAJOR, s, OVM_PROF_PROJ));
      }
    } else {
    { ProfScope ps(h, OVM_PROF_LN, s); KCHECK(h, launch_ln_rows(h->X, D, M, D, y.ln1g, y.ln1b, eps, o, s)); }
    {
      GemmParams p = gp_base(h->HN, am * D, y.qkv, M); p.a_il = il;
      p.Qhi = h->Q.hi; p.Qlo = h->Q.lo; p.Khi = h->Kx.hi; p.Klo = h->Kx.lo; p.Vhi = h->Vt.hi; p.Vlo = h->Vt.lo;
      p.T = T; p.Tpad = h->Tpad; p.heads = c.heads; p.qscale = kQScale;
      KCHECK(h, gemm(h, p, EPI_QKV, A_ROWMAJOR, s, OVM_PROF_QKV));
    }
    {
      AttnParams a; memset(&a, 0, sizeof(a));
      a.Qhi = h->Q.hi; a.Qlo = h->Q.lo; a.Khi = h->Kx.hi; a.Klo = h->Kx.lo; a.Vhi = h->Vt.hi; a.Vlo = h->Vt.lo;
      a.Ohi = h->AO.hi; a.Olo = h->AO.lo; a.ldo = am * D; a.o_il = il; a.B = B; a.heads = c.heads; a.T = T; a.Tpad = h->Tpad;
      a.corun = h->corun ? 1 : 0;
      a.tail_ws = h->attn_tail_ws; a.tail_cnt = h->attn_tail_cnt;     // leftover queries (T = 4097: one per head) split over the keys
      { ProfScope ps(h, OVM_PROF_ATTN, s); KCHECK(h, launch_attention(a, h->npass, s)); }
    }
    {
      GemmParams p = gp_base(h->AO, am * D, y.proj, M); p.a_il = il;
      p.gamma = y.ls1; p.X = h->X; p.ldx = D;
      KCHECK(h, gemm(h, p, EPI_RESID, A_ROWMAJOR, s, OVM_PROF_PROJ));
    }
    }
    { ProfScope ps(h, OVM_PROF_LN, s); KCHECK(h, launch_ln_rows(h->X, D, M, D, y.ln2g, y.ln2b, eps, o, s)); }
    {
      GemmParams p = gp_base(h->HN, am * D, y.fc1, M); p.a_il = il;
      p.Ohi = h->F1.hi; p.Olo = h->F1.lo; p.ldo = am * 4 * D; p.o_il = il; p.relu = h->mlp_act;
      KCHECK(h, gemm(h, p, EPI_GELU, A_ROWMAJOR, s, OVM_PROF_FC1));
    }
    {
      GemmParams p = gp_base(h->F1, am * 4 * D, y.fc2, M); p.a_il = il;
      p.gamma = y.ls2; p.X = h->X; p.ldx = D;
      KCHECK(h, gemm(h, p, EPI_RESID, A_ROWMAJOR, s, OVM_PROF_FC2));
    }
  }
  // ---- depth fusion at the last block output (reference dino.py:91-105) ----
  if (prompt_depth) {
    KCHECK(h, launch_depth_resize(prompt_depth, B, depth_h, depth_w, G, h->dtok, s));
    KCHECK(h, launch_tokens_cast(h->X, B, T, G2, D, D + 64, h->dtok, h->DF.hi, h->DF.lo, s));
    GemmParams p = gp_base(h->DF, D + 64, h->dfuse, B * G2);
    p.C = h->FUS; p.ldc = D;
    KCHECK(h, gemm(h, p, EPI_STORE, A_ROWMAJOR, s));
    KCHECK(h, launch_tokens_writeback(h->X, h->FUS, B, T, G2, D, s));
  }
  // ---- dense tokens (no final LayerNorm: reference dino.py:88-110) ----
  KCHECK(h, launch_tokens_cast(h->X, B, T, G2, D, D, nullptr, h->DT.hi, h->DT.lo, s));
  // ---- SFP (reference dino.py:143-152,208-224; stages nohup.out:565-596; 4-level form clip.py:155-166) ----
  {
    int li = 0;
    if (h->nlev == 4) {                                    // scale 4: ConvT . LN . GELU . ConvT
      KCHECK(h, convt_up(h, h->DT, D, B, G, h->convt4a, h->CT4a, s));
      KCHECK(h, launch_ln_gelu_split(h->CT4a.hi, h->CT4a.lo, B * 4 * G2, D / 2, h->up_ln_g, h->up_ln_b, 1e-6f, s));
      KCHECK(h, convt_up(h, h->CT4a, D / 2, B, 2 * G, h->convt4b, h->CT4b, s));
      KCHECK(h, sfp_branch(h, h->CT4b, D / 4, B, h->lv[li++], s));
    }
    KCHECK(h, convt_up(h, h->DT, D, B, G, h->convt, h->CT, s));
    KCHECK(h, sfp_branch(h, h->CT, D / 2, B, h->lv[li++], s));
    KCHECK(h, sfp_branch(h, h->DT, D, B, h->lv[li++], s));
    KCHECK(h, launch_maxpool2(h->DT.hi, h->DT.lo, B, G, D, h->DT4.hi, h->DT4.lo, s));
    KCHECK(h, sfp_branch(h, h->DT4, D, B, h->lv[li++], s));
  }
  return OVM_OK;
}

int ovm_backbone_num_levels(const OvmHandle* h) { return h ? h->nlev : OVM_ERR_INVALID; }

int ovm_backbone_level(const OvmHandle* h, int32_t level, const float** data, int32_t* side, float* stride) {
  if (!h || level < 0 || level >= h->nlev) return OVM_ERR_INVALID;
  if (data) *data = h->lv[level].p;
  if (side) *side = h->lv[level].side;
  if (stride) *stride = h->lv[level].stride;
  return OVM_OK;
}

static void roi_params(OvmHandle* h, RoiParams* rp) {
  memset(rp, 0, sizeof(*rp));
  for (int l = 0; l < h->nlev; ++l) {
    rp->feat[l] = h->lv[l].p; rp->fh[l] = rp->fw[l] = h->lv[l].side; rp->scale[l] = 1.0f / h->lv[l].stride;
  }
  rp->C = h->C; rp->nlevels = h->nlev; rp->min_level = h->cfg.pooler_min_level; rp->max_level = h->cfg.pooler_max_level;
  rp->out = h->cfg.pooler_res;
}

int ovm_cube_forward(OvmHandle* h, const OvmImage* images, int32_t B, const float* boxes, const float* scores,
                     const int32_t* classes, const int32_t* image_idx, int32_t n, int32_t postprocess, OvmDet3D* out,
                     int32_t* out_counts, ovm_stream_t stream) {
  if (!h) return OVM_ERR_INVALID;
  h->err.clear();
  hipStream_t s = (hipStream_t)stream;
  if (B < 1 || B > h->cfg.max_batch) { h->err = "batch exceeds max_batch"; return OVM_ERR_CAPACITY; }
  if (n > h->cfg.max_rois * h->cfg.max_batch) { h->err = "n exceeds max_rois*max_batch"; return OVM_ERR_CAPACITY; }
  HCHECK(h, hipSetDevice(h->device));
  if (n <= 0) {                                       // roi_heads.py:371-372: nothing to do
    HCHECK(h, hipMemsetAsync(out_counts, 0, sizeof(int) * B, s));
    return OVM_OK;
  }
  fill_meta(h, images, B);
  HCHECK(h, hipMemcpyAsync(h->d_meta, h->h_meta, sizeof(ImageMeta) * B, hipMemcpyHostToDevice, s));
  h->lastN = n;
  const int F = h->cfg.fc_dim;
  RoiParams rp; roi_params(h, &rp);
  rp.boxes = boxes; rp.batch_idx = image_idx; rp.n = n; rp.Ohi = h->RF.hi; rp.Olo = h->RF.lo; rp.ldo = h->roiK;
  KCHECK(h, launch_roi_align(rp, s));
  {
    GemmParams p = gp_base(h->RF, h->roiK, h->cube_fc1, n);
    p.Ohi = h->H1.hi; p.Olo = h->H1.lo; p.ldo = F; p.relu = 1;
    KCHECK(h, gemm(h, p, EPI_STORE, A_ROWMAJOR, s));
  }
  {
    GemmParams p = gp_base(h->H1, F, h->cube_fc2, n);
    p.Ohi = h->H2.hi; p.Olo = h->H2.lo; p.ldo = F; p.relu = 1;
    KCHECK(h, gemm(h, p, EPI_STORE, A_ROWMAJOR, s));
  }
  {
    GemmParams p = gp_base(h->H2, F, h->cube_out, n);
    p.C = h->HO; p.ldc = 16;
    KCHECK(h, gemm(h, p, EPI_STORE, A_ROWMAJOR, s));
  }
  CubeDecodeParams cp; memset(&cp, 0, sizeof(cp));
  cp.head = h->HO; cp.ldh = 16; cp.boxes = boxes; cp.scores = scores; cp.classes = classes; cp.batch_idx = image_idx;
  cp.meta = h->d_meta; cp.n = n; cp.virtual_focal = h->cfg.virtual_focal; cp.rec = h->rec; cp.keep = h->keep;
  cp.postprocess = postprocess;
  KCHECK(h, launch_cube_decode(cp, s));
  KCHECK(h, launch_compact_records(h->rec, h->keep, n, B, (float*)out, out_counts, s));
  return OVM_OK;
}

int ovm_rpn_box_forward(OvmHandle* h, const OvmImage* images, int32_t B, float* boxes, float* scores, int32_t* classes,
                        int32_t* image_idx, float* scores_full, int32_t* out_counts, ovm_stream_t stream) {
  if (!h) return OVM_ERR_INVALID;
  h->err.clear();
  if (!h->has_rpn || !h->has_box) { h->err = "checkpoint has no RPN / box-head weights"; return OVM_ERR_MISSING_WEIGHT; }
  if (B < 1 || B > h->cfg.max_batch) { h->err = "batch exceeds max_batch"; return OVM_ERR_CAPACITY; }
  hipStream_t s = (hipStream_t)stream;
  HCHECK(h, hipSetDevice(h->device));
  fill_meta(h, images, B);
  HCHECK(h, hipMemcpyAsync(h->d_meta, h->h_meta, sizeof(ImageMeta) * B, hipMemcpyHostToDevice, s));
  Det2dModel m; memset(&m, 0, sizeof(m));
  m.npass = h->npass; m.B = B; m.C = h->C; m.F = h->cfg.fc_dim; m.roiK = h->roiK;
  m.num_classes = h->cfg.num_classes;
  m.nlev = h->nlev;
  for (int l = 0; l < h->nlev; ++l) { m.rpad[l] = {h->lv[l].rpad.hi, h->lv[l].rpad.lo}; m.stride[l] = h->lv[l].stride; }
  m.rpn_conv_hi = h->rpn_conv.w.hi; m.rpn_conv_lo = h->rpn_conv.w.lo; m.rpn_conv_bias = h->rpn_conv.bias;
  m.rpn_out_hi = h->rpn_out.w.hi; m.rpn_out_lo = h->rpn_out.w.lo; m.rpn_out_bias = h->rpn_out.bias;
  m.fc1_hi = h->box_fc1.w.hi; m.fc1_lo = h->box_fc1.w.lo; m.fc1_bias = h->box_fc1.bias;
  m.fc2_hi = h->box_fc2.w.hi; m.fc2_lo = h->box_fc2.w.lo; m.fc2_bias = h->box_fc2.bias;
  m.out_hi = h->box_out.w.hi; m.out_lo = h->box_out.w.lo; m.out_bias = h->box_out.bias;
  for (int i = 0; i < kMaxLevels; ++i) m.anchor_sizes[i] = h->cfg.anchor_sizes[i];
  for (int i = 0; i < 3; ++i) m.anchor_ratios[i] = h->cfg.anchor_ratios[i];
  m.pre_topk = h->cfg.rpn_pre_topk; m.post_topk = h->cfg.rpn_post_topk; m.rpn_nms = h->cfg.rpn_nms_thresh;
  m.score_thresh = h->cfg.score_thresh; m.nms_thresh = h->cfg.nms_thresh; m.topk = h->cfg.detections_per_image;
  m.meta = h->d_meta;
  RoiParams rp; roi_params(h, &rp);
  m.roi = rp; m.RF = {h->RF.hi, h->RF.lo}; m.H1 = {h->H1.hi, h->H1.lo}; m.H2 = {h->H2.hi, h->H2.lo}; m.HO = h->HO;
  int r = det2d_forward(m, h->det, boxes, scores, classes, image_idx, scores_full, out_counts, s);
  if (r) { h->err = "det2d_forward failed (" + std::to_string(r) + ")"; return r; }
  return OVM_OK;
}

// RCNN3D.inference for ONE image with the text-prompted head, as one call (SURVEY.md 8b `ovm_infer`; reference
// cubercnn/modeling/meta_arch/rcnn3d.py:79-117 with `category_list`: preprocess -> DINOv2 + SFP -> ROIHeads3DGDINO (GroundingDINO
// network, phrase scores, threshold, NMS, class index: roi_heads_gdino.py:93-171) -> _forward_cube -> detector_postprocess).
// The detector runs on an internal side stream beside the backbone (it reads only the input image); the caller's stream joins it
// before the output glue. One host synchronisation (the number of kept 2D boxes sizes the cube-head launch).
// Everything of ovm_infer behind its argument checks: fork, the two networks, join, glue, cube head. Split off so that EVERY error exit
// (a failed detector or backbone launch, an allocation, the capacity check) passes through one place that drains both streams.
static int infer_forked(OvmHandle* h, OvmGdino* g, const OvmImage* image, const int32_t* token_ids, int32_t ntok, const int32_t* spans,
                        int32_t n_phrases, float box_threshold, float nms_threshold, OvmDet3D* out, int32_t out_capacity, int32_t* n_out,
                        ovm_stream_t stream, int nq) {
  hipStream_t s = (hipStream_t)stream;
  // fork: detector on the side stream
  HCHECK(h, hipEventRecord(h->ev_fork, s));
  HCHECK(h, hipStreamWaitEvent(h->side, h->ev_fork, 0));
  int r = ovm_gdino_forward(g, image, token_ids, ntok, nullptr, nullptr, nullptr, (ovm_stream_t)h->side);
  if (r) { h->err = std::string("ovm_gdino_forward: ") + ovm_gdino_last_error(g); return r; }
  HCHECK(h, hipEventRecord(h->ev_join, h->side));
  // backbone on the caller's stream
  r = ovm_backbone_forward(h, image, 1, nullptr, 0, 0, nullptr, nullptr, nullptr, stream);
  if (r) return r;
  // join, output glue (three launches on the handle's scratch), kept count to the host (the one synchronisation in front of the
  // cube head, whose grid it sizes), cube head
  HCHECK(h, hipStreamWaitEvent(s, h->ev_join, 0));
  const float *logits = nullptr, *gboxes = nullptr; int ld = 0;
  r = ovm_gdino_last_outputs(g, &logits, &gboxes, &ld);
  if (r) { h->err = "detector outputs unavailable"; return r; }
  if (nq <= 2048) {
    const size_t need = gdino_post_ws_bytes(nq, n_phrases);
    if (h->inf_ws_bytes < need) {
      char* q = nullptr;
      if ((r = dalloc(h, &q, need + need / 4))) return r;        // (the previous block stays in h->allocs until ovm_destroy)
      h->inf_ws = q; h->inf_ws_bytes = need + need / 4;
    }
    r = launch_gdino_post_ws(logits, nq, ld, gboxes, spans, n_phrases, image->height, image->width, box_threshold, nms_threshold, h->inf_ws,
                             h->inf_ws_bytes, h->inf_boxes, h->inf_scores, h->inf_classes, h->inf_n, s);
  } else {
    r = ovm_gdino_postprocess(logits, nq, ld, gboxes, spans, n_phrases, image->height, image->width, box_threshold, nms_threshold,
                              h->inf_boxes, h->inf_scores, h->inf_classes, h->inf_n, stream);
  }
  if (r) { h->err = "GroundingDINO output glue failed (" + std::to_string(r) + ")"; return r; }
  if (!h->inf_host) HCHECK(h, hipHostMalloc((void**)&h->inf_host, 2 * sizeof(int), hipHostMallocDefault));
  HCHECK(h, hipMemcpyAsync(&h->inf_host[0], h->inf_n, sizeof(int), hipMemcpyDeviceToHost, s));
  HCHECK(h, hipStreamSynchronize(s));
  const int n2d = h->inf_host[0];
  if (n2d > out_capacity) { h->err = "output capacity too small"; return OVM_ERR_CAPACITY; }
  r = ovm_cube_forward(h, image, 1, h->inf_boxes, h->inf_scores, h->inf_classes, h->inf_idx, n2d, 1, out, h->inf_counts, stream);
  if (r) return r;
  HCHECK(h, hipMemcpyAsync(&h->inf_host[1], h->inf_counts, sizeof(int), hipMemcpyDeviceToHost, s));
  HCHECK(h, hipStreamSynchronize(s));
  *n_out = h->inf_host[1];
  return OVM_OK;
}

int ovm_infer(OvmHandle* h, OvmGdino* g, const OvmImage* image, const int32_t* token_ids, int32_t ntok, const int32_t* spans,
              int32_t n_phrases, float box_threshold, float nms_threshold, OvmDet3D* out, int32_t out_capacity, int32_t* n_out,
              ovm_stream_t stream) {
  if (!h || !g || !image || !token_ids || !out || !n_out) return OVM_ERR_INVALID;
  h->err.clear();
  hipStream_t s = (hipStream_t)stream;
  HCHECK(h, hipSetDevice(h->device));
  if (!h->side) {
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    HCHECK(h, hipStreamCreateWithPriority(&h->side, hipStreamNonBlocking, hi));      // short kernels: let them jump the ViT's queue
    HCHECK(h, hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
    HCHECK(h, hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming));
  }
  const int nq = ovm_gdino_num_queries(g);
  if (nq <= 0) { h->err = "bad detector handle"; return OVM_ERR_INVALID; }
  if (nq > h->cfg.max_rois * h->cfg.max_batch) { h->err = "detector queries exceed max_rois"; return OVM_ERR_CAPACITY; }
  if (h->inf_cap < nq) {
    int r;
    if ((r = dalloc(h, &h->inf_boxes, (size_t)nq * 4)) || (r = dalloc(h, &h->inf_scores, (size_t)nq)) || (r = dalloc(h, &h->inf_classes, (size_t)nq)) ||
        (r = dalloc(h, &h->inf_idx, (size_t)nq, true)) || (r = dalloc(h, &h->inf_n, 1)) || (r = dalloc(h, &h->inf_counts, 1))) return r;
    h->inf_cap = nq;
  }
  const int r_all = infer_forked(h, g, image, token_ids, ntok, spans, n_phrases, box_threshold, nms_threshold, out, out_capacity, n_out, stream, nq);
  if (r_all != OVM_OK) {
    // The detector may still be running on the side stream, reading the caller's image and the plan arena: the caller is entitled to
    // free or reuse the image once this call has returned, and the next call's fork must not race with leftover work on h->inf_*.
    (void)hipStreamSynchronize(h->side);
    (void)hipStreamSynchronize(s);
  }
  return r_all;
}

// Co-run mode: the caller runs other work (the GroundingDINO detector) on a second stream while ovm_backbone_forward executes. The
// attention launches then keep to one 4-wave workgroup per CU (slower alone: 6.8 -> 9.3 ms per ViT-L image) so that the other stream's
// short kernels find free wave slots instead of queueing behind 270-us workgroups; measured end to end 28.3 -> 26.7 ms per image.
int ovm_set_corun(OvmHandle* h, int32_t on) {
  if (!h) return OVM_ERR_INVALID;
  h->corun = on != 0;
  return OVM_OK;
}

int ovm_profile_enable(OvmHandle* h, int32_t on) {
  if (!h) return OVM_ERR_INVALID;
  h->prof = on != 0;
  h->prof_mask = (on == 0 || on == 1) ? ~0u : ((unsigned)on >> 1);       // 1 = every category; else bit (c + 1) selects category c
  for (int c = 0; c < OVM_PROF_NCAT; ++c) h->prof_used[c] = 0;
  return OVM_OK;
}

// Sums the elapsed time of every bracketed launch since ovm_profile_enable(h, 1) per category and
// resets the counters. Synchronises the device.
int ovm_profile_read(OvmHandle* h, float* ms, int32_t* launches) {
  if (!h) return OVM_ERR_INVALID;
  HCHECK(h, hipSetDevice(h->device));
  HCHECK(h, hipDeviceSynchronize());
  for (int c = 0; c < OVM_PROF_NCAT; ++c) {
    double tot = 0.0;
    for (size_t i = 0; i < h->prof_used[c]; ++i) {
      float t = 0.f;
      if (hipEventElapsedTime(&t, h->prof_ev[c][i].first, h->prof_ev[c][i].second) == hipSuccess) tot += t;
    }
    ms[c] = (float)tot; launches[c] = (int32_t)h->prof_used[c];
    h->prof_used[c] = 0;
  }
  return OVM_OK;
}

// RCCL communicator bootstrap for ovm_gather_records (the unique id travels over the host's own
// rendezvous, e.g. torch.distributed's store).
int ovm_comm_unique_id(uint8_t* id128) {
  ncclUniqueId id;
  if (ncclGetUniqueId(&id) != ncclSuccess) return OVM_ERR_HIP;
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId size");
  memcpy(id128, &id, 128);
  return OVM_OK;
}
int ovm_comm_init(const uint8_t* id128, int32_t rank, int32_t world, int32_t device, void** comm) {
  ncclUniqueId id; memcpy(&id, id128, 128);
  if (hipSetDevice(device) != hipSuccess) return OVM_ERR_HIP;
  ncclComm_t c;
  if (ncclCommInitRank(&c, world, id, rank) != ncclSuccess) return OVM_ERR_HIP;
  *comm = (void*)c;
  return OVM_OK;
}
int ovm_comm_destroy(void* comm) {
  if (comm) ncclCommDestroy((ncclComm_t)comm);
  return OVM_OK;
}

int64_t ovm_debug_copy(OvmHandle* h, const char* name, float* dst, int64_t capacity, ovm_stream_t stream) {
  if (!h || !name) return OVM_ERR_INVALID;
  const int B = h->lastB, G = h->G, C = h->C;
  const float* src = nullptr; int64_t n = 0;
  const std::string k(name);
  if (k == "tokens") { src = h->X; n = (int64_t)B * h->T * h->D; }
  else if (k.size() == 2 && k[0] == 'p' && k[1] >= '2' && k[1] < '2' + h->nlev) {
    const FpnLevel& f = h->lv[k[1] - '2'];
    src = f.p; n = (int64_t)B * f.side * f.side * C;
  }
  // the RPN's per-image proposals after top-k / NMS (boxes [B][R][4], objectness logits [B][R], counts [B] as int32 bits) and the
  // cube head's raw outputs of the last ovm_cube_forward ([n][16]: deltas 2, dims 3, pose 6, depth 1, uncertainty 1) - parity tests
  else if (k == "rpn_boxes" && h->has_rpn && h->has_box) { src = h->det.prop_boxes; n = (int64_t)B * h->det.R * 4; }
  else if (k == "rpn_scores" && h->has_rpn && h->has_box) { src = h->det.prop_scores; n = (int64_t)B * h->det.R; }
  else if (k == "rpn_counts" && h->has_rpn && h->has_box) { src = (const float*)h->det.prop_count; n = B; }
  else if (k == "cube_head") { src = h->HO; n = (int64_t)h->lastN * 16; }
  else { h->err = "unknown debug tensor"; return OVM_ERR_INVALID; }
  if (n > capacity) { h->err = "debug copy capacity too small"; return OVM_ERR_CAPACITY; }
  if (hipMemcpyAsync(dst, src, (size_t)n * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess) return OVM_ERR_HIP;
  return n;
}

// Counts exchange alone (every rank calls it): lets rank 0 size its receive buffer before ovm_gather_records.
int ovm_gather_counts(void* comm, int32_t rank, int32_t world, int32_t n_send, int32_t* counts_all, ovm_stream_t stream) {
  if (!comm || world < 1 || rank < 0 || rank >= world || !counts_all || n_send < 0) return OVM_ERR_INVALID;
  ncclComm_t cm = (ncclComm_t)comm;
  hipStream_t s = (hipStream_t)stream;
  int* d_counts = nullptr;
  if (hipMalloc((void**)&d_counts, sizeof(int) * (world + 1)) != hipSuccess) return OVM_ERR_HIP;
  int rc = OVM_OK;
  if (hipMemcpyAsync(d_counts + world, &n_send, sizeof(int), hipMemcpyHostToDevice, s) != hipSuccess) rc = OVM_ERR_HIP;
  else if (ncclAllGather(d_counts + world, d_counts, 1, ncclInt32, cm, s) != ncclSuccess) rc = OVM_ERR_HIP;
  else if (hipMemcpyAsync(counts_all, d_counts, sizeof(int) * world, hipMemcpyDeviceToHost, s) != hipSuccess) rc = OVM_ERR_HIP;
  if (hipStreamSynchronize(s) != hipSuccess) rc = OVM_ERR_HIP;
  hipFree(d_counts);
  return rc;
}

// One gather of fixed-width records to rank 0: counts all-gather, then grouped send/recv (a gatherv).
// Every peer uses its own xGMI link to rank 0's GPU; the payload is ~200 B per detection.
int ovm_gather_records(void* comm, int32_t rank, int32_t world, const OvmDet3D* send, int32_t n_send, OvmDet3D* recv,
                       int32_t* counts_all, ovm_stream_t stream) {
  if (!comm || world < 1 || rank < 0 || rank >= world) return OVM_ERR_INVALID;
  ncclComm_t cm = (ncclComm_t)comm;
  hipStream_t s = (hipStream_t)stream;
  int* d_counts = nullptr;
  if (hipMalloc((void**)&d_counts, sizeof(int) * (world + 1)) != hipSuccess) return OVM_ERR_HIP;
  int rc = OVM_OK;
  do {
    if (hipMemcpyAsync(d_counts + world, &n_send, sizeof(int), hipMemcpyHostToDevice, s) != hipSuccess) { rc = OVM_ERR_HIP; break; }
    if (ncclAllGather(d_counts + world, d_counts, 1, ncclInt32, cm, s) != ncclSuccess) { rc = OVM_ERR_HIP; break; }
    if (hipMemcpyAsync(counts_all, d_counts, sizeof(int) * world, hipMemcpyDeviceToHost, s) != hipSuccess) { rc = OVM_ERR_HIP; break; }
    if (hipStreamSynchronize(s) != hipSuccess) { rc = OVM_ERR_HIP; break; }
    const size_t rb = sizeof(OvmDet3D);
    if (ncclGroupStart() != ncclSuccess) { rc = OVM_ERR_HIP; break; }
    if (rank == 0) {
      size_t off = 0;
      for (int r = 0; r < world; ++r) {
        if (r == 0) {
          if (n_send > 0) hipMemcpyAsync(recv, send, rb * n_send, hipMemcpyDeviceToDevice, s);
        } else if (counts_all[r] > 0) {
          ncclRecv((char*)recv + off * rb, rb * counts_all[r], ncclUint8, r, cm, s);
        }
        off += counts_all[r];
      }
    } else if (n_send > 0) {
      ncclSend(send, rb * n_send, ncclUint8, 0, cm, s);
    }
    if (ncclGroupEnd() != ncclSuccess) { rc = OVM_ERR_HIP; break; }
  } while (0);
  hipStreamSynchronize(s);
  hipFree(d_counts);
  return rc;
}

}  // extern "C"
