// Native GroundingDINO engine: the network `ROIHeads3DGDINO` calls at reference cubercnn/modeling/roi_heads/roi_heads_gdino.py:186
// (built at :16-23 from configs/GroundingDINO_SwinB_cfg.py; IDEA-Research/GroundingDINO @856dde2, source not in the reference
// tree), sequenced in C++ behind ovm_gdino_create / ovm_gdino_forward / ovm_gdino_detect / ovm_infer.
//
// BERT text encoder -> Swin backbone -> input projections -> 6 x (image<->text fusion, text enhancer, multi-scale deformable
// self-attention) -> two-stage query selection -> 6 x decoder -> contrastive class logits + boxes. Module structure and parameter
// names follow the Hugging Face port (the independent implementation the parity tests compare against).
//
// Execution model. Everything that depends only on (image size, caption) - index maps of the window partition / shift / patch
// merging, shift masks, sine position embeddings, reference grids, text masks - is built on the host once per *plan* and uploaded.
// A plan also owns the activation arena (sized by a dry pass over the same code) and every scratch buffer, so nothing is
// allocated, freed or re-sized on the hot path, and the whole forward is captured into ONE HIP graph per plan, replayed afterwards.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstring>
#include <list>
#include <map>
#include <memory>
#include <array>
#include <string>
#include <unordered_map>
#include <vector>
#include "../../include/ovm3d.h"
#include "det2d.hpp"
#include "gdino.hpp"
#include "dec_chain.hpp"
#include "kernels.hpp"

using namespace ovm;

namespace {

struct Lin {                       // packed nn.Linear: split-fp16 weight image (gemm.hpp layout) + fp32 bias
  half_t* hi = nullptr; half_t* lo = nullptr; float* bias = nullptr;
  half_t* frag = nullptr;          // decoder weights only: the same image in MFMA-fragment order (make_frag), for the row-chain kernels
  int N = 0, K = 0, Kpad = 0;
};
struct Ln { float* g = nullptr; float* b = nullptr; };
struct SplitBuf { half_t* hi = nullptr; half_t* lo = nullptr; int ld = 0; bool il = false; };   // il: one interleaved image [row][k/32][hi 32 | lo 32], lo = hi + 32, ld = 2 K

struct SwinBlock { Ln ln1, ln2; Lin qkv, proj, fc1, fc2; float* relbias = nullptr; };
struct SwinStage { std::vector<SwinBlock> blocks; int nh = 0, C = 0; bool has_red = false; Lin red; Ln dn; bool has_out = false; Ln on; };
struct BertLayer { Lin qkv, ao, fi, fo; Ln aln, oln; };
struct Mha { Lin qk, v, out; Lin q, kv; int heads = 0; };       // qk: [query | key] rows; kv: [key | value]; q alone for cross attention
struct MsdaW { Lin offw, value, out; };
struct EncLayer {
  Ln lnv, lnt; Lin vqv, tkv, ov, ot;            // fusion: [vision_proj | values_vision_proj], [text_proj | values_text_proj], gated output projections
  Mha te; Ln te_ln1, te_ln2; Lin te_fc1, te_fc2;
  MsdaW msda; Ln de_ln1, de_ln2; Lin de_fc1, de_fc2;
};
struct DecLayer { Mha sa, ca; MsdaW msda; Ln ln1, ln2, ln3, ln4; Lin fc1, fc2; };

struct Plan;

}  // namespace

namespace ovm {
static int g_gdino_branches = 1;
void set_gdino_branches(int v) { g_gdino_branches = v ? 1 : 0; }
static int g_gdino_dec_chain = 1;      // decoder layers as row-chain kernels (dec_chain.hip); read at capture time, like the branches
void set_gdino_dec_chain(int v) { g_gdino_dec_chain = v ? 1 : 0; }
static int g_gdino_ffn_split = 0;      // decoder chain B over (row blocks) x (FFN chunks) + chain C; read when a plan is built. Bit-identical; 8.61 -> 8.34 ms
                                       // for the detector ALONE, but 51.14 -> 50.97 images/s beside the ViT (four times the workgroups on the chip): off
void set_gdino_ffn_split(int v) { g_gdino_ffn_split = v ? 1 : 0; }
static int g_gdino_swin_fused = 1;     // Swin blocks: qkv projection inside the window-attention kernel; read when a plan is built
void set_gdino_swin_fused(int v) { g_gdino_swin_fused = v ? 1 : 0; }
static int g_gdino_gemm256 = 1;        // the wide K <= 256 contractions on the 256 x 256 GEMM (interleaved activations); read when a plan is built
void set_gdino_gemm256(int v) { g_gdino_gemm256 = v ? 1 : 0; }
}  // namespace ovm
using ovm::g_gdino_branches;
using ovm::g_gdino_dec_chain;
using ovm::g_gdino_gemm256;
using ovm::g_gdino_ffn_split;
using ovm::g_gdino_swin_fused;

constexpr size_t kSlabBytes = (size_t)256 << 20;

struct OvmGdino {
  OvmGdinoConfig cfg;
  int device = 0, npass = 3;
  std::string err;
  std::vector<void*> allocs;
  char* slab = nullptr; size_t slab_off = 0, slab_cap = 0;      // current slab of dmal()
  float* sine_dim_t = nullptr;                                 // [d_model / 4] frequency table of the decoder's sine embedding (dec_chain.hip)
  // ---- weights
  float *word = nullptr, *posemb = nullptr, *typemb = nullptr; Ln emb_ln; int bertD = 0, n_pos = 0, vocab = 0;
  std::vector<BertLayer> bert;
  Lin text_proj;
  Lin pe; Ln pe_ln;
  std::vector<SwinStage> stages;
  struct InProj { Lin w; int k = 1; Ln gn; } inproj[8];
  std::vector<float> level_embed;             // host [L][D]
  std::vector<EncLayer> enc;
  Lin enc_output; Ln enc_output_ln; Lin enc_bbox[3];
  float* tgt = nullptr;
  std::vector<DecLayer> dec;
  Lin dec_kv_text, dec_value;                 // all decoder layers' text key|value and deformable value projections, concatenated
  Ln dec_ln; Lin ref_head[2]; std::vector<std::array<Lin, 3>> bbox;
  // ---- plans
  std::list<Plan*> plans;
  Plan* last = nullptr;
  const int* force_topk = nullptr;            // device int32 [num_queries] (tests: pin the two-stage selection)
  int graphs_enabled = 1;
  long launches_last = 0;
  // ---- second branch of the forward: the text side (BERT, the text enhancers) has no data dependence on the image side (Swin,
  // deformable attention) between their joins, so it runs on a stream of its own - in a captured plan two branches of the graph
  int branches = 1;                           // ovm_tune_set("gdino_branches", 0 | 1)
  hipStream_t aux = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
};

namespace {

#define GCHECK(g, call)                                                                     \
  do {                                                                                     \
    hipError_t e_ = (call);                                                                \
    if (e_ != hipSuccess) { (g)->err = std::string(#call) + ": " + hipGetErrorString(e_); return OVM_ERR_HIP; } \
  } while (0)
#define RCHECK(g, call)                                                                     \
  do {                                                                                     \
    int r_ = (call);                                                                       \
    if (r_ != OVM_OK) { if ((g)->err.empty()) (g)->err = std::string(#call) + " failed (" + std::to_string(r_) + ")"; return r_; } \
  } while (0)

typedef std::unordered_map<std::string, const OvmTensor*> WMap;

int64_t numel(const OvmTensor* t) { int64_t n = 1; for (int i = 0; i < t->ndim; ++i) n *= t->shape[i]; return n; }

int get(OvmGdino* g, const WMap& wm, const std::string& name, const OvmTensor** out) {
  auto it = wm.find(name);
  if (it == wm.end()) { g->err = "missing weight: " + name; return OVM_ERR_MISSING_WEIGHT; }
  *out = it->second;
  return OVM_OK;
}

// Weights and tables live in a few large slabs, not in one hipMalloc each: ~700 separate allocations scatter the checkpoint over as
// many small VM mappings, and the latency-bound kernels of this branch (every workgroup touches every page of a weight matrix once)
// then pay an address-translation miss per 4-KiB page; a slab is mapped with large fragments.
template <typename T>
int dmal(OvmGdino* g, T** p, size_t count) {
  size_t bytes = count * sizeof(T); if (bytes == 0) bytes = 16;
  bytes = (bytes + 255) & ~(size_t)255;
  if (g->slab_off + bytes > g->slab_cap) {
    const size_t cap = bytes > kSlabBytes ? bytes : kSlabBytes;
    void* q = nullptr;
    GCHECK(g, hipMalloc(&q, cap));
    g->allocs.push_back(q);
    g->slab = (char*)q; g->slab_cap = cap; g->slab_off = 0;
  }
  *p = (T*)(g->slab + g->slab_off);
  g->slab_off += bytes;
  return OVM_OK;
}

int up_vec(OvmGdino* g, const std::vector<float>& v, float** out) {
  RCHECK(g, dmal(g, out, v.size()));
  GCHECK(g, hipMemcpy(*out, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice));
  return OVM_OK;
}
int up_f32(OvmGdino* g, const WMap& wm, const std::string& name, float** out, int64_t expect = -1) {
  const OvmTensor* t; RCHECK(g, get(g, wm, name, &t));
  if (expect >= 0 && numel(t) != expect) { g->err = "bad shape: " + name; return OVM_ERR_SHAPE; }
  RCHECK(g, dmal(g, out, (size_t)numel(t)));
  GCHECK(g, hipMemcpy(*out, t->data, (size_t)numel(t) * sizeof(float), hipMemcpyHostToDevice));
  return OVM_OK;
}
int up_ln(OvmGdino* g, const WMap& wm, const std::string& prefix, Ln* ln) {
  RCHECK(g, up_f32(g, wm, prefix + ".weight", &ln->g));
  return up_f32(g, wm, prefix + ".bias", &ln->b);
}

// host rows [N][K] (+ bias [N] or empty) -> device weight image of gemm.hpp / gemm_small.hip
int pack_host(OvmGdino* g, const std::vector<float>& w, const std::vector<float>& bias, int N, int K, Lin* out) {
  const int Kpad = (K + 63) / 64 * 64, Npad = (N + 127) / 128 * 128;
  const bool il = g->npass == 3;
  const size_t ld = il ? (size_t)2 * Kpad : (size_t)Kpad;
  std::vector<half_t> buf((size_t)Npad * ld, (half_t)0.f);
  for (int n = 0; n < N; ++n)
    for (int k = 0; k < K; ++k) {
      const float x = w[(size_t)n * K + k];
      const half_t hh = (half_t)x;
      if (il) {
        const size_t o = (size_t)n * ld + (size_t)(k >> 5) * 64 + (k & 31);
        buf[o] = hh; buf[o + 32] = (half_t)(x - (float)hh);
      } else {
        buf[(size_t)n * ld + k] = hh;
      }
    }
  RCHECK(g, dmal(g, &out->hi, buf.size()));
  GCHECK(g, hipMemcpy(out->hi, buf.data(), buf.size() * sizeof(half_t), hipMemcpyHostToDevice));
  out->lo = il ? out->hi + 32 : nullptr;
  out->N = N; out->K = K; out->Kpad = Kpad;
  out->bias = nullptr;
  if (!bias.empty()) RCHECK(g, up_vec(g, bias, &out->bias));
  return OVM_OK;
}

// The row-chain kernels (dec_chain.hip) feed weight fragments from global memory straight into v_mfma_f32_16x16x32_f16: lane l wants
// row l % 16, k-chunk l / 16 of a 16-row tile - read from the row-major image that is 16 different rows per quarter-wave, 64 cache
// lines per load instruction with 16 bytes used of each (measured: ~13 us per 256 x 256 projection, 20 GB/s per CU). This copy holds
// the image in the order the lanes consume it: [tile of 16 rows][k-step of 32][hi | lo][lane 0..63][8 halves] - one load
// instruction = 1 KiB contiguous.
int make_frag(OvmGdino* g, Lin* w) {
  if (g->npass != 3 || !w->hi) return OVM_OK;
  const int Npad = (w->N + 127) / 128 * 128, KS = w->Kpad / 32;
  const size_t n = (size_t)Npad * 2 * w->Kpad;
  std::vector<half_t> src(n), dst(n);
  GCHECK(g, hipMemcpy(src.data(), w->hi, n * sizeof(half_t), hipMemcpyDeviceToHost));
  for (int tile = 0; tile < Npad / 16; ++tile)
    for (int ks = 0; ks < KS; ++ks)
      for (int part = 0; part < 2; ++part)
        for (int lane = 0; lane < 64; ++lane) {
          const size_t so = (size_t)(tile * 16 + (lane & 15)) * 2 * w->Kpad + (size_t)ks * 64 + part * 32 + (lane >> 4) * 8;
          const size_t dof = ((((size_t)tile * KS + ks) * 2 + part) * 64 + lane) * 8;
          for (int e = 0; e < 8; ++e) dst[dof + e] = src[so + e];
        }
  RCHECK(g, dmal(g, &w->frag, n));
  GCHECK(g, hipMemcpy(w->frag, dst.data(), n * sizeof(half_t), hipMemcpyHostToDevice));
  return OVM_OK;
}

// concatenation along N of several nn.Linear (weight [n_i][K], optional bias), optional per-row scale of every part
int pack_cat(OvmGdino* g, const WMap& wm, const std::vector<std::string>& prefixes, Lin* out, bool with_bias = true,
             const float* row_scale = nullptr) {
  std::vector<float> w, b;
  int K = -1, N = 0;
  for (auto& p : prefixes) {
    const OvmTensor* t; RCHECK(g, get(g, wm, p + ".weight", &t));
    const int n = (int)t->shape[0]; const int k = (int)(numel(t) / n);
    if (K < 0) K = k; else if (K != k) { g->err = "pack_cat: K mismatch at " + p; return OVM_ERR_SHAPE; }
    w.insert(w.end(), t->data, t->data + (size_t)n * k);
    if (with_bias) {
      const OvmTensor* bt; RCHECK(g, get(g, wm, p + ".bias", &bt));
      b.insert(b.end(), bt->data, bt->data + n);
    }
    N += n;
  }
  if (row_scale) {
    for (int n = 0; n < N; ++n) {
      for (int k = 0; k < K; ++k) w[(size_t)n * K + k] *= row_scale[n];
      if (with_bias) b[n] *= row_scale[n];
    }
  }
  return pack_host(g, w, b, N, K, out);
}
int pack_lin(OvmGdino* g, const WMap& wm, const std::string& prefix, Lin* out, bool with_bias = true) {
  return pack_cat(g, wm, {prefix}, out, with_bias);
}
// conv weight [Cout][Cin][k][k] -> rows [Cout][(ky*k + kx)*Cin + c]
int pack_conv(OvmGdino* g, const WMap& wm, const std::string& prefix, Lin* out, int* ksize) {
  const OvmTensor* t; RCHECK(g, get(g, wm, prefix + ".weight", &t));
  const OvmTensor* bt; RCHECK(g, get(g, wm, prefix + ".bias", &bt));
  if (t->ndim != 4) { g->err = "conv weight must be 4-d: " + prefix; return OVM_ERR_SHAPE; }
  const int Co = (int)t->shape[0], Ci = (int)t->shape[1], kh = (int)t->shape[2], kw = (int)t->shape[3];
  std::vector<float> w((size_t)Co * Ci * kh * kw);
  for (int o = 0; o < Co; ++o)
    for (int c = 0; c < Ci; ++c)
      for (int y = 0; y < kh; ++y)
        for (int x = 0; x < kw; ++x) w[((size_t)o * kh * kw + y * kw + x) * Ci + c] = t->data[(((size_t)o * Ci + c) * kh + y) * kw + x];
  std::vector<float> b(bt->data, bt->data + Co);
  if (ksize) *ksize = kh;
  return pack_host(g, w, b, Co, Ci * kh * kw, out);
}

int load_mha(OvmGdino* g, const WMap& wm, const std::string& p, int heads, Mha* m, bool cross) {
  m->heads = heads;
  if (cross) {
    RCHECK(g, pack_lin(g, wm, p + "query", &m->q));
  } else {
    RCHECK(g, pack_cat(g, wm, {p + "query", p + "key"}, &m->qk));
    RCHECK(g, pack_lin(g, wm, p + "value", &m->v));
  }
  return pack_lin(g, wm, p + "out_proj", &m->out);
}
int load_msda(OvmGdino* g, const WMap& wm, const std::string& p, MsdaW* m, bool with_value) {
  RCHECK(g, pack_cat(g, wm, {p + "sampling_offsets", p + "attention_weights"}, &m->offw));
  if (with_value) RCHECK(g, pack_lin(g, wm, p + "value_proj", &m->value));
  return pack_lin(g, wm, p + "output_proj", &m->out);
}

// ------------------------------------------------------------------------------------------------------------------------------
// Plan: everything derived from (H, W, token ids, position ids)
// ------------------------------------------------------------------------------------------------------------------------------
struct WinMaps { int* win = nullptr; float* mask = nullptr; int nW = 0; };
struct StageGeo { int h = 0, w = 0; WinMaps wm[2]; int* merge = nullptr; int h2 = 0, w2 = 0; };

struct Plan {
  int H = 0, W = 0, T = 0;
  std::vector<int> ids, pids;
  std::vector<void*> allocs; size_t bytes = 0;      // device memory this plan holds (the plan cache's budget counts it)
  // text
  int* d_ids = nullptr; int* d_pids = nullptr; float* text_bias = nullptr; float* text_pos = nullptr;
  // swin
  int Hp = 0, Wp = 0; int* pe_map = nullptr;
  std::vector<StageGeo> geo;
  // neck / encoder tables
  int nlev = 0; int lh[8] = {0}, lw[8] = {0}, lstart[8] = {0}; int S = 0;
  int* conv_map = nullptr; int conv_h = 0, conv_w = 0;
  float* pos = nullptr; float* ref = nullptr; float* prop_logit = nullptr; int* valid_idx = nullptr;
  // scratch owned by the plan
  float* img = nullptr;                          // normalised input image [H*W][3]
  char* arena = nullptr; size_t arena_cap = 0;
  float* gemm_ws = nullptr; size_t gemm_ws_cap = 0;     // split-K partials (both GEMM kernels)
  unsigned long long* topk_keys = nullptr; int topk_N = 0;
  float* out_logits = nullptr; float* out_boxes = nullptr;
  // post-processing scratch (ovm_gdino_detect)
  // debug taps (pointers into the arena, valid after a forward)
  std::map<std::string, std::pair<const void*, int64_t>> taps;
  hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr;
  long launches = 0;
  ~Plan() {
    if (exec) (void)hipGraphExecDestroy(exec);
    if (graph) (void)hipGraphDestroy(graph);
    for (void* p : allocs) (void)hipFree(p);
  }
};

template <typename T>
int pup(OvmGdino* g, Plan* pl, const std::vector<T>& v, T** out) {
  void* q = nullptr;
  size_t bytes = v.size() * sizeof(T); if (bytes == 0) bytes = 16;
  GCHECK(g, hipMalloc(&q, bytes));
  pl->allocs.push_back(q); pl->bytes += bytes;
  if (!v.empty()) GCHECK(g, hipMemcpy(q, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  *out = (T*)q;
  return OVM_OK;
}
template <typename T>
int pal(OvmGdino* g, Plan* pl, T** out, size_t count) {
  void* q = nullptr;
  size_t bytes = count * sizeof(T); if (bytes == 0) bytes = 16;
  GCHECK(g, hipMalloc(&q, bytes));
  pl->allocs.push_back(q); pl->bytes += bytes;
  *out = (T*)q;
  return OVM_OK;
}

// GroundingDINO generate_masks_with_special_tokens_and_transfer_map: tokens attend inside their own sub-sentence (delimited by
// [CLS] [SEP] . ?); position ids restart per phrase, the closing delimiter included (upstream numbering)
void text_masks(const std::vector<int>& ids, std::vector<char>* mask, std::vector<int>* pos) {
  const int T = (int)ids.size();
  mask->assign((size_t)T * T, 0);
  pos->assign(T, 0);
  for (int i = 0; i < T; ++i) (*mask)[(size_t)i * T + i] = 1;
  int prev = 0;
  for (int col = 0; col < T; ++col) {
    const int t = ids[col];
    if (!(t == 101 || t == 102 || t == 1012 || t == 1029)) continue;
    if (col == 0 || col == T - 1) {
      (*mask)[(size_t)col * T + col] = 1; (*pos)[col] = 0;
    } else {
      for (int a = prev + 1; a <= col; ++a) {
        for (int b = prev + 1; b <= col; ++b) (*mask)[(size_t)a * T + b] = 1;
        (*pos)[a] = a - prev - 1;
      }
    }
    prev = col;
  }
}

void window_maps(int H, int W, int ws, int shift, std::vector<int>* win, std::vector<float>* mask, int* nW) {
  const int Hp = (H + ws - 1) / ws * ws, Wp = (W + ws - 1) / ws * ws, nwh = Hp / ws, nww = Wp / ws, ws2 = ws * ws;
  win->assign((size_t)nwh * nww * ws2, -1);
  for (int y = 0; y < Hp; ++y)
    for (int x = 0; x < Wp; ++x) {
      const int sy = (y + shift) % Hp, sx = (x + shift) % Wp;      // source (padded) coordinates of shifted-map position (y, x)
      const int src = (sy < H && sx < W) ? sy * W + sx : -1;
      (*win)[((size_t)(y / ws) * nww + x / ws) * ws2 + (y % ws) * ws + x % ws] = src;
    }
  *nW = nwh * nww;
  mask->clear();
  if (shift > 0) {
    std::vector<int> img((size_t)Hp * Wp);
    for (int y = 0; y < Hp; ++y)
      for (int x = 0; x < Wp; ++x) {
        const int hr = (y >= Hp - ws) + (y >= Hp - shift), wr = (x >= Wp - ws) + (x >= Wp - shift);
        img[(size_t)y * Wp + x] = hr * 3 + wr;
      }
    mask->assign((size_t)nwh * nww * ws2 * ws2, 0.f);
    for (int wy = 0; wy < nwh; ++wy)
      for (int wx = 0; wx < nww; ++wx) {
        const size_t base = ((size_t)wy * nww + wx) * ws2 * ws2;
        for (int a = 0; a < ws2; ++a) {
          const int ia = img[(size_t)(wy * ws + a / ws) * Wp + wx * ws + a % ws];
          for (int b = 0; b < ws2; ++b) {
            const int ib = img[(size_t)(wy * ws + b / ws) * Wp + wx * ws + b % ws];
            (*mask)[base + (size_t)a * ws2 + b] = (ia != ib) ? -100.0f : 0.f;
          }
        }
      }
  }
}

// GroundingDINO PositionEmbeddingSineHW with an all-valid mask, [h*w][2*dhalf] = (pos_y | pos_x); float32 arithmetic as torch
void sine_pos(int h, int w, int dhalf, float temperature, std::vector<float>* out) {
  out->assign((size_t)h * w * 2 * dhalf, 0.f);
  const float eps = 1e-6f, scale = 2.0f * 3.14159265358979323846f;
  std::vector<float> dim_t(dhalf);
  for (int i = 0; i < dhalf; ++i) dim_t[i] = powf(temperature, 2.0f * (float)(i / 2) / (float)dhalf);
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) {
      const float ye = (float)(y + 1) / ((float)h + eps) * scale, xe = (float)(x + 1) / ((float)w + eps) * scale;
      float* o = out->data() + ((size_t)y * w + x) * 2 * dhalf;
      for (int i = 0; i < dhalf; ++i) {
        const float py = ye / dim_t[i], px = xe / dim_t[i];
        o[i] = (i & 1) ? cosf(py) : sinf(py);
        o[dhalf + i] = (i & 1) ? cosf(px) : sinf(px);
      }
    }
}

int build_plan(OvmGdino* g, int H, int W, const std::vector<int>& ids, const std::vector<int>& pids_in, Plan** out) {
  const OvmGdinoConfig& c = g->cfg;
  Plan* pl = new Plan();
  std::unique_ptr<Plan> guard(pl);
  pl->H = H; pl->W = W; pl->T = (int)ids.size(); pl->ids = ids;
  const int T = pl->T, D = c.d_model;
  // ---- text tables
  std::vector<char> mask; std::vector<int> pids;
  text_masks(ids, &mask, &pids);
  if (!pids_in.empty()) pids = pids_in;
  pl->pids = pids_in;
  for (int t = 0; t < T; ++t) {
    if (ids[t] < 0 || ids[t] >= g->vocab) { g->err = "token id out of the vocabulary"; return OVM_ERR_INVALID; }
    if (pids[t] < 0 || pids[t] >= g->n_pos) { g->err = "position id out of range"; return OVM_ERR_INVALID; }
  }
  RCHECK(g, pup(g, pl, ids, &pl->d_ids));
  RCHECK(g, pup(g, pl, pids, &pl->d_pids));
  std::vector<float> bias((size_t)T * T);
  for (size_t i = 0; i < bias.size(); ++i) bias[i] = mask[i] ? 0.f : -3.4028234663852886e38f;     // torch.finfo(float32).min
  RCHECK(g, pup(g, pl, bias, &pl->text_bias));
  {
    std::vector<float> pf(T); for (int t = 0; t < T; ++t) pf[t] = (float)pids[t];
    float* d_pf; RCHECK(g, pup(g, pl, pf, &d_pf));
    RCHECK(g, pal(g, pl, &pl->text_pos, (size_t)T * D));
    RCHECK(g, ovm_g_sine_embed(d_pf, T, 1, D, 10000.0f, pl->text_pos, nullptr));
    GCHECK(g, hipDeviceSynchronize());
  }
  // ---- swin geometry
  const int P = 4, ws = c.swin_window;
  pl->Hp = (H + P - 1) / P; pl->Wp = (W + P - 1) / P;
  {
    std::vector<int> pm((size_t)pl->Hp * pl->Wp * P * P);
    for (int oy = 0; oy < pl->Hp; ++oy)
      for (int ox = 0; ox < pl->Wp; ++ox)
        for (int py = 0; py < P; ++py)
          for (int px = 0; px < P; ++px) {
            const int y = oy * P + py, x = ox * P + px;
            pm[((size_t)oy * pl->Wp + ox) * P * P + py * P + px] = (y < H && x < W) ? y * W + x : -1;
          }
    RCHECK(g, pup(g, pl, pm, &pl->pe_map));
  }
  int h = pl->Hp, w = pl->Wp;
  pl->geo.resize(g->stages.size());
  std::vector<std::pair<int, int>> feat_hw;
  for (size_t s = 0; s < g->stages.size(); ++s) {
    StageGeo& ge = pl->geo[s];
    ge.h = h; ge.w = w;
    for (int sh = 0; sh < 2; ++sh) {
      std::vector<int> win; std::vector<float> mk; int nW;
      window_maps(h, w, ws, sh ? ws / 2 : 0, &win, &mk, &nW);
      ge.wm[sh].nW = nW;
      RCHECK(g, pup(g, pl, win, &ge.wm[sh].win));
      if (!mk.empty()) RCHECK(g, pup(g, pl, mk, &ge.wm[sh].mask));
    }
    if (g->stages[s].has_out) feat_hw.push_back({h, w});
    if (g->stages[s].has_red) {
      const int h2 = (h + 1) / 2, w2 = (w + 1) / 2;
      std::vector<int> mm((size_t)h2 * w2 * 4, -1);
      for (int oy = 0; oy < h2; ++oy)
        for (int ox = 0; ox < w2; ++ox) {
          int k = 0;
          for (int col = 0; col < 2; ++col)                       // HF order: for col in 2: for row in 2
            for (int row = 0; row < 2; ++row) {
              const int y = 2 * oy + row, x = 2 * ox + col;
              mm[((size_t)oy * w2 + ox) * 4 + k++] = (y < h && x < w) ? y * w + x : -1;
            }
        }
      RCHECK(g, pup(g, pl, mm, &ge.merge));
      ge.h2 = h2; ge.w2 = w2;
      h = h2; w = w2;
    }
  }
  // ---- levels
  pl->nlev = c.n_levels;
  int nfeat = (int)feat_hw.size();
  if (nfeat > c.n_levels || c.n_levels > 8) { g->err = "level count"; return OVM_ERR_SHAPE; }
  int st = 0, lh = 0, lw = 0;
  for (int l = 0; l < c.n_levels; ++l) {
    if (l < nfeat) { lh = feat_hw[l].first; lw = feat_hw[l].second; }
    else {
      const int h2 = (lh + 2 - 3) / 2 + 1, w2 = (lw + 2 - 3) / 2 + 1;
      if (l == nfeat) {                                            // 3x3 stride-2 pad-1 conv on the last backbone stage: im2col map
        std::vector<int> cm((size_t)h2 * w2 * 9, -1);
        for (int oy = 0; oy < h2; ++oy)
          for (int ox = 0; ox < w2; ++ox)
            for (int ky = 0; ky < 3; ++ky)
              for (int kx = 0; kx < 3; ++kx) {
                const int y = 2 * oy + ky - 1, x = 2 * ox + kx - 1;
                cm[((size_t)oy * w2 + ox) * 9 + ky * 3 + kx] = (y >= 0 && y < lh && x >= 0 && x < lw) ? y * lw + x : -1;
              }
        RCHECK(g, pup(g, pl, cm, &pl->conv_map));
        pl->conv_h = lh; pl->conv_w = lw;
      } else { g->err = "more than one extra feature level is not supported"; return OVM_ERR_SHAPE; }
      lh = h2; lw = w2;
    }
    pl->lh[l] = lh; pl->lw[l] = lw; pl->lstart[l] = st; st += lh * lw;
  }
  pl->S = st;
  const int S = pl->S;
  {
    std::vector<float> pos((size_t)S * D), ref((size_t)S * 2), prop((size_t)S * 4);
    std::vector<int> valid(S);
    for (int l = 0; l < c.n_levels; ++l) {
      const int hh = pl->lh[l], ww = pl->lw[l];
      std::vector<float> sp; sine_pos(hh, ww, D / 2, c.pe_temperature, &sp);
      for (int i = 0; i < hh * ww; ++i) {
        float* o = pos.data() + (size_t)(pl->lstart[l] + i) * D;
        for (int d = 0; d < D; ++d) o[d] = sp[(size_t)i * D + d] + g->level_embed[(size_t)l * D + d];
        const int y = i / ww, x = i % ww;
        // reference points: linspace(0.5, n - 0.5, n) / n (valid ratios are 1: no padding)
        const float rx = ((float)x + 0.5f) / (float)ww, ry = ((float)y + 0.5f) / (float)hh;
        ref[(size_t)(pl->lstart[l] + i) * 2] = rx; ref[(size_t)(pl->lstart[l] + i) * 2 + 1] = ry;
        // proposals: ((grid + 0.5) / (w, h), 0.05 * 2^l)
        const float gx = ((float)x + 0.5f) / (float)ww, gy = ((float)y + 0.5f) / (float)hh, wh = 0.05f * (float)(1 << l);
        const float pr[4] = {gx, gy, wh, wh};
        bool ok = true;
        for (int k = 0; k < 4; ++k) ok = ok && (pr[k] > 0.01f) && (pr[k] < 0.99f);
        for (int k = 0; k < 4; ++k) prop[(size_t)(pl->lstart[l] + i) * 4 + k] = ok ? logf(pr[k] / (1.0f - pr[k])) : INFINITY;
        valid[pl->lstart[l] + i] = ok ? pl->lstart[l] + i : -1;
      }
    }
    RCHECK(g, pup(g, pl, pos, &pl->pos));
    RCHECK(g, pup(g, pl, ref, &pl->ref));
    RCHECK(g, pup(g, pl, prop, &pl->prop_logit));
    RCHECK(g, pup(g, pl, valid, &pl->valid_idx));
  }
  if (S < c.num_queries) {       // torch.topk in the upstream two-stage selection raises the same way
    g->err = "selected index k out of range: " + std::to_string(S) + " encoder tokens < " + std::to_string(c.num_queries) + " queries (image too small)";
    return OVM_ERR_SHAPE;
  }
  RCHECK(g, pal(g, pl, &pl->img, (size_t)H * W * 3));
  pl->topk_N = 2048; while (pl->topk_N < S) pl->topk_N <<= 1;      // the bitonic sort's minimum length is one 2048-key tile
  RCHECK(g, pal(g, pl, &pl->topk_keys, (size_t)pl->topk_N));
  RCHECK(g, pal(g, pl, &pl->out_logits, (size_t)c.num_queries * c.max_text_len));
  RCHECK(g, pal(g, pl, &pl->out_boxes, (size_t)c.num_queries * 4));
  pl->gemm_ws_cap = (size_t)64 << 20;
  RCHECK(g, pal(g, pl, (char**)&pl->gemm_ws, pl->gemm_ws_cap));
  guard.release();
  *out = pl;
  return OVM_OK;
}

// split-K workspace of the text branch (tail of the plan's workspace): its largest user is BERT's output projection at the
// maximum caption length, 12 slices x 256 tokens x 768 columns of fp32 partials = 9.4 MB
constexpr size_t kAuxWs = (size_t)16 << 20;

// ------------------------------------------------------------------------------------------------------------------------------
// One forward over a plan. `dry` = size the arena only (no launches).
// ------------------------------------------------------------------------------------------------------------------------------
struct Run {
  OvmGdino* g; Plan* pl; hipStream_t s; bool dry;
  size_t off = 0, peak = 0;
  long launches = 0;
  int rc = OVM_OK;
  // the two branches (see OvmGdino::aux). `s` is the stream the op wrappers launch on; fork() lets the text branch start from
  // the current point of the main stream, join() makes the main stream wait for it. Each branch has its own slice of the split-K
  // workspace. With branches off (or in the sizing pass) everything stays on the caller's stream.
  hipStream_t s_main = nullptr;
  float* ws = nullptr; size_t ws_cap = 0;
  bool two() const { return g->branches && g->aux && !dry; }
  void init_streams() { s_main = s; ws = pl->gemm_ws; ws_cap = two() ? pl->gemm_ws_cap - kAuxWs : pl->gemm_ws_cap; }
  void fork() {
    if (!two() || rc != OVM_OK) return;
    if (hipEventRecord(g->ev_fork, s_main) != hipSuccess || hipStreamWaitEvent(g->aux, g->ev_fork, 0) != hipSuccess) fail(OVM_ERR_HIP, "fork");
  }
  void on_text() { if (two()) { s = g->aux; ws = (float*)((char*)pl->gemm_ws + (pl->gemm_ws_cap - kAuxWs)); ws_cap = kAuxWs; } }
  void on_image() { if (two()) { s = s_main; ws = pl->gemm_ws; ws_cap = pl->gemm_ws_cap - kAuxWs; } }
  void join() {
    if (!two()) return;
    on_image();
    if (rc != OVM_OK) return;
    if (hipEventRecord(g->ev_join, g->aux) != hipSuccess || hipStreamWaitEvent(s_main, g->ev_join, 0) != hipSuccess) fail(OVM_ERR_HIP, "join");
  }

  void* alloc(size_t bytes) {
    off = (off + 255) & ~(size_t)255;
    void* p = dry ? (void*)(uintptr_t)(0x1000 + off) : (void*)(pl->arena + off);
    off += bytes;
    if (off > peak) peak = off;
    if (!dry && off > pl->arena_cap) { fail(OVM_ERR_CAPACITY, "arena overflow"); return pl->arena; }
    return p;
  }
  float* f32(size_t n) { return (float*)alloc(n * sizeof(float)); }
  int* i32(size_t n) { return (int*)alloc(n * sizeof(int)); }
  // split-fp16 rows of logical width K; the row stride is K rounded up to the GEMM's k-step (64). Producers write columns
  // [0, K) only, so when a pad exists (K = 32 or 48: test-size models, the 4x4x3 patch rows) the buffer is cleared first -
  // arena memory is recycled and NaN bit patterns in the pad would survive the multiplication by the zero weight columns.
  SplitBuf split(size_t rows, int K) {
    SplitBuf b; b.ld = (K + 63) / 64 * 64;
    const size_t bytes = rows * b.ld * sizeof(half_t);
    b.hi = (half_t*)alloc(bytes);
    b.lo = g->npass == 3 ? (half_t*)alloc(bytes) : nullptr;
    if (b.ld != K && go()) {
      if (hipMemsetAsync(b.hi, 0, bytes, s) != hipSuccess) fail(OVM_ERR_HIP, "memset");
      if (b.lo && hipMemsetAsync(b.lo, 0, bytes, s) != hipSuccess) fail(OVM_ERR_HIP, "memset");
    }
    return b;
  }
  // Interleaved split rows for the operands of the 256 x 256 GEMM (gemm256.hip): the encoder's and Swin stage 1's wide contractions
  // over K <= 256 (752 / 564 / 556 tiles of 128 x 128, i.e. 2-3 rounds of a kernel whose per-round cost hardly depends on K) are one
  // round of 256 x 256 tiles there. Falls back to planar rows when the kernel cannot take the shape (one-pass precision, K % 32).
  SplitBuf split_for256(size_t rows, int K) {
    if (g->npass != 3 || K % 32 || !g_gdino_gemm256) return split(rows, K);
    SplitBuf b; b.il = true; b.ld = 2 * K;
    b.hi = (half_t*)alloc(rows * b.ld * sizeof(half_t)); b.lo = b.hi ? b.hi + 32 : nullptr;
    return b;
  }
  size_t mark() const { return off; }
  void release(size_t m) { off = m; }
  void fail(int r, const char* what) { if (rc == OVM_OK) { rc = r; if (g->err.empty()) g->err = what; } }
  void chk(int r, const char* what) { ++launches; if (r != OVM_OK) fail(r, what); }
  void tap(const char* name, const void* p, int64_t n) { if (!dry) pl->taps[name] = {p, n}; }

  // ---- op wrappers (all skip the launch in a dry pass) ----
  bool go() const { return !dry && rc == OVM_OK; }      // after a failure nothing further is launched (later kernels would read its garbage)
  void rowop(RowOpParams p) { if (go()) chk(launch_rowop(p, s), "rowop"); }

  // big GEMM on the LDS-DMA kernels of gemm.hpp: A split fp16 [M][lda]
  GemmParams gp(const SplitBuf& A, int M, const Lin& W) {
    GemmParams p; memset(&p, 0, sizeof(p));
    p.Ahi = A.hi; p.Alo = A.lo; p.lda = A.ld; p.a_il = A.il ? 1 : 0; p.Whi = W.hi; p.Wlo = W.lo; p.M = M; p.N = W.N; p.K = W.Kpad; p.bias = W.bias;
    p.ws_slot = 1; p.part_ws = ws; p.part_cap = ws_cap;
    return p;
  }
  void gemm(const GemmParams& p, int epi) {
    if (!go()) return;
    if (p.a_il && gemm256_supported(p, g->npass)) chk(launch_gemm256(p, epi, 1, s), "gemm256");
    else chk(launch_gemm(p, g->npass, epi, A_ROWMAJOR, s), "gemm");
  }

  // small / mid GEMM reading fp32 activations directly (gemm_small.hip): y = act((A + A2) W^T + b) (+ R)
  void lin(const float* A, const float* A2, int lda, int M, const Lin& W, int act, const float* R, int ldr, float* C, int ldc) {
    if (!go() || M <= 0) return;
    if (!gemm_small_supported(A, lda, W.K) || (A2 && (((uintptr_t)A2) & 15))) { fail(OVM_ERR_SHAPE, "lin: unaligned fp32 operand"); return; }
    chk(launch_gemm_small_ex(A, A2, lda, M, W.K, W.hi, W.lo, W.N, W.Kpad, W.bias, act, R, ldr, C, ldc, g->npass, ws, ws_cap, s),
        "lin");
  }
  void ln(const float* x, int M, int D, const Ln& w, float eps, const float* res, float* y, SplitBuf* sp = nullptr) {
    RowOpParams p; memset(&p, 0, sizeof(p));
    p.x = x; p.ldx = D; p.res = res; p.ldr = D; p.gamma = w.g; p.beta = w.b; p.eps = eps; p.M = M; p.D = D; p.y = y; p.ldy = D;
    if (sp) { p.hi = sp->hi; p.lo = sp->lo; p.ldh = sp->ld; p.il = sp->il ? 1 : 0; }
    rowop(p);
  }
  void attn(AttnF32Params p) { if (go()) chk(launch_attn_f32(p, s), "attn_f32"); }
};

inline int kpad64(int k) { return (k + 63) / 64 * 64; }

// multi-head attention over fp32 rows: q [Tq][D] (at qp, stride ldq), k / v likewise
void mha_core(Run& r, const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, int Tq, int Tk, int heads, int D, const float* bias,
              int ldb, float* out, int ldo) {
  AttnF32Params a; memset(&a, 0, sizeof(a));
  const int dh = D / heads;
  a.q = q; a.k = k; a.v = v; a.ldq = ldq; a.ldk = ldk; a.ldv = ldv;
  a.sq2 = dh; a.sk2 = dh; a.sv2 = dh; a.o = out; a.ldo = ldo; a.so2 = dh;
  a.nb1 = 1; a.nb2 = heads; a.Tq = Tq; a.Tk = Tk; a.DH = dh; a.scale = 1.0f / sqrtf((float)dh);
  a.bias_b = bias; a.sbb = 0; a.ldbb = ldb;
  r.attn(a);
}

void deform(Run& r, const MsdaW& w, const float* value, int ldv, const float* ow, int ldow, const float* ref, int ldref, int mode, int Q,
            float* out, SplitBuf* osp) {
  if (!r.go()) return;
  const OvmGdinoConfig& c = r.g->cfg;
  MsDeformParams p; memset(&p, 0, sizeof(p));
  p.value = value; p.ldv = ldv; p.ow = ow; p.ldow = ldow; p.ref = ref; p.ldref = ldref; p.mode = mode;
  p.Q = Q; p.H = c.heads; p.dh = c.d_model / c.heads; p.L = c.n_levels; p.P = c.n_points;
  for (int l = 0; l < c.n_levels; ++l) { p.lh[l] = r.pl->lh[l]; p.lw[l] = r.pl->lw[l]; p.lstart[l] = r.pl->lstart[l]; }
  p.out = out; p.ldo = c.d_model;
  if (osp) { p.ohi = osp->hi; p.olo = osp->lo; p.ldoh = osp->ld; }
  (void)w;
  r.chk(launch_msdeform_fused(p, r.s), "msdeform");
}

int forward_impl(Run& r) {
  OvmGdino* g = r.g; Plan* pl = r.pl;
  r.init_streams();
  const hipStream_t s = r.s_main;                        // image branch / joined sections; text-branch launches use r.s
  const OvmGdinoConfig& c = g->cfg;
  const int D = c.d_model, T = pl->T, S = pl->S, Q = c.num_queries;
  const float eps = c.eps;
  const bool dry = r.dry;

  // =============================== text: BERT + projection ===============================
  // (text branch: runs beside the Swin backbone and the neck below, joined before the encoder)
  const int BD = g->bertD, BH = c.bert_heads;
  float* tx = r.f32((size_t)T * BD);
  r.fork(); r.on_text();
  if (r.go()) r.chk(launch_bert_embed(g->word, g->posemb, g->typemb, pl->d_ids, pl->d_pids, T, BD, g->emb_ln.g, g->emb_ln.b, 1e-12f, tx, r.s), "bert_embed");
  {
    float* qkv = r.f32((size_t)T * 3 * BD);
    float* ctx = r.f32((size_t)T * BD);
    float* a = r.f32((size_t)T * BD);
    float* hbuf = r.f32((size_t)T * 4 * BD);
    for (auto& ly : g->bert) {
      r.lin(tx, nullptr, BD, T, ly.qkv, 0, nullptr, 0, qkv, 3 * BD);
      mha_core(r, qkv, 3 * BD, qkv + BD, 3 * BD, qkv + 2 * BD, 3 * BD, T, T, BH, BD, pl->text_bias, T, ctx, BD);
      r.lin(ctx, nullptr, BD, T, ly.ao, 0, nullptr, 0, a, BD);
      r.ln(a, T, BD, ly.aln, 1e-12f, tx, tx);
      r.lin(tx, nullptr, BD, T, ly.fi, 2, nullptr, 0, hbuf, ly.fi.N);
      r.lin(hbuf, nullptr, ly.fi.N, T, ly.fo, 0, nullptr, 0, a, BD);
      r.ln(a, T, BD, ly.oln, 1e-12f, tx, tx);
    }
  }
  r.tap("bert_out", tx, (int64_t)T * BD);
  float* text0 = r.f32((size_t)T * D);
  r.lin(tx, nullptr, BD, T, g->text_proj, 0, nullptr, 0, text0, D);
  r.tap("text_features", text0, (int64_t)T * D);
  float* text = r.f32((size_t)T * D);                     // encoder output (the input above stays intact for the debug tap)
  r.on_image();

  // =============================== image: Swin backbone ===============================
  const int ws = c.swin_window, ws2 = ws * ws;
  int h = pl->Hp, w = pl->Wp;
  int C = g->stages[0].C;
  // residual streams of the stages live for the whole backbone (the output norms read them at the end of each stage)
  float* x = r.f32((size_t)h * w * C);
  {
    const size_t mk = r.mark();
    SplitBuf pa = r.split((size_t)h * w, 48);
    RowOpParams p; memset(&p, 0, sizeof(p));
    p.x = pl->img; p.ldx = 3; p.idx = pl->pe_map; p.nidx = 16; p.seg = 3; p.M = h * w; p.D = 48; p.hi = pa.hi; p.lo = pa.lo; p.ldh = pa.ld;
    r.rowop(p);
    float* y = r.f32((size_t)h * w * C);
    GemmParams q = r.gp(pa, h * w, g->pe); q.C = y; q.ldc = C;
    r.gemm(q, EPI_STORE);
    r.ln(y, h * w, C, g->pe_ln, eps, nullptr, x);
    r.release(mk);
  }
  struct Feat { float* f; int h, w, C; };
  std::vector<Feat> feats;
  for (size_t si = 0; si < g->stages.size(); ++si) {
    SwinStage& stg = g->stages[si];
    const StageGeo& ge = pl->geo[si];
    const int nh = stg.nh, dh = C / nh, ntok = h * w;
    for (size_t b = 0; b < stg.blocks.size(); ++b) {
      SwinBlock& blk = stg.blocks[b];
      const WinMaps& wmaps = ge.wm[b & 1];
      const int nW = wmaps.nW, M = nW * ws2;
      const size_t mk = r.mark();
      // LN1 + pad + cyclic shift + window partition -> split fp16 rows (padding rows are zero AFTER the norm)
      SplitBuf xw = r.split((size_t)M, C);
      {
        RowOpParams p; memset(&p, 0, sizeof(p));
        p.x = x; p.ldx = C; p.idx = wmaps.win; p.nidx = 1; p.seg = C; p.gamma = blk.ln1.g; p.beta = blk.ln1.b; p.eps = eps; p.zero_masked = 1;
        p.M = M; p.D = C; p.hi = xw.hi; p.lo = xw.lo; p.ldh = xw.ld;
        r.rowop(p);
      }
      SplitBuf ctx = r.split((size_t)M, C);
      const bool fused = g_gdino_swin_fused && blk.qkv.frag && dh == 32 && swin_qkv_attn_supported(C, nh, ws, g->npass) && xw.ld % 8 == 0;
      if (fused) {
        // qkv projection inside the window kernel (one workgroup per (window, head)): no [M][3 C] fp32 round trip, one launch less
        SwinQkvAttnParams a; memset(&a, 0, sizeof(a));
        a.xhi = xw.hi; a.xlo = xw.lo; a.ldx = xw.ld; a.wfrag = blk.qkv.frag; a.KS = blk.qkv.Kpad / 32; a.bias = blk.qkv.bias;
        a.C = C; a.nh = nh; a.nW = nW; a.relbias = blk.relbias; a.mask = wmaps.mask; a.ohi = ctx.hi; a.olo = ctx.lo; a.ldo = ctx.ld;
        a.scale = 1.0f / sqrtf((float)dh);
        if (r.go()) r.chk(launch_swin_qkv_attn(a, s), "swin_qkv_attn");
      } else {
      float* qkv = r.f32((size_t)M * 3 * C);
      { GemmParams q = r.gp(xw, M, blk.qkv); q.C = qkv; q.ldc = 3 * C; r.gemm(q, EPI_STORE); }
      // window attention: QK^T + relative-position bias + shift mask + softmax + PV, one workgroup per (window, head)
      {
        AttnF32Params a; memset(&a, 0, sizeof(a));
        a.q = qkv; a.k = qkv + C; a.v = qkv + 2 * C; a.ldq = a.ldk = a.ldv = 3 * C;
        a.sq1 = a.sk1 = a.sv1 = (long)ws2 * 3 * C; a.sq2 = a.sk2 = a.sv2 = dh;
        a.ohi = ctx.hi; a.olo = ctx.lo; a.ldoh = ctx.ld; a.soh1 = (long)ws2 * ctx.ld; a.soh2 = dh;
        a.nb1 = nW; a.nb2 = nh; a.Tq = ws2; a.Tk = ws2; a.DH = dh; a.scale = 1.0f / sqrtf((float)dh);
        a.bias_h = blk.relbias; a.sbh = (long)ws2 * ws2; a.ldbh = ws2;
        a.bias_b = wmaps.mask; a.sbb = (long)ws2 * ws2; a.ldbb = ws2;
        r.attn(a);
      }
      }
      // output projection; the epilogue un-partitions / un-shifts / crops through the same index map and adds the shortcut
      { GemmParams q = r.gp(ctx, M, blk.proj); q.X = x; q.ldx = C; q.row_map = wmaps.win; r.gemm(q, EPI_RESID); }
      // stage 1 (17,689 tokens x 512 outputs over K = 128: 556 tiles of 128 x 128, or 140 of 256 x 256 in one round): interleaved rows
      const bool wide = ((ntok + 127) / 128) * ((4 * C + 127) / 128) > 512 && (4 * C) % 256 == 0;
      SplitBuf hn = wide ? r.split_for256((size_t)ntok, C) : r.split((size_t)ntok, C);
      r.ln(x, ntok, C, blk.ln2, eps, nullptr, nullptr, &hn);
      SplitBuf f1 = r.split((size_t)ntok, 4 * C);
      { GemmParams q = r.gp(hn, ntok, blk.fc1); q.Ohi = f1.hi; q.Olo = f1.lo; q.ldo = f1.ld; r.gemm(q, EPI_GELU); }
      { GemmParams q = r.gp(f1, ntok, blk.fc2); q.X = x; q.ldx = C; r.gemm(q, EPI_RESID); }
      r.release(mk);
    }
    if (stg.has_out) {
      float* f = r.f32((size_t)ntok * C);
      r.ln(x, ntok, C, stg.on, eps, nullptr, f);
      feats.push_back({f, h, w, C});
      r.tap(("swin_stage" + std::to_string(feats.size())).c_str(), f, (int64_t)ntok * C);
    }
    if (stg.has_red) {
      const int h2 = ge.h2, w2 = ge.w2;
      float* xn = r.f32((size_t)h2 * w2 * 2 * C);
      const size_t mk = r.mark();
      SplitBuf xm = r.split((size_t)h2 * w2, 4 * C);
      RowOpParams p; memset(&p, 0, sizeof(p));
      p.x = x; p.ldx = C; p.idx = ge.merge; p.nidx = 4; p.seg = C; p.gamma = stg.dn.g; p.beta = stg.dn.b; p.eps = eps;
      p.M = h2 * w2; p.D = 4 * C; p.hi = xm.hi; p.lo = xm.lo; p.ldh = xm.ld;
      r.rowop(p);
      GemmParams q = r.gp(xm, h2 * w2, stg.red); q.C = xn; q.ldc = 2 * C;
      r.gemm(q, EPI_STORE);
      r.release(mk);
      x = xn; h = h2; w = w2; C = 2 * C;
    }
  }

  // =============================== neck: input projections + GroupNorm ===============================
  float* vis0 = r.f32((size_t)S * D);
  float* vis = r.f32((size_t)S * D);                      // encoder state / output
  {
    const size_t mk = r.mark();
    for (int l = 0; l < c.n_levels; ++l) {
      const int n = pl->lh[l] * pl->lw[l];
      float* y = r.f32((size_t)n * D);
      if (l < (int)feats.size()) {
        const Feat& f = feats[l];
        SplitBuf a = r.split((size_t)n, f.C);
        RowOpParams p; memset(&p, 0, sizeof(p));
        p.x = f.f; p.ldx = f.C; p.M = n; p.D = f.C; p.hi = a.hi; p.lo = a.lo; p.ldh = a.ld;
        r.rowop(p);
        GemmParams q = r.gp(a, n, g->inproj[l].w); q.C = y; q.ldc = D;
        r.gemm(q, EPI_STORE);
      } else {
        const Feat& f = feats.back();
        SplitBuf a = r.split((size_t)n, 9 * f.C);
        RowOpParams p; memset(&p, 0, sizeof(p));
        p.x = f.f; p.ldx = f.C; p.idx = pl->conv_map; p.nidx = 9; p.seg = f.C; p.M = n; p.D = 9 * f.C; p.hi = a.hi; p.lo = a.lo; p.ldh = a.ld;
        r.rowop(p);
        GemmParams q = r.gp(a, n, g->inproj[l].w); q.C = y; q.ldc = D;
        r.gemm(q, EPI_STORE);
      }
      if (r.go()) r.chk(ovm_g_groupnorm(y, 1, n, D, 32, g->inproj[l].gn.g, g->inproj[l].gn.b, 1e-5f, vis0 + (size_t)pl->lstart[l] * D, s), "groupnorm");
    }
    r.release(mk);
  }

  // =============================== encoder ===============================
  r.join();                                               // text features and image features meet in the fusion layers
  const int HF = c.heads / 2, E = c.ffn_dim / 2, dhf = E / HF;
  {
    const size_t mk = r.mark();
    float* v = r.f32((size_t)S * D); SplitBuf vsp = r.split_for256((size_t)S, D);      // A of vqv / de_fc1: S x 2048 outputs over K = 256
    float* t = r.f32((size_t)T * D);
    float* qvv = r.f32((size_t)S * 2 * E);              // [vision_proj | values_vision_proj]
    float* tkv = r.f32((size_t)T * 2 * E);              // [text_proj | values_text_proj]
    SplitBuf cv = r.split((size_t)S, E);
    float* ct = r.f32((size_t)T * E);
    const int chunk = 128, nchunk = (S + chunk - 1) / chunk;
    float* sc = r.f32((size_t)HF * T * S);
    float* stat = r.f32((size_t)HF * T * 2);
    float* part = r.f32((size_t)nchunk * T * E);
    float* text2 = r.f32((size_t)T * D);
    float* tqk = r.f32((size_t)T * 2 * D); float* tv = r.f32((size_t)T * D); float* tctx = r.f32((size_t)T * D);
    float* tff = r.f32((size_t)T * c.ffn_dim);
    SplitBuf vis_sp = r.split((size_t)S, D), visp_sp = r.split((size_t)S, D);
    float* val = r.f32((size_t)S * D);
    const int NOW = c.heads * c.n_levels * c.n_points * 3;
    float* ow = r.f32((size_t)S * NOW);
    SplitBuf dsp = r.split((size_t)S, D);
    float* pre = r.f32((size_t)S * D);
    SplitBuf ff = r.split((size_t)S, c.ffn_dim);
    const float* vis_in = vis0; const float* text_in = text0;
    for (auto& ly : g->enc) {
      // ---- fusion layer (bi-directional image <-> text attention)
      r.ln(vis_in, S, D, ly.lnv, eps, nullptr, v, &vsp);
      r.ln(text_in, T, D, ly.lnt, eps, nullptr, t);
      vis_in = vis; text_in = text;
      { GemmParams q = r.gp(vsp, S, ly.vqv); q.C = qvv; q.ldc = 2 * E; r.gemm(q, EPI_STORE); }
      r.lin(t, nullptr, D, T, ly.tkv, 0, nullptr, 0, tkv, 2 * E);
      if (r.go()) {
        BiAttnParams b; memset(&b, 0, sizeof(b));
        b.qv = qvv; b.ldq = 2 * E; b.kt = tkv; b.ldk = 2 * E; b.vv = qvv + E; b.ldvv = 2 * E; b.vt = tkv + E; b.ldvt = 2 * E;
        b.S = S; b.T = T; b.H = HF; b.dh = dhf; b.scale = 1.0f / sqrtf((float)dhf);
        b.cv_hi = cv.hi; b.cv_lo = cv.lo; b.ldcv = cv.ld; b.ct = ct; b.sc = sc; b.stat = stat; b.part = part; b.chunk = chunk; b.nchunk = nchunk;
        r.chk(launch_biattn(b, s), "biattn"); r.launches += 3;
      }
      // the two halves of the layer from here on touch disjoint buffers: text side (ot, text enhancer -> text) on the text
      // branch, image side (ov, deformable self-attention, FFN -> vis) on the main one; joined at the end of the layer
      r.fork(); r.on_text();
      r.lin(ct, nullptr, E, T, ly.ot, 0, t, D, text2, D);
      // ---- text enhancer
      r.lin(text2, pl->text_pos, D, T, ly.te.qk, 0, nullptr, 0, tqk, 2 * D);
      r.lin(text2, nullptr, D, T, ly.te.v, 0, nullptr, 0, tv, D);
      mha_core(r, tqk, 2 * D, tqk + D, 2 * D, tv, D, T, T, ly.te.heads, D, pl->text_bias, T, tctx, D);
      r.lin(tctx, nullptr, D, T, ly.te.out, 0, text2, D, t, D);
      r.ln(t, T, D, ly.te_ln1, eps, nullptr, text2);
      r.lin(text2, nullptr, D, T, ly.te_fc1, 1, nullptr, 0, tff, ly.te_fc1.N);
      r.lin(tff, nullptr, ly.te_fc1.N, T, ly.te_fc2, 0, text2, D, t, D);
      r.ln(t, T, D, ly.te_ln2, eps, nullptr, text);
      r.on_image();
      { GemmParams q = r.gp(cv, S, ly.ov); q.C = vis; q.ldc = D; q.R = v; q.ldr = D; r.gemm(q, EPI_STORE); }
      // ---- deformable self-attention over the image tokens
      {
        RowOpParams p; memset(&p, 0, sizeof(p));
        p.x = vis; p.ldx = D; p.M = S; p.D = D; p.hi = vis_sp.hi; p.lo = vis_sp.lo; p.ldh = vis_sp.ld;
        p.add = pl->pos; p.ld_add = D; p.add_rows = S; p.hi2 = visp_sp.hi; p.lo2 = visp_sp.lo; p.ldh2 = visp_sp.ld;
        r.rowop(p);
      }
      { GemmParams q = r.gp(vis_sp, S, ly.msda.value); q.C = val; q.ldc = D; r.gemm(q, EPI_STORE); }
      { GemmParams q = r.gp(visp_sp, S, ly.msda.offw); q.C = ow; q.ldc = NOW; r.gemm(q, EPI_STORE); }
      deform(r, ly.msda, val, D, ow, NOW, pl->ref, 2, 0, S, nullptr, &dsp);
      { GemmParams q = r.gp(dsp, S, ly.msda.out); q.C = pre; q.ldc = D; q.R = vis; q.ldr = D; r.gemm(q, EPI_STORE); }
      r.ln(pre, S, D, ly.de_ln1, eps, nullptr, v, &vsp);
      { GemmParams q = r.gp(vsp, S, ly.de_fc1); q.Ohi = ff.hi; q.Olo = ff.lo; q.ldo = ff.ld; q.relu = 1; r.gemm(q, EPI_STORE); }
      { GemmParams q = r.gp(ff, S, ly.de_fc2); q.C = pre; q.ldc = D; q.R = v; q.ldr = D; r.gemm(q, EPI_STORE); }
      r.ln(pre, S, D, ly.de_ln2, eps, nullptr, vis);
      r.join();
    }
    r.release(mk);
  }
  r.tap("source_flatten", vis0, (int64_t)S * D);
  r.tap("enc_vision", vis, (int64_t)S * D);
  r.tap("enc_text", text, (int64_t)T * D);

  // =============================== two-stage query selection ===============================
  float* ref = r.f32((size_t)Q * 4);
  int* topk = r.i32((size_t)Q);
  {
    const size_t mk = r.mark();
    SplitBuf oqs = r.split((size_t)S, D);
    {
      RowOpParams p; memset(&p, 0, sizeof(p));
      p.x = vis; p.ldx = D; p.idx = pl->valid_idx; p.nidx = 1; p.seg = D; p.M = S; p.D = D; p.hi = oqs.hi; p.lo = oqs.lo; p.ldh = oqs.ld;
      r.rowop(p);                                                               // invalid proposals -> zero rows
    }
    float* oq0 = r.f32((size_t)S * D);
    { GemmParams q = r.gp(oqs, S, g->enc_output); q.C = oq0; q.ldc = D; r.gemm(q, EPI_STORE); }
    float* oq = r.f32((size_t)S * D);
    r.ln(oq0, S, D, g->enc_output_ln, eps, nullptr, oq, &oqs);
    float* cls = r.f32((size_t)S * T);
    if (r.go()) r.chk(ovm_g_bmm(oq, text, cls, 1, S, T, D, D, D, T, 0, 0, 0, 1, 1.0f, s), "bmm cls");
    float* mx = r.f32((size_t)S);
    if (r.go()) r.chk(ovm_g_rowmax(cls, S, T, T, mx, s), "rowmax");
    const int* sel = topk;
    if (g->force_topk) sel = g->force_topk;
    else if (r.go()) r.chk(launch_topk_keys(mx, S, Q, topk, pl->topk_keys, pl->topk_N, s), "topk");
    SplitBuf h1 = r.split((size_t)S, D), h2 = r.split((size_t)S, D);
    { GemmParams q = r.gp(oqs, S, g->enc_bbox[0]); q.Ohi = h1.hi; q.Olo = h1.lo; q.ldo = h1.ld; q.relu = 1; r.gemm(q, EPI_STORE); }
    { GemmParams q = r.gp(h1, S, g->enc_bbox[1]); q.Ohi = h2.hi; q.Olo = h2.lo; q.ldo = h2.ld; q.relu = 1; r.gemm(q, EPI_STORE); }
    float* coord = r.f32((size_t)S * 4);
    { GemmParams q = r.gp(h2, S, g->enc_bbox[2]); q.C = coord; q.ldc = 4; r.gemm(q, EPI_STORE); }
    if (r.go()) r.chk(launch_select_ref(coord, 4, pl->prop_logit, sel, Q, ref, s), "select_ref");
    if (r.go() && g->force_topk) GCHECK(g, hipMemcpyAsync(topk, g->force_topk, sizeof(int) * Q, hipMemcpyDeviceToDevice, s));
    r.release(mk);
  }
  r.tap("topk", topk, Q);
  r.tap("init_ref", ref, (int64_t)Q * 4);

  // =============================== decoder ===============================
  const int NL = (int)g->dec.size();
  float* hs = r.f32((size_t)Q * D);
  float* last_ref = r.f32((size_t)Q * 4);
  float* hn = r.f32((size_t)Q * D);
  {
    const size_t mk = r.mark();
    // projections that do not depend on the decoder state, all layers at once: text keys | values, deformable values of the memory
    float* tkv_all = r.f32((size_t)T * NL * 2 * D);
    r.lin(text, nullptr, D, T, g->dec_kv_text, 0, nullptr, 0, tkv_all, NL * 2 * D);
    SplitBuf vsp = r.split_for256((size_t)S, D);
    {
      RowOpParams p; memset(&p, 0, sizeof(p));
      p.x = vis; p.ldx = D; p.M = S; p.D = D; p.hi = vsp.hi; p.lo = vsp.lo; p.ldh = vsp.ld; p.il = vsp.il ? 1 : 0;
      r.rowop(p);
    }
    float* val_all = r.f32((size_t)S * NL * D);
    { GemmParams q = r.gp(vsp, S, g->dec_value); q.C = val_all; q.ldc = NL * D; r.gemm(q, EPI_STORE); }
    if (r.go()) GCHECK(g, hipMemcpyAsync(hs, g->tgt, sizeof(float) * (size_t)Q * D, hipMemcpyDeviceToDevice, s));
    float* sine = r.f32((size_t)Q * 2 * D);
    float* qh = r.f32((size_t)Q * D); float* qpos = r.f32((size_t)Q * D);
    float* qk = r.f32((size_t)Q * 2 * D); float* vq = r.f32((size_t)Q * D); float* ctx = r.f32((size_t)Q * D);
    float* pre = r.f32((size_t)Q * D);
    const int NOW = c.heads * c.n_levels * c.n_points * 3;
    float* ow = r.f32((size_t)Q * NOW);
    float* ffb = r.f32((size_t)Q * c.ffn_dim);
    float* b1 = r.f32((size_t)Q * D); float* b2 = r.f32((size_t)Q * D); float* delta = r.f32((size_t)Q * 4);
    float* refs[2] = {ref, r.f32((size_t)Q * 4)};
    int cur = 0;
    // Row-chain form of a layer (dec_chain.hip): everything but the query self-attention is local to a query row, so a workgroup
    // walks 16 rows through the whole layer in LDS - 3 launches per layer instead of ~35 (ovm_tune_set "gdino_dec_chain" 0: the
    // launch-per-op sequence below, kept as the cross-check).
    const bool chain = g_gdino_dec_chain && dec_chain_supported(D, c.heads, c.ffn_dim, c.n_levels, c.n_points, T, g->npass);
    auto cl = [](const Lin& w) { return ChainLin{w.frag, w.bias, w.N, w.K, w.Kpad}; };
    auto cn = [](const Ln& w) { return ChainLn{w.g, w.b}; };
    const int ffn_chunks = c.ffn_dim > 512 ? c.ffn_dim / 512 : 1;
    const bool ffn_split = chain && g_gdino_ffn_split && ffn_chunks > 1;
    float* ffn_x = ffn_split ? r.f32((size_t)Q * D) : nullptr;
    float* ffn_part = ffn_split ? r.f32((size_t)ffn_chunks * Q * D) : nullptr;
    for (int i = 0; i < NL && chain; ++i) {
      DecLayer& ly = g->dec[i];
      float* rf = refs[cur];
      DecChainParams dp; memset(&dp, 0, sizeof(dp));
      if (ffn_split) { dp.ffn_split = ffn_chunks; dp.ffn_x = ffn_x; dp.ffn_part = ffn_part; }
      dp.Q = Q; dp.D = D; dp.T = T; dp.heads = c.heads; dp.ffn = c.ffn_dim; dp.eps = eps;
      dp.sine_dim_t = g->sine_dim_t;
      dp.hs = hs; dp.ref = rf; dp.ref_next = (i + 1 < NL) ? refs[cur ^ 1] : nullptr;
      dp.qpos = qpos; dp.qk = qk; dp.v = vq; dp.ctx = ctx;
      dp.tk = tkv_all + (size_t)i * 2 * D; dp.tv = dp.tk + D; dp.ldt = NL * 2 * D;
      dp.val = val_all + (size_t)i * D; dp.ldv = NL * D;
      dp.L = c.n_levels; dp.P = c.n_points;
      for (int l = 0; l < c.n_levels; ++l) { dp.lh[l] = pl->lh[l]; dp.lw[l] = pl->lw[l]; dp.lstart[l] = pl->lstart[l]; }
      dp.ref0 = cl(g->ref_head[0]); dp.ref1 = cl(g->ref_head[1]); dp.sa_qk = cl(ly.sa.qk); dp.sa_v = cl(ly.sa.v); dp.sa_out = cl(ly.sa.out);
      dp.ca_q = cl(ly.ca.q); dp.ca_out = cl(ly.ca.out); dp.offw = cl(ly.msda.offw); dp.msda_out = cl(ly.msda.out);
      dp.fc1 = cl(ly.fc1); dp.fc2 = cl(ly.fc2);
      if (i + 1 < NL) { dp.bb0 = cl(g->bbox[i][0]); dp.bb1 = cl(g->bbox[i][1]); dp.bb2 = cl(g->bbox[i][2]); }
      dp.ln1 = cn(ly.ln1); dp.ln2 = cn(ly.ln2); dp.ln3 = cn(ly.ln3); dp.ln4 = cn(ly.ln4);
#ifdef OVM_DIAG
      if (const char* e = getenv("OVM_DEC_CHAIN_SKIP")) dp.dbg_skip = atoi(e);
      static unsigned long long* d_st = nullptr;
      if (getenv("OVM_DEC_CHAIN_STAMPS") && i == 0 && !dry) {
        if (!d_st) { (void)hipMalloc((void**)&d_st, 96 * 8); }
        (void)hipMemsetAsync(d_st, 0, 96 * 8, s);
        dp.dbg_stamps = d_st;
      }
#endif
      if (r.go()) r.chk(launch_dec_chain(dp, 0, s), "dec_chain_a");
#ifdef OVM_DIAG
      if (dp.dbg_stamps && r.go()) {
        unsigned long long hst[96];
        (void)hipStreamSynchronize(s);
        (void)hipMemcpy(hst, dp.dbg_stamps, sizeof(hst), hipMemcpyDeviceToHost);
        fprintf(stderr, "[dec_chain_a stamps, cycles since entry]");
        for (int k = 1; k < 32 && hst[k]; ++k) fprintf(stderr, " %llu", hst[k] - hst[0]);
        fprintf(stderr, "\n");
      }
#endif
      mha_core(r, qk, 2 * D, qk + D, 2 * D, vq, D, Q, Q, ly.sa.heads, D, nullptr, 0, ctx, D);
      if (r.go()) r.chk(launch_dec_chain(dp, 1, s), "dec_chain_b");
      if (ffn_split && r.go()) r.chk(launch_dec_chain(dp, 2, s), "dec_chain_c");
      if (i == NL - 1) { if (r.go()) GCHECK(g, hipMemcpyAsync(last_ref, rf, sizeof(float) * (size_t)Q * 4, hipMemcpyDeviceToDevice, s)); }
      r.tap(("dec_hs" + std::to_string(i)).c_str(), hs, (int64_t)Q * D);
      if (i + 1 < NL) cur ^= 1;
    }
    for (int i = 0; i < NL && !chain; ++i) {
      DecLayer& ly = g->dec[i];
      float* rf = refs[cur];
      if (r.go()) r.chk(ovm_g_sine_embed(rf, Q, 4, D / 2, 10000.0f, sine, s), "sine_embed");
      r.lin(sine, nullptr, 2 * D, Q, g->ref_head[0], 1, nullptr, 0, qh, D);
      r.lin(qh, nullptr, D, Q, g->ref_head[1], 0, nullptr, 0, qpos, D);
      // self-attention
      r.lin(hs, qpos, D, Q, ly.sa.qk, 0, nullptr, 0, qk, 2 * D);
      r.lin(hs, nullptr, D, Q, ly.sa.v, 0, nullptr, 0, vq, D);
      mha_core(r, qk, 2 * D, qk + D, 2 * D, vq, D, Q, Q, ly.sa.heads, D, nullptr, 0, ctx, D);
      r.lin(ctx, nullptr, D, Q, ly.sa.out, 0, hs, D, pre, D);
      r.ln(pre, Q, D, ly.ln1, eps, nullptr, hs);
      // text cross-attention
      r.lin(hs, qpos, D, Q, ly.ca.q, 0, nullptr, 0, qk, D);
      mha_core(r, qk, D, tkv_all + (size_t)i * 2 * D, NL * 2 * D, tkv_all + (size_t)i * 2 * D + D, NL * 2 * D, Q, T, ly.ca.heads, D, nullptr, 0, ctx, D);
      r.lin(ctx, nullptr, D, Q, ly.ca.out, 0, hs, D, pre, D);
      r.ln(pre, Q, D, ly.ln2, eps, nullptr, hs);
      // deformable cross-attention on the encoder memory
      r.lin(hs, qpos, D, Q, ly.msda.offw, 0, nullptr, 0, ow, NOW);
      deform(r, ly.msda, val_all + (size_t)i * D, NL * D, ow, NOW, rf, 4, 1, Q, ctx, nullptr);
      r.lin(ctx, nullptr, D, Q, ly.msda.out, 0, hs, D, pre, D);
      r.ln(pre, Q, D, ly.ln3, eps, nullptr, hs);
      // FFN
      r.lin(hs, nullptr, D, Q, ly.fc1, 1, nullptr, 0, ffb, c.ffn_dim);
      r.lin(ffb, nullptr, c.ffn_dim, Q, ly.fc2, 0, hs, D, pre, D);
      r.ln(pre, Q, D, ly.ln4, eps, nullptr, hs);
      if (i == NL - 1) { if (r.go()) GCHECK(g, hipMemcpyAsync(last_ref, rf, sizeof(float) * (size_t)Q * 4, hipMemcpyDeviceToDevice, s)); }
      r.tap(("dec_hs" + std::to_string(i)).c_str(), hs, (int64_t)Q * D);
      // iterative box refinement (the update after the last layer is unused)
      if (i + 1 < NL) {
        r.lin(hs, nullptr, D, Q, g->bbox[i][0], 1, nullptr, 0, b1, D);
        r.lin(b1, nullptr, D, Q, g->bbox[i][1], 1, nullptr, 0, b2, D);
        r.lin(b2, nullptr, D, Q, g->bbox[i][2], 0, nullptr, 0, delta, 4);
        if (r.go()) r.chk(launch_box_refine(delta, 4, rf, 1e-5f, refs[cur ^ 1], Q, s), "box_refine");
        cur ^= 1;
      }
    }
    // ---- heads on the normalised last hidden state
    r.ln(hs, Q, D, g->dec_ln, eps, nullptr, hn);
    float* lt = r.f32((size_t)Q * T);
    if (r.go()) {
      r.chk(ovm_g_bmm(hn, text, lt, 1, Q, T, D, D, D, T, 0, 0, 0, 1, 1.0f, s), "bmm logits");
      r.chk(launch_pad_logits(lt, T, Q, T, pl->out_logits, c.max_text_len, s), "pad_logits");
    }
    r.lin(hn, nullptr, D, Q, g->bbox[NL - 1][0], 1, nullptr, 0, b1, D);
    r.lin(b1, nullptr, D, Q, g->bbox[NL - 1][1], 1, nullptr, 0, b2, D);
    r.lin(b2, nullptr, D, Q, g->bbox[NL - 1][2], 0, nullptr, 0, delta, 4);
    if (r.go()) r.chk(launch_box_refine(delta, 4, last_ref, 1e-5f, pl->out_boxes, Q, s), "box_refine");
    r.release(mk);
  }
  return r.rc;
}

int plan_matches(const Plan* p, int H, int W, const std::vector<int>& ids, const std::vector<int>& pids) {
  return p->H == H && p->W == W && p->ids == ids && p->pids == pids;
}

}  // namespace

extern "C" {

int ovm_gdino_create(const OvmGdinoConfig* cfg, const OvmTensor* weights, int32_t n_weights, int32_t device, OvmGdino** out) {
  if (!cfg || !weights || !out) return OVM_ERR_INVALID;
  OvmGdino* g = new OvmGdino();
  *out = g;                                      // returned even on failure so that ovm_gdino_last_error can be read; destroy it
  g->cfg = *cfg; g->device = device; g->npass = cfg->precision == 1 ? 1 : 3;
  g->graphs_enabled = cfg->use_graphs;
  g->branches = g_gdino_branches;
  {
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);         // hi = numerically lowest = highest priority
    if (hipStreamCreateWithPriority(&g->aux, hipStreamNonBlocking, hi) != hipSuccess || hipEventCreateWithFlags(&g->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&g->ev_join, hipEventDisableTiming) != hipSuccess) {
      g->err = "could not create the text-branch stream"; return OVM_ERR_HIP;
    }
  }
  const OvmGdinoConfig& c = g->cfg;
  if (c.d_model % c.heads || c.n_levels > 8 || c.n_levels < 1 || c.swin_window <= 0) { g->err = "bad GroundingDINO config"; return OVM_ERR_INVALID; }
  GCHECK(g, hipSetDevice(device));
  WMap wm;
  for (int i = 0; i < n_weights; ++i) wm[weights[i].name] = &weights[i];
  const std::string M = "model.";
  // ---- BERT
  {
    const std::string e = M + "text_backbone.embeddings.";
    const OvmTensor* t; RCHECK(g, get(g, wm, e + "word_embeddings.weight", &t));
    g->vocab = (int)t->shape[0]; g->bertD = (int)t->shape[1];
    RCHECK(g, up_f32(g, wm, e + "word_embeddings.weight", &g->word));
    RCHECK(g, get(g, wm, e + "position_embeddings.weight", &t)); g->n_pos = (int)t->shape[0];
    RCHECK(g, up_f32(g, wm, e + "position_embeddings.weight", &g->posemb));
    RCHECK(g, up_f32(g, wm, e + "token_type_embeddings.weight", &g->typemb));
    RCHECK(g, up_ln(g, wm, e + "LayerNorm", &g->emb_ln));
    for (int i = 0;; ++i) {
      const std::string p = M + "text_backbone.encoder.layer." + std::to_string(i) + ".";
      if (!wm.count(p + "attention.self.query.weight")) break;
      g->bert.emplace_back();
      BertLayer& ly = g->bert.back();
      RCHECK(g, pack_cat(g, wm, {p + "attention.self.query", p + "attention.self.key", p + "attention.self.value"}, &ly.qkv));
      RCHECK(g, pack_lin(g, wm, p + "attention.output.dense", &ly.ao));
      RCHECK(g, up_ln(g, wm, p + "attention.output.LayerNorm", &ly.aln));
      RCHECK(g, pack_lin(g, wm, p + "intermediate.dense", &ly.fi));
      RCHECK(g, pack_lin(g, wm, p + "output.dense", &ly.fo));
      RCHECK(g, up_ln(g, wm, p + "output.LayerNorm", &ly.oln));
    }
    if (g->bertD % c.bert_heads) { g->err = "bert heads"; return OVM_ERR_INVALID; }
    RCHECK(g, pack_lin(g, wm, M + "text_projection", &g->text_proj));
  }
  // ---- Swin
  {
    const std::string bb = M + "backbone.conv_encoder.model.", p = bb + "swin.";
    RCHECK(g, pack_conv(g, wm, p + "embeddings.patch_embeddings.projection", &g->pe, nullptr));
    RCHECK(g, up_ln(g, wm, p + "embeddings.norm", &g->pe_ln));
    const int ws = c.swin_window, ws2 = ws * ws;
    std::vector<int> rel_index((size_t)ws2 * ws2);
    for (int a = 0; a < ws2; ++a)
      for (int b = 0; b < ws2; ++b) {
        const int dy = a / ws - b / ws + ws - 1, dx = a % ws - b % ws + ws - 1;
        rel_index[(size_t)a * ws2 + b] = dy * (2 * ws - 1) + dx;
      }
    int C = c.swin_embed;
    for (int s = 0; s < 4; ++s) {
      if (c.swin_depths[s] <= 0) break;
      g->stages.emplace_back();
      SwinStage& st = g->stages.back();
      st.nh = c.swin_heads[s]; st.C = C;
      if (C % st.nh || (C / st.nh != 16 && C / st.nh != 32 && C / st.nh != 64)) { g->err = "Swin head dim must be 16, 32 or 64"; return OVM_ERR_SHAPE; }
      for (int b = 0; b < c.swin_depths[s]; ++b) {
        const std::string q = p + "encoder.layers." + std::to_string(s) + ".blocks." + std::to_string(b) + ".";
        st.blocks.emplace_back();
        SwinBlock& blk = st.blocks.back();
        RCHECK(g, up_ln(g, wm, q + "layernorm_before", &blk.ln1));
        RCHECK(g, up_ln(g, wm, q + "layernorm_after", &blk.ln2));
        RCHECK(g, pack_cat(g, wm, {q + "attention.q_proj", q + "attention.k_proj", q + "attention.v_proj"}, &blk.qkv));
        if (swin_qkv_attn_supported(C, st.nh, ws, g->npass)) RCHECK(g, make_frag(g, &blk.qkv));      // the window kernel projects q | k | v itself
        RCHECK(g, pack_lin(g, wm, q + "attention.o_proj", &blk.proj));
        RCHECK(g, pack_lin(g, wm, q + "mlp.fc1", &blk.fc1));
        RCHECK(g, pack_lin(g, wm, q + "mlp.fc2", &blk.fc2));
        const OvmTensor* tb; RCHECK(g, get(g, wm, q + "attention.relative_position_bias.relative_position_bias_table", &tb));
        if (tb->shape[0] != (2 * ws - 1) * (2 * ws - 1) || tb->shape[1] != st.nh) { g->err = "relative position bias table shape"; return OVM_ERR_SHAPE; }
        std::vector<float> rb((size_t)st.nh * ws2 * ws2);
        for (int hh = 0; hh < st.nh; ++hh)
          for (size_t i = 0; i < (size_t)ws2 * ws2; ++i) rb[(size_t)hh * ws2 * ws2 + i] = tb->data[(size_t)rel_index[i] * st.nh + hh];
        RCHECK(g, up_vec(g, rb, &blk.relbias));
      }
      const std::string dk = p + "encoder.layers." + std::to_string(s) + ".downsample.";
      if (wm.count(dk + "reduction.weight")) {
        st.has_red = true;
        RCHECK(g, pack_lin(g, wm, dk + "reduction", &st.red, false));
        RCHECK(g, up_ln(g, wm, dk + "norm", &st.dn));
      }
      const std::string nk = bb + "hidden_states_norms.stage" + std::to_string(s + 1);
      if (wm.count(nk + ".weight")) { st.has_out = true; RCHECK(g, up_ln(g, wm, nk, &st.on)); }
      if (st.has_red) C *= 2;
    }
  }
  // ---- neck
  for (int l = 0; l < c.n_levels; ++l) {
    const std::string p = M + "input_proj_vision." + std::to_string(l);
    RCHECK(g, pack_conv(g, wm, p + ".0", &g->inproj[l].w, &g->inproj[l].k));
    RCHECK(g, up_ln(g, wm, p + ".1", &g->inproj[l].gn));
  }
  {
    const OvmTensor* t; RCHECK(g, get(g, wm, M + "level_embed", &t));
    g->level_embed.assign(t->data, t->data + numel(t));
  }
  const int D = c.d_model;
  // ---- encoder
  for (int i = 0; i < c.enc_layers; ++i) {
    const std::string p = M + "encoder.layers." + std::to_string(i) + ".";
    const std::string fu = p + "fusion_layer.", te = p + "text_enhancer_layer.", de = p + "deformable_layer.";
    g->enc.emplace_back();
    EncLayer& ly = g->enc.back();
    RCHECK(g, up_ln(g, wm, fu + "layer_norm_vision", &ly.lnv));
    RCHECK(g, up_ln(g, wm, fu + "layer_norm_text", &ly.lnt));
    RCHECK(g, pack_cat(g, wm, {fu + "attn.vision_proj", fu + "attn.values_vision_proj"}, &ly.vqv));
    RCHECK(g, pack_cat(g, wm, {fu + "attn.text_proj", fu + "attn.values_text_proj"}, &ly.tkv));
    const OvmTensor *gv, *gt; RCHECK(g, get(g, wm, fu + "vision_param", &gv)); RCHECK(g, get(g, wm, fu + "text_param", &gt));
    RCHECK(g, pack_cat(g, wm, {fu + "attn.out_vision_proj"}, &ly.ov, true, gv->data));     // layer scale folded into the projection
    RCHECK(g, pack_cat(g, wm, {fu + "attn.out_text_proj"}, &ly.ot, true, gt->data));
    RCHECK(g, load_mha(g, wm, te + "self_attn.", c.heads / 2, &ly.te, false));
    RCHECK(g, up_ln(g, wm, te + "layer_norm_before", &ly.te_ln1));
    RCHECK(g, up_ln(g, wm, te + "layer_norm_after", &ly.te_ln2));
    RCHECK(g, pack_lin(g, wm, te + "fc1", &ly.te_fc1));
    RCHECK(g, pack_lin(g, wm, te + "fc2", &ly.te_fc2));
    RCHECK(g, load_msda(g, wm, de + "self_attn.", &ly.msda, true));
    RCHECK(g, up_ln(g, wm, de + "self_attn_layer_norm", &ly.de_ln1));
    RCHECK(g, up_ln(g, wm, de + "final_layer_norm", &ly.de_ln2));
    RCHECK(g, pack_lin(g, wm, de + "fc1", &ly.de_fc1));
    RCHECK(g, pack_lin(g, wm, de + "fc2", &ly.de_fc2));
  }
  RCHECK(g, pack_lin(g, wm, M + "enc_output", &g->enc_output));
  RCHECK(g, up_ln(g, wm, M + "enc_output_norm", &g->enc_output_ln));
  for (int k = 0; k < 3; ++k) RCHECK(g, pack_lin(g, wm, M + "encoder_output_bbox_embed.layers." + std::to_string(k), &g->enc_bbox[k]));
  RCHECK(g, up_f32(g, wm, M + "query_position_embeddings.weight", &g->tgt, (int64_t)c.num_queries * D));
  // ---- decoder
  {
    std::vector<std::string> kvnames, valnames;
    for (int i = 0; i < c.dec_layers; ++i) {
      const std::string p = M + "decoder.layers." + std::to_string(i) + ".";
      g->dec.emplace_back();
      DecLayer& ly = g->dec.back();
      RCHECK(g, load_mha(g, wm, p + "self_attn.", c.heads, &ly.sa, false));
      RCHECK(g, up_ln(g, wm, p + "self_attn_layer_norm", &ly.ln1));
      RCHECK(g, load_mha(g, wm, p + "encoder_attn_text.", c.heads, &ly.ca, true));
      RCHECK(g, up_ln(g, wm, p + "encoder_attn_text_layer_norm", &ly.ln2));
      RCHECK(g, load_msda(g, wm, p + "encoder_attn.", &ly.msda, false));
      RCHECK(g, up_ln(g, wm, p + "encoder_attn_layer_norm", &ly.ln3));
      RCHECK(g, pack_lin(g, wm, p + "fc1", &ly.fc1));
      RCHECK(g, pack_lin(g, wm, p + "fc2", &ly.fc2));
      RCHECK(g, up_ln(g, wm, p + "final_layer_norm", &ly.ln4));
      kvnames.push_back(p + "encoder_attn_text.key"); kvnames.push_back(p + "encoder_attn_text.value");
      valnames.push_back(p + "encoder_attn.value_proj");
    }
    RCHECK(g, pack_cat(g, wm, kvnames, &g->dec_kv_text));
    RCHECK(g, pack_cat(g, wm, valnames, &g->dec_value));
  }
  RCHECK(g, up_ln(g, wm, M + "decoder.layer_norm", &g->dec_ln));
  for (int k = 0; k < 2; ++k) RCHECK(g, pack_lin(g, wm, M + "decoder.reference_points_head.layers." + std::to_string(k), &g->ref_head[k]));
  g->bbox.resize(c.dec_layers);
  for (int i = 0; i < c.dec_layers; ++i)
    for (int k = 0; k < 3; ++k) RCHECK(g, pack_lin(g, wm, "bbox_embed." + std::to_string(i) + ".layers." + std::to_string(k), &g->bbox[i][k]));
  {
    const int F = D / 2;
    std::vector<float> dt((size_t)F / 2);
    for (int i = 0; i < F / 2; ++i) dt[i] = powf(10000.0f, 2.f * (float)i / (float)F);       // sine_embed_kernel's dim_t for f / 2 = i
    RCHECK(g, up_vec(g, dt, &g->sine_dim_t));
  }
  // fragment-ordered copies of everything the decoder's row-chain kernels multiply by
  for (int k = 0; k < 2; ++k) RCHECK(g, make_frag(g, &g->ref_head[k]));
  for (int i = 0; i < c.dec_layers; ++i) {
    DecLayer& ly = g->dec[i];
    for (Lin* w : {&ly.sa.qk, &ly.sa.v, &ly.sa.out, &ly.ca.q, &ly.ca.out, &ly.msda.offw, &ly.msda.out, &ly.fc1, &ly.fc2}) RCHECK(g, make_frag(g, w));
    if (i + 1 < c.dec_layers) for (int k = 0; k < 3; ++k) RCHECK(g, make_frag(g, &g->bbox[i][k]));
  }
  GCHECK(g, hipDeviceSynchronize());
  return OVM_OK;
}

int ovm_gdino_destroy(OvmGdino* g) {
  if (!g) return OVM_OK;
  (void)hipSetDevice(g->device);
  (void)hipDeviceSynchronize();
  for (Plan* p : g->plans) delete p;
  for (void* p : g->allocs) (void)hipFree(p);
  if (g->aux) (void)hipStreamDestroy(g->aux);
  if (g->ev_fork) (void)hipEventDestroy(g->ev_fork);
  if (g->ev_join) (void)hipEventDestroy(g->ev_join);
  delete g;
  return OVM_OK;
}

const char* ovm_gdino_last_error(const OvmGdino* g) { return g ? g->err.c_str() : "null handle"; }

int ovm_gdino_set_force_topk(OvmGdino* g, const int32_t* idx_device) {
  if (!g) return OVM_ERR_INVALID;
  g->force_topk = idx_device;
  for (Plan* p : g->plans) {                      // captured graphs bake the selection source in
    if (p->exec) { (void)hipGraphExecDestroy(p->exec); p->exec = nullptr; }
    if (p->graph) { (void)hipGraphDestroy(p->graph); p->graph = nullptr; }
  }
  return OVM_OK;
}

// The network: image (uint8, any C/H/W strides; normalised here with the reference's images[0][[2,1,0]] convention when
// flip_channels is set, roi_heads_gdino.py:146) + caption token ids -> pred_logits [num_queries][max_text_len] (pre-sigmoid,
// -inf beyond the caption) and pred_boxes [num_queries][4] (cx, cy, w, h in [0, 1]). position_ids: null = upstream numbering.
int ovm_gdino_forward(OvmGdino* g, const OvmImage* image, const int32_t* token_ids, int32_t ntok, const int32_t* position_ids,
                      float* pred_logits, float* pred_boxes, ovm_stream_t stream) {
  if (!g || !image || !token_ids || ntok <= 0 || ntok > g->cfg.max_text_len) return OVM_ERR_INVALID;
  g->err.clear();
  hipStream_t s = (hipStream_t)stream;
  const int H = image->height, W = image->width;
  if (H <= 0 || W <= 0) return OVM_ERR_INVALID;
  std::vector<int> ids(token_ids, token_ids + ntok), pids;
  if (position_ids) pids.assign(position_ids, position_ids + ntok);
  Plan* pl = nullptr;
  for (auto it = g->plans.begin(); it != g->plans.end(); ++it)
    if (plan_matches(*it, H, W, ids, pids)) { pl = *it; g->plans.erase(it); break; }
  if (!pl) {
    RCHECK(g, build_plan(g, H, W, ids, pids, &pl));
    Run dry{g, pl, s, true};
    int r = forward_impl(dry);
    if (r) { delete pl; return r; }
    pl->arena_cap = dry.peak + 4096;
    void* q = nullptr;
    if (hipMalloc(&q, pl->arena_cap) != hipSuccess) { delete pl; g->err = "arena allocation failed"; return OVM_ERR_HIP; }
    pl->allocs.push_back(q); pl->arena = (char*)q; pl->bytes += pl->arena_cap;
    // Least recently used plans go when the count or - what matters on a dataset with many aspect ratios - the bytes they hold
    // together exceed the configured bounds (defaults: 128 plans, 32 GiB of the 288 GB).
    const int maxp = g->cfg.max_plans > 0 ? g->cfg.max_plans : 128;
    const size_t budget = (size_t)(g->cfg.plan_budget_mb > 0 ? g->cfg.plan_budget_mb : 32768) << 20;
    size_t held = pl->bytes;
    for (Plan* q2 : g->plans) held += q2->bytes;
    while (!g->plans.empty() && ((int)g->plans.size() >= maxp || held > budget)) {
      (void)hipDeviceSynchronize();
      held -= g->plans.back()->bytes;
      delete g->plans.back(); g->plans.pop_back();
    }
  }
  g->plans.push_front(pl);
  g->last = pl;
  // input normalisation reads the caller's buffer: outside the graph
  RCHECK(g, ovm_g_normalize_image(image, g->cfg.pixel_mean, g->cfg.pixel_std, g->cfg.flip_channels, pl->img, s));
  if (pl->exec) {
    GCHECK(g, hipGraphLaunch(pl->exec, s));
  } else {
    Run run{g, pl, s, false};
    RCHECK(g, forward_impl(run));
    pl->launches = run.launches;
    if (g->graphs_enabled) {
      // captured right after the first (eager) run of a plan, so every later call of this shape replays; a failed capture
      // leaves the eager path in place (same results)
      hipGraph_t graph = nullptr;
      if (hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) == hipSuccess) {
        Run cap{g, pl, s, false};
        const int r = forward_impl(cap);
        const hipError_t e = hipStreamEndCapture(s, &graph);
        if (r == OVM_OK && e == hipSuccess && graph) {
          hipGraphExec_t ex = nullptr;
          if (hipGraphInstantiate(&ex, graph, nullptr, nullptr, 0) == hipSuccess) { pl->graph = graph; pl->exec = ex; }
          else (void)hipGraphDestroy(graph);
        } else if (graph) {
          (void)hipGraphDestroy(graph);
        }
        g->err.clear();
      }
      (void)hipGetLastError();
    }
  }
  g->launches_last = pl->launches;
  const size_t nl = (size_t)g->cfg.num_queries * g->cfg.max_text_len;
  if (pred_logits) GCHECK(g, hipMemcpyAsync(pred_logits, pl->out_logits, nl * sizeof(float), hipMemcpyDeviceToDevice, s));
  if (pred_boxes) GCHECK(g, hipMemcpyAsync(pred_boxes, pl->out_boxes, (size_t)g->cfg.num_queries * 4 * sizeof(float), hipMemcpyDeviceToDevice, s));
  return OVM_OK;
}

// network + the reference-owned output glue (ovm_gdino_postprocess): boxes / scores / class indices of the kept queries
int ovm_gdino_detect(OvmGdino* g, const OvmImage* image, const int32_t* token_ids, int32_t ntok, const int32_t* spans, int32_t n_phrases,
                     float box_threshold, float nms_threshold, float* out_boxes, float* out_scores, int32_t* out_classes, int32_t* n_out,
                     ovm_stream_t stream) {
  int r = ovm_gdino_forward(g, image, token_ids, ntok, nullptr, nullptr, nullptr, stream);
  if (r) return r;
  Plan* pl = g->last;
  return ovm_gdino_postprocess(pl->out_logits, g->cfg.num_queries, g->cfg.max_text_len, pl->out_boxes, spans, n_phrases, image->height,
                               image->width, box_threshold, nms_threshold, out_boxes, out_scores, out_classes, n_out, stream);
}

int32_t ovm_gdino_num_queries(const OvmGdino* g) { return g ? g->cfg.num_queries : 0; }

// device pointers of the last forward's raw outputs (owned by the handle's current plan; valid until the next forward)
int ovm_gdino_last_outputs(OvmGdino* g, const float** pred_logits, const float** pred_boxes, int32_t* logits_ld) {
  if (!g || !g->last) return OVM_ERR_INVALID;
  if (pred_logits) *pred_logits = g->last->out_logits;
  if (pred_boxes) *pred_boxes = g->last->out_boxes;
  if (logits_ld) *logits_ld = g->cfg.max_text_len;
  return OVM_OK;
}

int64_t ovm_gdino_debug_copy(OvmGdino* g, const char* name, void* dst, int64_t capacity_elems, ovm_stream_t stream) {
  if (!g || !g->last || !name) return OVM_ERR_INVALID;
  if (std::string(name) == "launches") return g->launches_last;
  if (std::string(name) == "plans") return (int64_t)g->plans.size();                      // plan-cache occupancy (tests, bench)
  if (std::string(name) == "plan_bytes") { int64_t b = 0; for (Plan* q : g->plans) b += (int64_t)q->bytes; return b; }
  auto it = g->last->taps.find(name);
  if (it == g->last->taps.end()) { g->err = std::string("unknown debug tensor ") + name; return OVM_ERR_INVALID; }
  if (it->second.second > capacity_elems) return OVM_ERR_CAPACITY;
  if (hipMemcpyAsync(dst, it->second.first, (size_t)it->second.second * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess) return OVM_ERR_HIP;
  return it->second.second;
}

}  // extern "C"
