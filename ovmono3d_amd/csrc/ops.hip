// Kernel-level C entry points (parity tests / micro-benchmarks): thin adapters that convert fp32 test
// tensors to the split-fp16 layouts and launch the same kernels the model path uses.
#include <hip/hip_runtime.h>
#include <cstring>
#include <vector>
#include "../../include/ovm3d.h"
#include "kernels.hpp"
#include "det2d.hpp"

using namespace ovm;

namespace {
__global__ void split_kernel(const float* __restrict__ x, int64_t n, half_t* __restrict__ hi, half_t* __restrict__ lo) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  half_t h, l; split_f16(x[i], h, l);
  hi[i] = h;
  if (lo) lo[i] = l;
}
// qkv [B*T][3D] fp32 (as nn.Linear emits it) -> Q (pre-scaled), K, V^T split layouts of the attention kernel
__global__ void qkv_layout_kernel(const float* __restrict__ qkv, int B, int T, int Tpad, int heads,
                                  half_t* Qh, half_t* Ql, half_t* Kh, half_t* Kl, half_t* Vh, half_t* Vl) {
  const int D = heads * 64;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)B * T * 3 * D;
  if (i >= total) return;
  const int n = (int)(i % (3 * D)); const int64_t m = i / (3 * D);
  const int b = (int)(m / T), t = (int)(m - (int64_t)b * T);
  const int which = n / D, f = n - which * D, head = f >> 6, d = f & 63;
  float v = qkv[i]; if (which == 0) v *= kQScale;
  half_t h, l; split_f16(v, h, l);
  if (which < 2) {
    const size_t o = ((size_t)(b * heads + head) * T + t) * 64 + d;
    if (which == 0) { Qh[o] = h; if (Ql) Ql[o] = l; } else { Kh[o] = h; if (Kl) Kl[o] = l; }
  } else {
    const int tp = (t & ~15) | (t & 3) | ((t & 4) << 1) | ((t & 8) >> 1);
    const size_t o = ((size_t)(b * heads + head) * 64 + d) * Tpad + tp;
    Vh[o] = h; if (Vl) Vl[o] = l;
  }
}
__global__ void join_kernel(const half_t* __restrict__ hi, const half_t* __restrict__ lo, int64_t n, float* __restrict__ y) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  y[i] = (float)hi[i] + (lo ? (float)lo[i] * kLoInv : 0.f);
}
struct Tmp {
  std::vector<void*> p;
  template <typename Tp> Tp* get(size_t n, bool zero = false) {
    void* q = nullptr;
    if (hipMalloc(&q, n * sizeof(Tp) + 16) != hipSuccess) return nullptr;
    if (zero) hipMemset(q, 0, n * sizeof(Tp) + 16);
    p.push_back(q);
    return (Tp*)q;
  }
  ~Tmp() { for (void* q : p) hipFree(q); }
};
}  // namespace

namespace ovm { void set_use_gemm256(int v); void set_gdino_branches(int v); void msdeform_set_vec(int v); void set_gdino_dec_chain(int v); void set_gdino_gemm256(int v); void set_gdino_swin_fused(int v); void set_gdino_ffn_split(int v); void gemm256_set_n192(int v); void set_gemm256_ksplit(int v); }

extern "C" {

int ovm_op_split_f16(const float* x, int64_t n, uint16_t* hi, uint16_t* lo, ovm_stream_t stream) {
  if (n <= 0) return OVM_OK;
  hipLaunchKernelGGL(split_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, n, (half_t*)hi, (half_t*)lo);
  return hipGetLastError() == hipSuccess ? OVM_OK : OVM_ERR_HIP;
}

// C[M][N] = A[M][K] W[N][K]^T (+bias)(+relu). W must be padded to a multiple of 128 rows by the caller.
static int g_op_gemm256 = 0;
static unsigned long long* g_gemm256_stamps = nullptr;   // diagnostic: device buffer [8][128] set through ovm_debug_set_ptr

int ovm_op_gemm(const uint16_t* a_hi, const uint16_t* a_lo, int32_t lda, const uint16_t* w_hi, const uint16_t* w_lo,
                int32_t M, int32_t N, int32_t K, const float* bias, int32_t relu, float* c, int32_t ldc,
                int32_t precision, ovm_stream_t stream) {
  GemmParams p; memset(&p, 0, sizeof(p));
  p.Ahi = (const half_t*)a_hi; p.Alo = (const half_t*)a_lo; p.lda = lda;
  p.Whi = (const half_t*)w_hi; p.Wlo = (const half_t*)w_lo;
  if (precision == 3) {
    if (w_lo != w_hi + 32) return OVM_ERR_INVALID;            // split weights are an interleaved image (ovm_op_interleave)
    p.a_il = (a_lo == a_hi + 32);                             // activations: either layout
  }
  p.M = M; p.N = N; p.K = K; p.bias = bias; p.relu = relu; p.C = c; p.ldc = ldc;
  if (g_op_gemm256 > 0) {                                     // tests / micro-benchmarks: force the 256 x 256 kernel (value = split-K hint)
    if (!gemm256_supported(p, precision)) return OVM_ERR_INVALID;
    p.stamps = g_gemm256_stamps;
    return launch_gemm256(p, EPI_STORE, g_op_gemm256, (hipStream_t)stream);
  }
  return launch_gemm(p, precision, EPI_STORE, A_ROWMAJOR, (hipStream_t)stream);
}

namespace {
__global__ void interleave_kernel(const half_t* __restrict__ hi, const half_t* __restrict__ lo, long rows, int K, half_t* __restrict__ out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * K) return;
  const long r = i / K; const int k = (int)(i - r * K);
  const size_t o = (size_t)r * 2 * K + (size_t)(k >> 5) * 64 + (k & 31);
  out[o] = hi[i];
  out[o + 32] = lo[i];
}
}  // namespace

// hi, lo [rows][K] (K % 32 == 0) -> out [rows][K/32][hi 32 | lo 32], the operand image of the split-mode GEMM
int ovm_op_interleave(const uint16_t* hi, const uint16_t* lo, int64_t rows, int32_t K, uint16_t* out, ovm_stream_t stream) {
  if (rows <= 0) return OVM_OK;
  if (!hi || !lo || !out || K % 32 != 0) return OVM_ERR_INVALID;
  const long n = (long)rows * K;
  hipLaunchKernelGGL(interleave_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const half_t*)hi, (const half_t*)lo,
                     (long)rows, K, (half_t*)out);
  return hipGetLastError() == hipSuccess ? OVM_OK : OVM_ERR_HIP;
}

int ovm_op_layernorm(const float* x, int32_t M, int32_t D, const float* gamma, const float* beta, float eps, float* y,
                     ovm_stream_t stream) {
  LnOut o; memset(&o, 0, sizeof(o)); o.f32 = y; o.ldf = D;
  return launch_ln_rows(x, D, M, D, gamma, beta, eps, o, (hipStream_t)stream);
}

// qkv [B*T][3*heads*64] fp32 -> out [B*T][heads*64] fp32 (synchronous helper: allocates scratch)
int ovm_op_attention(const float* qkv, int32_t B, int32_t T, int32_t heads, float* out, int32_t precision, ovm_stream_t stream) {
  hipStream_t s = (hipStream_t)stream;
  const int D = heads * 64, Tpad = (T + 63) / 64 * 64;
  const size_t nqk = (size_t)B * T * D, nv = (size_t)B * D * Tpad;
  Tmp tmp;
  half_t *Qh = tmp.get<half_t>(nqk), *Kh = tmp.get<half_t>(nqk), *Vh = tmp.get<half_t>(nv, true), *Oh = tmp.get<half_t>(nqk);
  half_t *Ql = nullptr, *Kl = nullptr, *Vl = nullptr, *Ol = nullptr;
  if (precision == 3) { Ql = tmp.get<half_t>(nqk); Kl = tmp.get<half_t>(nqk); Vl = tmp.get<half_t>(nv, true); Ol = tmp.get<half_t>(nqk); }
  if (!Qh || !Kh || !Vh || !Oh) return OVM_ERR_HIP;
  const int64_t total = (int64_t)B * T * 3 * D;
  hipLaunchKernelGGL(qkv_layout_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, qkv, B, T, Tpad, heads, Qh, Ql, Kh, Kl, Vh, Vl);
  AttnParams a; memset(&a, 0, sizeof(a));
  a.Qhi = Qh; a.Qlo = Ql; a.Khi = Kh; a.Klo = Kl; a.Vhi = Vh; a.Vlo = Vl; a.Ohi = Oh; a.Olo = Ol; a.ldo = D;
  a.B = B; a.heads = heads; a.T = T; a.Tpad = Tpad;
  a.tail_ws = tmp.get<float>(attn_tail_ws_floats(B, heads)); a.tail_cnt = tmp.get<int>((size_t)B * heads * 8, true);
  if (!a.tail_ws || !a.tail_cnt) return OVM_ERR_HIP;
  int r = launch_attention(a, precision, s);
  if (r) return r;
  hipLaunchKernelGGL(join_kernel, dim3((unsigned)((nqk + 255) / 256)), dim3(256), 0, s, Oh, Ol, (int64_t)nqk, out);
  if (hipStreamSynchronize(s) != hipSuccess) return OVM_ERR_HIP;
  return OVM_OK;
}

int ovm_op_roi_align(const float* p2, const float* p3, const float* p4, const int32_t* hw, const float* scales, int32_t C,
                     int32_t out_res, int32_t min_level, int32_t max_level, const float* boxes, const int32_t* image_idx,
                     int32_t n, float* out, ovm_stream_t stream) {
  hipStream_t s = (hipStream_t)stream;
  if (n <= 0) return OVM_OK;
  Tmp tmp;
  const size_t ne = (size_t)n * out_res * out_res * C;
  half_t *hi = tmp.get<half_t>(ne), *lo = tmp.get<half_t>(ne);
  if (!hi || !lo) return OVM_ERR_HIP;
  RoiParams rp; memset(&rp, 0, sizeof(rp));
  const float* f[3] = {p2, p3, p4};
  int nl = 0;
  for (int i = 0; i < 3; ++i) if (f[i]) { rp.feat[nl] = f[i]; rp.fh[nl] = hw[2 * i]; rp.fw[nl] = hw[2 * i + 1]; rp.scale[nl] = scales[i]; ++nl; }
  rp.C = C; rp.nlevels = nl; rp.min_level = min_level; rp.max_level = max_level; rp.out = out_res;
  rp.boxes = boxes; rp.batch_idx = image_idx; rp.n = n; rp.Ohi = hi; rp.Olo = lo; rp.ldo = out_res * out_res * C;
  int r = launch_roi_align(rp, s);
  if (r) return r;
  hipLaunchKernelGGL(join_kernel, dim3((unsigned)((ne + 255) / 256)), dim3(256), 0, s, hi, lo, (int64_t)ne, out);
  if (hipStreamSynchronize(s) != hipSuccess) return OVM_ERR_HIP;
  return OVM_OK;
}

int ovm_op_cube_decode(const float* head13, int32_t ld, const float* boxes, const float* scores, const int32_t* classes,
                       const int32_t* image_idx, const OvmImage* images, int32_t B, int32_t n, float virtual_focal,
                       int32_t postprocess, OvmDet3D* rec, int32_t* keep, ovm_stream_t stream) {
  hipStream_t s = (hipStream_t)stream;
  if (n <= 0) return OVM_OK;
  std::vector<ImageMeta> hm(B);
  for (int b = 0; b < B; ++b) {
    for (int i = 0; i < 9; ++i) hm[b].K[i] = images[b].K[i];
    hm[b].ratio = (float)((double)images[b].orig_height / (double)images[b].height);
    hm[b].net_h = images[b].height; hm[b].net_w = images[b].width;
    hm[b].orig_h = images[b].orig_height; hm[b].orig_w = images[b].orig_width;
  }
  Tmp tmp;
  ImageMeta* dm = tmp.get<ImageMeta>(B);
  if (!dm) return OVM_ERR_HIP;
  if (hipMemcpy(dm, hm.data(), sizeof(ImageMeta) * B, hipMemcpyHostToDevice) != hipSuccess) return OVM_ERR_HIP;
  CubeDecodeParams cp; memset(&cp, 0, sizeof(cp));
  cp.head = head13; cp.ldh = ld; cp.boxes = boxes; cp.scores = scores; cp.classes = classes; cp.batch_idx = image_idx;
  cp.meta = dm; cp.n = n; cp.virtual_focal = virtual_focal; cp.rec = (float*)rec; cp.keep = keep; cp.postprocess = postprocess;
  int r = launch_cube_decode(cp, s);
  if (r) return r;
  if (hipStreamSynchronize(s) != hipSuccess) return OVM_ERR_HIP;
  return OVM_OK;
}

// Tuning knobs for experiments (not part of the stable surface): "gemm_bm" = 0 (heuristic) | 128 | 256.
int ovm_tune_set(const char* key, int32_t value) {
  if (!key) return OVM_ERR_INVALID;
  if (!strcmp(key, "gemm_bm")) { gemm_set_force_bm(value); return OVM_OK; }
  if (!strcmp(key, "gemm_stages")) { gemm_set_stages(value); return OVM_OK; }
  if (!strcmp(key, "gemm_splitk")) { gemm_set_splitk(value); return OVM_OK; }
  if (!strcmp(key, "gemm_tail")) { gemm_set_tail_rows(value); return OVM_OK; }
  if (!strcmp(key, "attn_waves")) { attn_set_waves(value); return OVM_OK; }
  if (!strcmp(key, "attn_pp")) {
#ifndef OVM_DIAG
    if (value < 0 || value > 1) return OVM_ERR_INVALID;    // 2 / 3 select timing-only ablations with wrong results: -DOVM_DIAG builds only
#endif
    attn_set_pp(value); return OVM_OK;
  }
  if (!strcmp(key, "gdino_ffn_split")) { ovm::set_gdino_ffn_split(value); return OVM_OK; }   // plans built afterwards: 0 = one workgroup per row block runs the whole FFN
  if (!strcmp(key, "gdino_swin_fused")) { ovm::set_gdino_swin_fused(value); return OVM_OK; }  // plans built afterwards: 0 = qkv GEMM + window attention as two launches
  if (!strcmp(key, "gdino_gemm256")) { ovm::set_gdino_gemm256(value); return OVM_OK; }       // plans built afterwards: 0 = planar rows, 128 x 128 tiles
  if (!strcmp(key, "gdino_dec_chain")) { ovm::set_gdino_dec_chain(value); return OVM_OK; }   // plans built afterwards: 0 = one launch per decoder op
  if (!strcmp(key, "msdeform_vec")) { msdeform_set_vec(value); return OVM_OK; }
  if (!strcmp(key, "attn_q64")) { attn_set_q64(value); return OVM_OK; }      // 0: the 8-wave x 32-query attention kernel of rounds 1-2
  if (!strcmp(key, "attn_prio")) { attn_set_prio(value); return OVM_OK; }
  if (!strcmp(key, "attn_lds_pad")) { attn_set_lds_pad(value); return OVM_OK; }
  if (!strcmp(key, "attn_tail")) { attn_set_tail_rows(value); return OVM_OK; }
  if (!strcmp(key, "attn_tail_split")) { attn_set_tail_split(value); return OVM_OK; }
  if (!strcmp(key, "glin_small_max_tiles")) { glinear_set_small_max_tiles(value); return OVM_OK; }
  if (!strcmp(key, "glin_target_blocks")) { gemm_small_set(value, -1); return OVM_OK; }
  if (!strcmp(key, "gbmm_tiled")) { gbmm_set_tiled(value); return OVM_OK; }
  if (!strcmp(key, "glin_stages")) { gemm_small_set_stages(value); return OVM_OK; }
  if (!strcmp(key, "glin_wpe")) { gemm_small_set_wpe(value); return OVM_OK; }
  if (!strcmp(key, "glin_max_ksplit")) { gemm_small_set(-1, value); return OVM_OK; }
  if (!strcmp(key, "gemm256_ksplit")) { ovm::set_gemm256_ksplit(value); return OVM_OK; }
  if (!strcmp(key, "gemm256_n192")) { ovm::gemm256_set_n192(value); return OVM_OK; }     // qkv: 256 x 192 tiles when they fill the chip better (default 1)
  if (!strcmp(key, "gemm256")) { ovm::set_use_gemm256(value); return OVM_OK; }          // engine: 256 x 256 kernel for qkv / fc1 (default 1)
  if (!strcmp(key, "op_gemm256")) { g_op_gemm256 = value; return OVM_OK; }
  if (!strcmp(key, "gdino_branches")) { ovm::set_gdino_branches(value); return OVM_OK; }  // engines created afterwards: text branch on its own stream (default 1)              // ovm_op_gemm: force it, value = split-K hint
  return OVM_ERR_INVALID;
}

/* diagnostics: hands a device pointer to a named debug hook ("gemm256_stamps": u64 [8][128], NULL switches it off) */
int ovm_debug_set_ptr(const char* key, void* ptr) {
  if (key && !strcmp(key, "gemm256_stamps")) { g_gemm256_stamps = (unsigned long long*)ptr; return OVM_OK; }
#ifdef OVM_DIAG
  if (key && !strcmp(key, "attn_stamps")) { attn_set_stamps((unsigned long long*)ptr); return OVM_OK; }
#else
  if (key && !strcmp(key, "attn_stamps")) return OVM_ERR_UNSUPPORTED;      // stamp / ablation kernels exist in -DOVM_DIAG builds only (OVM_DIAG=1 csrc/build.sh)
#endif
  return OVM_ERR_INVALID;
}

int ovm_op_nms(const float* boxes, const float* scores, int32_t n, float thresh, int32_t* keep_idx, int32_t* n_keep,
               ovm_stream_t stream) {
  return launch_nms_single(boxes, scores, nullptr, n, thresh, keep_idx, n_keep, (hipStream_t)stream);
}

int ovm_gdino_postprocess(const float* pred_logits, int32_t nq, int32_t ld, const float* pred_boxes, const int32_t* spans, int32_t n_phrases,
                          int32_t img_h, int32_t img_w, float box_threshold, float nms_threshold, float* out_boxes, float* out_scores,
                          int32_t* out_classes, int32_t* n_out, ovm_stream_t stream) {
  return launch_gdino_post(pred_logits, nq, ld, pred_boxes, spans, n_phrases, img_h, img_w, box_threshold, nms_threshold, out_boxes,
                           out_scores, out_classes, n_out, (hipStream_t)stream);
}

}  // extern "C"
