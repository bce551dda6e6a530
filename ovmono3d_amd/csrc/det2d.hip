// RPN inference + 2D box head + Fast R-CNN inference on the device (see det2d.hpp).
//
// Follows detectron2's RPN.predict_proposals / find_top_rpn_proposals (reached from reference
// cubercnn/modeling/meta_arch/rcnn3d.py:106 through RPNWithIgnore, rpn.py:19-39), ROIHeads3D._forward_box
// (roi_heads.py:252-296) and fast_rcnn_inference_single_image (fast_rcnn.py:57-116); SURVEY.md Appendix
// A4, A6, A10 for the third-party semantics (top-k per level, Box2BoxTransform, clip, batched NMS).
#include "det2d.hpp"
#include "gdino.hpp"

#include <cmath>
#include <cstring>

namespace ovm {

namespace {

constexpr int TILE = 2048;
constexpr unsigned long long kInvalidKey = ~0ull;
constexpr unsigned long long kIdMask = (1ull << 24) - 1;

__device__ __forceinline__ unsigned int ord32(float f) {        // order-preserving float -> uint (ascending)
  const unsigned int u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// ------------------------------------------------------------------------------------------------
// Batched bitonic sort (ascending) of u64 keys [B][N], N a power of two >= TILE.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void bitonic_tile_kernel(unsigned long long* __restrict__ keys, int N, int kfix) {
  __shared__ unsigned long long s[TILE];
  const int base = blockIdx.x * TILE;
  unsigned long long* g = keys + (size_t)blockIdx.y * N + base;
  const int t = threadIdx.x;
  s[t] = g[t];
  s[t + 1024] = g[t + 1024];
  __syncthreads();
  const int k0 = kfix ? kfix : 2;
  const int k1 = kfix ? kfix : TILE;
  for (int k = k0; k <= k1; k <<= 1) {
    for (int j = (kfix ? (TILE >> 1) : (k >> 1)); j > 0; j >>= 1) {
      const int i = ((t / j) * 2 * j) + (t % j);
      const int l = i + j;
      const bool asc = (((base + i) & k) == 0);
      const unsigned long long a = s[i], b = s[l];
      if ((a > b) == asc) { s[i] = b; s[l] = a; }
      __syncthreads();
    }
    if (kfix) break;
  }
  g[t] = s[t];
  g[t + 1024] = s[t + 1024];
}

__global__ void bitonic_global_kernel(unsigned long long* __restrict__ keys, int N, int k, int j) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= N / 2) return;
  unsigned long long* g = keys + (size_t)blockIdx.y * N;
  const int i = ((t / j) * 2 * j) + (t % j);
  const int l = i + j;
  const bool asc = ((i & k) == 0);
  const unsigned long long a = g[i], b = g[l];
  if ((a > b) == asc) { g[i] = b; g[l] = a; }
}

int sort_keys(unsigned long long* keys, int N, int B, hipStream_t s) {
  if (N < TILE || (N & (N - 1))) return OVM_ERR_SHAPE;
  hipLaunchKernelGGL(bitonic_tile_kernel, dim3(N / TILE, B), dim3(1024), 0, s, keys, N, 0);
  for (int k = 2 * TILE; k <= N; k <<= 1) {
    for (int j = k >> 1; j >= TILE; j >>= 1)
      hipLaunchKernelGGL(bitonic_global_kernel, dim3((N / 2 + 255) / 256, B), dim3(256), 0, s, keys, N, k, j);
    hipLaunchKernelGGL(bitonic_tile_kernel, dim3(N / TILE, B), dim3(1024), 0, s, keys, N, k);
  }
  return hipGetLastError() == hipSuccess ? OVM_OK : OVM_ERR_HIP;
}

int pow2_at_least(int n) { int p = TILE; while (p < n) p <<= 1; return p; }

// ------------------------------------------------------------------------------------------------
// Bit-matrix NMS over candidates that are already sorted by decreasing score inside each group
// (group = FPN level for the RPN, class for the box head). Candidate i may be suppressed only by an
// earlier candidate of its own group with IoU > thr (strict), as torchvision.ops.nms.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool iou_gt(const float* a, const float* b, float thr) {
  const float xx1 = fmaxf(a[0], b[0]), yy1 = fmaxf(a[1], b[1]);
  const float xx2 = fminf(a[2], b[2]), yy2 = fminf(a[3], b[3]);
  const float w = fmaxf(xx2 - xx1, 0.f), h = fmaxf(yy2 - yy1, 0.f);
  const float inter = w * h;
  const float aa = (a[2] - a[0]) * (a[3] - a[1]), ab = (b[2] - b[0]) * (b[3] - b[1]);
  return inter / (aa + ab - inter) > thr;
}

// mask[(b*N + i)*W + w] bit t  <=>  candidate (gstart + w*64 + t) is suppressed by candidate i
__global__ void nms_mask_kernel(const float* __restrict__ box, const int* __restrict__ group, const int* __restrict__ gstart,
                                const int* __restrict__ gend, int ngroups, int N, int W, float thr,
                                unsigned long long* __restrict__ mask) {
  // one thread per (candidate i, 64-candidate word w): 64 IoU tests each
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int b = blockIdx.y;
  if (idx >= (long)N * W) return;
  const int i = (int)(idx / W), w = (int)(idx - (long)i * W);
  unsigned long long* out = mask + ((size_t)b * N + i) * W + w;
  const int g = group[(size_t)b * N + i];
  if (g < 0) { *out = 0ull; return; }
  const int st = gstart[b * ngroups + g], en = gend[b * ngroups + g];
  const float* bx = box + (size_t)b * N * 4;
  unsigned long long bits = 0ull;
  const int j0 = st + w * 64;
  if (j0 + 63 > i && j0 < en) {
    const float me[4] = {bx[i * 4], bx[i * 4 + 1], bx[i * 4 + 2], bx[i * 4 + 3]};
    for (int t = 0; t < 64; ++t) {
      const int j = j0 + t;
      if (j > i && j < en && group[(size_t)b * N + j] == g && iou_gt(me, bx + (size_t)j * 4, thr)) bits |= (1ull << t);
    }
  }
  *out = bits;
}

// one wave per (group, image): sequential greedy pass; lane w owns removed-word w
__global__ __launch_bounds__(64) void nms_scan_kernel(const int* __restrict__ group, const int* __restrict__ gstart,
                                                      const int* __restrict__ gend, int ngroups, int N, int W,
                                                      const unsigned long long* __restrict__ mask, int* __restrict__ keep) {
  const int g = blockIdx.x, b = blockIdx.y, lane = threadIdx.x;
  const int st = gstart[b * ngroups + g], en = gend[b * ngroups + g];
  unsigned long long remv = 0ull;
  for (int i0 = st; i0 < en; i0 += 8) {
    unsigned long long m[8]; int gv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = i0 + u;
      m[u] = (i < en && lane < W) ? mask[((size_t)b * N + i) * W + lane] : 0ull;
      gv[u] = (i < en) ? group[(size_t)b * N + i] : -1;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = i0 + u;
      if (i >= en) break;
      const int rel = i - st;
      const unsigned long long wv = __shfl(remv, rel >> 6, 64);
      const bool removed = (wv >> (rel & 63)) & 1ull;
      const bool kp = (!removed) && gv[u] == g;
      if (kp) remv |= m[u];
      if (lane == 0) keep[(size_t)b * N + i] = kp ? 1 : 0;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// RPN
// ------------------------------------------------------------------------------------------------
struct RpnGeom {
  int nlev;
  int HW[kMaxLevels], Wl[kMaxLevels], A_off[kMaxLevels + 1];     // anchors per level prefix (A = 3 per cell)
  float base[kMaxLevels][3][4];   // [level][ratio][xyxy] cell anchors (computed in double on the host)
  float stride[kMaxLevels];
  const float* o[kMaxLevels];     // per level [B*HW][16]: 3 objectness logits + 12 deltas
};

// key = level (2 bits @58) | ~ord(score) (32 bits @24) | local anchor index (24 bits)
__global__ void rpn_keys_kernel(RpnGeom gm, int N, unsigned long long* __restrict__ keys) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  const int b = blockIdx.y;
  if (g >= N) return;
  unsigned long long key = kInvalidKey;
  if (g < gm.A_off[gm.nlev]) {
    int l = 0;
    for (int q = 1; q < gm.nlev; ++q) l = (g >= gm.A_off[q]) ? q : l;
    const int local = g - gm.A_off[l];
    const int cell = local / 3, a = local - cell * 3;
    const float* o = gm.o[l];
    const float sc = o[((size_t)b * gm.HW[l] + cell) * 16 + a];
    key = ((unsigned long long)l << 58) | ((unsigned long long)(~ord32(sc)) << 24) | (unsigned long long)local;
  }
  keys[(size_t)b * N + g] = key;
}

__global__ void rpn_decode_kernel(const unsigned long long* __restrict__ keys, RpnGeom gm, int N, int pre_topk,
                                  const ImageMeta* __restrict__ meta, float scale_clamp, float* __restrict__ cbox,
                                  float* __restrict__ cscore, int* __restrict__ cgroup, int* __restrict__ gstart,
                                  int* __restrict__ gend) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  const int b = blockIdx.y;
  const int NS = gm.nlev * pre_topk;
  if (s >= NS) return;
  const int l = s / pre_topk, i = s - l * pre_topk;
  const int A_l = gm.A_off[l + 1] - gm.A_off[l];
  const int k_l = A_l < pre_topk ? A_l : pre_topk;
  if (i == 0) { gstart[b * gm.nlev + l] = l * pre_topk; gend[b * gm.nlev + l] = l * pre_topk + k_l; }
  const size_t so = (size_t)b * NS + s;
  if (i >= k_l) { cgroup[so] = -1; cscore[so] = 0.f; cbox[so * 4] = cbox[so * 4 + 1] = cbox[so * 4 + 2] = cbox[so * 4 + 3] = 0.f; return; }
  const unsigned long long key = keys[(size_t)b * N + gm.A_off[l] + i];
  const int local = (int)(key & kIdMask);
  const int cell = local / 3, a = local - cell * 3;
  const int y = cell / gm.Wl[l], x = cell - y * gm.Wl[l];
  const float* o = gm.o[l] + ((size_t)b * gm.HW[l] + cell) * 16;
  const float sc = o[a];
  const float sx = (float)x * gm.stride[l], sy = (float)y * gm.stride[l];
  const float ax1 = sx + gm.base[l][a][0], ay1 = sy + gm.base[l][a][1];
  const float ax2 = sx + gm.base[l][a][2], ay2 = sy + gm.base[l][a][3];
  // Box2BoxTransform(1,1,1,1).apply_deltas
  const float widths = ax2 - ax1, heights = ay2 - ay1;
  const float ctr_x = ax1 + 0.5f * widths, ctr_y = ay1 + 0.5f * heights;
  const float dx = o[3 + a * 4 + 0], dy = o[3 + a * 4 + 1];
  const float dw = fminf(o[3 + a * 4 + 2], scale_clamp), dh = fminf(o[3 + a * 4 + 3], scale_clamp);
  const float pcx = dx * widths + ctr_x, pcy = dy * heights + ctr_y;
  const float pw = expf(dw) * widths, ph = expf(dh) * heights;
  float x1 = pcx - 0.5f * pw, y1 = pcy - 0.5f * ph, x2 = pcx + 0.5f * pw, y2 = pcy + 0.5f * ph;
  bool ok = isfinite(x1) && isfinite(y1) && isfinite(x2) && isfinite(y2) && isfinite(sc);
  const float H = (float)meta[b].net_h, Wd = (float)meta[b].net_w;
  x1 = fminf(fmaxf(x1, 0.f), Wd); x2 = fminf(fmaxf(x2, 0.f), Wd);
  y1 = fminf(fmaxf(y1, 0.f), H);  y2 = fminf(fmaxf(y2, 0.f), H);
  ok = ok && (x2 - x1) > 0.f && (y2 - y1) > 0.f;                         // min_box_size 0
  cbox[so * 4 + 0] = x1; cbox[so * 4 + 1] = y1; cbox[so * 4 + 2] = x2; cbox[so * 4 + 3] = y2;
  cscore[so] = sc;
  cgroup[so] = ok ? l : -1;
}

__global__ void rpn_merge_keys_kernel(const float* __restrict__ cscore, const int* __restrict__ ckeep, int NS, int N,
                                      unsigned long long* __restrict__ mkeys) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  const int b = blockIdx.y;
  if (s >= N) return;
  unsigned long long key = kInvalidKey;
  if (s < NS && ckeep[(size_t)b * NS + s])
    key = ((unsigned long long)(~ord32(cscore[(size_t)b * NS + s])) << 24) | (unsigned long long)s;
  mkeys[(size_t)b * N + s] = key;
}

__global__ void rpn_emit_kernel(const unsigned long long* __restrict__ mkeys, int N, int NS, const float* __restrict__ cbox,
                                const float* __restrict__ cscore, int R, float* __restrict__ pbox, float* __restrict__ pscore,
                                int* __restrict__ pbidx, int* __restrict__ pcount) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  const int b = blockIdx.y;
  if (r >= R) return;
  const size_t o = (size_t)b * R + r;
  const unsigned long long key = (r < N) ? mkeys[(size_t)b * N + r] : kInvalidKey;
  pbidx[o] = b;
  if (key == kInvalidKey) { pbox[o * 4] = pbox[o * 4 + 1] = pbox[o * 4 + 2] = pbox[o * 4 + 3] = 0.f; pscore[o] = 0.f; return; }
  const int s = (int)(key & kIdMask);
  const size_t so = (size_t)b * NS + s;
#pragma unroll
  for (int c = 0; c < 4; ++c) pbox[o * 4 + c] = cbox[so * 4 + c];
  pscore[o] = cscore[so];
  atomicAdd(&pcount[b], 1);
}

// ------------------------------------------------------------------------------------------------
// Box head post-processing
// ------------------------------------------------------------------------------------------------
// One wave per proposal row: softmax over K+1 logits, per-class box decode (weights 10,10,5,5), finite
// test on the raw decoded row, clip, score threshold -> dense candidate table + sort keys.
__global__ __launch_bounds__(256) void boxhead_dense_kernel(const float* __restrict__ HO, int ldh, const float* __restrict__ pbox,
                                                            const int* __restrict__ pcount, const ImageMeta* __restrict__ meta,
                                                            int R, int K, int M, float thresh, float scale_clamp, int Ncand,
                                                            float* __restrict__ probs, float* __restrict__ dbox,
                                                            unsigned long long* __restrict__ keys) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= M) return;
  const int b = row / R, r = row - b * R;
  const float* h = HO + (size_t)row * ldh;
  const float logit = (lane <= K) ? h[lane] : -INFINITY;
  const float mx = wave_max(logit);
  const float e = (lane <= K) ? expf(logit - mx) : 0.f;
  const float sum = wave_sum(e);
  const float p = e / sum;
  if (lane <= K) probs[(size_t)row * 64 + lane] = p;
  bool fin = (lane <= K) ? isfinite(p) : true;
  float x1 = 0.f, y1 = 0.f, x2 = 0.f, y2 = 0.f;
  if (lane < K) {
    const float* d = h + (K + 1) + lane * 4;
    const float bx1 = pbox[(size_t)row * 4], by1 = pbox[(size_t)row * 4 + 1], bx2 = pbox[(size_t)row * 4 + 2], by2 = pbox[(size_t)row * 4 + 3];
    const float widths = bx2 - bx1, heights = by2 - by1;
    const float ctr_x = bx1 + 0.5f * widths, ctr_y = by1 + 0.5f * heights;
    const float dx = d[0] / 10.f, dy = d[1] / 10.f;
    const float dw = fminf(d[2] / 5.f, scale_clamp), dh = fminf(d[3] / 5.f, scale_clamp);
    const float pcx = dx * widths + ctr_x, pcy = dy * heights + ctr_y;
    const float pw = expf(dw) * widths, ph = expf(dh) * heights;
    x1 = pcx - 0.5f * pw; y1 = pcy - 0.5f * ph; x2 = pcx + 0.5f * pw; y2 = pcy + 0.5f * ph;
    fin = fin && isfinite(x1) && isfinite(y1) && isfinite(x2) && isfinite(y2);
    const float H = (float)meta[b].net_h, Wd = (float)meta[b].net_w;
    x1 = fminf(fmaxf(x1, 0.f), Wd); x2 = fminf(fmaxf(x2, 0.f), Wd);
    y1 = fminf(fmaxf(y1, 0.f), H);  y2 = fminf(fmaxf(y2, 0.f), H);
    float* o = dbox + ((size_t)row * K + lane) * 4;
    o[0] = x1; o[1] = y1; o[2] = x2; o[3] = y2;
  }
  const bool row_ok = __all(fin) && (r < pcount[b]);                    // fast_rcnn.py:76-79 drops the whole row
  if (lane < K) {
    unsigned long long key = kInvalidKey;
    if (row_ok && p > thresh)                                           // fast_rcnn.py:91 (strict)
      key = ((unsigned long long)lane << 58) | ((unsigned long long)(~ord32(p)) << 24) | (unsigned long long)(r * K + lane);
    keys[(size_t)b * Ncand + r * K + lane] = key;
  }
}

__global__ void cand_gather_kernel(const unsigned long long* __restrict__ keys, int N, int R, int K, const float* __restrict__ dbox,
                                   float* __restrict__ sbox, int* __restrict__ sgroup) {
  const int pos = blockIdx.x * blockDim.x + threadIdx.x;
  const int b = blockIdx.y;
  if (pos >= N) return;
  const unsigned long long key = keys[(size_t)b * N + pos];
  const size_t o = (size_t)b * N + pos;
  if (key == kInvalidKey) { sgroup[o] = -1; return; }
  const int id = (int)(key & kIdMask);
  sgroup[o] = (int)(key >> 58);
  const float* d = dbox + ((size_t)b * R * K + id) * 4;
#pragma unroll
  for (int c = 0; c < 4; ++c) sbox[o * 4 + c] = d[c];
}

__global__ void seg_bounds_kernel(const int* __restrict__ sgroup, int N, int ngroups, int* __restrict__ gstart, int* __restrict__ gend) {
  const int pos = blockIdx.x * blockDim.x + threadIdx.x;
  const int b = blockIdx.y;
  if (pos >= N) return;
  const int g = sgroup[(size_t)b * N + pos];
  if (g < 0) return;
  if (pos == 0 || sgroup[(size_t)b * N + pos - 1] != g) gstart[b * ngroups + g] = pos;
  if (pos == N - 1 || sgroup[(size_t)b * N + pos + 1] != g) gend[b * ngroups + g] = pos + 1;
}

__global__ void final_keys_kernel(unsigned long long* __restrict__ keys, const int* __restrict__ skeep, int N) {
  const int pos = blockIdx.x * blockDim.x + threadIdx.x;
  const int b = blockIdx.y;
  if (pos >= N) return;
  const size_t o = (size_t)b * N + pos;
  const unsigned long long key = keys[o];
  keys[o] = (key != kInvalidKey && skeep[o]) ? (key & ((1ull << 56) - 1)) : kInvalidKey;   // drop the class bits
}

// single workgroup: walks the images in order so outputs are image-major and compact
__global__ __launch_bounds__(1024) void boxhead_emit_kernel(const unsigned long long* __restrict__ keys, int N, int B, int R, int K,
                                                            int topk, const float* __restrict__ dbox, const float* __restrict__ probs,
                                                            float* __restrict__ boxes, float* __restrict__ scores, int* __restrict__ classes,
                                                            int* __restrict__ image_idx, float* __restrict__ scores_full,
                                                            int* __restrict__ out_counts) {
  const int t = threadIdx.x;
  int base = 0;
  for (int b = 0; b < B; ++b) {
    unsigned long long key = kInvalidKey;
    if (t < topk && t < N) key = keys[(size_t)b * N + t];
    const int valid = key != kInvalidKey;
    const int cnt = __syncthreads_count(valid);
    if (valid) {
      const int id = (int)(key & kIdMask);
      const int r = id / K, c = id - r * K;
      const int o = base + t;                       // sorted => valid keys occupy ranks 0..cnt-1
      const float* d = dbox + ((size_t)b * R * K + id) * 4;
#pragma unroll
      for (int q = 0; q < 4; ++q) boxes[(size_t)o * 4 + q] = d[q];
      const float* pr = probs + ((size_t)b * R + r) * 64;
      scores[o] = pr[c];
      classes[o] = c;
      image_idx[o] = b;
      if (scores_full) for (int q = 0; q < K; ++q) scores_full[(size_t)o * K + q] = pr[q];
    }
    if (t == 0) out_counts[b] = cnt;
    base += cnt;
    __syncthreads();
  }
}

template <typename Tp>
int dmalloc(Tp** p, size_t n, std::vector<void*>* allocs) {
  void* q = nullptr;
  if (hipMalloc(&q, n * sizeof(Tp) + 16) != hipSuccess) return OVM_ERR_HIP;
  if (hipMemset(q, 0, n * sizeof(Tp) + 16) != hipSuccess) return OVM_ERR_HIP;
  allocs->push_back(q);
  *p = (Tp*)q;
  return OVM_OK;
}

}  // namespace

int det2d_alloc(Det2dWorkspace* w, int B, int nlev, const int* sides, int C, int num_classes, int maxR, int pre_topk, int post_topk,
                int topk, std::vector<void*>* allocs) {
  memset(w, 0, sizeof(*w));
  if (nlev < 1 || nlev > kMaxLevels) return OVM_ERR_INVALID;
  w->maxB = B; w->nlev = nlev; w->C = C; w->num_classes = num_classes; w->pre_topk = pre_topk; w->topk = topk;
  w->R = post_topk < maxR ? post_topk : maxR;
  int A = 0;
  for (int l = 0; l < nlev; ++l) { w->Wl[l] = sides[l]; w->HW[l] = sides[l] * sides[l]; A += w->HW[l] * 3; }
  if (A >= (1 << 24)) return OVM_ERR_CAPACITY;                          // the sort key carries a 24-bit local anchor index
  w->A_tot = A;
  w->Nrpn = pow2_at_least(A);
  w->Ncand = pow2_at_least(w->R * num_classes);
  w->Nmerge = pow2_at_least(nlev * pre_topk);
  if (pre_topk > 1024 || w->R > 1024 || topk > 1024 || num_classes > 63 || (size_t)w->R * num_classes >= (1u << 24)) return OVM_ERR_CAPACITY;
  int r;
  for (int l = 0; l < nlev; ++l) {
    if ((r = dmalloc(&w->rpn_t[l].hi, (size_t)B * w->HW[l] * C, allocs))) return r;
    if ((r = dmalloc(&w->rpn_t[l].lo, (size_t)B * w->HW[l] * C, allocs))) return r;
    if ((r = dmalloc(&w->rpn_o[l], (size_t)B * w->HW[l] * 16, allocs))) return r;
  }
  const size_t nk = (size_t)B * (w->Nrpn > w->Ncand ? w->Nrpn : w->Ncand);
  if ((r = dmalloc(&w->keys, nk, allocs))) return r;
  const size_t NS = (size_t)B * nlev * pre_topk;
  if ((r = dmalloc(&w->cbox, NS * 4, allocs))) return r;
  if ((r = dmalloc(&w->cscore, NS, allocs))) return r;
  if ((r = dmalloc(&w->cgroup, NS, allocs))) return r;
  if ((r = dmalloc(&w->ckeep, NS, allocs))) return r;
  const int ng = num_classes > nlev ? num_classes : nlev;
  if ((r = dmalloc(&w->gstart, (size_t)B * ng, allocs))) return r;
  if ((r = dmalloc(&w->gend, (size_t)B * ng, allocs))) return r;
  const size_t nmask = (size_t)B * (w->Ncand > nlev * pre_topk ? w->Ncand : nlev * pre_topk) * 16;
  if ((r = dmalloc(&w->mask, nmask, allocs))) return r;
  if ((r = dmalloc(&w->mkeys, (size_t)B * w->Nmerge, allocs))) return r;
  const size_t BR = (size_t)B * w->R;
  if ((r = dmalloc(&w->prop_boxes, BR * 4, allocs))) return r;
  if ((r = dmalloc(&w->prop_scores, BR, allocs))) return r;
  if ((r = dmalloc(&w->prop_bidx, BR, allocs))) return r;
  if ((r = dmalloc(&w->prop_count, (size_t)B, allocs))) return r;
  if ((r = dmalloc(&w->probs, BR * 64, allocs))) return r;
  if ((r = dmalloc(&w->dbox, BR * num_classes * 4, allocs))) return r;
  if ((r = dmalloc(&w->sbox, (size_t)B * w->Ncand * 4, allocs))) return r;
  if ((r = dmalloc(&w->sgroup, (size_t)B * w->Ncand, allocs))) return r;
  if ((r = dmalloc(&w->skeep, (size_t)B * w->Ncand, allocs))) return r;
  return OVM_OK;
}

int det2d_forward(const Det2dModel& m, Det2dWorkspace& w, float* boxes, float* scores, int* classes, int* image_idx,
                  float* scores_full, int* out_counts, hipStream_t s) {
  const int B = m.B, C = m.C, K = m.num_classes;
  if (B > w.maxB || m.pre_topk != w.pre_topk || K != w.num_classes) return OVM_ERR_CAPACITY;
  const float scale_clamp = (float)std::log(1000.0 / 16.0);
  const int nlev = m.nlev;
  if (nlev != w.nlev) return OVM_ERR_CAPACITY;
  RpnGeom gm; memset(&gm, 0, sizeof(gm));
  gm.nlev = nlev;
  gm.A_off[0] = 0;
  for (int l = 0; l < nlev; ++l) {
    gm.HW[l] = w.HW[l]; gm.Wl[l] = w.Wl[l]; gm.A_off[l + 1] = gm.A_off[l] + w.HW[l] * 3;
    gm.stride[l] = m.stride[l]; gm.o[l] = w.rpn_o[l];
    for (int a = 0; a < 3; ++a) {                         // DefaultAnchorGenerator.generate_cell_anchors
      const double area = (double)m.anchor_sizes[l] * (double)m.anchor_sizes[l];
      const double ww = std::sqrt(area / (double)m.anchor_ratios[a]);
      const double hh = (double)m.anchor_ratios[a] * ww;
      gm.base[l][a][0] = (float)(-ww / 2.0); gm.base[l][a][1] = (float)(-hh / 2.0);
      gm.base[l][a][2] = (float)(ww / 2.0);  gm.base[l][a][3] = (float)(hh / 2.0);
    }
  }
  // ---- RPN head: conv3x3+bias+ReLU (implicit GEMM over the zero-bordered fp16 pyramid), then the 1x1s ----
  for (int l = 0; l < nlev; ++l) {
    const int Hs = w.Wl[l], M = B * w.HW[l];
    GemmParams p; memset(&p, 0, sizeof(p));
    p.Ahi = m.rpad[l].hi; p.Alo = m.rpad[l].lo; p.Whi = m.rpn_conv_hi; p.Wlo = m.rpn_conv_lo;
    p.M = M; p.N = C; p.K = 9 * C; p.cH = Hs; p.cW = Hs; p.cC = C; p.bias = m.rpn_conv_bias; p.relu = 1;
    p.Ohi = w.rpn_t[l].hi; p.Olo = (m.npass == 3) ? w.rpn_t[l].lo : nullptr; p.ldo = C;
    int r = launch_gemm(p, m.npass, EPI_STORE, A_CONV3X3, s);
    if (r) return r;
    GemmParams q; memset(&q, 0, sizeof(q));
    q.Ahi = w.rpn_t[l].hi; q.Alo = w.rpn_t[l].lo; q.lda = C; q.Whi = m.rpn_out_hi; q.Wlo = m.rpn_out_lo;
    q.M = M; q.N = 15; q.K = C; q.bias = m.rpn_out_bias; q.C = w.rpn_o[l]; q.ldc = 16;
    r = launch_gemm(q, m.npass, EPI_STORE, A_ROWMAJOR, s);
    if (r) return r;
  }
  // ---- per-level top-k (sort of (level, score, index) keys), decode, clip ----
  const int N = w.Nrpn, NS = nlev * m.pre_topk;
  hipLaunchKernelGGL(rpn_keys_kernel, dim3((N + 255) / 256, B), dim3(256), 0, s, gm, N, w.keys);
  int r = sort_keys(w.keys, N, B, s);
  if (r) return r;
  hipLaunchKernelGGL(rpn_decode_kernel, dim3((NS + 255) / 256, B), dim3(256), 0, s, w.keys, gm, N,
                     m.pre_topk, m.meta, scale_clamp, w.cbox, w.cscore, w.cgroup, w.gstart, w.gend);
  // ---- per-level NMS, merge by score, keep post_topk ----
  hipLaunchKernelGGL(nms_mask_kernel, dim3((NS * 16 + 127) / 128, B), dim3(128), 0, s, w.cbox, w.cgroup, w.gstart, w.gend, nlev, NS, 16,
                     m.rpn_nms, w.mask);
  hipLaunchKernelGGL(nms_scan_kernel, dim3(nlev, B), dim3(64), 0, s, w.cgroup, w.gstart, w.gend, nlev, NS, 16, w.mask, w.ckeep);
  hipLaunchKernelGGL(rpn_merge_keys_kernel, dim3((w.Nmerge + 255) / 256, B), dim3(256), 0, s, w.cscore, w.ckeep, NS, w.Nmerge, w.mkeys);
  r = sort_keys(w.mkeys, w.Nmerge, B, s);
  if (r) return r;
  if (hipMemsetAsync(w.prop_count, 0, sizeof(int) * B, s) != hipSuccess) return OVM_ERR_HIP;
  const int R = w.R;
  hipLaunchKernelGGL(rpn_emit_kernel, dim3((R + 255) / 256, B), dim3(256), 0, s, w.mkeys, w.Nmerge, NS, w.cbox, w.cscore, R,
                     w.prop_boxes, w.prop_scores, w.prop_bidx, w.prop_count);
  // ---- box head: ROIAlign -> fc1 -> fc2 -> (cls_score | bbox_pred) ----
  const int M = B * R;
  RoiParams rp = m.roi;
  rp.boxes = w.prop_boxes; rp.batch_idx = w.prop_bidx; rp.n = M; rp.Ohi = m.RF.hi; rp.Olo = m.RF.lo; rp.ldo = m.roiK;
  r = launch_roi_align(rp, s);
  if (r) return r;
  {
    GemmParams p; memset(&p, 0, sizeof(p));
    p.Ahi = m.RF.hi; p.Alo = m.RF.lo; p.lda = m.roiK; p.Whi = m.fc1_hi; p.Wlo = m.fc1_lo; p.M = M; p.N = m.F; p.K = m.roiK;
    p.bias = m.fc1_bias; p.relu = 1; p.Ohi = m.H1.hi; p.Olo = m.H1.lo; p.ldo = m.F;
    if ((r = launch_gemm(p, m.npass, EPI_STORE, A_ROWMAJOR, s))) return r;
    GemmParams q; memset(&q, 0, sizeof(q));
    q.Ahi = m.H1.hi; q.Alo = m.H1.lo; q.lda = m.F; q.Whi = m.fc2_hi; q.Wlo = m.fc2_lo; q.M = M; q.N = m.F; q.K = m.F;
    q.bias = m.fc2_bias; q.relu = 1; q.Ohi = m.H2.hi; q.Olo = m.H2.lo; q.ldo = m.F;
    if ((r = launch_gemm(q, m.npass, EPI_STORE, A_ROWMAJOR, s))) return r;
    GemmParams o; memset(&o, 0, sizeof(o));
    o.Ahi = m.H2.hi; o.Alo = m.H2.lo; o.lda = m.F; o.Whi = m.out_hi; o.Wlo = m.out_lo; o.M = M; o.N = (K + 1) + 4 * K; o.K = m.F;
    o.bias = m.out_bias; o.C = m.HO; o.ldc = 256;
    if (o.N > 256) return OVM_ERR_CAPACITY;
    if ((r = launch_gemm(o, m.npass, EPI_STORE, A_ROWMAJOR, s))) return r;
  }
  // ---- softmax + decode + threshold -> per-class NMS -> top-k ----
  const int Nc = w.Ncand;
  if (hipMemsetAsync(w.keys, 0xFF, sizeof(unsigned long long) * (size_t)B * Nc, s) != hipSuccess) return OVM_ERR_HIP;
  hipLaunchKernelGGL(boxhead_dense_kernel, dim3((M + 3) / 4), dim3(256), 0, s, m.HO, 256, w.prop_boxes, w.prop_count, m.meta, R, K, M,
                     m.score_thresh, scale_clamp, Nc, w.probs, w.dbox, w.keys);
  if ((r = sort_keys(w.keys, Nc, B, s))) return r;
  hipLaunchKernelGGL(cand_gather_kernel, dim3((Nc + 255) / 256, B), dim3(256), 0, s, w.keys, Nc, R, K, w.dbox, w.sbox, w.sgroup);
  if (hipMemsetAsync(w.gstart, 0, sizeof(int) * (size_t)B * K, s) != hipSuccess) return OVM_ERR_HIP;
  if (hipMemsetAsync(w.gend, 0, sizeof(int) * (size_t)B * K, s) != hipSuccess) return OVM_ERR_HIP;
  hipLaunchKernelGGL(seg_bounds_kernel, dim3((Nc + 255) / 256, B), dim3(256), 0, s, w.sgroup, Nc, K, w.gstart, w.gend);
  hipLaunchKernelGGL(nms_mask_kernel, dim3((Nc * 16 + 127) / 128, B), dim3(128), 0, s, w.sbox, w.sgroup, w.gstart, w.gend, K, Nc, 16,
                     m.nms_thresh, w.mask);
  hipLaunchKernelGGL(nms_scan_kernel, dim3(K, B), dim3(64), 0, s, w.sgroup, w.gstart, w.gend, K, Nc, 16, w.mask, w.skeep);
  hipLaunchKernelGGL(final_keys_kernel, dim3((Nc + 255) / 256, B), dim3(256), 0, s, w.keys, w.skeep, Nc);
  if ((r = sort_keys(w.keys, Nc, B, s))) return r;
  hipLaunchKernelGGL(boxhead_emit_kernel, dim3(1), dim3(1024), 0, s, w.keys, Nc, B, R, K, m.topk, w.dbox, w.probs, boxes, scores,
                     classes, image_idx, scores_full, out_counts);
  return hipGetLastError() == hipSuccess ? OVM_OK : OVM_ERR_HIP;
}

// ------------------------------------------------------------------------------------------------
// Standalone class-agnostic NMS (op-level entry; also the GroundingDINO glue's NMS, roi_heads_gdino.py:254)
// ------------------------------------------------------------------------------------------------
namespace {
__global__ void single_keys_kernel(const float* __restrict__ scores, const int* __restrict__ valid, int n, int N,
                                   unsigned long long* __restrict__ keys) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const bool ok = (i < n) && (!valid || valid[i]);
  keys[i] = ok ? (((unsigned long long)(~ord32(scores[i])) << 24) | (unsigned long long)i) : kInvalidKey;
}
__global__ void single_gather_kernel(const unsigned long long* __restrict__ keys, const float* __restrict__ boxes, int n, int N,
                                     float* __restrict__ sbox, int* __restrict__ sgroup, int* __restrict__ gse) {
  const int pos = blockIdx.x * blockDim.x + threadIdx.x;
  if (pos == 0) { gse[0] = 0; gse[1] = n; }
  if (pos >= N) return;
  const unsigned long long key = keys[pos];
  if (key == kInvalidKey) { sgroup[pos] = -1; return; }
  const int i = (int)(key & kIdMask);
  sgroup[pos] = 0;
#pragma unroll
  for (int c = 0; c < 4; ++c) sbox[pos * 4 + c] = boxes[i * 4 + c];
}
__global__ __launch_bounds__(1024) void single_emit_kernel(const unsigned long long* __restrict__ keys, const int* __restrict__ keep,
                                                           int n, int* __restrict__ keep_idx, int* __restrict__ n_keep) {
  __shared__ int wsum[16];
  __shared__ int base_s;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  if (t == 0) base_s = 0;
  __syncthreads();
  for (int st = 0; st < n; st += 1024) {
    const int pos = st + t;
    const int k = (pos < n) ? keep[pos] : 0;
    const unsigned long long bal = __ballot(k != 0);
    const int pre = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) wsum[wave] = __popcll(bal);
    __syncthreads();
    int woff = 0, tot = 0;
    for (int w2 = 0; w2 < 16; ++w2) { if (w2 < wave) woff += wsum[w2]; tot += wsum[w2]; }
    const int base = base_s;
    if (k) keep_idx[base + woff + pre] = (int)(keys[pos] & kIdMask);
    __syncthreads();
    if (t == 0) base_s = base + tot;
    __syncthreads();
  }
  if (t == 0) *n_keep = base_s;
}
}  // namespace

int launch_nms_single(const float* boxes, const float* scores, const int* valid, int n, float thresh, int* keep_idx, int* n_keep,
                      hipStream_t s) {
  if (n <= 0) return hipMemsetAsync(n_keep, 0, sizeof(int), s) == hipSuccess ? OVM_OK : OVM_ERR_HIP;
  if (n > 4096) return OVM_ERR_CAPACITY;
  const int N = pow2_at_least(n), W = (n + 63) / 64;
  unsigned long long *keys = nullptr, *mask = nullptr; float* sbox = nullptr; int *sgroup = nullptr, *gse = nullptr, *keep = nullptr;
  std::vector<void*> tmp;
  int r;
  if ((r = dmalloc(&keys, (size_t)N, &tmp)) || (r = dmalloc(&mask, (size_t)N * W, &tmp)) || (r = dmalloc(&sbox, (size_t)N * 4, &tmp)) ||
      (r = dmalloc(&sgroup, (size_t)N, &tmp)) || (r = dmalloc(&gse, 2, &tmp)) || (r = dmalloc(&keep, (size_t)N, &tmp))) {
    for (void* p : tmp) (void)hipFree(p);
    return r;
  }
  hipLaunchKernelGGL(single_keys_kernel, dim3((N + 255) / 256), dim3(256), 0, s, scores, valid, n, N, keys);
  r = sort_keys(keys, N, 1, s);
  if (!r) {
    hipLaunchKernelGGL(single_gather_kernel, dim3((N + 255) / 256), dim3(256), 0, s, keys, boxes, n, N, sbox, sgroup, gse);
    hipLaunchKernelGGL(nms_mask_kernel, dim3((unsigned)(((long)N * W + 127) / 128), 1), dim3(128), 0, s, sbox, sgroup, gse, gse + 1, 1, N, W, thresh, mask);
    hipLaunchKernelGGL(nms_scan_kernel, dim3(1, 1), dim3(64), 0, s, sgroup, gse, gse + 1, 1, N, W, mask, keep);
    hipLaunchKernelGGL(single_emit_kernel, dim3(1), dim3(1024), 0, s, keys, keep, n, keep_idx, n_keep);
    if (hipStreamSynchronize(s) != hipSuccess) r = OVM_ERR_HIP;
  }
  for (void* p : tmp) (void)hipFree(p);
  return r;
}

// ------------------------------------------------------------------------------------------------
// GroundingDINO output glue (reference cubercnn/modeling/roi_heads/roi_heads_gdino.py:186-202,253,266-294):
// sigmoid of the token logits, per-phrase SUM over the phrase's token span, max / first-argmax over phrases,
// strict threshold, cxcywh (normalised) -> xyxy pixels, class-agnostic NMS.
// ------------------------------------------------------------------------------------------------
namespace {
__global__ void gdino_phrase_kernel(const float* __restrict__ logits, int nq, int ld, const float* __restrict__ cxcywh,
                                    const int* __restrict__ spans, int K, float img_h, float img_w, float thr,
                                    float* __restrict__ xyxy, float* __restrict__ score, int* __restrict__ cls,
                                    int* __restrict__ valid) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= nq) return;
  const float* l = logits + (size_t)q * ld;
  float best = -INFINITY; int arg = 0;
  for (int k = 0; k < K; ++k) {
    float sum = 0.f;
    for (int t = spans[2 * k]; t < spans[2 * k + 1]; ++t) sum += 1.0f / (1.0f + expf(-l[t]));
    if (sum > best) { best = sum; arg = k; }          // torch.max returns the first maximal index
  }
  const float cx = cxcywh[q * 4] * img_w, cy = cxcywh[q * 4 + 1] * img_h, w = cxcywh[q * 4 + 2] * img_w, h = cxcywh[q * 4 + 3] * img_h;
  xyxy[q * 4 + 0] = cx - 0.5f * w; xyxy[q * 4 + 1] = cy - 0.5f * h;
  xyxy[q * 4 + 2] = cx + 0.5f * w; xyxy[q * 4 + 3] = cy + 0.5f * h;
  score[q] = best; cls[q] = arg;
  valid[q] = (K > 0 && best > thr) ? 1 : 0;
}
__global__ void gdino_gather_kernel(const int* __restrict__ keep_idx, const int* __restrict__ n_keep, const float* __restrict__ xyxy,
                                    const float* __restrict__ score, const int* __restrict__ cls, float* __restrict__ ob,
                                    float* __restrict__ os, int* __restrict__ oc) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= *n_keep) return;
  const int q = keep_idx[i];
#pragma unroll
  for (int c = 0; c < 4; ++c) ob[i * 4 + c] = xyxy[q * 4 + c];
  os[i] = score[q]; oc[i] = cls[q];
}
// ---- the same glue as three launches on caller-owned scratch, no host round trip (nq <= 2048: the sort fits one workgroup) ----
struct SpanArgs { int se[2 * kGdinoPostSpansByValue]; };

// One workgroup: phrase scores of every query, (~score | query) keys, bitonic sort in LDS, boxes / scores / class indices gathered
// in decreasing-score order (thresholded-out queries sort behind the valid ones and are dropped), *nvalid = number of valid queries.
__global__ __launch_bounds__(1024) void gdino_post_sort_kernel(const float* __restrict__ logits, int nq, int ld,
                                                               const float* __restrict__ cxcywh, const int* __restrict__ spans_dev,
                                                               SpanArgs sa, int K, float img_h, float img_w, float thr, int N,
                                                               float* __restrict__ sbox, float* __restrict__ sscore,
                                                               int* __restrict__ scls, int* __restrict__ nvalid) {
  __shared__ unsigned long long keys[2048];
  __shared__ float sc[2048];
  __shared__ int cl[2048];
  const int t = threadIdx.x;
  for (int q = t; q < N; q += 1024) {
    unsigned long long key = kInvalidKey;
    if (q < nq) {
      const float* l = logits + (size_t)q * ld;
      float best = -INFINITY; int arg = 0;
      for (int k = 0; k < K; ++k) {
        const int b0 = spans_dev ? spans_dev[2 * k] : sa.se[2 * k], b1 = spans_dev ? spans_dev[2 * k + 1] : sa.se[2 * k + 1];
        float sum = 0.f;
        for (int u = b0; u < b1; ++u) sum += 1.0f / (1.0f + expf(-l[u]));
        if (sum > best) { best = sum; arg = k; }          // torch.max returns the first maximal index
      }
      sc[q] = best; cl[q] = arg;
      if (K > 0 && best > thr) key = (((unsigned long long)(~ord32(best))) << 24) | (unsigned long long)q;
    }
    keys[q] = key;
  }
  __syncthreads();
  for (int k = 2; k <= N; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int x = t; x < (N >> 1); x += 1024) {
        const int i = ((x / j) * 2 * j) + (x % j), l2 = i + j;
        const bool asc = ((i & k) == 0);
        const unsigned long long a = keys[i], b = keys[l2];
        if ((a > b) == asc) { keys[i] = b; keys[l2] = a; }
      }
      __syncthreads();
    }
  if (t == 0 && keys[0] == kInvalidKey) *nvalid = 0;
  for (int pos = t; pos < N; pos += 1024) {
    const unsigned long long key = keys[pos];
    if (key == kInvalidKey) continue;
    const int q = (int)(key & kIdMask);
    // boxes * [w, h, w, h], then cxcywh -> xyxy, as separate roundings (roi_heads_gdino.py:253 and box_ops.box_cxcywh_to_xyxy)
    const float cx = __fmul_rn(cxcywh[q * 4], img_w), cy = __fmul_rn(cxcywh[q * 4 + 1], img_h);
    const float w = __fmul_rn(cxcywh[q * 4 + 2], img_w), h = __fmul_rn(cxcywh[q * 4 + 3], img_h);
    sbox[pos * 4 + 0] = __fsub_rn(cx, __fmul_rn(0.5f, w)); sbox[pos * 4 + 1] = __fsub_rn(cy, __fmul_rn(0.5f, h));
    sbox[pos * 4 + 2] = __fadd_rn(cx, __fmul_rn(0.5f, w)); sbox[pos * 4 + 3] = __fadd_rn(cy, __fmul_rn(0.5f, h));
    sscore[pos] = sc[q]; scls[pos] = cl[q];
    if (pos + 1 == N || keys[pos + 1] == kInvalidKey) *nvalid = pos + 1;
  }
}

// mask[i*W + w] bit t  <=>  sorted candidate (w*64 + t) > i overlaps candidate i with IoU > thr; rows i < *nvalid only
__global__ void nms_mask1_kernel(const float* __restrict__ sbox, const int* __restrict__ nvalid, int W, float thr,
                                 unsigned long long* __restrict__ mask) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int n = *nvalid;
  const int i = idx / W, w = idx - i * W;
  if (i >= n) return;
  unsigned long long bits = 0ull;
  const int j0 = w * 64;
  if (j0 + 63 > i && j0 < n) {
    const float me[4] = {sbox[i * 4], sbox[i * 4 + 1], sbox[i * 4 + 2], sbox[i * 4 + 3]};
    for (int u = 0; u < 64; ++u) {
      const int j = j0 + u;
      if (j > i && j < n && iou_gt(me, sbox + (size_t)j * 4, thr)) bits |= (1ull << u);
    }
  }
  mask[(size_t)i * W + w] = bits;
}

// One workgroup: greedy pass over the sorted candidates in blocks of 64 - wave 0 settles a block from its 64 x 64 diagonal words
// with scalar bit operations (only surviving candidates cost an iteration), all 16 waves OR the kept rows into the removed words
// of the later blocks - then the kept candidates are emitted in order (boxes, scores, class indices, count).
__global__ __launch_bounds__(1024) void nms_resolve_emit_kernel(const unsigned long long* __restrict__ mask, const int* __restrict__ nvalid,
                                                                int W, const float* __restrict__ sbox, const float* __restrict__ sscore,
                                                                const int* __restrict__ scls, float* __restrict__ ob,
                                                                float* __restrict__ os, int* __restrict__ oc, int* __restrict__ n_out) {
  __shared__ unsigned long long part[16][64];
  __shared__ unsigned long long keptw[64];
  __shared__ int pre[64];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int n = *nvalid, nb = (n + 63) >> 6;
  unsigned long long remv = 0ull;                                   // wave 0: lane w owns the removed word of block w
  for (int b = 0; b < nb; ++b) {
    if (wave == 0) {
      const int row = 64 * b + lane;
      const unsigned long long diag = row < n ? mask[(size_t)row * W + b] : 0ull;
      const unsigned dlo = (unsigned)diag, dhi = (unsigned)(diag >> 32);
      const unsigned long long rw = __shfl(remv, b, 64);
      unsigned long long alive = ~rw;
      if (n - 64 * b < 64) alive &= (1ull << (n - 64 * b)) - 1ull;
      unsigned alo = __builtin_amdgcn_readfirstlane((unsigned)alive), ahi = __builtin_amdgcn_readfirstlane((unsigned)(alive >> 32));
      unsigned klo = 0u, khi = 0u;
      while (alo | ahi) {
        const int u = alo ? (__builtin_ffs((int)alo) - 1) : (32 + __builtin_ffs((int)ahi) - 1);
        if (u < 32) klo |= 1u << u; else khi |= 1u << (u - 32);
        const unsigned slo = __builtin_amdgcn_readlane(dlo, u), shi = __builtin_amdgcn_readlane(dhi, u);
        alo &= ~slo; ahi &= ~shi;
        if (u < 32) alo &= ~(1u << u); else ahi &= ~(1u << (u - 32));
      }
      if (lane == 0) keptw[b] = ((unsigned long long)khi << 32) | klo;
    }
    __syncthreads();
    const unsigned long long kept = keptw[b];
    unsigned long long acc = 0ull;
    if (lane < W && lane > b)
      for (int r = wave; r < 64; r += 16)
        if ((kept >> r) & 1ull) acc |= mask[(size_t)(64 * b + r) * W + lane];
    part[wave][lane] = acc;
    __syncthreads();
    if (wave == 0) {
#pragma unroll
      for (int v = 0; v < 16; ++v) remv |= part[v][lane];
    }
  }
  __syncthreads();
  if (t == 0) {
    int run = 0;
    for (int b = 0; b < nb; ++b) { pre[b] = run; run += __popcll(keptw[b]); }
    *n_out = run;
  }
  __syncthreads();
  for (int pos = t; pos < n; pos += 1024) {
    const unsigned long long kw = keptw[pos >> 6];
    const int l = pos & 63;
    if (!((kw >> l) & 1ull)) continue;
    const int o = pre[pos >> 6] + __popcll(kw & ((1ull << l) - 1ull));
#pragma unroll
    for (int c = 0; c < 4; ++c) ob[o * 4 + c] = sbox[pos * 4 + c];
    os[o] = sscore[pos]; oc[o] = scls[pos];
  }
}
}  // namespace

// scratch layout: sbox [N][4] f32 | sscore [N] f32 | scls [N] i32 | nvalid (16 B) | spans [2*K] i32 (only if K > by-value limit) | mask [nq][W] u64
size_t gdino_post_ws_bytes(int nq, int K) {
  if (nq <= 0 || nq > 2048) return 0;
  const size_t N = nq <= 1024 ? 1024 : 2048, W = ((size_t)nq + 63) / 64;
  const size_t sp = K > kGdinoPostSpansByValue ? (((size_t)2 * K * 4 + 15) & ~(size_t)15) : 0;
  return N * 16 + N * 4 + N * 4 + 16 + sp + (size_t)nq * W * 8;
}

int launch_gdino_post_ws(const float* logits, int nq, int ld, const float* cxcywh, const int* spans, int K, int img_h, int img_w,
                         float box_thr, float nms_thr, void* ws, size_t ws_bytes, float* out_boxes, float* out_scores, int* out_classes,
                         int* n_out, hipStream_t s) {
  if (nq == 0) return hipMemsetAsync(n_out, 0, sizeof(int), s) == hipSuccess ? OVM_OK : OVM_ERR_HIP;
  if (nq < 0 || nq > 2048 || K < 0 || !ws || ws_bytes < gdino_post_ws_bytes(nq, K)) return OVM_ERR_CAPACITY;
  const int N = nq <= 1024 ? 1024 : 2048, W = (nq + 63) / 64;
  char* p = (char*)ws;
  float* sbox = (float*)p; p += (size_t)N * 16;
  float* sscore = (float*)p; p += (size_t)N * 4;
  int* scls = (int*)p; p += (size_t)N * 4;
  int* nvalid = (int*)p; p += 16;
  int* dsp = nullptr;
  SpanArgs sa; memset(&sa, 0, sizeof(sa));
  if (K > kGdinoPostSpansByValue) {
    dsp = (int*)p; p += ((size_t)2 * K * 4 + 15) & ~(size_t)15;
    if (hipMemcpyAsync(dsp, spans, sizeof(int) * 2 * K, hipMemcpyHostToDevice, s) != hipSuccess) return OVM_ERR_HIP;
  } else if (K > 0) {
    memcpy(sa.se, spans, sizeof(int) * 2 * K);
  }
  unsigned long long* mask = (unsigned long long*)p;
  hipLaunchKernelGGL(gdino_post_sort_kernel, dim3(1), dim3(1024), 0, s, logits, nq, ld, cxcywh, (const int*)dsp, sa, K, (float)img_h,
                     (float)img_w, box_thr, N, sbox, sscore, scls, nvalid);
  hipLaunchKernelGGL(nms_mask1_kernel, dim3((unsigned)((nq * W + 127) / 128)), dim3(128), 0, s, sbox, nvalid, W, nms_thr, mask);
  hipLaunchKernelGGL(nms_resolve_emit_kernel, dim3(1), dim3(1024), 0, s, mask, nvalid, W, sbox, sscore, scls, out_boxes, out_scores,
                     out_classes, n_out);
  return hipGetLastError() == hipSuccess ? OVM_OK : OVM_ERR_HIP;
}

int launch_gdino_post(const float* logits, int nq, int ld, const float* cxcywh, const int* spans, int K, int img_h, int img_w,
                      float box_thr, float nms_thr, float* out_boxes, float* out_scores, int* out_classes, int* n_out, hipStream_t s) {
  if (nq == 0) return hipMemsetAsync(n_out, 0, sizeof(int), s) == hipSuccess ? OVM_OK : OVM_ERR_HIP;
  if (nq < 0 || nq > 4096 || K < 0) return OVM_ERR_CAPACITY;
  if (nq <= 2048) {                        // three launches on one scratch block
    const size_t bytes = gdino_post_ws_bytes(nq, K);
    void* ws = nullptr;
    if (hipMalloc(&ws, bytes) != hipSuccess) return OVM_ERR_HIP;
    int r = launch_gdino_post_ws(logits, nq, ld, cxcywh, spans, K, img_h, img_w, box_thr, nms_thr, ws, bytes, out_boxes, out_scores,
                                 out_classes, n_out, s);
    if (hipStreamSynchronize(s) != hipSuccess && !r) r = OVM_ERR_HIP;
    (void)hipFree(ws);
    return r;
  }
  std::vector<void*> tmp;
  float *xyxy = nullptr, *score = nullptr; int *cls = nullptr, *valid = nullptr, *keep = nullptr, *dsp = nullptr;
  int r;
  if ((r = dmalloc(&xyxy, (size_t)nq * 4, &tmp)) || (r = dmalloc(&score, (size_t)nq, &tmp)) || (r = dmalloc(&cls, (size_t)nq, &tmp)) ||
      (r = dmalloc(&valid, (size_t)nq, &tmp)) || (r = dmalloc(&keep, (size_t)nq, &tmp)) || (r = dmalloc(&dsp, (size_t)(2 * K + 2), &tmp))) {
    for (void* p : tmp) (void)hipFree(p);
    return r;
  }
  if (K > 0 && hipMemcpyAsync(dsp, spans, sizeof(int) * 2 * K, hipMemcpyHostToDevice, s) != hipSuccess) r = OVM_ERR_HIP;
  if (!r) {
    hipLaunchKernelGGL(gdino_phrase_kernel, dim3((nq + 127) / 128), dim3(128), 0, s, logits, nq, ld, cxcywh, dsp, K, (float)img_h,
                       (float)img_w, box_thr, xyxy, score, cls, valid);
    r = launch_nms_single(xyxy, score, valid, nq, nms_thr, keep, n_out, s);      // synchronises
    if (!r) {
      hipLaunchKernelGGL(gdino_gather_kernel, dim3((nq + 127) / 128), dim3(128), 0, s, keep, n_out, xyxy, score, cls, out_boxes,
                         out_scores, out_classes);
      if (hipStreamSynchronize(s) != hipSuccess) r = OVM_ERR_HIP;
    }
  }
  for (void* p : tmp) (void)hipFree(p);
  return r;
}

// top-k indices by decreasing score (ties: lower index first), n <= 2^24
namespace {
__global__ void topk_emit_kernel(const unsigned long long* __restrict__ keys, int k, int* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < k) out[i] = (int)(keys[i] & kIdMask);
}
}  // namespace
int launch_topk_keys(const float* scores, int n, int k, int* out_idx, unsigned long long* keys, int N, hipStream_t s) {
  if (k > n || n <= 0 || N < pow2_at_least(n)) return OVM_ERR_INVALID;
  N = pow2_at_least(n);
  hipLaunchKernelGGL(single_keys_kernel, dim3((N + 255) / 256), dim3(256), 0, s, scores, (const int*)nullptr, n, N, keys);
  int r = sort_keys(keys, N, 1, s);
  if (r) return r;
  hipLaunchKernelGGL(topk_emit_kernel, dim3((k + 255) / 256), dim3(256), 0, s, keys, k, out_idx);
  return hipGetLastError() == hipSuccess ? OVM_OK : OVM_ERR_HIP;
}

int launch_topk(const float* scores, int n, int k, int* out_idx, hipStream_t s) {
  if (k > n || n <= 0) return OVM_ERR_INVALID;
  const int N = pow2_at_least(n);
  static unsigned long long* keys = nullptr; static int cap = 0;
  if (cap < N) { if (keys) { (void)hipDeviceSynchronize(); (void)hipFree(keys); } if (hipMalloc((void**)&keys, sizeof(unsigned long long) * N) != hipSuccess) { keys = nullptr; cap = 0; return OVM_ERR_HIP; } cap = N; }
  hipLaunchKernelGGL(single_keys_kernel, dim3((N + 255) / 256), dim3(256), 0, s, scores, (const int*)nullptr, n, N, keys);
  int r = sort_keys(keys, N, 1, s);
  if (r) return r;
  hipLaunchKernelGGL(topk_emit_kernel, dim3((k + 255) / 256), dim3(256), 0, s, keys, k, out_idx);
  return hipGetLastError() == hipSuccess ? OVM_OK : OVM_ERR_HIP;
}

}  // namespace ovm
