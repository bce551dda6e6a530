// Generic device ops used by the GroundingDINO branch (scope row a10): the network is sequenced by the Python
// host (ovmono3d_amd/gdino/), one C-ABI call per op; every arithmetic step runs here.
// Tensors are fp32 row-major in HBM; dense projections go through the split-fp16 MFMA GEMM of gemm.hpp.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstring>
#include <vector>
#include "../../include/ovm3d.h"
#include "kernels.hpp"
#include "det2d.hpp"

using namespace ovm;

namespace {

// ---- scratch (grown on demand; one stream / one thread per process, like the rest of the library) ----
struct Scratch { void* p = nullptr; size_t cap = 0; };
Scratch g_s[2];
long g_small_max_tiles = -1;      // -1: built-in heuristic
int g_bmm_tiled = 1;
void* scratch(int i, size_t bytes) {
  if (g_s[i].cap < bytes) {
    if (g_s[i].p) { (void)hipDeviceSynchronize(); (void)hipFree(g_s[i].p); }
    size_t cap = bytes + bytes / 4 + (1 << 20);
    if (hipMalloc(&g_s[i].p, cap) != hipSuccess) { g_s[i].p = nullptr; g_s[i].cap = 0; return nullptr; }
    g_s[i].cap = cap;
  }
  return g_s[i].p;
}

// x [M][K] fp32 (row stride ldx) -> split fp16 [M][Kpad], zero in the K padding
__global__ void split_pad_kernel(const float* __restrict__ x, int ldx, int M, int K, int Kpad, half_t* __restrict__ hi, half_t* __restrict__ lo) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)M * Kpad) return;
  const int k = (int)(i % Kpad); const size_t m = i / Kpad;
  float v = (k < K) ? x[m * ldx + k] : 0.f;
  half_t h, l; split_f16(v, h, l);
  hi[i] = h;
  if (lo) lo[i] = l;
}

// lo == nullptr: one-pass image [Npad][Kpad]; else the interleaved split image [Npad][Kpad/32][hi 32 | lo 32] (lo = hi + 32)
__global__ void pack_weight_kernel(const float* __restrict__ w, int N, int K, int Npad, int Kpad, half_t* __restrict__ hi, half_t* __restrict__ lo) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)Npad * Kpad) return;
  const int k = (int)(i % Kpad); const int n = (int)(i / Kpad);
  float v = (n < N && k < K) ? w[(size_t)n * K + k] : 0.f;
  half_t h, l; split_f16(v, h, l);
  if (lo) {
    const size_t o = (size_t)n * 2 * Kpad + (size_t)(k >> 5) * 64 + (k & 31);
    hi[o] = h; hi[o + 32] = l;
  } else {
    hi[i] = h;
  }
}

// y = LayerNorm(x (+ residual)) over the last dim; one wave per row, any D (looped)
__global__ __launch_bounds__(256) void ln_generic_kernel(const float* __restrict__ x, const float* __restrict__ res, int M, int D,
                                                         const float* __restrict__ g, const float* __restrict__ b, float eps,
                                                         float* __restrict__ y) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= M) return;
  const float* xr = x + (size_t)row * D;
  const float* rr = res ? res + (size_t)row * D : nullptr;
  float s = 0.f;
  for (int j = lane; j < D; j += 64) s += xr[j] + (rr ? rr[j] : 0.f);
  const float mean = wave_sum(s) / (float)D;
  float q = 0.f;
  for (int j = lane; j < D; j += 64) { const float d = xr[j] + (rr ? rr[j] : 0.f) - mean; q += d * d; }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)D + eps);
  for (int j = lane; j < D; j += 64) y[(size_t)row * D + j] = (xr[j] + (rr ? rr[j] : 0.f) - mean) * rstd * g[j] + b[j];
}

// c[b] = alpha * a[b] (M x K) * op(b[b]) ; op = transpose when transB (b is N x K), else b is K x N. fp32 FMA, 16x16 tiles.
// batch index z = z1 * nb2 + z2 with independent strides per level (e.g. window x head)
__global__ __launch_bounds__(256) void bmm_kernel(const float* __restrict__ A, const float* __restrict__ Bm, float* __restrict__ Cm,
                                                  int M, int N, int K, int lda, int ldb, int ldc, long sA, long sB, long sC,
                                                  int nb2, long sA2, long sB2, long sC2, int transB, float alpha, int tiles_n, int kchunk,
                                                  float* __restrict__ part) {
  __shared__ float As[16][17], Bs[16][17];
  const int ks = blockIdx.x / tiles_n, bx = blockIdx.x - ks * tiles_n;            // split-K slice (kchunk % 16 == 0)
  const int kbeg = ks * kchunk, kend = (kbeg + kchunk < K) ? kbeg + kchunk : K;
  const int z1 = blockIdx.z / nb2, z2 = blockIdx.z - z1 * nb2;
  const float* a = A + (size_t)z1 * sA + (size_t)z2 * sA2; const float* b = Bm + (size_t)z1 * sB + (size_t)z2 * sB2;
  float* c = Cm + (size_t)z1 * sC + (size_t)z2 * sC2;
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int m = blockIdx.y * 16 + ty, n = bx * 16 + tx;
  float acc = 0.f;
  for (int k0 = kbeg; k0 < kend; k0 += 16) {
    As[ty][tx] = (m < M && k0 + tx < kend) ? a[(size_t)m * lda + k0 + tx] : 0.f;
    const int nn = bx * 16 + ty;                                              // for the transposed load
    if (transB) Bs[tx][ty] = (nn < N && k0 + tx < kend) ? b[(size_t)nn * ldb + k0 + tx] : 0.f;   // Bs[k][n]
    else Bs[ty][tx] = (k0 + ty < kend && n < N) ? b[(size_t)(k0 + ty) * ldb + n] : 0.f;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) acc = fmaf(As[ty][k], Bs[k][tx], acc);
    __syncthreads();
  }
  if (m < M && n < N) {
    if (part) part[(((size_t)ks * gridDim.z + blockIdx.z) * M + m) * N + n] = acc;
    else c[(size_t)m * ldc + n] = alpha * acc;
  }
}


// Register-blocked variant for M >= 48: a workgroup computes a (16*TM) x (16*TN) tile, each thread TM x TN outputs (rows
// ty + 16 i, columns tx + 16 j: stores coalesced along tx), k-steps of 16 through LDS. Same fp32 FMA order per output as
// bmm_kernel (k ascending), so the two produce identical results.
template <int TM, int TN>
__global__ __launch_bounds__(256) void bmm_tile_kernel(const float* __restrict__ A, const float* __restrict__ Bm, float* __restrict__ Cm, int M,
                                                       int N, int K, int lda, int ldb, int ldc, long sA, long sB, long sC, int nb2, long sA2,
                                                       long sB2, long sC2, int transB, float alpha) {
  __shared__ float As[16][16 * TM + 1], Bs[16][16 * TN + 1];
  const int z1 = blockIdx.z / nb2, z2 = blockIdx.z - z1 * nb2;
  const float* a = A + (size_t)z1 * sA + (size_t)z2 * sA2; const float* b = Bm + (size_t)z1 * sB + (size_t)z2 * sB2;
  float* c = Cm + (size_t)z1 * sC + (size_t)z2 * sC2;
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int m0 = blockIdx.y * 16 * TM, n0 = blockIdx.x * 16 * TN;
  float acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = 0.f;
  for (int k0 = 0; k0 < K; k0 += 16) {
#pragma unroll
    for (int i = 0; i < TM; ++i) {                               // A rows (k contiguous): thread (row ty + 16 i, k tx)
      const int m = m0 + ty + 16 * i;
      As[tx][ty + 16 * i] = (m < M && k0 + tx < K) ? a[(size_t)m * lda + k0 + tx] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      if (transB) {                                              // B is N x K: thread (row ty + 16 j, k tx)
        const int n = n0 + ty + 16 * j;
        Bs[tx][ty + 16 * j] = (n < N && k0 + tx < K) ? b[(size_t)n * ldb + k0 + tx] : 0.f;
      } else {                                                   // B is K x N: thread (k ty, column tx + 16 j)
        const int n = n0 + tx + 16 * j;
        Bs[ty][tx + 16 * j] = (k0 + ty < K && n < N) ? b[(size_t)(k0 + ty) * ldb + n] : 0.f;
      }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      float av[TM], bv[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) av[i] = As[k][ty + 16 * i];
#pragma unroll
      for (int j = 0; j < TN; ++j) bv[j] = Bs[k][tx + 16 * j];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = fmaf(av[i], bv[j], acc[i][j]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int m = m0 + ty + 16 * i;
    if (m >= M) continue;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + tx + 16 * j;
      if (n < N) c[(size_t)m * ldc + n] = alpha * acc[i][j];
    }
  }
}

// sums the split-K partials of bmm_kernel: part [ksplit][nz][M][N] -> c (batch strides as in bmm_kernel)
__global__ void bmm_reduce_kernel(const float* __restrict__ part, float* __restrict__ Cm, int ksplit, int nz, int M, int N, int ldc, long sC,
                                  int nb2, long sC2, float alpha) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)nz * M * N) return;
  const int n = (int)(i % N); const int m = (int)((i / N) % M); const int z = (int)(i / ((long)M * N));
  float acc = 0.f;
  for (int ks = 0; ks < ksplit; ++ks) acc += part[(size_t)ks * nz * M * N + i];
  const int z1 = z / nb2, z2 = z - z1 * nb2;
  Cm[(size_t)z1 * sC + (size_t)z2 * sC2 + (size_t)m * ldc + n] = alpha * acc;
}

// x[r][:] = softmax(x[r][:] + bias[(r / bias_div) % bias_rows][:]) ; one wave per row
// optional second additive term bias2[(row / d2) * m2 + row % m2][:]  (e.g. the per-window shift mask next to the per-head
// relative-position bias of Swin attention)
__global__ __launch_bounds__(256) void softmax_kernel(float* __restrict__ x, int rows, int cols, int ld, const float* __restrict__ bias,
                                                      int bias_rows, int bias_div, int bias_ld, const float* __restrict__ bias2, int d2,
                                                      int m2) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  float* xr = x + (size_t)row * ld;
  const float* br = bias ? bias + (size_t)((row / bias_div) % bias_rows) * bias_ld : nullptr;
  const float* b2 = bias2 ? bias2 + ((size_t)(row / d2) * m2 + (row % m2)) * bias_ld : nullptr;
  if (cols <= 512) {
    // short rows (Swin windows of 144 keys, 16 text tokens): keep the row in registers - one read of x and of the biases, one
    // write; same operation order as the generic path below (bit-identical results)
    float v[8];
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int j = lane + 64 * t;
      v[t] = (j < cols) ? xr[j] + (br ? br[j] : 0.f) + (b2 ? b2[j] : 0.f) : -INFINITY;
      mx = fmaxf(mx, v[t]);
    }
    mx = wave_max(mx);
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int j = lane + 64 * t;
      if (j < cols) { v[t] = expf(v[t] - mx); s += v[t]; }
    }
    s = wave_sum(s);
    const float inv = 1.0f / s;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int j = lane + 64 * t;
      if (j < cols) xr[j] = v[t] * inv;
    }
    return;
  }
  float mx = -INFINITY;
  for (int j = lane; j < cols; j += 64) mx = fmaxf(mx, xr[j] + (br ? br[j] : 0.f) + (b2 ? b2[j] : 0.f));
  mx = wave_max(mx);
  float s = 0.f;
  for (int j = lane; j < cols; j += 64) { const float e = expf(xr[j] + (br ? br[j] : 0.f) + (b2 ? b2[j] : 0.f) - mx); xr[j] = e; s += e; }
  s = wave_sum(s);
  const float inv = 1.0f / s;
  for (int j = lane; j < cols; j += 64) xr[j] *= inv;
}

enum { E_ADD = 0, E_MUL = 1, E_RELU = 2, E_GELU = 3, E_SIGMOID = 4, E_CLAMP = 5, E_AXPY = 6, E_INVSIG = 7, E_COPY = 8, E_MASKFILL = 9 };
// out[i] = f(a[i], b[i % bmod]) ; bmod = n for a full tensor, = cols for a row vector
__global__ void eltwise_kernel(int op, const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, long n, long bmod,
                               float alpha, float beta) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float x = a[i];
  const float y = b ? b[i % bmod] : 0.f;
  float r;
  switch (op) {
    case E_ADD: r = x + y; break;
    case E_MUL: r = x * y; break;
    case E_RELU: r = fmaxf(x, 0.f); break;
    case E_GELU: r = gelu_erf(x); break;
    case E_SIGMOID: r = 1.0f / (1.0f + expf(-x)); break;
    case E_CLAMP: r = fminf(fmaxf(x, alpha), beta); break;
    case E_AXPY: r = x + alpha * y; break;
    case E_INVSIG: { const float xc = fminf(fmaxf(x, alpha), 1.f - alpha); r = logf(xc / (1.f - xc)); break; }   // torch.special.logit(x, eps)
    case E_MASKFILL: r = (y != 0.f) ? alpha : x; break;
    default: r = x;
  }
  out[i] = r;
}

// out[r] = max_j x[r][j] ; one wave per row
__global__ __launch_bounds__(256) void rowmax_kernel(const float* __restrict__ x, int rows, int cols, int ld, float* __restrict__ out) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= rows) return;
  float m = -INFINITY;
  for (int j = lane; j < cols; j += 64) m = fmaxf(m, x[(size_t)row * ld + j]);
  m = wave_max(m);
  if (lane == 0) out[row] = m;
}

// dst[i][j*cols .. (j+1)*cols) = idx[i*nidx+j] >= 0 ? src[idx][0..cols) : 0
__global__ void gather_rows_kernel(const float* __restrict__ src, int ld_src, const int* __restrict__ idx, long n_out, int nidx, int cols,
                                   float* __restrict__ dst) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = n_out * nidx * cols;
  if (t >= total) return;
  const int c = (int)(t % cols); const long ij = t / cols;
  const int id = idx[ij];
  dst[t] = id >= 0 ? src[(size_t)id * ld_src + c] : 0.f;
}

// GroupNorm over NHWC-flattened x [B][HW][C]: statistics per (batch, group) over HW x (C/groups)
__global__ __launch_bounds__(256) void groupnorm_kernel(const float* __restrict__ x, int HW, int Cc, int groups, const float* __restrict__ g,
                                                        const float* __restrict__ b, float eps, float* __restrict__ y) {
  __shared__ float red[8];
  const int grp = blockIdx.x, bz = blockIdx.y, cpg = Cc / groups, n = HW * cpg;
  const float* xb = x + (size_t)bz * HW * Cc; float* yb = y + (size_t)bz * HW * Cc;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float s = 0.f;
  for (int i = tid; i < n; i += 256) s += xb[(size_t)(i / cpg) * Cc + grp * cpg + (i % cpg)];
  s = wave_sum(s); if (lane == 0) red[wave] = s; __syncthreads();
  const float mean = (red[0] + red[1] + red[2] + red[3]) / (float)n;
  float q = 0.f;
  for (int i = tid; i < n; i += 256) { const float d = xb[(size_t)(i / cpg) * Cc + grp * cpg + (i % cpg)] - mean; q += d * d; }
  q = wave_sum(q); if (lane == 0) red[4 + wave] = q; __syncthreads();
  const float rstd = 1.0f / sqrtf((red[4] + red[5] + red[6] + red[7]) / (float)n + eps);
  for (int i = tid; i < n; i += 256) {
    const int c = grp * cpg + (i % cpg); const size_t o = (size_t)(i / cpg) * Cc + c;
    yb[o] = (xb[o] - mean) * rstd * g[c] + b[c];
  }
}

// The same statistics with the group's slice held in registers: the neck's largest level (67 x 67 x 256 at a 532 x 532 image, 32 groups
// of 8 channels) is 8,978 float4 per group - one workgroup of 1,024 threads reads each once (9 independent 16-byte loads per thread in
// flight instead of three dependent strided passes of 140 iterations: 98.6 -> ~10 us per launch in the detector's trace), then mean,
// centred sum of squares and the normalisation run on registers. Same two-pass formula; the order of the partial sums differs.
constexpr int kGnMaxV = 12;
__global__ __launch_bounds__(1024) void groupnorm_reg_kernel(const float* __restrict__ x, int HW, int Cc, int groups, const float* __restrict__ g,
                                                             const float* __restrict__ b, float eps, float* __restrict__ y) {
  __shared__ float red[32];
  const int grp = blockIdx.x, bz = blockIdx.y, cpg = Cc / groups, q4 = cpg >> 2, n4 = HW * q4;
  const float* xb = x + (size_t)bz * HW * Cc + (size_t)grp * cpg; float* yb = y + (size_t)bz * HW * Cc + (size_t)grp * cpg;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  f32x4 v[kGnMaxV];
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < kGnMaxV; ++j) {
    const int i = tid + j * 1024;
    v[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (i < n4) { v[j] = *(const f32x4*)(xb + (size_t)(i / q4) * Cc + (i % q4) * 4); s += (v[j][0] + v[j][1]) + (v[j][2] + v[j][3]); }
  }
  s = wave_sum(s); if (lane == 0) red[wave] = s; __syncthreads();
  float tot = 0.f;
#pragma unroll
  for (int w = 0; w < 16; ++w) tot += red[w];
  const float mean = tot / (float)(n4 * 4);
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < kGnMaxV; ++j)
    if (tid + j * 1024 < n4) {
#pragma unroll
      for (int r = 0; r < 4; ++r) { const float d = v[j][r] - mean; q += d * d; }
    }
  q = wave_sum(q); if (lane == 0) red[16 + wave] = q; __syncthreads();
  float qt = 0.f;
#pragma unroll
  for (int w = 0; w < 16; ++w) qt += red[16 + w];
  const float rstd = 1.0f / sqrtf(qt / (float)(n4 * 4) + eps);
#pragma unroll
  for (int j = 0; j < kGnMaxV; ++j) {
    const int i = tid + j * 1024;
    if (i < n4) {
      const int c = (i % q4) * 4;
      const f32x4 gg = *(const f32x4*)(g + grp * cpg + c), bb = *(const f32x4*)(b + grp * cpg + c);
      f32x4 o;
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = (v[j][r] - mean) * rstd * gg[r] + bb[r];
      *(f32x4*)(yb + (size_t)(i / q4) * Cc + c) = o;
    }
  }
}

// Multi-scale deformable attention sampling (Deformable-DETR; F.grid_sample bilinear, zeros padding, align_corners=False).
// value [B][S][H][dh]; loc [B][Q][H][L][P][2] in [0,1]; w [B][Q][H][L][P]; out [B][Q][H*dh]. One thread per (b,q,h,d).
struct MsdShapes { int h[8], w[8], start[8]; };
__global__ void msdeform_kernel(const float* __restrict__ value, MsdShapes sh, int B, int S, int Q, int H, int dh, int L, int P,
                                const float* __restrict__ loc, const float* __restrict__ w, float* __restrict__ out) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)B * Q * H * dh;
  if (t >= total) return;
  const int d = (int)(t % dh); long r = t / dh;
  const int hh = (int)(r % H); r /= H;
  const int q = (int)(r % Q); const int b = (int)(r / Q);
  const float* lp = loc + ((((size_t)b * Q + q) * H + hh) * L) * P * 2;
  const float* wp = w + ((((size_t)b * Q + q) * H + hh) * L) * P;
  float acc = 0.f;
  for (int l = 0; l < L; ++l) {
    const int Hh = sh.h[l], Ww = sh.w[l];
    const float* vb = value + ((size_t)b * S + sh.start[l]) * H * dh + (size_t)hh * dh + d;
    for (int p = 0; p < P; ++p) {
      const float gx = 2.f * lp[(l * P + p) * 2] - 1.f, gy = 2.f * lp[(l * P + p) * 2 + 1] - 1.f;
      const float ix = ((gx + 1.f) * (float)Ww - 1.f) * 0.5f, iy = ((gy + 1.f) * (float)Hh - 1.f) * 0.5f;
      const float fx = floorf(ix), fy = floorf(iy);
      const int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
      const float wx1 = ix - fx, wy1 = iy - fy, wx0 = 1.f - wx1, wy0 = 1.f - wy1;
      float v = 0.f;
      if (y0 >= 0 && y0 < Hh) {
        if (x0 >= 0 && x0 < Ww) v += wy0 * wx0 * vb[(size_t)(y0 * Ww + x0) * H * dh];
        if (x1 >= 0 && x1 < Ww) v += wy0 * wx1 * vb[(size_t)(y0 * Ww + x1) * H * dh];
      }
      if (y1 >= 0 && y1 < Hh) {
        if (x0 >= 0 && x0 < Ww) v += wy1 * wx0 * vb[(size_t)(y1 * Ww + x0) * H * dh];
        if (x1 >= 0 && x1 < Ww) v += wy1 * wx1 * vb[(size_t)(y1 * Ww + x1) * H * dh];
      }
      acc += v * wp[l * P + p];
    }
  }
  out[t] = acc;
}

// sinusoidal embedding of coordinates (DETR convention, x/y swapped): pos [n][nc] -> out [n][nc*F]
__global__ void sine_embed_kernel(const float* __restrict__ pos, long n, int nc, int F, float temperature, float* __restrict__ out) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n * nc * F) return;
  const int f = (int)(t % F); long r = t / F;
  const int slot = (int)(r % nc); const long i = r / nc;
  int c = slot;
  if (nc >= 2) { if (slot == 0) c = 1; else if (slot == 1) c = 0; }           // [pos_y, pos_x, ...]
  const float dim_t = powf(temperature, 2.f * (float)(f / 2) / (float)F);
  const float e = pos[i * nc + c] * 6.283185307179586f / dim_t;
  out[t] = (f & 1) ? cosf(e) : sinf(e);
}

// normalised float image of the reference's GDINO call: images[0][[2,1,0]] (reference roi_heads_gdino.py:146), NHWC out
__global__ void normalize_image_kernel(ImageDesc d, float m0, float m1, float m2, float s0, float s1, float s2, int flip, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= d.H * d.W * 3) return;
  const int c = i % 3; const int p = i / 3; const int x = p % d.W, y = p / d.W;
  const int cs = flip ? 2 - c : c;
  const float mean[3] = {m0, m1, m2}, sd[3] = {s0, s1, s2};
  out[i] = ((float)d.data[(size_t)cs * d.sC + (size_t)y * d.sH + (size_t)x * d.sW] - mean[cs]) / sd[cs];
}

inline dim3 g1(long n, int bs = 256) { return dim3((unsigned)((n + bs - 1) / bs)); }

}  // namespace

namespace ovm { void glinear_set_small_max_tiles(int t) { g_small_max_tiles = t; } void gbmm_set_tiled(int v) { g_bmm_tiled = v; } }

extern "C" {

int ovm_g_pack_weight(const float* w, int32_t N, int32_t K, int32_t Kpad, uint16_t* hi, uint16_t* lo, ovm_stream_t stream) {
  const int Npad = (N + 127) / 128 * 128;
  if (lo && (lo != hi + 32 || Kpad % 32 != 0)) return OVM_ERR_INVALID;      // split image: lo interleaved 32 halves after hi
  hipLaunchKernelGGL(pack_weight_kernel, g1((long)Npad * Kpad), dim3(256), 0, (hipStream_t)stream, w, N, K, Npad, Kpad, (half_t*)hi, (half_t*)lo);
  return hipGetLastError() == hipSuccess ? OVM_OK : OVM_ERR_HIP;
}

// y[M][N] (row stride ldy) = act(x[M][K] W^T + bias) (+ residual). W packed by ovm_g_pack_weight (split mode: interleaved image, w_lo = w_hi + 32).
int ovm_g_linear(const float* x, int32_t ldx, int32_t M, int32_t K, const uint16_t* w_hi, const uint16_t* w_lo, int32_t N, int32_t Kpad,
                 const float* bias, int32_t act, const float* residual, int32_t ldr, float* y, int32_t ldy, int32_t precision,
                 ovm_stream_t stream) {
  hipStream_t s = (hipStream_t)stream;
  if (M <= 0) return OVM_OK;
  // small / mid-size problems: fp32-A 64x64-tile kernel (splits x in registers, split-K on thin grids); very large grids keep
  // the 128x128 LDS-DMA kernel behind a split pre-pass
  const long tiles128 = (long)((M + 127) / 128) * ((N + 127) / 128);
  const long small_max = g_small_max_tiles >= 0 ? g_small_max_tiles : (K <= 512 ? 100 : (K >= 2048 ? 31 : 64));     // measured crossovers (scratch/bench_linear.py); long K: the MFMA kernel's split-K wins earlier
  if (tiles128 <= small_max && gemm_small_supported(x, ldx, K))
    return launch_gemm_small(x, ldx, M, K, (const half_t*)w_hi, (const half_t*)w_lo, N, Kpad, bias, act, residual, ldr, y, ldy, precision, s);
  const size_t ne = (size_t)M * Kpad;
  half_t* ahi = (half_t*)scratch(0, ne * 2);
  half_t* alo = (half_t*)scratch(1, ne * 2);
  if (!ahi || !alo) return OVM_ERR_HIP;
  hipLaunchKernelGGL(split_pad_kernel, g1((long)ne), dim3(256), 0, s, x, ldx, M, K, Kpad, ahi, precision == 3 ? alo : nullptr);
  GemmParams p; memset(&p, 0, sizeof(p));
  p.Ahi = ahi; p.Alo = alo; p.lda = Kpad; p.Whi = (const half_t*)w_hi; p.Wlo = (const half_t*)w_lo;
  p.M = M; p.N = N; p.K = Kpad; p.bias = bias; p.relu = act; p.R = residual; p.ldr = ldr; p.C = y; p.ldc = ldy;
  p.ws_slot = 1;                                             // this branch may run beside the engine's GEMMs on another stream
  return launch_gemm(p, precision, EPI_STORE, A_ROWMAJOR, s);
}

int ovm_g_layernorm(const float* x, const float* residual, int32_t M, int32_t D, const float* gamma, const float* beta, float eps, float* y,
                    ovm_stream_t stream) {
  if (M <= 0) return OVM_OK;
  hipLaunchKernelGGL(ln_generic_kernel, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, residual, M, D, gamma, beta, eps, y);
  return hipGetLastError() == hipSuccess ? OVM_OK : OVM_ERR_HIP;
}

int ovm_g_bmm(const float* a, const float* b, float* c, int32_t batch, int32_t M, int32_t N, int32_t K, int32_t lda, int32_t ldb, int32_t ldc,
              int64_t sA, int64_t sB, int64_t sC, int32_t transB, float alpha, ovm_stream_t stream) {
  return ovm_g_bmm2(a, b, c, batch, 1, M, N, K, lda, ldb, ldc, sA, sB, sC, 0, 0, 0, transB, alpha, stream);
}

int ovm_g_bmm2(const float* a, const float* b, float* c, int32_t nb1, int32_t nb2, int32_t M, int32_t N, int32_t K, int32_t lda, int32_t ldb,
               int32_t ldc, int64_t sA1, int64_t sB1, int64_t sC1, int64_t sA2, int64_t sB2, int64_t sC2, int32_t transB, float alpha,
               ovm_stream_t stream) {
  if (nb1 <= 0 || nb2 <= 0 || M <= 0 || N <= 0) return OVM_OK;
  const long nz = (long)nb1 * nb2;
  const int tiles_n = (N + 15) / 16;
  // long reductions on a thin grid (text <- image attention: 16 x 6015 per head): split K over workgroups
  const long blocks = (long)tiles_n * ((M + 15) / 16) * nz;
  if (blocks < 512 && K >= 1024 && nz <= 65535) {
    int ksplit = (int)((1024 + blocks - 1) / blocks);
    if (ksplit > K / 128) ksplit = K / 128;
    if (ksplit > 64) ksplit = 64;
    if (ksplit > 1) {
      const int kchunk = ((K + ksplit - 1) / ksplit + 15) / 16 * 16;
      ksplit = (K + kchunk - 1) / kchunk;
      float* part = (float*)scratch(0, (size_t)ksplit * nz * M * N * sizeof(float));
      if (!part) return OVM_ERR_HIP;
      hipLaunchKernelGGL(bmm_kernel, dim3(tiles_n * ksplit, (M + 15) / 16, (unsigned)nz), dim3(256), 0, (hipStream_t)stream, a, b, c, M, N, K,
                         lda, ldb, ldc, (long)sA1, (long)sB1, (long)sC1, nb2, (long)sA2, (long)sB2, (long)sC2, transB, alpha, tiles_n, kchunk, part);
      const long n_out = nz * M * N;
      hipLaunchKernelGGL(bmm_reduce_kernel, g1(n_out), dim3(256), 0, (hipStream_t)stream, part, c, ksplit, (int)nz, M, N, ldc, (long)sC1, nb2,
                         (long)sC2, alpha);
      return hipGetLastError() == hipSuccess ? OVM_OK : OVM_ERR_HIP;
    }
  }
  for (long z0 = 0; z0 < nz; z0 += 65535 / nb2 * nb2) {          // grid.z limit; chunks keep z2 aligned
    const long cnt = (nz - z0 < (long)(65535 / nb2 * nb2)) ? nz - z0 : (long)(65535 / nb2 * nb2);
    const long o1 = z0 / nb2;
    if (M >= 48 && g_bmm_tiled) {                                 // register-blocked tiles: 48 x 16 / 32 / 48
#define OVM_BMM_TILE(TN_)                                                                                                          \
      hipLaunchKernelGGL((bmm_tile_kernel<3, TN_>), dim3((N + 16 * TN_ - 1) / (16 * TN_), (M + 47) / 48, (unsigned)cnt), dim3(256), 0,       \
                         (hipStream_t)stream, a + o1 * sA1, b + o1 * sB1, c + o1 * sC1, M, N, K, lda, ldb, ldc, (long)sA1, (long)sB1, (long)sC1, \
                         nb2, (long)sA2, (long)sB2, (long)sC2, transB, alpha)
      if (N <= 16) OVM_BMM_TILE(1); else if (N <= 32) OVM_BMM_TILE(2); else OVM_BMM_TILE(3);
#undef OVM_BMM_TILE
      continue;
    }
    hipLaunchKernelGGL(bmm_kernel, dim3((N + 15) / 16, (M + 15) / 16, (unsigned)cnt), dim3(256), 0, (hipStream_t)stream, a + o1 * sA1,
                       b + o1 * sB1, c + o1 * sC1, M, N, K, lda, ldb, ldc, (long)sA1, (long)sB1, (long)sC1, nb2, (long)sA2, (long)sB2,
                       (long)sC2, transB, alpha, tiles_n, (K + 15) / 16 * 16, (float*)nullptr);
  }
  return hipGetLastError() == hipSuccess ? OVM_OK : OVM_ERR_HIP;
}

int ovm_g_softmax(float* x, int32_t rows, int32_t cols, int32_t ld, const float* bias, int32_t bias_rows, int32_t bias_div, int32_t bias_ld,
                  ovm_stream_t stream) {
  return ovm_g_softmax2(x, rows, cols, ld, bias, bias_rows, bias_div, bias_ld, nullptr, 1, 1, stream);
}

int ovm_g_softmax2(float* x, int32_t rows, int32_t cols, int32_t ld, const float* bias, int32_t bias_rows, int32_t bias_div, int32_t bias_ld,
                   const float* bias2, int32_t d2, int32_t m2, ovm_stream_t stream) {
  if (rows <= 0) return OVM_OK;
  hipLaunchKernelGGL(softmax_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, rows, cols, ld, bias, bias_rows > 0 ? bias_rows : 1,
                     bias_div > 0 ? bias_div : 1, bias_ld, bias2, d2 > 0 ? d2 : 1, m2 > 0 ? m2 : 1);
  return hipGetLastError() == hipSuccess ? OVM_OK : OVM_ERR_HIP;
}

int ovm_g_eltwise(int32_t op, const float* a, const float* b, float* out, int64_t n, int64_t bmod, float alpha, float beta, ovm_stream_t stream) {
  if (n <= 0) return OVM_OK;
  hipLaunchKernelGGL(eltwise_kernel, g1(n), dim3(256), 0, (hipStream_t)stream, op, a, b, out, (long)n, (long)(bmod > 0 ? bmod : n), alpha, beta);
  return hipGetLastError() == hipSuccess ? OVM_OK : OVM_ERR_HIP;
}

int ovm_g_gather_rows(const float* src, int32_t ld_src, const int32_t* idx, int64_t n_out, int32_t nidx, int32_t cols, float* dst, ovm_stream_t stream) {
  if (n_out <= 0) return OVM_OK;
  hipLaunchKernelGGL(gather_rows_kernel, g1(n_out * nidx * cols), dim3(256), 0, (hipStream_t)stream, src, ld_src, idx, (long)n_out, nidx, cols, dst);
  return hipGetLastError() == hipSuccess ? OVM_OK : OVM_ERR_HIP;
}

int ovm_g_groupnorm(const float* x, int32_t B, int32_t HW, int32_t Cc, int32_t groups, const float* gamma, const float* beta, float eps, float* y,
                    ovm_stream_t stream) {
  const int cpg = groups > 0 ? Cc / groups : 0;
  const bool al16 = (((uintptr_t)x | (uintptr_t)y | (uintptr_t)gamma | (uintptr_t)beta) & 15) == 0;
  if (cpg > 0 && cpg % 4 == 0 && Cc % 4 == 0 && al16 && (long)HW * (cpg / 4) <= 1024L * kGnMaxV)
    hipLaunchKernelGGL(groupnorm_reg_kernel, dim3(groups, B), dim3(1024), 0, (hipStream_t)stream, x, HW, Cc, groups, gamma, beta, eps, y);
  else
    hipLaunchKernelGGL(groupnorm_kernel, dim3(groups, B), dim3(256), 0, (hipStream_t)stream, x, HW, Cc, groups, gamma, beta, eps, y);
  return hipGetLastError() == hipSuccess ? OVM_OK : OVM_ERR_HIP;
}

int ovm_g_msdeform(const float* value, const int32_t* shapes_hw, int32_t L, int32_t B, int32_t S, int32_t Q, int32_t H, int32_t dh, int32_t P,
                   const float* loc, const float* w, float* out, ovm_stream_t stream) {
  if (L > 8) return OVM_ERR_CAPACITY;
  MsdShapes sh; int st = 0;
  for (int l = 0; l < L; ++l) { sh.h[l] = shapes_hw[2 * l]; sh.w[l] = shapes_hw[2 * l + 1]; sh.start[l] = st; st += sh.h[l] * sh.w[l]; }
  if (st != S) return OVM_ERR_SHAPE;
  hipLaunchKernelGGL(msdeform_kernel, g1((long)B * Q * H * dh), dim3(256), 0, (hipStream_t)stream, value, sh, B, S, Q, H, dh, L, P, loc, w, out);
  return hipGetLastError() == hipSuccess ? OVM_OK : OVM_ERR_HIP;
}

int ovm_g_sine_embed(const float* pos, int64_t n, int32_t nc, int32_t F, float temperature, float* out, ovm_stream_t stream) {
  hipLaunchKernelGGL(sine_embed_kernel, g1(n * nc * F), dim3(256), 0, (hipStream_t)stream, pos, (long)n, nc, F, temperature, out);
  return hipGetLastError() == hipSuccess ? OVM_OK : OVM_ERR_HIP;
}

int ovm_g_normalize_image(const OvmImage* image, const float* mean, const float* stdv, int32_t flip_channels, float* out_nhwc, ovm_stream_t stream) {
  ImageDesc d{image->data, image->height, image->width, image->stride_c, image->stride_h, image->stride_w};
  hipLaunchKernelGGL(normalize_image_kernel, g1((long)d.H * d.W * 3), dim3(256), 0, (hipStream_t)stream, d, mean[0], mean[1], mean[2], stdv[0],
                     stdv[1], stdv[2], flip_channels, out_nhwc);
  return hipGetLastError() == hipSuccess ? OVM_OK : OVM_ERR_HIP;
}

int ovm_g_rowmax(const float* x, int32_t rows, int32_t cols, int32_t ld, float* out, ovm_stream_t stream) {
  if (rows <= 0) return OVM_OK;
  hipLaunchKernelGGL(rowmax_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, rows, cols, ld, out);
  return hipGetLastError() == hipSuccess ? OVM_OK : OVM_ERR_HIP;
}

int ovm_g_topk(const float* scores, int32_t n, int32_t k, int32_t* out_idx, ovm_stream_t stream) {
  return launch_topk(scores, n, k, out_idx, (hipStream_t)stream);
}

}  // extern "C"
