// HBM-bound helper kernels: patch gather (+normalise, +zero pad), LayerNorm (token rows and NHWC
// channel rows), cls/pos init, dense-token cast, 2x2 max-pool, depth-prompt resize.
#include "kernels.hpp"

namespace ovm {

// ---------------------------------------------------------------------------------------------
// Patch gather: u8 image (arbitrary C/H/W strides, so CHW dict tensors and native NHWC both work)
// -> normalised fp16 split rows A[b*G2 + p][k], k = (py*14 + px)*3 + c, zero beyond the image
// (ImageList pad value 0 is applied AFTER normalisation) and zero in the K padding columns.
// One thread per (patch, py): P px * 3 c outputs (P = 14: 84 contiguous bytes per part; P = 16: 96).
// Follows detectron2 preprocess_image as called at reference rcnn3d.py:88 + the ViT's patch conv
// (dinov2 PatchEmbed 14x14/14; open_clip conv1 16x16/16, reference clip.py:66).
// ---------------------------------------------------------------------------------------------
template <int P>
__global__ void patch_gather_kernel(const ImageDesc* __restrict__ imgs, int B, int G, int Kpad,
                                    float m0, float m1, float m2, float s0, float s1, float s2,
                                    half_t* __restrict__ Ahi, half_t* __restrict__ Alo) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int G2 = G * G;
  constexpr int RPP = (P == 14) ? 16 : P;           // threads per patch; P = 14: 14 py rows + 2 "rows" that zero the K padding
  const int total = B * G2 * RPP;
  if (idx >= total) return;
  const int py = idx % RPP;
  const int pid = idx / RPP;
  const int b = pid / G2, p = pid - b * G2;
  const int gy = p / G, gx = p - gy * G;
  half_t* oh = Ahi + (size_t)pid * Kpad;
  half_t* ol = Alo ? Alo + (size_t)pid * Kpad : nullptr;
  if (py >= P) {                                    // K padding: columns 3 P^2..Kpad-1, split over 2 threads
    const int kpad0 = 3 * P * P, n = Kpad - kpad0;
    const int half_n = (n + 1) / 2;
    const int beg = kpad0 + (py - P) * half_n;
    const int end = min(Kpad, beg + half_n);
    for (int k = beg; k < end; ++k) { oh[k] = (half_t)0.f; if (ol) ol[k] = (half_t)0.f; }
    return;
  }
  const ImageDesc d = imgs[b];
  const int y = gy * P + py;
  const float mean[3] = {m0, m1, m2}, stdv[3] = {s0, s1, s2};
  const int k0 = py * 3 * P;
  for (int px = 0; px < P; ++px) {
    const int x = gx * P + px;
    const bool in = (y < d.H) && (x < d.W);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float v = 0.f;
      if (in) v = ((float)d.data[(size_t)c * d.sC + (size_t)y * d.sH + (size_t)x * d.sW] - mean[c]) / stdv[c];
      half_t h, l; split_f16(v, h, l);
      oh[k0 + px * 3 + c] = h;
      if (ol) ol[k0 + px * 3 + c] = l;
    }
  }
}

int launch_patch_gather(const ImageDesc* d_imgs, int B, int G, int patch, int Kpad, const float* mean, const float* stdv,
                        half_t* Ahi, half_t* Alo, hipStream_t s) {
  if (patch == 14) {
    const int total = B * G * G * 16;
    hipLaunchKernelGGL(patch_gather_kernel<14>, dim3((total + 255) / 256), dim3(256), 0, s, d_imgs, B, G, Kpad,
                       mean[0], mean[1], mean[2], stdv[0], stdv[1], stdv[2], Ahi, Alo);
  } else if (patch == 16 && Kpad == 768) {
    const int total = B * G * G * 16;
    hipLaunchKernelGGL(patch_gather_kernel<16>, dim3((total + 255) / 256), dim3(256), 0, s, d_imgs, B, G, Kpad,
                       mean[0], mean[1], mean[2], stdv[0], stdv[1], stdv[2], Ahi, Alo);
  } else {
    return OVM_ERR_INVALID;
  }
  return hipGetLastError() == hipSuccess ? OVM_OK : OVM_ERR_HIP;
}

// x[b][0][:] = cls + pos[0]   (dinov2 prepare_tokens_with_masks, reference dino.py:75)
__global__ void cls_init_kernel(float* __restrict__ X, const float* __restrict__ cls, const float* __restrict__ pos,
                                int B, int T, int D) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * D) return;
  const int b = i / D, n = i - b * D;
  X[(size_t)b * T * D + n] = cls[n] + pos[n];
}
int launch_cls_init(float* X, const float* cls, const float* pos, int B, int T, int D, hipStream_t s) {
  hipLaunchKernelGGL(cls_init_kernel, dim3((B * D + 255) / 256), dim3(256), 0, s, X, cls, pos, B, T, D);
  return hipGetLastError() == hipSuccess ? OVM_OK : OVM_ERR_HIP;
}

// ---------------------------------------------------------------------------------------------
// LayerNorm over the last dimension of fp32 rows; one wave per row, values kept in registers,
// wave-shuffle reductions (biased variance, eps inside the sqrt: F.layer_norm and detectron2's
// channel LayerNorm agree). Writes any of: fp16 split (optionally into a zero-bordered NHWC
// image, the layout the implicit-GEMM 3x3 conv reads) and fp32.
// ---------------------------------------------------------------------------------------------
template <int MAXV>
__global__ __launch_bounds__(256) void ln_rows_kernel(const float* __restrict__ X, int ldx, int M, int D,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      float eps, LnOut o) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= M) return;
  const int nv = D >> 2;
  const f32x4* x = (const f32x4*)(X + (size_t)row * ldx);
  f32x4 v[MAXV];
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int j = lane + i * 64;
    if (j < nv) { v[i] = x[j]; sum += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]); }
  }
  const float mean = wave_sum(sum) / (float)D;
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int j = lane + i * 64;
    if (j < nv) {
#pragma unroll
      for (int r = 0; r < 4; ++r) { const float d = v[i][r] - mean; sq += d * d; }
    }
  }
  const float var = wave_sum(sq) / (float)D;
  const float rstd = 1.0f / sqrtf(var + eps);
  size_t orow = (size_t)row;
  if (o.padH > 0) {
    const int xx = row % o.padW; const int t = row / o.padW; const int yy = t % o.padH; const int b = t / o.padH;
    orow = ((size_t)b * (o.padH + 2) + yy + 1) * (o.padW + 2) + xx + 1;
  }
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int j = lane + i * 64;
    if (j < nv) {
      const f32x4 g = ((const f32x4*)gamma)[j];
      const f32x4 bt = ((const f32x4*)beta)[j];
      f32x4 y; half4 h, l;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        y[r] = (v[i][r] - mean) * rstd * g[r] + bt[r];
        half_t hh, ll; split_f16(y[r], hh, ll); h[r] = hh; l[r] = ll;
      }
      if (o.f32) ((f32x4*)(o.f32 + (size_t)row * o.ldf))[j] = y;
      const int oc = o.il ? il_col(j * 4) : j * 4;
      if (o.hi) *(half4*)(o.hi + orow * o.ld + oc) = h;
      if (o.lo) *(half4*)(o.lo + orow * o.ld + oc) = l;
    }
  }
}

int launch_ln_rows(const float* X, int ldx, int M, int D, const float* gamma, const float* beta, float eps,
                   const LnOut& o, hipStream_t s) {
  if (D % 4 != 0 || D > 2048) return OVM_ERR_SHAPE;
  const dim3 grid((M + 3) / 4), block(256);
  if (D <= 256) hipLaunchKernelGGL(ln_rows_kernel<1>, grid, block, 0, s, X, ldx, M, D, gamma, beta, eps, o);
  else if (D <= 1024) hipLaunchKernelGGL(ln_rows_kernel<4>, grid, block, 0, s, X, ldx, M, D, gamma, beta, eps, o);
  else hipLaunchKernelGGL(ln_rows_kernel<8>, grid, block, 0, s, X, ldx, M, D, gamma, beta, eps, o);
  return hipGetLastError() == hipSuccess ? OVM_OK : OVM_ERR_HIP;
}

// ---------------------------------------------------------------------------------------------
// Channel LayerNorm + erf-GELU, in place, on fp16 split rows [M][D] (value = hi + lo; lo null in one-pass mode): the
// middle of the scale-4 stage of the simple feature pyramid, ConvT -> LN -> GELU -> ConvT (detectron2 SimpleFeaturePyramid,
// built at reference clip.py:155-166). One wave per row, D <= 1024.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ln_gelu_split_kernel(half_t* __restrict__ Hi, half_t* __restrict__ Lo, int M, int D,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            float eps) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= M) return;
  const int nv = D >> 2;
  half_t* hi = Hi + (size_t)row * D;
  half_t* lo = Lo ? Lo + (size_t)row * D : nullptr;
  f32x4 v[4];
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int j = lane + i * 64;
    if (j < nv) {
      const half4 h = *(const half4*)(hi + j * 4);
#pragma unroll
      for (int r = 0; r < 4; ++r) v[i][r] = (float)h[r];
      if (lo) {
        const half4 l = *(const half4*)(lo + j * 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[i][r] += (float)l[r];
      }
      sum += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    }
  }
  const float mean = wave_sum(sum) / (float)D;
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int j = lane + i * 64;
    if (j < nv) {
#pragma unroll
      for (int r = 0; r < 4; ++r) { const float d = v[i][r] - mean; sq += d * d; }
    }
  }
  const float rstd = 1.0f / sqrtf(wave_sum(sq) / (float)D + eps);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int j = lane + i * 64;
    if (j < nv) {
      const f32x4 g = ((const f32x4*)gamma)[j];
      const f32x4 bt = ((const f32x4*)beta)[j];
      half4 h, l;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float y = gelu_erf((v[i][r] - mean) * rstd * g[r] + bt[r]);
        half_t hh, ll; split_f16(y, hh, ll); h[r] = hh; l[r] = ll;
      }
      *(half4*)(hi + j * 4) = h;
      if (lo) *(half4*)(lo + j * 4) = l;
    }
  }
}

int launch_ln_gelu_split(half_t* hi, half_t* lo, int M, int D, const float* gamma, const float* beta, float eps, hipStream_t s) {
  if (D % 4 != 0 || D > 1024) return OVM_ERR_SHAPE;
  hipLaunchKernelGGL(ln_gelu_split_kernel, dim3((M + 3) / 4), dim3(256), 0, s, hi, lo, M, D, gamma, beta, eps);
  return hipGetLastError() == hipSuccess ? OVM_OK : OVM_ERR_HIP;
}

// ---------------------------------------------------------------------------------------------
// Dense tokens -> fp16 split rows (drops the cls token; reference dino.py:112-117 + tokens_to_output
// :168-170 is a pure re-layout, NHWC here makes it a copy). Optional extra column D carrying the
// resized depth prompt (depth fusion input, reference dino.py:91-99) with zero K padding after it.
// ---------------------------------------------------------------------------------------------
__global__ void tokens_cast_kernel(const float* __restrict__ X, int B, int T, int G2, int D, int ldo,
                                   const float* __restrict__ depth_tok, half_t* __restrict__ Ohi,
                                   half_t* __restrict__ Olo) {
  const int nv = ldo >> 2;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)B * G2 * nv;
  if (idx >= total) return;
  const int j = (int)(idx % nv);
  const size_t r = idx / nv;
  const int b = (int)(r / G2), p = (int)(r - (size_t)b * G2);
  half4 h, l;
  const int n = j * 4;
  if (n < D) {
    const f32x4 v = *(const f32x4*)(X + ((size_t)b * T + (T - G2) + p) * D + n);     // T - G2 leading (class) tokens
#pragma unroll
    for (int q = 0; q < 4; ++q) { half_t hh, ll; split_f16(v[q], hh, ll); h[q] = hh; l[q] = ll; }
  } else {
#pragma unroll
    for (int q = 0; q < 4; ++q) { h[q] = (half_t)0.f; l[q] = (half_t)0.f; }
    if (n == D && depth_tok) { half_t hh, ll; split_f16(depth_tok[r], hh, ll); h[0] = hh; l[0] = ll; }
  }
  *(half4*)(Ohi + r * ldo + n) = h;
  if (Olo) *(half4*)(Olo + r * ldo + n) = l;
}
int launch_tokens_cast(const float* X, int B, int T, int G2, int D, int ldo, const float* depth_tok,
                       half_t* Ohi, half_t* Olo, hipStream_t s) {
  const size_t total = (size_t)B * G2 * (ldo / 4);
  hipLaunchKernelGGL(tokens_cast_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, X, B, T, G2, D, ldo,
                     depth_tok, Ohi, Olo);
  return hipGetLastError() == hipSuccess ? OVM_OK : OVM_ERR_HIP;
}

// fused-token write-back: X[b][1+p][:] = F[b*G2+p][:]   (reference dino.py:101-105)
__global__ void tokens_writeback_kernel(float* __restrict__ X, const float* __restrict__ F, int B, int T, int G2, int D) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)B * G2 * (D / 4);
  if (idx >= total) return;
  const int j = (int)(idx % (D / 4));
  const size_t r = idx / (D / 4);
  const int b = (int)(r / G2), p = (int)(r - (size_t)b * G2);
  ((f32x4*)(X + ((size_t)b * T + (T - G2) + p) * D))[j] = ((const f32x4*)(F + r * D))[j];
}
int launch_tokens_writeback(float* X, const float* F, int B, int T, int G2, int D, hipStream_t s) {
  const size_t total = (size_t)B * G2 * (D / 4);
  hipLaunchKernelGGL(tokens_writeback_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, X, F, B, T, G2, D);
  return hipGetLastError() == hipSuccess ? OVM_OK : OVM_ERR_HIP;
}

// 2x2/2 max-pool on split fp16 NHWC rows [B][G][G][D] -> [B][G/2][G/2][D]. split() is monotone, so
// the pair of the larger reconstructed value is exactly split(max(x)).
__global__ void maxpool2_kernel(const half_t* __restrict__ Ihi, const half_t* __restrict__ Ilo, int B, int G, int D,
                                half_t* __restrict__ Ohi, half_t* __restrict__ Olo) {
  const int Go = G / 2, nv = D >> 2;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)B * Go * Go * nv;
  if (idx >= total) return;
  const int j = (int)(idx % nv);
  size_t r = idx / nv;
  const int xo = (int)(r % Go); r /= Go;
  const int yo = (int)(r % Go); const int b = (int)(r / Go);
  half4 bh, bl; float best[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int y = 2 * yo + (t >> 1), x = 2 * xo + (t & 1);
    const size_t off = (((size_t)b * G + y) * G + x) * D + j * 4;
    const half4 h = *(const half4*)(Ihi + off);
    half4 l = {(half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f};
    if (Ilo) l = *(const half4*)(Ilo + off);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float v = (float)h[q] + (float)l[q] * kLoInv;
      if (t == 0 || v > best[q]) { best[q] = v; bh[q] = h[q]; bl[q] = l[q]; }
    }
  }
  const size_t oo = (((size_t)b * Go + yo) * Go + xo) * D + j * 4;
  *(half4*)(Ohi + oo) = bh;
  if (Olo) *(half4*)(Olo + oo) = bl;
}
int launch_maxpool2(const half_t* Ihi, const half_t* Ilo, int B, int G, int D, half_t* Ohi, half_t* Olo, hipStream_t s) {
  const size_t total = (size_t)B * (G / 2) * (G / 2) * (D / 4);
  hipLaunchKernelGGL(maxpool2_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, Ihi, Ilo, B, G, D, Ohi, Olo);
  return hipGetLastError() == hipSuccess ? OVM_OK : OVM_ERR_HIP;
}

// Bilinear resize (align_corners=False, no antialias) of fp32 planes [B][Hd][Wd] -> [B][Ho][Wo]: F.interpolate(mode='bilinear')
// as the reference applies it to depth prompts - to the image size and through ResizeShortestEdge in the mapper
// (dataset_mapper.py:45-52,70-72: detectron2's ResizeTransform uses F.interpolate for non-uint8 input), then to the token grid
// in the backbone (dino.py:85).
__global__ void resize_bilinear_f32_kernel(const float* __restrict__ Dp, int B, int Hd, int Wd, int Ho, int Wo, float* __restrict__ out) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * Ho * Wo) return;
  const int x = idx % Wo; const int t = idx / Wo; const int y = t % Ho; const int b = t / Ho;
  const float sy = (float)Hd / (float)Ho, sx = (float)Wd / (float)Wo;
  float fy = ((float)y + 0.5f) * sy - 0.5f; if (fy < 0.f) fy = 0.f;
  float fx = ((float)x + 0.5f) * sx - 0.5f; if (fx < 0.f) fx = 0.f;
  int y0 = (int)fy, x0 = (int)fx;
  if (y0 > Hd - 1) y0 = Hd - 1;
  if (x0 > Wd - 1) x0 = Wd - 1;
  const int y1 = y0 + ((y0 < Hd - 1) ? 1 : 0), x1 = x0 + ((x0 < Wd - 1) ? 1 : 0);
  const float ly = fy - (float)y0, lx = fx - (float)x0;
  const float* d = Dp + (size_t)b * Hd * Wd;
  const float v = (1.f - ly) * ((1.f - lx) * d[y0 * Wd + x0] + lx * d[y0 * Wd + x1]) +
                  ly * ((1.f - lx) * d[y1 * Wd + x0] + lx * d[y1 * Wd + x1]);
  out[idx] = v;
}
int launch_resize_bilinear_f32(const float* src, int B, int Hd, int Wd, int Ho, int Wo, float* out, hipStream_t s) {
  if (B < 1 || Hd < 1 || Wd < 1 || Ho < 1 || Wo < 1) return OVM_ERR_INVALID;
  const int total = B * Ho * Wo;
  hipLaunchKernelGGL(resize_bilinear_f32_kernel, dim3((total + 255) / 256), dim3(256), 0, s, src, B, Hd, Wd, Ho, Wo, out);
  return hipGetLastError() == hipSuccess ? OVM_OK : OVM_ERR_HIP;
}
int launch_depth_resize(const float* Dp, int B, int Hd, int Wd, int G, float* out, hipStream_t s) {
  return launch_resize_bilinear_f32(Dp, B, Hd, Wd, G, G, out, s);
}

// zero-fill helper for split fp16 / fp32 buffers on a stream
int launch_zero(void* p, size_t bytes, hipStream_t s) {
  return hipMemsetAsync(p, 0, bytes, s) == hipSuccess ? OVM_OK : OVM_ERR_HIP;
}

}  // namespace ovm
