// ROIAlign on the virtual-depth feature pyramid, cube decode, detector postprocess + compaction.
#include "kernels.hpp"

namespace ovm {

// ---------------------------------------------------------------------------------------------
// ROIAlign (aligned=True, sampling_ratio=0 => ceil(roi/7) samples per bin axis) over NHWC fp32
// levels, with detectron2's ROIPooler level rule evaluated in-kernel. One wave per output bin,
// 4 channels per lane (float4 loads: 1 KiB contiguous per tap for C=256) - HBM/L2-bound gather.
// Arithmetic follows torchvision roi_align's bilinear_interpolate, reached
// from reference roi_heads.py:270 (box_pooler) and :366 (cube_pooler).
// Output row = flattened (ph, pw, c): the FC weights are re-ordered to match at pack time.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void roi_align_kernel(const RoiParams p) {
  const int roi = blockIdx.x / (p.out * p.out);
  const int bin = blockIdx.x - roi * (p.out * p.out);
  const int ph = bin / p.out, pw = bin - ph * p.out;
  const int lane = threadIdx.x;
  const float bx1 = p.boxes[roi * 4 + 0], by1 = p.boxes[roi * 4 + 1];
  const float bx2 = p.boxes[roi * 4 + 2], by2 = p.boxes[roi * 4 + 3];
  int lvl = 0;
  if (p.nlevels > 1) {
    const float size = sqrtf((bx2 - bx1) * (by2 - by1));
    float l = floorf(4.0f + log2f(size / 224.0f + 1e-8f));
    l = fminf(fmaxf(l, (float)p.min_level), (float)p.max_level);   // NaN (negative area) -> clamps like torch.clamp
    if (!(l == l)) l = (float)p.min_level;
    lvl = (int)l - p.min_level;
  }
  const int H = p.fh[lvl], W = p.fw[lvl];
  const float sc = p.scale[lvl];
  const float* feat = p.feat[lvl] + (size_t)p.batch_idx[roi] * H * W * p.C;
  const float x1 = bx1 * sc - 0.5f, y1 = by1 * sc - 0.5f;
  const float x2 = bx2 * sc - 0.5f, y2 = by2 * sc - 0.5f;
  const float roi_w = x2 - x1, roi_h = y2 - y1;
  const float bin_h = roi_h / (float)p.out, bin_w = roi_w / (float)p.out;
  const int gh = (int)ceilf(roi_h / (float)p.out), gw = (int)ceilf(roi_w / (float)p.out);
  const float count = fmaxf((float)(gh * gw), 1.0f);
  const int nv = p.C >> 2;
  for (int cv = lane; cv < nv; cv += 64) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int iy = 0; iy < gh; ++iy) {
      const float yy = y1 + (float)ph * bin_h + ((float)iy + 0.5f) * bin_h / (float)gh;
      for (int ix = 0; ix < gw; ++ix) {
        const float xx = x1 + (float)pw * bin_w + ((float)ix + 0.5f) * bin_w / (float)gw;
        if (yy < -1.0f || yy > (float)H || xx < -1.0f || xx > (float)W) continue;
        float y = fmaxf(yy, 0.f), x = fmaxf(xx, 0.f);
        int yl = (int)y, xl = (int)x, yh, xh;
        if (yl >= H - 1) { yh = yl = H - 1; y = (float)yl; } else yh = yl + 1;
        if (xl >= W - 1) { xh = xl = W - 1; x = (float)xl; } else xh = xl + 1;
        const float ly = y - (float)yl, lx = x - (float)xl, hy = 1.f - ly, hx = 1.f - lx;
        const float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
        const f32x4 v1 = ((const f32x4*)(feat + ((size_t)yl * W + xl) * p.C))[cv];
        const f32x4 v2 = ((const f32x4*)(feat + ((size_t)yl * W + xh) * p.C))[cv];
        const f32x4 v3 = ((const f32x4*)(feat + ((size_t)yh * W + xl) * p.C))[cv];
        const f32x4 v4 = ((const f32x4*)(feat + ((size_t)yh * W + xh) * p.C))[cv];
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] += w1 * v1[r] + w2 * v2[r] + w3 * v3[r] + w4 * v4[r];
      }
    }
    half4 h, l;
#pragma unroll
    for (int r = 0; r < 4; ++r) { half_t hh, ll; split_f16(acc[r] / count, hh, ll); h[r] = hh; l[r] = ll; }
    const size_t o = (size_t)roi * p.ldo + (size_t)bin * p.C + cv * 4;
    *(half4*)(p.Ohi + o) = h;
    if (p.Olo) *(half4*)(p.Olo + o) = l;
  }
}

int launch_roi_align(const RoiParams& p, hipStream_t s) {
  if (p.n <= 0) return OVM_OK;
  if (p.C % 4 != 0) return OVM_ERR_SHAPE;
  hipLaunchKernelGGL(roi_align_kernel, dim3(p.n * p.out * p.out), dim3(64), 0, s, p);
  return hipGetLastError() == hipSuccess ? OVM_OK : OVM_ERR_HIP;
}

// ---------------------------------------------------------------------------------------------
// Cube decode: one thread per box. Restates ROIHeads3D._forward_cube's eval tail
// (reference roi_heads.py:378-408, :480-515, :548-549, :798-843) with Base.yaml:71-86 settings
// (Z direct, 6d pose, allocentric, virtual depth, confidence), CubeHead's 6D->R and uncertainty clip
// (cube_head.py:164,177), util.R_from_allocentric (math_util.py:651-679), get_cuboid_verts_faces
// (math_util.py:116-196) and detectron2's detector_postprocess box rescale/clip/non-empty test.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void normalize3(float& x, float& y, float& z) {
  const float n = fmaxf(sqrtf(x * x + y * y + z * z), 1e-12f);   // F.normalize eps
  x /= n; y /= n; z /= n;
}

__global__ void cube_decode_kernel(const CubeDecodeParams p) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= p.n) return;
  const float* hd = p.head + (size_t)i * p.ldh;
  const int bi = p.batch_idx[i];
  const ImageMeta mt = p.meta[bi];
  // Ks_scaled_per_box = K / ratio, [2][2] = 1      (roi_heads.py:378-382)
  const float fx = mt.K[0] / mt.ratio, fy = mt.K[4] / mt.ratio;
  const float cx = mt.K[2] / mt.ratio, cy = mt.K[5] / mt.ratio;
  const float focal = mt.K[4];                                     // :384
  const float im_scale = (float)mt.net_h;                          // :395-398
  const float im_scale_orig = im_scale * mt.ratio;                 // :400
  const float v2r = (im_scale * focal) / (p.virtual_focal * im_scale_orig);   // math_util.py:592 with (f,H,f0,H0)
  const float bx1 = p.boxes[i * 4 + 0], by1 = p.boxes[i * 4 + 1], bx2 = p.boxes[i * 4 + 2], by2 = p.boxes[i * 4 + 3];
  const float sw = bx2 - bx1, sh = by2 - by1;
  const float ctr_x = bx1 + 0.5f * sw, ctr_y = by1 + 0.5f * sh;
  const float cube_x = ctr_x + sw * hd[0], cube_y = ctr_y + sh * hd[1];        // :480-481
  const float dW = expf(fminf(hd[2], 5.f)), dH = expf(fminf(hd[3], 5.f)), dL = expf(fminf(hd[4], 5.f));   // :507
  // rotation_6d_to_matrix (rows b1, b2, b3)
  float a1x = hd[5], a1y = hd[6], a1z = hd[7], a2x = hd[8], a2y = hd[9], a2z = hd[10];
  normalize3(a1x, a1y, a1z);
  const float dt = a1x * a2x + a1y * a2y + a1z * a2z;
  float b2x = a2x - dt * a1x, b2y = a2y - dt * a1y, b2z = a2z - dt * a1z;
  normalize3(b2x, b2y, b2z);
  const float b3x = a1y * b2z - a1z * b2y, b3y = a1z * b2x - a1x * b2z, b3z = a1x * b2y - a1y * b2x;
  float R[9] = {a1x, a1y, a1z, b2x, b2y, b2z, b3x, b3y, b3z};
  // allocentric -> egocentric (math_util.py:658-679)
  {
    float ox = (cube_x - cx) / fx, oy = (cube_y - cy) / fy, oz = 1.f;
    const float on = sqrtf(ox * ox + oy * oy + oz * oz);
    ox /= on; oy /= on; oz /= on;
    const float angle = acosf(oz);
    if (angle > 0.f) {
      const float ax = -oy, ay = ox;
      const float an = sqrtf(ax * ax + ay * ay);
      const float vx = angle * ax / an, vy = angle * ay / an, vz = angle * 0.f / an;
      // pytorch3d axis_angle_to_matrix via quaternion
      const float ang = sqrtf(vx * vx + vy * vy + vz * vz);
      const float half = ang * 0.5f;
      const float sh_over = (fabsf(ang) < 1e-6f) ? (0.5f - ang * ang / 48.f) : (sinf(half) / ang);
      const float qr = cosf(half), qi = vx * sh_over, qj = vy * sh_over, qk = vz * sh_over;
      const float two_s = 2.0f / (qr * qr + qi * qi + qj * qj + qk * qk);
      const float M[9] = {1.f - two_s * (qj * qj + qk * qk), two_s * (qi * qj - qk * qr), two_s * (qi * qk + qj * qr),
                          two_s * (qi * qj + qk * qr), 1.f - two_s * (qi * qi + qk * qk), two_s * (qj * qk - qi * qr),
                          two_s * (qi * qk - qj * qr), two_s * (qj * qk + qi * qr), 1.f - two_s * (qi * qi + qj * qj)};
      float Rn[9];
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int c = 0; c < 3; ++c) Rn[a * 3 + c] = M[a * 3 + 0] * R[c] + M[a * 3 + 1] * R[3 + c] + M[a * 3 + 2] * R[6 + c];
#pragma unroll
      for (int a = 0; a < 9; ++a) R[a] = Rn[a];
    }
  }
  const float z = hd[11] * v2r;                                                // :515, :548-549
  const float uncert = fmaxf(hd[12], 0.01f);                                  // cube_head.py:164
  const float x3d = z * (cube_x - cx) / fx, y3d = z * (cube_y - cy) / fy;     // :802-803
  const float conf = expf(-uncert);                                           // :807
  const float score = sqrtf(p.scores[i] * conf);                              // :825
  float* rec = p.rec + (size_t)i * kRecFloats;
  // detector_postprocess: scale to original resolution, clip, non-empty
  const float sx = (float)mt.orig_w / (float)mt.net_w, sy = (float)mt.orig_h / (float)mt.net_h;
  const float ox1 = fminf(fmaxf(bx1 * sx, 0.f), (float)mt.orig_w), oy1 = fminf(fmaxf(by1 * sy, 0.f), (float)mt.orig_h);
  const float ox2 = fminf(fmaxf(bx2 * sx, 0.f), (float)mt.orig_w), oy2 = fminf(fmaxf(by2 * sy, 0.f), (float)mt.orig_h);
  if (p.postprocess) { rec[0] = ox1; rec[1] = oy1; rec[2] = ox2; rec[3] = oy2; }
  else { rec[0] = bx1; rec[1] = by1; rec[2] = bx2; rec[3] = by2; }
  rec[4] = score;
  ((int*)rec)[5] = p.classes[i];
  // cuboid corners: X = +-l/2 (0,3,4,7 negative), Y = +-h/2 (0,1,4,5 negative), Z = +-w/2 (0..3 negative)
#pragma unroll
  for (int v = 0; v < 8; ++v) {
    const float vx = ((v == 0 || v == 3 || v == 4 || v == 7) ? -dL : dL) * 0.5f;
    const float vy = ((v == 0 || v == 1 || v == 4 || v == 5) ? -dH : dH) * 0.5f;
    const float vz = ((v < 4) ? -dW : dW) * 0.5f;
    rec[6 + v * 3 + 0] = (R[0] * vx + R[1] * vy + R[2] * vz) + x3d;
    rec[6 + v * 3 + 1] = (R[3] * vx + R[4] * vy + R[5] * vz) + y3d;
    rec[6 + v * 3 + 2] = (R[6] * vx + R[7] * vy + R[8] * vz) + z;
  }
  rec[30] = x3d; rec[31] = y3d; rec[32] = z;
  rec[33] = cube_x * mt.ratio; rec[34] = cube_y * mt.ratio;                   // :804
  rec[35] = dW; rec[36] = dH; rec[37] = dL;
#pragma unroll
  for (int a = 0; a < 9; ++a) rec[38 + a] = R[a];
  ((int*)rec)[47] = bi;
  p.keep[i] = (!p.postprocess || ((ox2 - ox1) > 0.f && (oy2 - oy1) > 0.f)) ? 1 : 0;
}

int launch_cube_decode(const CubeDecodeParams& p, hipStream_t s) {
  if (p.n <= 0) return OVM_OK;
  hipLaunchKernelGGL(cube_decode_kernel, dim3((p.n + 127) / 128), dim3(128), 0, s, p);
  return hipGetLastError() == hipSuccess ? OVM_OK : OVM_ERR_HIP;
}

// Stable compaction of kept records (order preserved), plus per-image counts. Single workgroup:
// n <= a few thousand; wave ballot + LDS prefix over 16 waves.
__global__ __launch_bounds__(1024) void compact_records_kernel(const float* __restrict__ rec, const int* __restrict__ keep,
                                                               int n, int B, float* __restrict__ out, int* __restrict__ counts) {
  __shared__ int wsum[16];
  __shared__ int base_s;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) base_s = 0;
  for (int b = tid; b < B; b += 1024) counts[b] = 0;
  __syncthreads();
  for (int start = 0; start < n; start += 1024) {
    const int i = start + tid;
    const int k = (i < n) ? keep[i] : 0;
    const unsigned long long bal = __ballot(k != 0);
    const int pre = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) wsum[wave] = __popcll(bal);
    __syncthreads();
    int woff = 0, tot = 0;
    for (int w = 0; w < 16; ++w) { if (w < wave) woff += wsum[w]; tot += wsum[w]; }
    const int base = base_s;
    if (k) {
      const int dst = base + woff + pre;
      const f32x4* src = (const f32x4*)(rec + (size_t)i * kRecFloats);
      f32x4* d = (f32x4*)(out + (size_t)dst * kRecFloats);
#pragma unroll
      for (int q = 0; q < kRecFloats / 4; ++q) d[q] = src[q];
      atomicAdd(&counts[((const int*)(rec + (size_t)i * kRecFloats))[47]], 1);
    }
    __syncthreads();
    if (tid == 0) base_s = base + tot;
    __syncthreads();
  }
}

int launch_compact_records(const float* rec, const int* keep, int n, int B, float* out, int* counts, hipStream_t s) {
  hipLaunchKernelGGL(compact_records_kernel, dim3(1), dim3(1024), 0, s, rec, keep, n, B, out, counts);
  return hipGetLastError() == hipSuccess ? OVM_OK : OVM_ERR_HIP;
}

}  // namespace ovm
