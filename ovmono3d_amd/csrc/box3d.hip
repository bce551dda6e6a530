// Exact intersection volume / IoU of oriented 3D boxes given by their 8 corners (pytorch3d corner order), one thread per
// (detection, ground-truth) pair. Replaces pytorch3d's `_C.iou_box3d` behind the reference's `box3d_overlap`
// (cubercnn/evaluation/omni3d_evaluation.py:109-169), including its validity screening of the detections
// (`_check_coplanar` :68-87, `_check_nonzero` :90-107).
//
// Method: the intersection of two convex boxes is a convex polyhedron whose faces lie on faces of A or of B. By the
// divergence theorem  V = 1/3 * sum_f (p_f . n_f) * area(f),  so
//     V = 1/3 * sum_{faces f of A} (p_f . n_f) * area(f clipped to B)  +  1/3 * sum_{faces g of B} (p_g . n_g) * area(g clipped to A)
// with outward unit normals n and any point p on the face. Each quad face is clipped against the other box's six
// half-spaces (Sutherland-Hodgman, at most 10 vertices). A's faces are clipped against B's CLOSED half-spaces and B's faces
// against A's OPEN ones, so a coincident face pair is counted once (IoU of a box with itself is exactly 1).
#include <hip/hip_runtime.h>
#include <cstdint>
#include "common.hpp"
#include "../../include/ovm3d.h"

namespace {

struct V3 { float x, y, z; };
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
__device__ __forceinline__ float norm(V3 a) { return sqrtf(dot(a, a)); }

// pytorch3d `_box_planes`: corner indices of the six quad faces, each in cyclic order
__constant__ int kFace[6][4] = {{0, 1, 2, 3}, {3, 2, 6, 7}, {0, 1, 5, 4}, {0, 3, 7, 4}, {1, 2, 6, 5}, {4, 5, 6, 7}};
// pytorch3d `_box_triangles`
__constant__ int kTri[12][3] = {{0, 1, 2}, {0, 3, 2}, {4, 5, 6}, {4, 6, 7}, {1, 5, 6}, {1, 6, 2}, {0, 4, 7}, {0, 7, 3}, {3, 2, 6}, {3, 6, 7}, {0, 1, 5}, {0, 4, 5}};

struct Box {
  V3 c[8];
  V3 n[6];        // outward unit normals
  float d[6];     // plane offsets: inside <=> n . x <= d
  float vol;
  float scale;    // largest |coordinate|: sets the snapping tolerance of the plane tests
};

__device__ void load_box(const float* p, Box& b) {
  V3 ctr = {0.f, 0.f, 0.f};
  b.scale = 0.f;
  for (int i = 0; i < 8; ++i) {
    b.c[i] = {p[3 * i], p[3 * i + 1], p[3 * i + 2]}; ctr = ctr + b.c[i];
    b.scale = fmaxf(b.scale, fmaxf(fabsf(b.c[i].x), fmaxf(fabsf(b.c[i].y), fabsf(b.c[i].z))));
  }
  ctr = ctr * 0.125f;
  for (int f = 0; f < 6; ++f) {
    const V3 v0 = b.c[kFace[f][0]], v1 = b.c[kFace[f][1]], v2 = b.c[kFace[f][2]], v3 = b.c[kFace[f][3]];
    V3 n = cross(v1 - v0, v3 - v0);
    const float l = norm(n);
    n = l > 0.f ? n * (1.f / l) : V3{0.f, 0.f, 0.f};
    const V3 fc = (v0 + v1 + v2 + v3) * 0.25f;
    if (dot(n, fc - ctr) < 0.f) n = n * -1.f;
    b.n[f] = n; b.d[f] = dot(n, fc);
  }
  // volume of the hexahedron: the same face sum with unclipped faces
  float v = 0.f;
  for (int f = 0; f < 6; ++f) {
    const V3 v0 = b.c[kFace[f][0]], v1 = b.c[kFace[f][1]], v2 = b.c[kFace[f][2]], v3 = b.c[kFace[f][3]];
    const float area = 0.5f * (norm(cross(v1 - v0, v2 - v0)) + norm(cross(v2 - v0, v3 - v0)));
    v += b.d[f] * area;
  }
  b.vol = v * (1.f / 3.f);
}

// area of face `f` of `a` inside box `o`; closed = keep points on o's planes
__device__ float clipped_face_area(const Box& a, int f, const Box& o, bool closed) {
  V3 poly[12], tmp[12];
  int n = 4;
  for (int i = 0; i < 4; ++i) poly[i] = a.c[kFace[f][i]];
  for (int pl = 0; pl < 6 && n > 0; ++pl) {
    const V3 pn = o.n[pl]; const float pd = o.d[pl];
    // Signed distances within rounding noise of the plane are snapped to "on the plane" (a few fp32 ulps of the coordinate
    // magnitude, ~1e-5 m at 10 m): a face of `a` lying in a plane of `o` is then inside for the closed test and outside for
    // the open one, whichever way the noise fell.
    const float eps = 4e-6f * (fmaxf(a.scale, o.scale) + fabsf(pd));
    int m = 0;
    for (int i = 0; i < n; ++i) {
      const V3 cur = poly[i], nxt = poly[(i + 1 == n) ? 0 : i + 1];
      float sc = dot(pn, cur) - pd, sn = dot(pn, nxt) - pd;
      if (fabsf(sc) <= eps) sc = 0.f;
      if (fabsf(sn) <= eps) sn = 0.f;
      const bool in_c = closed ? (sc <= 0.f) : (sc < 0.f);
      const bool in_n = closed ? (sn <= 0.f) : (sn < 0.f);
      if (in_c) tmp[m++] = cur;
      if (in_c != in_n) {
        const float t = sc / (sc - sn);
        tmp[m++] = cur + (nxt - cur) * t;
      }
    }
    n = m < 12 ? m : 12;
    for (int i = 0; i < n; ++i) poly[i] = tmp[i];
  }
  if (n < 3) return 0.f;
  V3 acc = {0.f, 0.f, 0.f};
  for (int i = 1; i + 1 < n; ++i) acc = acc + cross(poly[i] - poly[0], poly[i + 1] - poly[0]);
  return 0.5f * fabsf(dot(acc, a.n[f]));
}

__device__ bool box_valid(const Box& b, float eps_coplanar, float eps_nonzero) {
  // reference _check_coplanar: |(v3 - v0) . normalize(cross(normalize(v1 - v0), normalize(v2 - v0)))| summed over the 6 planes < eps
  float s = 0.f;
  for (int f = 0; f < 6; ++f) {
    const V3 v0 = b.c[kFace[f][0]], v1 = b.c[kFace[f][1]], v2 = b.c[kFace[f][2]], v3 = b.c[kFace[f][3]];
    V3 e0 = v1 - v0, e1 = v2 - v0;
    const float l0 = fmaxf(norm(e0), 1e-12f), l1 = fmaxf(norm(e1), 1e-12f);
    e0 = e0 * (1.f / l0); e1 = e1 * (1.f / l1);
    V3 nn = cross(e0, e1);
    const float ln = fmaxf(norm(nn), 1e-12f);
    nn = nn * (1.f / ln);
    s += dot(v3 - v0, nn);
  }
  if (!(fabsf(s) < eps_coplanar)) return false;
  // reference _check_nonzero: every triangle area > eps
  for (int t = 0; t < 12; ++t) {
    const V3 v0 = b.c[kTri[t][0]], v1 = b.c[kTri[t][1]], v2 = b.c[kTri[t][2]];
    if (!(0.5f * norm(cross(v1 - v0, v2 - v0)) > eps_nonzero)) return false;
  }
  return true;
}

__global__ __launch_bounds__(128) void box3d_iou_kernel(const float* __restrict__ dt, const float* __restrict__ gt, int N, int M, float eps_coplanar,
                                                        float eps_nonzero, float* __restrict__ iou, float* __restrict__ vol) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)N * M) return;
  const int i = (int)(idx / M), j = (int)(idx % M);
  Box a, b;
  load_box(dt + (size_t)i * 24, a);
  load_box(gt + (size_t)j * 24, b);
  float v = 0.f;
  for (int f = 0; f < 6; ++f) v += a.d[f] * clipped_face_area(a, f, b, true);
  for (int f = 0; f < 6; ++f) v += b.d[f] * clipped_face_area(b, f, a, false);
  v = fmaxf(v * (1.f / 3.f), 0.f);
  float u = a.vol + b.vol - v;
  float r = (u > 0.f) ? v / u : 0.f;
  if (!box_valid(a, eps_coplanar, eps_nonzero)) { r = 0.f; }             // offending detections get IoU 0 (:160-167)
  iou[idx] = r;
  if (vol) vol[idx] = v;
}

}  // namespace

extern "C" int ovm_box3d_iou(const float* boxes_dt, const float* boxes_gt, int32_t N, int32_t M, float eps_coplanar, float eps_nonzero, float* iou,
                             float* vol, ovm_stream_t stream) {
  if (N <= 0 || M <= 0) return OVM_OK;
  if (!boxes_dt || !boxes_gt || !iou) return OVM_ERR_INVALID;
  const long n = (long)N * M;
  hipLaunchKernelGGL(box3d_iou_kernel, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, (hipStream_t)stream, boxes_dt, boxes_gt, N, M, eps_coplanar,
                     eps_nonzero, iou, vol);
  return hipGetLastError() == hipSuccess ? OVM_OK : OVM_ERR_HIP;
}
