// Instantiations + host dispatch of the MFMA GEMM (see gemm.hpp).
#include "gemm.hpp"

namespace ovm {

template <int NPASS, int BK, int EPI, int AMODE>
static int launch_one(const GemmParams& p, hipStream_t s) {
  constexpr int NPART = (NPASS == 3) ? 4 : 2;
  constexpr int smem = 2 * NPART * 128 * BK * 2;
  const int tiles_m = (p.M + 127) / 128;
  const int tiles_n = (p.N + 127) / 128;
  if (p.M <= 0 || p.N <= 0) return OVM_OK;
  if (p.K % BK != 0) return OVM_ERR_SHAPE;
  hipLaunchKernelGGL((gemm_kernel<NPASS, BK, EPI, AMODE>), dim3(tiles_m * tiles_n), dim3(256), smem, s, p);
  return hipGetLastError() == hipSuccess ? OVM_OK : OVM_ERR_HIP;
}

template <int EPI, int AMODE>
static int launch_prec(const GemmParams& p, int npass, hipStream_t s) {
  if (npass == 3) return launch_one<3, 32, EPI, AMODE>(p, s);
  return launch_one<1, 64, EPI, AMODE>(p, s);
}

int launch_gemm(const GemmParams& p, int npass, int epi, int amode, hipStream_t s) {
  if (npass == 3 && (p.Alo == nullptr || p.Wlo == nullptr)) return OVM_ERR_INVALID;
  if (amode == A_CONV3X3) {
    if (epi != EPI_STORE) return OVM_ERR_INVALID;
    return launch_prec<EPI_STORE, A_CONV3X3>(p, npass, s);
  }
  switch (epi) {
    case EPI_STORE: return launch_prec<EPI_STORE, A_ROWMAJOR>(p, npass, s);
    case EPI_RESID: return launch_prec<EPI_RESID, A_ROWMAJOR>(p, npass, s);
    case EPI_GELU:  return launch_prec<EPI_GELU, A_ROWMAJOR>(p, npass, s);
    case EPI_QKV:   return launch_prec<EPI_QKV, A_ROWMAJOR>(p, npass, s);
    case EPI_PATCH: return launch_prec<EPI_PATCH, A_ROWMAJOR>(p, npass, s);
    case EPI_CONVT: return launch_prec<EPI_CONVT, A_ROWMAJOR>(p, npass, s);
  }
  return OVM_ERR_INVALID;
}

}  // namespace ovm
