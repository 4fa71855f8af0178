// sh_eval.hip -- utils/sh_utils.py:57-112 `eval_sh(deg, sh[N,3,K], dirs[N,3]) -> [N,3]` as one kernel each way
// (the SH evaluation of the MLP-toned coefficients in render()'s appearance path,
// gaussian_renderer/__init__.py:111-116, and of `convert_SHs_python`, :120-125). In torch the degree-1 evaluation is
// ~10 elementwise kernels forward and ~25 backward over N x 3 tensors (degree 3: ~60 / ~150); here each direction is
// one pass: 12(K+2) B read + 12 B written per Gaussian forward. Layout is the reference's channel-major [N,3,K]
// (NOT the rasterizer's [N,K,3]); only the first (deg+1)^2 coefficients of the K stored ones are used, exactly like
// the reference, and the backward writes zeros to the unused ones.
#include "sfgs_internal.h"

namespace sfgs {

// (degree 4 exists only in the reference's Python eval_sh, utils/sh_utils.py:44-54,101-111; since round 4 the basis and its
// partial derivatives up to degree 4 live in raster_math.h and the rasterizer's in-kernel paths evaluate it too)
template <int DEG>
__device__ __forceinline__ void sh_basis_any(float x, float y, float z, float* Bk) { sh_basis(DEG, x, y, z, Bk); }
template <int DEG>
__device__ __forceinline__ void sh_basis_grad_any(float x, float y, float z, float* dBx, float* dBy, float* dBz) {
  sh_basis_grad(DEG, x, y, z, dBx, dBy, dBz);
}

template <int DEG>
__global__ void __launch_bounds__(256)
sh_eval_fwd_kernel(int N, int K, const float* __restrict__ sh, const float* __restrict__ dirs,
                   float* __restrict__ out) {
  const int g = blockIdx.x * 256 + threadIdx.x;
  if (g >= N) return;
  constexpr int M = (DEG + 1) * (DEG + 1);
  float Bk[M];
  const float x = dirs[3 * (size_t)g], y = dirs[3 * (size_t)g + 1], z = dirs[3 * (size_t)g + 2];
  sh_basis_any<DEG>(x, y, z, Bk);
  const float* s = sh + (size_t)g * 3 * K;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < M; ++k) acc += Bk[k] * s[c * K + k];
    out[3 * (size_t)g + c] = acc;
  }
}

template <int DEG>
__global__ void __launch_bounds__(256)
sh_eval_bwd_kernel(int N, int K, const float* __restrict__ sh, const float* __restrict__ dirs,
                   const float* __restrict__ g_out, float* __restrict__ g_sh, float* __restrict__ g_dirs) {
  const int g = blockIdx.x * 256 + threadIdx.x;
  if (g >= N) return;
  constexpr int M = (DEG + 1) * (DEG + 1);
  float Bk[M], dBx[M], dBy[M], dBz[M];
  const float x = dirs[3 * (size_t)g], y = dirs[3 * (size_t)g + 1], z = dirs[3 * (size_t)g + 2];
  sh_basis_any<DEG>(x, y, z, Bk);
  sh_basis_grad_any<DEG>(x, y, z, dBx, dBy, dBz);
  const float* s = sh + (size_t)g * 3 * K;
  float* gs = g_sh + (size_t)g * 3 * K;
  float gx = 0.f, gy = 0.f, gz = 0.f;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float go = g_out[3 * (size_t)g + c];
#pragma unroll
    for (int k = 0; k < M; ++k) {
      gs[c * K + k] = Bk[k] * go;
      const float w = go * s[c * K + k];
      gx += dBx[k] * w; gy += dBy[k] * w; gz += dBz[k] * w;
    }
    for (int k = M; k < K; ++k) gs[c * K + k] = 0.f;
  }
  if (g_dirs) { g_dirs[3 * (size_t)g] = gx; g_dirs[3 * (size_t)g + 1] = gy; g_dirs[3 * (size_t)g + 2] = gz; }
}

}  // namespace sfgs

using namespace sfgs;

static int check_sh_args(int32_t N, int32_t deg, int32_t K) {
  SFGS_REQUIRE(N >= 0, SFGS_E_ARG, "negative Gaussian count");
  SFGS_REQUIRE(deg >= 0 && deg <= 4, SFGS_E_UNSUPPORTED, "SH degree %d not in 0..4", deg);
  SFGS_REQUIRE(K >= (deg + 1) * (deg + 1), SFGS_E_ARG, "%d SH coefficients stored, degree %d needs %d", K, deg,
               (deg + 1) * (deg + 1));
  return SFGS_OK;
}

#define SH_DISPATCH(deg, KERNEL, ...)                                                                           \
  switch (deg) {                                                                                                \
    case 0: hipLaunchKernelGGL(KERNEL<0>, dim3((N + 255) / 256), dim3(256), 0, stream, __VA_ARGS__); break;     \
    case 1: hipLaunchKernelGGL(KERNEL<1>, dim3((N + 255) / 256), dim3(256), 0, stream, __VA_ARGS__); break;     \
    case 2: hipLaunchKernelGGL(KERNEL<2>, dim3((N + 255) / 256), dim3(256), 0, stream, __VA_ARGS__); break;     \
    case 3: hipLaunchKernelGGL(KERNEL<3>, dim3((N + 255) / 256), dim3(256), 0, stream, __VA_ARGS__); break;     \
    default: hipLaunchKernelGGL(KERNEL<4>, dim3((N + 255) / 256), dim3(256), 0, stream, __VA_ARGS__); break;    \
  }

extern "C" int sfgs_sh_eval_forward(int32_t N, int32_t deg, int32_t K, const float* sh, const float* dirs,
                                    float* out, void* stream_) {
  if (int rc = check_sh_args(N, deg, K)) return rc;
  if (N == 0) return SFGS_OK;
  SFGS_REQUIRE(sh && dirs && out, SFGS_E_ARG, "NULL argument");
  hipStream_t stream = (hipStream_t)stream_;
  { ProfScope ps_(KID_SH_EVAL_FWD, stream);
    SH_DISPATCH(deg, sh_eval_fwd_kernel, N, K, sh, dirs, out) }
  SFGS_POST_LAUNCH("sh_eval_fwd", stream, 0);
  return SFGS_OK;
}

extern "C" int sfgs_sh_eval_backward(int32_t N, int32_t deg, int32_t K, const float* sh, const float* dirs,
                                     const float* g_out, float* g_sh, float* g_dirs_or_null, void* stream_) {
  if (int rc = check_sh_args(N, deg, K)) return rc;
  if (N == 0) return SFGS_OK;
  SFGS_REQUIRE(sh && dirs && g_out && g_sh, SFGS_E_ARG, "NULL argument");
  hipStream_t stream = (hipStream_t)stream_;
  { ProfScope ps_(KID_SH_EVAL_BWD, stream);
    SH_DISPATCH(deg, sh_eval_bwd_kernel, N, K, sh, dirs, g_out, g_sh, g_dirs_or_null) }
  SFGS_POST_LAUNCH("sh_eval_bwd", stream, 0);
  return SFGS_OK;
}
