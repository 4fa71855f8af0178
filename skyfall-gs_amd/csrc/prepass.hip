// prepass.hip -- fused per-Gaussian pre-pass of render(): activations + Mip-Splatting 3D smoothing filter.
// (the first "next" row after the rasterizer.)
//
// Replaces, in ONE kernel each way, what the reference computes every render() call with ~15 forward and ~25
// backward elementwise torch kernels over N-sized tensors (scene/gaussian_model.py):
//   scales    = sqrt(exp(_scaling)^2 + filter_3D^2)                         get_scaling_with_3D_filter  :207-213
//   opacity   = sigmoid(_opacity) * sqrt(prod s^2 / prod (s^2 + filter_3D^2)) get_opacity_with_3D_filter  :237-249
//   rotation  = normalize(_rotation)  (F.normalize, eps 1e-12)               get_rotation                :216-217
// filter_3D is float64 during training (compute_3D_filter, :258-308) and float32 after load_ply (:545): the
// reference then evaluates in float64 and render() casts to float32 (gaussian_renderer/__init__.py:137-138).
// The kernels follow torch's type promotion in both cases (float32 square / prod of the scales, then float64 or
// float32 arithmetic according to the filter's dtype), so results agree with the reference to the last bit
// except where libm's expf differs from the device's.
// 36-40 B read, 32 B written per Gaussian forward; HBM-bound.
#include "sfgs_internal.h"

namespace sfgs {

// the reference's intermediate quantities, in its own promotion rules (FT = dtype of filter_3D):
//   sq_i  = square(exp(raw_i))                    float32
//   det1  = prod_i sq_i                           float32
//   t_i   = sq_i + square(filter)                 FT
//   det2  = prod_i t_i                            FT
//   coef  = sqrt(det1 / det2)                     FT
// OT = dtype of the raw opacity parameter: float32, or float64 after the reference's reset_opacity
// (scene/gaussian_model.py:483-501 divides by a float64 coefficient, so from the first opacity reset on -- iteration
// 3000 of the default schedule -- `_opacity` and its Adam moments are float64 tensors and sigmoid runs in float64).
template <typename FT, typename OT>
struct PrepassTerms {
  float sq[3], det1;
  OT o;
  FT f2, t[3], det2, coef;
};

template <typename FT>
__device__ __forceinline__ FT sqrt_t(FT v);
template <> __device__ __forceinline__ float sqrt_t<float>(float v) { return sqrtf(v); }
template <> __device__ __forceinline__ double sqrt_t<double>(double v) { return sqrt(v); }

template <typename FT, typename OT>
__device__ __forceinline__ PrepassTerms<FT, OT> prepass_terms(const float* __restrict__ scaling_raw,
                                                              const OT* __restrict__ opacity_raw,
                                                              const FT* __restrict__ filter3d, int g) {
  PrepassTerms<FT, OT> p;
  const FT f = filter3d[g];
  p.f2 = f * f;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const float s = expf(scaling_raw[3 * (size_t)g + i]);
    p.sq[i] = s * s;
    p.t[i] = (FT)p.sq[i] + p.f2;
  }
  p.det1 = (p.sq[0] * p.sq[1]) * p.sq[2];
  p.det2 = (p.t[0] * p.t[1]) * p.t[2];
  p.coef = sqrt_t<FT>((FT)p.det1 / p.det2);
  if constexpr (sizeof(OT) == 8) p.o = 1.0 / (1.0 + exp(-opacity_raw[g]));
  else p.o = 1.0f / (1.0f + expf(-opacity_raw[g]));
  return p;
}

template <typename FT, typename OT>
__global__ void __launch_bounds__(256)
prepass_fwd_kernel(int N, const float* __restrict__ scaling_raw, const OT* __restrict__ opacity_raw,
                   const float* __restrict__ rotation_raw, const FT* __restrict__ filter3d,
                   float* __restrict__ scales, float* __restrict__ opacities, float* __restrict__ rotations) {
  const int g = blockIdx.x * 256 + threadIdx.x;
  if (g >= N) return;
  const PrepassTerms<FT, OT> p = prepass_terms<FT, OT>(scaling_raw, opacity_raw, filter3d, g);
#pragma unroll
  for (int i = 0; i < 3; ++i) scales[3 * (size_t)g + i] = (float)sqrt_t<FT>(p.t[i]);
  if constexpr (sizeof(OT) == 8) opacities[g] = (float)(p.o * (double)p.coef);   // torch promotes to float64
  else opacities[g] = (float)((FT)p.o * p.coef);
  const float4 q = *reinterpret_cast<const float4*>(rotation_raw + 4 * (size_t)g);
  const float n = fmaxf(sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w), 1e-12f);
  *reinterpret_cast<float4*>(rotations + 4 * (size_t)g) = make_float4(q.x / n, q.y / n, q.z / n, q.w / n);
}

template <typename FT, typename OT>
__global__ void __launch_bounds__(256)
prepass_bwd_kernel(int N, const float* __restrict__ scaling_raw, const OT* __restrict__ opacity_raw,
                   const float* __restrict__ rotation_raw, const FT* __restrict__ filter3d,
                   const float* __restrict__ g_scales, const float* __restrict__ g_opacities,
                   const float* __restrict__ g_rotations, float* __restrict__ g_scaling_raw,
                   OT* __restrict__ g_opacity_raw, float* __restrict__ g_rotation_raw) {
  const int g = blockIdx.x * 256 + threadIdx.x;
  if (g >= N) return;
  const PrepassTerms<FT, OT> p = prepass_terms<FT, OT>(scaling_raw, opacity_raw, filter3d, g);
  const double coef = (double)p.coef, o = (double)p.o;
  const double go = g_opacities ? (double)g_opacities[g] : 0.0;
  g_opacity_raw[g] = (OT)(go * coef * o * (1.0 - o));
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    // d scales_i / d raw_i = s_i^2 / sqrt(s_i^2 + f^2) ; d (o coef) / d raw_i = o coef f^2 / (s_i^2 + f^2)
    const double gs = g_scales ? (double)g_scales[3 * (size_t)g + i] : 0.0;
    const double t = (double)p.t[i];
    g_scaling_raw[3 * (size_t)g + i] = (float)(gs * (double)p.sq[i] / sqrt(t) + go * o * coef * (double)p.f2 / t);
  }
  const float4 q = *reinterpret_cast<const float4*>(rotation_raw + 4 * (size_t)g);
  float4 gr = make_float4(0.f, 0.f, 0.f, 0.f);
  if (g_rotations) gr = *reinterpret_cast<const float4*>(g_rotations + 4 * (size_t)g);
  const float nn = sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
  float4 out;
  if (nn > 1e-12f) {  // d (q/|q|) : (g - q_hat (q_hat . g)) / |q|
    const float inv = 1.0f / nn;
    const float4 h = make_float4(q.x * inv, q.y * inv, q.z * inv, q.w * inv);
    const float dot = h.x * gr.x + h.y * gr.y + h.z * gr.z + h.w * gr.w;
    out = make_float4((gr.x - h.x * dot) * inv, (gr.y - h.y * dot) * inv, (gr.z - h.z * dot) * inv,
                      (gr.w - h.w * dot) * inv);
  } else {  // clamped denominator: q / 1e-12
    out = make_float4(gr.x * 1e12f, gr.y * 1e12f, gr.z * 1e12f, gr.w * 1e12f);
  }
  *reinterpret_cast<float4*>(g_rotation_raw + 4 * (size_t)g) = out;
}

}  // namespace sfgs

using namespace sfgs;

// dispatch on (filter dtype, raw opacity dtype)
#define SFGS_PREPASS_DISPATCH(MASK, LAUNCH)                                      \
  do {                                                                           \
    switch ((MASK) & 3) {                                                        \
      case 0: LAUNCH(float, float); break;                                       \
      case 1: LAUNCH(double, float); break;                                      \
      case 2: LAUNCH(float, double); break;                                      \
      default: LAUNCH(double, double); break;                                    \
    }                                                                            \
  } while (0)

extern "C" int sfgs_prepass_forward(int32_t N, const float* scaling_raw, const void* opacity_raw,
                                    const float* rotation_raw, const void* filter3d, int32_t f64_mask,
                                    float* scales, float* opacities, float* rotations, void* stream_) {
  SFGS_REQUIRE(N >= 0, SFGS_E_ARG, "negative Gaussian count");
  if (N == 0) return SFGS_OK;
  SFGS_REQUIRE(scaling_raw && opacity_raw && rotation_raw && filter3d && scales && opacities && rotations, SFGS_E_ARG,
               "NULL argument");
  SFGS_REQUIRE((f64_mask & ~3) == 0, SFGS_E_ARG, "f64_mask: bit 0 = filter3d is float64, bit 1 = opacity_raw is float64");
  hipStream_t stream = (hipStream_t)stream_;
  const dim3 grid((N + 255) / 256), block(256);
  { ProfScope ps_(KID_PREPASS_FWD, stream);
#define SFGS_LAUNCH_PF(FT, OT)                                                                                      \
  hipLaunchKernelGGL((prepass_fwd_kernel<FT, OT>), grid, block, 0, stream, N, scaling_raw, (const OT*)opacity_raw, \
                     rotation_raw, (const FT*)filter3d, scales, opacities, rotations)
    SFGS_PREPASS_DISPATCH(f64_mask, SFGS_LAUNCH_PF);
#undef SFGS_LAUNCH_PF
  }
  SFGS_POST_LAUNCH("prepass_fwd", stream, 0);
  return SFGS_OK;
}

extern "C" int sfgs_prepass_backward(int32_t N, const float* scaling_raw, const void* opacity_raw,
                                     const float* rotation_raw, const void* filter3d, int32_t f64_mask,
                                     const float* g_scales, const float* g_opacities, const float* g_rotations,
                                     float* g_scaling_raw, void* g_opacity_raw, float* g_rotation_raw, void* stream_) {
  SFGS_REQUIRE(N >= 0, SFGS_E_ARG, "negative Gaussian count");
  if (N == 0) return SFGS_OK;
  SFGS_REQUIRE(scaling_raw && opacity_raw && rotation_raw && filter3d && g_scaling_raw && g_opacity_raw && g_rotation_raw,
               SFGS_E_ARG, "NULL argument");
  SFGS_REQUIRE((f64_mask & ~3) == 0, SFGS_E_ARG, "f64_mask: bit 0 = filter3d is float64, bit 1 = opacity_raw is float64");
  hipStream_t stream = (hipStream_t)stream_;
  const dim3 grid((N + 255) / 256), block(256);
  { ProfScope ps_(KID_PREPASS_BWD, stream);
#define SFGS_LAUNCH_PB(FT, OT)                                                                                      \
  hipLaunchKernelGGL((prepass_bwd_kernel<FT, OT>), grid, block, 0, stream, N, scaling_raw, (const OT*)opacity_raw, \
                     rotation_raw, (const FT*)filter3d, g_scales, g_opacities, g_rotations, g_scaling_raw,         \
                     (OT*)g_opacity_raw, g_rotation_raw)
    SFGS_PREPASS_DISPATCH(f64_mask, SFGS_LAUNCH_PB);
#undef SFGS_LAUNCH_PB
  }
  SFGS_POST_LAUNCH("prepass_bwd", stream, 0);
  return SFGS_OK;
}
