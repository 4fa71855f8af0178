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
#include "act_math.h"
#include "sfgs_internal.h"

namespace sfgs {

template <typename FT, typename OT>
__global__ void __launch_bounds__(256)
prepass_fwd_kernel(int N, const float* __restrict__ scaling_raw, const OT* __restrict__ opacity_raw,
                   const float* __restrict__ rotation_raw, const FT* __restrict__ filter3d,
                   float* __restrict__ scales, float* __restrict__ opacities, float* __restrict__ rotations) {
  const int g = blockIdx.x * 256 + threadIdx.x;
  if (g >= N) return;
  const float rs[3] = {scaling_raw[3 * (size_t)g], scaling_raw[3 * (size_t)g + 1], scaling_raw[3 * (size_t)g + 2]};
  const ActTerms<FT, OT> p = act_terms<FT, OT>(rs, opacity_raw[g], filter3d[g]);
  float sc[3], op;
  act_outputs(p, sc, &op);
#pragma unroll
  for (int i = 0; i < 3; ++i) scales[3 * (size_t)g + i] = sc[i];
  opacities[g] = op;
  *reinterpret_cast<float4*>(rotations + 4 * (size_t)g) =
      act_rotation(*reinterpret_cast<const float4*>(rotation_raw + 4 * (size_t)g));
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
  const float rs[3] = {scaling_raw[3 * (size_t)g], scaling_raw[3 * (size_t)g + 1], scaling_raw[3 * (size_t)g + 2]};
  const ActTerms<FT, OT> p = act_terms<FT, OT>(rs, opacity_raw[g], filter3d[g]);
  float gs[3] = {0.f, 0.f, 0.f}, grs[3];
  if (g_scales) { gs[0] = g_scales[3 * (size_t)g]; gs[1] = g_scales[3 * (size_t)g + 1]; gs[2] = g_scales[3 * (size_t)g + 2]; }
  OT gro;
  act_backward(p, gs, g_opacities ? g_opacities[g] : 0.f, grs, &gro);
  g_opacity_raw[g] = gro;
#pragma unroll
  for (int i = 0; i < 3; ++i) g_scaling_raw[3 * (size_t)g + i] = grs[i];
  float4 gr = make_float4(0.f, 0.f, 0.f, 0.f);
  if (g_rotations) gr = *reinterpret_cast<const float4*>(g_rotations + 4 * (size_t)g);
  *reinterpret_cast<float4*>(g_rotation_raw + 4 * (size_t)g) =
      act_rotation_backward(*reinterpret_cast<const float4*>(rotation_raw + 4 * (size_t)g), gr);
}

}  // namespace sfgs

using namespace sfgs;

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
    SFGS_ACT_DISPATCH(f64_mask, SFGS_LAUNCH_PF);
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
    SFGS_ACT_DISPATCH(f64_mask, SFGS_LAUNCH_PB);
#undef SFGS_LAUNCH_PB
  }
  SFGS_POST_LAUNCH("prepass_bwd", stream, 0);
  return SFGS_OK;
}
