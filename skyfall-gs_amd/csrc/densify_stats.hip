// densify_stats.hip -- GaussianModel.add_densification_stats as one in-place kernel (the
// per-step part). Reference: scene/gaussian_model.py:744-749
//   xyz_gradient_accum[f]         += || viewspace.grad[f, :2] ||
//   xyz_gradient_accum_abs[f]     += || viewspace.grad[f, 2:] ||
//   xyz_gradient_accum_abs_max[f]  = max(., || viewspace.grad[f, 2:] ||)
//   denom[f]                      += 1
// with f = update_filter (bool [N]). In torch these are boolean-mask gathers / scatters: ~20 kernels and several
// host syncs (mask -> index conversion) per training step. Here: one thread per Gaussian, no sync.
#include "sfgs_internal.h"

namespace sfgs {

__global__ void __launch_bounds__(256)
densify_stats_kernel(int N, const float* __restrict__ vs_grad /* [N,3] */, const unsigned char* __restrict__ filter,
                     float* __restrict__ accum, float* __restrict__ accum_abs, float* __restrict__ accum_abs_max,
                     float* __restrict__ denom) {
  const int g = blockIdx.x * 256 + threadIdx.x;
  if (g >= N || !filter[g]) return;
  const float gx = vs_grad[3 * (size_t)g], gy = vs_grad[3 * (size_t)g + 1], ga = vs_grad[3 * (size_t)g + 2];
  accum[g] += sqrtf(gx * gx + gy * gy);   // torch.norm over 2 elements
  const float na = fabsf(ga);             // torch.norm over the single "abs" column
  accum_abs[g] += na;
  if (accum_abs_max) accum_abs_max[g] = fmaxf(accum_abs_max[g], na);
  denom[g] += 1.0f;
}

}  // namespace sfgs

using namespace sfgs;

extern "C" int sfgs_densify_stats(int32_t N, const float* viewspace_grad, const unsigned char* update_filter,
                                  float* xyz_gradient_accum, float* xyz_gradient_accum_abs,
                                  float* xyz_gradient_accum_abs_max_or_null, float* denom, void* stream_) {
  SFGS_REQUIRE(N >= 0, SFGS_E_ARG, "negative Gaussian count");
  if (N == 0) return SFGS_OK;
  SFGS_REQUIRE(viewspace_grad && update_filter && xyz_gradient_accum && xyz_gradient_accum_abs && denom, SFGS_E_ARG,
               "NULL argument");
  hipStream_t stream = (hipStream_t)stream_;
  { ProfScope ps_(KID_DENSIFY_STATS, stream);
    hipLaunchKernelGGL(densify_stats_kernel, dim3((N + 255) / 256), dim3(256), 0, stream, N, viewspace_grad,
                       update_filter, xyz_gradient_accum, xyz_gradient_accum_abs, xyz_gradient_accum_abs_max_or_null,
                       denom); }
  SFGS_POST_LAUNCH("densify_stats", stream, 0);
  return SFGS_OK;
}
