// filter3d.hip -- GaussianModel.compute_3D_filter as two kernels (Mip-Splatting 3D filter).
//
// Reference: scene/gaussian_model.py:255-308 -- a Python loop over the training cameras that, per camera, runs
// ~12 float64 torch kernels over all N points (transform, depth test z > 0.2, projection with the camera's
// focal / principal point, 15 % screen margin, running min of z), then filter_3D = min_z / max_focal * sqrt(0.2)
// with never-seen points set to the largest seen distance. Here: one thread per Gaussian loops over the cameras
// (18 doubles each, read through the scalar cache), float64 throughout as in the reference.
#include "sfgs_internal.h"

namespace sfgs {

constexpr int CAM_DOUBLES = 18;  // R[9] (as stored by the reference: xyz @ R), T[3], focal_x, focal_y, cx_ori, cy_ori, W, H

__global__ void __launch_bounds__(256)
filter3d_min_depth_kernel(int N, const float* __restrict__ xyz, int C, const double* __restrict__ cams,
                          double* __restrict__ dist, double* __restrict__ block_max) {
  __shared__ double s_max[4];
  const int g = blockIdx.x * 256 + threadIdx.x;
  double d = 1e8;
  bool seen = false;
  if (g < N) {
    const double x = (double)xyz[3 * (size_t)g], y = (double)xyz[3 * (size_t)g + 1], z = (double)xyz[3 * (size_t)g + 2];
    for (int c = 0; c < C; ++c) {
      const double* k = cams + (size_t)c * CAM_DOUBLES;
      // xyz_cam = xyz @ R + T  (row vector times matrix)
      const double xc = x * k[0] + y * k[3] + z * k[6] + k[9];
      const double yc = x * k[1] + y * k[4] + z * k[7] + k[10];
      const double zc = x * k[2] + y * k[5] + z * k[8] + k[11];
      const bool valid_depth = zc > 0.2;
      const double zz = zc < 0.001 ? 0.001 : zc;
      const double px = xc / zz * k[12] + k[14];
      const double py = yc / zz * k[13] + k[15];
      const double W = k[16], H = k[17];
      const bool in_screen = px >= -0.15 * W && px <= W * 1.15 && py >= -0.15 * H && py <= 1.15 * H;
      if (valid_depth && in_screen) { d = zz < d ? zz : d; seen = true; }
    }
    dist[g] = seen ? d : -1.0;  // -1 marks "never seen"
  }
  // largest seen distance of the block
  double m = (g < N && seen) ? d : -1.0;
#pragma unroll
  for (int s = 32; s >= 1; s >>= 1) { const double o = __shfl_xor(m, s); m = o > m ? o : m; }
  if ((threadIdx.x & 63) == 0) s_max[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) block_max[blockIdx.x] = fmax(fmax(s_max[0], s_max[1]), fmax(s_max[2], s_max[3]));
}

__global__ void __launch_bounds__(1024)
filter3d_reduce_kernel(int NB, double* __restrict__ block_max) {
  __shared__ double sm[16];
  double m = -1.0;
  for (int i = threadIdx.x; i < NB; i += 1024) m = fmax(m, block_max[i]);
#pragma unroll
  for (int s = 32; s >= 1; s >>= 1) m = fmax(m, __shfl_xor(m, s));
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = -1.0;
    for (int w = 0; w < 16; ++w) t = fmax(t, sm[w]);
    block_max[0] = t;
  }
}

__global__ void __launch_bounds__(256)
filter3d_finish_kernel(int N, const double* __restrict__ dist, const double* __restrict__ max_seen, double focal,
                       double* __restrict__ out) {
  const int g = blockIdx.x * 256 + threadIdx.x;
  if (g >= N) return;
  // No camera sees ANY point: the reference raises here (`distance[valid_points].max()` of an empty tensor,
  // scene/gaussian_model.py:299); an asynchronous kernel cannot, so such points keep the reference's initial
  // distance of 1e8 (:260) instead of a negative filter size.
  const double far = max_seen[0] < 0.0 ? 1e8 : max_seen[0];
  const double d = dist[g] < 0.0 ? far : dist[g];
  out[g] = d / focal * 0.4472135954999579;  // 0.2 ** 0.5 as Python evaluates it
}

}  // namespace sfgs

using namespace sfgs;

extern "C" size_t sfgs_filter3d_scratch_bytes(int32_t N) {
  if (N <= 0) return 0;
  return align_up((size_t)N * 8, 256) + align_up((size_t)((N + 255) / 256) * 8, 256);
}

extern "C" int sfgs_filter3d(const float* xyz, int32_t N, const double* cams, int32_t C, double max_focal,
                             double* filter_out, void* scratch, size_t scratch_sz, void* stream_) {
  SFGS_REQUIRE(N >= 0 && C >= 0, SFGS_E_ARG, "negative count");
  if (N == 0) return SFGS_OK;
  SFGS_REQUIRE(xyz && filter_out && scratch && (C == 0 || cams), SFGS_E_ARG, "NULL argument");
  SFGS_REQUIRE(scratch_sz >= sfgs_filter3d_scratch_bytes(N), SFGS_E_CAPACITY, "filter3d scratch too small");
  SFGS_REQUIRE(max_focal > 0.0, SFGS_E_ARG, "max_focal must be positive");
  hipStream_t stream = (hipStream_t)stream_;
  double* dist = (double*)scratch;
  double* bmax = (double*)((char*)scratch + align_up((size_t)N * 8, 256));
  const int NB = (N + 255) / 256;
  { ProfScope ps_(KID_FILTER3D, stream);
    hipLaunchKernelGGL(filter3d_min_depth_kernel, dim3(NB), dim3(256), 0, stream, N, xyz, C, cams, dist, bmax);
    hipLaunchKernelGGL(filter3d_reduce_kernel, dim3(1), dim3(1024), 0, stream, NB, bmax);
    hipLaunchKernelGGL(filter3d_finish_kernel, dim3(NB), dim3(256), 0, stream, N, dist, bmax, max_focal, filter_out); }
  SFGS_POST_LAUNCH("filter3d", stream, 0);
  return SFGS_OK;
}
