// knn.hip -- simple_knn._C.distCUDA2 for gfx950 (reference call site scene/gaussian_model.py:324):
// out[i] = mean squared distance from point i to its 3 nearest OTHER points (index-excluded).
//
// Exact brute force, tiled through LDS: a workgroup owns 256 query points (one per lane) and streams
// all N points in 1024-point LDS tiles read with uniform (broadcast) addresses; each lane keeps its
// three smallest squared distances in registers. O(N^2) VALU work, no scratch, no sort: the call
// happens once per training run on the SfM cloud (1e4..1e6 points), where this is milliseconds to a
// fraction of a second on 256 CUs. N < 4 averages over the neighbours that exist.
#include "sfgs_internal.h"

namespace sfgs {

constexpr int KNN_BLOCK = 256, KNN_TILE = 1024;

__global__ void __launch_bounds__(KNN_BLOCK)
knn_dist2_kernel(const float* __restrict__ xyz, int N, float* __restrict__ out) {
  __shared__ float4 tile[KNN_TILE];
  const int i = blockIdx.x * KNN_BLOCK + threadIdx.x;
  float px = 0.f, py = 0.f, pz = 0.f;
  if (i < N) { px = xyz[3 * (size_t)i]; py = xyz[3 * (size_t)i + 1]; pz = xyz[3 * (size_t)i + 2]; }
  float b0 = INFINITY, b1 = INFINITY, b2 = INFINITY;
  for (int base = 0; base < N; base += KNN_TILE) {
    const int cnt = min(KNN_TILE, N - base);
    __syncthreads();
    for (int t = threadIdx.x; t < cnt; t += KNN_BLOCK) {
      const size_t j = (size_t)(base + t) * 3;
      tile[t] = make_float4(xyz[j], xyz[j + 1], xyz[j + 2], 0.f);
    }
    __syncthreads();
    const int self = i - base;  // position of the query itself inside this tile (if any)
    for (int t = 0; t < cnt; ++t) {
      const float4 q = tile[t];
      const float dx = q.x - px, dy = q.y - py, dz = q.z - pz;
      float d = dx * dx + dy * dy + dz * dz;
      d = (t == self) ? INFINITY : d;
      if (d < b2) {
        if (d < b0) { b2 = b1; b1 = b0; b0 = d; }
        else if (d < b1) { b2 = b1; b1 = d; }
        else b2 = d;
      }
    }
  }
  if (i < N) {
    float sum = 0.f;
    int c = 0;
    if (b0 < INFINITY) { sum += b0; ++c; }
    if (b1 < INFINITY) { sum += b1; ++c; }
    if (b2 < INFINITY) { sum += b2; ++c; }
    out[i] = c ? sum / (float)c : 0.f;
  }
}

}  // namespace sfgs

using namespace sfgs;

extern "C" size_t sfgs_knn_scratch_bytes(int32_t N) { (void)N; return 0; }

extern "C" int sfgs_knn_dist2(const float* xyz, int32_t N, float* out, void* scratch, size_t scratch_sz, void* stream_) {
  (void)scratch; (void)scratch_sz;
  SFGS_REQUIRE(N >= 0, SFGS_E_ARG, "negative point count");
  if (N == 0) return SFGS_OK;
  SFGS_REQUIRE(xyz && out, SFGS_E_ARG, "NULL argument");
  hipStream_t stream = (hipStream_t)stream_;
  hipLaunchKernelGGL(knn_dist2_kernel, dim3((N + KNN_BLOCK - 1) / KNN_BLOCK), dim3(KNN_BLOCK), 0, stream, xyz, N, out);
  SFGS_POST_LAUNCH("knn_dist2", stream, 0);
  return SFGS_OK;
}
