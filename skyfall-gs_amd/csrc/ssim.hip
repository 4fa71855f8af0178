// ssim.hip -- fused SSIM forward/backward for gfx950. Replaces fused_ssim.fused_ssim(img1, img2)
// (reference call sites train.py:222,778; semantics == utils/loss_utils.py:23-63: 11x11 Gaussian window
// sigma 1.5, zero "same" padding, C1 = 0.01^2, C2 = 0.03^2, mean over all elements, gradient w.r.t. img1).
//
// One 16x16 output tile per 256-thread workgroup: the 26x26 input halos of both images are staged in
// LDS, the separable window runs as a horizontal pass into LDS followed by a vertical pass in
// registers (5 moments: mu1, mu2, E[x^2], E[y^2], E[xy]). HBM traffic: forward reads 2 planes and
// writes 3 partial-derivative maps (training) -- backward reads those 3 maps + 2 planes, writes 1.
// The mean is reduced without float atomics (per-block partials + a fixed-order final sum), so the
// loss is bit-reproducible.
#include "sfgs_internal.h"

namespace sfgs {

__constant__ float SSIM_W[11] = {1.028380124e-03f, 7.598758209e-03f, 3.600077331e-02f, 1.093606874e-01f,
                                 2.130055279e-01f, 2.660117149e-01f, 2.130055279e-01f, 1.093606874e-01f,
                                 3.600077331e-02f, 7.598758209e-03f, 1.028380124e-03f};
constexpr int ST = 16, SHALO = 5, SIN = ST + 2 * SHALO;  // 26
constexpr float SSIM_C1 = 0.01f * 0.01f, SSIM_C2 = 0.03f * 0.03f;

__device__ __forceinline__ float block_sum_256(float v, float* smem) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
  const int tid = threadIdx.y * blockDim.x + threadIdx.x;
  if ((tid & 63) == 0) smem[tid >> 6] = v;
  __syncthreads();
  return smem[0] + smem[1] + smem[2] + smem[3];
}

__global__ void __launch_bounds__(256)
ssim_fwd_kernel(const float* __restrict__ img1, const float* __restrict__ img2, int H, int W,
                float* __restrict__ ssim_map, float* __restrict__ block_partials, float* __restrict__ dm_dmu1,
                float* __restrict__ dm_dsig1, float* __restrict__ dm_dsig12) {
  __shared__ float s1[SIN][SIN + 1], s2[SIN][SIN + 1];
  __shared__ float hz[5][SIN][ST];
  __shared__ float red[4];
  const int plane = blockIdx.z;
  const size_t poff = (size_t)plane * H * W;
  const int x0 = blockIdx.x * ST, y0 = blockIdx.y * ST;
  const int tid = threadIdx.y * ST + threadIdx.x;
  for (int i = tid; i < SIN * SIN; i += 256) {
    const int ly = i / SIN, lx = i - ly * SIN;
    const int gy = y0 + ly - SHALO, gx = x0 + lx - SHALO;
    const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
    s1[ly][lx] = in ? img1[poff + (size_t)gy * W + gx] : 0.f;
    s2[ly][lx] = in ? img2[poff + (size_t)gy * W + gx] : 0.f;
  }
  __syncthreads();
  for (int i = tid; i < SIN * ST; i += 256) {
    const int ly = i / ST, lx = i - ly * ST;
    float m1 = 0.f, m2 = 0.f, e11 = 0.f, e22 = 0.f, e12 = 0.f;
#pragma unroll
    for (int k = 0; k < 11; ++k) {
      const float w = SSIM_W[k], a = s1[ly][lx + k], b = s2[ly][lx + k];
      m1 = fmaf(w, a, m1); m2 = fmaf(w, b, m2);
      e11 = fmaf(w, a * a, e11); e22 = fmaf(w, b * b, e22); e12 = fmaf(w, a * b, e12);
    }
    hz[0][ly][lx] = m1; hz[1][ly][lx] = m2; hz[2][ly][lx] = e11; hz[3][ly][lx] = e22; hz[4][ly][lx] = e12;
  }
  __syncthreads();
  const int lx = threadIdx.x, ly = threadIdx.y;
  float mu1 = 0.f, mu2 = 0.f, e11 = 0.f, e22 = 0.f, e12 = 0.f;
#pragma unroll
  for (int k = 0; k < 11; ++k) {
    const float w = SSIM_W[k];
    mu1 = fmaf(w, hz[0][ly + k][lx], mu1); mu2 = fmaf(w, hz[1][ly + k][lx], mu2);
    e11 = fmaf(w, hz[2][ly + k][lx], e11); e22 = fmaf(w, hz[3][ly + k][lx], e22);
    e12 = fmaf(w, hz[4][ly + k][lx], e12);
  }
  const int gx = x0 + lx, gy = y0 + ly;
  float val = 0.f;
  if (gx < W && gy < H) {
    const float mu1sq = mu1 * mu1, mu2sq = mu2 * mu2, mu12 = mu1 * mu2;
    const float sg1 = e11 - mu1sq, sg2 = e22 - mu2sq, sg12 = e12 - mu12;
    const float A1 = 2.f * mu12 + SSIM_C1, A2 = 2.f * sg12 + SSIM_C2;
    const float B1 = mu1sq + mu2sq + SSIM_C1, B2 = sg1 + sg2 + SSIM_C2;
    const float inv = 1.0f / (B1 * B2);
    val = A1 * A2 * inv;
    const size_t idx = poff + (size_t)gy * W + gx;
    if (ssim_map) ssim_map[idx] = val;
    if (dm_dmu1) {
      // partials w.r.t. the three convolution outputs that depend on img1: mu1, E[x^2], E[xy]
      const float d_sig1 = -val / B2;            // d/d sigma1_sq
      const float d_sig12 = 2.f * A1 * inv;      // d/d sigma12
      dm_dmu1[idx] = 2.f * mu2 * A2 * inv - 2.f * mu1 * val / B1 - 2.f * mu1 * d_sig1 - mu2 * d_sig12;
      dm_dsig1[idx] = d_sig1;
      dm_dsig12[idx] = d_sig12;
    }
  }
  const float bs = block_sum_256(val, red);
  if (tid == 0) block_partials[((size_t)plane * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = bs;
}

// fixed-order final reduction: mean = sum(partials) / count
__global__ void __launch_bounds__(1024) ssim_mean_kernel(const float* __restrict__ partials, int n, float inv_count,
                                                         float* __restrict__ out) {
  __shared__ double sm[16];
  double acc = 0.0;
  for (int i = threadIdx.x; i < n; i += 1024) acc += (double)partials[i];
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int w = 0; w < 16; ++w) t += sm[w];
    out[0] = (float)(t * (double)inv_count);
  }
}

__global__ void __launch_bounds__(256)
ssim_bwd_kernel(const float* __restrict__ img1, const float* __restrict__ img2, int H, int W,
                const float* __restrict__ dm_dmu1, const float* __restrict__ dm_dsig1,
                const float* __restrict__ dm_dsig12, const float* __restrict__ dL_dmean, float inv_count,
                float* __restrict__ dL_dimg1) {
  __shared__ float s[3][SIN][SIN + 1];
  __shared__ float hz[3][SIN][ST];
  const int plane = blockIdx.z;
  const size_t poff = (size_t)plane * H * W;
  const int x0 = blockIdx.x * ST, y0 = blockIdx.y * ST;
  const int tid = threadIdx.y * ST + threadIdx.x;
  for (int i = tid; i < SIN * SIN; i += 256) {
    const int ly = i / SIN, lx = i - ly * SIN;
    const int gy = y0 + ly - SHALO, gx = x0 + lx - SHALO;
    const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
    const size_t idx = poff + (size_t)gy * W + gx;
    s[0][ly][lx] = in ? dm_dmu1[idx] : 0.f;
    s[1][ly][lx] = in ? dm_dsig1[idx] : 0.f;
    s[2][ly][lx] = in ? dm_dsig12[idx] : 0.f;
  }
  __syncthreads();
  for (int i = tid; i < SIN * ST; i += 256) {
    const int ly = i / ST, lx = i - ly * ST;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
#pragma unroll
    for (int k = 0; k < 11; ++k) {
      const float w = SSIM_W[k];
      a0 = fmaf(w, s[0][ly][lx + k], a0); a1 = fmaf(w, s[1][ly][lx + k], a1); a2 = fmaf(w, s[2][ly][lx + k], a2);
    }
    hz[0][ly][lx] = a0; hz[1][ly][lx] = a1; hz[2][ly][lx] = a2;
  }
  __syncthreads();
  const int lx = threadIdx.x, ly = threadIdx.y;
  const int gx = x0 + lx, gy = y0 + ly;
  if (gx >= W || gy >= H) return;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f;
#pragma unroll
  for (int k = 0; k < 11; ++k) {
    const float w = SSIM_W[k];
    a0 = fmaf(w, hz[0][ly + k][lx], a0); a1 = fmaf(w, hz[1][ly + k][lx], a1); a2 = fmaf(w, hz[2][ly + k][lx], a2);
  }
  const size_t idx = poff + (size_t)gy * W + gx;
  const float p1 = img1[idx], p2 = img2[idx];
  dL_dimg1[idx] = dL_dmean[0] * inv_count * (a0 + 2.f * p1 * a1 + p2 * a2);
}

}  // namespace sfgs

using namespace sfgs;

static inline size_t ssim_nblocks(int B, int C, int H, int W) {
  return (size_t)B * C * ((H + ST - 1) / ST) * ((W + ST - 1) / ST);
}

extern "C" size_t sfgs_ssim_scratch_bytes(int32_t B, int32_t C, int32_t H, int32_t W, int32_t with_grad) {
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0) return 0;
  size_t b = align_up(ssim_nblocks(B, C, H, W) * 4, 256);
  if (with_grad) b += 3 * align_up((size_t)B * C * H * W * 4, 256);
  return b;
}

extern "C" int sfgs_ssim_forward(const float* img1, const float* img2, int32_t B, int32_t C, int32_t H, int32_t W,
                                 float* ssim_map_or_null, float* ssim_mean, void* scratch, size_t scratch_sz,
                                 int32_t with_grad, void* stream_) {
  SFGS_REQUIRE(img1 && img2 && ssim_mean && scratch, SFGS_E_ARG, "NULL argument");
  SFGS_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, SFGS_E_ARG, "bad image shape [%d,%d,%d,%d]", B, C, H, W);
  SFGS_REQUIRE((int64_t)B * C <= 65535, SFGS_E_UNSUPPORTED, "B*C > 65535");
  SFGS_REQUIRE(scratch_sz >= sfgs_ssim_scratch_bytes(B, C, H, W, with_grad), SFGS_E_CAPACITY, "ssim scratch too small");
  hipStream_t stream = (hipStream_t)stream_;
  const size_t nblk = ssim_nblocks(B, C, H, W), plane = align_up((size_t)B * C * H * W * 4, 256);
  float* partials = (float*)scratch;
  char* maps = (char*)scratch + align_up(nblk * 4, 256);
  float* m0 = with_grad ? (float*)maps : nullptr;
  float* m1 = with_grad ? (float*)(maps + plane) : nullptr;
  float* m2 = with_grad ? (float*)(maps + 2 * plane) : nullptr;
  const dim3 grid((W + ST - 1) / ST, (H + ST - 1) / ST, B * C), block(ST, ST);
  { ProfScope ps_(KID_SSIM_FWD, stream);
    hipLaunchKernelGGL(ssim_fwd_kernel, grid, block, 0, stream, img1, img2, H, W, ssim_map_or_null, partials, m0, m1, m2); }
  SFGS_POST_LAUNCH("ssim_fwd", stream, 0);
  { ProfScope ps_(KID_SSIM_MEAN, stream);
    hipLaunchKernelGGL(ssim_mean_kernel, dim3(1), dim3(1024), 0, stream, partials, (int)nblk,
                       1.0f / (float)((double)B * C * H * W), ssim_mean); }
  SFGS_POST_LAUNCH("ssim_mean", stream, 0);
  return SFGS_OK;
}

extern "C" int sfgs_ssim_backward(const float* img1, const float* img2, int32_t B, int32_t C, int32_t H, int32_t W,
                                  const void* scratch, const float* dL_dmean, float* dL_dimg1, void* stream_) {
  SFGS_REQUIRE(img1 && img2 && scratch && dL_dmean && dL_dimg1, SFGS_E_ARG, "NULL argument");
  SFGS_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, SFGS_E_ARG, "bad image shape");
  hipStream_t stream = (hipStream_t)stream_;
  const size_t nblk = ssim_nblocks(B, C, H, W), plane = align_up((size_t)B * C * H * W * 4, 256);
  const char* maps = (const char*)scratch + align_up(nblk * 4, 256);
  const dim3 grid((W + ST - 1) / ST, (H + ST - 1) / ST, B * C), block(ST, ST);
  { ProfScope ps_(KID_SSIM_BWD, stream);
    hipLaunchKernelGGL(ssim_bwd_kernel, grid, block, 0, stream, img1, img2, H, W, (const float*)maps,
                       (const float*)(maps + plane), (const float*)(maps + 2 * plane), dL_dmean,
                       1.0f / (float)((double)B * C * H * W), dL_dimg1); }
  SFGS_POST_LAUNCH("ssim_bwd", stream, 0);
  return SFGS_OK;
}
