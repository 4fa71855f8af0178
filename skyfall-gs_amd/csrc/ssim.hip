// ssim.hip -- fused SSIM forward/backward for gfx950. Replaces fused_ssim.fused_ssim(img1, img2)
// (reference call sites train.py:222,778; semantics == utils/loss_utils.py:23-63: 11x11 Gaussian window
// sigma 1.5, zero "same" padding, C1 = 0.01^2, C2 = 0.03^2, mean over all elements, gradient w.r.t. img1).
//
// One 32x32 output tile per 256-thread workgroup: the 42x42 input halos of both images are staged in LDS, the
// separable window runs as a horizontal pass into LDS followed by a vertical pass, both with REGISTER sliding windows
// (a thread produces 4 neighbouring outputs from 14 staged values: 3.5 LDS reads per output and moment instead of 11);
// 5 moments: mu1, mu2, E[x^2], E[y^2], E[xy]. Every output is the same ascending-k fma chain as a direct 11-tap sum.
// HBM traffic: forward reads 2 planes (x 1.7 halo) and writes 3 partial-derivative maps (training) -- backward reads
// those 3 maps + 2 planes, writes 1. The mean is reduced without float atomics (per-block partials + a fixed-order
// final sum), so the loss is bit-reproducible.
#include "sfgs_internal.h"

namespace sfgs {

__constant__ float SSIM_W[11] = {1.028380124e-03f, 7.598758209e-03f, 3.600077331e-02f, 1.093606874e-01f,
                                 2.130055279e-01f, 2.660117149e-01f, 2.130055279e-01f, 1.093606874e-01f,
                                 3.600077331e-02f, 7.598758209e-03f, 1.028380124e-03f};
constexpr int ST = 32, SHALO = 5, SIN = ST + 2 * SHALO;  // 42
constexpr int SPITCH = SIN + 2;                          // LDS row pitch of the staged inputs
constexpr int SQ = 4;                                    // outputs per thread and pass (sliding window of SQ + 10)
constexpr float SSIM_C1 = 0.01f * 0.01f, SSIM_C2 = 0.03f * 0.03f;

__device__ __forceinline__ float block_sum_256(float v, float* smem) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
  const int tid = threadIdx.x;
  if ((tid & 63) == 0) smem[tid >> 6] = v;
  __syncthreads();
  return smem[0] + smem[1] + smem[2] + smem[3];
}

// out[q] = sum_k W[k] * v[q + k], k ascending (the order of a direct 11-tap sum)
__device__ __forceinline__ void window4(const float (&v)[SQ + 10], float (&out)[SQ]) {
#pragma unroll
  for (int q = 0; q < SQ; ++q) {
    float a = 0.f;
#pragma unroll
    for (int k = 0; k < 11; ++k) a = fmaf(SSIM_W[k], v[q + k], a);
    out[q] = a;
  }
}

__global__ void __launch_bounds__(256)
ssim_fwd_kernel(const float* __restrict__ img1, const float* __restrict__ img2, int H, int W,
                float* __restrict__ ssim_map, float* __restrict__ block_partials, float* __restrict__ dm_dmu1,
                float* __restrict__ dm_dsig1, float* __restrict__ dm_dsig12) {
  // the staged inputs and the horizontal-pass results share LDS (the results are held in registers across the
  // barrier that retires the inputs): 26.9 KB per workgroup -> five workgroups per CU instead of three
  __shared__ float smem[5 * SIN * ST];
  __shared__ float red[4];
  float (*s1)[SPITCH] = reinterpret_cast<float (*)[SPITCH]>(smem);
  float (*s2)[SPITCH] = reinterpret_cast<float (*)[SPITCH]>(smem + SIN * SPITCH);
  float (*hz)[SIN][ST] = reinterpret_cast<float (*)[SIN][ST]>(smem);
  static_assert(2 * SIN * SPITCH <= 5 * SIN * ST, "inputs must fit under the results");
  const int plane = blockIdx.z;
  const size_t poff = (size_t)plane * H * W;
  const int x0 = blockIdx.x * ST, y0 = blockIdx.y * ST;
  const int tid = threadIdx.x;
  for (int i = tid; i < SIN * SIN; i += 256) {
    const int ly = i / SIN, lx = i - ly * SIN;
    const int gy = y0 + ly - SHALO, gx = x0 + lx - SHALO;
    const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
    s1[ly][lx] = in ? img1[poff + (size_t)gy * W + gx] : 0.f;
    s2[ly][lx] = in ? img2[poff + (size_t)gy * W + gx] : 0.f;
  }
  __syncthreads();
  // horizontal pass: item = (row, group of SQ output columns); SIN * ST / SQ = 336 items = at most 2 per thread
  constexpr int HITEMS = SIN * (ST / SQ), HROUNDS = (HITEMS + 255) / 256;
  float ho[HROUNDS][5][SQ];
#pragma unroll
  for (int r = 0; r < HROUNDS; ++r) {
    const int i = tid + 256 * r;
    if (i < HITEMS) {
      const int ly = i / (ST / SQ), lx = (i - ly * (ST / SQ)) * SQ;
      float a[SQ + 10], b[SQ + 10], t[SQ + 10];
#pragma unroll
      for (int k = 0; k < SQ + 10; ++k) { a[k] = s1[ly][lx + k]; b[k] = s2[ly][lx + k]; }
      window4(a, ho[r][0]);
      window4(b, ho[r][1]);
#pragma unroll
      for (int k = 0; k < SQ + 10; ++k) t[k] = a[k] * a[k];
      window4(t, ho[r][2]);
#pragma unroll
      for (int k = 0; k < SQ + 10; ++k) t[k] = b[k] * b[k];
      window4(t, ho[r][3]);
#pragma unroll
      for (int k = 0; k < SQ + 10; ++k) t[k] = a[k] * b[k];
      window4(t, ho[r][4]);
    }
  }
  __syncthreads();   // every read of s1 / s2 is done: their space becomes hz
#pragma unroll
  for (int r = 0; r < HROUNDS; ++r) {
    const int i = tid + 256 * r;
    if (i < HITEMS) {
      const int ly = i / (ST / SQ), lx = (i - ly * (ST / SQ)) * SQ;
#pragma unroll
      for (int m = 0; m < 5; ++m)
#pragma unroll
        for (int q = 0; q < SQ; ++q) hz[m][ly][lx + q] = ho[r][m][q];
    }
  }
  __syncthreads();
  // vertical pass: thread = (column, group of SQ output rows)
  const int lx = tid & (ST - 1), ly0 = (tid / ST) * SQ;
  float mo[5][SQ];
#pragma unroll
  for (int m = 0; m < 5; ++m) {
    float v[SQ + 10];
#pragma unroll
    for (int k = 0; k < SQ + 10; ++k) v[k] = hz[m][ly0 + k][lx];
    window4(v, mo[m]);
  }
  const int gx = x0 + lx;
  float vsum = 0.f;
#pragma unroll
  for (int q = 0; q < SQ; ++q) {
    const int gy = y0 + ly0 + q;
    if (gx < W && gy < H) {
      const float mu1 = mo[0][q], mu2 = mo[1][q], e11 = mo[2][q], e22 = mo[3][q], e12 = mo[4][q];
      const float mu1sq = mu1 * mu1, mu2sq = mu2 * mu2, mu12 = mu1 * mu2;
      const float sg1 = e11 - mu1sq, sg2 = e22 - mu2sq, sg12 = e12 - mu12;
      const float A1 = 2.f * mu12 + SSIM_C1, A2 = 2.f * sg12 + SSIM_C2;
      const float B1 = mu1sq + mu2sq + SSIM_C1, B2 = sg1 + sg2 + SSIM_C2;
      const float inv = 1.0f / (B1 * B2);
      const float val = A1 * A2 * inv;
      vsum += val;
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
  }
  const float bs = block_sum_256(vsum, red);
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
  __shared__ float s[3][SIN][SPITCH];
  __shared__ float hz[3][SIN][ST];
  const int plane = blockIdx.z;
  const size_t poff = (size_t)plane * H * W;
  const int x0 = blockIdx.x * ST, y0 = blockIdx.y * ST;
  const int tid = threadIdx.x;
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
  for (int i = tid; i < SIN * (ST / SQ); i += 256) {
    const int ly = i / (ST / SQ), lx = (i - ly * (ST / SQ)) * SQ;
#pragma unroll
    for (int m = 0; m < 3; ++m) {
      float v[SQ + 10], o[SQ];
#pragma unroll
      for (int k = 0; k < SQ + 10; ++k) v[k] = s[m][ly][lx + k];
      window4(v, o);
#pragma unroll
      for (int q = 0; q < SQ; ++q) hz[m][ly][lx + q] = o[q];
    }
  }
  __syncthreads();
  const int lx = tid & (ST - 1), ly0 = (tid / ST) * SQ;
  float mo[3][SQ];
#pragma unroll
  for (int m = 0; m < 3; ++m) {
    float v[SQ + 10];
#pragma unroll
    for (int k = 0; k < SQ + 10; ++k) v[k] = hz[m][ly0 + k][lx];
    window4(v, mo[m]);
  }
  const int gx = x0 + lx;
  const float scale = dL_dmean[0] * inv_count;
#pragma unroll
  for (int q = 0; q < SQ; ++q) {
    const int gy = y0 + ly0 + q;
    if (gx < W && gy < H) {
      const size_t idx = poff + (size_t)gy * W + gx;
      const float p1 = img1[idx], p2 = img2[idx];
      dL_dimg1[idx] = scale * (mo[0][q] + 2.f * p1 * mo[1][q] + p2 * mo[2][q]);
    }
  }
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
  const dim3 grid((W + ST - 1) / ST, (H + ST - 1) / ST, B * C), block(256);
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
  const dim3 grid((W + ST - 1) / ST, (H + ST - 1) / ST, B * C), block(256);
  { ProfScope ps_(KID_SSIM_BWD, stream);
    hipLaunchKernelGGL(ssim_bwd_kernel, grid, block, 0, stream, img1, img2, H, W, (const float*)maps,
                       (const float*)(maps + plane), (const float*)(maps + 2 * plane), dL_dmean,
                       1.0f / (float)((double)B * C * H * W), dL_dimg1); }
  SFGS_POST_LAUNCH("ssim_bwd", stream, 0);
  return SFGS_OK;
}
