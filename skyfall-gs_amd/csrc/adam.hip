// adam.hip -- one-launch multi-tensor Adam step for the per-Gaussian parameter groups.
// Reference: scene/gaussian_model.py:382 `torch.optim.Adam(l, lr=0.0, eps=1e-15)` stepped every iteration at
// train.py:339,906. torch's default CUDA path for that call is the "foreach" implementation
// (torch/optim/adam.py::_multi_tensor_adam): eight elementwise passes over every parameter, i.e. ~ 25 float
// reads/writes per element. The arithmetic below follows that path operation by operation (same order, same f32
// rounding points, scalars converted from the host's double exactly as the foreach kernels do):
//     g' = g + wd * p                                 (only when weight_decay != 0)
//     m  = m + (1-b1) * (g' - m)                      (_foreach_lerp_, weight < 0.5 branch)
//     v  = v * b2 ;  v = v + (1-b2) * g' * g'         (_foreach_mul_, _foreach_addcmul_)
//     d  = sqrt(v) / sqrt(1-b2^t) + eps               (_foreach_sqrt, _foreach_div_, _foreach_add_)
//     p  = p + (-lr/(1-b1^t)) * (m / d)               (_foreach_addcdiv_)
// HBM-bound: 16 B read + 12 B written per element; one block streams 4096 consecutive elements of one tensor.
#include "sfgs_internal.h"

namespace sfgs {

constexpr int ADAM_MAX_TENSORS = 24;    // per launch (kernel-argument table); longer lists are split
constexpr int ADAM_BLOCK = 256;
constexpr int ADAM_VEC_PER_THREAD = 4;  // float4s per thread
constexpr int ADAM_CHUNK = ADAM_BLOCK * ADAM_VEC_PER_THREAD * 4;

struct AdamTable {
  SfgsAdamTensor t[ADAM_MAX_TENSORS];
  unsigned block_end[ADAM_MAX_TENSORS];  // exclusive prefix: blocks [block_end[i-1], block_end[i]) work on tensor i
  int vec_ok[ADAM_MAX_TENSORS];          // all four pointers 16-B aligned
  int count;
};

template <typename T>
struct AdamScalars { T wd, w1, b2, w2, bc2s, eps, step; };

template <typename T>
__device__ __forceinline__ T sqrt_of(T v);
template <> __device__ __forceinline__ float sqrt_of<float>(float v) { return sqrtf(v); }
template <> __device__ __forceinline__ double sqrt_of<double>(double v) { return sqrt(v); }

template <typename T>
__device__ __forceinline__ void adam_update(T& p, T g, T& m, T& v, const AdamScalars<T>& s) {
  if (s.wd != (T)0) g = g + s.wd * p;
  m = m + s.w1 * (g - m);
  v = v * s.b2;
  v = v + s.w2 * (g * g);
  const T d = sqrt_of<T>(v) / s.bc2s + s.eps;
  p = p + s.step * (m / d);
}

__global__ void __launch_bounds__(ADAM_BLOCK)
adam_kernel(const AdamTable tab) {
  // block -> tensor (uniform; the table sits in the kernel-argument segment, i.e. scalar loads)
  int ti = 0;
  while (ti + 1 < tab.count && blockIdx.x >= tab.block_end[ti]) ++ti;
  const SfgsAdamTensor& T = tab.t[ti];
  const unsigned first = ti ? tab.block_end[ti - 1] : 0u;
  const int64_t base = (int64_t)(blockIdx.x - first) * ADAM_CHUNK;
  const int64_t n = T.count;
  if (T.flags & SFGS_ADAM_F64) {
    // float64 tensors (the reference's `_opacity` after reset_opacity): torch runs the same sequence in float64
    const AdamScalars<double> s{T.weight_decay, T.one_minus_beta1, T.beta2, T.one_minus_beta2, T.bias_correction2_sqrt,
                                T.eps, T.neg_step_size};
    double* __restrict__ P = (double*)T.param;
    const double* __restrict__ G = (const double*)T.grad;
    double* __restrict__ M = (double*)T.exp_avg;
    double* __restrict__ V = (double*)T.exp_avg_sq;
    for (int64_t e = base + threadIdx.x; e < n && e < base + ADAM_CHUNK; e += ADAM_BLOCK) {
      double p = P[e], m = M[e], v = V[e];
      adam_update<double>(p, G[e], m, v, s);
      P[e] = p; M[e] = m; V[e] = v;
    }
    return;
  }
  // the host formed the scalars in double like torch does; the float32 path rounds them to float as torch's kernels do
  const AdamScalars<float> s{(float)T.weight_decay, (float)T.one_minus_beta1, (float)T.beta2, (float)T.one_minus_beta2,
                             (float)T.bias_correction2_sqrt, (float)T.eps, (float)T.neg_step_size};
  float* __restrict__ P = (float*)T.param;
  const float* __restrict__ G = (const float*)T.grad;
  float* __restrict__ M = (float*)T.exp_avg;
  float* __restrict__ V = (float*)T.exp_avg_sq;

  if (tab.vec_ok[ti] && base + ADAM_CHUNK <= n) {
    float4 p[ADAM_VEC_PER_THREAD], g[ADAM_VEC_PER_THREAD], m[ADAM_VEC_PER_THREAD], v[ADAM_VEC_PER_THREAD];
#pragma unroll
    for (int k = 0; k < ADAM_VEC_PER_THREAD; ++k) {
      const int64_t e = base + ((int64_t)k * ADAM_BLOCK + threadIdx.x) * 4;
      p[k] = *reinterpret_cast<const float4*>(P + e);
      g[k] = *reinterpret_cast<const float4*>(G + e);
      m[k] = *reinterpret_cast<const float4*>(M + e);
      v[k] = *reinterpret_cast<const float4*>(V + e);
    }
#pragma unroll
    for (int k = 0; k < ADAM_VEC_PER_THREAD; ++k) {
      adam_update<float>(p[k].x, g[k].x, m[k].x, v[k].x, s);
      adam_update<float>(p[k].y, g[k].y, m[k].y, v[k].y, s);
      adam_update<float>(p[k].z, g[k].z, m[k].z, v[k].z, s);
      adam_update<float>(p[k].w, g[k].w, m[k].w, v[k].w, s);
      const int64_t e = base + ((int64_t)k * ADAM_BLOCK + threadIdx.x) * 4;
      *reinterpret_cast<float4*>(P + e) = p[k];
      *reinterpret_cast<float4*>(M + e) = m[k];
      *reinterpret_cast<float4*>(V + e) = v[k];
    }
    return;
  }
  // ragged tail / unaligned tensors
  for (int64_t e = base + threadIdx.x; e < n && e < base + ADAM_CHUNK; e += ADAM_BLOCK) {
    float p = P[e], m = M[e], v = V[e];
    adam_update<float>(p, G[e], m, v, s);
    P[e] = p; M[e] = m; V[e] = v;
  }
}

}  // namespace sfgs

using namespace sfgs;

extern "C" int sfgs_adam_step(const SfgsAdamTensor* tensors, int32_t count, void* stream_) {
  SFGS_REQUIRE(count >= 0, SFGS_E_ARG, "negative tensor count");
  SFGS_REQUIRE(count == 0 || tensors, SFGS_E_ARG, "NULL tensor table");
  hipStream_t stream = (hipStream_t)stream_;
  int i = 0;
  while (i < count) {
    AdamTable tab;
    tab.count = 0;
    unsigned blocks = 0;
    for (; i < count && tab.count < ADAM_MAX_TENSORS; ++i) {
      const SfgsAdamTensor& t = tensors[i];
      SFGS_REQUIRE(t.count >= 0, SFGS_E_ARG, "tensor %d: negative element count", i);
      if (t.count == 0) continue;
      SFGS_REQUIRE(t.param && t.grad && t.exp_avg && t.exp_avg_sq, SFGS_E_ARG, "tensor %d: NULL pointer", i);
      const uint64_t nb = (uint64_t)((t.count + ADAM_CHUNK - 1) / ADAM_CHUNK);
      SFGS_REQUIRE(nb < (1ull << 31), SFGS_E_UNSUPPORTED, "tensor %d: too many elements for one launch", i);
      if (blocks + nb >= (1ull << 31)) break;  // grid full: launch what is queued, continue with this tensor
      const uintptr_t bits = (uintptr_t)t.param | (uintptr_t)t.grad | (uintptr_t)t.exp_avg | (uintptr_t)t.exp_avg_sq;
      blocks += (unsigned)nb;
      tab.t[tab.count] = t;
      tab.block_end[tab.count] = blocks;
      tab.vec_ok[tab.count] = (bits & 15) == 0;
      ++tab.count;
    }
    if (!blocks) continue;
    { ProfScope ps_(KID_ADAM, stream);
      hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(ADAM_BLOCK), 0, stream, tab); }
    SFGS_POST_LAUNCH("adam", stream, 0);
  }
  return SFGS_OK;
}
