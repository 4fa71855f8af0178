// act_math.h -- activations + Mip-Splatting 3D smoothing filter of ONE Gaussian (device code).
//
// Shared by prepass.hip (the stand-alone pre-pass kernels) and by the rasterizer's raw-parameter mode, in which
// preprocess / preprocess_bwd take the model's raw parameters and run this math themselves (the
// pre-pass "folded into preprocess fwd/bwd"). One source, one float sequence: both routes give the same bits.
//
// Reference semantics (scene/gaussian_model.py), in torch's own promotion rules (FT = dtype of filter_3D, OT = dtype
// of the raw opacity parameter, float64 after the reference's reset_opacity, :483-501):
//   scales    = sqrt(exp(_scaling)^2 + filter_3D^2)                            get_scaling_with_3D_filter  :207-213
//   opacity   = sigmoid(_opacity) * sqrt(prod s^2 / prod (s^2 + filter_3D^2))  get_opacity_with_3D_filter  :237-249
//   rotation  = normalize(_rotation)  (F.normalize, eps 1e-12)                 get_rotation                :216-217
// followed by render()'s .float() casts (gaussian_renderer/__init__.py:137-138).
#pragma once
#include <hip/hip_runtime.h>

namespace sfgs {

//   sq_i  = square(exp(raw_i))                    float32
//   det1  = prod_i sq_i                           float32
//   t_i   = sq_i + square(filter)                 FT
//   det2  = prod_i t_i                            FT
//   coef  = sqrt(det1 / det2)                     FT
//   o     = sigmoid(raw opacity)                  OT
template <typename FT, typename OT>
struct ActTerms {
  float sq[3], det1;
  OT o;
  FT f2, t[3], det2, coef;
};

template <typename FT>
__device__ __forceinline__ FT sqrt_t(FT v);
template <> __device__ __forceinline__ float sqrt_t<float>(float v) { return sqrtf(v); }
template <> __device__ __forceinline__ double sqrt_t<double>(double v) { return sqrt(v); }

template <typename FT, typename OT>
__device__ __forceinline__ ActTerms<FT, OT> act_terms(const float raw_s[3], OT raw_o, FT f) {
  ActTerms<FT, OT> p;
  p.f2 = f * f;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const float s = expf(raw_s[i]);
    p.sq[i] = s * s;
    p.t[i] = (FT)p.sq[i] + p.f2;
  }
  p.det1 = (p.sq[0] * p.sq[1]) * p.sq[2];
  p.det2 = (p.t[0] * p.t[1]) * p.t[2];
  p.coef = sqrt_t<FT>((FT)p.det1 / p.det2);
  if constexpr (sizeof(OT) == 8) p.o = 1.0 / (1.0 + exp(-raw_o));
  else p.o = 1.0f / (1.0f + expf(-raw_o));
  return p;
}

// the float32 values render() hands the rasterizer
template <typename FT, typename OT>
__device__ __forceinline__ void act_outputs(const ActTerms<FT, OT>& p, float scales[3], float* opacity) {
#pragma unroll
  for (int i = 0; i < 3; ++i) scales[i] = (float)sqrt_t<FT>(p.t[i]);
  if constexpr (sizeof(OT) == 8) *opacity = (float)(p.o * (double)p.coef);   // torch promotes to float64
  else *opacity = (float)((FT)p.o * p.coef);
}

__device__ __forceinline__ float4 act_rotation(float4 q) {
  const float n = fmaxf(sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w), 1e-12f);
  return make_float4(q.x / n, q.y / n, q.z / n, q.w / n);
}

// gradients w.r.t. the raw scaling / raw opacity from those w.r.t. the activated scales (gs) and opacity (go); all in
// float64 like the autograd graph the reference builds when either dtype is float64 (and at least as exact otherwise)
template <typename FT, typename OT>
__device__ __forceinline__ void act_backward(const ActTerms<FT, OT>& p, const float gs[3], float go_, float g_raw_s[3],
                                             OT* g_raw_o) {
  const double coef = (double)p.coef, o = (double)p.o, go = (double)go_;
  *g_raw_o = (OT)(go * coef * o * (1.0 - o));
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    // d scales_i / d raw_i = s_i^2 / sqrt(s_i^2 + f^2) ; d (o coef) / d raw_i = o coef f^2 / (s_i^2 + f^2)
    const double t = (double)p.t[i];
    g_raw_s[i] = (float)((double)gs[i] * (double)p.sq[i] / sqrt(t) + go * o * coef * (double)p.f2 / t);
  }
}

__device__ __forceinline__ float4 act_rotation_backward(float4 q, float4 gr) {
  const float nn = sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
  if (nn > 1e-12f) {  // d (q/|q|) : (g - q_hat (q_hat . g)) / |q|
    const float inv = 1.0f / nn;
    const float4 h = make_float4(q.x * inv, q.y * inv, q.z * inv, q.w * inv);
    const float dot = h.x * gr.x + h.y * gr.y + h.z * gr.z + h.w * gr.w;
    return make_float4((gr.x - h.x * dot) * inv, (gr.y - h.y * dot) * inv, (gr.z - h.z * dot) * inv,
                       (gr.w - h.w * dot) * inv);
  }
  return make_float4(gr.x * 1e12f, gr.y * 1e12f, gr.z * 1e12f, gr.w * 1e12f);   // clamped denominator: q / 1e-12
}

// (filter dtype, raw opacity dtype) of a launch: bit 0 = filter_3D is float64, bit 1 = the raw opacity is float64.
// CALL(FT, OT) is expanded in the matching branch (the mask is uniform over the launch).
#define SFGS_ACT_DISPATCH(MASK, CALL)                                            \
  do {                                                                           \
    switch ((MASK) & 3) {                                                        \
      case 0: CALL(float, float); break;                                         \
      case 1: CALL(double, float); break;                                        \
      case 2: CALL(float, double); break;                                        \
      default: CALL(double, double); break;                                      \
    }                                                                            \
  } while (0)

// element g of a float or double array as raw bits / back: lets a kernel issue the load where its other loads are and
// interpret the word inside the dtype switch later
template <typename T> __device__ __forceinline__ unsigned long long raw_bits(const void* p, size_t g) {
  if constexpr (sizeof(T) == 4) return static_cast<const unsigned*>(p)[g];
  else return static_cast<const unsigned long long*>(p)[g];
}
template <typename T> __device__ __forceinline__ T from_bits(unsigned long long b) {
  if constexpr (sizeof(T) == 4) return __uint_as_float((unsigned)b);
  else return __longlong_as_double((long long)b);
}

}  // namespace sfgs
