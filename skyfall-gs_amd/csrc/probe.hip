// probe.hip -- box probe of libsfgs.so (include/sfgs.h: sfgs_box_probe). Not on the rendering path: bench.py runs it once, before
// its timed region, so that numbers from different boxes of a pool (same binary, 3-8 % apart) can be told from code changes.
#include "sfgs_internal.h"

namespace sfgs {

// A fixed VALU loop: every thread runs ITERS x 8 independent v_fma_f32 chains (no memory traffic), 8 waves per SIMD on every CU.
// Reports what the box sustains on plain FP32 multiply-adds -- the instruction class the compositing kernels are bound by.
__global__ void __launch_bounds__(256) valu_probe_kernel(float* __restrict__ out, unsigned long long* __restrict__ clk, int iters) {
  const unsigned gid = blockIdx.x * 256u + threadIdx.x;
  float a0 = (float)(gid & 7u) * 0.125f, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f,
        a6 = a0 + 6.f, a7 = a0 + 7.f;
  const float m = 0.999f + (float)(gid & 3u) * 1e-4f, c = 1e-3f;
  const unsigned long long w0 = wall_clock64();
#pragma unroll 4
  for (int i = 0; i < iters; ++i) {
    a0 = fmaf(a0, m, c); a1 = fmaf(a1, m, c); a2 = fmaf(a2, m, c); a3 = fmaf(a3, m, c);
    a4 = fmaf(a4, m, c); a5 = fmaf(a5, m, c); a6 = fmaf(a6, m, c); a7 = fmaf(a7, m, c);
  }
  const unsigned long long w1 = wall_clock64();
  out[gid] = ((a0 + a1) + (a2 + a3)) + ((a4 + a5) + (a6 + a7));
  if (gid == 0) clk[0] = w1 - w0;
}

}  // namespace sfgs

using namespace sfgs;

extern "C" int sfgs_box_probe(double* valu_tflops, double* sclk_mhz_effective, void* stream_) {
  SFGS_REQUIRE(valu_tflops && sclk_mhz_effective, SFGS_E_ARG, "sfgs_box_probe: NULL output");
  hipStream_t stream = (hipStream_t)stream_;
  int dev = 0, cus = 0;
  SFGS_CHECK_HIP(hipGetDevice(&dev));
  SFGS_CHECK_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  const int blocks = cus * 8, iters = 2048;   // 8 workgroups of 4 waves per CU = 8 waves per SIMD
  float* out = nullptr;
  unsigned long long* clk = nullptr;
  SFGS_CHECK_HIP(hipMalloc(&out, (size_t)blocks * 256 * sizeof(float) + 64));
  clk = (unsigned long long*)(out + (size_t)blocks * 256);
  hipEvent_t e0 = nullptr, e1 = nullptr;
  SFGS_CHECK_HIP(hipEventCreate(&e0));
  SFGS_CHECK_HIP(hipEventCreate(&e1));
  float best = 0.f;
  for (int rep = 0; rep < 4; ++rep) {   // the first launch warms the clocks; the fastest of the rest is reported
    SFGS_CHECK_HIP(hipEventRecord(e0, stream));
    hipLaunchKernelGGL(valu_probe_kernel, dim3(blocks), dim3(256), 0, stream, out, clk, iters);
    SFGS_CHECK_HIP(hipEventRecord(e1, stream));
    SFGS_CHECK_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    SFGS_CHECK_HIP(hipEventElapsedTime(&ms, e0, e1));
    if (rep > 0 && (best == 0.f || ms < best)) best = ms;
  }
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  SFGS_CHECK_HIP(hipFree(out));
  SFGS_REQUIRE(best > 0.f, SFGS_E_HIP, "sfgs_box_probe: no time measured");
  const double fma_per_thread = (double)iters * 8.0;
  const double flops = (double)blocks * 256.0 * fma_per_thread * 2.0;
  *valu_tflops = flops / (best * 1e-3) / 1e12;
  // a SIMD issues one wave64 v_fma_f32 per 2 cycles (32 lanes per cycle): cycles per SIMD = 2 x the wave instructions it ran
  const double wave_instr_per_simd = (double)blocks * 4.0 / ((double)cus * 4.0) * fma_per_thread;
  *sclk_mhz_effective = 2.0 * wave_instr_per_simd / (best * 1e-3) / 1e6;
  return SFGS_OK;
}
