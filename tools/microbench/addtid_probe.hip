// addtid_probe.hip -- ds_write_addtid_b32 semantics probe (for the zero fill of raster_bwd.hip): four waves clear their
// own 8 448-byte LDS regions (33 stores of 256 bytes) at bases wave * 9 656; prints which dwords ended up cleared.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void __launch_bounds__(256) k(float* out) {
  __shared__ float sm[9664];   // 38 656 bytes
  for (int i = threadIdx.x; i < 9664; i += 256) sm[i] = 1.f;
  __syncthreads();
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const unsigned base = (unsigned)(uintptr_t)sm + 9656u * (unsigned)wave;
  const float zero = 0.f;
#define Z(O) "ds_write_addtid_b32 %1 offset:" #O "\n\t"
  asm volatile("s_mov_b32 m0, %0\n\t" Z(0) Z(256) Z(512) Z(768) Z(1024) Z(1280) Z(1536) Z(1792) Z(2048) Z(2304) Z(2560)
               Z(2816) Z(3072) Z(3328) Z(3584) Z(3840) Z(4096) Z(4352) Z(4608) Z(4864) Z(5120) Z(5376) Z(5632) Z(5888)
               Z(6144) Z(6400) Z(6656) Z(6912) Z(7168) Z(7424) Z(7680) Z(7936) Z(8192)
               :: "s"(__builtin_amdgcn_readfirstlane((int)base)), "v"(zero) : "memory");
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __syncthreads();
  int bad = 0;
  for (int i = threadIdx.x; i < 9664; i += 256) {
    const int w = i / 2414, r = i - w * 2414;
    const bool should_clear = w < 4 && r < 2112;
    bad += (sm[i] == 0.f) != should_clear;
  }
  if (bad) atomicAdd((int*)out + 9664 + (blockIdx.x & 1), bad);
  if (blockIdx.x == 0) for (int i = threadIdx.x; i < 9664; i += 256) out[i] = sm[i];
}
int main() {
  float* d; hipMalloc(&d, 9666 * 4); hipMemset(d, 0, 9666 * 4); static float h[9666];
  hipLaunchKernelGGL(k, dim3(4096), dim3(256), 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int w = 0; w < 4; ++w) {
    const int b = 9656 / 4 * w;
    int cleared = 0, first_bad = -1;
    for (int i = 0; i < 2112; ++i) { if (h[b + i] == 0.f) ++cleared; else if (first_bad < 0) first_bad = i; }
    int outside = 0;
    for (int i = 2112; i < 2414 && b + i < 9664; ++i) outside += h[b + i] == 0.f;
    printf("wave %d: %d of 2112 dwords cleared (first not cleared: %d), %d cleared beyond\n", w, cleared, first_bad, outside);
  }
  printf("4096 workgroups (4 per CU resident): mismatching dwords (even / odd blocks): %d %d\n", ((int*)h)[9664], ((int*)h)[9665]);
  return 0;
}
