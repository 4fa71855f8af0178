// lds_probe.hip -- two questions behind composite_bwd's occupancy (round 4):
//  A. how many ONE-WAVE workgroups with X bytes of LDS are resident per CU (census: every workgroup counts itself into
//     its CU's slot while it spins)?  -> the LDS allocation granule of gfx950 and the byte budget for 17 / 18 / 20 waves
//  B. what happens to DS accesses beyond the workgroup's allocation (ISA: reads return 0, writes are dropped)?
// build: hipcc --offload-arch=gfx950 -O3 lds_probe.hip -o lds_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

__device__ __forceinline__ unsigned cu_key() {
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  return ((xcc & 0xfu) << 8) | ((hw >> 8) & 0xffu);   // cu_id[11:8] sh_id[12] se_id[15:13]
}

__global__ void __launch_bounds__(64) census(int* active, int* peak, int spin) {
  extern __shared__ float sm[];
  if (threadIdx.x == 0) {
    sm[0] = 1.f;
    const unsigned k = cu_key();
    const int n = atomicAdd(&active[k], 1) + 1;
    atomicMax(&peak[k], n);
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) __builtin_amdgcn_s_sleep(8);
    atomicSub(&active[k], 1);
  }
}

// B: static allocation of exactly ALLOC bytes; every wave tags its own LDS, writes a different pattern to the 2 KB BEYOND
// it, reads that range back, spins (so that neighbours run their phases meanwhile) and re-checks its own bytes.
constexpr int ALLOC = 8960;
__global__ void __launch_bounds__(64) oob(int* res) {
  __shared__ unsigned sm[ALLOC / 4];
  const unsigned tag = 0x10000u + blockIdx.x;
  for (int i = threadIdx.x; i < ALLOC / 4; i += 64) sm[i] = tag;
  __builtin_amdgcn_wave_barrier();
  const unsigned base = (unsigned)(uintptr_t)sm;
  unsigned nonzero = 0;
  for (int i = threadIdx.x; i < 512; i += 64) {
    const unsigned addr = base + ALLOC + 4u * (unsigned)i;
    unsigned v;
    asm volatile("ds_write_b32 %1, %2\n\ts_waitcnt lgkmcnt(0)\n\tds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)"
                 : "=v"(v) : "v"(addr), "v"(0xdead0000u + blockIdx.x) : "memory");
    nonzero += v != 0u;
  }
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < 2000) __builtin_amdgcn_s_sleep(4);
  unsigned bad = 0;
  for (int i = threadIdx.x; i < ALLOC / 4; i += 64) bad += sm[i] != tag;
  if (nonzero) atomicAdd(&res[0], (int)nonzero);
  if (bad) atomicAdd(&res[1], (int)bad);
  if (threadIdx.x == 0) atomicAdd(&res[2], 1);
}

int main() {
  int *active, *peak;
  hipMalloc(&active, 4096 * 4); hipMalloc(&peak, 4096 * 4);
  std::vector<int> h(4096);
  hipFuncSetAttribute((const void*)census, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  printf("A. resident one-wave workgroups per CU vs LDS bytes per workgroup (census max over CUs / occupancy API)\n");
  int last = -1;
  for (int lds = 4096; lds <= 16384; lds += 64) {
    hipMemset(active, 0, 4096 * 4); hipMemset(peak, 0, 4096 * 4);
    hipLaunchKernelGGL(census, dim3(256 * 40), dim3(64), lds, 0, active, peak, 3000);
    hipMemcpy(h.data(), peak, 4096 * 4, hipMemcpyDeviceToHost);
    int mx = 0, cus = 0; long long sum = 0;
    for (int v : h) { if (v) { ++cus; sum += v; } if (v > mx) mx = v; }
    int api = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&api, census, 64, lds);
    if (mx != last) { printf("  lds %5d B: census peak %2d (%d CUs seen, mean %.1f), API %2d\n", lds, mx, cus, cus ? (double)sum / cus : 0., api); last = mx; }
  }
  int* res; hipMalloc(&res, 16); hipMemset(res, 0, 16);
  hipLaunchKernelGGL(oob, dim3(256 * 64), dim3(64), 0, 0, res);
  int r[3]; hipMemcpy(r, res, 12, hipMemcpyDeviceToHost);
  printf("B. %d one-wave workgroups of %d B static LDS: out-of-range reads that returned non-zero: %d; own words found "
         "corrupted after the neighbours' out-of-range writes: %d\n", r[2], ALLOC, r[0], r[1]);
  return 0;
}
