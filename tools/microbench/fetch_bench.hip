// fetch_bench.hip -- kernels with a KNOWN memory-side byte count, to calibrate rocprofv3's FETCH_SIZE / WRITE_SIZE for
// the access patterns of the compositing kernels (MI355X_MICROARCH.md: "FETCH_SIZE reports exactly 1/2 of the bytes of a
// wide coalesced streaming read ... other access widths and WRITE_SIZE are uncalibrated: calibrate on a known byte count
// in your own access pattern"). tools/prof_summary.py doubles FETCH_SIZE for every kernel; whether that holds for
// 48-byte record GATHERS (three 16-byte loads per lane at a random record) decides whether composite_bwd's "2.65 x
// algorithmic" traffic is real. Every working set is larger than the 256 MiB Infinity Cache; every record / element is
// touched exactly once per launch (ids are a random permutation), so the compulsory bytes are exact:
//   stream16      1 GiB, 16 B per lane, coalesced                          expected = bytes
//   stream4       256 MiB, 4 B per lane, coalesced                         expected = bytes
//   gather48      8 Mi records of 48 B at permuted ids, 3 x 16 B per lane  useful 48 B, 64-B sectors 96 B, 128-B lines 160 B per record (+ 4 B id)
//   gather64      8 Mi records of 64 B (aligned), 4 x 16 B per lane        useful = sectors 64 B, lines 128 B per record (+ 4 B id)
//   gather48_coh  gather48 with ids sorted inside blocks of 4 096          same bytes; neighbouring lanes share lines
//   wstream16     1 GiB of 16-byte stores, coalesced                       expected = bytes
//   wscatter48    48-B records at permuted ids, written like composite_bwd: 4 lanes x 3 dword stores (floats r, 4 + r, 8 + r)
//   wscatter48v   the same records as three 16-byte stores of one lane
// build: hipcc --offload-arch=gfx950 -O3 fetch_bench.hip -o fetch_bench ; run under tools/calibrate_fetch.sh
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <numeric>
#include <random>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void __launch_bounds__(256) stream16(const float4* __restrict__ p, size_t n, float* __restrict__ out) {
  float acc = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { const float4 v = p[i]; acc += v.x + v.y + v.z + v.w; }
  if (acc == 1234.5f) out[0] = acc;
}
__global__ void __launch_bounds__(256) stream4(const float* __restrict__ p, size_t n, float* __restrict__ out) {
  float acc = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc += p[i];
  if (acc == 1234.5f) out[0] = acc;
}
template <int F4, int TAG>   // TAG: a second instantiation = a second kernel name for the profiler
__global__ void __launch_bounds__(256) gather(const float4* __restrict__ rec, const unsigned* __restrict__ ids, size_t m, float* __restrict__ out) {
  float acc = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < m; i += (size_t)gridDim.x * 256) {
    const size_t id = ids[i];
#pragma unroll
    for (int q = 0; q < F4; ++q) { const float4 v = rec[id * F4 + q]; acc += (v.x + v.y) + (v.z + v.w); }
  }
  if (acc == 1234.5f) out[0] = acc;
}
__global__ void __launch_bounds__(256) wstream16(float4* __restrict__ p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = make_float4(1.f, 2.f, 3.f, (float)i);
}
// composite_bwd's store: lane (entry e, row r) of a wave writes floats r, 4 + r, 8 + r of entry e's record (16 entries per wave)
__global__ void __launch_bounds__(256) wscatter48(float* __restrict__ rec, const unsigned* __restrict__ ids, size_t m) {
  const int lane = threadIdx.x & 63, e = lane & 15, r = lane >> 4;
  const size_t wave = ((size_t)blockIdx.x * 256 + threadIdx.x) >> 6, nwave = ((size_t)gridDim.x * 256) >> 6;
  for (size_t b = wave * 16; b < m; b += nwave * 16) {
    const size_t id = ids[b + e];
    float* dst = rec + id * 12 + r;
    dst[0] = 1.f; dst[4] = 2.f; dst[8] = (float)r;
  }
}
__global__ void __launch_bounds__(256) wscatter48v(float4* __restrict__ rec, const unsigned* __restrict__ ids, size_t m) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < m; i += (size_t)gridDim.x * 256) {
    const size_t id = ids[i];
    rec[id * 3] = make_float4(1.f, 2.f, 3.f, 4.f); rec[id * 3 + 1] = make_float4(5.f, 6.f, 7.f, 8.f); rec[id * 3 + 2] = make_float4(9.f, 1.f, 2.f, 3.f);
  }
}

int main() {
  const size_t M = 8u << 20;                 // records
  const size_t GIB = 1ull << 30;
  float4* big; float* out; unsigned *ids, *ids_coh;
  CK(hipMalloc(&big, GIB)); CK(hipMalloc(&out, 256)); CK(hipMalloc(&ids, M * 4)); CK(hipMalloc(&ids_coh, M * 4));
  CK(hipMemset(big, 0, GIB));
  std::vector<unsigned> perm(M);
  std::iota(perm.begin(), perm.end(), 0u);
  std::mt19937 rng(12345);
  std::shuffle(perm.begin(), perm.end(), rng);
  CK(hipMemcpy(ids, perm.data(), M * 4, hipMemcpyHostToDevice));
  for (size_t b = 0; b < M; b += 4096) std::sort(perm.begin() + b, perm.begin() + std::min(M, b + 4096));   // coherent inside blocks
  CK(hipMemcpy(ids_coh, perm.data(), M * 4, hipMemcpyHostToDevice));
  const dim3 grid(256 * 16), blk(256);
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  auto timeit = [&](const char* name, double useful, double sectors, double lines, auto launch) {
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(a)); launch(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b)); best = std::min(best, ms);
    }
    printf("{\"kernel\": \"%s\", \"ms\": %.4f, \"useful_bytes\": %.0f, \"bytes_64B_sectors\": %.0f, \"bytes_128B_lines\": %.0f, \"useful_GBps\": %.1f}\n",
           name, best, useful, sectors, lines, useful / best / 1e6);
  };
  const double idb = (double)M * 4;
  timeit("stream16", (double)GIB, (double)GIB, (double)GIB, [&] { stream16<<<grid, blk>>>(big, GIB / 16, out); });
  timeit("stream4", (double)GIB / 4, (double)GIB / 4, (double)GIB / 4, [&] { stream4<<<grid, blk>>>((const float*)big, GIB / 16, out); });
  timeit("gather48", M * 48.0 + idb, M * 96.0 + idb, M * 160.0 + idb, [&] { gather<3, 0><<<grid, blk>>>(big, ids, M, out); });
  timeit("gather64", M * 64.0 + idb, M * 64.0 + idb, M * 128.0 + idb, [&] { gather<4, 0><<<grid, blk>>>(big, ids, M, out); });
  timeit("gather48_coh", M * 48.0 + idb, M * 96.0 + idb, M * 160.0 + idb, [&] { gather<3, 1><<<grid, blk>>>(big, ids_coh, M, out); });
  timeit("wstream16", (double)GIB, (double)GIB, (double)GIB, [&] { wstream16<<<grid, blk>>>(big, GIB / 16); });
  timeit("wscatter48", M * 48.0, M * 96.0, M * 160.0, [&] { wscatter48<<<grid, blk>>>((float*)big, ids, M); });
  timeit("wscatter48v", M * 48.0, M * 96.0, M * 160.0, [&] { wscatter48v<<<grid, blk>>>(big, ids, M); });
  return 0;
}
