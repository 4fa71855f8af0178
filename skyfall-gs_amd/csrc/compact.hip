// compact.hip -- row compaction of MANY tensors by one keep-mask in one pass.
// Reference: GaussianModel.prune_points / _prune_optimizer (scene/gaussian_model.py:563-603): every parameter, both
// Adam moments of every parameter and five per-Gaussian statistics tensors are filtered with `tensor[mask]` -- 26
// boolean-index operations, each of which runs its own mask -> index conversion (nonzero) and waits for the host to
// learn the output size. Here: ONE prefix scan of the mask (one host read-back: the kept-row count, needed to size the
// outputs) and ONE multi-tensor gather launch. Pure data movement: bit-exact by construction.
#include "sfgs_internal.h"

namespace sfgs {

constexpr int CP_NT = 256, CP_ROWS_PER_THREAD = 16, CP_CHUNK = CP_NT * CP_ROWS_PER_THREAD;
constexpr int CP_MAX_TENSORS = 48;
constexpr int CP_WORDS_PER_THREAD = 8, CP_WORD_CHUNK = CP_NT * CP_WORDS_PER_THREAD;

struct CompactScratch {
  unsigned long long* total;  // [1] (+ padding)
  unsigned* block_sum;        // [NB]
  unsigned* dst_index;        // [N]
};
static inline int64_t cp_blocks(int64_t N) { return (N + CP_CHUNK - 1) / CP_CHUNK; }
static inline size_t cp_scratch_bytes(int64_t N) {
  return 256 + align_up((size_t)cp_blocks(N) * 4, 256) + align_up((size_t)N * 4, 256);
}
static inline CompactScratch cp_view(void* scratch, int64_t N) {
  char* p = (char*)scratch;
  CompactScratch s;
  s.total = (unsigned long long*)p; p += 256;
  s.block_sum = (unsigned*)p; p += align_up((size_t)cp_blocks(N) * 4, 256);
  s.dst_index = (unsigned*)p;
  return s;
}

__global__ void __launch_bounds__(CP_NT)
compact_count_kernel(const unsigned char* __restrict__ keep, long long N, unsigned* __restrict__ block_sum) {
  __shared__ unsigned smem[CP_NT / 64 + 1];
  const long long base = (long long)blockIdx.x * CP_CHUNK + (long long)threadIdx.x * CP_ROWS_PER_THREAD;
  unsigned c = 0;
  unsigned char kv[CP_ROWS_PER_THREAD];
#pragma unroll
  for (int k = 0; k < CP_ROWS_PER_THREAD; ++k) kv[k] = keep[min(base + k, N - 1)];   // loads first (see compact_gather)
#pragma unroll
  for (int k = 0; k < CP_ROWS_PER_THREAD; ++k) c += (base + k < N && kv[k]) ? 1u : 0u;
  unsigned total;
  block_excl_scan_u32<CP_NT>(c, &total, smem);
  if (threadIdx.x == 0) block_sum[blockIdx.x] = total;
}

__global__ void __launch_bounds__(1024)
compact_scan_blocks_kernel(unsigned* __restrict__ block_sum, long long NB, unsigned long long* __restrict__ total_out) {
  __shared__ unsigned smem[1024 / 64 + 1];
  unsigned long long carry = 0;
  for (long long b0 = 0; b0 < NB; b0 += 1024) {
    const long long i = b0 + threadIdx.x;
    const unsigned v = i < NB ? block_sum[i] : 0u;
    unsigned total;
    const unsigned ex = block_excl_scan_u32<1024>(v, &total, smem);
    if (i < NB) block_sum[i] = (unsigned)carry + ex;
    carry += total;
  }
  if (threadIdx.x == 0) *total_out = carry;
}

__global__ void __launch_bounds__(CP_NT)
compact_index_kernel(const unsigned char* __restrict__ keep, long long N, const unsigned* __restrict__ block_sum,
                     unsigned* __restrict__ dst_index) {
  __shared__ unsigned smem[CP_NT / 64 + 1];
  const long long base = (long long)blockIdx.x * CP_CHUNK + (long long)threadIdx.x * CP_ROWS_PER_THREAD;
  unsigned flags = 0, c = 0;
  unsigned char kv[CP_ROWS_PER_THREAD];
#pragma unroll
  for (int k = 0; k < CP_ROWS_PER_THREAD; ++k) kv[k] = keep[min(base + k, N - 1)];
#pragma unroll
  for (int k = 0; k < CP_ROWS_PER_THREAD; ++k)
    if (base + k < N && kv[k]) { flags |= 1u << k; ++c; }
  unsigned total;
  unsigned pos = block_sum[blockIdx.x] + block_excl_scan_u32<CP_NT>(c, &total, smem);
#pragma unroll
  for (int k = 0; k < CP_ROWS_PER_THREAD; ++k)
    if (base + k < N) { dst_index[base + k] = pos; pos += (flags >> k) & 1u; }
}

struct CompactTable {
  const void* src[CP_MAX_TENSORS];
  void* dst[CP_MAX_TENSORS];
  unsigned row_units[CP_MAX_TENSORS];   // row size in units of U
  unsigned block_end[CP_MAX_TENSORS];
  int count;
};

template <typename U>
__global__ void __launch_bounds__(CP_NT)
compact_gather_kernel(const CompactTable tab, const unsigned char* __restrict__ keep,
                      const unsigned* __restrict__ dst_index, long long N) {
  int ti = 0;
  while (ti + 1 < tab.count && blockIdx.x >= tab.block_end[ti]) ++ti;
  const unsigned first = ti ? tab.block_end[ti - 1] : 0u;
  const unsigned ru = tab.row_units[ti];
  const U* __restrict__ src = (const U*)tab.src[ti];
  U* __restrict__ dst = (U*)tab.dst[ti];
  const long long words = N * (long long)ru;
  const long long base = (long long)(blockIdx.x - first) * CP_WORD_CHUNK;   // block-uniform
  const long long row0 = base / ru;                                         // one 64-bit division per block
  const unsigned rem0 = (unsigned)(base - row0 * ru);
  const float inv = 1.0f / (float)ru;
  // three phases so that every load of the thread is in flight before the first use: rows and keep flags, then
  // destination rows and source words (unconditional, from clamped indices), then the predicated stores. (One loop with
  // `if (keep[row]) dst[..dst_index[row]..] = src[w]` serialises three dependent round trips per word.)
  long long rowv[CP_WORDS_PER_THREAD];
  unsigned colv[CP_WORDS_PER_THREAD];
  unsigned char kp[CP_WORDS_PER_THREAD];
#pragma unroll
  for (int k = 0; k < CP_WORDS_PER_THREAD; ++k) {
    const unsigned off = threadIdx.x + k * CP_NT;
    // (rem0 + off) / ru with x < 2^24: float estimate + one correction step is exact
    const unsigned x = rem0 + off;
    unsigned q = (unsigned)((float)x * inv);
    if (q * ru > x) --q; else if ((q + 1) * ru <= x) ++q;
    rowv[k] = min(row0 + (long long)q, N - 1);
    colv[k] = x - q * ru;
    kp[k] = keep[rowv[k]];
  }
  U val[CP_WORDS_PER_THREAD];
  unsigned di[CP_WORDS_PER_THREAD];
#pragma unroll
  for (int k = 0; k < CP_WORDS_PER_THREAD; ++k) {
    const long long w = min(base + (long long)(threadIdx.x + k * CP_NT), words - 1);
    val[k] = src[w];
    di[k] = dst_index[rowv[k]];
  }
#pragma unroll
  for (int k = 0; k < CP_WORDS_PER_THREAD; ++k) {
    const long long w = base + (long long)(threadIdx.x + k * CP_NT);
    if (w < words && kp[k]) dst[(long long)di[k] * ru + colv[k]] = val[k];
  }
}

}  // namespace sfgs

using namespace sfgs;

extern "C" size_t sfgs_compact_scratch_bytes(int64_t N) { return N < 0 ? 0 : cp_scratch_bytes(N); }

extern "C" int sfgs_compact_plan(const unsigned char* keep, int64_t N, void* scratch, size_t scratch_sz,
                                 int64_t* kept_out, void* stream_) {
  SFGS_REQUIRE(N >= 0 && N < (1ll << 32), SFGS_E_ARG, "row count out of range");
  SFGS_REQUIRE(scratch && scratch_sz >= cp_scratch_bytes(N), SFGS_E_CAPACITY, "compact scratch too small");
  hipStream_t stream = (hipStream_t)stream_;
  const CompactScratch s = cp_view(scratch, N);
  if (N == 0) {
    SFGS_CHECK_HIP(hipMemsetAsync(s.total, 0, 8, stream));
  } else {
    SFGS_REQUIRE(keep, SFGS_E_ARG, "NULL mask");
    const long long NB = cp_blocks(N);
    { ProfScope ps_(KID_COMPACT_SCAN, stream);
      hipLaunchKernelGGL(compact_count_kernel, dim3((unsigned)NB), dim3(CP_NT), 0, stream, keep, (long long)N, s.block_sum);
      hipLaunchKernelGGL(compact_scan_blocks_kernel, dim3(1), dim3(1024), 0, stream, s.block_sum, NB, s.total);
      hipLaunchKernelGGL(compact_index_kernel, dim3((unsigned)NB), dim3(CP_NT), 0, stream, keep, (long long)N,
                         s.block_sum, s.dst_index); }
    SFGS_POST_LAUNCH("compact_scan", stream, 0);
  }
  if (kept_out) {  // the one host synchronisation of a prune: the caller sizes its outputs with this
    unsigned long long h = 0;
    SFGS_CHECK_HIP(hipMemcpyAsync(&h, s.total, 8, hipMemcpyDeviceToHost, stream));
    SFGS_CHECK_HIP(hipStreamSynchronize(stream));
    *kept_out = (int64_t)h;
  }
  return SFGS_OK;
}

extern "C" int sfgs_compact_rows(const unsigned char* keep, int64_t N, const void* scratch,
                                 const SfgsCompactTensor* tensors, int32_t count, void* stream_) {
  SFGS_REQUIRE(N >= 0 && N < (1ll << 32) && count >= 0, SFGS_E_ARG, "bad row / tensor count");
  if (N == 0 || count == 0) return SFGS_OK;
  SFGS_REQUIRE(keep && scratch && tensors, SFGS_E_ARG, "NULL argument");
  hipStream_t stream = (hipStream_t)stream_;
  const CompactScratch s = cp_view(const_cast<void*>(scratch), N);
  bool words_ok = true;  // 4-byte units when every tensor allows it, bytes otherwise
  for (int i = 0; i < count; ++i) {
    SFGS_REQUIRE(tensors[i].row_bytes >= 0 && tensors[i].row_bytes < (1ll << 22), SFGS_E_ARG, "tensor %d: bad row size", i);
    if (tensors[i].row_bytes == 0) continue;
    SFGS_REQUIRE(tensors[i].src && tensors[i].dst, SFGS_E_ARG, "tensor %d: NULL pointer", i);
    if ((tensors[i].row_bytes & 3) || (((uintptr_t)tensors[i].src | (uintptr_t)tensors[i].dst) & 3)) words_ok = false;
  }
  const int unit = words_ok ? 4 : 1;
  int i = 0;
  while (i < count) {
    CompactTable tab;
    tab.count = 0;
    uint64_t blocks = 0;
    for (; i < count && tab.count < CP_MAX_TENSORS; ++i) {
      if (tensors[i].row_bytes == 0) continue;
      const uint64_t ru = (uint64_t)tensors[i].row_bytes / unit;
      const uint64_t nb = ((uint64_t)N * ru + CP_WORD_CHUNK - 1) / CP_WORD_CHUNK;
      SFGS_REQUIRE(nb < (1ull << 31), SFGS_E_UNSUPPORTED, "tensor %d: too large for one launch", i);
      if (blocks + nb >= (1ull << 31)) break;
      blocks += nb;
      tab.src[tab.count] = tensors[i].src; tab.dst[tab.count] = tensors[i].dst;
      tab.row_units[tab.count] = (unsigned)ru; tab.block_end[tab.count] = (unsigned)blocks;
      ++tab.count;
    }
    if (!blocks) continue;
    { ProfScope ps_(KID_COMPACT_GATHER, stream);
      if (words_ok) hipLaunchKernelGGL(compact_gather_kernel<uint32_t>, dim3((unsigned)blocks), dim3(CP_NT), 0, stream, tab, keep, s.dst_index, (long long)N);
      else hipLaunchKernelGGL(compact_gather_kernel<unsigned char>, dim3((unsigned)blocks), dim3(CP_NT), 0, stream, tab, keep, s.dst_index, (long long)N); }
    SFGS_POST_LAUNCH("compact_gather", stream, 0);
  }
  return SFGS_OK;
}
