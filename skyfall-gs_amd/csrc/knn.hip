// knn.hip -- simple_knn._C.distCUDA2 for gfx950 (reference call site scene/gaussian_model.py:324):
// out[i] = mean squared distance from point i to its 3 nearest OTHER points (index-excluded; duplicates at distance 0
// count). EXACT for every input; the reference's simple-knn (un-vendored submodule) is a Morton sort with
// box pruning, and so is this -- laid out for wave64:
//
//   N <= KNN_BRUTE_MAX   brute force, tiled through LDS (one launch; 1e4 points: tens of microseconds).
//   larger               1. robust bounds: per-axis 1 % / 99 % quantiles of <= 4096 sampled points, widened by a quarter
//                           (a handful of far outliers -- SfM clouds have them -- must not decide the grid resolution);
//                        2. counting sort by a Z-curve cell key (2^(3b) cells on the bounds, b = 4..8 bits per axis from
//                           N; outside points clamp into the border cells): histogram with device atomics, one exclusive
//                           scan, scatter. The order INSIDE a cell is whatever the atomics gave -- irrelevant: the
//                           result is a set property (the three smallest distances, added in ascending order);
//                        3. bounding boxes of every 64 consecutive sorted points ("box" = one wave of queries AND one
//                           wave-load of candidates) and of every 64 boxes ("superbox");
//                        4. query: one wave per box. The wave scans its own box, then walks the superboxes; a superbox
//                           / box is opened only if ANY lane's box distance is below that lane's current third-best
//                           (wave-uniform decisions: ballots). Opening a box = one coalesced 1 KB load, then 64
//                           broadcast steps (v_readlane operands) of sub/mul/add + a 3-instruction top-3 insert
//                           (v_min, v_med3, v_med3). Pruning is exact in floating point: float subtraction, squaring
//                           and the fixed-order sum are monotonic in |d|, so a box's computed distance never exceeds
//                           the computed distance of a point inside it.
//   Dense clusters below the grid resolution degrade towards scanning the cluster, never beyond brute force.
// Squared distances are the same float32 expression as the brute-force kernel and the oracle: (dx dx + dy dy) + dz dz.
#include <cstdlib>
#include <cstring>

#include "sfgs_internal.h"

namespace sfgs {

constexpr int KNN_BLOCK = 256, KNN_TILE = 1024;
constexpr int KNN_BRUTE_MAX = 32768;
constexpr int KNN_SAMPLES = 4096;

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

// ---- spatial path ----------------------------------------------------------------------------------------------------
struct KnnBounds { float lo[3], inv[3]; };   // cell coordinate along axis a = (p[a] - lo[a]) * inv[a], clamped to the grid

__host__ __device__ inline int knn_bits(int N) {
  int lg = 0;
  while ((1ll << lg) < (long long)N) ++lg;
  const int b = (lg + 2) / 3 + 1;             // ~ 2..8 points per cell if the cloud filled its bounds; surfaces fill less
  return b < 4 ? 4 : b > 8 ? 8 : b;
}

// one workgroup: sample <= 4096 points, sort each axis (bitonic in LDS), take the 1 % / 99 % quantiles, widen
__global__ void __launch_bounds__(1024)
knn_bounds_kernel(const float* __restrict__ xyz, int N, int bits, KnnBounds* __restrict__ bounds) {
  __shared__ float v[KNN_SAMPLES];
  const int S = min(N, KNN_SAMPLES);
  int P = 1;
  while (P < S) P <<= 1;                     // padded with +inf (sorts to the end)
  const double stride = (double)N / S;
  for (int a = 0; a < 3; ++a) {
    __syncthreads();
    for (int s = threadIdx.x; s < P; s += 1024) {
      float x = INFINITY;
      if (s < S) {
        x = xyz[3 * (size_t)min((long long)(s * stride), (long long)N - 1) + a];
        if (!(fabsf(x) < INFINITY)) x = INFINITY;        // non-finite coordinates do not take part
      }
      v[s] = x;
    }
    __syncthreads();
    for (int k = 2; k <= P; k <<= 1)
      for (int j = k >> 1; j > 0; j >>= 1) {
        for (int s = threadIdx.x; s < P; s += 1024) {
          const int o = s ^ j;
          if (o > s) {
            const float x = v[s], y = v[o];
            const bool up = (s & k) == 0;
            if ((x > y) == up) { v[s] = y; v[o] = x; }
          }
        }
        __syncthreads();
      }
    if (threadIdx.x == 0) {
      int nf = S;
      while (nf > 0 && !(v[nf - 1] < INFINITY)) --nf;   // finite samples
      float lo = 0.f, hi = 1.f;
      if (nf > 0) { lo = v[(int)(0.01 * (nf - 1))]; hi = v[(int)(0.99 * (nf - 1) + 0.5)]; }
      const float w = hi - lo;
      lo -= 0.25f * w; hi += 0.25f * w;
      const float ext = hi - lo;
      bounds->lo[a] = lo;
      bounds->inv[a] = ext > 0.f ? (float)(1 << bits) / ext : 0.f;    // a flat axis: every point in cell 0
    }
  }
}

__device__ __forceinline__ unsigned knn_spread3(unsigned x) {   // 8 bits -> every third bit
  x = (x | (x << 16)) & 0x030000ffu;
  x = (x | (x << 8)) & 0x0300f00fu;
  x = (x | (x << 4)) & 0x030c30c3u;
  x = (x | (x << 2)) & 0x09249249u;
  return x;
}

__device__ __forceinline__ unsigned knn_key(const KnnBounds& b, int bits, float x, float y, float z) {
  const float top = (float)((1 << bits) - 1);
  float u[3] = {(x - b.lo[0]) * b.inv[0], (y - b.lo[1]) * b.inv[1], (z - b.lo[2]) * b.inv[2]};
  unsigned c[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    float t = u[a];
    t = t >= 0.f ? t : 0.f;       // also NaN -> 0
    t = t <= top ? t : top;
    c[a] = (unsigned)t;
  }
  return knn_spread3(c[0]) | (knn_spread3(c[1]) << 1) | (knn_spread3(c[2]) << 2);
}

__global__ void __launch_bounds__(256)
knn_count_kernel(const float* __restrict__ xyz, int N, int bits, const KnnBounds* __restrict__ bounds,
                 unsigned* __restrict__ keys, unsigned* __restrict__ count) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  const KnnBounds b = *bounds;
  const unsigned k = knn_key(b, bits, xyz[3 * (size_t)i], xyz[3 * (size_t)i + 1], xyz[3 * (size_t)i + 2]);
  keys[i] = k;
  atomicAdd(&count[k], 1u);
}

// exclusive scan of M = nblk * 4096 counters: (a) per-block sums, (b) scan of the sums, (c) local scan + offset
constexpr int KNN_SCAN_TILE = 4096;
__global__ void __launch_bounds__(256)
knn_scan_sums_kernel(const unsigned* __restrict__ count, unsigned* __restrict__ block_sum) {
  __shared__ unsigned sm[8];
  const uint4* src = reinterpret_cast<const uint4*>(count + (size_t)blockIdx.x * KNN_SCAN_TILE) + threadIdx.x * 4;
  unsigned s = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) { const uint4 q = src[k]; s += q.x + q.y + q.z + q.w; }
  unsigned total;
  block_excl_scan_u32<256>(s, &total, sm);
  if (threadIdx.x == 0) block_sum[blockIdx.x] = total;
}

__global__ void __launch_bounds__(1024)
knn_scan_top_kernel(unsigned* __restrict__ block_sum, int nblk) {   // nblk <= 4096
  __shared__ unsigned sm[20];
  unsigned v[4], s = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) { const int j = threadIdx.x * 4 + k; v[k] = j < nblk ? block_sum[j] : 0u; s += v[k]; }
  unsigned total;
  unsigned run = block_excl_scan_u32<1024>(s, &total, sm);
#pragma unroll
  for (int k = 0; k < 4; ++k) { const int j = threadIdx.x * 4 + k; if (j < nblk) block_sum[j] = run; run += v[k]; }
}

__global__ void __launch_bounds__(256)
knn_scan_apply_kernel(const unsigned* __restrict__ count, const unsigned* __restrict__ block_sum,
                      unsigned* __restrict__ start) {
  __shared__ unsigned sm[8];
  const size_t base = (size_t)blockIdx.x * KNN_SCAN_TILE + threadIdx.x * 16;
  unsigned v[16], s = 0;
  const uint4* src = reinterpret_cast<const uint4*>(count + base);
#pragma unroll
  for (int k = 0; k < 4; ++k) { const uint4 q = src[k]; v[4 * k] = q.x; v[4 * k + 1] = q.y; v[4 * k + 2] = q.z; v[4 * k + 3] = q.w; }
#pragma unroll
  for (int k = 0; k < 16; ++k) s += v[k];
  unsigned total;
  unsigned run = block_excl_scan_u32<256>(s, &total, sm) + block_sum[blockIdx.x];
  uint4* dst = reinterpret_cast<uint4*>(start + base);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    uint4 o;
    o.x = run; run += v[4 * k];
    o.y = run; run += v[4 * k + 1];
    o.z = run; run += v[4 * k + 2];
    o.w = run; run += v[4 * k + 3];
    dst[k] = o;
  }
}

__global__ void __launch_bounds__(256)
knn_scatter_kernel(const float* __restrict__ xyz, int N, const unsigned* __restrict__ keys,
                   const unsigned* __restrict__ start, unsigned* __restrict__ count, float4* __restrict__ sorted) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  const unsigned k = keys[i];
  const unsigned pos = start[k] + atomicSub(&count[k], 1u) - 1u;   // the cell's slots, back to front
  sorted[pos] = make_float4(xyz[3 * (size_t)i], xyz[3 * (size_t)i + 1], xyz[3 * (size_t)i + 2], __int_as_float(i));
}

// box b = sorted points [64 b, 64 b + 64): its bounding box; level 1 reads the points, level 2 the level-1 boxes
__global__ void __launch_bounds__(256)
knn_boxes_kernel(const float4* __restrict__ pts, int n, const float4* __restrict__ lo_in, const float4* __restrict__ hi_in,
                 float4* __restrict__ lo_out, float4* __restrict__ hi_out) {
  const int g = blockIdx.x * 256 + threadIdx.x, lane = threadIdx.x & 63;
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  if (g < n) {
    if (pts) {
      const float4 p = pts[g];
      const float c[3] = {p.x, p.y, p.z};
#pragma unroll
      for (int a = 0; a < 3; ++a) if (fabsf(c[a]) < INFINITY) { lo[a] = c[a]; hi[a] = c[a]; }
    } else {
      const float4 l = lo_in[g], h = hi_in[g];
      lo[0] = l.x; lo[1] = l.y; lo[2] = l.z; hi[0] = h.x; hi[1] = h.y; hi[2] = h.z;
    }
  }
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { lo[a] = fminf(lo[a], __shfl_xor(lo[a], d)); hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], d)); }
  if (lane == 0 && (g >> 6) < (n + 63) / 64) {
    lo_out[g >> 6] = make_float4(lo[0], lo[1], lo[2], 0.f);
    hi_out[g >> 6] = make_float4(hi[0], hi[1], hi[2], 0.f);
  }
}

__device__ __forceinline__ float knn_box_dist2(float px, float py, float pz, const float4 lo, const float4 hi) {
  // max(lo - p, p - hi, 0) per axis; the same (dx dx + dy dy) + dz dz as the point distance (monotonic: exact pruning)
  const float dx = fmaxf(fmaxf(lo.x - px, px - hi.x), 0.f);
  const float dy = fmaxf(fmaxf(lo.y - py, py - hi.y), 0.f);
  const float dz = fmaxf(fmaxf(lo.z - pz, pz - hi.z), 0.f);
  return dx * dx + dy * dy + dz * dz;
}

__device__ __forceinline__ void knn_insert(float d, float& b0, float& b1, float& b2) {
  // b0 <= b1 <= b2 stays sorted: 3 instructions (v_min, v_med3, v_med3)
  const float n2 = __builtin_amdgcn_fmed3f(b1, b2, d);
  const float n1 = __builtin_amdgcn_fmed3f(b0, b1, d);
  b0 = fminf(b0, d); b1 = n1; b2 = n2;
}

template <bool SELF>
__device__ __forceinline__ void knn_scan_box(const float4* __restrict__ sorted, int N, int box, int lane, float px, float py,
                                             float pz, float& b0, float& b1, float& b2) {
  const int j = box * 64 + lane;
  float4 c = make_float4(INFINITY, INFINITY, INFINITY, 0.f);   // past the end: distance inf, never inserted
  if (j < N) {
    c = sorted[j];
    if (!(fabsf(c.x) < INFINITY && fabsf(c.y) < INFINITY && fabsf(c.z) < INFINITY)) c.x = c.y = c.z = INFINITY;
  }
#pragma unroll 8
  for (int t = 0; t < 64; ++t) {
    const float qx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(c.x), t)),
                qy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(c.y), t)),
                qz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(c.z), t));
    const float dx = qx - px, dy = qy - py, dz = qz - pz;
    float d = dx * dx + dy * dy + dz * dz;
    if (SELF) d = (t == lane) ? INFINITY : d;
    knn_insert(d, b0, b1, b2);
  }
}

__global__ void __launch_bounds__(256)
knn_query_kernel(const float4* __restrict__ sorted, int N, const float4* __restrict__ blo, const float4* __restrict__ bhi,
                 int nb, const float4* __restrict__ slo, const float4* __restrict__ shi, int nsb, float* __restrict__ out) {
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
  const int box = blockIdx.x * 4 + wave;
  if (box >= nb) return;
  const int i = box * 64 + lane;
  const bool valid = i < N;
  float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
  if (valid) p = sorted[i];
  const bool finite = valid && fabsf(p.x) < INFINITY && fabsf(p.y) < INFINITY && fabsf(p.z) < INFINITY;
  const float px = finite ? p.x : 0.f, py = finite ? p.y : 0.f, pz = finite ? p.z : 0.f;
  float b0 = INFINITY, b1 = INFINITY, b2 = INFINITY;
  knn_scan_box<true>(sorted, N, box, lane, px, py, pz, b0, b1, b2);
  // neighbouring boxes first (Z-curve neighbours are usually spatial neighbours): tightens b2 before the walk
  for (int off = 1; off <= 2; ++off) {
    if (box - off >= 0) knn_scan_box<false>(sorted, N, box - off, lane, px, py, pz, b0, b1, b2);
    if (box + off < nb) knn_scan_box<false>(sorted, N, box + off, lane, px, py, pz, b0, b1, b2);
  }
  for (int S = 0; S < nsb; ++S) {
    const float ds = knn_box_dist2(px, py, pz, slo[S], shi[S]);     // wave-uniform addresses: scalar loads
    if (__ballot(finite && ds < b2) == 0ull) continue;
    // lane k tests box 64 S + k against every lane's need: first a cheap superset -- the box of this wave's queries,
    // grown by the largest third-best -- then, box by box, the exact any-lane test
    const int cb = S * 64 + lane;
    unsigned long long cand = 0ull;
    {
      float4 l = make_float4(INFINITY, INFINITY, INFINITY, 0.f), h = make_float4(-INFINITY, -INFINITY, -INFINITY, 0.f);
      if (cb < nb) { l = blo[cb]; h = bhi[cb]; }
      const float4 ql = blo[box], qh = bhi[box];
      float r2 = finite ? b2 : 0.f;
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) r2 = fmaxf(r2, __shfl_xor(r2, d));
      const float gx = fmaxf(fmaxf(l.x - qh.x, ql.x - h.x), 0.f), gy = fmaxf(fmaxf(l.y - qh.y, ql.y - h.y), 0.f),
                  gz = fmaxf(fmaxf(l.z - qh.z, ql.z - h.z), 0.f);
      const float g2 = gx * gx + gy * gy + gz * gz;
      const bool skip = (cb >= box - 2 && cb <= box + 2);            // already scanned
      cand = __ballot(cb < nb && !skip && g2 < r2);
    }
    while (cand) {
      const int k = __builtin_ctzll(cand);
      cand &= cand - 1;
      const int c = S * 64 + k;
      const float db = knn_box_dist2(px, py, pz, blo[c], bhi[c]);
      if (__ballot(finite && db < b2) == 0ull) continue;
      knn_scan_box<false>(sorted, N, c, lane, px, py, pz, b0, b1, b2);
    }
  }
  if (valid) {
    float sum = 0.f;
    int c = 0;
    if (finite) {
      if (b0 < INFINITY) { sum += b0; ++c; }
      if (b1 < INFINITY) { sum += b1; ++c; }
      if (b2 < INFINITY) { sum += b2; ++c; }
    }
    out[__float_as_int(p.w)] = c ? sum / (float)c : 0.f;
  }
}

struct KnnScratch {
  KnnBounds* bounds;
  unsigned *keys, *count, *start, *block_sum;
  float4 *sorted, *blo, *bhi, *slo, *shi;
  size_t bytes;
  int bits, nb, nsb;
  size_t M;
};

static KnnScratch knn_layout(void* base, int N) {
  KnnScratch s;
  s.bits = knn_bits(N);
  s.M = (size_t)1 << (3 * s.bits);
  s.nb = (N + 63) / 64;
  s.nsb = (s.nb + 63) / 64;
  char* p = (char*)base;
  size_t o = 0;
  auto take = [&](size_t n) { char* r = p + o; o += (n + 255) / 256 * 256; return r; };
  s.bounds = (KnnBounds*)take(sizeof(KnnBounds));
  s.block_sum = (unsigned*)take(4096 * 4);
  s.keys = (unsigned*)take((size_t)N * 4);
  s.count = (unsigned*)take(s.M * 4);
  s.start = (unsigned*)take(s.M * 4);
  s.sorted = (float4*)take((size_t)N * 16);
  s.blo = (float4*)take((size_t)s.nb * 16);
  s.bhi = (float4*)take((size_t)s.nb * 16);
  s.slo = (float4*)take((size_t)s.nsb * 16);
  s.shi = (float4*)take((size_t)s.nsb * 16);
  s.bytes = o;
  return s;
}

}  // namespace sfgs

using namespace sfgs;

extern "C" size_t sfgs_knn_scratch_bytes(int32_t N) {
  if (N <= KNN_BRUTE_MAX) return 0;
  return knn_layout(nullptr, N).bytes;
}

extern "C" int sfgs_knn_dist2(const float* xyz, int32_t N, float* out, void* scratch, size_t scratch_sz, void* stream_) {
  SFGS_REQUIRE(N >= 0, SFGS_E_ARG, "negative point count");
  if (N == 0) return SFGS_OK;
  SFGS_REQUIRE(xyz && out, SFGS_E_ARG, "NULL argument");
  hipStream_t stream = (hipStream_t)stream_;
  ProfScope ps_(KID_KNN, stream);
  // option "knn" = "brute" (sfgs_set_option): test knob (the exact reference every size is compared with)
  if (N <= KNN_BRUTE_MAX || option(OPT_KNN) != 0) {
    hipLaunchKernelGGL(knn_dist2_kernel, dim3((N + KNN_BLOCK - 1) / KNN_BLOCK), dim3(KNN_BLOCK), 0, stream, xyz, N, out);
    SFGS_POST_LAUNCH("knn_dist2", stream, 0);
    return SFGS_OK;
  }
  const size_t need = sfgs_knn_scratch_bytes(N);
  SFGS_REQUIRE(scratch && scratch_sz >= need, SFGS_E_CAPACITY, "knn scratch: %zu bytes given, %zu needed", scratch_sz, need);
  SFGS_REQUIRE(((uintptr_t)scratch & 255) == 0, SFGS_E_ARG, "knn scratch must be 256-byte aligned");
  const KnnScratch s = knn_layout(scratch, N);
  const int nblk = (int)(s.M / KNN_SCAN_TILE), pb = (N + 255) / 256;
  (void)hipMemsetAsync(s.count, 0, s.M * 4, stream);
  hipLaunchKernelGGL(knn_bounds_kernel, dim3(1), dim3(1024), 0, stream, xyz, N, s.bits, s.bounds);
  hipLaunchKernelGGL(knn_count_kernel, dim3(pb), dim3(256), 0, stream, xyz, N, s.bits, s.bounds, s.keys, s.count);
  hipLaunchKernelGGL(knn_scan_sums_kernel, dim3(nblk), dim3(256), 0, stream, s.count, s.block_sum);
  hipLaunchKernelGGL(knn_scan_top_kernel, dim3(1), dim3(1024), 0, stream, s.block_sum, nblk);
  hipLaunchKernelGGL(knn_scan_apply_kernel, dim3(nblk), dim3(256), 0, stream, s.count, s.block_sum, s.start);
  hipLaunchKernelGGL(knn_scatter_kernel, dim3(pb), dim3(256), 0, stream, xyz, N, s.keys, s.start, s.count, s.sorted);
  hipLaunchKernelGGL(knn_boxes_kernel, dim3((N + 255) / 256), dim3(256), 0, stream, s.sorted, N, nullptr, nullptr, s.blo,
                     s.bhi);
  hipLaunchKernelGGL(knn_boxes_kernel, dim3((s.nb + 255) / 256), dim3(256), 0, stream, nullptr, s.nb, s.blo, s.bhi, s.slo,
                     s.shi);
  hipLaunchKernelGGL(knn_query_kernel, dim3((s.nb + 3) / 4), dim3(256), 0, stream, s.sorted, N, s.blo, s.bhi, s.nb, s.slo,
                     s.shi, s.nsb, out);
  SFGS_POST_LAUNCH("knn_dist2", stream, 0);
  return SFGS_OK;
}
