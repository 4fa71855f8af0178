// densify.hip -- GaussianModel.densify_and_prune as a handful of kernels (the densification half).
//
// Reference (scene/gaussian_model.py): densify_and_prune :707-742 -> densify_and_clone :686-705 -> densify_and_split
// :653-684 -> densification_postfix / cat_tensors_to_optimizer :603-651 (x2) -> prune_points :586-601 (x2). Every
// parameter, both Adam moments of every parameter and the statistics tensors go through boolean indexing, repeat, cat
// and again boolean indexing -- several dozen kernels, each mask -> index conversion with its own host sync -- and the
// quantile is a full sort that torch refuses above 16 M elements (the reference then silently uses Q = 0.99, :716-723).
//
// Here:
//   sfgs_select_kth        exact order statistics by radix select (3 histogram passes over the float bits + one pass
//                          for the successor): the two values torch.quantile interpolates between, for ANY N;
//   sfgs_densify_decide    one pass: per Gaussian the clone / split decision and which of its rows (original, clone,
//                          two children) survive the final prune; block sums of five counters;
//   sfgs_densify_scan      exclusive offsets + the five totals (one host read-back sizes the outputs);
//   sfgs_densify_gather    ONE multi-tensor launch writes every output tensor in the reference's final row order
//                          [surviving originals | clones | first children | second children]: parameters copied, Adam
//                          moments copied for originals and zero for new rows;
//   sfgs_densify_children  the two tensors whose child rows are computed, not copied: xyz = parent + R(q) * sample and
//                          raw scaling = log(scaling / 1.6).
// The DECISIONS are bit-exact with the reference's rule (tests/test_densify_masks.py, masks captured from the real
// methods); the children's positions use the caller's normal samples (torch.randn on the device: the reference draws
// from the same generator, so the random stream differs in order only).
#include "sfgs_internal.h"

namespace sfgs {

// ---------------------------------------------------------------------------------------------------------------------
// radix select on non-negative floats (their bit patterns order like unsigned integers)
constexpr int SEL_NT = 256, SEL_PER_THREAD = 16, SEL_CHUNK = SEL_NT * SEL_PER_THREAD;
constexpr int SEL_BINS = 2048;

struct SelectState {          // device
  unsigned long long k;       // rank still to resolve inside the current prefix
  unsigned prefix;            // value bits decided so far
  unsigned mask;              // which bits those are
  unsigned long long mult;    // after the last pass: multiplicity of the selected value
  unsigned hist[SEL_BINS];
  unsigned succ;              // bits of the smallest value greater than the selected one (0xffffffff: none)
};

__global__ void __launch_bounds__(256) select_init_kernel(SelectState* st, const float* rank_lo, long long N) {
  if (threadIdx.x == 0) {
    float r = rank_lo[0];
    long long k = (long long)floorf(r);
    if (!(r >= 0.f)) k = 0;
    if (k > N - 1) k = N - 1;
    st->k = (unsigned long long)k; st->prefix = 0u; st->mask = 0u; st->mult = 0ull; st->succ = 0xffffffffu;
  }
  for (int i = threadIdx.x; i < SEL_BINS; i += 256) st->hist[i] = 0u;
}

template <int SHIFT, int BITS>
__global__ void __launch_bounds__(SEL_NT)
select_hist_kernel(const float* __restrict__ v, long long N, SelectState* st) {
  __shared__ unsigned h[1 << BITS];
  for (int i = threadIdx.x; i < (1 << BITS); i += SEL_NT) h[i] = 0u;
  __syncthreads();
  const unsigned prefix = st->prefix, mask = st->mask;
  const long long base = (long long)blockIdx.x * SEL_CHUNK;
  // loads first, all in flight (unconditional, clamped index), then the histogram: a load under `if (i < N)` gets an
  // exec-masked block of its own with s_waitcnt vmcnt(0) behind it
  unsigned bv[SEL_PER_THREAD];
#pragma unroll
  for (int k = 0; k < SEL_PER_THREAD; ++k)
    bv[k] = __float_as_uint(v[min(base + (long long)k * SEL_NT + threadIdx.x, N - 1)]);
#pragma unroll
  for (int k = 0; k < SEL_PER_THREAD; ++k) {
    const long long i = base + (long long)k * SEL_NT + threadIdx.x;
    if (i < N && (bv[k] & mask) == prefix) atomicAdd(&h[(bv[k] >> SHIFT) & ((1u << BITS) - 1u)], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < (1 << BITS); i += SEL_NT)
    if (h[i]) atomicAdd(&st->hist[i], h[i]);
}

template <int SHIFT, int BITS>
__global__ void __launch_bounds__(1024) select_pick_kernel(SelectState* st) {
  // one workgroup: find the bin that holds rank k (serial over wave-level partial sums: 2048 bins, negligible)
  __shared__ unsigned long long cum[1 << BITS];
  const int n = 1 << BITS;
  for (int i = threadIdx.x; i < n; i += 1024) cum[i] = st->hist[i];
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long run = 0, k = st->k;
    int bin = n - 1;
    for (int i = 0; i < n; ++i) {
      const unsigned long long c = cum[i];
      if (k < run + c) { bin = i; break; }
      run += c;
    }
    st->k = k - run;
    st->mult = cum[bin];
    st->prefix |= (unsigned)bin << SHIFT;
    st->mask |= ((1u << BITS) - 1u) << SHIFT;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < SEL_BINS; i += 1024) st->hist[i] = 0u;
}

__global__ void __launch_bounds__(SEL_NT)
select_successor_kernel(const float* __restrict__ v, long long N, SelectState* st) {
  const unsigned a = st->prefix;
  unsigned best = 0xffffffffu;
  const long long base = (long long)blockIdx.x * SEL_CHUNK;
  unsigned bv[SEL_PER_THREAD];
#pragma unroll
  for (int k = 0; k < SEL_PER_THREAD; ++k)   // (a clamped index repeats the last element: harmless for a minimum)
    bv[k] = __float_as_uint(v[min(base + (long long)k * SEL_NT + threadIdx.x, N - 1)]);
#pragma unroll
  for (int k = 0; k < SEL_PER_THREAD; ++k)
    if (bv[k] > a && bv[k] < best) best = bv[k];
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) best = min(best, (unsigned)__shfl_xor((int)best, d));
  if (lane_id() == 0 && best != 0xffffffffu) atomicMin(&st->succ, best);
}

__global__ void select_finish_kernel(const SelectState* st, const float* rank, float* out2) {
  if (threadIdx.x == 0) {
    const float a = __uint_as_float(st->prefix);
    // value at rank lo + 1: the same value while duplicates remain, otherwise its successor
    const bool dup = st->k + 1 < st->mult;
    const float b = (dup || st->succ == 0xffffffffu) ? a : __uint_as_float(st->succ);
    const float r = rank[0];
    out2[0] = a;                                   // value at rank floor(r)
    out2[1] = ceilf(r) > floorf(r) ? b : a;        // value at rank ceil(r)
  }
}

__global__ void __launch_bounds__(256)
densify_masks_kernel(long long N, const unsigned char* __restrict__ code, unsigned char* __restrict__ clone_out,
                     unsigned char* __restrict__ split_out, unsigned char* __restrict__ keep_out /* [N][3] or null */);

// ---------------------------------------------------------------------------------------------------------------------
constexpr int DN_NT = 256;
enum { DN_KEEP_ORIG = 1, DN_KEEP_CLONE = 2, DN_KEEP_CHILD = 4, DN_CLONE = 8, DN_SPLIT = 16 };
constexpr int DN_CATS = 5;   // scan order: keep_orig, keep_clone, keep_child, clone (raw), split (raw)

template <typename OT>
__global__ void __launch_bounds__(DN_NT)
densify_decide_kernel(long long N, const float* __restrict__ gnorm, const float* __restrict__ gabs,
                      const float* __restrict__ scaling, const OT* __restrict__ opacity, const float* __restrict__ Q_dev,
                      float Q_host, float max_grad, double min_opacity, float dense_thr, float big_thr, int use_big,
                      unsigned char* __restrict__ code, unsigned* __restrict__ block_sum /* [NB][5] */) {
  __shared__ unsigned smem[DN_NT / 64 + 1];
  const long long i = (long long)blockIdx.x * DN_NT + threadIdx.x;
  unsigned c = 0;
  if (i < N) {
    const float Q = Q_dev ? Q_dev[0] : Q_host;
    const float s0 = scaling[3 * i], s1 = scaling[3 * i + 1], s2 = scaling[3 * i + 2];
    const float smax = fmaxf(fmaxf(s0, s1), s2);
    const bool sel = (gnorm[i] >= max_grad) || (gabs[i] >= Q);              // :688-690 / :656-662
    const bool clone = sel && (smax <= dense_thr);                            // :691-692
    const bool split = sel && (smax > dense_thr);                             // :663-664
    // :730 `get_opacity < min_opacity`: torch compares in the tensor's dtype (the Python scalar is rounded to it)
    bool low;
    if constexpr (sizeof(OT) == 8) low = opacity[i] < min_opacity;
    else low = opacity[i] < (float)min_opacity;
    const bool prune_self = low || (use_big && smax > big_thr);               // :731-735 (max_radii2D was reset: never)
    // children: scaling / (0.8 * 2) through the inverse activation and back (:672 then get_scaling). On the GPU torch
    // divides a tensor by a Python scalar by MULTIPLYING with the reciprocal formed in double (1 / 1.6 -> 0.625f); the
    // CPU path divides -- 1 ulp apart for some inputs. This kernel replaces the GPU path.
    const float cmax = expf(logf(smax * 0.625f));
    const bool prune_child = low || (use_big && cmax > big_thr);
    if (!split && !prune_self) c |= DN_KEEP_ORIG;
    if (clone && !prune_self) c |= DN_KEEP_CLONE;
    if (split && !prune_child) c |= DN_KEEP_CHILD;
    if (clone) c |= DN_CLONE;
    if (split) c |= DN_SPLIT;
    code[i] = (unsigned char)c;
  }
#pragma unroll
  for (int k = 0; k < DN_CATS; ++k) {
    unsigned total;
    block_excl_scan_u32<DN_NT>((c >> k) & 1u, &total, smem);
    if (threadIdx.x == 0) block_sum[(size_t)blockIdx.x * DN_CATS + k] = total;
  }
}

__global__ void __launch_bounds__(1024)
densify_scan_blocks_kernel(unsigned* __restrict__ block_sum, long long NB, unsigned long long* __restrict__ totals) {
  __shared__ unsigned smem[1024 / 64 + 1];
  for (int k = 0; k < DN_CATS; ++k) {
    unsigned long long carry = 0;
    for (long long b0 = 0; b0 < NB; b0 += 1024) {
      const long long i = b0 + threadIdx.x;
      const unsigned v = i < NB ? block_sum[i * DN_CATS + k] : 0u;
      unsigned total;
      const unsigned ex = block_excl_scan_u32<1024>(v, &total, smem);
      if (i < NB) block_sum[i * DN_CATS + k] = (unsigned)carry + ex;
      carry += total;
    }
    if (threadIdx.x == 0) totals[k] = carry;
  }
}

__global__ void __launch_bounds__(DN_NT)
densify_index_kernel(long long N, const unsigned char* __restrict__ code, const unsigned* __restrict__ block_sum,
                     unsigned* __restrict__ idx /* [N][4]: orig, clone, child (kept ranks), split (raw rank) */) {
  __shared__ unsigned smem[DN_NT / 64 + 1];
  const long long i = (long long)blockIdx.x * DN_NT + threadIdx.x;
  const unsigned c = i < N ? code[i] : 0u;
  const int cats[4] = {0, 1, 2, 4};
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    unsigned total;
    const unsigned ex = block_excl_scan_u32<DN_NT>((c >> cats[q]) & 1u, &total, smem);
    if (i < N) idx[4 * i + q] = block_sum[(size_t)blockIdx.x * DN_CATS + cats[q]] + ex;
  }
}

constexpr int DG_MAX_TENSORS = 40;
constexpr int DG_WORDS_PER_THREAD = 8, DG_WORD_CHUNK = DN_NT * DG_WORDS_PER_THREAD;
struct DensifyTable {
  const void* src[DG_MAX_TENSORS];
  void* dst[DG_MAX_TENSORS];
  unsigned row_units[DG_MAX_TENSORS];   // 4-byte words per row
  unsigned zero_new[DG_MAX_TENSORS];    // 1: new rows (clones, children) are zero (Adam moments)
  unsigned block_end[DG_MAX_TENSORS];
  int count;
};

__global__ void __launch_bounds__(DN_NT)
densify_gather_kernel(const DensifyTable tab, long long N, const unsigned char* __restrict__ code,
                      const unsigned* __restrict__ idx, unsigned n_orig, unsigned n_clone, unsigned n_child) {
  int ti = 0;
  while (ti + 1 < tab.count && blockIdx.x >= tab.block_end[ti]) ++ti;
  const unsigned first = ti ? tab.block_end[ti - 1] : 0u;
  const unsigned ru = tab.row_units[ti];
  const bool zero_new = tab.zero_new[ti] != 0;
  const uint32_t* __restrict__ src = (const uint32_t*)tab.src[ti];
  uint32_t* __restrict__ dst = (uint32_t*)tab.dst[ti];
  const long long words = N * (long long)ru;
  const long long base = (long long)(blockIdx.x - first) * DG_WORD_CHUNK;
  const long long row0 = base / ru;
  const unsigned rem0 = (unsigned)(base - row0 * ru);
  const float inv = 1.0f / (float)ru;
  // loads first (row codes; then source words and target indices, unconditional from clamped indices), stores last: one
  // loop with the loads under `if (code & ...)` serialises three dependent round trips to memory per word
  long long rowv[DG_WORDS_PER_THREAD];
  unsigned colv[DG_WORDS_PER_THREAD], cv[DG_WORDS_PER_THREAD];
#pragma unroll
  for (int k = 0; k < DG_WORDS_PER_THREAD; ++k) {
    const unsigned x = rem0 + threadIdx.x + k * DN_NT;
    unsigned q = (unsigned)((float)x * inv);
    if (q * ru > x) --q; else if ((q + 1) * ru <= x) ++q;
    rowv[k] = min(row0 + (long long)q, N - 1);
    colv[k] = x - q * ru;
    cv[k] = code[rowv[k]];
  }
  uint32_t val[DG_WORDS_PER_THREAD];
  uint4 ixv[DG_WORDS_PER_THREAD];
#pragma unroll
  for (int k = 0; k < DG_WORDS_PER_THREAD; ++k) {
    val[k] = src[min(base + (long long)(threadIdx.x + k * DN_NT), words - 1)];
    ixv[k] = *reinterpret_cast<const uint4*>(idx + 4 * rowv[k]);
  }
#pragma unroll
  for (int k = 0; k < DG_WORDS_PER_THREAD; ++k) {
    const long long w = base + (long long)(threadIdx.x + k * DN_NT);
    const unsigned c = cv[k], col = colv[k];
    if (w < words && (c & (DN_KEEP_ORIG | DN_KEEP_CLONE | DN_KEEP_CHILD))) {
      const uint32_t nv = zero_new ? 0u : val[k];
      if (c & DN_KEEP_ORIG) dst[(long long)ixv[k].x * ru + col] = val[k];
      if (c & DN_KEEP_CLONE) dst[(long long)(n_orig + ixv[k].y) * ru + col] = nv;
      if (c & DN_KEEP_CHILD) {
        dst[(long long)(n_orig + n_clone + ixv[k].z) * ru + col] = nv;
        dst[(long long)(n_orig + n_clone + n_child + ixv[k].z) * ru + col] = nv;
      }
    }
  }
}

// child rows of xyz and raw scaling (:666-672): samples [2 * n_split_raw, 3] hold std * z for child k of the parent with
// raw split rank r at row k * n_split_raw + r, exactly the layout of the reference's `samples`; with unit_samples they
// hold z alone and the parent's scaling is applied here
__global__ void __launch_bounds__(DN_NT)
densify_children_kernel(long long N, const unsigned char* __restrict__ code, const unsigned* __restrict__ idx,
                        const float* __restrict__ xyz, const float* __restrict__ rot_raw, const float* __restrict__ scaling,
                        const float* __restrict__ samples, int unit_samples, unsigned n_split_raw, unsigned n_orig,
                        unsigned n_clone,
                        unsigned n_child, float* __restrict__ xyz_out, float* __restrict__ scaling_raw_out) {
  const long long i = (long long)blockIdx.x * DN_NT + threadIdx.x;
  if (i >= N || !(code[i] & DN_KEEP_CHILD)) return;
  float q[4] = {rot_raw[4 * i], rot_raw[4 * i + 1], rot_raw[4 * i + 2], rot_raw[4 * i + 3]};
  const float nrm = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);   // build_rotation normalises (:80-83)
  for (int k = 0; k < 4; ++k) q[k] /= nrm;
  float R[9];
  quat_to_rot(q, R);
  const unsigned rr = idx[4 * i + 3], rk = idx[4 * i + 2];
  for (int child = 0; child < 2; ++child) {
    const float* sp = samples + 3 * ((size_t)child * n_split_raw + rr);
    float s[3] = {sp[0], sp[1], sp[2]};
    if (unit_samples) { for (int a = 0; a < 3; ++a) s[a] *= scaling[3 * i + a]; }   // torch.normal(0, std) = std * z
    const size_t row = (size_t)n_orig + n_clone + (size_t)child * n_child + rk;
    for (int a = 0; a < 3; ++a)
      xyz_out[3 * row + a] = R[3 * a] * s[0] + R[3 * a + 1] * s[1] + R[3 * a + 2] * s[2] + xyz[3 * i + a];
    for (int a = 0; a < 3; ++a) scaling_raw_out[3 * row + a] = logf(scaling[3 * i + a] * 0.625f);
  }
}

__global__ void __launch_bounds__(256)
densify_masks_kernel(long long N, const unsigned char* __restrict__ code, unsigned char* __restrict__ clone_out,
                     unsigned char* __restrict__ split_out, unsigned char* __restrict__ keep_out) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  const unsigned c = code[i];
  if (clone_out) clone_out[i] = (c & DN_CLONE) ? 1 : 0;
  if (split_out) split_out[i] = (c & DN_SPLIT) ? 1 : 0;
  if (keep_out) {
    keep_out[3 * i] = (c & DN_KEEP_ORIG) ? 1 : 0; keep_out[3 * i + 1] = (c & DN_KEEP_CLONE) ? 1 : 0;
    keep_out[3 * i + 2] = (c & DN_KEEP_CHILD) ? 1 : 0;
  }
}

static inline long long dn_blocks(long long N) { return (N + DN_NT - 1) / DN_NT; }
struct DensifyScratch { unsigned long long* totals; unsigned char* code; unsigned* block_sum; unsigned* idx; };
static inline size_t dn_scratch_bytes(long long N) {
  return 256 + align_up((size_t)N, 256) + align_up((size_t)dn_blocks(N) * DN_CATS * 4, 256) + align_up((size_t)N * 16, 256);
}
static inline DensifyScratch dn_view(void* scratch, long long N) {
  char* p = (char*)scratch;
  DensifyScratch s;
  s.totals = (unsigned long long*)p; p += 256;
  s.code = (unsigned char*)p; p += align_up((size_t)N, 256);
  s.block_sum = (unsigned*)p; p += align_up((size_t)dn_blocks(N) * DN_CATS * 4, 256);
  s.idx = (unsigned*)p;
  return s;
}

}  // namespace sfgs

using namespace sfgs;

extern "C" size_t sfgs_select_scratch_bytes(void) { return align_up(sizeof(SelectState), 256); }

extern "C" int sfgs_select_kth(const float* values, int64_t N, const float* rank_lo_dev, float* out2, void* scratch,
                               size_t scratch_sz, void* stream_) {
  SFGS_REQUIRE(N > 0 && values && rank_lo_dev && out2 && scratch, SFGS_E_ARG, "bad argument");
  SFGS_REQUIRE(scratch_sz >= sfgs_select_scratch_bytes(), SFGS_E_CAPACITY, "select scratch too small");
  hipStream_t stream = (hipStream_t)stream_;
  SelectState* st = (SelectState*)scratch;
  const unsigned nb = (unsigned)((N + SEL_CHUNK - 1) / SEL_CHUNK);
  { ProfScope ps_(KID_DENSIFY, stream);
    hipLaunchKernelGGL(select_init_kernel, dim3(1), dim3(256), 0, stream, st, rank_lo_dev, (long long)N);
    hipLaunchKernelGGL((select_hist_kernel<21, 11>), dim3(nb), dim3(SEL_NT), 0, stream, values, (long long)N, st);
    hipLaunchKernelGGL((select_pick_kernel<21, 11>), dim3(1), dim3(1024), 0, stream, st);
    hipLaunchKernelGGL((select_hist_kernel<10, 11>), dim3(nb), dim3(SEL_NT), 0, stream, values, (long long)N, st);
    hipLaunchKernelGGL((select_pick_kernel<10, 11>), dim3(1), dim3(1024), 0, stream, st);
    hipLaunchKernelGGL((select_hist_kernel<0, 10>), dim3(nb), dim3(SEL_NT), 0, stream, values, (long long)N, st);
    hipLaunchKernelGGL((select_pick_kernel<0, 10>), dim3(1), dim3(1024), 0, stream, st);
    hipLaunchKernelGGL(select_successor_kernel, dim3(nb), dim3(SEL_NT), 0, stream, values, (long long)N, st);
    hipLaunchKernelGGL(select_finish_kernel, dim3(1), dim3(64), 0, stream, st, rank_lo_dev, out2); }
  SFGS_POST_LAUNCH("select_kth", stream, 0);
  return SFGS_OK;
}

extern "C" size_t sfgs_densify_scratch_bytes(int64_t N) { return N < 0 ? 0 : dn_scratch_bytes(N); }

extern "C" int sfgs_densify_decide(int64_t N, const float* grad_norm, const float* grad_abs, const float* scaling,
                                   const void* opacity, int32_t opacity_is_f64, const float* Q_dev, float Q_host,
                                   float max_grad, double min_opacity, float dense_threshold, float big_threshold,
                                   int32_t use_big_threshold, void* scratch, size_t scratch_sz, int64_t totals_out[5],
                                   void* stream_) {
  SFGS_REQUIRE(N >= 0 && N < (1ll << 31), SFGS_E_ARG, "row count out of range");
  SFGS_REQUIRE(scratch && scratch_sz >= dn_scratch_bytes(N), SFGS_E_CAPACITY, "densify scratch too small");
  hipStream_t stream = (hipStream_t)stream_;
  const DensifyScratch s = dn_view(scratch, N);
  if (N == 0) {
    if (totals_out) for (int k = 0; k < 5; ++k) totals_out[k] = 0;
    return SFGS_OK;
  }
  SFGS_REQUIRE(grad_norm && grad_abs && scaling && opacity, SFGS_E_ARG, "NULL argument");
  const long long NB = dn_blocks(N);
  { ProfScope ps_(KID_DENSIFY, stream);
    if (opacity_is_f64)
      hipLaunchKernelGGL(densify_decide_kernel<double>, dim3((unsigned)NB), dim3(DN_NT), 0, stream, (long long)N, grad_norm,
                         grad_abs, scaling, (const double*)opacity, Q_dev, Q_host, max_grad, min_opacity, dense_threshold,
                         big_threshold, use_big_threshold, s.code, s.block_sum);
    else
      hipLaunchKernelGGL(densify_decide_kernel<float>, dim3((unsigned)NB), dim3(DN_NT), 0, stream, (long long)N, grad_norm,
                         grad_abs, scaling, (const float*)opacity, Q_dev, Q_host, max_grad, min_opacity, dense_threshold,
                         big_threshold, use_big_threshold, s.code, s.block_sum);
    hipLaunchKernelGGL(densify_scan_blocks_kernel, dim3(1), dim3(1024), 0, stream, s.block_sum, NB, s.totals);
    hipLaunchKernelGGL(densify_index_kernel, dim3((unsigned)NB), dim3(DN_NT), 0, stream, (long long)N, s.code, s.block_sum,
                       s.idx); }
  SFGS_POST_LAUNCH("densify_decide", stream, 0);
  if (totals_out) {  // the one host synchronisation of a densification: the caller sizes its outputs with these
    unsigned long long h[DN_CATS];
    SFGS_CHECK_HIP(hipMemcpyAsync(h, s.totals, sizeof(h), hipMemcpyDeviceToHost, stream));
    SFGS_CHECK_HIP(hipStreamSynchronize(stream));
    for (int k = 0; k < DN_CATS; ++k) totals_out[k] = (int64_t)h[k];
  }
  return SFGS_OK;
}

extern "C" int sfgs_densify_masks(int64_t N, const void* scratch, unsigned char* clone_out, unsigned char* split_out,
                                  unsigned char* keep_out, void* stream_) {
  // the raw clone / split decisions (and, optionally, [N][3] keep flags: original, clone, children) as byte masks
  SFGS_REQUIRE(N >= 0 && scratch, SFGS_E_ARG, "bad argument");
  if (N == 0) return SFGS_OK;
  hipStream_t stream = (hipStream_t)stream_;
  const DensifyScratch s = dn_view(const_cast<void*>(scratch), N);
  hipLaunchKernelGGL(densify_masks_kernel, dim3((unsigned)dn_blocks(N)), dim3(DN_NT), 0, stream, (long long)N, s.code,
                     clone_out, split_out, keep_out);
  SFGS_POST_LAUNCH("densify_masks", stream, 0);
  return SFGS_OK;
}

extern "C" int sfgs_densify_gather(int64_t N, const void* scratch, const int64_t totals[5],
                                   const SfgsDensifyTensor* tensors, int32_t count, void* stream_) {
  SFGS_REQUIRE(N >= 0 && N < (1ll << 31) && count >= 0 && totals, SFGS_E_ARG, "bad argument");
  if (N == 0 || count == 0) return SFGS_OK;
  SFGS_REQUIRE(scratch && tensors, SFGS_E_ARG, "NULL argument");
  hipStream_t stream = (hipStream_t)stream_;
  const DensifyScratch s = dn_view(const_cast<void*>(scratch), N);
  int i = 0;
  while (i < count) {
    DensifyTable tab;
    tab.count = 0;
    uint64_t blocks = 0;
    for (; i < count && tab.count < DG_MAX_TENSORS; ++i) {
      const SfgsDensifyTensor& t = tensors[i];
      SFGS_REQUIRE(t.row_bytes >= 0 && t.row_bytes < (1ll << 22) && (t.row_bytes & 3) == 0, SFGS_E_ARG,
                   "tensor %d: row size must be a multiple of 4 bytes", i);
      if (t.row_bytes == 0) continue;
      SFGS_REQUIRE(t.src && t.dst && (((uintptr_t)t.src | (uintptr_t)t.dst) & 3) == 0, SFGS_E_ARG, "tensor %d: bad pointer", i);
      const uint64_t ru = (uint64_t)t.row_bytes / 4;
      const uint64_t nb = ((uint64_t)N * ru + DG_WORD_CHUNK - 1) / DG_WORD_CHUNK;
      if (blocks + nb >= (1ull << 31)) break;
      blocks += nb;
      tab.src[tab.count] = t.src; tab.dst[tab.count] = t.dst; tab.row_units[tab.count] = (unsigned)ru;
      tab.zero_new[tab.count] = t.zero_new_rows ? 1u : 0u; tab.block_end[tab.count] = (unsigned)blocks;
      ++tab.count;
    }
    if (!blocks) continue;
    { ProfScope ps_(KID_DENSIFY, stream);
      hipLaunchKernelGGL(densify_gather_kernel, dim3((unsigned)blocks), dim3(DN_NT), 0, stream, tab, (long long)N, s.code,
                         s.idx, (unsigned)totals[0], (unsigned)totals[1], (unsigned)totals[2]); }
    SFGS_POST_LAUNCH("densify_gather", stream, 0);
  }
  return SFGS_OK;
}

extern "C" int sfgs_densify_children(int64_t N, const void* scratch, const int64_t totals[5], const float* xyz,
                                     const float* rotation_raw, const float* scaling, const float* samples,
                                     int32_t unit_samples, float* xyz_out, float* scaling_raw_out, void* stream_) {
  SFGS_REQUIRE(N >= 0 && totals, SFGS_E_ARG, "bad argument");
  if (N == 0 || totals[2] == 0) return SFGS_OK;
  SFGS_REQUIRE(scratch && xyz && rotation_raw && scaling && samples && xyz_out && scaling_raw_out, SFGS_E_ARG, "NULL argument");
  hipStream_t stream = (hipStream_t)stream_;
  const DensifyScratch s = dn_view(const_cast<void*>(scratch), N);
  { ProfScope ps_(KID_DENSIFY, stream);
    hipLaunchKernelGGL(densify_children_kernel, dim3((unsigned)dn_blocks(N)), dim3(DN_NT), 0, stream, (long long)N, s.code,
                       s.idx, xyz, rotation_raw, scaling, samples, (int)unit_samples, (unsigned)totals[4], (unsigned)totals[0],
                       (unsigned)totals[1], (unsigned)totals[2], xyz_out, scaling_raw_out); }
  SFGS_POST_LAUNCH("densify_children", stream, 0);
  return SFGS_OK;
}
