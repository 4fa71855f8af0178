// raster_bwd.hip -- backward of the Gaussian-splat rasterizer for gfx950 (MI355X).
//
// Replaces the autograd backward of diff_gauss.GaussianRasterizer (reference: triggered at
// train.py:279,845; gradient contract scene/gaussian_model.py:744-749). Two kernels:
//
//   composite_bwd   (composite_bwd.hip, a translation unit of its own: it is compiled with another scheduling strategy)
//                   one wave per 8x8 tile, back to front over the tile's sorted list; ONE 48-byte record of raw
//                   gradient sums per (splat, tile) duplicate, no float atomics.
//   preprocess_bwd  one thread per Gaussian: sums its duplicates' lines in a fixed order and applies
//                   the 2D -> 3D chain rule (raster_math.h: preprocess_backward_one).
//
// Compile with -ffp-contract=off (the forward's alpha/skip decisions must be reproduced exactly; FMA
// only where spelled fmaf, identically to raster_fwd.hip).
#include <cstring>

#include "act_math.h"
#include "sfgs_internal.h"

namespace sfgs {

// Dead list entries -- behind their tile's last contributor: opaque surfaces seen at a low angle leave a third to four
// fifths of every list dead, near-camera overdraw 95 % -- own a gradient record like every other entry, and preprocess_bwd
// adds up a Gaussian's records. Three ways to make a dead record read as zero, in the order they were built:
//   * composite_bwd stores a zero record per dead entry: ~38 ps each (near-camera regime: 4.4 of 4.8 ms of composite_bwd);
//   * (rounds 2 - 5) zero the WHOLE record array with streaming stores when more than 30 % are dead: ~10 ps per entry, dead
//     or alive -- and preprocess_bwd still READS all of it (low elevation, 2 M Gaussians: 79 % of 17.4 M entries dead,
//     0.83 GB written and 0.83 GB read back for nothing);
//   * (round 6) LIVE FLAGS: one byte per duplicate index behind the records. This kernel clears the bytes (1 / 48 of the
//     array), composite_bwd sets the byte of every record it writes and writes nothing for the dead, and the readers
//     (dupgrad_reduce_kernel, preprocess_bwd) fetch a record only where the byte is set -- everything else is read from
//     one line of zeros. Sums are bit-identical: the skipped addends were +0.
// The training forward leaves every tile's dead-entry count (tile_dead); this kernel sums them, decides
// (prefill_wanted), and publishes the decision in hdr[HDR_PREFILLED] for the three kernels behind it.
// (On the headline scene 0.1 % are dead: the kernel returns after the sum.)
__global__ void __launch_bounds__(256)
dupgrad_prefill_kernel(int T8, const uint16_t* __restrict__ tile_dead, unsigned long long n_dup, int mode,
                       uint4* __restrict__ live16, uint4* __restrict__ zero_line, unsigned long long* __restrict__ hdr,
                       unsigned long long* __restrict__ feedback, const unsigned long long* __restrict__ dup_pool,
                       unsigned long long dup_capacity, unsigned npools) {
  __shared__ unsigned part[4];
  // every workgroup sums the per-tile counts for itself (2 bytes per tile, 16-byte loads: 64 KB from L2 at 1080p)
  unsigned dead = 0;   // <= 65535 * T8 < 2^32 up to 65 k tiles... accumulate in 64 bits across lanes below
  unsigned long long dead64 = 0ull;
  const uint4* v = reinterpret_cast<const uint4*>(tile_dead);
  const int n8 = T8 / 8;
  for (int i = threadIdx.x; i < n8; i += 256) {
    const uint4 q = v[i];
    dead += (q.x & 0xffffu) + (q.x >> 16) + (q.y & 0xffffu) + (q.y >> 16) + (q.z & 0xffffu) + (q.z >> 16) +
            (q.w & 0xffffu) + (q.w >> 16);
    if (dead > 0x7fffffffu) { dead64 += dead; dead = 0; }
  }
  for (int t = n8 * 8 + threadIdx.x; t < T8; t += 256) dead += tile_dead[t];
  dead64 += dead;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) dead64 += (unsigned long long)__shfl_xor((long long)dead64, d);
  // saturating 32-bit partials are enough for the decision
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = (unsigned)min(dead64, 0xffffffffull >> 2);
  __syncthreads();
  const unsigned long long total = (unsigned long long)part[0] + part[1] + part[2] + part[3];
  const bool fill = mode == 1 ? true : mode == 2 ? false : prefill_wanted(total, n_dup);   // mode: test knob
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    hdr[HDR_PREFILLED] = fill ? 1ull : 0ull;
    if (feedback) feedback[FB_PREFILLED] = fill ? 1ull : 0ull;   // for the next frame's plan (SfgsFrame.feedback)
  }
  if (!fill) return;
  const uint4 zero4 = make_uint4(0u, 0u, 0u, 0u);
  if (blockIdx.x == 0 && threadIdx.x < 16) zero_line[threadIdx.x] = zero4;
  const unsigned long long R = dup_capacity / npools;   // the used part of every pool's index range (sfgs_internal.h)
  for (unsigned q = 0; q < npools; ++q) {
    // whole 16-byte words around the pool's used byte range (a word shared with the next pool's range is cleared twice:
    // every flag of a used range starts at zero, the gaps are never read)
    const unsigned long long b0 = (unsigned long long)q * R, b1 = b0 + min(dup_pool[q * DP_STRIDE], R);
    const size_t w0 = (size_t)(b0 / 16), w1 = (size_t)((b1 + 15) / 16);
    for (size_t i = w0 + (size_t)blockIdx.x * 256 + threadIdx.x; i < w1; i += (size_t)gridDim.x * 256) live16[i] = zero4;
  }
}

// Parallel pre-reduction of the records of Gaussians with more than BWD_BIG duplicates (sfgs_internal.h): one
// workgroup per 1024-record chunk sums the chunk in a fixed order and overwrites the chunk's FIRST record with the sum.
__global__ void __launch_bounds__(256)
dupgrad_reduce_kernel(const unsigned long long* __restrict__ hdr, const uint2* __restrict__ big_chunks,
                      unsigned chunk_cap, const uint2* __restrict__ dup, float4* __restrict__ dupgrad,
                      const uint8_t* __restrict__ live) {
  constexpr int NF = DUPGRAD_FLOATS;   // every float of the record is summed in place (padding floats are zeros)
  __shared__ float part[4][NF];
  const unsigned n_chunks = (unsigned)min(hdr[HDR_BIG_CHUNKS], (unsigned long long)chunk_cap);
  const bool flags = live != nullptr && hdr[HDR_PREFILLED] != 0ull;   // live-flag frame: unflagged records hold garbage
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (unsigned j = blockIdx.x; j < n_chunks; j += gridDim.x) {
    const uint2 gc = big_chunks[j];
    const uint2 dr = dup[gc.x];
    const unsigned first = gc.y * BWD_CHUNK;
    const unsigned n = min(BWD_CHUNK, dr.y - first);
    const size_t d0 = (size_t)dr.x + first;
    float v[NF];
#pragma unroll
    for (int i = 0; i < NF; ++i) v[i] = 0.f;
    static_assert(BWD_CHUNK == 4 * 256, "four records per thread");
    bool take[4];   // (the four flags of a thread in flight together, then the records of the flagged ones)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const unsigned i = threadIdx.x + 256u * j;
      take[j] = i < n && (!flags || live[d0 + min(i, n - 1u)] != 0);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (!take[j]) continue;
      const unsigned i = threadIdx.x + 256u * j;
#pragma unroll
      for (int q = 0; q < DG_F4; ++q) {
        const float4 x = dupgrad[(d0 + i) * DG_F4 + q];
        v[4 * q] += x.x; v[4 * q + 1] += x.y; v[4 * q + 2] += x.z; v[4 * q + 3] += x.w;
      }
    }
#pragma unroll
    for (int i = 0; i < NF; ++i) {
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) v[i] += __shfl_xor(v[i], d);
    }
    __syncthreads();   // every record of the chunk has been read (and `part` is free again)
    if (lane == 0) {
#pragma unroll
      for (int i = 0; i < NF; ++i) part[wave][i] = v[i];
    }
    __syncthreads();
    if (threadIdx.x < NF) {
      const float s = ((part[0][threadIdx.x] + part[1][threadIdx.x]) + part[2][threadIdx.x]) + part[3][threadIdx.x];
      reinterpret_cast<float*>(dupgrad + d0 * DG_F4)[threadIdx.x] = s;
    }
  }
}

// accumulators of one Gaussian's record sum: DG_F4 float4s, added componentwise in record layout
struct DupAcc {
  float4 q[DG_F4];
  __device__ __forceinline__ void zero() {
#pragma unroll
    for (int i = 0; i < DG_F4; ++i) q[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  __device__ __forceinline__ void add(const float4* __restrict__ r) {
#pragma unroll
    for (int i = 0; i < DG_F4; ++i) {
      const float4 x = r[i];
      q[i].x += x.x; q[i].y += x.y; q[i].z += x.z;
      if (DG_F4 == 3) q[i].w += x.w;      // the fourth float of a 64-byte record's quarters is padding
    }
  }
  // float f of the record in Grad2D order (gmx gmy absx absy | gA gB gC gop | r g b depth)
  __device__ __forceinline__ float get(int f) const {
    if constexpr (DG_F4 == 3) { const float4 v = q[f >> 2]; const int c = f & 3; return c == 0 ? v.x : c == 1 ? v.y : c == 2 ? v.z : v.w; }
    else { const float4 v = q[f & 3]; const int c = f >> 2; return c == 0 ? v.x : c == 1 ? v.y : v.z; }   // float 4 k + r at quarter r, slot k
  }
};

// one thread per Gaussian. K = SH coefficients stored per Gaussian (0: colors_precomp), DEG = active degree:
// compile-time so that the coefficient / gradient rows live in registers, not scratch.
// RAW (SfgsGaussians raw-parameter mode): `scales`, `rots`, `opac_` are the model's raw parameters; the activations are
// recomputed here and the chain rule continues through them, so g_scales / g_rots / g_opac_ receive the RAW parameters'
// gradients (what sfgs_prepass_backward would make of this kernel's non-raw outputs: the same functions, act_math.h).
// CM != 0 (SfgsGaussians.sh_dirs): the view direction is given per Gaussian and its gradient goes to g_sh_dirs instead of
// into g_means3D; CM == 1: coefficients and their gradients channel-major [N,3,K], CM == 2: [N,K,3].
// CMX = CM, or CM + 3 for SPLIT SH STORAGE (SfgsGaussians.shs_rest; CM 0 or 2): an instantiation of its own, because a
// run-time choice between the unsplit row's twelve 16-byte accesses and the split rows' 12-byte ones costs the UNSPLIT path
// 10 - 20 % at 16 coefficients (profiles/r4_split_sh_rows_ab.txt).
template <int K, int DEG, bool RAW, int CMX>
__global__ void __launch_bounds__(PRE_BLOCK)
preprocess_bwd_kernel(KFrame kf, int N, const float* __restrict__ means3D, const float* __restrict__ scales,
                      const float* __restrict__ rots, const void* __restrict__ opac_, const void* __restrict__ filt,
                      int raw_mask, const float* __restrict__ shs, const float* __restrict__ shs_rest,
                      const float* __restrict__ sh_dirs, int dirs_are_centers,
                      const int* __restrict__ radii,
                      const uint2* __restrict__ dup, const float4* __restrict__ dupgrad, const uint8_t* __restrict__ live,
                      size_t zero_f4, const unsigned long long* __restrict__ hdr,
                      float* __restrict__ g_means3D, float* __restrict__ g_means2D, float* __restrict__ g_scales,
                      float* __restrict__ g_rots, void* __restrict__ g_opac_, float* __restrict__ g_colors,
                      float* __restrict__ g_shs, float* __restrict__ g_shs_rest, float* __restrict__ g_sh_dirs) {
  constexpr bool SPLIT = CMX >= 3;
  constexpr int CM = SPLIT ? CMX - 3 : CMX;
  static_assert(!SPLIT || (K > 1 && CM != 1), "split SH storage: more than one coefficient, coefficient-major");
  constexpr int PB_CHUNK = 128;   // records per staging chunk and wave: 6 KB of LDS
  static_assert((PB_CHUNK * DG_F4) % 64 == 0, "whole load rounds");
  __shared__ float4 pb_stage[PRE_BLOCK / 64][PB_CHUNK * DG_F4];
  const int g = blockIdx.x * PRE_BLOCK + threadIdx.x;
  const bool valid = g < N;
  const int lane = threadIdx.x & 63;
  // ---- sum this Gaussian's per-duplicate records (fixed order -> deterministic) -----------------------------------
  // A splat that covers many tiles owns thousands of records: summing them in its own lane would leave the other 63
  // lanes idle for that long, so above COOP records the whole wave strides over them and reduces across lanes.
  constexpr unsigned COOP = 32;
  const bool vis = valid && radii[g] > 0;
  unsigned d0 = 0, cnt = 0;
  if (vis) { const uint2 dr = dup[g]; d0 = dr.x; cnt = dr.y; }
  // live-flag frame (dupgrad_prefill_kernel above): a record is fetched only where its flag is set; the others read
  // float4 `zero_f4` of the blob, a line of zeros (all the lanes of a dead stretch share ONE request)
  const bool flags = live != nullptr && hdr[HDR_PREFILLED] != 0ull;
  DupAcc acc;
  acc.zero();
  if (cnt > BWD_BIG) {   // pre-reduced by dupgrad_reduce_kernel: add the chunk heads (<= a few dozen)
    for (unsigned d = d0; d < d0 + cnt; d += BWD_CHUNK) acc.add(dupgrad + (size_t)d * DG_F4);
  }
  // Gaussians with at most COOP records (almost all): the records of a wave's Gaussians are CONTIGUOUS (the forward
  // reserves duplicate indices in thread order), so the wave streams them through LDS in chunks of PB_CHUNK records
  // with full-line loads -- lane i fetches the i-th 16-byte piece -- and every lane then adds up its own records from
  // LDS, in the same order as before (bit-identical sums). Read per lane straight from memory (64 different cache lines
  // per load instruction) the summation was 0.085 of the kernel's 0.14 ms.
  {
    const bool small = cnt > 0 && cnt <= COOP;
    const unsigned rb = small ? d0 : 0xffffffffu, re = small ? d0 + cnt : 0u;
    unsigned w0 = rb, w1 = re;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
      w0 = min(w0, (unsigned)__shfl_xor((int)w0, d));
      w1 = max(w1, (unsigned)__shfl_xor((int)w1, d));
    }
    float4* stage = pb_stage[threadIdx.x >> 6];
    for (unsigned c0 = w0; c0 < w1; c0 += PB_CHUNK) {   // w0 >= w1 when the wave has no such Gaussian
      const unsigned c1 = min(c0 + PB_CHUNK, w1);
      if (__ballot(small && rb < c1 && re > c0) == 0ull) continue;   // a stretch that belongs to bigger splats only
      const unsigned n4 = (c1 - c0) * DG_F4;
      // all of the chunk's loads in flight together: unconditional from a clamped index.
      // (Written as `if (u < n4) stage[u] = dupgrad[..]`, every load got its own exec-masked block with an s_waitcnt
      // vmcnt(0) behind it: six serialised round trips to memory per chunk -- found by reading the ISA, round 3.)
      constexpr int NLD = PB_CHUNK * DG_F4 / 64;
      float4 tmp[NLD];
      size_t src[NLD];   // float4 index of this lane's k-th piece
#pragma unroll
      for (int k = 0; k < NLD; ++k) src[k] = (size_t)c0 * DG_F4 + min((unsigned)(k * 64 + lane), n4 - 1u);
      if (flags) {   // (wave-uniform. The flag loads and the selects live in this block, the record loads below are common
                     // to both kinds of frame: with two sets of loads the compiler moved `tmp` to scratch memory)
        static_assert(DG_F4 == 3, "piece -> record: u / 3");
        unsigned char lv[NLD];
#pragma unroll
        for (int k = 0; k < NLD; ++k) lv[k] = live[(size_t)c0 + min((unsigned)(k * 64 + lane), n4 - 1u) / 3u];
        unsigned any = 0;
#pragma unroll
        for (int k = 0; k < NLD; ++k) { src[k] = lv[k] ? src[k] : zero_f4; any |= lv[k]; }
        if (__ballot(any != 0u) == 0ull) continue;   // a stretch without a single record: nothing to fetch, nothing to add
      }
#pragma unroll
      for (int k = 0; k < NLD; ++k) tmp[k] = dupgrad[src[k]];
#pragma unroll
      for (int k = 0; k < NLD; ++k) stage[k * 64 + lane] = tmp[k];   // unconditional too (the stage holds PB_CHUNK records;
                                                                     // its tail beyond n4 is never read): a predicated
                                                                     // store lets the compiler sink the load back under it
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      if (small) {
        const unsigned lo = max(rb, c0), hi = min(re, c1);
        for (unsigned d = lo; d < hi; ++d) acc.add(stage + (size_t)(d - c0) * DG_F4);
      }
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
  }
  for (unsigned long long todo = __ballot(cnt > COOP && cnt <= BWD_BIG); todo; todo &= todo - 1) {
    const int src = __builtin_ctzll(todo);
    const unsigned b0 = (unsigned)__shfl((int)d0, src), bn = (unsigned)__shfl((int)cnt, src);
    DupAcc w;
    w.zero();
    if (flags) {   // four flags per lane and round trip; a round without a record (big splats are dead over whole regions of
                   // the image) is skipped, the others fetch under the flags
      constexpr int U = 4;
      for (unsigned r = 0; r < bn; r += 64 * U) {
        bool lv[U];
        bool any = false;
#pragma unroll
        for (int j = 0; j < U; ++j) {
          const unsigned i = r + 64u * j + lane;
          lv[j] = live[(size_t)b0 + min(i, bn - 1u)] != 0 && i < bn;
          any |= lv[j];
        }
        if (__ballot(any) == 0ull) continue;
#pragma unroll
        for (int j = 0; j < U; ++j)
          if (lv[j]) w.add(dupgrad + (size_t)(b0 + r + 64u * j + lane) * DG_F4);
      }
    } else {
      for (unsigned i = lane; i < bn; i += 64) w.add(dupgrad + (size_t)(b0 + i) * DG_F4);
    }
#pragma unroll
    for (int i = 0; i < DG_F4; ++i) {
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) {
        w.q[i].x += __shfl_xor(w.q[i].x, d); w.q[i].y += __shfl_xor(w.q[i].y, d); w.q[i].z += __shfl_xor(w.q[i].z, d);
        if (DG_F4 == 3) w.q[i].w += __shfl_xor(w.q[i].w, d);
      }
    }
    if (lane == src) acc = w;
  }
  if (!valid) return;
  FrameParams f = load_frame(kf);
  f.sh_degree = DEG; f.sh_coeffs = K;
  GaussGrads out;
#pragma unroll
  for (int i = 0; i < 3; ++i) { out.means3D[i] = 0.f; out.means2D[i] = 0.f; out.scales[i] = 0.f; out.rgb[i] = 0.f; }
#pragma unroll
  for (int i = 0; i < 4; ++i) out.rot[i] = 0.f;
  out.opacity = 0.f;
  constexpr int ROW = K > 0 ? 3 * K : 1;
  float gshl[ROW];
#pragma unroll
  for (int i = 0; i < ROW; ++i) gshl[i] = 0.f;
  float gdir[3] = {0.f, 0.f, 0.f};
  if (vis) {
    // the records hold raw sums; op and the conic are applied once, to the total (raster_math.h: grad2d_from_sums)
    GradSums A;
    A.x = acc.get(0); A.y = acc.get(1); A.ax = acc.get(2); A.ay = acc.get(3);
    A.xx = acc.get(4); A.xy = acc.get(5); A.yy = acc.get(6); A.u = acc.get(7);
    A.r = acc.get(8); A.g = acc.get(9); A.b = acc.get(10); A.d = acc.get(11);
    float p[3], s[3];
    load3(means3D + 3 * (size_t)g, p);
    load3(scales + 3 * (size_t)g, s);
    const float4 qraw = *reinterpret_cast<const float4*>(rots + 4 * (size_t)g);
    float4 qv = qraw;
    float opacity;
    const float sraw[3] = {s[0], s[1], s[2]};
    unsigned long long o_bits = 0ull, f_bits = 0ull;   // RAW: the raw opacity / filter_3D words, loaded ONCE (as bits:
    if constexpr (!RAW) {                               // their types are a launch-uniform switch), used twice below
      opacity = static_cast<const float*>(opac_)[g];
    } else {
#define SFGS_ACT_LOAD(FT, OT) do { o_bits = raw_bits<OT>(opac_, g); f_bits = raw_bits<FT>(filt, g); } while (0)
      SFGS_ACT_DISPATCH(raw_mask, SFGS_ACT_LOAD);
#undef SFGS_ACT_LOAD
#define SFGS_ACT_FWD(FT, OT) act_outputs(act_terms<FT, OT>(sraw, from_bits<OT>(o_bits), from_bits<FT>(f_bits)), s, &opacity)
      SFGS_ACT_DISPATCH(raw_mask, SFGS_ACT_FWD);
#undef SFGS_ACT_FWD
      qv = act_rotation(qraw);
    }
    const float q[4] = {qv.x, qv.y, qv.z, qv.w};
    if constexpr (K > 0) {
      float shl[ROW];
      if constexpr (SPLIT) load_sh_rows<K>(shs, shs_rest, (size_t)g, shl);   // coefficient 0 in `shs`, the others in `shs_rest`
      else load_row<ROW>(shs + (size_t)ROW * g, shl);
      if constexpr (CM != 0) {
        float din[3];
        load3(sh_dirs + 3 * (size_t)g, din);
        // dirs_are_centers (SfgsGaussians.sh_centers): direction = normalize(p - centre), its gradient goes into means3D
        // like the in-kernel SH path's; otherwise the direction was an input and its gradient is one (g_sh_dirs)
        preprocess_backward_sums(f, p, s, q, opacity, shl, A, out, gshl, CM == 1, dirs_are_centers ? nullptr : din, gdir,
                                 dirs_are_centers ? din : nullptr);
      } else {
        preprocess_backward_sums(f, p, s, q, opacity, shl, A, out, gshl);
      }
    } else {
      preprocess_backward_sums(f, p, s, q, opacity, nullptr, A, out, gshl);
    }
    if constexpr (RAW) {   // ... and on through the activations (terms recomputed: cheaper than carrying 11 values)
      const float gs[3] = {out.scales[0], out.scales[1], out.scales[2]};
#define SFGS_ACT_BWD(FT, OT)                                                                                         \
  do {                                                                                                              \
    OT gro;                                                                                                         \
    act_backward(act_terms<FT, OT>(sraw, from_bits<OT>(o_bits), from_bits<FT>(f_bits)), gs, out.opacity, out.scales, \
                 &gro);                                                                                             \
    static_cast<OT*>(g_opac_)[g] = gro;                                                                             \
  } while (0)
      SFGS_ACT_DISPATCH(raw_mask, SFGS_ACT_BWD);
#undef SFGS_ACT_BWD
      const float4 gq = act_rotation_backward(qraw, make_float4(out.rot[0], out.rot[1], out.rot[2], out.rot[3]));
      out.rot[0] = gq.x; out.rot[1] = gq.y; out.rot[2] = gq.z; out.rot[3] = gq.w;
    }
  } else if constexpr (RAW) {   // not visible: every gradient is zero (the raw opacity's in its own dtype)
    if (raw_mask & 2) static_cast<double*>(g_opac_)[g] = 0.0; else static_cast<float*>(g_opac_)[g] = 0.f;
  }
  store3(g_means3D + 3 * (size_t)g, out.means3D[0], out.means3D[1], out.means3D[2]);
  store3(g_means2D + 3 * (size_t)g, out.means2D[0], out.means2D[1], out.means2D[2]);
  store3(g_scales + 3 * (size_t)g, out.scales[0], out.scales[1], out.scales[2]);
  *reinterpret_cast<float4*>(g_rots + 4 * (size_t)g) = make_float4(out.rot[0], out.rot[1], out.rot[2], out.rot[3]);
  if constexpr (!RAW) static_cast<float*>(g_opac_)[g] = out.opacity;
  if constexpr (K > 0) {
    if constexpr (SPLIT) store_sh_rows<K>(g_shs, g_shs_rest, (size_t)g, gshl);
    else store_row<ROW>(g_shs + (size_t)ROW * g, gshl);
    if constexpr (CM != 0) { if (!dirs_are_centers) store3(g_sh_dirs + 3 * (size_t)g, gdir[0], gdir[1], gdir[2]); }
  } else {
    store3(g_colors + 3 * (size_t)g, out.rgb[0], out.rgb[1], out.rgb[2]);
  }
}

}  // namespace sfgs

using namespace sfgs;

// option "prefill" = "always" | "never" (sfgs_set_option) forces / forbids the dead-entry prefill (default: decided per
// frame on the device). Both paths produce bit-identical gradients (tests/test_gpu_raster.py); the option exists so that
// the tests can run each. The kernel takes 0 = decide, 1 = always, 2 = never.
static int prefill_mode() { return option(OPT_PREFILL); }

extern "C" int sfgs_raster_backward(const SfgsFrame* frame, const SfgsGaussians* g, const int32_t* radii,
                                    const void* geom, const void* tiles, const void* bins, int64_t dup_capacity,
                                    int64_t coarse_capacity, int64_t num_duplicates, const void* image,
                                    const float* dL_dcolor, const float* dL_ddepth,
                                    const float* dL_dalpha, void* dupgrad, size_t dupgrad_sz,
                                    const SfgsGaussianGrads* grads, void* stream_) {
  SFGS_REQUIRE(frame && frame->struct_size == sizeof(SfgsFrame), SFGS_E_ARG, "SfgsFrame.struct_size mismatch");
  SFGS_REQUIRE(g && g->struct_size == sizeof(SfgsGaussians), SFGS_E_ARG, "SfgsGaussians.struct_size mismatch");
  SFGS_REQUIRE(grads && grads->struct_size == sizeof(SfgsGaussianGrads), SFGS_E_ARG,
               "SfgsGaussianGrads.struct_size mismatch");
  hipStream_t stream = (hipStream_t)stream_;
  const int N = g->count, W = frame->image_width, H = frame->image_height;
  if (N == 0) return SFGS_OK;
  SFGS_REQUIRE(radii && geom && tiles && image, SFGS_E_ARG, "forward state pointer is NULL");
  SFGS_REQUIRE(grads->means3D && grads->means2D && grads->scales && grads->rotations && grads->opacities, SFGS_E_ARG,
               "gradient output pointer is NULL");
  SFGS_REQUIRE((g->colors_precomp != nullptr) == (grads->colors_precomp != nullptr) &&
                   (g->shs != nullptr) == (grads->shs != nullptr) && (g->sh_dirs != nullptr) == (grads->sh_dirs != nullptr),
               SFGS_E_ARG, "colour gradient outputs must match the colour inputs");
  SFGS_REQUIRE(!(g->sh_dirs || g->sh_centers) || g->shs, SFGS_E_ARG, "sh_dirs / sh_centers without shs");
  SFGS_REQUIRE(!(g->sh_dirs && g->sh_centers), SFGS_E_ARG, "sh_dirs and sh_centers are alternatives");
  SFGS_REQUIRE((g->shs_rest != nullptr) == (grads->shs_rest != nullptr) &&
                   (!g->shs_rest || (g->shs && frame->sh_coeffs > 1 && g->shs_channel_major == 0)),
               SFGS_E_ARG, "shs_rest (split SH storage): gradient output must match, needs shs, sh_coeffs > 1, coefficient-major");
  SFGS_REQUIRE(g->filter_3D ? (g->raw_f64_mask & ~3) == 0 : g->raw_f64_mask == 0, SFGS_E_ARG,
               "raw_f64_mask %d: bit 0 = filter_3D is float64, bit 1 = raw opacities are float64; 0 without filter_3D",
               g->raw_f64_mask);
  SFGS_REQUIRE(dup_capacity >= 0 && num_duplicates >= 0 && num_duplicates <= dup_capacity, SFGS_E_ARG,
               "bad dup_capacity / num_duplicates");
  // duplicate indices come from DUP_POOLS ranges of [0, dup_capacity): the record array spans the whole index space
  SFGS_REQUIRE(num_duplicates == 0 || dupgrad_sz >= dupgrad_bytes(dup_capacity), SFGS_E_CAPACITY,
               "dupgrad blob: %zu bytes given, %zu needed", dupgrad_sz, dupgrad_bytes(dup_capacity));
  SFGS_REQUIRE(num_duplicates == 0 || (bins && dupgrad), SFGS_E_ARG, "bins / dupgrad is NULL");
  const TilesView tv = tiles_view(const_cast<void*>(tiles), W, H, N, nullptr);
  const GeomView gv = geom_view(const_cast<void*>(geom), N);
  const BinsView bv = bins_view(const_cast<void*>(bins), dup_capacity, coarse_bins(W, H), coarse_capacity);
  const ImageView iv = image_view(const_cast<void*>(image), W, H, dup_capacity);
  const KFrame kf = make_kframe(frame);
  const int TX8 = (W + TILE_BIN - 1) / TILE_BIN, TY8 = (H + TILE_BIN - 1) / TILE_BIN;
  constexpr int BE = composite_block_edge<BWG_WAVES>();
  const int SX = (TX8 + BE - 1) / BE, SY = (TY8 + BE - 1) / BE;
  // the live flags behind the records (sfgs_internal.h: dupgrad_bytes). Without the prefill kernel its decision word keeps
  // the plan's zero -- or an EARLIER backward's decision over the same forward state: the readers then get no flag pointer
  // and composite_bwd writes the zero records itself
  const bool no_prefill = (frame->launch_hints & SFGS_HINT_NO_PREFILL) != 0;
  uint8_t* live = (uint8_t*)dupgrad + dupgrad_flags_offset(dup_capacity);
  const uint8_t* live_r = no_prefill ? nullptr : live;
  const size_t zero_f4 = dupgrad_zero_offset(dup_capacity) / 16;
  { ProfScope ps_(KID_COMPOSITE_BWD, stream);
    if (!no_prefill)
    hipLaunchKernelGGL(dupgrad_prefill_kernel, dim3(512), dim3(256), 0, stream, TX8 * TY8, iv.tile_dead,
                       (unsigned long long)num_duplicates, prefill_mode(), (uint4*)live, (uint4*)((char*)dupgrad + dupgrad_zero_offset(dup_capacity)), tv.hdr,
                       (unsigned long long*)frame->feedback, (const unsigned long long*)tv.dup_pool,
                       (unsigned long long)dup_capacity, dup_pools_used(pre_blocks(N)));
    launch_composite_bwd(composite_grid(SX, SY, BE * BE / BWG_WAVES), stream, kf, TX8, TY8, SX, SY, tv.tile_range, bv.sorted_id,
                         bv.sorted_dup, gv.rec, iv.n_contrib, iv.final_T, iv.dacc, dL_dcolor, dL_ddepth, dL_dalpha, iv.hitmask,
                         iv.tile_kmax, (float4*)dupgrad, live, tv.hdr, no_prefill ? 1 : 0,
                         tv.tile_order + (size_t)8 * order_slots(W, H), (unsigned)order_slots(W, H));
  }
  SFGS_POST_LAUNCH("composite_bwd", stream, frame->debug);
  const int NB = (int)pre_blocks(N);
  { ProfScope ps_(KID_PREPROCESS_BWD, stream);
    if (!(frame->launch_hints & SFGS_HINT_NO_BIG_CHUNKS))   // the caller read num_big_chunks == 0 from this frame's plan
    hipLaunchKernelGGL(dupgrad_reduce_kernel, dim3(1024), dim3(256), 0, stream, tv.hdr, bv.big_chunks,
                       (unsigned)big_chunk_capacity(dup_capacity), gv.dup, (float4*)dupgrad, live_r);
#define SFGS_LAUNCH_PBWD_(K, D, RAW, CM)                                                                               \
  hipLaunchKernelGGL((preprocess_bwd_kernel<K, D, RAW, CM>), dim3(NB), dim3(PRE_BLOCK), 0, stream, kf, N, g->means3D,  \
                     g->scales, g->rotations, (const void*)g->opacities, g->filter_3D, (int)g->raw_f64_mask, g->shs,   \
                     g->shs_rest, g->sh_dirs ? g->sh_dirs : g->sh_centers, g->sh_centers ? 1 : 0, radii, gv.dup,       \
                     (const float4*)dupgrad, live_r, zero_f4, tv.hdr, grads->means3D, grads->means2D,                  \
                     grads->scales, grads->rotations, (void*)grads->opacities, grads->colors_precomp, grads->shs,      \
                     grads->shs_rest, grads->sh_dirs)
#define SFGS_LAUNCH_PBWD(K, D)                                                                                         \
  do {                                                                                                                 \
    if constexpr ((K) > 1) {   /* split SH storage (shs_rest): instantiations of their own (CMX = CM + 3) */            \
      if (g->shs_rest && (g->sh_dirs || g->sh_centers)) { if (g->filter_3D) SFGS_LAUNCH_PBWD_(K, D, true, 5); else SFGS_LAUNCH_PBWD_(K, D, false, 5); break; } \
      if (g->shs_rest) { if (g->filter_3D) SFGS_LAUNCH_PBWD_(K, D, true, 3); else SFGS_LAUNCH_PBWD_(K, D, false, 3); break; } \
    }                                                                                                                  \
    if constexpr ((K) > 0) {                                                                                           \
      if ((g->sh_dirs || g->sh_centers) && g->shs_channel_major) { if (g->filter_3D) SFGS_LAUNCH_PBWD_(K, D, true, 1); else SFGS_LAUNCH_PBWD_(K, D, false, 1); break; } \
      if (g->sh_dirs || g->sh_centers) { if (g->filter_3D) SFGS_LAUNCH_PBWD_(K, D, true, 2); else SFGS_LAUNCH_PBWD_(K, D, false, 2); break; } \
    }                                                                                                                  \
    if (g->filter_3D) SFGS_LAUNCH_PBWD_(K, D, true, 0); else SFGS_LAUNCH_PBWD_(K, D, false, 0);                        \
  } while (0)
    SFGS_DISPATCH_SH(g->shs ? frame->sh_coeffs : 0, frame->sh_degree, SFGS_LAUNCH_PBWD);
#undef SFGS_LAUNCH_PBWD
#undef SFGS_LAUNCH_PBWD_
  }
  SFGS_POST_LAUNCH("preprocess_bwd", stream, frame->debug);
  return SFGS_OK;
}

namespace sfgs { int scratch_layout(int32_t N, int32_t W, int32_t H, int64_t D, int64_t ccap, bool with_image, SfgsScratchLayout* out); }

extern "C" int sfgs_raster_backward_scratch(const SfgsFrame* frame, const SfgsGaussians* g, const int32_t* radii,
                                            const void* scratch, size_t scratch_bytes, int64_t dup_capacity,
                                            int64_t coarse_capacity, int64_t num_duplicates, const float* dL_dcolor,
                                            const float* dL_ddepth, const float* dL_dalpha, void* dupgrad,
                                            size_t dupgrad_sz, const SfgsGaussianGrads* grads, void* stream_) {
  SFGS_REQUIRE(frame && frame->struct_size == sizeof(SfgsFrame), SFGS_E_ARG, "SfgsFrame.struct_size mismatch");
  SFGS_REQUIRE(g && g->struct_size == sizeof(SfgsGaussians), SFGS_E_ARG, "SfgsGaussians.struct_size mismatch");
  SFGS_REQUIRE(scratch != nullptr, SFGS_E_ARG, "scratch is NULL");
  SfgsScratchLayout lay;
  if (int rc = scratch_layout(g->count, frame->image_width, frame->image_height, dup_capacity, coarse_capacity, true, &lay))
    return rc;
  SFGS_REQUIRE(scratch_bytes >= lay.total_bytes, SFGS_E_CAPACITY, "scratch: %zu bytes given, %zu needed", scratch_bytes,
               lay.total_bytes);
  const char* base = (const char*)scratch;
  return sfgs_raster_backward(frame, g, radii, base + lay.geom_offset, base + lay.tiles_offset, base + lay.bins_offset,
                              dup_capacity, coarse_capacity, num_duplicates, base + lay.image_offset, dL_dcolor, dL_ddepth,
                              dL_dalpha, dupgrad, dupgrad_sz, grads, stream_);
}
