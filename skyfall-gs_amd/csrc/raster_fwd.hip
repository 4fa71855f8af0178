// raster_fwd.hip -- forward of the Gaussian-splat rasterizer for gfx950 (MI355X), hand-written HIP.
//
// Replaces diff_gauss.GaussianRasterizer.__call__ (reference: gaussian_renderer/__init__.py:132-140)
// behind the C ABI of include/sfgs.h. Kernel chain (all wave64, no MFMA: the path is gather/sort/VALU
// bound):
//
//   plan:    subpix_bound -> preprocess (per Gaussian: project, radii, 48-B record; opacity-aware COARSE binning:
//            one 16-B item per (Gaussian, 32x32-px coarse bin) with the mask of the 8x8 tiles it can contribute to)
//            -> plan_scan (counters for the host, published straight into pinned host memory)
//   render:  fine_bin (per coarse bin: LDS-ranked expansion into per-tile item segments)
//            -> sort_tiles (per tile: normalised bitonic network in registers up to 2 048 entries, LDS and an
//               LDS/global hybrid beyond)
//            -> composite (one wave per 8x8 tile, front to back, records staged through LDS)
//
// Compile with -ffp-contract=off: integer outputs depend on exact float32 sequences (raster_math.h).
#include <stdlib.h>
#include <string.h>

#include <algorithm>

#include <type_traits>

#include "act_math.h"
#include "sfgs_internal.h"

namespace sfgs {

// Variants built, measured and not kept (A/B files; the code is in the history, not in this source):
//   preprocess: records leave through LDS as lane-contiguous stores         profiles/r4_preprocess_rec_transpose_ab_not_kept.txt
//   composite_fwd: records gathered in piece order (the backward's trick)   profiles/r4_fwd_gather3_ab_not_kept.txt
//   composite_fwd: exact ellipse-vs-pixel-row strip test                     profiles/r4_fwd_strip_exact_ab_not_kept.txt
//   the plan's head cleared by hipMemsetAsync instead of zero_head_kernel    profiles/r4_plan_memset_ab.txt
constexpr int REG_SORT_SMALL = 512;  // lists up to here: sort_tiles_reg_kernel (<= 8 keys per lane, 8 waves per SIMD)
constexpr int REG_SORT_MAX = 1024;   // lists up to here: register network too (16 keys per lane), separate kernel

// ------------------------------------------------------------------------------------------------
// max |subpixel_offset| -> header (float bits; non-negative floats order like unsigned ints)
__global__ void __launch_bounds__(256) subpix_bound_kernel(const float* __restrict__ subpix, int64_t n,
                                                           unsigned long long* __restrict__ hdr) {
  // float4 loads, a few hundred workgroups, ONE atomic per workgroup and none at all for an all-zero tensor (what the
  // reference's render() passes when ray jitter is off): atomics to one address serialise at ~12 ns each
  __shared__ float s_m[4];
  float m = 0.f;
  const int64_t n4 = (reinterpret_cast<uintptr_t>(subpix) & 15) ? 0 : (n >> 2);  // unaligned view: scalar tail only
  const float4* __restrict__ v4 = reinterpret_cast<const float4*>(subpix);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    const float4 v = v4[i];
    m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
  }
  for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    m = fmaxf(m, fabsf(subpix[i]));
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d));
  if (lane_id() == 0) s_m[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmaxf(fmaxf(s_m[0], s_m[1]), fmaxf(s_m[2], s_m[3]));
    if (m > 0.f) atomicMax((unsigned int*)&hdr[HDR_SUBPIX_BOUND], __float_as_uint(m));
  }
}

// 16-bit hit mask of the 4x4 tiles of coarse bin (cbx, cby) that lie inside the walk range `br`
// (bit = 4 * local_y + local_x).
__device__ __forceinline__ unsigned coarse_hits(const SplatRec& r, float thr, const BinRange& br, int cbx, int cby,
                                                int W, int H, float bound) {
  unsigned m = 0;
  const int tx0 = imax(br.x0, cbx * COARSE), tx1 = imin(br.x1, cbx * COARSE + COARSE);
  const int ty0 = imax(br.y0, cby * COARSE), ty1 = imin(br.y1, cby * COARSE + COARSE);
  for (int ty = ty0; ty < ty1; ++ty)
    for (int tx = tx0; tx < tx1; ++tx)
      if (bin_test(r, thr, tx, ty, W, H, bound)) m |= 1u << ((ty - cby * COARSE) * COARSE + (tx - cbx * COARSE));
  return m;
}

// small walks handed to the wave (preprocess_kernel): at most this many tiles per Gaussian (so a wave has at most 2 048
// tasks), the owner's test inputs, its mask words and the task -> owner map, 7 KB of LDS per wave
constexpr int WALK_COOP_MAX = 32;
struct WalkLds {
  float4 rec[64][4];
  uint4 mask[64];
  unsigned char owner[64 * WALK_COOP_MAX];
};
constexpr int BIG_WALK = 6;  // coarse bins above which a splat's walk is done by a whole wave (big_walk_kernel)
constexpr int BIG_WALK_BLOCKS = 1024;   // persistent grid of big_walk_kernel: 4 096 waves = 4 per SIMD, what its 120 VGPRs allow (round 4:
                                        // with 1 per SIMD nothing hid the tile tests' dependent chains -- near-camera regime); the
                                        // kernel is not launched at all under the NO_HUGE_SPLATS hint
constexpr int BIG_WALK_CACHE = 2048;    // coarse-bin masks a wave keeps in LDS between its count and emit passes (16-bit each)
static_assert(BIG_WALK <= 8, "the per-lane walk keeps one 16-bit mask per coarse bin in two 64-bit registers");

struct WalkArgs { SplatRec r; BinRange br; float thr; int cx0, cx1, cy0, cy1; };

// lane L's walk parameters, broadcast to the wave
__device__ __forceinline__ WalkArgs broadcast_walk(const SplatRec& r, const BinRange& br, float thr, int cx0, int cx1,
                                                  int cy0, int cy1, int L) {
  WalkArgs a;
  a.r.mx = __shfl(r.mx, L); a.r.my = __shfl(r.my, L); a.r.qa = __shfl(r.qa, L); a.r.qb = __shfl(r.qb, L);
  a.r.qc = __shfl(r.qc, L); a.r.op = __shfl(r.op, L); a.r.depth = 0.f; a.r.r = a.r.g = a.r.b = 0.f;
  a.r.ex = __shfl(r.ex, L); a.r.ey = __shfl(r.ey, L);
  a.br.x0 = __shfl(br.x0, L); a.br.x1 = __shfl(br.x1, L); a.br.y0 = __shfl(br.y0, L); a.br.y1 = __shfl(br.y1, L);
  a.thr = __shfl(thr, L);
  a.cx0 = __shfl(cx0, L); a.cx1 = __shfl(cx1, L); a.cy0 = __shfl(cy0, L); a.cy1 = __shfl(cy1, L);
  return a;
}

// Cooperative walk, lane = TILE: lanes 16q..16q+15 test the 16 tiles of coarse bin number 4*it + q of the walk
// (row-major over [cx0,cx1) x [cy0,cy1)); the ballot's 16-bit slice q is that bin's mask, identical to coarse_hits().
// A mid-size splat (6..63 coarse bins) keeps all 64 lanes busy this way; one lane per coarse bin would use 6..63.
__device__ __forceinline__ unsigned long long walk_ballot4(const WalkArgs& a, int it, int lane, int W, int H,
                                                           float bound, int* cb_index, int CX) {
  const int nx = a.cx1 - a.cx0, ncb = nx * (a.cy1 - a.cy0);
  const int i = it * 4 + (lane >> 4), t = lane & 15;
  bool hit = false;
  int cb = 0;
  if (i < ncb) {
    const int cx = a.cx0 + i % nx, cy = a.cy0 + i / nx;
    const int tx = cx * COARSE + (t & (COARSE - 1)), ty = cy * COARSE + (t / COARSE);
    cb = cy * CX + cx;
    if (tx >= a.br.x0 && tx < a.br.x1 && ty >= a.br.y0 && ty < a.br.y1) hit = bin_test(a.r, a.thr, tx, ty, W, H, bound);
  }
  *cb_index = cb;
  return __ballot(hit);
}

__device__ __forceinline__ unsigned wave_sum_u32(unsigned v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += (unsigned)__shfl_xor((int)v, d);
  return v;
}

// ------------------------------------------------------------------------------------------------
// K1: one thread per Gaussian: project, write radii + the compositing record, and bin COARSELY.
// The thread walks the coarse bins (32x32 px) its splat can reach; for each it computes the 16-bit mask of
// the 8x8 tiles the splat can contribute to (opacity-aware test, raster_math.h). A non-empty mask becomes
// one 16-byte coarse item (id, depth, first duplicate index, mask) appended to the bin's slab with ONE
// returning device atomic. The block reserves its duplicate indices (one per set mask bit, in walk order)
// with one atomic per block.
// RAW: `scales`, `rots`, `opac` are the model's raw parameters and `filt` its 3D filter (SfgsGaussians raw-parameter
// mode): the activations of act_math.h run here instead of in a pre-pass kernel that writes N-sized intermediates.
// CM != 0 (SfgsGaussians.sh_dirs): the view direction of a Gaussian is read from `sh_dirs` -- the eval_sh + 0.5 +
// clamp_min of render()'s Python colour paths folded in (sh_to_rgb); CM == 1: `shs` is channel-major [N,3,K] (eval_sh's
// layout), CM == 2: coefficient-major [N,K,3].
template <int K, int DEG, bool RAW, int CM>  // K = SH coefficients stored per Gaussian (0: colors_precomp), DEG = active degree
__global__ void __launch_bounds__(PRE_BLOCK)
preprocess_kernel(KFrame kf, int N, const float* __restrict__ means3D, const float* __restrict__ scales,
                  const float* __restrict__ rots, const void* __restrict__ opac_, const void* __restrict__ filt,
                  int raw_mask,
                  const float* __restrict__ colors, const float* __restrict__ shs, const float* __restrict__ shs_rest,
                  const float* __restrict__ sh_dirs, int dirs_are_centers, int* __restrict__ radii,
                  float4* __restrict__ rec_out, uint2* __restrict__ dup_out, uint32_t* __restrict__ coarse_count,
                  uint4* __restrict__ slabs, unsigned coarse_capacity, unsigned long long dup_capacity,
                  uint32_t* __restrict__ block_nvis, unsigned long long* __restrict__ block_dref,
                  uint4* __restrict__ big_list, unsigned long long* __restrict__ hdr,
                  unsigned long long* __restrict__ dup_pool, uint4* __restrict__ pairs,
                  uint32_t* __restrict__ block_items) {
  __shared__ unsigned s_red[PRE_BLOCK / 64 + 1];
  __shared__ unsigned long long s_base;
  const FrameParams f = load_frame(kf);
  const float bound = __uint_as_float((unsigned)hdr[HDR_SUBPIX_BOUND]);
  const int g = blockIdx.x * PRE_BLOCK + threadIdx.x;
  const int CX = (((f.W + TILE_BIN - 1) / TILE_BIN) + COARSE - 1) / COARSE;
  unsigned n_dup = 0, vis = 0, dref = 0, depth_bits = 0;
  unsigned long long mask_lo = 0ull, mask_hi = 0ull;  // 16-bit tile masks of the (<= BIG_WALK) coarse bins of the walk
  unsigned walk_tiles = 0;           // tiles of a walk of at most BIG_WALK coarse bins (<= 96): walked after the per-thread part
  bool big = false;                  // walk handled cooperatively by the wave (BIG_WALK < coarse bins < 64)
  bool huge = false;                 // >= 64 coarse bins: walked by big_walk_kernel
  const int lane = threadIdx.x & 63;
  SplatRec r;
  BinRange br;
  float thr = 0.f;
  int cx0 = 0, cx1 = 0, cy0 = 0, cy1 = 0;
  br.x0 = br.x1 = br.y0 = br.y1 = 0;
  if (g < N) {
    // every input of the Gaussian is requested before the first one is used (the empty asm below consumes one word of
    // each load, which keeps the compiler from sinking the loads behind the near-plane test inside project_gaussian and
    // behind `if (pr.visible)`: three dependent round trips to memory per thread otherwise)
    float p[3], s[3];
    load3(means3D + 3 * (size_t)g, p);
    load3(scales + 3 * (size_t)g, s);
    float4 qv = *reinterpret_cast<const float4*>(rots + 4 * (size_t)g);
    float opacity_in = 0.f;
    if constexpr (!RAW) opacity_in = static_cast<const float*>(opac_)[g];
    float rgb_in[3] = {0.f, 0.f, 0.f};
    if constexpr (K == 0) load3(colors + 3 * (size_t)g, rgb_in);
    // RAW: the raw opacity and filter_3D words belong to the same round trip (as bits: their types are a launch-uniform
    // switch, and a load inside the switch BELOW the asm would be a second, dependent trip to memory)
    unsigned long long o_bits = 0ull, f_bits = 0ull;
    if constexpr (RAW) {
#define SFGS_ACT_LOAD(FT, OT) do { o_bits = raw_bits<OT>(opac_, g); f_bits = raw_bits<FT>(filt, g); } while (0)
      SFGS_ACT_DISPATCH(raw_mask, SFGS_ACT_LOAD);
#undef SFGS_ACT_LOAD
    }
    asm volatile("" :: "v"(p[2]), "v"(s[0]), "v"(qv.x), "v"(opacity_in), "v"(rgb_in[0]), "v"((unsigned)o_bits),
                 "v"((unsigned)f_bits));   // all of them in registers here
    if constexpr (RAW) {   // raw parameters -> what render() would have passed (the mask is uniform over the launch)
#define SFGS_ACT_FWD(FT, OT) act_outputs(act_terms<FT, OT>(s, from_bits<OT>(o_bits), from_bits<FT>(f_bits)), s, &opacity_in)
      SFGS_ACT_DISPATCH(raw_mask, SFGS_ACT_FWD);
#undef SFGS_ACT_FWD
      qv = act_rotation(qv);
    }
    float q[4] = {qv.x, qv.y, qv.z, qv.w};
    const Projected pr = project_gaussian(f, p, s, q);
    radii[g] = pr.radius;
    if (pr.visible) {
      vis = 1;
      dref = (unsigned)((pr.rmaxx - pr.rminx) * (pr.rmaxy - pr.rminy));
      float rgb[3];
      if constexpr (K == 0) {
        rgb[0] = rgb_in[0]; rgb[1] = rgb_in[1]; rgb[2] = rgb_in[2];
      } else {
        float shl[3 * K];
        // coefficient-major rows: K 12-byte loads from one array or, split storage (SfgsGaussians.shs_rest), from two --
        // the same instructions either way (load_sh_rows); the channel-major form [N,3,K] reads its 3 K floats as they lie
        if constexpr (K > 1 && CM != 1) load_sh_rows<K>(shs, shs_rest, (size_t)g, shl);
        else load_row<3 * K>(shs + 3 * (size_t)K * g, shl);
        unsigned cm; float dir[3], len;
        if constexpr (CM != 0) {
          float din[3];
          load3(sh_dirs + 3 * (size_t)g, din);
          // dirs_are_centers (SfgsGaussians.sh_centers): `sh_dirs` holds per-Gaussian centres c and the direction is
          // normalize(p - c), computed here like the in-kernel SH path's normalize(p - campos); wave-uniform
          sh_to_rgb(DEG, shl, p, dirs_are_centers ? din : f.campos, rgb, &cm, dir, &len, CM == 1 ? 1 : 3, CM == 1 ? K : 1,
                    dirs_are_centers ? nullptr : din);
        } else {
          sh_to_rgb(DEG, shl, p, f.campos, rgb, &cm, dir, &len);
        }
      }
      r = make_record(pr, opacity_in, rgb);
      rec_out[REC_F4 * (size_t)g + 0] = make_float4(r.mx, r.my, r.qa, r.qb);
      // (r, g) and (b, depth) sit in aligned pairs: the compositing loops fetch them with one 16-byte and one 8-byte
      // LDS read into the register pairs their packed multiply-adds take
      rec_out[REC_F4 * (size_t)g + 1] = make_float4(r.qc, r.op, r.r, r.g);
      rec_out[REC_F4 * (size_t)g + 2] = make_float4(r.b, r.depth, r.ex, r.ey);
      depth_bits = __float_as_uint(r.depth);
      br = bin_range(r, f.W, f.H, pr.rminx, pr.rminy, pr.rmaxx, pr.rmaxy, bound);
      br.y0 = imax(br.y0, kf.band0); br.y1 = imax(br.y0, imin(br.y1, kf.band1));  // band rendering
      thr = alpha_threshold_log2(r.op);
      if (br.x1 > br.x0 && br.y1 > br.y0) {
        cx0 = br.x0 / COARSE; cx1 = (br.x1 - 1) / COARSE + 1;
        cy0 = br.y0 / COARSE; cy1 = (br.y1 - 1) / COARSE + 1;
        const int ncb_ = (cx1 - cx0) * (cy1 - cy0);
        huge = ncb_ >= 64;
        big = ncb_ > BIG_WALK && !huge;
        if (ncb_ <= BIG_WALK) {
          walk_tiles = (unsigned)(br.y1 - br.y0) * (unsigned)(br.x1 - br.x0);   // walked below, by the wave or by this thread
        }
      }
    }
  }
  // Small walks, balanced over the wave. One thread walking ITS Gaussian's tiles makes the wave run max over its 64
  // lanes of the tile counts -- 11.4 steps on the headline scene for 4.4 tiles per Gaussian (tools/workmodel) -- in a kernel
  // that is bound by VALU issue with this loop as half of its instructions. Here lane = one (Gaussian, tile) task: the
  // owners publish what the test needs (48 bytes) and their lane number in the task slots they own, every lane takes task
  // 64 p + lane in pass p (4.9 passes), hits are ORed into the owner's mask words with LDS atomics. Same tests on the
  // same tiles: masks and duplicate counts are bit-identical.
  // A walk of more than WALK_COOP_MAX tiles (33 .. 96: its tasks would not fit the wave's task map) is not left to its
  // owner thread -- until round 6 such a lane sent its WHOLE wave back to the thread-per-Gaussian loop, up to 96 serial tile
  // tests with 63 lanes waiting: most waves of a low-elevation or UHD frame (preprocess 0.078 -> 0.276 ms on the pitched
  // camera) -- but walked by the whole wave, lane = tile, as the mid-size splats below are: its at most 6 coarse bins are
  // two ballots, whose 16-bit slices ARE the owner's mask words (walk_ballot4: same tests, same bit positions).
  const bool wide = walk_tiles > (unsigned)WALK_COOP_MAX;
  const unsigned long long wide_lanes = __ballot(wide);
  if (wide) walk_tiles = 0;   // not a task of the cooperative map
  for (unsigned long long wl_ = wide_lanes; wl_; wl_ &= wl_ - 1) {
    const int L = __builtin_ctzll(wl_);
    const WalkArgs a = broadcast_walk(r, br, thr, cx0, cx1, cy0, cy1, L);
    int cb_;
    const unsigned long long m0 = walk_ballot4(a, 0, lane, f.W, f.H, bound, &cb_, CX);
    const unsigned long long m1 = (a.cx1 - a.cx0) * (a.cy1 - a.cy0) > 4 ? walk_ballot4(a, 1, lane, f.W, f.H, bound, &cb_, CX) : 0ull;
    if (lane == L) { mask_lo = m0; mask_hi = m1; n_dup = (unsigned)__popcll(m0) + (unsigned)__popcll(m1); }
  }
  {
    __shared__ WalkLds s_walk[PRE_BLOCK / 64];
    const unsigned incl = wave_incl_scan_u32(walk_tiles);
    const unsigned total_t = (unsigned)__builtin_amdgcn_readlane((int)incl, 63);
    if (total_t) {   // wave-uniform
      WalkLds& wl = s_walk[threadIdx.x >> 6];
      const unsigned excl = incl - walk_tiles;
      wl.mask[lane] = make_uint4(0u, 0u, 0u, 0u);
      if (walk_tiles) {
        const unsigned w_t = (unsigned)(br.x1 - br.x0);
        wl.rec[lane][0] = make_float4(r.mx, r.my, r.qa, r.qb);
        wl.rec[lane][1] = make_float4(r.qc, thr, (float)(br.x0 * TILE_BIN), (float)(br.y0 * TILE_BIN));
        // k / w_t for k < 32 as (k * M) >> 16 with M = ceil(2^16 / w_t): exact while k (M - 2^16 / w_t) < 2^16 / w_t
        const unsigned M = (65536u + w_t - 1u) / w_t;
        wl.rec[lane][2] = make_float4(__uint_as_float(w_t | (M << 8)), __uint_as_float((unsigned)br.x0 | ((unsigned)br.y0 << 16)),
                                      __uint_as_float((unsigned)cx0 | ((unsigned)cy0 << 16)),
                                      __uint_as_float((unsigned)(cx1 - cx0) | (excl << 8)));
        wl.rec[lane][3] = make_float4(-0.5f / r.qc, -0.5f / r.qa, 0.f, 0.f);   // the test's two divisions, once per Gaussian
        for (unsigned k = 0; k < walk_tiles; ++k) wl.owner[excl + k] = (unsigned char)lane;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const float wm1 = (float)(f.W - 1), hm1 = (float)(f.H - 1);
      for (unsigned t0 = 0; t0 < total_t; t0 += 64) {
        const unsigned t = t0 + (unsigned)lane;
        if (t < total_t) {
          const unsigned o = wl.owner[t];
          const float4 a0 = wl.rec[o][0], a1 = wl.rec[o][1], a2 = wl.rec[o][2];
          const float2 a3 = *reinterpret_cast<const float2*>(&wl.rec[o][3]);
          const unsigned wM = __float_as_uint(a2.x), xy = __float_as_uint(a2.y), cxy = __float_as_uint(a2.z),
                         ne = __float_as_uint(a2.w);
          const unsigned w_t = wM & 0xffu, k = t - (ne >> 8);
          const unsigned row = __umul24(k, wM >> 8) >> 16, col = k - __umul24(row, w_t);   // k < 32, M <= 2^16
          SplatRec q;
          q.mx = a0.x; q.my = a0.y; q.qa = a0.z; q.qb = a0.w; q.qc = a1.x;
          if (bin_test_at_h(q, a1.y, a1.z + (float)(col * TILE_BIN), a1.w + (float)(row * TILE_BIN), wm1, hm1, bound, a3.x, a3.y)) {
            const unsigned tx = (xy & 0xffffu) + col, ty = (xy >> 16) + row;
            const unsigned slot = ((ty >> 2) - (cxy >> 16)) * (ne & 0xffu) + ((tx >> 2) - (cxy & 0xffffu));
            const unsigned pos = slot * 16u + (ty & 3u) * 4u + (tx & 3u);
            atomicOr(&wl.mask[o].x + (pos >> 5), 1u << (pos & 31u));
          }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      if (walk_tiles) {
        const uint4 m = wl.mask[lane];
        mask_lo = (unsigned long long)m.x | ((unsigned long long)m.y << 32);
        mask_hi = (unsigned long long)m.z;
        n_dup = (unsigned)__popcll(mask_lo) + (unsigned)__popc(m.z);
      }
      __builtin_amdgcn_wave_barrier();   // the next user of this wave's LDS block is the next launch
    }
  }
  // Mid-size splats (more than BIG_WALK, fewer than 64 coarse bins) are walked by the whole wave, lane = tile, instead
  // of serially by their owner thread.
  const unsigned long long big_lanes = __ballot(big);
  for (unsigned long long bl = big_lanes; bl; bl &= bl - 1) {
    const int L = __builtin_ctzll(bl);
    const WalkArgs a = broadcast_walk(r, br, thr, cx0, cx1, cy0, cy1, L);
    const int ncb = (a.cx1 - a.cx0) * (a.cy1 - a.cy0);
    unsigned total = 0;
    for (int it = 0; it * 4 < ncb; ++it) {
      int cb;
      total += (unsigned)__popcll(walk_ballot4(a, it, lane, f.W, f.H, bound, &cb, CX));
    }
    if (lane == L) n_dup = total;
  }
  // Huge splats (64 coarse bins and more, up to the whole screen) are not walked here: they go to the work list of
  // big_walk_kernel, which spreads them over the whole chip one wave per splat (walked by the wave that owns them,
  // 2 000 screen-filling splats kept 8 waves busy for 10 ms while the other 4 000 wave slots idled). That kernel also
  // reserves their duplicate indices and writes their dup_out entry.
  const unsigned long long huge_lanes = __ballot(huge);
  if (huge_lanes) {
    unsigned long long slot0 = 0ull;
    if (lane == 0) slot0 = atomicAdd(&hdr[HDR_BIG_COUNT], (unsigned long long)__popcll(huge_lanes));
    const unsigned s_lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)slot0);
    if (huge) {
      const unsigned rk = __builtin_amdgcn_mbcnt_hi((unsigned)(huge_lanes >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)huge_lanes, 0u));
      big_list[s_lo + rk] = make_uint4((unsigned)g, (unsigned)br.x0 | ((unsigned)br.x1 << 16),
                                       (unsigned)br.y0 | ((unsigned)br.y1 << 16), 0u);
    }
  }
  // reserve duplicate indices: block scan + one returning atomic per block. Two-pass binning (`pairs`): the same scan
  // ranks the thread's coarse ITEMS inside the workgroup's stretch of the pair list -- packed into one word: a
  // workgroup has < 2^20 duplicates (256 threads x at most 63 coarse bins x 16 tiles) and at most 1 536 own items
  unsigned n_items = 0;
  if (pairs && n_dup && !big && !huge) {
#pragma unroll
    for (int k = 0; k < BIG_WALK; ++k)
      n_items += ((k < 4 ? mask_lo >> (16 * k) : mask_hi >> (16 * (k - 4))) & 0xffffull) != 0ull;
  }
  static_assert(PAIRS_PER_BLOCK == PRE_BLOCK * BIG_WALK, "a thread emits at most BIG_WALK items itself");
  unsigned total;
  const unsigned ex_packed = block_excl_scan_u32<PRE_BLOCK>(n_dup | (n_items << 20), &total, s_red);
  const unsigned ex = ex_packed & 0xfffffu, ex_items = ex_packed >> 20, total_items = total >> 20;
  total &= 0xfffffu;
  __shared__ int s_fits;
  if (threadIdx.x == 0) {
    bool ok = true;
    s_base = total ? dup_alloc(dup_pool, dup_pools_used(gridDim.x), blockIdx.x, total, dup_capacity, &ok) : 0ull;
    s_fits = ok;
  }
  __syncthreads();
  const unsigned long long base = s_base;
  const bool fits = s_fits != 0;
  if (!fits && threadIdx.x == 0) hdr[HDR_OVERFLOW] = 1ull;
  if (g < N && !huge) dup_out[g] = make_uint2((unsigned)(base + ex), n_dup);
  {
    // Emission, one coarse bin of the walk per iteration and lane. Lanes that append to the SAME bin in the same
    // iteration -- the rule rather than the exception when the Gaussians are stored in a spatially coherent order
    // (a loaded point cloud, clones next to their parents): 64 returning atomics on one counter serialise, the kernel
    // was 2.5x slower on a Z-curve-ordered scene than on a shuffled one -- are merged per RUN of consecutive lanes: the
    // run's first lane reserves the run's ranks with one atomic. On a shuffled scene every run has length 1 (same
    // number of atomics as before, a few lane-mask instructions more).
    const bool emit = fits && n_dup && !big;
    const int nb = emit ? (cx1 - cx0) * (cy1 - cy0) : 0;
    unsigned dup = (unsigned)(base + ex);
    int cx = cx0, cy = cy0;
    if (pairs) {
      // TWO-PASS BINNING (round 3): the items go, with plain stores and the coarse bin in the upper half of the mask
      // word, into this workgroup's stretch of the pair list; bin_scatter_kernel then ranks them per bin with LDS
      // atomics and reserves slab ranges with ONE device atomic per (scatter workgroup, bin) -- 0.5 M instead of 2.7 M
      // returning device atomics on the headline scene, which is what this kernel was bound by.
      unsigned pos = blockIdx.x * (unsigned)PAIRS_PER_BLOCK + ex_items;
#pragma unroll
      for (int k = 0; k < BIG_WALK; ++k) {
        if (k < nb) {
          const unsigned m = (unsigned)((k < 4 ? mask_lo >> (16 * k) : mask_hi >> (16 * (k - 4))) & 0xffffull);
          if (m) {
            pairs[pos++] = make_uint4((unsigned)g, depth_bits, dup, m | ((unsigned)(cy * CX + cx) << 16));
            dup += (unsigned)__popc(m);
          }
          if (++cx == cx1) { cx = cx0; ++cy; }
        }
      }
      if (threadIdx.x == 0) block_items[blockIdx.x] = fits ? total_items : 0u;
    } else {
    const unsigned long long le = (2ull << lane) - 1ull;   // lanes <= this one
    for (int k = 0; __ballot(k < nb) != 0ull; ++k) {
      unsigned m = 0;
      if (k < nb) m = (unsigned)((k < 4 ? mask_lo >> (16 * k) : mask_hi >> (16 * (k - 4))) & 0xffffull);
      const int cb = cy * CX + cx;
      const unsigned key = m ? (unsigned)cb : 0x80000000u | (unsigned)lane;   // idle lanes break the runs
      const unsigned prev = (unsigned)__shfl_up((int)key, 1);
      const unsigned long long heads = __ballot(lane == 0 || key != prev);
      const int h = 63 - __clzll(heads & le);                      // first lane of this lane's run
      const unsigned long long above = heads & ~le;
      const int next = above ? __ffsll((long long)above) - 1 : 64;  // first lane of the next run
      // the counter is 64 bits wide: items appended in the low word (the returned value is the run's first rank), the
      // TILE hits of those items in the high word -- plan_scan turns the per-bin hit totals into list-slot bases, so that
      // fine_bin needs no allocator (2 040 returning atomics on one word were a third of its time)
      const unsigned pc = m ? (unsigned)__popc(m) : 0u;
      unsigned run_hits = pc;                      // every lane its own run (a shuffled scene): no scan needed
      if (heads != ~0ull) {                        // wave-uniform
        const unsigned pincl = wave_incl_scan_u32(pc);
        run_hits = (unsigned)__shfl((int)pincl, next - 1) - (pincl - pc);   // meaningful in lane h
      }
      unsigned rank0 = 0;
      // (the 64-bit atomic costs preprocess ~5 us on the shuffled headline scene -- 14 + 18 bits packed into a 32-bit
      // one ~3 us, not worth a capacity-dependent format -- against 20 us of allocator queueing removed from fine_bin)
      if (m && h == lane)
        rank0 = (unsigned)atomicAdd(reinterpret_cast<unsigned long long*>(&coarse_count[(size_t)cb * CC_STRIDE]),
                                    (unsigned long long)(unsigned)(next - h) | ((unsigned long long)run_hits << 32));
      rank0 = (unsigned)__shfl((int)rank0, h);
      if (m) {
        const unsigned rank = rank0 + (unsigned)(lane - h);
        if (rank < coarse_capacity) slabs[(size_t)cb * coarse_capacity + rank] = make_uint4((unsigned)g, depth_bits, dup, m);
        else hdr[HDR_OVERFLOW] = 1ull;
        dup += (unsigned)__popc(m);
      }
      if (k < nb && ++cx == cx1) { cx = cx0; ++cy; }
    }
    }
  }
  for (unsigned long long bl = fits ? big_lanes : 0ull; bl; bl &= bl - 1) {  // cooperative emission (fits is block-uniform)
    const int L = __builtin_ctzll(bl);
    const WalkArgs a = broadcast_walk(r, br, thr, cx0, cx1, cy0, cy1, L);
    const unsigned g_L = (unsigned)__shfl((int)g, L), depth_L = __shfl(depth_bits, L);
    unsigned dup = __shfl((unsigned)(base + ex), L);
    const int ncb = (a.cx1 - a.cx0) * (a.cy1 - a.cy0);
    // ONE returning atomic instruction per splat (round 4). The walk takes up to 16 iterations of four coarse bins (a
    // mid-size splat has < 64 of them); the item of (iteration it, group q) is appended by lane 16 q + it, which keeps the
    // iteration's mask / bin / first duplicate index when its turn comes -- so all of the splat's <= 63 atomics are in
    // flight TOGETHER and the wave waits for one round trip per splat instead of one per iteration (the near-camera
    // regime -- 200 k splats of ~36 coarse bins -- spent most of this kernel's 2.4 ms in those waits, one rank at a time).
    const int q = lane >> 4, turn = lane & 15;
    const unsigned long long below_q = (1ull << (16 * q)) - 1ull;
    unsigned m_mine = 0u, dp_mine = 0u;
    int cb_mine = 0;
    for (int it = 0; it * 4 < ncb; ++it) {
      int cb;
      const unsigned long long bal = walk_ballot4(a, it, lane, f.W, f.H, bound, &cb, CX);
      if (turn == it) {
        m_mine = (unsigned)(bal >> (16 * q)) & 0xffffu;
        cb_mine = cb;
        dp_mine = dup + (unsigned)__popcll(bal & below_q);
      }
      dup += (unsigned)__popcll(bal);
    }
    if (m_mine) {
      const unsigned rank = (unsigned)atomicAdd(reinterpret_cast<unsigned long long*>(&coarse_count[(size_t)cb_mine * CC_STRIDE]),
                                                1ull | ((unsigned long long)(unsigned)__popc(m_mine) << 32));
      if (rank < coarse_capacity) slabs[(size_t)cb_mine * coarse_capacity + rank] = make_uint4(g_L, depth_L, dp_mine, m_mine);
      else hdr[HDR_OVERFLOW] = 1ull;
    }
  }
  // per-block statistics (summed by plan_scan; no contended atomics)
  block_excl_scan_u32<PRE_BLOCK>(vis, &total, s_red);
  if (threadIdx.x == 0) block_nvis[blockIdx.x] = total;
  block_excl_scan_u32<PRE_BLOCK>(dref, &total, s_red);
  if (threadIdx.x == 0) block_dref[blockIdx.x] = total;
}

// ------------------------------------------------------------------------------------------------
// K1s: second pass of the two-pass binning = ONE RADIX PASS over the (coarse bin, item) pairs with the coarse bin as the
// digit, in the classic three steps -- no device atomics at all:
//   bin_count_kernel    one workgroup per SCATTER_BLOCKS consecutive preprocess workgroups (~11 000 pairs on the headline
//                       scene) counts its pairs per coarse bin in LDS (items, tile hits) and writes its row of the
//                       [workgroup][bin] count matrices;
//   bin_rank_kernel     column scan: for every bin the first slab rank of every workgroup's run (behind the items the big
//                       splats appended directly, with atomics, while preprocess ran), and the bin's totals into its
//                       64-bit counter (items | hits, the format of the direct path);
//   bin_scatter_kernel  the same workgroups again: counting-sort their pairs by bin (LDS cursors, 16-bit pair indices)
//                       and write them in SORTED order -- consecutive lanes store consecutive items of a bin's run.
// With one returning device atomic per (workgroup, bin) instead (first version of this round: 0.5 M atomics, all
// workgroups hitting the same 2 040 counter lines at the same moment) the reservation alone took 36 of the pass's 73 us
// (profiles/r3_bin_scatter_ablation.txt). The order of a bin's items is irrelevant (its tiles are sorted afterwards).
// (ablation builds of this kernel -- no sorted stores / no sweeps / no counting: profiles/r3_bin_scatter_ablation.txt)
constexpr int SCATTER_NT = 1024, SCATTER_BINS = 4096, SCATTER_IDX = 15360, SCATTER_MLP = 4;
static_assert(SCATTER_BLOCKS * PAIRS_PER_BLOCK <= 65536, "16-bit pair indices");
static_assert(SCATTER_BLOCKS <= 64, "one wave scans the block counts");

// prefix sums of the workgroup's SCATTER_BLOCKS block counts -> s_prefix[0 .. nblk]; returns the total
__device__ __forceinline__ unsigned scatter_prefix(const uint32_t* __restrict__ block_items, int b0, int nblk,
                                                   unsigned* s_prefix) {
  const int tid = threadIdx.x;
  if (tid < 64) {
    const unsigned c = tid < nblk ? block_items[b0 + tid] : 0u;
    const unsigned incl = wave_incl_scan_u32(c);
    if (tid < SCATTER_BLOCKS) s_prefix[tid + 1] = incl;
    if (tid == 0) s_prefix[0] = 0u;
  }
  __syncthreads();
  return s_prefix[nblk];
}
// flat pair index -> address: the preprocess workgroup it belongs to by binary search over <= 33 prefix sums
__device__ __forceinline__ const uint4* scatter_pair(const uint4* __restrict__ pairs, const unsigned* s_prefix, int b0,
                                                     int nblk, unsigned i) {
  int lo = 0, hi = nblk;                 // s_prefix[lo] <= i < s_prefix[hi]
#pragma unroll
  for (int it = 0; it < 6; ++it) {       // 2^6 > SCATTER_BLOCKS
    const int mid = (lo + hi) >> 1;
    if (hi - lo > 1) { if (s_prefix[mid] <= i) lo = mid; else hi = mid; }
  }
  return pairs + (size_t)(b0 + lo) * PAIRS_PER_BLOCK + (i - s_prefix[lo]);
}

__global__ void __launch_bounds__(SCATTER_NT)
bin_count_kernel(int NB, int NCB, const uint4* __restrict__ pairs, const uint32_t* __restrict__ block_items,
                 uint32_t* __restrict__ sc_cnt, uint32_t* __restrict__ sc_hits) {
  __shared__ unsigned s_prefix[SCATTER_BLOCKS + 1];
  __shared__ unsigned s_items[SCATTER_BINS], s_hits[SCATTER_BINS];
  const int tid = threadIdx.x;
  const int b0 = blockIdx.x * SCATTER_BLOCKS, nblk = min(SCATTER_BLOCKS, NB - b0);
  const unsigned total = scatter_prefix(block_items, b0, nblk, s_prefix);
  for (int r0 = 0; r0 < NCB; r0 += SCATTER_BINS) {   // one round up to 4 096 bins (1080p: 2 040, 2160p: 8 160)
    const int nbins = min(SCATTER_BINS, NCB - r0);
    for (int i = tid; i < nbins; i += SCATTER_NT) { s_items[i] = 0u; s_hits[i] = 0u; }
    __syncthreads();
    for (unsigned i0 = tid; i0 < total; i0 += SCATTER_MLP * SCATTER_NT) {   // independent loads in flight per thread
      unsigned w[SCATTER_MLP];
#pragma unroll
      for (int u = 0; u < SCATTER_MLP; ++u)
        w[u] = i0 + u * SCATTER_NT < total ? scatter_pair(pairs, s_prefix, b0, nblk, i0 + u * SCATTER_NT)->w : 0u;
#pragma unroll
      for (int u = 0; u < SCATTER_MLP; ++u) {
        const int cb = (int)(w[u] >> 16) - r0;
        if ((w[u] & 0xffffu) && cb >= 0 && cb < nbins) {
          atomicAdd(&s_items[cb], 1u);
          atomicAdd(&s_hits[cb], (unsigned)__popc(w[u] & 0xffffu));
        }
      }
    }
    __syncthreads();
    for (int i = tid; i < nbins; i += SCATTER_NT) {   // the whole row, zeros included: the matrices are not cleared
      sc_cnt[(size_t)blockIdx.x * NCB + r0 + i] = s_items[i];
      sc_hits[(size_t)blockIdx.x * NCB + r0 + i] = s_hits[i];
    }
    __syncthreads();
  }
}

// 16 bins (columns) per workgroup, 64 row groups: thread (g, c) owns positions [g R, (g + 1) R) of column c's rank order
// (R = 4 with the headline's 245 rows); 128 workgroups at 1080p
constexpr int RANK_COLS = 16, RANK_GROUPS = 64;
// Rank ORDER of the scatter workgroups inside a bin's slab: XCD-major (workgroup w runs on XCD w mod 8), so that the ~31
// workgroups of one XCD own ONE contiguous stretch of every slab and its partial lines are completed inside that XCD's L2
// instead of being written back piecemeal from eight of them. k-th in rank order -> workgroup (matrix row).
// (rank order = workgroup order measured slower: profiles/r3_bin_rank_xcd_major_ab.txt)
__device__ __forceinline__ int rank_row(int k, int NWG) {
  const int q = NWG >> 3, r = NWG & 7;               // XCDs x < r have q + 1 workgroups, the others q
  int x = 0, first = 0;                              // (compares instead of integer divisions)
#pragma unroll
  for (int t = 1; t < 8; ++t) {
    const int st = min(t, r) * (q + 1) + max(t - r, 0) * q;   // rank position of XCD t's first workgroup
    if (k >= st) { x = t; first = st; }
  }
  return (k - first) * 8 + x;
}
__global__ void __launch_bounds__(RANK_COLS * RANK_GROUPS)
bin_rank_kernel(int NWG, int NCB, const uint32_t* __restrict__ sc_cnt, const uint32_t* __restrict__ sc_hits,
                uint32_t* __restrict__ sc_base, uint32_t* __restrict__ coarse_count, unsigned long long* __restrict__ hdr) {
  __shared__ unsigned s_c[RANK_GROUPS][RANK_COLS], s_h[RANK_GROUPS][RANK_COLS];
  __shared__ unsigned s_tot[RANK_COLS], s_first[RANK_COLS];
  const int c = threadIdx.x % RANK_COLS, g = threadIdx.x / RANK_COLS;
  const int bin = min(blockIdx.x * RANK_COLS + c, NCB - 1);      // (lanes beyond the last bin repeat it and write nothing)
  const bool live = blockIdx.x * RANK_COLS + c < NCB;
  const int R = (NWG + RANK_GROUPS - 1) / RANK_GROUPS;
  const int w0 = min(g * R, NWG), w1 = min(w0 + R, NWG);
  unsigned long long* counter = reinterpret_cast<unsigned long long*>(&coarse_count[(size_t)bin * CC_STRIDE]);
  const unsigned long long old = *counter;           // what the big splats appended directly (preprocess / big_walk are done)
  unsigned sc = 0u, sh = 0u;
  for (int w = w0; w < w1; w += 4) {                  // four rows in flight (clamped: no predicated loads)
    unsigned tc[4], th[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const size_t at = (size_t)rank_row(min(w + u, w1 - 1), NWG) * NCB + bin;
      tc[u] = sc_cnt[at]; th[u] = sc_hits[at];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (w + u < w1) { sc += tc[u]; sh += th[u]; }
  }
  s_c[g][c] = sc; s_h[g][c] = sh;
  __syncthreads();                                   // (every row group has read the counter BEFORE the last one rewrites it)
  // The bins' runs in BinsView::csr are EXACTLY sized: this workgroup takes the room of its 16 bins with ONE device atomic
  // on the header's cursor (128 atomics per frame at 1080p) and lays the bins out one after the other. Which workgroup comes
  // first is a race -- it decides where a bin's run lies, never what it holds.
  if (g == 0) {
    unsigned tot = 0u;
    for (int q = 0; q < RANK_GROUPS; ++q) tot += s_c[q][c];
    s_tot[c] = live ? tot : 0u;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned sum = 0u;
#pragma unroll
    for (int k = 0; k < RANK_COLS; ++k) { s_first[k] = sum; sum += s_tot[k]; }
    const unsigned at = (unsigned)atomicAdd(&hdr[HDR_CSR_CURSOR], (unsigned long long)sum);
#pragma unroll
    for (int k = 0; k < RANK_COLS; ++k) s_first[k] += at;
  }
  __syncthreads();
  unsigned run = s_first[c];                         // position in BinsView::csr
  for (int q = 0; q < g; ++q) run += s_c[q][c];
  for (int w = w0; w < w1; w += 4) {
    unsigned tc[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) tc[u] = sc_cnt[(size_t)rank_row(min(w + u, w1 - 1), NWG) * NCB + bin];
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (w + u < w1) {
        if (live) sc_base[(size_t)rank_row(w + u, NWG) * NCB + bin] = run;
        run += tc[u];
      }
  }
  if (g == RANK_GROUPS - 1 && live) {
    unsigned hits = (unsigned)(old >> 32);
    for (int q = 0; q < RANK_GROUPS; ++q) hits += s_h[q][c];
    // word 0: ALL items of the bin (the directly appended ones + this run), word 1: their tile hits
    *counter = (unsigned long long)((unsigned)old + s_tot[c]) | ((unsigned long long)hits << 32);
    coarse_count[(size_t)bin * CC_STRIDE + 6] = s_first[c];
    coarse_count[(size_t)bin * CC_STRIDE + 7] = s_tot[c];
  }
}

// the plan's two single-workgroup epilogues (defined below: counters for the host; list-slot bases of the bins)
__device__ void plan_scan_role(int role, int NCB, int NB, uint32_t* __restrict__ coarse_count,
                               const uint32_t* __restrict__ block_nvis, const unsigned long long* __restrict__ block_dref,
                               unsigned long long* __restrict__ hdr, const unsigned long long* __restrict__ feedback,
                               const unsigned long long* __restrict__ dup_pool, unsigned long long* __restrict__ host_out,
                               unsigned coarse_capacity);

// plan_roles = 2: the launch's LAST two workgroups run the plan's epilogues (plan_scan_role) next to the scatter
// workgroups instead of a launch of their own behind them -- everything they read is final before this kernel starts (the
// bins' totals come from bin_rank, the per-block statistics from preprocess), and the one thing the scatter may still
// add, the overflow flag of a bin beyond its capacity, role 0 derives from the fullest bin's count itself. (At the END of
// the grid: scatter workgroup w must be workgroup w of the launch, i.e. run on XCD w mod 8 -- rank_row's slab order
// relies on it)
__global__ void __launch_bounds__(SCATTER_NT)
bin_scatter_kernel(int NB, int NCB, const uint4* __restrict__ pairs, const uint32_t* __restrict__ block_items,
                   const uint32_t* __restrict__ sc_cnt, const uint32_t* __restrict__ sc_base,
                   uint4* __restrict__ csr, unsigned csr_capacity, unsigned coarse_capacity,
                   unsigned long long* __restrict__ hdr, int plan_roles, uint32_t* __restrict__ coarse_count, const uint32_t* __restrict__ block_nvis,
                   const unsigned long long* __restrict__ block_dref, const unsigned long long* __restrict__ feedback,
                   const unsigned long long* __restrict__ dup_pool, unsigned long long* __restrict__ host_out) {
  const int n_scatter = (int)gridDim.x - plan_roles;
  if ((int)blockIdx.x >= n_scatter) {   // uniform per workgroup
    plan_scan_role((int)blockIdx.x - n_scatter, NCB, NB, coarse_count, block_nvis, block_dref, hdr, feedback, dup_pool,
                   host_out, coarse_capacity);
    return;
  }
  const int wg = (int)blockIdx.x;   // scatter workgroup
  __shared__ unsigned s_prefix[SCATTER_BLOCKS + 1];
  __shared__ unsigned s_red[SCATTER_NT / 64 + 1];
  __shared__ unsigned s_delta[SCATTER_BINS];   // first position of the workgroup's run of the bin in `csr` - its first sorted position
  __shared__ unsigned s_cur[SCATTER_BINS];     // the bin's cursor in the sorted order
  __shared__ unsigned short s_idx[SCATTER_IDX];   // pair index at every sorted position
  const int tid = threadIdx.x;
  const int b0 = wg * SCATTER_BLOCKS, nblk = min(SCATTER_BLOCKS, NB - b0);
  const unsigned total = scatter_prefix(block_items, b0, nblk, s_prefix);
  if (total == 0u) return;
  for (int r0 = 0; r0 < NCB; r0 += SCATTER_BINS) {
    const int nbins = min(SCATTER_BINS, NCB - r0);
    // counts -> sorted positions (a thread's four bins are neighbours in that order)
    static_assert(SCATTER_BINS <= 4 * SCATTER_NT, "four bins per thread");
    unsigned c[4], base[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = tid + u * SCATTER_NT;
      c[u] = i < nbins ? sc_cnt[(size_t)wg * NCB + r0 + i] : 0u;
      base[u] = i < nbins ? sc_base[(size_t)wg * NCB + r0 + i] : 0u;
    }
    unsigned round_total;
    unsigned p = block_excl_scan_u32<SCATTER_NT>(c[0] + c[1] + c[2] + c[3], &round_total, s_red);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = tid + u * SCATTER_NT;
      if (i < nbins) { s_delta[i] = base[u] - p; s_cur[i] = p; }
      p += c[u];
    }
    __syncthreads();
    // sorted position of every pair; those beyond the index buffer (a workgroup with > 15 360 pairs in this round of bins)
    // are written at once, unsorted
    for (unsigned i0 = tid; i0 < total; i0 += SCATTER_MLP * SCATTER_NT) {
      uint4 it[SCATTER_MLP];
#pragma unroll
      for (int u = 0; u < SCATTER_MLP; ++u)
        it[u] = i0 + u * SCATTER_NT < total ? *scatter_pair(pairs, s_prefix, b0, nblk, i0 + u * SCATTER_NT)
                                            : make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
      for (int u = 0; u < SCATTER_MLP; ++u) {
        const unsigned m = it[u].w & 0xffffu;
        const int cb = (int)(it[u].w >> 16) - r0;
        if (m && cb >= 0 && cb < nbins) {
          const unsigned pos = atomicAdd(&s_cur[cb], 1u);
          if (pos < (unsigned)SCATTER_IDX) {
            s_idx[pos] = (unsigned short)(i0 + u * SCATTER_NT);
          } else {
            const unsigned at = s_delta[cb] + pos;
            if (at < csr_capacity) csr[at] = make_uint4(it[u].x, it[u].y, it[u].z, m);
            else hdr[HDR_OVERFLOW] = 1ull;   // (cannot happen: the runs hold exactly the pairs that exist)
          }
        }
      }
    }
    __syncthreads();
    const unsigned nsorted = min(round_total, (unsigned)SCATTER_IDX);
    for (unsigned q0 = tid; q0 < nsorted; q0 += SCATTER_MLP * SCATTER_NT) {
      uint4 it[SCATTER_MLP];
#pragma unroll
      for (int u = 0; u < SCATTER_MLP; ++u)
        it[u] = q0 + u * SCATTER_NT < nsorted ? *scatter_pair(pairs, s_prefix, b0, nblk, s_idx[q0 + u * SCATTER_NT])
                                              : make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
      for (int u = 0; u < SCATTER_MLP; ++u) {
        if (q0 + u * SCATTER_NT < nsorted) {
          const int cb = (int)(it[u].w >> 16) - r0;
          const unsigned at = s_delta[cb] + (q0 + u * SCATTER_NT);
          if (at < csr_capacity) csr[at] = make_uint4(it[u].x, it[u].y, it[u].z, it[u].w & 0xffffu);
          else hdr[HDR_OVERFLOW] = 1ull;
        }
      }
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------
// K1b: the binning walk of the splats preprocess_kernel listed as huge (64 coarse bins and more), ONE WAVE PER SPLAT,
// persistent waves over the work list. Two passes over the splat's coarse bins, lane = coarse bin: count the hit
// tiles, reserve the duplicate indices (one atomic per splat), then emit one item per non-empty coarse bin.
// (Mid-size splats stay with the wave that owns them: sent here too, every splat pays the latency chain list entry ->
// record -> walk -> atomic -> stores on its own, and a pitched camera's 600 k mid-size splats took 1.2 ms instead of 0.45.)
__global__ void __launch_bounds__(256)
big_walk_kernel(KFrame kf, const uint4* __restrict__ big_list, const float4* __restrict__ rec,
                uint2* __restrict__ dup_out, uint32_t* __restrict__ coarse_count, uint4* __restrict__ slabs,
                unsigned coarse_capacity, unsigned long long dup_capacity, uint2* __restrict__ big_chunks,
                unsigned big_chunk_cap, unsigned long long* __restrict__ hdr, unsigned long long* __restrict__ dup_pool,
                unsigned npools) {
  // the count pass leaves every coarse bin's 16-bit tile mask in LDS for the emit pass (round 4: the 16 tile tests per bin
  // were done twice); splats of more than BIG_WALK_CACHE coarse bins (images beyond ~2 K) recompute the tail
  __shared__ uint16_t bw_masks[4][BIG_WALK_CACHE];
  uint16_t* my_masks = bw_masks[threadIdx.x >> 6];
  const unsigned n_big = (unsigned)hdr[HDR_BIG_COUNT];
  const int lane = threadIdx.x & 63;
  const unsigned nwaves = gridDim.x * (blockDim.x >> 6);
  const int W = kf.W, H = kf.H;
  const int CX = (((W + TILE_BIN - 1) / TILE_BIN) + COARSE - 1) / COARSE;
  const float bound = __uint_as_float((unsigned)hdr[HDR_SUBPIX_BOUND]);
  for (unsigned i = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); i < n_big; i += nwaves) {
    const uint4 it = big_list[i];
    const unsigned g = it.x;
    const float4 r0 = rec[REC_F4 * (size_t)g], r1 = rec[REC_F4 * (size_t)g + 1], r2 = rec[REC_F4 * (size_t)g + 2];
    WalkArgs a;
    a.r.mx = r0.x; a.r.my = r0.y; a.r.qa = r0.z; a.r.qb = r0.w; a.r.qc = r1.x; a.r.op = r1.y; a.r.depth = r2.y;
    a.r.r = a.r.g = a.r.b = 0.f; a.r.ex = r2.z; a.r.ey = r2.w;
    a.br.x0 = (int)(it.y & 0xffffu); a.br.x1 = (int)(it.y >> 16);
    a.br.y0 = (int)(it.z & 0xffffu); a.br.y1 = (int)(it.z >> 16);
    a.thr = alpha_threshold_log2(a.r.op);
    a.cx0 = a.br.x0 / COARSE; a.cx1 = (a.br.x1 - 1) / COARSE + 1;
    a.cy0 = a.br.y0 / COARSE; a.cy1 = (a.br.y1 - 1) / COARSE + 1;
    const unsigned depth_bits = __float_as_uint(a.r.depth);
    const int nx = a.cx1 - a.cx0, ncb = nx * (a.cy1 - a.cy0);
    // pass 1: count
    unsigned total = 0;   // lane = coarse bin (16 tile tests per lane and iteration, index math amortised)
    for (int i0 = 0; i0 < ncb; i0 += 64) {
      const int j = i0 + lane;
      unsigned c = 0;
      if (j < ncb) {
        const unsigned m = coarse_hits(a.r, a.thr, a.br, a.cx0 + j % nx, a.cy0 + j / nx, W, H, bound);
        if (j < BIG_WALK_CACHE) my_masks[j] = (uint16_t)m;
        c = (unsigned)__popc(m);
      }
      total += wave_sum_u32(c);
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    // reserve the duplicate indices
    unsigned long long base = 0ull;
    bool ok = true;
    if (lane == 0 && total) base = dup_alloc(dup_pool, npools, i, total, dup_capacity, &ok);
    base = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(base >> 32)) << 32) |
           (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)base);
    const bool fits = __builtin_amdgcn_readfirstlane((int)ok) != 0;
    if (lane == 0) {
      dup_out[g] = make_uint2((unsigned)base, total);
      if (!fits) hdr[HDR_OVERFLOW] = 1ull;
    }
    if (!fits || total == 0) continue;
    if (total > BWD_BIG) {  // list this Gaussian's records in chunks for the backward's parallel reduction
      const unsigned nch = (total + BWD_CHUNK - 1) / BWD_CHUNK;
      unsigned long long cb0 = 0ull;
      if (lane == 0) cb0 = atomicAdd(&hdr[HDR_BIG_CHUNKS], (unsigned long long)nch);
      const unsigned c0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)cb0);
      if ((unsigned long long)c0 + nch <= big_chunk_cap) {
        for (unsigned c = lane; c < nch; c += 64) big_chunks[c0 + c] = make_uint2(g, c);
      } else if (lane == 0) {
        hdr[HDR_OVERFLOW] = 1ull;   // cannot happen while dup_capacity >= D (big_chunk_capacity's bound); kept for safety
      }
    }
    // pass 2: emit
    unsigned dup = (unsigned)base;
    for (int i0 = 0; i0 < ncb; i0 += 64) {
      const int j = i0 + lane;
      unsigned m = 0;
      int cb = 0;
      if (j < ncb) {
        const int cx = a.cx0 + j % nx, cy = a.cy0 + j / nx;
        m = j < BIG_WALK_CACHE ? (unsigned)my_masks[j] : coarse_hits(a.r, a.thr, a.br, cx, cy, W, H, bound);
        cb = cy * CX + cx;
      }
      const unsigned c = (unsigned)__popc(m);
      const unsigned incl = wave_incl_scan_u32(c);
      if (m) {
        const unsigned rank = (unsigned)atomicAdd(reinterpret_cast<unsigned long long*>(&coarse_count[(size_t)cb * CC_STRIDE]),
                                                  1ull | ((unsigned long long)c << 32));
        if (rank < coarse_capacity) slabs[(size_t)cb * coarse_capacity + rank] = make_uint4(g, depth_bits, dup + incl - c, m);
        else hdr[HDR_OVERFLOW] = 1ull;
      }
      dup += __shfl(incl, 63);
    }
  }
}

// List-slot bases of the coarse bins (one workgroup, any thread count that is a multiple of 64 up to 1024): bin cb's
// tiles will need at most hits(cb) + 16 x 63 slots (every list starts on a multiple of 64), rounded up to 64; the
// exclusive scan over the bins goes into word 2 of the bin's counter line, the total into the header. Replaces a device
// atomic per bin on ONE allocator word in fine_bin (2 040 returning atomics at ~10 ns: 20 of its 61 us).
__device__ __forceinline__ unsigned bin_slots_needed(unsigned hits) {
  return hits ? ((hits + COARSE_TILES * (LIST_ALIGN - 1) + LIST_ALIGN - 1) & ~(unsigned)(LIST_ALIGN - 1)) : 0u;
}
__device__ void bin_base_scan(int NCB, uint32_t* __restrict__ coarse_count, unsigned long long* __restrict__ hdr) {
  __shared__ unsigned long long bb_wave[17];
  const int nt = blockDim.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = nt >> 6;
  unsigned long long carry = 0;
  for (int c0 = 0; c0 < NCB; c0 += nt) {
    const int cb = c0 + threadIdx.x;
    const unsigned need = cb < NCB ? bin_slots_needed(coarse_count[(size_t)cb * CC_STRIDE + 1]) : 0u;
    unsigned long long incl = need;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const unsigned long long t = __shfl_up(incl, o); if (lane >= o) incl += t; }
    __syncthreads();
    if (lane == 63) bb_wave[wave] = incl;
    __syncthreads();
    unsigned long long before = carry, all = 0;
    for (int w = 0; w < nw; ++w) { const unsigned long long t = bb_wave[w]; if (w < wave) before += t; all += t; }
    if (cb < NCB) coarse_count[(size_t)cb * CC_STRIDE + 2] = (unsigned)min(before + incl - need, 0xffffffffull);
    carry += all;
  }
  if (threadIdx.x == 0) hdr[HDR_ITEM_ALLOC] = carry;
}

__global__ void __launch_bounds__(1024) bin_base_scan_kernel(int NCB, uint32_t* __restrict__ coarse_count,
                                                            unsigned long long* __restrict__ hdr) {
  bin_base_scan(NCB, coarse_count, hdr);
}

// ------------------------------------------------------------------------------------------------
// K2: single workgroup: counters for the host (visible count, reference duplicate total, fullest coarse bin).
constexpr int SCAN_NT = 1024;
static_assert(SCAN_NT == SCATTER_NT, "the plan's epilogues can ride in the scatter launch");
__device__ void plan_scan_role(int role, int NCB, int NB, uint32_t* __restrict__ coarse_count,
                               const uint32_t* __restrict__ block_nvis, const unsigned long long* __restrict__ block_dref,
                               unsigned long long* __restrict__ hdr, const unsigned long long* __restrict__ feedback,
                               const unsigned long long* __restrict__ dup_pool, unsigned long long* __restrict__ host_out,
                               unsigned coarse_capacity) {
  if (role == 1) { bin_base_scan(NCB, coarse_count, hdr); return; }   // second workgroup, concurrently
  __shared__ unsigned long long s_acc[SCAN_NT / 64];
  unsigned long long nvis = 0, dref = 0, cmax = 0, dmax = 0;
  for (int i = threadIdx.x; i < NB; i += SCAN_NT) { nvis += block_nvis[i]; dref += block_dref[i]; }
  for (int i = threadIdx.x; i < NCB; i += SCAN_NT) {
    const uint2 c01 = *reinterpret_cast<const uint2*>(&coarse_count[(size_t)i * CC_STRIDE]);
    const unsigned in_csr = coarse_count[(size_t)i * CC_STRIDE + 7];
    cmax = max(cmax, (unsigned long long)c01.x);               // all items of the bin: what the sort route is chosen from
    dmax = max(dmax, (unsigned long long)(c01.x - in_csr));    // the directly appended ones: what the slab has to hold
  }
  for (int pass = 0; pass < 4; ++pass) {
    unsigned long long v = pass == 0 ? nvis : pass == 1 ? dref : pass == 2 ? cmax : dmax;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
      const unsigned long long o = __shfl_xor(v, d);
      v = pass >= 2 ? max(v, o) : v + o;
    }
    if (lane_id() == 0) s_acc[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
      unsigned long long t = 0;
      for (int w = 0; w < SCAN_NT / 64; ++w) t = pass >= 2 ? max(t, s_acc[w]) : t + s_acc[w];
      hdr[pass == 0 ? HDR_N_VIS : pass == 1 ? HDR_D_REF : pass == 2 ? HDR_MAX_BIN_ITEMS : HDR_MAX_COARSE] = t;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {   // duplicates of the frame = what the pools handed out
    unsigned long long d = 0;
    for (int q = 0; q < DUP_POOLS; ++q) d += dup_pool[q * DP_STRIDE];
    hdr[HDR_D_EFF] = d;
  }
  __syncthreads();
  // publish the first 8 header words straight into the caller's pinned host buffer (fine-grained memory: visible
  // to the host once this kernel has completed) -- the host can then decide about a capacity retry while the render
  // stage is already running, without a copy engine round trip in the middle of the stream
  if (threadIdx.x == 0 && host_out) {
#pragma unroll
    for (int i = 0; i < 8; ++i) host_out[i] = hdr[i];
    // When this role rides in the scatter launch, scatter workgroups may still be about to flag a bin that runs past its
    // slab: the bins' totals are final (bin_rank), so the published overflow word is completed from them
    if (hdr[HDR_MAX_COARSE] > (unsigned long long)coarse_capacity) host_out[HDR_OVERFLOW] = 1ull;
    // words 8..15: this frame's optional-work counts, and the late statistics of the PREVIOUS frame (its render /
    // backward stages completed before this kernel started: same stream) -- what the caller picks launch hints from
    host_out[8] = hdr[HDR_BIG_COUNT];
    host_out[9] = hdr[HDR_BIG_CHUNKS];
    host_out[10] = feedback ? feedback[FB_VALID] : 0ull;
    host_out[11] = feedback ? feedback[FB_LONG_TILES] : 0ull;
    host_out[12] = feedback ? feedback[FB_MAX_LIST] : 0ull;
    host_out[13] = feedback ? feedback[FB_PREFILLED] : 0ull;
    host_out[14] = feedback ? feedback[FB_OVER_512] : 0ull;
    host_out[15] = hdr[HDR_MAX_BIN_ITEMS];
  }
}

// the same two roles as a launch of their own (one-pass binning, SFGS_PLAN_SCAN=separate)
__global__ void __launch_bounds__(SCAN_NT)
plan_scan_kernel(int NCB, int NB, uint32_t* __restrict__ coarse_count, const uint32_t* __restrict__ block_nvis,
                 const unsigned long long* __restrict__ block_dref, unsigned long long* __restrict__ hdr,
                 const unsigned long long* __restrict__ feedback, const unsigned long long* __restrict__ dup_pool,
                 unsigned long long* __restrict__ host_out, unsigned coarse_capacity) {
  plan_scan_role((int)blockIdx.x, NCB, NB, coarse_count, block_nvis, block_dref, hdr, feedback, dup_pool, host_out,
                 coarse_capacity);
}

// ------------------------------------------------------------------------------------------------
// K3: fine binning. One 256-thread workgroup per coarse bin: count the bin's 16 tiles with LDS atomics,
// reserve the bin's list slots with ONE device atomic, publish (start, length) of its tiles, then expand
// every coarse item into its per-tile duplicates: items[slot] = (id, depth, dup index, 0).
__global__ void __launch_bounds__(256)
fine_bin_kernel(int TX8, int TY8, int CX, int NCB, const uint32_t* __restrict__ coarse_count,
                const uint4* __restrict__ csr, const uint4* __restrict__ slabs, unsigned coarse_capacity,
                unsigned long long slot_capacity,
                uint2* __restrict__ tile_range, uint4* __restrict__ items, uint32_t* __restrict__ long_tiles,
                unsigned long long* __restrict__ hdr) {
  __shared__ unsigned cnt[COARSE_TILES];
  __shared__ unsigned long long s_base;
  const int cb = blockIdx.x, tid = threadIdx.x;
  const BinItems slab = bin_items(coarse_count + (size_t)cb * CC_STRIDE, csr, slabs, (size_t)cb, coarse_capacity);
  const unsigned n = slab.n;
  if (tid < COARSE_TILES) cnt[tid] = 0;
  __syncthreads();
  // the bin's coarse items are read ONCE, all loads of a thread in flight together (the kernel is latency-bound: one
  // round of workgroups, each a chain of dependent global loads and LDS atomics); bins with more than 256 * FB_R items
  // take the remainder from memory in both passes
  constexpr int FB_R = 12;
  uint4 reg[FB_R];
  // UNCONDITIONAL loads from a clamped index, the tile mask cleared afterwards: written as `i < n ? slab[i] : 0` every
  // load sat in its own exec-masked block with an s_waitcnt vmcnt(0) right behind it -- twelve serialised round trips to
  // memory per workgroup instead of one (found by reading the ISA, round 3)
  const unsigned last = n ? n - 1u : 0u;
#pragma unroll
  for (int k = 0; k < FB_R; ++k) reg[k] = slab[min(tid + 256u * k, last)];
#pragma unroll
  for (int k = 0; k < FB_R; ++k)
    if (tid + 256u * k >= n) reg[k].w = 0u;
#pragma unroll
  for (int k = 0; k < FB_R; ++k) {
    unsigned m = reg[k].w;
    while (m) { const int b = __builtin_ctz(m); m &= m - 1; atomicAdd(&cnt[b], 1u); }
  }
  for (unsigned i = tid + 256u * FB_R; i < n; i += 256) {
    unsigned m = slab[i].w;
    while (m) { const int b = __builtin_ctz(m); m &= m - 1; atomicAdd(&cnt[b], 1u); }
  }
  __syncthreads();
  if (tid < 64) {  // first wave: exclusive scan of the 16 counts, one allocation for the whole bin
    const unsigned c = tid < COARSE_TILES ? cnt[tid] : 0u;
    // every list starts on a multiple of LIST_ALIGN (= 64) slots: the compositing kernels load ids 64 at a time from
    // 256-byte aligned addresses, and slot s + 64 b + lane doubles as the index of the per-pixel hit-mask word of the
    // tile's b-th batch (image blob). Costs index space only (<= 63 unused slots per tile), no traffic.
    const unsigned ca = (c + LIST_ALIGN - 1) & ~(unsigned)(LIST_ALIGN - 1);
    unsigned incl = ca;
#pragma unroll
    for (int d = 1; d < COARSE_TILES; d <<= 1) { const unsigned t = __shfl_up(incl, d); if (tid >= d) incl += t; }
    const unsigned total = __shfl(incl, COARSE_TILES - 1);
    unsigned long long base = 0;
    if (tid == 0) {
      // the bin's first list slot: scanned by the plan from the bins' tile-hit totals (an upper bound of what the bin's
      // lists take: `total` <= hits + 16 x 63), no allocator atomic
      base = coarse_count[(size_t)cb * CC_STRIDE + 2];
      // memory safety when the caller's slot capacity is too small (the wrapper sizes it with
      // sfgs_raster_slot_capacity, which always suffices): the bin's lists are dropped and the frame is flagged
      if (base + total > slot_capacity) { hdr[HDR_OVERFLOW] = 1ull; base = ~0ull; }
      s_base = base;
    }
    base = __shfl(base, 0);
    const bool dropped = base == ~0ull;
    const unsigned off = incl - ca;
    if (tid < COARSE_TILES) {
      const int tx = (cb % CX) * COARSE + (tid & (COARSE - 1)), ty = (cb / CX) * COARSE + (tid / COARSE);
      if (tx < TX8 && ty < TY8) {
        tile_range[ty * TX8 + tx] = dropped ? make_uint2(0u, 0u) : make_uint2((unsigned)base + off, c);
        if (c > REG_SORT_SMALL && !dropped) long_tiles[atomicAdd(&hdr[HDR_LONG_COUNT], 1ull)] = (unsigned)(ty * TX8 + tx);  // rare
      }
      cnt[tid] = off;  // becomes the per-tile cursor
    }
    {  // longest list of the bin: ONE atomic per workgroup (16 same-address atomics per bin serialise in the L2)
      unsigned cm = tid < COARSE_TILES ? c : 0u;
#pragma unroll
      for (int d = 8; d >= 1; d >>= 1) cm = max(cm, (unsigned)__shfl_xor((int)cm, d));
      if (tid == 0 && cm) atomicMax((unsigned int*)&hdr[HDR_MAX_LIST], cm);
    }
  }
  __syncthreads();
  const unsigned long long base = s_base;
  if (base == ~0ull) return;
  auto expand = [&](const uint4 it) {
    unsigned m = it.w, dup = it.z;
    while (m) {
      const int b = __builtin_ctz(m);
      m &= m - 1;
      const unsigned slot = atomicAdd(&cnt[b], 1u);
      items[base + slot] = make_uint4(it.x, it.y, dup, 0u);
      ++dup;
    }
  };
#pragma unroll
  for (int k = 0; k < FB_R; ++k) expand(reg[k]);
  for (unsigned i = tid + 256u * FB_R; i < n; i += 256) expand(slab[i]);
}

// ------------------------------------------------------------------------------------------------
// K4a: per-tile sort for lists of up to 512 entries, entirely in registers. One wave per tile; element
// (lane, r) of the wave carries label e = lane * EPL + r. Normalised bitonic network: for every size the
// first step pairs e with e ^ (size - 1), the following steps with e ^ stride; the lower label keeps the
// smaller key. Label bits below log2(EPL) are register indices (compile-time), the bits above are lane
// bits: those exchanges are cross-lane shuffles (3 per element: key hi/lo + payload). No LDS traffic, no
// bank conflicts, no barriers. Keys: (depth bits << 32 | Gaussian id) == SURVEY A.3 order.
// value of lane (l ^ K): DPP where the hardware has the permutation (no LDS latency), the LDS crossbar
// (ds_swizzle / ds_bpermute) otherwise.
template <int K>
__device__ __forceinline__ unsigned xor_shuffle(unsigned v) {
  if constexpr (K == 1) return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, false);       // quad_perm [1,0,3,2]
  else if constexpr (K == 2) return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, false);  // quad_perm [2,3,0,1]
  else if constexpr (K == 3) return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x1B, 0xf, 0xf, false);  // quad_perm [3,2,1,0]
  else if constexpr (K == 7) return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xf, 0xf, false); // row_half_mirror
  else if constexpr (K == 15) return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xf, 0xf, false); // row_mirror
  else if constexpr (K < 32) return (unsigned)__builtin_amdgcn_ds_swizzle((int)v, (K << 10) | 0x1f);           // bit mode: xor K
  else return (unsigned)__shfl_xor((int)v, K);
}

template <int EPL, int SIZE, int STEP>
__device__ __forceinline__ void bitonic_stage(unsigned long long (&key)[EPL], unsigned (&pay)[EPL], int lane) {
  constexpr int mask = STEP == 0 ? SIZE - 1 : (SIZE >> (STEP + 1));
  constexpr int rmask = mask & (EPL - 1);   // register-index part of the label
  constexpr int lmask = mask / EPL;         // lane part
  if constexpr (lmask == 0) {
#pragma unroll
    for (int r = 0; r < EPL; ++r) {
      const int q = r ^ rmask;
      if (q > r) {
        const bool sw = key[q] < key[r];
        const unsigned long long kr = key[r], kq = key[q];
        const unsigned pr = pay[r], pq = pay[q];
        key[r] = sw ? kq : kr; key[q] = sw ? kr : kq;
        pay[r] = sw ? pq : pr; pay[q] = sw ? pr : pq;
      }
    }
  } else {
    const bool upper = (lane & (lmask & ~(lmask >> 1))) != 0;  // highest bit of the mask decides who is lower
    unsigned long long nk[EPL];
    unsigned np[EPL];
#pragma unroll
    for (int r = 0; r < EPL; ++r) {
      const int q = r ^ rmask;  // partner's register index
      const unsigned hi = xor_shuffle<lmask>((unsigned)(key[q] >> 32));
      const unsigned lo = xor_shuffle<lmask>((unsigned)(key[q] & 0xffffffffull));
      const unsigned pp = xor_shuffle<lmask>(pay[q]);
      const unsigned long long theirs = ((unsigned long long)hi << 32) | lo;
      const bool take = (theirs < key[r]) != upper;  // the lower label keeps the minimum, the upper the maximum
      nk[r] = take ? theirs : key[r];
      np[r] = take ? pp : pay[r];
    }
#pragma unroll
    for (int r = 0; r < EPL; ++r) { key[r] = nk[r]; pay[r] = np[r]; }
  }
}

template <int EPL, int SIZE, int STEP>
__device__ __forceinline__ void bitonic_steps(unsigned long long (&key)[EPL], unsigned (&pay)[EPL], int lane) {
  if constexpr (STEP == 0 || (SIZE >> (STEP + 1)) >= 1) {
    bitonic_stage<EPL, SIZE, STEP>(key, pay, lane);
    bitonic_steps<EPL, SIZE, STEP + 1>(key, pay, lane);
  }
}

template <int EPL, int SIZE>
__device__ __forceinline__ void bitonic_sizes(unsigned long long (&key)[EPL], unsigned (&pay)[EPL], int lane) {
  if constexpr (SIZE <= 64 * EPL) {
    bitonic_steps<EPL, SIZE, 0>(key, pay, lane);
    bitonic_sizes<EPL, SIZE * 2>(key, pay, lane);
  }
}

template <int EPL>
__device__ __forceinline__ void wave_bitonic_sort(unsigned long long (&key)[EPL], unsigned (&pay)[EPL], int lane) {
  bitonic_sizes<EPL, 2>(key, pay, lane);
}

// A lane's EPL consecutive list entries (positions lane * EPL ..) as 16-byte (8- / 4-byte) stores instead of EPL dword
// stores 4 * EPL bytes apart (round 4: one scattered dword store per entry was the expensive way to write a list, cf.
// profiles/r4_bwd_store3_ab.txt). Every list owns its slots up to the next multiple of 64 (LIST_ALIGN), so a lane whose
// group starts inside that range writes all of it; the entries beyond the list's length are padding nobody reads.
template <int EPL>
__device__ __forceinline__ void store_list_entries(uint32_t* __restrict__ dst, const unsigned (&v)[EPL]) {
  if constexpr (EPL >= 4) {
#pragma unroll
    for (int i = 0; i < EPL / 4; ++i)
      reinterpret_cast<uint4*>(dst)[i] = make_uint4(v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]);
  } else if constexpr (EPL == 2) {
    *reinterpret_cast<uint2*>(dst) = make_uint2(v[0], v[1]);
  } else {
    dst[0] = v[0];
  }
}

template <int EPL>
__device__ __forceinline__ void sort_tile_in_registers(unsigned s, int L, int lane, const uint4* __restrict__ items,
                                                       uint32_t* __restrict__ sorted_id,
                                                       uint32_t* __restrict__ sorted_dup) {
  unsigned long long key[EPL];
  unsigned pay[EPL];
  // coalesced (striped) loads, all in flight together: unconditional from a clamped index, padding selected afterwards
  // (a load under `if (i < L)` gets its own exec-masked block and, every few of them, an s_waitcnt vmcnt(0))
  uint4 it[EPL];
#pragma unroll
  for (int r = 0; r < EPL; ++r) it[r] = items[s + min(r * 64 + lane, L - 1)];
#pragma unroll
  for (int r = 0; r < EPL; ++r) {   // the network sorts any initial arrangement
    const bool in = r * 64 + lane < L;
    key[r] = in ? (((unsigned long long)it[r].y << 32) | it[r].x) : ~0ull;
    pay[r] = in ? it[r].z : 0u;
  }
  wave_bitonic_sort<EPL>(key, pay, lane);
  static_assert(LIST_ALIGN % EPL == 0, "a lane's entries lie inside or outside the list's slot range as a whole");
  if (lane * EPL < ((L + LIST_ALIGN - 1) & ~(LIST_ALIGN - 1))) {
    unsigned ids[EPL];
#pragma unroll
    for (int r = 0; r < EPL; ++r) ids[r] = (unsigned)(key[r] & 0xffffffffull);
    store_list_entries<EPL>(sorted_id + s + lane * EPL, ids);
    store_list_entries<EPL>(sorted_dup + s + lane * EPL, pay);
  }
}

__global__ void __launch_bounds__(256)
sort_tiles_reg_kernel(int T8, const uint2* __restrict__ tile_range, const uint4* __restrict__ items,
                      uint32_t* __restrict__ sorted_id, uint32_t* __restrict__ sorted_dup,
                      const unsigned long long* __restrict__ hdr, unsigned long long* __restrict__ feedback) {
  if (feedback && blockIdx.x == 0 && threadIdx.x == 0) {   // the list statistics are final (fine_bin has run): leave
    feedback[FB_VALID] = 1ull;                              // them for the next frame's plan (SfgsFrame.feedback)
    feedback[FB_LONG_TILES] = hdr[HDR_LONG_COUNT];
    feedback[FB_MAX_LIST] = hdr[HDR_MAX_LIST];
    feedback[FB_OVER_512] = hdr[HDR_LONG_COUNT];           // this route lists exactly the tiles beyond 512 entries
  }
  const int t = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (t >= T8) return;
  const uint2 tr = tile_range[t];
  const int L = (int)tr.y;
  if (L == 0 || L > REG_SORT_SMALL) return;
  if (L <= 64) sort_tile_in_registers<1>(tr.x, L, lane, items, sorted_id, sorted_dup);
  else if (L <= 128) sort_tile_in_registers<2>(tr.x, L, lane, items, sorted_id, sorted_dup);
  else if (L <= 256) sort_tile_in_registers<4>(tr.x, L, lane, items, sorted_id, sorted_dup);
  else sort_tile_in_registers<8>(tr.x, L, lane, items, sorted_id, sorted_dup);
}

// K3+K4a fused (round 3): SELECT + SORT, no per-tile item round trip through memory.
// fine_bin expands the coarse items of a bin into 16-byte per-tile items in memory (111 MB written at the headline
// scene) which the sort kernel reads straight back. Here a workgroup takes one ROW of four tiles of a coarse bin (wave =
// tile): it reads the bin's slab once (21 KB on average; the bin's four workgroups are XCD-contiguous and share the L2),
// hands every coarse item to the wave-private LDS lists of the row's tiles its mask names (12 bytes per entry: key =
// depth bits << 32 | id, payload = duplicate index = the item's first index + the mask bits below the tile's; positions
// from LDS atomics -- the order inside a list does not matter before the sort), then every wave sorts its list with the
// register network and writes ids / duplicate indices in their final order. The tile's first slot inside the bin's slot
// range (the plan's scan, bin_slots_needed) comes from one atomic per tile on the bin's OWN counter line (word 3: a cursor
// in units of slots, 16 atomics per line; placement only -- which tile of a bin sits where never reaches a result). Lists
// beyond REG_SORT_SMALL entries (rare) are written out as unsorted items by a second scan of the tile's wave and left to
// the long-list kernels, exactly as fine_bin leaves them. The longest list of the frame is collected per bin (word 4 of
// the line) and reduced by list_stats.
// SS_CAP = entries a tile's LDS list holds = the longest list sorted here: 512 (24 KB of LDS per workgroup, <= 8 keys
// per lane: the SHORT_LISTS form) or 1 024 (round 4: 48 KB, three workgroups per CU, the 16-key network for the lists
// beyond 512 -- the MEDIUM_LISTS form for frames whose lists reach 513..1 024 entries, e.g. the reference's 45 / 25 degree
// IDU cameras (arguments/__init__.py:238-249), which otherwise fell back to fine_bin + two sort kernels).
// (A wave-level LDS radix sort for the lists beyond 256 entries was built and measured in round 4 -- slower below ~700
// entries, not kept: profiles/r4_radix_sort_ab_not_kept.txt.)
template <int SS_CAP>
struct alignas(16) SelectSortLds {
  unsigned long long key[SS_CAP];
  uint32_t pay[SS_CAP];
  uint32_t pad_[4];
};


template <int SS_CAP>
__global__ void __launch_bounds__(256)
select_sort_kernel(int TX8, int TY8, int CX, int NCB, uint32_t* __restrict__ coarse_count,
                   const uint4* __restrict__ csr, const uint4* __restrict__ slabs, unsigned coarse_capacity,
                   unsigned long long slot_capacity,
                   uint2* __restrict__ tile_range, uint4* __restrict__ items, uint32_t* __restrict__ long_tiles,
                   unsigned long long* __restrict__ hdr, uint32_t* __restrict__ sorted_id,
                   uint32_t* __restrict__ sorted_dup) {
  __shared__ SelectSortLds<SS_CAP> lds_all[COARSE];
  __shared__ unsigned s_cnt[COARSE];
  // workgroup = one row of four tiles of a coarse bin (wave = tile); the four workgroups of a bin are neighbours on one XCD
  // (chunks of 8 bins = 32 workgroups per XCD turn, dealt round-robin: balanced on frames whose bins are not equally full)
  const unsigned lb = xcd_chunked(blockIdx.x, (unsigned)NCB * COARSE, 8u * COARSE);
  const int cb = (int)(lb / COARSE), q = (int)(lb % COARSE);
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int tx = (cb % CX) * COARSE + wave, ty = (cb / CX) * COARSE + q;
  if (ty >= TY8) return;                             // the whole row lies below the image (uniform per workgroup)
  const int t = ty * TX8 + tx;
  const int bit = q * COARSE + wave;                 // this tile's bit of the coarse items' masks
  uint32_t* line = coarse_count + (size_t)cb * CC_STRIDE;
  const BinItems slab = bin_items(line, csr, slabs, (size_t)cb, coarse_capacity);
  const unsigned n = slab.n;
  if (n == 0) { if (lane == 0 && tx < TX8) tile_range[t] = make_uint2(0u, 0u); return; }   // uniform per workgroup
  SelectSortLds<SS_CAP>& lds = lds_all[wave];
  // ---- scan: the workgroup reads the slab ONCE (R items per thread and round trip, unconditional loads from clamped
  // indices) and hands every item to the lists of the row's tiles it touches. Positions come from LDS atomics: the order
  // inside a list is irrelevant here, the sort below fixes it ((depth, id) keys are unique within a tile).
  if (tid < COARSE) s_cnt[tid] = 0u;
  __syncthreads();
  constexpr int R = 8;
  const unsigned rowmask = 0xfu << (q * COARSE);
  for (unsigned i0 = 0; i0 < n; i0 += 256u * R) {
    uint4 it[R];
#pragma unroll
    for (int k = 0; k < R; ++k) it[k] = slab[min(i0 + 256u * k + tid, n - 1u)];
#pragma unroll
    for (int k = 0; k < R; ++k) {
      unsigned m = i0 + 256u * k + tid < n ? it[k].w & rowmask : 0u;
      while (m) {
        const int b = __builtin_ctz(m);
        m &= m - 1;
        const unsigned pos = atomicAdd(&s_cnt[b - q * COARSE], 1u);
        if (pos < (unsigned)SS_CAP) {
          SelectSortLds<SS_CAP>& dst = lds_all[b - q * COARSE];
          dst.key[pos] = ((unsigned long long)it[k].y << 32) | it[k].x;
          dst.pay[pos] = it[k].z + (unsigned)__popc(it[k].w & ((1u << b) - 1u));
        }
      }
    }
  }
  __syncthreads();
  if (tx >= TX8) return;                             // a tile right of the image: no item has its bit
  const unsigned below = (1u << bit) - 1u;
  const unsigned c = __builtin_amdgcn_readfirstlane((int)s_cnt[wave]);
  // ---- the tile's slots: a stretch of the bin's range (scanned by the plan from the bin's tile hits) ---------------
  const unsigned long long bin_base = line[2];
  const unsigned need = bin_slots_needed(line[1]);
  const bool dropped = bin_base + need > slot_capacity;   // the caller's slot capacity is too small (see fine_bin)
  unsigned off = 0;
  if (lane == 0) {
    if (dropped) hdr[HDR_OVERFLOW] = 1ull;
    else if (c) {
      off = atomicAdd(&line[3], (c + LIST_ALIGN - 1) & ~(unsigned)(LIST_ALIGN - 1));
      atomicMax(&line[4], c);
      if (c > (unsigned)REG_SORT_SMALL) atomicAdd(&line[5], 1u);   // how many of the frame's lists need the 1 024 form
    }
  }
  if (dropped || c == 0) { if (lane == 0) tile_range[t] = make_uint2(0u, 0u); return; }
  if (c <= (unsigned)SS_CAP) {
    // sort first, then address: the returning atomic is only needed by the stores behind the sort
    const int L = (int)c;
    auto finish = [&](auto epl_tag) {
      constexpr int EPL = decltype(epl_tag)::value;
      unsigned long long key[EPL];
      unsigned pay[EPL];
#pragma unroll
      for (int r = 0; r < EPL; ++r) {
        const int i = r * 64 + lane;
        const unsigned long long k = lds.key[min(i, SS_CAP - 1)];
        const unsigned p = lds.pay[min(i, SS_CAP - 1)];
        key[r] = i < L ? k : ~0ull;
        pay[r] = i < L ? p : 0u;
      }
      wave_bitonic_sort<EPL>(key, pay, lane);
      const unsigned s = (unsigned)bin_base + (unsigned)__builtin_amdgcn_readfirstlane((int)off);
      if (lane == 0) tile_range[t] = make_uint2(s, c);
      if (lane * EPL < ((L + LIST_ALIGN - 1) & ~(LIST_ALIGN - 1))) {   // the tile owns its slots up to the next multiple of 64
        unsigned ids[EPL];
#pragma unroll
        for (int r = 0; r < EPL; ++r) ids[r] = (unsigned)(key[r] & 0xffffffffull);
        store_list_entries<EPL>(sorted_id + s + lane * EPL, ids);
        store_list_entries<EPL>(sorted_dup + s + lane * EPL, pay);
      }
    };
    if (L <= 64) finish(std::integral_constant<int, 1>{});
    else if (L <= 128) finish(std::integral_constant<int, 2>{});
    else if (L <= 256) finish(std::integral_constant<int, 4>{});
    else if (SS_CAP <= 512 || L <= 512) finish(std::integral_constant<int, 8>{});
    else if constexpr (SS_CAP > 512) finish(std::integral_constant<int, 16>{});
    return;
  }
  // ---- long list: unsorted 16-byte items for the long-list kernels, like fine_bin ------------------------------------
  const unsigned s = (unsigned)bin_base + (unsigned)__builtin_amdgcn_readfirstlane((int)off);
  if (lane == 0) {
    tile_range[t] = make_uint2(s, c);
    long_tiles[atomicAdd(&hdr[HDR_LONG_COUNT], 1ull)] = (unsigned)t;
  }
  unsigned done = 0;
  for (unsigned i0 = 0; i0 < n; i0 += 64u * R) {   // R loads of the wave in flight per round trip
    uint4 lt[R];
#pragma unroll
    for (int k = 0; k < R; ++k) lt[k] = slab[min(i0 + 64u * k + lane, n - 1u)];
#pragma unroll
    for (int k = 0; k < R; ++k) {
      const bool hit = i0 + 64u * k + lane < n && ((lt[k].w >> bit) & 1u);
      const unsigned long long b = __ballot(hit);
      const unsigned pos = done + __builtin_amdgcn_mbcnt_hi((unsigned)(b >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)b, 0u));
      if (hit) items[s + pos] = make_uint4(lt[k].x, lt[k].y, lt[k].z + (unsigned)__popc(lt[k].w & below), 0u);
      done += (unsigned)__popcll(b);
    }
  }
}

// longest list of the frame = max over the bins' maxima (select_sort_kernel), and the list statistics for the next
// frame's plan (SfgsFrame.feedback). Run by the first workgroup of the (always launched) long-list kernel: 256 threads.
__device__ void list_stats(int NCB, uint32_t* __restrict__ coarse_count, unsigned long long* hdr,
                           unsigned long long* __restrict__ feedback) {
  __shared__ unsigned ls_part[4], ls_over[4];
  unsigned m = 0, over = 0;
  for (int i = threadIdx.x; i < NCB; i += 256) {
    m = max(m, coarse_count[(size_t)i * CC_STRIDE + 4]);
    over += coarse_count[(size_t)i * CC_STRIDE + 5];
    // (a plan is single-use, include/sfgs.h: the bins' slot cursors, their maxima and the header's long-list words are
    // only reset by the next plan's memset)
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) {
    m = max(m, (unsigned)__shfl_xor((int)m, d));
    over += (unsigned)__shfl_xor((int)over, d);
  }
  if ((threadIdx.x & 63) == 0) { ls_part[threadIdx.x >> 6] = m; ls_over[threadIdx.x >> 6] = over; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned long long mx = max(max(ls_part[0], ls_part[1]), max(ls_part[2], ls_part[3]));
    hdr[HDR_MAX_LIST] = mx;
    if (feedback) {
      feedback[FB_VALID] = 1ull;
      feedback[FB_LONG_TILES] = hdr[HDR_LONG_COUNT];
      feedback[FB_MAX_LIST] = mx;
      feedback[FB_OVER_512] = (unsigned long long)ls_over[0] + ls_over[1] + ls_over[2] + ls_over[3];
    }
  }
  __syncthreads();
}

// K4a': the same register network with EPL = 16 keys per lane for lists of 513..1024 entries (low-elevation views, dense
// frames). A kernel of its own: 99 VGPRs would otherwise cut the occupancy of the common short lists. One wave per tile,
// persistent over the device-side list of long tiles. (Longer lists: the bucketed sort of sort_tiles_long_kernel.)
template <int EPL>
__global__ void __launch_bounds__(256)
sort_tiles_reg_long_kernel(const uint32_t* __restrict__ long_tiles, const unsigned long long* __restrict__ hdr,
                           const uint2* __restrict__ tile_range, const uint4* __restrict__ items,
                           uint32_t* __restrict__ sorted_id, uint32_t* __restrict__ sorted_dup) {
  const unsigned n_long = (unsigned)hdr[HDR_LONG_COUNT];
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
  for (unsigned li = blockIdx.x * 4 + wave; li < n_long; li += gridDim.x * 4) {
    const uint2 tr = tile_range[long_tiles[li]];
    const int L = (int)tr.y;
    if (L > 32 * EPL && L <= 64 * EPL) sort_tile_in_registers<EPL>(tr.x, L, lane, items, sorted_id, sorted_dup);
  }
}

// K4b: per-tile sort of the lists the register network does not take (L > 512), by 64-bit key
// (depth bits << 32 | Gaussian id) == ascending (depth, Gaussian id), SURVEY A.3. One 256-thread workgroup per tile,
// persistent over the device-side list of long tiles; normalised bitonic network (every comparator sorts ascending,
// so the +inf padding stays at the tail and comparators reaching past the end of the list are no-ops).
//   L <= CAP : one pass in LDS (keys + payload, 48 KB: three workgroups per CU). Dense frames (16 M Gaussians at
//              1080p: every list ~1 800 long) run entirely through this path.
//   L  > CAP : hybrid -- every CAP-chunk is sorted in LDS, then for each larger network size only the exchanges
//              with stride >= CAP touch global memory (in place, on the tile's own segment of 16-byte items); the
//              strides below CAP of that size are again one LDS pass per chunk. Global accesses are agent-scope
//              relaxed (they bypass the per-CU L1, so the waves of the workgroup see each other's exchanges after
//              the barrier).
template <int CAP>
__global__ void __launch_bounds__(256, 3)   // three workgroups per CU: what the 52 KB of LDS allow
sort_tiles_long_kernel(int lo, const uint32_t* __restrict__ long_tiles, const unsigned long long* __restrict__ hdr,
                       const uint2* __restrict__ tile_range, uint4* items, uint32_t* __restrict__ sorted_id,
                       uint32_t* __restrict__ sorted_dup, int stats_bins, uint32_t* __restrict__ coarse_count,
                       unsigned long long* hdr_w, unsigned long long* __restrict__ feedback) {
  __shared__ unsigned long long k[CAP];
  __shared__ uint32_t pl[CAP];
  constexpr int NT = 256;
  // behind select_sort_kernel (stats_bins > 0): the frame's list statistics are final now
  if (stats_bins > 0 && blockIdx.x == 0) list_stats(stats_bins, coarse_count, hdr_w, feedback);
  const unsigned n_long = (unsigned)hdr[HDR_LONG_COUNT];
  const int tid = threadIdx.x;
  // ---- bucketed sort (lists of up to 2 CAP entries): O(n) partition by depth, then small register sorts ----------------
  // The list is split into 256 equal-width depth bins (LDS histogram, counting scatter); consecutive bins are grouped
  // into segments of at most SEG entries and every segment is sorted by ONE WAVE in registers on the full 64-bit key
  // (depth bits, id) -- the network of the short lists, no barriers, no LDS exchanges. Only the depth bits and the
  // entry's 16-bit position live in LDS (6 bytes per entry: the same 48 KB hold 2 CAP entries); ids and duplicate indices
  // are gathered from the tile's item segment when a segment is loaded. A bitonic network over 4 096 entries has 78
  // barrier-separated LDS steps, over 8 192 entries 23 of its 91 steps went through global memory; this needs three
  // barriers and log^2(512) = 45 register steps per segment. Falls back to the network when one bin alone exceeds SEG
  // (a list concentrated in < 0.4 % of its own depth range) or all depths are equal.
  constexpr int BK_BINS = 256, SEG = 512, BK_CAP = 2 * CAP;
  uint32_t* bk_depth = reinterpret_cast<uint32_t*>(k);          // [BK_CAP] depth bits, grouped by bin
  uint16_t* bk_pos = reinterpret_cast<uint16_t*>(pl);           // [BK_CAP] position in the tile's item segment
  __shared__ unsigned bk_hist[BK_BINS], bk_start[BK_BINS + 1], bk_red[2 * (NT / 64)];
  __shared__ unsigned short bk_seg_bin[BK_BINS + 1], bk_next[BK_BINS];   // first bin of every segment (+ end marker)
  static_assert(NT == BK_BINS, "one thread per depth bin");
  __shared__ unsigned bk_nseg, bk_fallback;
  // Two LDS layouts: lists of up to CAP entries keep (depth, id, duplicate index) in LDS -- the tile's items are read from
  // memory exactly once; lists of up to 2 CAP entries keep (depth, 16-bit position) and gather id / duplicate index from
  // the item segment when a segment is sorted (a second, scattered read of the items).
  uint32_t* bk_id = reinterpret_cast<uint32_t*>(k) + CAP;         // [CAP]  (narrow layout only)
  auto bucket_sort = [&](unsigned s, int L, auto wide_tag) -> bool {
    constexpr bool WIDE = decltype(wide_tag)::value;
    const int lane = tid & 63, wave = tid >> 6;
    // the thread's depths stay in registers for all three passes (min / max, histogram, scatter): read from memory in
    // every pass, each pass was a chain of dependent global-load latencies
    constexpr int PER = (WIDE ? BK_CAP : CAP) / NT;
    unsigned dreg[PER], idreg[WIDE ? 1 : PER], dupreg[WIDE ? 1 : PER];
    unsigned dmin = 0xffffffffu, dmax = 0u;
#pragma unroll
    for (int r = 0; r < PER; ++r) {
      const int i = tid + NT * r;
      // unconditional loads from a clamped index (entries beyond L are never used: every later pass tests i < L)
      if constexpr (WIDE) {
        dreg[r] = items[s + min(i, L - 1)].y;
      } else {
        const uint4 it = items[s + min(i, L - 1)];
        dreg[r] = it.y; idreg[r] = it.x; dupreg[r] = it.z;
      }
    }
#pragma unroll
    for (int r = 0; r < PER; ++r)
      if (tid + NT * r < L) { dmin = min(dmin, dreg[r]); dmax = max(dmax, dreg[r]); }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { dmin = min(dmin, (unsigned)__shfl_xor((int)dmin, o)); dmax = max(dmax, (unsigned)__shfl_xor((int)dmax, o)); }
    if (lane == 0) { bk_red[wave] = dmin; bk_red[NT / 64 + wave] = dmax; }
    if (tid < BK_BINS) bk_hist[tid] = 0u;
    __syncthreads();
    dmin = min(min(bk_red[0], bk_red[1]), min(bk_red[2], bk_red[3]));
    dmax = max(max(bk_red[4], bk_red[5]), max(bk_red[6], bk_red[7]));
    if (dmin == dmax) return false;                               // uniform across the workgroup
    // view depths are positive floats: their bit patterns order like the values, and the float arithmetic below is
    // monotonic in the depth (subtract, scale, truncate)
    const float fmin = __uint_as_float(dmin), scale = 255.5f / (__uint_as_float(dmax) - fmin);
    auto bin_of = [&](unsigned d) { return min((unsigned)((__uint_as_float(d) - fmin) * scale), (unsigned)(BK_BINS - 1)); };
#pragma unroll
    for (int r = 0; r < PER; ++r)
      if (tid + NT * r < L) atomicAdd(&bk_hist[bin_of(dreg[r])], 1u);
    __syncthreads();
    if (wave == 0) {   // exclusive scan of the 256 counts (4 per lane), largest bin
      unsigned c[4], sum = 0, big = 0;
#pragma unroll
      for (int q = 0; q < 4; ++q) { c[q] = bk_hist[lane * 4 + q]; sum += c[q]; big = max(big, c[q]); }
      unsigned incl = sum;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) { const unsigned t = (unsigned)__shfl_up((int)incl, o); if (lane >= o) incl += t; }
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) big = max(big, (unsigned)__shfl_xor((int)big, o));
      unsigned run = incl - sum;
#pragma unroll
      for (int q = 0; q < 4; ++q) { bk_start[lane * 4 + q] = run; bk_hist[lane * 4 + q] = run; run += c[q]; }   // hist -> cursors
      if (lane == 63) bk_start[BK_BINS] = run;
      if (lane == 0) bk_fallback = big > (unsigned)SEG ? 1u : 0u;
    }
    __syncthreads();
    if (bk_fallback) return false;
#pragma unroll
    for (int r = 0; r < PER; ++r) {
      const int i = tid + NT * r;
      if (i < L) {
        const unsigned pos = atomicAdd(&bk_hist[bin_of(dreg[r])], 1u);
        bk_depth[pos] = dreg[r];
        if constexpr (WIDE) bk_pos[pos] = (uint16_t)i;
        else { bk_id[pos] = idreg[r]; pl[pos] = dupreg[r]; }
      }
    }
    // greedy grouping of consecutive bins into segments of at most SEG entries: every bin finds, by binary search on the
    // prefix sums, the first bin that no longer fits a segment starting at itself; one thread then hops along
    {
      const unsigned base = bk_start[tid];          // NT == BK_BINS threads
      int lo_b = tid + 1, hi_b = BK_BINS;           // largest e in (tid, 256] with bk_start[e] - base <= SEG
      while (lo_b < hi_b) {
        const int mid = (lo_b + hi_b + 1) >> 1;
        if (bk_start[mid] - base <= (unsigned)SEG) lo_b = mid; else hi_b = mid - 1;
      }
      bk_next[tid] = (unsigned short)lo_b;
    }
    __syncthreads();
    if (tid == 0) {
      unsigned ns = 0, bb = 0;
      while (bb < (unsigned)BK_BINS && bk_start[bb] < (unsigned)L) { bk_seg_bin[ns++] = (unsigned short)bb; bb = bk_next[bb]; }
      bk_seg_bin[ns] = (unsigned short)BK_BINS;
      bk_nseg = ns;
    }
    __syncthreads();
    const unsigned nseg = bk_nseg;
    for (unsigned g = wave; g < nseg; g += NT / 64) {
      // a segment ends where the next one begins (empty bins in between belong to nobody)
      const unsigned a = bk_start[bk_seg_bin[g]], e = bk_start[bk_seg_bin[g + 1]];
      const int n = (int)(e - a);
      auto run_seg = [&](auto epl_tag) {
        constexpr int E = decltype(epl_tag)::value;
        unsigned long long key[E];
        unsigned pay[E];
#pragma unroll
        for (int r = 0; r < E; ++r) {
          const int i = r * 64 + lane;
          key[r] = ~0ull; pay[r] = 0u;
          if (i < n) {
            if constexpr (WIDE) {
              const uint4 it = items[s + bk_pos[a + i]];
              key[r] = ((unsigned long long)bk_depth[a + i] << 32) | it.x;
              pay[r] = it.z;
            } else {
              key[r] = ((unsigned long long)bk_depth[a + i] << 32) | bk_id[a + i];
              pay[r] = pl[a + i];
            }
          }
        }
        wave_bitonic_sort<E>(key, pay, lane);
#pragma unroll
        for (int r = 0; r < E; ++r) {
          const int q = lane * E + r;
          if (q < n) { sorted_id[s + a + q] = (unsigned)(key[r] & 0xffffffffull); sorted_dup[s + a + q] = pay[r]; }
        }
      };
      if (n <= 64) run_seg(std::integral_constant<int, 1>{});
      else if (n <= 128) run_seg(std::integral_constant<int, 2>{});
      else if (n <= 256) run_seg(std::integral_constant<int, 4>{});
      else run_seg(std::integral_constant<int, 8>{});
    }
    return true;
  };

  // comparators (i, i ^ mask-ish) of one network step restricted to LDS-resident positions [0, m)
  auto lds_mirror = [&](int m, int size) {   // first step of `size`: i <-> block_end - 1 - offset
    const int half = size >> 1;
    for (int c = tid; c < (m >> 1); c += NT) {
      const int blk = c / half, o = c - blk * half;
      const int i = blk * size + o, j = blk * size + size - 1 - o;
      const unsigned long long a = k[i], b = k[j];
      if (a > b) { k[i] = b; k[j] = a; const uint32_t x = pl[i]; pl[i] = pl[j]; pl[j] = x; }
    }
    __syncthreads();
  };
  auto lds_strides = [&](int m, int first_stride) {  // strides first_stride, first_stride/2, ..., 1
    for (int stride = first_stride; stride >= 1; stride >>= 1) {
      for (int c = tid; c < (m >> 1); c += NT) {
        const int i = 2 * stride * (c / stride) + (c % stride), j = i + stride;
        const unsigned long long a = k[i], b = k[j];
        if (a > b) { k[i] = b; k[j] = a; const uint32_t x = pl[i]; pl[i] = pl[j]; pl[j] = x; }
      }
      __syncthreads();
    }
  };

  // full sort of the m (power of two, <= CAP) entries in LDS: the bitonic network, one barrier per step. Only the
  // bucketed sort's fallback and the chunks of lists beyond 2 CAP entries come here (a register presort of the four
  // 1 024-entry quarters used to save 55 of the 78 steps at m = 4096, for 127 VGPRs: with the bucketed sort in front it
  // would only cost the common path its occupancy).
  auto sort_lds = [&](int m) {
    for (int size = 2; size <= m; size <<= 1) { lds_mirror(m, size); lds_strides(m, size >> 2); }
  };

  for (unsigned li = blockIdx.x; li < n_long; li += gridDim.x) {  // uniform per workgroup
    const uint2 tr = tile_range[long_tiles[li]];
    const unsigned s = tr.x;
    const long long L = (long long)tr.y;
    if (L <= lo) continue;
    __syncthreads();
    if (L <= CAP ? bucket_sort(s, (int)L, std::false_type{}) : (L <= BK_CAP && bucket_sort(s, (int)L, std::true_type{}))) continue;
    __syncthreads();
    long long n = 1;
    while (n < L) n <<= 1;
    if (n <= CAP) {  // ---- whole list in LDS -------------------------------------------------------------------
      const int m = (int)n;
      for (int i = tid; i < m; i += NT) {
        unsigned long long key = ~0ull;
        uint32_t d = 0;
        if (i < L) { const uint4 it = items[s + i]; key = ((unsigned long long)it.y << 32) | it.x; d = it.z; }
        k[i] = key; pl[i] = d;
      }
      __syncthreads();
      sort_lds(m);
      for (int i = tid; i < L; i += NT) { sorted_id[s + i] = (unsigned)(k[i] & 0xffffffffull); sorted_dup[s + i] = pl[i]; }
      continue;
    }
    // ---- hybrid: item i of the tile = 64-bit words 2i (key: id | depth << 32) and 2i + 1 (dup) ---------------------
    unsigned long long* w = reinterpret_cast<unsigned long long*>(items + s);
    auto ld = [&](long long i) { return __hip_atomic_load(&w[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    auto st = [&](long long i, unsigned long long v) { __hip_atomic_store(&w[i], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    auto chunk_load = [&](long long base) {
      for (int i = tid; i < CAP; i += NT) {
        unsigned long long key = ~0ull;
        uint32_t d = 0;
        if (base + i < L) { key = ld(2 * (base + i)); d = (uint32_t)ld(2 * (base + i) + 1); }
        k[i] = key; pl[i] = d;
      }
      __syncthreads();
    };
    auto chunk_store = [&](long long base) {
      for (int i = tid; i < CAP; i += NT)
        if (base + i < L) { st(2 * (base + i), k[i]); st(2 * (base + i) + 1, (unsigned long long)pl[i]); }
      __syncthreads();
    };
    auto global_cmpx = [&](long long i, long long j) {
      if (j >= L) return;  // +inf padding never moves down
      const unsigned long long a = ld(2 * i), b = ld(2 * j);
      if (a > b) {
        const unsigned long long pa = ld(2 * i + 1), pb = ld(2 * j + 1);
        st(2 * i, b); st(2 * j, a); st(2 * i + 1, pb); st(2 * j + 1, pa);
      }
    };
    for (long long base = 0; base < L; base += CAP) {  // every chunk fully sorted (network sizes 2 .. CAP)
      chunk_load(base);
      sort_lds(CAP);
      chunk_store(base);
    }
    for (long long size = 2 * (long long)CAP; size <= n; size <<= 1) {
      const long long half = size >> 1;
      for (long long c = tid; c < (n >> 1); c += NT) {  // mirror step of this size: always crosses chunks
        const long long blk = c / half, o = c - blk * half;
        global_cmpx(blk * size + o, blk * size + size - 1 - o);
      }
      __syncthreads();
      for (long long stride = half >> 1; stride >= CAP; stride >>= 1) {
        for (long long c = tid; c < (n >> 1); c += NT) global_cmpx(2 * stride * (c / stride) + (c % stride),
                                                                      2 * stride * (c / stride) + (c % stride) + stride);
        __syncthreads();
      }
      for (long long base = 0; base < L; base += CAP) {  // strides CAP/2 .. 1 stay inside a chunk
        chunk_load(base);
        lds_strides(CAP, CAP >> 1);
        chunk_store(base);
      }
    }
    for (long long i = tid; i < L; i += NT) {
      sorted_id[s + i] = (unsigned)(ld(2 * i) & 0xffffffffull);
      sorted_dup[s + i] = (unsigned)(ld(2 * i + 1) & 0xffffffffull);
    }
  }
}

// Longest-first tile order of the compositing kernels (SFGS_HINT_TILE_ORDER; round 6). A compositing wave owns one tile from
// its first to its last list entry, and a frame is only ~4 rounds of such waves deep: whatever is still running when the
// queue is empty runs alone. On a uniform frame that tail is one average tile; on a city seen from above -- lists of 200
// entries everywhere, 1 500 along the facades that are seen edge-on -- the wave timeline (tools/timeline.py) has composite_fwd
// drain for 70 of its 240 us, and list scheduling of the measured tile times longest first (within each XCD's share of the
// image, which keeps the chunked XCD mapping and its L2 sharing) predicts 187 us; composite_bwd 405 -> 370.
// One workgroup per XCD: a counting sort of the XCD's tiles by key (list length / last contributor) into 64 classes of 16
// entries, longest first; tiles without work last (a 4-wave workgroup of composite_bwd then holds four lists of one class
// and returns its LDS when all four are done -- in image order 28 % of that frame's tiles have nothing to blend and their
// workgroup-mates keep the LDS). Inside a class the order is whatever the LDS atomics make it: tiles are independent, any
// order composites the same bits. (A STABLE sort -- image order inside a class, in runs of 64 tiles -- was built and measured
// worse: city at 75 degrees 1.21 -> 1.10 ms instead of -> 1.04, composite_fwd 0.235 instead of 0.208 ms. And uniform frames lose
// with either -- headline +6 %, dense 8 M +12 % when forced on, almost all of it in composite_bwd: a class is a sparse subset of
// the image, the tiles in flight no longer gather the same records (FETCH_SIZE on the dense frame: composite_bwd 1.65 -> 2.27 GB
// per launch, 788 -> 1 161 us; composite_fwd 1.22 -> 2.17 GB at an unchanged 440 us) -- which is why
// the order is a HINT the caller only gives for frames whose longest list is several times their mean. Dealing the CHUNKS to the XCDs by work as well -- the XCDs still finish 4 - 8 % apart -- was built
// (tools/variants/chunk_deal_by_work_r6.patch) and bought the two kernels 1.4 / 2.4 % by rocprofv3, less than its own two extra
// passes cost: not kept. profiles/r6_tile_order_ab.txt, section 5.)
template <bool PAIRS>   // PAIRS: key = tile_range[t].y (uint2 array), else a plain uint32 array (tile_kmax)
__global__ void __launch_bounds__(1024)
tile_order_kernel(int TX8, int TY8, int SX, int SY, const void* __restrict__ keys_, uint32_t* __restrict__ order, unsigned P,
                  unsigned long long* __restrict__ hdr, unsigned bit) {
  __shared__ unsigned hist[65], base[65];
  const unsigned x = blockIdx.x;
  if (threadIdx.x < 65) hist[threadIdx.x] = 0u;
  __syncthreads();
  const unsigned SXc = (unsigned)(SX + CHUNK_BX - 1) / CHUNK_BX, SYc = (unsigned)(SY + CHUNK_BY - 1) / CHUNK_BY;
  const unsigned nch = SXc * SYc;
  constexpr unsigned PERCH = CHUNK_BX * CHUNK_BY * 4;   // tiles per chunk
  auto tile_of = [&](unsigned i, unsigned* cls) -> unsigned {   // slot i of this XCD in image (chunk) order -> tile, class
    const unsigned c = (i / PERCH) * 8u + x, within = i % PERCH;
    const unsigned st = within >> 2, sub = within & 3u;
    const unsigned tx = ((c % SXc) * CHUNK_BX + st % CHUNK_BX) * 2u + (sub & 1u);
    const unsigned ty = ((c / SXc) * CHUNK_BY + st / CHUNK_BX) * 2u + (sub >> 1);
    if (c >= nch || tx >= (unsigned)TX8 || ty >= (unsigned)TY8) { *cls = 64u; return 0xffffffffu; }
    const unsigned t = ty * (unsigned)TX8 + tx;
    const unsigned key = PAIRS ? static_cast<const uint2*>(keys_)[t].y : static_cast<const uint32_t*>(keys_)[t];
    *cls = key == 0u ? 63u : 62u - min(62u, (key - 1u) >> 4);   // class 0: more than 992 entries ... class 62: 1 .. 16; 63: none
    return t;
  };
  for (unsigned i = threadIdx.x; i < P; i += 1024) {
    unsigned cls;
    tile_of(i, &cls);
    atomicAdd(&hist[cls], 1u);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned run = 0;
    for (int k = 0; k < 65; ++k) { base[k] = run; run += hist[k]; }
  }
  __syncthreads();
  for (unsigned i = threadIdx.x; i < P; i += 1024) {
    unsigned cls;
    const unsigned t = tile_of(i, &cls);
    order[(size_t)x * P + atomicAdd(&base[cls], 1u)] = t;
  }
  if (x == 0 && threadIdx.x == 0) atomicOr(&hdr[HDR_TILE_ORDER], (unsigned long long)bit);
}



// ------------------------------------------------------------------------------------------------
// K5: compositing. Workgroup = 4 independent waves = a 2x2 block of 8x8 tiles; lane l of
// a wave owns pixel (l & 7, l >> 3) of its tile. Records of 64 list entries at a time are gathered
// (48 B each) into a wave-private LDS stage and consumed with uniform-address (broadcast) reads.
// No barriers: a wave only ever reads what it wrote itself.
// TRAIN (a backward will follow): every pixel also records WHICH entries it blended, one bit per entry, and the
// wave stores one 8-byte word per pixel and 64-entry batch (image blob, `hitmask`): the backward then walks each
// pixel's own blended entries instead of re-testing every (pixel, entry) pair of the list (raster_bwd.hip).
template <bool TRAIN>
__global__ void __launch_bounds__(64 * CWG_WAVES)
composite_fwd_kernel(KFrame kf, int TX8, int TY8, int SX, int SY, const uint2* __restrict__ tile_range,
                     const uint32_t* __restrict__ sorted_id, const float4* __restrict__ rec,
                     float* __restrict__ out_color, float* __restrict__ out_depth, float* __restrict__ out_alpha,
                     uint32_t* __restrict__ n_contrib, float* __restrict__ final_T, float* __restrict__ dacc_out,
                     uint2* __restrict__ hitmask, uint32_t* __restrict__ tile_kmax, uint16_t* __restrict__ tile_dead,
                     const unsigned long long* __restrict__ hdr, const uint32_t* __restrict__ order, unsigned order_P) {
  __shared__ float4 stage[CWG_WAVES][64 * 3];
  __shared__ __attribute__((aligned(8))) unsigned char rowlist[CWG_WAVES][4][64];
  // wave-uniform (readfirstlane / blockIdx): the tile, its list range and every loop bound below are scalars -> scalar
  // loads, SGPR loop counters and s_cbranch instead of exec-mask loops
  int tx, ty, lw;
  if (order) {   // (launch-uniform) longest list first within the XCD's share of the image: tile_order_kernel
    const unsigned ot = ordered_tile<CWG_WAVES>(order, order_P, lw);
    if (ot == 0xffffffffu) return;
    ty = (int)(ot / (unsigned)TX8); tx = (int)(ot - (unsigned)ty * (unsigned)TX8);
  } else {
    int sbx, sby, wave;
    if (!composite_wave_role(SX, SY, sbx, sby, wave, lw)) return;   // a surplus workgroup of the padded grid
    tx = sbx * 2 + (wave & 1); ty = sby * 2 + (wave >> 1);
  }
  const int lane = threadIdx.x & 63;
  if (tx >= TX8 || ty >= TY8 || ty < kf.band0 || ty >= kf.band1) return;
  const int W = kf.W, H = kf.H;
  const int px = tx * 8 + (lane & 7), py = ty * 8 + (lane >> 3);
  const bool inside = px < W && py < H;
  const size_t pix = (size_t)py * W + px;
  const int t = ty * TX8 + tx;
  const uint2 tr = tile_range[t];
  const unsigned s = tr.x, e = tr.x + tr.y;
  const float bound = __uint_as_float((unsigned)hdr[HDR_SUBPIX_BOUND]);
  float sx = (float)px, sy = (float)py;
  // (an all-zero offset tensor -- what the reference's render() passes without --ray_jitter -- is not even loaded: px + 0 = px)
  if (kf.subpix && bound != 0.f && inside) { sx += kf.subpix[pix * 2]; sy += kf.subpix[pix * 2 + 1]; }
  PixelFwd ps;
  pixel_fwd_init(ps, inside);
  float4* st = stage[lw];
  // Software pipeline over batches of 64 list entries: the (dependent) id -> record gathers of batch i+1
  // are issued before batch i is composited, so their latency hides behind ~1600 VALU instructions.
  float4 n0 = make_float4(0.f, 0.f, 0.f, 0.f), n1 = n0, n2 = n0;
  unsigned id_next = 0;
  if (s + lane < e) {
    const unsigned id = sorted_id[s + lane];
    n0 = rec[REC_F4 * (size_t)id]; n1 = rec[REC_F4 * (size_t)id + 1]; n2 = rec[REC_F4 * (size_t)id + 2];
  }
  if (s + 64 + lane < e) id_next = sorted_id[s + 64 + lane];
  for (unsigned b = s; b < e; b += 64) {
    if (__ballot(ps.T > 0.f) == 0ull) break;  // every pixel of the tile is saturated
    const unsigned cnt = min(64u, e - b);
    st[lane * 3] = n0; st[lane * 3 + 1] = n1; st[lane * 3 + 2] = n2;
    const float stage_my = n0.y, stage_ey = n2.w;
    if (b + 64 + lane < e) {  // prefetch: records of the next batch, ids of the one after
      n0 = rec[REC_F4 * (size_t)id_next]; n1 = rec[REC_F4 * (size_t)id_next + 1]; n2 = rec[REC_F4 * (size_t)id_next + 2];
    }
    if (b + 128 + lane < e) id_next = sorted_id[b + 128 + lane];
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    const unsigned k0 = b - s;
    // Strip skipping. Only ~25 % of the (pixel, splat) pairs of a tile list hit (splats of a few pixels on an 8x8
    // tile), and a wave cannot skip per lane -- but it can per strip: lanes 16r..16r+15 (one DPP row) own pixel rows
    // 2r, 2r+1, and an entry whose alpha >= 1/255 region (y-extent my +- ey, margins included) misses those two rows
    // cannot be accepted by any of their pixels. Each strip therefore walks its own compact list of the batch entries
    // that can touch it (~65 % of them); the four strips advance in lockstep and the wave leaves the batch when the
    // longest list is exhausted. Exact: the skipped pairs are pairs the per-pixel test would have rejected.
    // (Eight single-row lists need fewer steps but twice the list set-up: same time.)
    const float my = stage_my, ey = stage_ey;   // this lane's STAGED entry (entry index = lane)
    const float ylo = (float)(ty * 8) - bound, yhi = (float)(ty * 8 + 1) + bound;
    const bool live = (unsigned)lane < cnt;
    const unsigned long long b0 = __ballot(live && !(my + ey < ylo) && !(my - ey > yhi));
    const unsigned long long b1 = __ballot(live && !(my + ey < ylo + 2.f) && !(my - ey > yhi + 2.f));
    const unsigned long long b2 = __ballot(live && !(my + ey < ylo + 4.f) && !(my - ey > yhi + 4.f));
    const unsigned long long b3 = __ballot(live && !(my + ey < ylo + 6.f) && !(my - ey > yhi + 6.f));
    // per-row compact entry lists (bytes) in LDS
    unsigned char* Lw = &rowlist[lw][0][0];
    const int row = lane >> 4;
    const unsigned char* Lr = Lw + row * 64;
    if constexpr (!TRAIN) {
      // entry `lane` goes to position rank(lane) of every row it touches
      auto rank = [&](unsigned long long m) {
        return __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
      };
      if ((b0 >> lane) & 1ull) Lw[0 * 64 + rank(b0)] = (unsigned char)lane;
      if ((b1 >> lane) & 1ull) Lw[1 * 64 + rank(b1)] = (unsigned char)lane;
      if ((b2 >> lane) & 1ull) Lw[2 * 64 + rank(b2)] = (unsigned char)lane;
      if ((b3 >> lane) & 1ull) Lw[3 * 64 + rank(b3)] = (unsigned char)lane;
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      const int n0r = __popcll(b0), n1r = __popcll(b1), n2r = __popcll(b2), n3r = __popcll(b3);
      const int len = row == 0 ? n0r : row == 1 ? n1r : row == 2 ? n2r : n3r;
      const int maxlen = max(max(n0r, n1r), max(n2r, n3r));
      for (int i = 0; i < maxlen; i += 8) {   // 8 list positions per early-exit check
        const uint2 j8 = *reinterpret_cast<const uint2*>(Lr + i);   // eight byte indices at once (i % 8 == 0)
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          if (i + u < len) {
            const unsigned j = ((u < 4 ? j8.x : j8.y) >> (8 * (u & 3))) & 0xffu;
            const float4 r0 = st[j * 3], r1 = st[j * 3 + 1];
            const float2 r2 = *reinterpret_cast<const float2*>(&st[j * 3 + 2]);
            const SplatEval ev = eval_splat(r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, sx, sy);
            pixel_fwd_step(ps, ev, r2.y, r1.z, r1.w, r2.x, k0 + j);
          }
        }
        if (__ballot(ps.T > 0.f) == 0ull) break;
      }
    } else {
      // The batch is walked as two halves of 32 entries (row lists: positions 0.. for entries 0..31, positions 32..
      // for entries 32..63), so that a pixel's blended-entry bits of one half live in ONE 32-bit register and the
      // bit index is the entry index itself (v_lshl_or uses the low five bits of j).
      auto pos = [&](unsigned long long m) {
        const unsigned lo = __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u);
        const unsigned hi = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), 0u);
        return lane < 32 ? lo : 32u + hi;
      };
      if ((b0 >> lane) & 1ull) Lw[0 * 64 + pos(b0)] = (unsigned char)lane;
      if ((b1 >> lane) & 1ull) Lw[1 * 64 + pos(b1)] = (unsigned char)lane;
      if ((b2 >> lane) & 1ull) Lw[2 * 64 + pos(b2)] = (unsigned char)lane;
      if ((b3 >> lane) & 1ull) Lw[3 * 64 + pos(b3)] = (unsigned char)lane;
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      unsigned mlo = 0u, mhi = 0u;
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int n0r = __popc((unsigned)(b0 >> (32 * half))), n1r = __popc((unsigned)(b1 >> (32 * half)));
        const int n2r = __popc((unsigned)(b2 >> (32 * half))), n3r = __popc((unsigned)(b3 >> (32 * half)));
        const int len = row == 0 ? n0r : row == 1 ? n1r : row == 2 ? n2r : n3r;
        const int maxlen = max(max(n0r, n1r), max(n2r, n3r));
        unsigned& m = half ? mhi : mlo;
        bool done = false;
        for (int i = 0; i < maxlen; i += 8) {
          const uint2 j8 = *reinterpret_cast<const uint2*>(Lr + 32 * half + i);
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            if (i + u < len) {
              const unsigned j = ((u < 4 ? j8.x : j8.y) >> (8 * (u & 3))) & 0xffu;
              const float4 r0 = st[j * 3], r1 = st[j * 3 + 1];
              const float2 r2 = *reinterpret_cast<const float2*>(&st[j * 3 + 2]);
              const SplatEval ev = eval_splat(r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, sx, sy);
              pixel_fwd_step_mask(ps, ev, r2.y, r1.z, r1.w, r2.x, j, m);
            }
          }
          if (__ballot(ps.T > 0.f) == 0ull) { done = true; break; }
        }
        if (done) break;
      }
      if (mlo | mhi) ps.last = k0 + (mhi ? 63u - (unsigned)__clz(mhi) : 31u - (unsigned)__clz(mlo)) + 1u;
      hitmask[(size_t)b + lane] = make_uint2(mlo, mhi);   // b is a multiple of 64: one 512-byte store per batch
    }
    __builtin_amdgcn_wave_barrier();
  }
  const float T = pixel_fwd_final_T(ps), C0 = ps.C0, C1 = ps.C1, C2 = ps.C2, D = ps.D;
  const unsigned last = ps.last;
  if constexpr (TRAIN) {  // the tile's last contributor: where the backward starts, and how much of the list is dead
    unsigned kmax = last;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) kmax = max(kmax, (unsigned)__shfl_xor((int)kmax, d));
    if (lane == 0) {
      tile_kmax[t] = kmax;
      // (summed by dupgrad_prefill_kernel. One device atomic per tile on a frame counter instead: 32 400 atomics on one
      // address serialise at ~9 ns each -- near-camera regime: 0.13 -> 0.42 ms)
      tile_dead[t] = (uint16_t)min(e - s - kmax, 65535u);
    }
  }
  if (inside) {
    const size_t P = (size_t)W * H;
    out_color[pix] = fmaf(T, kf.bg[0], C0);
    out_color[P + pix] = fmaf(T, kf.bg[1], C1);
    out_color[2 * P + pix] = fmaf(T, kf.bg[2], C2);
    const float a = 1.0f - T;
    out_alpha[pix] = a;
    out_depth[pix] = kf.depth_mode == SFGS_DEPTH_NORMALISED ? D / a : D;
    if constexpr (TRAIN) { n_contrib[pix] = last; final_T[pix] = T; dacc_out[pix] = D; }
  }
}

}  // namespace sfgs

// =================================================================================================
// C ABI
// =================================================================================================
using namespace sfgs;

static int check_frame(const SfgsFrame* f) {
  SFGS_REQUIRE(f != nullptr, SFGS_E_ARG, "frame is NULL");
  SFGS_REQUIRE(f->struct_size == sizeof(SfgsFrame), SFGS_E_ARG, "SfgsFrame.struct_size %u != %zu (ABI mismatch)",
               f->struct_size, sizeof(SfgsFrame));
  SFGS_REQUIRE(f->image_width > 0 && f->image_height > 0, SFGS_E_ARG, "image size %dx%d", f->image_width,
               f->image_height);
  SFGS_REQUIRE(f->bg && f->viewmatrix && f->projmatrix && f->campos, SFGS_E_ARG, "frame tensor pointer is NULL");
  SFGS_REQUIRE(f->sh_degree >= 0 && f->sh_degree <= 4, SFGS_E_UNSUPPORTED, "sh_degree %d not in 0..4", f->sh_degree);
  SFGS_REQUIRE(f->tanfovx > 0.f && f->tanfovy > 0.f, SFGS_E_ARG, "tanfov must be positive");
  SFGS_REQUIRE((int64_t)f->image_width * f->image_height < (1ll << 31), SFGS_E_UNSUPPORTED, "image too large");
  return SFGS_OK;
}

// Route of the render stage's fine binning + short-list sort: the "sort" option (sfgs_set_option) forces one (tests, A/B
// runs); otherwise the caller's SHORT_LISTS hint picks the fused kernel. Both routes build bit-identical lists.
// Returns 0 (split), 512 or 1024 (the fused kernel's list capacity: "fused1024" / the MEDIUM_LISTS hint).
static int sort_fused(uint32_t launch_hints) {
  switch (option(OPT_SORT)) {
    case SORT_SPLIT: return 0;
    case SORT_FUSED: return 512;
    case SORT_FUSED1024: return 1024;
    case SORT_FUSED768: return 768;
    default: break;
  }
  if (launch_hints & SFGS_HINT_MEDIUM_LISTS) return (launch_hints & SFGS_HINT_LISTS_768) ? 768 : 1024;
  return (launch_hints & SFGS_HINT_SHORT_LISTS) ? 512 : 0;
}

static bool plan_scan_separate() { return option(OPT_PLAN_SCAN) != 0; }   // the plan's epilogues as a launch of their own (A/B, tests)

static bool binning_direct() { return option(OPT_BINNING) != 0; }

static int check_gaussians(const SfgsFrame* f, const SfgsGaussians* g) {
  SFGS_REQUIRE(g != nullptr, SFGS_E_ARG, "gaussians is NULL");
  SFGS_REQUIRE(g->struct_size == sizeof(SfgsGaussians), SFGS_E_ARG, "SfgsGaussians.struct_size mismatch");
  SFGS_REQUIRE(g->count >= 0, SFGS_E_ARG, "negative Gaussian count");
  if (g->count > 0) {
    SFGS_REQUIRE(g->means3D && g->scales && g->rotations && g->opacities, SFGS_E_ARG, "Gaussian tensor pointer is NULL");
    SFGS_REQUIRE((g->colors_precomp != nullptr) != (g->shs != nullptr), SFGS_E_ARG,
                 "provide exactly one of colors_precomp / shs");
    SFGS_REQUIRE(g->filter_3D ? (g->raw_f64_mask & ~3) == 0 : g->raw_f64_mask == 0, SFGS_E_ARG,
                 "raw_f64_mask %d: bit 0 = filter_3D is float64, bit 1 = raw opacities are float64; 0 without filter_3D",
                 g->raw_f64_mask);
    SFGS_REQUIRE(!(g->sh_dirs || g->sh_centers) || g->shs, SFGS_E_ARG, "sh_dirs / sh_centers (eval_sh-folded colour path) need shs");
    SFGS_REQUIRE(!(g->sh_dirs && g->sh_centers), SFGS_E_ARG, "sh_dirs and sh_centers are alternatives");
    SFGS_REQUIRE((g->sh_dirs || g->sh_centers) ? (g->shs_channel_major & ~1) == 0 : g->shs_channel_major == 0, SFGS_E_ARG,
                 "shs_channel_major %d: 0 or 1, and 0 without sh_dirs / sh_centers", g->shs_channel_major);
    SFGS_REQUIRE(!g->shs_rest || (g->shs && f->sh_coeffs > 1 && g->shs_channel_major == 0), SFGS_E_ARG,
                 "shs_rest (split SH storage) needs shs, sh_coeffs > 1 and coefficient-major storage");
    if (g->shs)
      SFGS_REQUIRE(f->sh_coeffs >= (f->sh_degree + 1) * (f->sh_degree + 1) &&
                       (f->sh_coeffs == 1 || f->sh_coeffs == 4 || f->sh_coeffs == 9 || f->sh_coeffs == 16 || f->sh_coeffs == 25),
                   SFGS_E_ARG, "sh_coeffs %d: must be 1, 4, 9, 16 or 25 and hold degree %d", f->sh_coeffs, f->sh_degree);
  }
  return SFGS_OK;
}

// The plan clears the head of the tiles blob (header, duplicate pools, the coarse bins' counter lines: 261 KB at 1080p).
// A kernel of our own instead of hipMemsetAsync (profiles/r4_plan_memset_ab.txt): the runtime's fill goes through its blit path,
// which showed up in the kernel traces with 4 - 6 us of idle queue in front of it on every frame.
__global__ void __launch_bounds__(256) zero_head_kernel(uint4* __restrict__ p, unsigned n16) {
  const unsigned i = blockIdx.x * 256u + threadIdx.x;
  if (i < n16) p[i] = make_uint4(0u, 0u, 0u, 0u);
}
static hipError_t zero_head(void* p, size_t bytes, hipStream_t stream) {
  const unsigned n16 = (unsigned)(bytes / 16);   // zero_bytes is a multiple of 256
  hipLaunchKernelGGL(zero_head_kernel, dim3((n16 + 255) / 256), dim3(256), 0, stream, (uint4*)p, n16);
  return hipGetLastError();
}

extern "C" int sfgs_raster_sizes(int32_t N, int32_t W, int32_t H, int64_t D, int64_t coarse_capacity,
                                 SfgsRasterSizes* out) {
  SFGS_REQUIRE(out && out->struct_size == sizeof(SfgsRasterSizes), SFGS_E_ARG, "SfgsRasterSizes.struct_size mismatch");
  SFGS_REQUIRE(N >= 0 && W > 0 && H > 0 && D >= 0 && coarse_capacity >= 0, SFGS_E_ARG,
               "bad sizes N=%d W=%d H=%d D=%lld coarse_capacity=%lld", N, W, H, (long long)D, (long long)coarse_capacity);
  SFGS_REQUIRE(D < (1ll << 32) && coarse_capacity < (1ll << 31), SFGS_E_UNSUPPORTED, "more than 2^32 duplicates");
  size_t tb = 0;
  tiles_view(nullptr, W, H, N, &tb);
  out->geom_bytes = geom_bytes(N);
  out->tiles_bytes = tb;
  out->bins_bytes = bins_bytes_plan(D, coarse_bins(W, H), coarse_capacity, N);
  out->image_bytes = image_bytes(W, H, D);
  out->dupgrad_bytes = dupgrad_bytes(D);   // D = the duplicate capacity the frame was planned with (sparse index space)
  out->coarse_bins = coarse_bins(W, H);
  return SFGS_OK;
}

extern "C" int64_t sfgs_raster_slot_capacity(int32_t W, int32_t H, int64_t num_duplicates) {
  if (W <= 0 || H <= 0 || num_duplicates < 0) return -1;
  // every coarse bin's lists take at most its tile hits + 16 x 63 alignment slots, rounded up to 64 (bin_slots_needed)
  return num_duplicates + (int64_t)(COARSE_TILES * (LIST_ALIGN - 1) + 2 * LIST_ALIGN - 1) / LIST_ALIGN * LIST_ALIGN * coarse_bins(W, H);
}

extern "C" int sfgs_raster_forward_plan(const SfgsFrame* frame, const SfgsGaussians* g, int32_t* radii, void* geom,
                                        size_t geom_sz, void* tiles, size_t tiles_sz, void* bins, size_t bins_sz,
                                        int64_t dup_capacity, int64_t coarse_capacity, void* counters_pinned_host,
                                        void* stream_) {
  if (int rc = check_frame(frame)) return rc;
  if (int rc = check_gaussians(frame, g)) return rc;
  hipStream_t stream = (hipStream_t)stream_;
  const int N = g->count, W = frame->image_width, H = frame->image_height;
  size_t tb = 0;
  SFGS_REQUIRE(tiles != nullptr, SFGS_E_ARG, "tiles blob is NULL");
  const TilesView tv = tiles_view(tiles, W, H, N, &tb);
  const int64_t NCB = coarse_bins(W, H);
  SFGS_REQUIRE(tiles_sz >= tb, SFGS_E_CAPACITY, "tiles blob: %zu bytes given, %zu needed", tiles_sz, tb);
  SFGS_REQUIRE(geom_sz >= geom_bytes(N), SFGS_E_CAPACITY, "geom blob: %zu bytes given, %zu needed", geom_sz, geom_bytes(N));
  SFGS_REQUIRE(N == 0 || (radii && geom), SFGS_E_ARG, "radii / geom is NULL");
  SFGS_REQUIRE(dup_capacity >= 0 && dup_capacity < (1ll << 32) && coarse_capacity >= 0 && coarse_capacity < (1ll << 31),
               SFGS_E_ARG, "bad dup_capacity / coarse_capacity");
  SFGS_REQUIRE(bins_sz >= bins_bytes_plan(dup_capacity, NCB, coarse_capacity, N), SFGS_E_CAPACITY,
               "bins blob: %zu bytes given, %zu needed (sfgs_raster_sizes; the plan keeps its pair list there)", bins_sz,
               bins_bytes_plan(dup_capacity, NCB, coarse_capacity, N));
  SFGS_REQUIRE(bins, SFGS_E_ARG, "bins blob is NULL");
  const GeomView gv = geom_view(geom, N);
  const BinsView bv = bins_view_csr(bins, dup_capacity, NCB, coarse_capacity, N);
  const KFrame kf = make_kframe(frame);
  const int NB = (int)pre_blocks(N);
  SFGS_CHECK_HIP(zero_head(tiles, tv.zero_bytes, stream));
  if (frame->subpixel_offset) {
    // (round 6 tried ONE launch for the clear and the reduction -- per-workgroup partial maxima that every preprocess wave
    // reduces: the launch saved is worth less than what the 31 250 waves pay for it, 88 -> 92 us for the three kernels;
    // profiles/r6_plan_head_merge_ab_not_kept.txt)
    const int64_t n = (int64_t)W * H * 2;
    const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>((n / 4 + 255) / 256, 512));
    { ProfScope ps_(KID_SUBPIX, stream);
      hipLaunchKernelGGL(subpix_bound_kernel, dim3(blocks), dim3(256), 0, stream, frame->subpixel_offset, n, tv.hdr); }
    SFGS_POST_LAUNCH("subpix_bound", stream, frame->debug);
  }
  // two-pass binning (pair list + bin_scatter_kernel) unless the coarse-bin index does not fit the pair's 16 bits (images
  // beyond 65 536 coarse bins = 8 192 x 8 192 pixels) or SFGS_BINNING=direct asks for the one-pass path (A/B, tests)
  const bool two_pass = NCB <= 65536 && !binning_direct();
  int plan_roles = 0;   // 2: the plan's epilogues ride in the scatter launch (two-pass binning)
  if (NB > 0) {
    { ProfScope ps_(KID_PREPROCESS, stream);
#define SFGS_LAUNCH_PRE_(K, D, RAW, CM)                                                                                \
  hipLaunchKernelGGL((preprocess_kernel<K, D, RAW, CM>), dim3(NB), dim3(PRE_BLOCK), 0, stream, kf, N, g->means3D,      \
                     g->scales, g->rotations, (const void*)g->opacities, g->filter_3D, (int)g->raw_f64_mask,           \
                     g->colors_precomp, g->shs, g->shs_rest, g->sh_dirs ? g->sh_dirs : g->sh_centers,                 \
                     g->sh_centers ? 1 : 0, radii, gv.rec, gv.dup, tv.coarse_count,                                    \
                     bv.slabs,                                                                                         \
                     (unsigned)coarse_capacity, (unsigned long long)dup_capacity, tv.block_nvis, tv.block_dref,        \
                     gv.big_list, tv.hdr, tv.dup_pool, two_pass ? bv.pairs : nullptr, gv.block_items)
#define SFGS_LAUNCH_PRE(K, D)                                                                                          \
  do {                                                                                                                 \
    if constexpr ((K) > 0) {                                                                                           \
      if ((g->sh_dirs || g->sh_centers) && g->shs_channel_major) { if (g->filter_3D) SFGS_LAUNCH_PRE_(K, D, true, 1); else SFGS_LAUNCH_PRE_(K, D, false, 1); break; } \
      if (g->sh_dirs || g->sh_centers) { if (g->filter_3D) SFGS_LAUNCH_PRE_(K, D, true, 2); else SFGS_LAUNCH_PRE_(K, D, false, 2); break; } \
    }                                                                                                                  \
    if (g->filter_3D) SFGS_LAUNCH_PRE_(K, D, true, 0); else SFGS_LAUNCH_PRE_(K, D, false, 0);                          \
  } while (0)
      SFGS_DISPATCH_SH(g->shs ? frame->sh_coeffs : 0, frame->sh_degree, SFGS_LAUNCH_PRE);
#undef SFGS_LAUNCH_PRE
#undef SFGS_LAUNCH_PRE_
      // the big splats' walk: persistent waves over the work list (returns at once when the list is empty; not launched
      // at all when the caller asserts there are none -- it checks counters.num_huge_splats afterwards)
      if (!(frame->launch_hints & SFGS_HINT_NO_HUGE_SPLATS))
      hipLaunchKernelGGL(big_walk_kernel, dim3(BIG_WALK_BLOCKS), dim3(256), 0, stream, kf, gv.big_list, gv.rec, gv.dup,
                         tv.coarse_count, bv.slabs, (unsigned)coarse_capacity, (unsigned long long)dup_capacity,
                         bv.big_chunks, (unsigned)big_chunk_capacity(dup_capacity), tv.hdr, tv.dup_pool,
                         dup_pools_used(NB));
    }
    SFGS_POST_LAUNCH("preprocess", stream, frame->debug);
    if (two_pass) {
      const int NWG = (int)scatter_groups(N);
      { ProfScope ps_(KID_BIN_COUNT, stream);
        hipLaunchKernelGGL(bin_count_kernel, dim3(NWG), dim3(SCATTER_NT), 0, stream, NB, (int)NCB, bv.pairs,
                           gv.block_items, tv.sc_cnt, tv.sc_hits); }
      { ProfScope ps_(KID_BIN_RANK, stream);
        hipLaunchKernelGGL(bin_rank_kernel, dim3(((int)NCB + RANK_COLS - 1) / RANK_COLS), dim3(RANK_COLS * RANK_GROUPS), 0,
                           stream, NWG, (int)NCB, tv.sc_cnt, tv.sc_hits, tv.sc_base, tv.coarse_count, tv.hdr); }
      plan_roles = plan_scan_separate() ? 0 : 2;
      { ProfScope ps_(KID_BIN_SCATTER, stream);
        hipLaunchKernelGGL(bin_scatter_kernel, dim3(NWG + plan_roles), dim3(SCATTER_NT), 0, stream, NB, (int)NCB, bv.pairs,
                           gv.block_items, tv.sc_cnt, tv.sc_base, bv.csr,
                           (unsigned)std::min<size_t>(pairs_bytes(N) / 16, 0xffffffffull), (unsigned)coarse_capacity, tv.hdr, plan_roles,
                           tv.coarse_count, tv.block_nvis, tv.block_dref, (const unsigned long long*)frame->feedback,
                           (const unsigned long long*)tv.dup_pool, (unsigned long long*)counters_pinned_host); }
    }
    SFGS_POST_LAUNCH("bin_scatter", stream, frame->debug);
  }
  if (plan_roles == 0)
  { ProfScope ps_(KID_PLAN_SCAN, stream);
    hipLaunchKernelGGL(plan_scan_kernel, dim3(2), dim3(SCAN_NT), 0, stream, (int)NCB, NB, tv.coarse_count, tv.block_nvis,
                       tv.block_dref, tv.hdr,
                       (const unsigned long long*)frame->feedback, (const unsigned long long*)tv.dup_pool,
                       (unsigned long long*)counters_pinned_host, (unsigned)coarse_capacity); }
  SFGS_POST_LAUNCH("plan_scan", stream, frame->debug);
  return SFGS_OK;
}

// ------------------------------------------------------------------------------------------------
// Joint render with the GAUSSIANS sharded over ranks: every rank plans its own Gaussians, EXPORTS the plan's
// per-Gaussian records and per-coarse-bin items in a layout that can be all-gathered (fixed strides), and MERGES the
// gathered parts into one plan whose Gaussian ids are the positions in the concatenated set -- the render stage then
// bins, sorts (by depth bits, then id: the single-process order) and composites this rank's band.
constexpr int MERGE_MAX_PARTS = 16;
struct MergeParts {
  const float4* rec[MERGE_MAX_PARTS];
  const uint32_t* count[MERGE_MAX_PARTS];
  const uint4* items[MERGE_MAX_PARTS];
  unsigned base[MERGE_MAX_PARTS + 1];   // first merged id of every part
  int parts;
};

__global__ void __launch_bounds__(256)
plan_export_kernel(int N, int NCB, const float4* __restrict__ rec, const uint32_t* __restrict__ coarse_count,
                   const uint4* __restrict__ csr, const uint4* __restrict__ slabs, unsigned coarse_capacity,
                   unsigned export_capacity, float4* __restrict__ rec_out, uint32_t* __restrict__ count_out,
                   uint4* __restrict__ items_out) {
  const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x, nthreads = (size_t)gridDim.x * 256;
  for (size_t i = t; i < (size_t)N * 3; i += nthreads) rec_out[i] = rec[(i / 3) * REC_F4 + i % 3];
  for (size_t cb = t; cb < (size_t)NCB; cb += nthreads)
    count_out[cb] = min(bin_items(coarse_count + cb * CC_STRIDE, csr, slabs, cb, coarse_capacity).n, export_capacity);
  for (size_t i = t; i < (size_t)NCB * export_capacity; i += nthreads) {
    const size_t cb = i / export_capacity, k = i % export_capacity;
    const BinItems b = bin_items(coarse_count + cb * CC_STRIDE, csr, slabs, cb, coarse_capacity);
    if (k < b.n) items_out[i] = b[(unsigned)k];
  }
}

__global__ void __launch_bounds__(256)
plan_merge_rec_kernel(MergeParts mp, float4* __restrict__ rec) {
  const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x, nthreads = (size_t)gridDim.x * 256;
  for (int p = 0; p < mp.parts; ++p) {
    const size_t n = (size_t)(mp.base[p + 1] - mp.base[p]) * 3;
    for (size_t i = t; i < n; i += nthreads) rec[((size_t)mp.base[p] + i / 3) * REC_F4 + i % 3] = mp.rec[p][i];
  }
}

// one workgroup per coarse bin: the parts' items one after the other (part order = id order), ids re-based
__global__ void __launch_bounds__(256)
plan_merge_items_kernel(MergeParts mp, unsigned export_capacity, unsigned coarse_capacity,
                        uint32_t* __restrict__ coarse_count, uint4* __restrict__ slabs, unsigned long long* __restrict__ hdr) {
  __shared__ unsigned s_hits[4];
  const size_t cb = blockIdx.x;
  unsigned off = 0, hits = 0;
  for (int p = 0; p < mp.parts; ++p) {
    const unsigned n = min(mp.count[p][cb], export_capacity);
    for (unsigned k = threadIdx.x; k < n; k += 256) {
      uint4 it = mp.items[p][cb * export_capacity + k];
      it.x += mp.base[p];
      it.z = 0u;            // duplicate indices belong to the backward: a merged plan is rendered, not differentiated
      if (off + k < coarse_capacity) slabs[cb * coarse_capacity + off + k] = it;
      hits += (unsigned)__popc(it.w);
    }
    off += n;
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) hits += (unsigned)__shfl_xor((int)hits, d);
  if ((threadIdx.x & 63) == 0) s_hits[threadIdx.x >> 6] = hits;
  __syncthreads();
  if (threadIdx.x == 0) {
    coarse_count[cb * CC_STRIDE] = off;
    coarse_count[cb * CC_STRIDE + 1] = s_hits[0] + s_hits[1] + s_hits[2] + s_hits[3];   // tile hits: list-slot need
    if (off > coarse_capacity) hdr[HDR_OVERFLOW] = 1ull;
  }
}

extern "C" int sfgs_raster_plan_export(const SfgsFrame* frame, int32_t N, const void* geom, const void* tiles,
                                       const void* bins, int64_t dup_capacity, int64_t coarse_capacity,
                                       int64_t export_capacity, float* rec_out, uint32_t* count_out, void* items_out,
                                       void* stream_) {
  if (int rc = check_frame(frame)) return rc;
  SFGS_REQUIRE(N >= 0 && geom && tiles && bins && count_out && items_out && (N == 0 || rec_out), SFGS_E_ARG, "NULL argument");
  SFGS_REQUIRE(export_capacity > 0 && export_capacity < (1ll << 31) && coarse_capacity >= 0, SFGS_E_ARG, "bad capacity");
  hipStream_t stream = (hipStream_t)stream_;
  const int W = frame->image_width, H = frame->image_height;
  const int64_t NCB = coarse_bins(W, H);
  const TilesView tv = tiles_view(const_cast<void*>(tiles), W, H, N, nullptr);
  const GeomView gv = geom_view(const_cast<void*>(geom), N);
  const BinsView bv = bins_view_csr(const_cast<void*>(bins), dup_capacity, NCB, coarse_capacity, N);
  hipLaunchKernelGGL(plan_export_kernel, dim3(2048), dim3(256), 0, stream, (int)N, (int)NCB, gv.rec, tv.coarse_count,
                     bv.csr, bv.slabs, (unsigned)coarse_capacity, (unsigned)export_capacity, (float4*)rec_out, count_out,
                     (uint4*)items_out);
  SFGS_POST_LAUNCH("plan_export", stream, frame->debug);
  return SFGS_OK;
}

extern "C" int sfgs_raster_plan_merge(const SfgsFrame* frame, int32_t parts, const int32_t* part_N,
                                      const float* const* part_rec, const uint32_t* const* part_count,
                                      const void* const* part_items, int64_t export_capacity, void* geom, size_t geom_sz,
                                      void* tiles, size_t tiles_sz, void* bins, size_t bins_sz, int64_t dup_capacity,
                                      int64_t coarse_capacity, void* stream_) {
  if (int rc = check_frame(frame)) return rc;
  SFGS_REQUIRE(parts >= 1 && parts <= MERGE_MAX_PARTS, SFGS_E_ARG, "1..%d parts", MERGE_MAX_PARTS);
  SFGS_REQUIRE(part_N && part_rec && part_count && part_items && geom && tiles && bins, SFGS_E_ARG, "NULL argument");
  SFGS_REQUIRE(export_capacity > 0 && export_capacity < (1ll << 31), SFGS_E_ARG, "bad export_capacity");
  hipStream_t stream = (hipStream_t)stream_;
  const int W = frame->image_width, H = frame->image_height;
  const int64_t NCB = coarse_bins(W, H);
  MergeParts mp;
  int64_t N = 0;
  for (int p = 0; p < parts; ++p) {
    SFGS_REQUIRE(part_N[p] >= 0 && part_count[p] && part_items[p] && (part_N[p] == 0 || part_rec[p]), SFGS_E_ARG, "bad part %d", p);
    mp.rec[p] = (const float4*)part_rec[p]; mp.count[p] = part_count[p]; mp.items[p] = (const uint4*)part_items[p];
    mp.base[p] = (unsigned)N;
    N += part_N[p];
  }
  mp.base[parts] = (unsigned)N;
  mp.parts = parts;
  SFGS_REQUIRE(N < (1ll << 31), SFGS_E_ARG, "too many Gaussians");
  size_t tb = 0;
  const TilesView tv = tiles_view(tiles, W, H, N, &tb);
  SFGS_REQUIRE(tiles_sz >= tb, SFGS_E_CAPACITY, "tiles blob: %zu bytes given, %zu needed", tiles_sz, tb);
  SFGS_REQUIRE(geom_sz >= geom_bytes(N), SFGS_E_CAPACITY, "geom blob: %zu bytes given, %zu needed", geom_sz, geom_bytes(N));
  SFGS_REQUIRE(dup_capacity >= 0 && dup_capacity < (1ll << 32) && coarse_capacity >= 0 && coarse_capacity < (1ll << 31),
               SFGS_E_ARG, "bad dup_capacity / coarse_capacity");
  // (the plan-sized blob: the render stage reads the bin-sorted items of the two-pass binning, which lie at its END)
  SFGS_REQUIRE(bins_sz >= bins_bytes_plan(dup_capacity, NCB, coarse_capacity, N), SFGS_E_CAPACITY,
               "bins blob: %zu bytes given, %zu needed (the blob the plan was given: sfgs_raster_sizes().bins_bytes)", bins_sz,
               bins_bytes_plan(dup_capacity, NCB, coarse_capacity, N));
  const GeomView gv = geom_view(geom, N);
  const BinsView bv = bins_view(bins, dup_capacity, NCB, coarse_capacity);
  SFGS_CHECK_HIP(zero_head(tiles, tv.zero_bytes, stream));
  if (frame->subpixel_offset) {
    const int64_t n = (int64_t)W * H * 2;
    const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>((n / 4 + 255) / 256, 512));
    hipLaunchKernelGGL(subpix_bound_kernel, dim3(blocks), dim3(256), 0, stream, frame->subpixel_offset, n, tv.hdr);
  }
  if (N > 0) hipLaunchKernelGGL(plan_merge_rec_kernel, dim3(2048), dim3(256), 0, stream, mp, gv.rec);
  hipLaunchKernelGGL(plan_merge_items_kernel, dim3((unsigned)NCB), dim3(256), 0, stream, mp, (unsigned)export_capacity,
                     (unsigned)coarse_capacity, tv.coarse_count, bv.slabs, tv.hdr);
  hipLaunchKernelGGL(bin_base_scan_kernel, dim3(1), dim3(1024), 0, stream, (int)NCB, tv.coarse_count, tv.hdr);
  SFGS_POST_LAUNCH("plan_merge", stream, frame->debug);
  return SFGS_OK;
}

static void unpack_counters(const unsigned long long* h, SfgsRasterCounters* out) {
  out->num_duplicates = (int64_t)h[HDR_D_EFF];
  out->num_duplicates_ref = (int64_t)h[HDR_D_REF];
  out->num_visible = (int64_t)h[HDR_N_VIS];
  out->max_tile_list = (int64_t)h[HDR_MAX_LIST];
  out->overflow = (int64_t)h[HDR_OVERFLOW];
  out->max_coarse_bin = (int64_t)h[HDR_MAX_COARSE];
  out->max_bin_items = (int64_t)h[HDR_MAX_BIN_ITEMS];
  out->num_huge_splats = (int64_t)h[HDR_BIG_COUNT];
  out->num_big_chunks = (int64_t)h[HDR_BIG_CHUNKS];
  out->prev_valid = out->prev_long_tiles = out->prev_max_tile_list = out->prev_prefilled = out->prev_tiles_over_512 = 0;
}

extern "C" int sfgs_raster_counters_decode(const void* host_128, SfgsRasterCounters* out) {
  SFGS_REQUIRE(host_128 && out, SFGS_E_ARG, "NULL argument");
  // words 0..7 = the first header words; 8..13 as plan_scan_kernel lays them out
  const unsigned long long* h = (const unsigned long long*)host_128;
  unsigned long long hdr[HDR_WORDS] = {0};
  for (int i = 0; i < 8; ++i) hdr[i] = h[i];
  hdr[HDR_BIG_COUNT] = h[8];
  hdr[HDR_BIG_CHUNKS] = h[9];
  unpack_counters(hdr, out);
  out->prev_valid = (int64_t)h[10];
  out->prev_long_tiles = (int64_t)h[11];
  out->prev_max_tile_list = (int64_t)h[12];
  out->prev_prefilled = (int64_t)h[13];
  out->prev_tiles_over_512 = (int64_t)h[14];
  out->max_bin_items = (int64_t)h[15];
  return SFGS_OK;
}

extern "C" int sfgs_raster_read_counters(const void* tiles, SfgsRasterCounters* out, void* stream_) {
  SFGS_REQUIRE(tiles && out, SFGS_E_ARG, "NULL argument");
  hipStream_t stream = (hipStream_t)stream_;
  unsigned long long h[HDR_WORDS] = {0};
  SFGS_CHECK_HIP(hipMemcpyAsync(h, tiles, 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream));
  SFGS_CHECK_HIP(hipStreamSynchronize(stream));
  unpack_counters(h, out);
  return SFGS_OK;
}

extern "C" int sfgs_raster_read_counters_pinned(const void* tiles, void* pinned_host_64, SfgsRasterCounters* out,
                                                void* stream_) {
  SFGS_REQUIRE(tiles && out && pinned_host_64, SFGS_E_ARG, "NULL argument");
  hipStream_t stream = (hipStream_t)stream_;
  SFGS_CHECK_HIP(hipMemcpyAsync(pinned_host_64, tiles, 64, hipMemcpyDeviceToHost, stream));
  SFGS_CHECK_HIP(hipStreamSynchronize(stream));
  unsigned long long hdr[HDR_WORDS] = {0};   // the first 8 header words; the optional-work counts read 0 (use
  for (int i = 0; i < 8; ++i) hdr[i] = ((const unsigned long long*)pinned_host_64)[i];   // sfgs_raster_read_counters)
  unpack_counters(hdr, out);
  return SFGS_OK;
}

constexpr int SORT_CAP = 4096;

extern "C" int sfgs_raster_forward_render(const SfgsFrame* frame, int32_t N, const void* geom, void* tiles,
                                          void* bins, size_t bins_sz, int64_t dup_capacity, int64_t coarse_capacity,
                                          int64_t num_duplicates, float* out_color, float* out_depth, float* out_alpha,
                                          void* image, size_t image_sz, void* stream_) {
  if (int rc = check_frame(frame)) return rc;
  hipStream_t stream = (hipStream_t)stream_;
  const int W = frame->image_width, H = frame->image_height;
  const int64_t NCB = coarse_bins(W, H);
  SFGS_REQUIRE(N >= 0 && tiles && out_color && out_depth && out_alpha && bins, SFGS_E_ARG, "NULL argument");
  SFGS_REQUIRE(dup_capacity >= 0 && dup_capacity < (1ll << 32) && coarse_capacity >= 0, SFGS_E_ARG, "bad capacity");
  SFGS_REQUIRE(num_duplicates < 0 || sfgs_raster_slot_capacity(W, H, num_duplicates) <= dup_capacity, SFGS_E_CAPACITY,
               "%lld duplicates need %lld list slots, dup_capacity is %lld: redo the plan with larger blobs",
               (long long)num_duplicates, (long long)sfgs_raster_slot_capacity(W, H, num_duplicates),
               (long long)dup_capacity);
  SFGS_REQUIRE(bins_sz >= bins_bytes(dup_capacity, NCB, coarse_capacity), SFGS_E_CAPACITY,
               "bins blob: %zu bytes given, %zu needed", bins_sz, bins_bytes(dup_capacity, NCB, coarse_capacity));
  SFGS_REQUIRE(image == nullptr || image_sz >= image_bytes(W, H, dup_capacity), SFGS_E_CAPACITY, "image blob too small");
  const TilesView tv = tiles_view(tiles, W, H, N, nullptr);
  const GeomView gv = geom_view(const_cast<void*>(geom), N);
  const BinsView bv = bins_view_csr(bins, dup_capacity, NCB, coarse_capacity, N);
  const KFrame kf = make_kframe(frame);
  const int TX8 = tiles8_x(W), TY8 = tiles8_y(H), T8 = TX8 * TY8, CX = coarse_x(W);
  // Two routes to the sorted per-tile lists (bit-identical results): select_sort_kernel (no per-tile items in memory, one
  // launch less: faster for lists of a few hundred entries and bins of a few thousand items -- the caller's SHORT_LISTS
  // hint) or fine_bin + the sort kernels with the items in memory between them (long lists, crowded bins)
  const int fused_cap = sort_fused(frame->launch_hints);
  const bool fused = fused_cap != 0;
  if (fused) {
    { ProfScope ps_(KID_SORT_SMALL, stream);
      if (fused_cap > 768)
        hipLaunchKernelGGL(select_sort_kernel<1024>, dim3((unsigned)NCB * COARSE), dim3(256), 0, stream, TX8, TY8, CX,
                           (int)NCB, tv.coarse_count, bv.csr, bv.slabs, (unsigned)coarse_capacity,
                           (unsigned long long)dup_capacity, tv.tile_range, bv.items, tv.long_tiles, tv.hdr, bv.sorted_id,
                           bv.sorted_dup);
      else if (fused_cap > 512)   // 36 KB of LDS per workgroup: four workgroups per CU instead of three
        hipLaunchKernelGGL(select_sort_kernel<768>, dim3((unsigned)NCB * COARSE), dim3(256), 0, stream, TX8, TY8, CX,
                           (int)NCB, tv.coarse_count, bv.csr, bv.slabs, (unsigned)coarse_capacity,
                           (unsigned long long)dup_capacity, tv.tile_range, bv.items, tv.long_tiles, tv.hdr, bv.sorted_id,
                           bv.sorted_dup);
      else
        hipLaunchKernelGGL(select_sort_kernel<512>, dim3((unsigned)NCB * COARSE), dim3(256), 0, stream, TX8, TY8, CX,
                           (int)NCB, tv.coarse_count, bv.csr, bv.slabs, (unsigned)coarse_capacity,
                           (unsigned long long)dup_capacity, tv.tile_range, bv.items, tv.long_tiles, tv.hdr, bv.sorted_id,
                           bv.sorted_dup); }
    SFGS_POST_LAUNCH("select_sort", stream, frame->debug);
  } else {
    { ProfScope ps_(KID_FINE_BIN, stream);
      hipLaunchKernelGGL(fine_bin_kernel, dim3((unsigned)NCB), dim3(256), 0, stream, TX8, TY8, CX, (int)NCB,
                         tv.coarse_count, bv.csr, bv.slabs, (unsigned)coarse_capacity, (unsigned long long)dup_capacity,
                         tv.tile_range, bv.items, tv.long_tiles, tv.hdr); }
    SFGS_POST_LAUNCH("fine_bin", stream, frame->debug);
  }
  if (num_duplicates != 0) {  // < 0: unknown (render enqueued before the counters were read)
    if (!fused) {
      { ProfScope ps_(KID_SORT_SMALL, stream);
        hipLaunchKernelGGL(sort_tiles_reg_kernel, dim3((T8 + 3) / 4), dim3(256), 0, stream, T8, tv.tile_range, bv.items,
                           bv.sorted_id, bv.sorted_dup, (const unsigned long long*)tv.hdr,
                           (unsigned long long*)frame->feedback); }
      SFGS_POST_LAUNCH("sort_tiles_small", stream, frame->debug);
    }
    const int stats_bins = fused ? (int)NCB : 0;
    if (frame->launch_hints & SFGS_HINT_FEW_LONG_LISTS) {
      // the caller expects (next to) no list beyond 512 entries: ONE catch-all launch -- the LDS kernel takes every long
      // list, whatever its size class -- instead of three that each cost ~5 us of queue time when they find nothing
      ProfScope ps_(KID_SORT_LDS, stream);
      hipLaunchKernelGGL(sort_tiles_long_kernel<SORT_CAP>, dim3(std::min(T8, 768)), dim3(256), 0, stream, REG_SORT_SMALL,
                         tv.long_tiles, tv.hdr, tv.tile_range, bv.items, bv.sorted_id, bv.sorted_dup, stats_bins,
                         tv.coarse_count, tv.hdr, (unsigned long long*)frame->feedback);
    } else {
    // size classes: 513..1024 -> register network with 16 keys per lane; longer -> bucketed sort (a 32-key register
    // network used to take 1025..2048: 195 VGPRs, one more launch, and slower than the buckets -- city 0.179 -> 0.152 ms)
    { ProfScope ps_(KID_SORT_REG_LONG, stream);
      hipLaunchKernelGGL(sort_tiles_reg_long_kernel<16>, dim3(std::min((T8 + 3) / 4, 2048)), dim3(256), 0, stream,
                         tv.long_tiles, tv.hdr, tv.tile_range, bv.items, bv.sorted_id, bv.sorted_dup); }
    SFGS_POST_LAUNCH("sort_tiles_reg_long", stream, frame->debug);
    { ProfScope ps_(KID_SORT_LDS, stream);
      hipLaunchKernelGGL(sort_tiles_long_kernel<SORT_CAP>, dim3(std::min(T8, 768)), dim3(256), 0, stream, REG_SORT_MAX,
                         tv.long_tiles, tv.hdr, tv.tile_range, bv.items, bv.sorted_id, bv.sorted_dup, stats_bins,
                         tv.coarse_count, tv.hdr, (unsigned long long*)frame->feedback); }
    }
    SFGS_POST_LAUNCH("sort_tiles_long", stream, frame->debug);
  }
  const int SX = (TX8 + 1) / 2, SY = (TY8 + 1) / 2;
  const unsigned cgrid = composite_grid(SX, SY, 4 / CWG_WAVES);
  // longest-first tile order (tile_order_kernel): asked for by the caller's hint, forced / forbidden by the "tile_order" option
  const unsigned OP = (unsigned)order_slots(W, H);
  const bool ordered = option(OPT_TILE_ORDER) == 1 || (option(OPT_TILE_ORDER) == 0 && (frame->launch_hints & SFGS_HINT_TILE_ORDER));
  const uint32_t* ord = ordered ? tv.tile_order : nullptr;
  { ProfScope ps_(KID_COMPOSITE_FWD, stream);
    if (ordered)
      hipLaunchKernelGGL(tile_order_kernel<true>, dim3(8), dim3(1024), 0, stream, TX8, TY8, SX, SY, (const void*)tv.tile_range,
                         tv.tile_order, OP, tv.hdr, 1u);
    if (image) {
      const ImageView iv = image_view(image, W, H, dup_capacity);
      hipLaunchKernelGGL(composite_fwd_kernel<true>, dim3(cgrid), dim3(64 * CWG_WAVES), 0, stream, kf, TX8, TY8, SX, SY,
                         tv.tile_range, bv.sorted_id, gv.rec, out_color, out_depth, out_alpha, iv.n_contrib, iv.final_T,
                         iv.dacc, iv.hitmask, iv.tile_kmax, iv.tile_dead, tv.hdr, ord, OP);
      if (ordered)   // the backward's order, by last contributor (known now)
        hipLaunchKernelGGL(tile_order_kernel<false>, dim3(8), dim3(1024), 0, stream, TX8, TY8, SX, SY, (const void*)iv.tile_kmax,
                           tv.tile_order + (size_t)8 * OP, OP, tv.hdr, 2u);
    } else {
      hipLaunchKernelGGL(composite_fwd_kernel<false>, dim3(cgrid), dim3(64 * CWG_WAVES), 0, stream, kf, TX8, TY8, SX, SY,
                         tv.tile_range, bv.sorted_id, gv.rec, out_color, out_depth, out_alpha, (uint32_t*)nullptr,
                         (float*)nullptr, (float*)nullptr, (uint2*)nullptr, (uint32_t*)nullptr, (uint16_t*)nullptr, tv.hdr,
                         ord, OP);
    } }
  SFGS_POST_LAUNCH("composite_fwd", stream, frame->debug);
  return SFGS_OK;
}

// ---- one scratch allocation, one call per stage (ABI 11) --------------------------------------------------------------
namespace sfgs {
int scratch_layout(int32_t N, int32_t W, int32_t H, int64_t D, int64_t ccap, bool with_image, SfgsScratchLayout* out) {
  SFGS_REQUIRE(N >= 0 && W > 0 && H > 0 && D >= 0 && ccap >= 0, SFGS_E_ARG, "bad sizes N=%d W=%d H=%d D=%lld coarse_capacity=%lld",
               N, W, H, (long long)D, (long long)ccap);
  SFGS_REQUIRE(D < (1ll << 32) && ccap < (1ll << 31), SFGS_E_UNSUPPORTED, "more than 2^32 duplicates");
  size_t tb = 0;
  tiles_view(nullptr, W, H, N, &tb);
  const size_t a = 256;
  out->geom_offset = 0;
  out->tiles_offset = align_up(std::max<size_t>(geom_bytes(N), 1), a);
  out->bins_offset = out->tiles_offset + align_up(tb, a);
  out->image_offset = out->bins_offset + align_up(std::max<size_t>(bins_bytes_plan(D, coarse_bins(W, H), ccap, N), 1), a);
  out->total_bytes = out->image_offset + (with_image ? align_up(image_bytes(W, H, D), a) : 0);
  out->dupgrad_bytes = dupgrad_bytes(D);
  out->coarse_bins = coarse_bins(W, H);
  out->slot_overhead = sfgs_raster_slot_capacity(W, H, 0);
  return SFGS_OK;
}
}  // namespace sfgs

extern "C" int sfgs_raster_scratch_layout(int32_t N, int32_t W, int32_t H, int64_t dup_capacity, int64_t coarse_capacity,
                                          int32_t with_image, SfgsScratchLayout* out) {
  SFGS_REQUIRE(out && out->struct_size == sizeof(SfgsScratchLayout), SFGS_E_ARG, "SfgsScratchLayout.struct_size mismatch");
  return scratch_layout(N, W, H, dup_capacity, coarse_capacity, with_image != 0, out);
}

extern "C" int sfgs_raster_forward(const SfgsFrame* frame, const SfgsGaussians* g, int32_t* radii, void* scratch,
                                   size_t scratch_bytes, int64_t dup_capacity, int64_t coarse_capacity, int32_t with_image,
                                   void* counters_pinned_host, void* plan_done_event, float* out_color, float* out_depth,
                                   float* out_alpha, void* stream_) {
  if (int rc = check_frame(frame)) return rc;
  SFGS_REQUIRE(g != nullptr && scratch != nullptr, SFGS_E_ARG, "NULL argument");
  SfgsScratchLayout lay;
  if (int rc = scratch_layout(g->count, frame->image_width, frame->image_height, dup_capacity, coarse_capacity,
                              with_image != 0, &lay)) return rc;
  SFGS_REQUIRE(scratch_bytes >= lay.total_bytes, SFGS_E_CAPACITY, "scratch: %zu bytes given, %zu needed", scratch_bytes,
               lay.total_bytes);
  SFGS_REQUIRE(((uintptr_t)scratch & 255u) == 0, SFGS_E_ARG, "scratch must be 256-byte aligned");
  char* base = (char*)scratch;
  const size_t geom_sz = lay.tiles_offset, tiles_sz = lay.bins_offset - lay.tiles_offset,
               bins_sz = lay.image_offset - lay.bins_offset, image_sz = lay.total_bytes - lay.image_offset;
  if (int rc = sfgs_raster_forward_plan(frame, g, radii, base + lay.geom_offset, geom_sz, base + lay.tiles_offset, tiles_sz,
                                        base + lay.bins_offset, bins_sz, dup_capacity, coarse_capacity,
                                        counters_pinned_host, stream_)) return rc;
  if (plan_done_event) SFGS_CHECK_HIP(hipEventRecord((hipEvent_t)plan_done_event, (hipStream_t)stream_));
  return sfgs_raster_forward_render(frame, g->count, base + lay.geom_offset, base + lay.tiles_offset, base + lay.bins_offset,
                                    bins_sz, dup_capacity, coarse_capacity, -1, out_color, out_depth, out_alpha,
                                    with_image ? base + lay.image_offset : nullptr, image_sz, stream_);
}
