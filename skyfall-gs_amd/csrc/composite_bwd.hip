// composite_bwd.hip -- the compositing half of the rasterizer's backward for gfx950 (MI355X): composite_bwd_kernel and its
// launcher. (The other half -- the dead-entry prefill, the per-Gaussian reduction and the 2D -> 3D chain rule -- and the C ABI
// entry points are in raster_bwd.hip.) A translation unit of its own because it is scheduled differently:
// -mllvm -amdgpu-sched-strategy=max-ilp makes this kernel 4.5 % faster (85 -> 103 VGPRs at the same LDS-bound occupancy) and
// sort_tiles three times slower, and costs preprocess_bwd a wave per SIMD (profiles/r5_sched_strategy_ab.txt); the flag is
// per file (Makefile).
//
// Replaces the autograd backward of diff_gauss.GaussianRasterizer (reference: triggered at train.py:279,845; gradient
// contract scene/gaussian_model.py:744-749).
//   composite_bwd   one wave per 8x8 tile, back to front over the tile's sorted list;
//                   the 12 per-pixel partial gradients of every (splat, tile) pair are reduced across
//                   the wave with DPP row operations and written as ONE 48-byte record per duplicate
//                   (no float atomics: the result is bit-reproducible).
//
// Compile with -ffp-contract=off (the forward's alpha/skip decisions must be reproduced exactly; FMA
// only where spelled fmaf, identically to raster_fwd.hip).
#include "sfgs_internal.h"

// Variants that were built, measured and NOT kept live as patches / A/B files, not in this source (tools/build_variant.sh
// applies tools/variants/*.patch to a scratch copy):
//   ablation bits (no phase 1 / phase 2 / stores / gathers / exp / rcp), registers instead of DPP for the upstream
//     gradients                                                  tools/variants/bwd_lab_r5.patch, profiles/r5_bwd_ablation_matrix_ab.txt
//   per-batch zero fill of UW instead of ds_wrxchg_rtn_b64        profiles/r4_bwd_ab.txt
//   three dword record stores 16 B apart instead of one dwordx3   profiles/r4_bwd_store3_ab.txt
//   three 16-byte gathers per entry instead of one 48-lane gather profiles/r4_gather48_ab.txt
//   batches of 8 entries (20 waves / CU)                          profiles/r4_bwd_batch8_ab_not_kept.txt
//   18 waves / CU on exactly 8 960 B of LDS (out-of-range dummies) profiles/r4_bwd_lds18_ab_not_kept.txt, r4_lds_probe.txt
//   packed FP32 in phase 2 (v_pk_fma_f32)                         profiles/r4_bwd_pk2_ab_not_kept.txt
//   row moments from symmetric pixel pairs, single-entry last phase-1 round, s_setprio around either phase, LDS padding
//     (occupancy sensitivity)                                     profiles/r4_bwd_trims_ab.txt, r4_bwd_ab.txt
//   1 / 2 / 8 / 16-wave workgroups                                profiles/r4_bwd_wg_waves_ab_not_kept.txt
//   phase-1 software pipelining, 3 / 4 entries per iteration      profiles/r3_bwd_p1pipe_ab_not_kept.txt, r3_bwd_phase1_entries_per_iteration_ab_not_kept.txt

namespace sfgs {

// Compositing backward. Workgroup = 4 independent waves = 2x2 tiles of 8x8 pixels (as the forward).
// The tile's list is walked back to front in batches of B entries, each batch in two phases:
//
//   phase 1 (lane = pixel): SPARSE. Only ~24 % of the (pixel, entry) pairs of a list were blended by the forward,
//            which recorded them: one bit per (pixel, entry) in the image blob's hit-mask words. Every lane walks ITS
//            OWN set bits of the batch (most significant first = back to front), reads that entry's record from the
//            LDS stage with a per-lane address, advances the pixel's transmittance / "colour behind" recurrences and
//            stores the two scalars all 12 gradients derive from -- u = G dL/dalpha and w = alpha T -- into
//            the wave-private LDS matrix UW[j][p] (row stride 65 pairs: phase 2's entry-major reads are bank-conflict free;
//            phase 1's writes are not -- every lane writes the row of ITS entry, bank = 2 (j_p + p) mod 32, ~3-way per
//            16-lane group: all of the kernel's 21 % LDS conflict cycles, profiles/r5_bwd_ablation_matrix_ab.txt; a
//            stride that fixes the writes makes the reads 16-way). The wave leaves the phase after
//            max_p popcount steps: 0.44 B on the headline scene instead of B (tools/workmodel), and no pair is
//            re-tested (no compare / select chain; the forward's decisions are replayed bit for bit).
//   phase 2 (lane = entry j, 64/B lanes per entry each owning B pixels): accumulate the 12 sums over pixels in
//            registers -- the per-(splat, tile) reduction becomes in-lane adds instead of a 12-value cross-lane
//            reduction per entry -- then combine the 64/B partial lanes and write ONE 48-byte record per duplicate.
//
// No float atomics anywhere: gradients are bit-reproducible run to run.
template <int B>
struct alignas(16) BwdLds {
  static constexpr int ROW = 65;
  // UW[j][p] = (u, w) of entry j at pixel p; row B is a dummy row (written by lanes that have no blended entry left in
  // the batch, never read). Row stride 65 pairs: the entry-major 8-byte reads of phase 2 are bank-conflict free on the
  // 64-bank LDS.
  static constexpr int REC_BYTES = 48;
  float2 UW[(B + 1) * ROW];
  float4 recs[(B + 1) * 3];   // staged records of the batch; record B is all zeros (the dummy entry: alpha = 0)
};

// value of lane I of the caller's 16-lane row, broadcast to the whole row (DPP row_newbcast; folds into
// the consuming VALU instruction). Phase 2 uses it to read per-PIXEL registers (sample position, upstream
// gradients) from per-ENTRY lanes: lane (entry, quarter q) needs pixel 16 q + I = lane I of row q.
template <int I>
__device__ __forceinline__ float row_bcast(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x150 + I, 0xf, 0xf, false));
}

// v_permlane32_swap (gfx950): a' = [a.lo, b.lo], b' = [a.hi, b.hi]  ->  a' + b' = a summed over the two half-waves in
// lanes 0..31 and b summed over them in lanes 32..63
__device__ __forceinline__ float swap32_add(float a, float b) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// v_permlane16_swap: a' = [a.r0, b.r0, a.r2, b.r2], b' = [a.r1, b.r1, a.r3, b.r3] (rows of 16 lanes)  ->  a' + b' =
// a.r0 + a.r1 | b.r0 + b.r1 | a.r2 + a.r3 | b.r2 + b.r3
__device__ __forceinline__ float swap16_add(float a, float b) {
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// SFGS_BWD_XCHG: read a (u, w) pair and leave zeros behind (ds_wrxchg_rtn_b64)
__device__ __forceinline__ float2 uw_take(const float2* p) {
  const unsigned long long v = __hip_atomic_exchange(reinterpret_cast<unsigned long long*>(const_cast<float2*>(p)), 0ull,
                                                     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
  return make_float2(__uint_as_float((unsigned)v), __uint_as_float((unsigned)(v >> 32)));
}

struct Phase2Acc {
  float u, x, y, ax, ay, xx, xy, yy, r, g, b, d;
};

template <int I>
__device__ __forceinline__ void phase2_step(Phase2Acc& a, float u, float w, float mx, float my, float cA, float cB,
                                            float cC, float sx, float sy, float g0, float g1, float g2, float g3) {
  // d = m - (lane I of this row's s): the DPP row broadcast is folded into the subtract / multiply-add
  // (hipcc keeps a separate v_mov_b32_dpp otherwise). sx, sy, g0..g3 are written once per kernel, far
  // ahead of these reads, so the VALU-write -> DPP-read wait states are trivially satisfied.
  float dx, dy;
  asm("v_subrev_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "=v"(dx) : "v"(sx), "v"(mx), "n"(I));
  asm("v_subrev_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "=v"(dy) : "v"(sy), "v"(my), "n"(I));
  const float udx = u * dx, udy = u * dy;
  a.u += u; a.x += udx; a.y += udy;
  // |u (cA dx + cB dy)| = |u| |cA dx + cB dy| accumulated with ONE multiply-add each (the |.| are free source modifiers), as the
  // grid form does: 3 instructions per sum instead of 4 (round 6)
  a.ax = fmaf(fabsf(u), fabsf(fmaf(cA, dx, cB * dy)), a.ax);
  a.ay = fmaf(fabsf(u), fabsf(fmaf(cC, dy, cB * dx)), a.ay);
  a.xx = fmaf(udx, dx, a.xx); a.xy = fmaf(udx, dy, a.xy); a.yy = fmaf(udy, dy, a.yy);
  asm("v_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(a.r) : "v"(g0), "v"(w), "n"(I));
  asm("v_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(a.g) : "v"(g1), "v"(w), "n"(I));
  asm("v_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(a.b) : "v"(g2), "v"(w), "n"(I));
  asm("v_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(a.d) : "v"(g3), "v"(w), "n"(I));
}

// Phase 2 when the sample points sit on the pixel grid (no ray jitter: subpixel_offset absent or all zero, the
// default of train.py and of every render script). In tile-centred coordinates pixel I of a lane's 2x8 group has the
// COMPILE-TIME column cx = (I & 7) - 3.5 and row r = I >> 3, so the six polynomial sums are carried as raw per-row
// moments  S_r = sum u,  X_r = sum u cx,  XX_r = sum u cx^2  (3 instructions per pixel instead of 10) and recentred
// on the splat's mean once per batch (phase2_grid_finish); the two |.| sums need the linear forms
//   lx = cA dx + cB dy = kx_r - cA cx,   ly = cC dy + cB dx = ky_r - cB cx     (dx = m_x - s_x, dy = m_y - s_y)
// which are one multiply-add each. |cx| <= 3.5, so recentring loses at most ~12 / dx^2 ulps -- far inside the
// gradient tolerance -- and nothing when the mean is far from the tile.
struct Phase2Grid {
  float S0, S1, X0, X1, XX0, XX1, ax, ay, r, g, b, d;
};

template <int I>
__device__ __forceinline__ void phase2_grid_step(Phase2Grid& a, float u, float w, float ncA, float ncB, float kx0,
                                                 float kx1, float ky0, float ky1, float g0, float g1, float g2,
                                                 float g3) {
  constexpr float cx = (float)(I & 7) - 3.5f;
  float lx, ly;
  // the first pixel of each row / of the group INITIALISES its accumulators (no zero-fill, no add)
  if constexpr (I == 0) { a.S0 = u; a.X0 = u * cx; a.XX0 = u * (cx * cx); }
  else if constexpr (I < 8) { a.S0 += u; a.X0 = fmaf(u, cx, a.X0); a.XX0 = fmaf(u, cx * cx, a.XX0); }
  else if constexpr (I == 8) { a.S1 = u; a.X1 = u * cx; a.XX1 = u * (cx * cx); }
  else { a.S1 += u; a.X1 = fmaf(u, cx, a.X1); a.XX1 = fmaf(u, cx * cx, a.XX1); }
  if constexpr (I < 8) { lx = fmaf(ncA, cx, kx0); ly = fmaf(ncB, cx, ky0); }
  else { lx = fmaf(ncA, cx, kx1); ly = fmaf(ncB, cx, ky1); }
  if constexpr (I == 0) {
    a.ax = fabsf(u) * fabsf(lx);
    a.ay = fabsf(u) * fabsf(ly);
    asm("v_mul_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "=v"(a.r) : "v"(g0), "v"(w), "n"(I));
    asm("v_mul_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "=v"(a.g) : "v"(g1), "v"(w), "n"(I));
    asm("v_mul_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "=v"(a.b) : "v"(g2), "v"(w), "n"(I));
    asm("v_mul_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "=v"(a.d) : "v"(g3), "v"(w), "n"(I));
  } else {
    a.ax = fmaf(fabsf(u), fabsf(lx), a.ax);
    a.ay = fmaf(fabsf(u), fabsf(ly), a.ay);
    asm("v_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(a.r) : "v"(g0), "v"(w), "n"(I));
    asm("v_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(a.g) : "v"(g1), "v"(w), "n"(I));
    asm("v_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(a.b) : "v"(g2), "v"(w), "n"(I));
    asm("v_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(a.d) : "v"(g3), "v"(w), "n"(I));
  }
}

// raw moments -> the sums about the mean that Phase2Acc carries (mxl = m_x - tile centre x, dy_r = m_y - y of row r)
__device__ __forceinline__ Phase2Acc phase2_grid_finish(const Phase2Grid& a, float mxl, float dy0, float dy1) {
  Phase2Acc o;
  const float Su = a.S0 + a.S1, X = a.X0 + a.X1, XX = a.XX0 + a.XX1;
  const float t0 = mxl * a.S0 - a.X0, t1 = mxl * a.S1 - a.X1;  // sum u dx of each row
  o.u = Su;
  o.x = t0 + t1;
  o.y = dy0 * a.S0 + dy1 * a.S1;
  o.xx = mxl * (o.x - X) + XX;
  o.xy = dy0 * t0 + dy1 * t1;
  o.yy = (dy0 * dy0) * a.S0 + (dy1 * dy1) * a.S1;
  o.ax = a.ax; o.ay = a.ay; o.r = a.r; o.g = a.g; o.b = a.b; o.d = a.d;
  return o;
}

// Phase 1 of one batch (see the kernel's header comment): every lane walks its own blended entries, most significant
// bit first; exhausted lanes step on the dummy entry B.
//
// One iteration = K entries per lane: their record reads, exponentials and reciprocals are independent and overlap; only
// the short transmittance / "colour behind" recurrences chain them.
// pm is LEFT-ALIGNED (bit 31 = entry B - 1): v_ffbh gives fb = B - 1 - j directly (0xffffffff for an exhausted lane,
// i.e. j = B, the dummy entry), both LDS addresses are ONE v_mad_i32_i24 of fb each and the bit is cleared with a shift
// and a v_bfi (2 instructions where xor / min / bfe took 3; round 4).
// recs_top / uw_top: LDS byte offsets (the low 32 bits of a generic LDS address) of record B - 1 and of this pixel's slot
// in row B - 1; both live in VGPRs across the loop (v_mad_i32_i24 takes one scalar operand: left to itself the compiler
// re-materialises the wave's LDS base with a v_mov in every iteration).
template <int K, int ROW, bool HAS_BG, int REC_BYTES>
__device__ __forceinline__ void phase1_iter(PixelBwd& ps, unsigned& pm, unsigned recs_top, unsigned uw_top, float sx,
                                            float sy) {
  typedef float v4f __attribute__((ext_vector_type(4)));
  static_assert(REC_BYTES % 16 == 0, "16-byte aligned staged records");
  typedef float v2f __attribute__((ext_vector_type(2)));
  typedef const __attribute__((address_space(3))) v4f* lds_c4;
  typedef const __attribute__((address_space(3))) v2f* lds_c2;
  typedef __attribute__((address_space(3))) v2f* lds_p2;
  int fb[K];
  float4 r0[K], r1[K];
  float2 r2[K];
#pragma unroll
  for (int q = 0; q < K; ++q) {
    unsigned f;
    asm("v_ffbh_u32 %0, %1" : "=v"(f) : "v"(pm));            // 0xffffffff (= -1) for pm == 0
    fb[q] = (int)f;
    pm &= ~(0x80000000u >> (f & 31u));                        // pm == 0: clears bit 0, which is never set
  }
#pragma unroll
  for (int q = 0; q < K; ++q) {
    const unsigned rp = recs_top + (unsigned)__mul24(fb[q], -REC_BYTES);
    const v4f a = *(lds_c4)(uintptr_t)rp, b = *(lds_c4)(uintptr_t)(rp + 16u);
    const v2f c = *(lds_c2)(uintptr_t)(rp + 32u);
    r0[q] = make_float4(a.x, a.y, a.z, a.w); r1[q] = make_float4(b.x, b.y, b.z, b.w); r2[q] = make_float2(c.x, c.y);
  }
  SplatEval e[K];
#pragma unroll
  for (int q = 0; q < K; ++q) e[q] = eval_splat(r0[q].x, r0[q].y, r0[q].z, r0[q].w, r1[q].x, r1[q].y, sx, sy);
  float u[K], w[K];
#pragma unroll
  for (int q = 0; q < K; ++q) pixel_bwd_scalars<HAS_BG>(ps, e[q], r2[q].y, r1[q].z, r1[q].w, r2[q].x, u[q], w[q]);
#pragma unroll
  for (int q = 0; q < K; ++q) {
    v2f uw; uw.x = u[q]; uw.y = w[q];
    *(lds_p2)(uintptr_t)(uw_top + (unsigned)__mul24(fb[q], -8 * ROW)) = uw;
  }
}

template <int B, bool HAS_BG>
__device__ __forceinline__ void phase1_walk(BwdLds<B>& lds, PixelBwd& ps, unsigned pm, float sx, float sy, int lane) {
  constexpr int ROW = BwdLds<B>::ROW;
  constexpr int K = 2;   // entries per iteration (3 and 4 measured slower: profiles/r3_bwd_phase1_entries_per_iteration_ab_not_kept.txt)
  static_assert(B == 16, "left-aligned batch masks of B bits");
  constexpr int RB = BwdLds<B>::REC_BYTES;
  unsigned recs_top = (unsigned)(uintptr_t)lds.recs + (unsigned)((B - 1) * RB);
  unsigned uw_top = (unsigned)(uintptr_t)&lds.UW[(B - 1) * ROW + lane];   // row of entry B - 1 (fb = 0)
  asm volatile("" : "+v"(recs_top), "+v"(uw_top));
  // the caller only enters with at least one blended entry in the wave (a batch without any skips the phase)
  do {
    phase1_iter<K, ROW, HAS_BG, RB>(ps, pm, recs_top, uw_top, sx, sy);
  } while (__ballot(pm != 0u) != 0ull);
}

template <int B>
__global__ void __launch_bounds__(64 * BWG_WAVES, 16 / BWG_WAVES)   // 16 waves per CU (the LDS allows no more)
composite_bwd_kernel(KFrame kf, int TX8, int TY8, int SX, int SY, const uint2* __restrict__ tile_range,
                     const uint32_t* __restrict__ sorted_id, const uint32_t* __restrict__ sorted_dup,
                     const float4* __restrict__ rec, const uint32_t* __restrict__ n_contrib,
                     const float* __restrict__ final_T, const float* __restrict__ dacc,
                     const float* __restrict__ dL_dcolor, const float* __restrict__ dL_ddepth,
                     const float* __restrict__ dL_dalpha, const uint2* __restrict__ hitmask,
                     const uint32_t* __restrict__ tile_kmax, float4* __restrict__ dupgrad, uint8_t* __restrict__ live,
                     const unsigned long long* __restrict__ hdr, int not_prefilled, const uint32_t* __restrict__ order,
                     unsigned order_P) {
  constexpr int ROW = BwdLds<B>::ROW;
  __shared__ BwdLds<B> lds_all[BWG_WAVES];
  // wave-uniform: wave index, tile, list range and all loop bounds become SGPRs (scalar loads / branches)
  int tx, ty, lw;
  if ((unsigned)hdr[HDR_TILE_ORDER] & 2u) {   // this frame's forward left a longest-first order (raster_fwd.hip: tile_order_kernel)
    const unsigned ot = ordered_tile<BWG_WAVES>(order, order_P, lw);
    if (ot == 0xffffffffu) return;
    ty = (int)(ot / (unsigned)TX8); tx = (int)(ot - (unsigned)ty * (unsigned)TX8);
  } else {
    int sbx, sby, wave;
    if (!composite_wave_role<BWG_WAVES>(SX, SY, sbx, sby, wave, lw)) return;   // a surplus workgroup of the padded grid
    constexpr int BE = composite_block_edge<BWG_WAVES>();
    tx = sbx * BE + (wave % BE); ty = sby * BE + (wave / BE);
  }
  const int lane = threadIdx.x & 63;
  if (tx >= TX8 || ty >= TY8 || ty < kf.band0 || ty >= kf.band1) return;
  BwdLds<B>& lds = lds_all[lw];
  const int W = kf.W, H = kf.H;
  const size_t P = (size_t)W * H;
  const int px = tx * 8 + (lane & 7), py = ty * 8 + (lane >> 3);
  const bool inside = px < W && py < H;
  const size_t pix = (size_t)py * W + px;
  const int t = ty * TX8 + tx;
  static_assert(B == 16, "phase 2 maps pixel groups onto DPP rows: 16 entries x 4 row pairs");
  static_assert(LIST_ALIGN == 64 && 64 % B == 0, "hit-mask words cover 64 list entries");
  static_assert(3 * B <= 64 && REC_F4 == 3, "one gather instruction per batch: 48-byte records, at most 21 entries");

  // ---- prologue: THREE dependent memory round trips per tile, every load of a round in flight together (round 6: the
  // wave timeline, tools/timeline.py, showed 16 % of all wave time in a prologue of seven serial round trips) -----------
  // round A (needs the tile's coordinates only): the tile's list range / last contributor / frame words (scalar loads) and
  // the per-pixel forward state and upstream gradients
  const uint2 tr = tile_range[t];
  const unsigned kmax = tile_kmax[t];   // max of `last` over the tile's pixels (written by the forward; scalar load)
  const unsigned prefilled = (unsigned)hdr[HDR_PREFILLED], subpix_bound = (unsigned)hdr[HDR_SUBPIX_BOUND];
  unsigned last = 0;
  float T_final = 1.f, dac = 0.f, gr = 0.f, gg = 0.f, gb = 0.f, gdep = 0.f, galp = 0.f;
  if (inside) {
    last = n_contrib[pix];
    T_final = final_T[pix];
    dac = dacc[pix];
    if (dL_dcolor) { gr = dL_dcolor[pix]; gg = dL_dcolor[P + pix]; gb = dL_dcolor[2 * P + pix]; }
    if (dL_ddepth) gdep = dL_ddepth[pix];
    if (dL_dalpha) galp = dL_dalpha[pix];
  }
  const unsigned s = tr.x, L = tr.y;
  if (L == 0) return;
  // wave-uniform: sample points on the pixel grid (max |subpixel_offset| of the plan == 0)? Then the (all-zero) offsets are
  // not even loaded: px + 0 = px
  const bool on_grid = !(kf.subpix && subpix_bound != 0u);
  float sx = (float)px, sy = (float)py;
  if (!on_grid && inside) { sx += kf.subpix[pix * 2]; sy += kf.subpix[pix * 2 + 1]; }

  // round B (needs the list range and kmax): ids / duplicate indices of the last batch, ids of the one before, the hit-mask
  // words of the last two 64-entry groups, and the duplicate indices of the first 64 DEAD entries (behind every pixel's last
  // contributor: they receive zero gradient records)
  const int ej = lane & (B - 1), grp = lane / B;  // phase-2 role of this lane
  const int orow = lane >> 4;   // which float of each record quarter this lane stores (its DPP row; = grp for B = 16)
  const int nbatch = (int)((kmax + B - 1) / B);
  // one gather INSTRUCTION per batch: lane = (entry g_rec = lane / 3, 16-byte piece g_piece = lane % 3) for lanes < 3 B, so
  // adjacent lanes fetch adjacent pieces of a record (one 48-byte request per record instead of three 16-byte ones from
  // three instructions) and the LDS stage is written with one contiguous ds_write_b128 (float4 index = lane)
  const int g_rec = lane / 3, g_piece = lane - 3 * g_rec;
  constexpr int BPG = 64 / B;   // batches per 64-entry hit-mask group
  // list entries behind every pixel's last contributor receive zero gradient records -- unless dupgrad_prefill_kernel
  // found so many of them in this frame that it switched the frame to LIVE FLAGS (sfgs_internal.h: prefill_wanted): it
  // cleared `live`, this kernel sets the byte of every record it writes, and the readers skip the others
  // (not_prefilled: the caller did not launch the prefill kernel for THIS backward -- the header word may still hold the
  // decision of an earlier backward over the same forward state, e.g. retain_graph)
  const bool zero_dead = not_prefilled || prefilled == 0u;
  if (kmax == 0u) {   // nothing blended in this tile: every entry is dead
    if (zero_dead) {
      const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
      for (unsigned k = lane; k < L; k += 64) {
        float4* dst = dupgrad + (size_t)sorted_dup[s + k] * DG_F4;
#pragma unroll
        for (int q = 0; q < DG_F4; ++q) dst[q] = zero4;
      }
    }
    return;
  }
  // Straight-line and UNCONDITIONAL (lanes without an entry re-load a valid neighbour's word: a load under a lane mask
  // whose result merges with a default value makes the compiler wait for it on the spot)
  const unsigned bl = (unsigned)(nbatch - 1) * B;   // first entry of the last batch
  const unsigned nl = kmax - bl;                     // its entry count (1 .. B)
  // the duplicate index of entry ej is needed by all four lanes (ej, row) of the entry: each stores a quarter of
  // the entry's gradient record (see the combine step)
  const unsigned id0 = sorted_id[s + bl + min((unsigned)g_rec, nl - 1u)];
  unsigned dup_cur = sorted_dup[s + bl + min((unsigned)ej, nl - 1u)];
  // (a one-batch tile has no batch before the last: the word is re-loaded, never used)
  unsigned id_next = sorted_id[s + (nbatch >= 2 ? bl - B + min((unsigned)g_rec, (unsigned)(B - 1)) : bl)];
  // hit-mask words of the tile's 64-entry groups (one uint2 per pixel and group), fetched one group ahead
  int g_cur = (nbatch - 1) / BPG;
  uint2 mw = hitmask[(size_t)s + 64u * (unsigned)g_cur + lane];
  uint2 mw_next = hitmask[(size_t)s + 64u * (unsigned)(g_cur > 0 ? g_cur - 1 : 0) + lane];
  const unsigned dead_dup = sorted_dup[s + min(kmax + lane, L - 1u)];

  // LDS set-up while the loads fly: the dummy entry (see phase 1) and the zero (u, w) matrix
  if (lane < 3) lds.recs[B * 3 + lane] = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int i = lane; i < B * ROW; i += 64) lds.UW[i] = make_float2(0.f, 0.f);

  PixelBwd ps;
  {
    const float bg[3] = {kf.bg[0], kf.bg[1], kf.bg[2]};
    pixel_bwd_init(ps, last, T_final, dac, gr, gg, gb, gdep, galp, kf.depth_mode, bg);
  }
  // wave-uniform (same for the whole frame): with a black background the bg term of dL/dalpha vanishes identically
  const bool has_bg = kf.bg[0] != 0.f || kf.bg[1] != 0.f || kf.bg[2] != 0.f;

  // round C (needs the ids): the last batch's records. Software pipeline over the batches (back to front): the dependent
  // id -> record gathers of the NEXT batch are in flight while this one is processed, the ids of the one after are fetched
  // alongside (as in the forward).
  float4 n0 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (lane < 3 * B && (unsigned)g_rec < nl) n0 = rec[REC_F4 * (size_t)id0 + g_piece];

  const float ocx = (float)(tx * 8) + 3.5f, ocy = (float)(ty * 8) + 3.5f;  // tile centre
  // A batch's three 16-byte record stores are issued at the START of the next iteration, right after that iteration's
  // prefetch loads: the s_waitcnt vmcnt(0) the compiler places at the loop's back edge (for the prefetched registers)
  // then only sees memory operations that had a whole batch of arithmetic to complete. Issued at the end of their own
  // iteration, the stores were waited for every batch (measured: the kernel had a 0.08 ms floor of pure store latency).
  float pq0 = 0.f, pq1 = 0.f, pq2 = 0.f;   // this lane's three floats of the record: floats 3 row .. 3 row + 2 (row = lane >> 4)
  unsigned p_dup = 0;
  bool p_valid = false;
  for (int bi = nbatch - 1; bi >= 0; --bi) {
    const unsigned b0 = (unsigned)bi * B;
    const unsigned cnt = min((unsigned)B, kmax - b0);
    const unsigned my_dup = dup_cur;
    if ((bi / BPG) != g_cur) {
      g_cur = bi / BPG;
      mw = mw_next;
      if (g_cur > 0) mw_next = hitmask[(size_t)s + 64u * (unsigned)(g_cur - 1) + lane];
    }
    // this pixel's blended entries of the batch, LEFT-ALIGNED (phase1_walk): bit 32 - B + j <=> entry b0 + j
    const int moff = (bi % BPG) * B;   // the batch's first bit in the group's 64-bit word (wave-uniform)
    unsigned pm = (((moff & 32) ? mw.y : mw.x) >> (moff & 31)) << (32 - B);
    if (lane < 3 * B && (unsigned)g_rec < cnt) lds.recs[lane] = n0;
    if (bi >= 1) {  // batches below the last one are always full
      if (lane < 3 * B) n0 = rec[REC_F4 * (size_t)id_next + g_piece];
      // the duplicate indices of the NEXT batch (needed only when its records are stored): loaded one batch ahead into
      // the register whose old value was copied (my_dup) at the top of this iteration. A two-deep rotation
      // (cur <- next <- load) made the compiler copy the freshly loaded value right away: an s_waitcnt vmcnt(0) directly
      // behind the record gathers, i.e. every wave sat out the full gather latency once per batch.
      dup_cur = sorted_dup[s + b0 - B + ej];
      if (bi >= 2) {
        if (lane < 3 * B) id_next = sorted_id[s + b0 - 2 * B + g_rec];
      }
    }
    if (p_valid) {  // the previous batch's gradient records
      if constexpr (DG_F4 == 4) {   // one 16-byte quarter per lane: the entry's four lanes fill a 64-byte sector
        dupgrad[(size_t)p_dup * 4 + orow] = make_float4(pq0, pq1, pq2, 0.f);
      } else {
        typedef float v3f __attribute__((ext_vector_type(3), aligned(4)));
        v3f v; v.x = pq0; v.y = pq1; v.z = pq2;
        *reinterpret_cast<v3f*>(reinterpret_cast<float*>(dupgrad) + (size_t)p_dup * 12 + 3 * orow) = v;
      }
      if (!zero_dead && orow == 0) live[p_dup] = 1;   // (wave-uniform branch; 16 byte stores per batch in live-flag frames only)
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    // ---- phase 1: lane = pixel, each lane walks its own blended entries back to front -------------------------
    // Divergence-free: a lane whose bits are exhausted keeps stepping on the DUMMY entry B (a zero record: alpha = 0,
    // so 1 / (1 - alpha) = 1 and w = 0 leave Tr untouched; the lazily applied "colour behind" update runs once and is
    // then a no-op because last_alpha becomes 0; its (u, w) goes to the dummy row). No exec masking, no state copies:
    // the loop body is one straight basic block.
    if (__ballot(pm != 0u) != 0ull) {
      if (has_bg) phase1_walk<B, true>(lds, ps, pm, sx, sy, lane);
      else phase1_walk<B, false>(lds, ps, pm, sx, sy, lane);
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    // ---- phase 2: lane = (entry ej, pixel group grp) ------------------------------------------------
    // Every lane runs the 16 steps (a DPP source lane must be active); lanes of entries beyond cnt read
    // zero U/Wm rows and their sums are discarded below.
    Phase2Acc pa;
    float cA, cB, cC;
    {
      const float4 r0 = lds.recs[ej * 3], r1 = lds.recs[ej * 3 + 1];
      const float mx = r0.x, my = r0.y;
      cA = -2.0f * LN2 * r0.z; cB = -LN2 * r0.w; cC = -2.0f * LN2 * r1.x;
      const float2* UWrow = &lds.UW[ej * ROW + grp * B];   // the lane's 16 pixels: rows 2 grp, 2 grp + 1
      const float g0 = ps.gch[0], g1 = ps.gch[1], g2 = ps.gch[2], g3 = ps.gch[3];
      // rolled loops over four 4-pixel groups (DPP controls are immediates, hence the switch): keeps the
      // compiler from hoisting all 32 LDS loads above the arithmetic, which costs ~30 VGPRs
      if (on_grid) {
        const float mxl = mx - ocx;
        const float dy0 = (my - ocy) - ((float)(grp * 2) - 3.5f), dy1 = dy0 - 1.0f;
        const float kx0 = fmaf(cA, mxl, cB * dy0), kx1 = fmaf(cA, mxl, cB * dy1);
        const float ky0 = fmaf(cC, dy0, cB * mxl), ky1 = fmaf(cC, dy1, cB * mxl);
        const float ncA = -cA, ncB = -cB;
        Phase2Grid pg;
        // straight line, four pixels per LDS round trip (the asm fences keep the compiler from hoisting all sixteen
        // 8-byte loads above the arithmetic, which would cost ~30 VGPRs and the fourth wave per SIMD)
#define SFGS_P2(I) phase2_grid_step<I>(pg, uw##I.x, uw##I.y, ncA, ncB, kx0, kx1, ky0, ky1, g0, g1, g2, g3)
#define SFGS_P2_LOAD(A, Bq, C, D) const float2 uw##A = uw_take(UWrow + A), uw##Bq = uw_take(UWrow + Bq), uw##C = uw_take(UWrow + C), uw##D = uw_take(UWrow + D);
#define SFGS_P2_DO(A, Bq, C, D) SFGS_P2(A); SFGS_P2(Bq); SFGS_P2(C); SFGS_P2(D);
#define SFGS_P2_FENCE asm volatile("" ::: "memory");
        {
          // software-pipelined by hand: the next four pairs are in flight while four are consumed (8 more live registers;
          // the kernel's occupancy is set by its LDS, 4 waves per SIMD = 128 VGPRs each). The fences pin the order: left
          // alone the compiler issues every group's loads right in front of their first use.
          SFGS_P2_LOAD(0, 1, 2, 3)
          SFGS_P2_LOAD(4, 5, 6, 7)
          SFGS_P2_FENCE
          SFGS_P2_DO(0, 1, 2, 3)
          SFGS_P2_FENCE
          SFGS_P2_LOAD(8, 9, 10, 11)
          SFGS_P2_FENCE
          SFGS_P2_DO(4, 5, 6, 7)
          SFGS_P2_FENCE
          SFGS_P2_LOAD(12, 13, 14, 15)
          SFGS_P2_FENCE
          SFGS_P2_DO(8, 9, 10, 11)
          SFGS_P2_DO(12, 13, 14, 15)
        }
#undef SFGS_P2_LOAD
#undef SFGS_P2_DO
#undef SFGS_P2_FENCE
#undef SFGS_P2
        pa = phase2_grid_finish(pg, mxl, dy0, dy1);
      } else {
        // jittered sample points (--ray_jitter): the generic form, software-pipelined like the grid form (round 6: rolled over
        // four 4-pixel groups with each (u, w) pair consumed right behind its LDS round trip, it cost +30 % of the kernel)
        pa = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#define SFGS_P2(I) phase2_step<I>(pa, uw##I.x, uw##I.y, mx, my, cA, cB, cC, sx, sy, g0, g1, g2, g3)
#define SFGS_P2_LOAD(A, Bq, C, D) const float2 uw##A = uw_take(UWrow + A), uw##Bq = uw_take(UWrow + Bq), uw##C = uw_take(UWrow + C), uw##D = uw_take(UWrow + D);
#define SFGS_P2_DO(A, Bq, C, D) SFGS_P2(A); SFGS_P2(Bq); SFGS_P2(C); SFGS_P2(D);
#define SFGS_P2_FENCE asm volatile("" ::: "memory");
        {
          SFGS_P2_LOAD(0, 1, 2, 3)
          SFGS_P2_LOAD(4, 5, 6, 7)
          SFGS_P2_FENCE
          SFGS_P2_DO(0, 1, 2, 3)
          SFGS_P2_FENCE
          SFGS_P2_LOAD(8, 9, 10, 11)
          SFGS_P2_FENCE
          SFGS_P2_DO(4, 5, 6, 7)
          SFGS_P2_FENCE
          SFGS_P2_LOAD(12, 13, 14, 15)
          SFGS_P2_FENCE
          SFGS_P2_DO(8, 9, 10, 11)
          SFGS_P2_DO(12, 13, 14, 15)
        }
#undef SFGS_P2_LOAD
#undef SFGS_P2_DO
#undef SFGS_P2_FENCE
#undef SFGS_P2
      }
    }
    // Combine the four partial lanes (ej, row 0..3) of every entry in a fixed order (deterministic). The record's 12
    // floats are linear in the sums, so every lane forms them from its PARTIAL sums first; then two rounds of the gfx950
    // half-wave / row SWAPS reduce two (then four) values per instruction pair and leave floats 3 row .. 3 row + 2 of the
    // record in lane (ej, row): 9 v_permlane*_swap + 9 adds instead of 24 ds_bpermute + 24 adds, and each lane stores its
    // three floats with one 12-byte store (the deferred store needs 3 registers instead of 12).
    // The record holds the RAW sums (GradSums order): op and the conic, which turn them into dL/dmean2D, dL/dconic ...,
    // are the same for all duplicates of a Gaussian, so preprocess_bwd applies them once to the summed record
    // (raster_math.h: grad2d_from_sums) instead of this kernel once per (Gaussian, tile) pair -- 20 instructions per batch.
    {
      const float O[12] = {pa.x, pa.y, pa.ax, pa.ay, pa.xx, pa.xy, pa.yy, pa.u, pa.r, pa.g, pa.b, pa.d};
      float q[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        // halves: lanes 0..31 get O[4k] summed over (row r, row r + 2), lanes 32..63 get O[4k+2]; likewise O[4k+1] / O[4k+3]
        // rows 0..3 end up with O[k], O[3 + k], O[6 + k], O[9 + k]: lane (ej, row) holds floats 3 row .. 3 row + 2 of the record in
        // q[0..2] -- ONE 12-byte store per lane, the entry's four lanes cover its 48 contiguous bytes with one instruction
        const float s02 = swap32_add(O[k], O[6 + k]);
        const float s13 = swap32_add(O[3 + k], O[9 + k]);
        // rows: row 0 = O[4k], row 1 = O[4k+1], row 2 = O[4k+2], row 3 = O[4k+3], each summed over the four rows
        q[k] = swap16_add(s02, s13);
      }
      pq0 = q[0]; pq1 = q[1]; pq2 = q[2]; p_dup = my_dup;
    }
    p_valid = (unsigned)ej < cnt;
    __builtin_amdgcn_wave_barrier();
  }
  if (p_valid) {
    if constexpr (DG_F4 == 4) {
      dupgrad[(size_t)p_dup * 4 + orow] = make_float4(pq0, pq1, pq2, 0.f);
    } else {
      typedef float v3f __attribute__((ext_vector_type(3), aligned(4)));
      v3f v; v.x = pq0; v.y = pq1; v.z = pq2;
      *reinterpret_cast<v3f*>(reinterpret_cast<float*>(dupgrad) + (size_t)p_dup * 12 + 3 * orow) = v;
    }
    if (!zero_dead && orow == 0) live[p_dup] = 1;
  }
  // the dead entries' zero records, last: nothing waits for these stores (their duplicate indices arrived with round B)
  if (zero_dead) {
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (kmax + lane < L) {
      float4* dst = dupgrad + (size_t)dead_dup * DG_F4;
#pragma unroll
      for (int q = 0; q < DG_F4; ++q) dst[q] = zero4;
    }
    for (unsigned k = kmax + 64 + lane; k < L; k += 64) {
      float4* dst = dupgrad + (size_t)sorted_dup[s + k] * DG_F4;
#pragma unroll
      for (int q = 0; q < DG_F4; ++q) dst[q] = zero4;
    }
  }
}

void launch_composite_bwd(unsigned grid, hipStream_t stream, KFrame kf, int TX8, int TY8, int SX, int SY,
                          const uint2* tile_range, const uint32_t* sorted_id, const uint32_t* sorted_dup, const float4* rec,
                          const uint32_t* n_contrib, const float* final_T, const float* dacc, const float* dL_dcolor,
                          const float* dL_ddepth, const float* dL_dalpha, const uint2* hitmask, const uint32_t* tile_kmax,
                          float4* dupgrad, uint8_t* live, const unsigned long long* hdr, int not_prefilled,
                          const uint32_t* order, unsigned order_slots_per_xcd) {
  hipLaunchKernelGGL(composite_bwd_kernel<16>, dim3(grid), dim3(64 * BWG_WAVES), 0, stream, kf, TX8, TY8, SX, SY, tile_range,
                     sorted_id, sorted_dup, rec, n_contrib, final_T, dacc, dL_dcolor, dL_ddepth, dL_dalpha, hitmask, tile_kmax,
                     dupgrad, live, hdr, not_prefilled, order, order_slots_per_xcd);
}

}  // namespace sfgs
