// raster_bwd.hip -- backward of the Gaussian-splat rasterizer for gfx950 (MI355X).
//
// Replaces the autograd backward of diff_gauss.GaussianRasterizer (reference: triggered at
// train.py:279,845; gradient contract scene/gaussian_model.py:744-749). Two kernels:
//
//   composite_bwd   one wave per 8x8 tile, back to front over the tile's sorted list;
//                   the 12 per-pixel partial gradients of every (splat, tile) pair are reduced across
//                   the wave with DPP row operations and written as ONE 64-byte line per duplicate
//                   (no float atomics: the result is bit-reproducible).
//   preprocess_bwd  one thread per Gaussian: sums its duplicates' lines in a fixed order and applies
//                   the 2D -> 3D chain rule (raster_math.h: preprocess_backward_one).
//
// Compile with -ffp-contract=off (the forward's alpha/skip decisions must be reproduced exactly; FMA
// only where spelled fmaf, identically to raster_fwd.hip).
#include <cstring>

#include "act_math.h"
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
  a.ax += fabsf(u * fmaf(cA, dx, cB * dy));
  a.ay += fabsf(u * fmaf(cC, dy, cB * dx));
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
composite_bwd_kernel(KFrame kf, int TX8, int TY8, int SX, int nblk, const uint2* __restrict__ tile_range,
                     const uint32_t* __restrict__ sorted_id, const uint32_t* __restrict__ sorted_dup,
                     const float4* __restrict__ rec, const uint32_t* __restrict__ n_contrib,
                     const float* __restrict__ final_T, const float* __restrict__ dacc,
                     const float* __restrict__ dL_dcolor, const float* __restrict__ dL_ddepth,
                     const float* __restrict__ dL_dalpha, const uint2* __restrict__ hitmask,
                     const uint32_t* __restrict__ tile_kmax, float4* __restrict__ dupgrad,
                     const unsigned long long* __restrict__ hdr, int not_prefilled) {
  constexpr int ROW = BwdLds<B>::ROW;
  __shared__ BwdLds<B> lds_all[BWG_WAVES];
  // wave-uniform: wave index, tile, list range and all loop bounds become SGPRs (scalar loads / branches)
  unsigned sb;
  int wave, lw;
  composite_wave_role<BWG_WAVES>((unsigned)nblk, sb, wave, lw);
  const int lane = threadIdx.x & 63;
  constexpr int BE = composite_block_edge<BWG_WAVES>();
  const int tx = (int)(sb % SX) * BE + (wave % BE), ty = (int)(sb / SX) * BE + (wave / BE);
  if (tx >= TX8 || ty >= TY8 || ty < kf.band0 || ty >= kf.band1) return;
  BwdLds<B>& lds = lds_all[lw];
  if (lane < 3) lds.recs[B * 3 + lane] = make_float4(0.f, 0.f, 0.f, 0.f);   // the dummy entry (see phase 1)
  const int W = kf.W, H = kf.H;
  const size_t P = (size_t)W * H;
  const int px = tx * 8 + (lane & 7), py = ty * 8 + (lane >> 3);
  const bool inside = px < W && py < H;
  const size_t pix = (size_t)py * W + px;
  const int t = ty * TX8 + tx;
  const uint2 tr = tile_range[t];
  const unsigned s = tr.x, e = tr.x + tr.y;
  const unsigned L = e - s;
  if (L == 0) return;

  float sx = (float)px, sy = (float)py;
  unsigned last = 0;
  PixelBwd ps;
  {
    float T_final = 1.f, dac = 0.f, gr = 0.f, gg = 0.f, gb = 0.f, gdep = 0.f, galp = 0.f;
    if (inside) {
      if (kf.subpix) { sx += kf.subpix[pix * 2]; sy += kf.subpix[pix * 2 + 1]; }
      last = n_contrib[pix];
      T_final = final_T[pix];
      dac = dacc[pix];
      if (dL_dcolor) { gr = dL_dcolor[pix]; gg = dL_dcolor[P + pix]; gb = dL_dcolor[2 * P + pix]; }
      if (dL_ddepth) gdep = dL_ddepth[pix];
      if (dL_dalpha) galp = dL_dalpha[pix];
    }
    const float bg[3] = {kf.bg[0], kf.bg[1], kf.bg[2]};
    pixel_bwd_init(ps, last, T_final, dac, gr, gg, gb, gdep, galp, kf.depth_mode, bg);
  }
  static_assert(B == 16, "phase 2 maps pixel groups onto DPP rows: 16 entries x 4 row pairs");
  // wave-uniform (same for the whole frame): with a black background the bg term of dL/dalpha vanishes identically
  const bool has_bg = kf.bg[0] != 0.f || kf.bg[1] != 0.f || kf.bg[2] != 0.f;

  const unsigned kmax = tile_kmax[t];   // max of `last` over the tile's pixels (written by the forward; scalar load)

  // list entries behind every pixel's last contributor receive zero gradient -- unless dupgrad_prefill_kernel found so
  // many of them in this frame that it zeroed the whole record array with streaming stores instead
  // (not_prefilled: the caller did not launch the prefill kernel for THIS backward -- the header word may still hold the
  // decision of an earlier backward over the same forward state, e.g. retain_graph)
  if (not_prefilled || (unsigned)hdr[HDR_PREFILLED] == 0u) {
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    for (unsigned k = kmax + lane; k < L; k += 64) {
      float4* dst = dupgrad + (size_t)sorted_dup[s + k] * DG_F4;
#pragma unroll
      for (int q = 0; q < DG_F4; ++q) dst[q] = zero4;
    }
  }
  if (kmax == 0) return;

  const int ej = lane & (B - 1), grp = lane / B;  // phase-2 role of this lane
  const int orow = lane >> 4;   // which float of each record quarter this lane stores (its DPP row; = grp for B = 16)
  // wave-uniform: sample points on the pixel grid (max |subpixel_offset| of the plan == 0)?
  const bool on_grid = !(kf.subpix && (unsigned)hdr[HDR_SUBPIX_BOUND] != 0u);
  const float ocx = (float)(tx * 8) + 3.5f, ocy = (float)(ty * 8) + 3.5f;  // tile centre
  const int nbatch = (int)((kmax + B - 1) / B);
  // Software pipeline over the batches (back to front): the dependent id -> record gathers of the NEXT batch are in
  // flight while this one is processed, the ids of the one after are fetched alongside (as in the forward).
  float4 n0 = make_float4(0.f, 0.f, 0.f, 0.f);
  [[maybe_unused]] float4 n1 = n0, n2 = n0;
  unsigned dup_cur = 0, id_next = 0;
  // one gather INSTRUCTION per batch: lane = (entry g_rec = lane / 3, 16-byte piece g_piece = lane % 3) for lanes < 3 B, so
  // adjacent lanes fetch adjacent pieces of a record (one 48-byte request per record instead of three 16-byte ones from
  // three instructions) and the LDS stage is written with one contiguous ds_write_b128 (float4 index = lane)
  static_assert(3 * B <= 64 && REC_F4 == 3, "one gather instruction per batch: 48-byte records, at most 21 entries");
  const int g_rec = lane / 3, g_piece = lane - 3 * g_rec;
  {
    const unsigned b0 = (unsigned)(nbatch - 1) * B;
    // the duplicate index of entry ej is needed by all four lanes (ej, row) of the entry: each stores a quarter of
    // the entry's gradient record (see the combine step)
    if ((unsigned)ej < kmax - b0) dup_cur = sorted_dup[s + b0 + ej];
    if (lane < 3 * B && (unsigned)g_rec < kmax - b0) {
      const unsigned id = sorted_id[s + b0 + g_rec];
      n0 = rec[REC_F4 * (size_t)id + g_piece];
    }
    if (nbatch >= 2) {
      if (lane < 3 * B) id_next = sorted_id[s + b0 - B + g_rec];
    }
  }
  // hit-mask words of the tile's 64-entry groups (one uint2 per pixel and group), fetched one group ahead
  static_assert(LIST_ALIGN == 64 && 64 % B == 0, "hit-mask words cover 64 list entries");
  constexpr int BPG = 64 / B;   // batches per 64-entry hit-mask group
  int g_cur = (nbatch - 1) / BPG;
  uint2 mw = hitmask[(size_t)s + 64u * (unsigned)g_cur + lane];
  uint2 mw_next = make_uint2(0u, 0u);
  if (g_cur > 0) mw_next = hitmask[(size_t)s + 64u * (unsigned)(g_cur - 1) + lane];
  // A batch's three 16-byte record stores are issued at the START of the next iteration, right after that iteration's
  // prefetch loads: the s_waitcnt vmcnt(0) the compiler places at the loop's back edge (for the prefetched registers)
  // then only sees memory operations that had a whole batch of arithmetic to complete. Issued at the end of their own
  // iteration, the stores were waited for every batch (measured: the kernel had a 0.08 ms floor of pure store latency).
  for (int i = lane; i < B * ROW; i += 64) lds.UW[i] = make_float2(0.f, 0.f);
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
        pa = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#define SFGS_P2(I) { const float2 t_ = uw_take(UWrow + I); phase2_step<I>(pa, t_.x, t_.y, mx, my, cA, cB, cC, sx, sy, g0, g1, g2, g3); }
#pragma nounroll
        for (int c = 0; c < 4; ++c) {
          switch (c) {
            case 0: SFGS_P2(0); SFGS_P2(1); SFGS_P2(2); SFGS_P2(3); break;
            case 1: SFGS_P2(4); SFGS_P2(5); SFGS_P2(6); SFGS_P2(7); break;
            case 2: SFGS_P2(8); SFGS_P2(9); SFGS_P2(10); SFGS_P2(11); break;
            default: SFGS_P2(12); SFGS_P2(13); SFGS_P2(14); SFGS_P2(15); break;
          }
        }
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
  }
}

// Dead list entries -- behind their tile's last contributor: opaque surfaces seen at a low angle leave a third to a
// half of every list dead, near-camera overdraw 95 % -- still own a gradient record that preprocess_bwd adds up, so it
// has to read zero. One scattered 48-byte store per dead entry costs ~38 ps (near-camera regime: 4.4 of 4.8 ms of
// composite_bwd); zeroing the WHOLE array with streaming stores costs ~10 ps per entry, dead or alive. The training
// forward leaves every tile's dead-entry count (tile_dead); this kernel sums them, fills when prefill_wanted() says
// so, and publishes the decision in hdr[HDR_PREFILLED] for composite_bwd, which then skips its zero records.
// (On the headline scene 0.1 % are dead: the kernel returns after the sum.)
__global__ void __launch_bounds__(256)
dupgrad_prefill_kernel(int T8, const uint16_t* __restrict__ tile_dead, unsigned long long n_dup, int mode,
                       float4* __restrict__ dupgrad, unsigned long long* __restrict__ hdr,
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
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  const unsigned long long R = dup_capacity / npools;   // the used part of every pool's index range (sfgs_internal.h)
  for (unsigned q = 0; q < npools; ++q) {
    const size_t n4 = (size_t)min(dup_pool[q * DP_STRIDE], R) * DG_F4;
    float4* dst = dupgrad + (size_t)q * R * DG_F4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) dst[i] = zero4;
  }
}

// Parallel pre-reduction of the records of Gaussians with more than BWD_BIG duplicates (sfgs_internal.h): one
// workgroup per 1024-record chunk sums the chunk in a fixed order and overwrites the chunk's FIRST record with the sum.
__global__ void __launch_bounds__(256)
dupgrad_reduce_kernel(const unsigned long long* __restrict__ hdr, const uint2* __restrict__ big_chunks,
                      unsigned chunk_cap, const uint2* __restrict__ dup, float4* __restrict__ dupgrad) {
  constexpr int NF = DUPGRAD_FLOATS;   // every float of the record is summed in place (padding floats are zeros)
  __shared__ float part[4][NF];
  const unsigned n_chunks = (unsigned)min(hdr[HDR_BIG_CHUNKS], (unsigned long long)chunk_cap);
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
    for (unsigned i = threadIdx.x; i < n; i += 256) {
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
                      const uint2* __restrict__ dup, const float4* __restrict__ dupgrad,
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
#pragma unroll
      for (int k = 0; k < NLD; ++k) tmp[k] = dupgrad[(size_t)c0 * DG_F4 + min((unsigned)(k * 64 + lane), n4 - 1u)];
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
    for (unsigned i = lane; i < bn; i += 64) w.add(dupgrad + (size_t)(b0 + i) * DG_F4);
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
  const int SX = (TX8 + BE - 1) / BE, SY = (TY8 + BE - 1) / BE, nblk = SX * SY;
  { ProfScope ps_(KID_COMPOSITE_BWD, stream);
    // (without the prefill kernel its decision word keeps the plan's zero: composite_bwd writes the zero records itself)
    if (!(frame->launch_hints & SFGS_HINT_NO_PREFILL))
    hipLaunchKernelGGL(dupgrad_prefill_kernel, dim3(512), dim3(256), 0, stream, TX8 * TY8, iv.tile_dead,
                       (unsigned long long)num_duplicates, prefill_mode(), (float4*)dupgrad, tv.hdr,
                       (unsigned long long*)frame->feedback, (const unsigned long long*)tv.dup_pool,
                       (unsigned long long)dup_capacity, dup_pools_used(pre_blocks(N)));
    hipLaunchKernelGGL(composite_bwd_kernel<16>, dim3(nblk * (BE * BE / BWG_WAVES)), dim3(64 * BWG_WAVES), 0, stream, kf, TX8, TY8, SX,
                       nblk, tv.tile_range, bv.sorted_id, bv.sorted_dup, gv.rec, iv.n_contrib, iv.final_T, iv.dacc, dL_dcolor, dL_ddepth,
                       dL_dalpha, iv.hitmask, iv.tile_kmax, (float4*)dupgrad, tv.hdr,
                       (frame->launch_hints & SFGS_HINT_NO_PREFILL) ? 1 : 0);
  }
  SFGS_POST_LAUNCH("composite_bwd", stream, frame->debug);
  const int NB = (int)pre_blocks(N);
  { ProfScope ps_(KID_PREPROCESS_BWD, stream);
    if (!(frame->launch_hints & SFGS_HINT_NO_BIG_CHUNKS))   // the caller read num_big_chunks == 0 from this frame's plan
    hipLaunchKernelGGL(dupgrad_reduce_kernel, dim3(1024), dim3(256), 0, stream, tv.hdr, bv.big_chunks,
                       (unsigned)big_chunk_capacity(dup_capacity), gv.dup, (float4*)dupgrad);
#define SFGS_LAUNCH_PBWD_(K, D, RAW, CM)                                                                               \
  hipLaunchKernelGGL((preprocess_bwd_kernel<K, D, RAW, CM>), dim3(NB), dim3(PRE_BLOCK), 0, stream, kf, N, g->means3D,  \
                     g->scales, g->rotations, (const void*)g->opacities, g->filter_3D, (int)g->raw_f64_mask, g->shs,   \
                     g->shs_rest, g->sh_dirs ? g->sh_dirs : g->sh_centers, g->sh_centers ? 1 : 0, radii, gv.dup,       \
                     (const float4*)dupgrad, grads->means3D, grads->means2D,                                           \
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
