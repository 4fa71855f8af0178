// workmodel.cpp -- CPU work model of the compositing kernels (design tool, not product, not oracle).
// Builds the product's per-8x8-tile lists with the product's own math header (raster_math.h compiled by g++),
// composites a window of tiles and counts, for candidate wave schedules of the backward, how many wave steps
// each would take. Used to size design decisions before spending GPU time.
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../skyfall-gs_amd/csrc/raster_math.h"

using namespace sfgs;

extern "C" {
struct WmFrame {
  int32_t W, H;
  float tanfovx, tanfovy, kernel_size, scale_modifier;
  const float* bg; const float* view; const float* proj; const float* campos;
};

// window of tiles [tx0,tx1) x [ty0,ty1); out[] receives the statistics (see tools/workmodel/run.py)
int wm_run(const WmFrame* hf, int32_t N, const float* means, const float* scales, const float* rots, const float* opac,
           int tx0, int tx1, int ty0, int ty1, double* out, int n_out) {
  FrameParams f;
  memset(&f, 0, sizeof(f));
  f.W = hf->W; f.H = hf->H; f.tanfovx = hf->tanfovx; f.tanfovy = hf->tanfovy;
  f.kernel_size = hf->kernel_size; f.scale_modifier = hf->scale_modifier;
  for (int i = 0; i < 16; ++i) { f.view[i] = hf->view[i]; f.proj[i] = hf->proj[i]; }
  for (int i = 0; i < 3; ++i) { f.campos[i] = hf->campos[i]; f.bg[i] = hf->bg[i]; }
  const int W = f.W, H = f.H;
  const int TX8 = (W + 7) / 8;
  const int wx = tx1 - tx0, wy = ty1 - ty0;
  std::vector<std::vector<unsigned long long>> lists((size_t)wx * wy);
  std::vector<SplatRec> recs;
  std::vector<int> dup_per_gauss;
  long long D_all = 0;
  // preprocess's inline tile walk (one thread per Gaussian, the wave runs max over its 64 lanes of the walked tiles):
  // wave steps now, and what a walk balanced over the wave (lane = one (Gaussian, tile) task) would take
  double walk_waves = 0, walk_steps_max = 0, walk_steps_balanced = 0, walk_tiles = 0, walk_hits = 0, walk_rows_max = 0, walk_waves_over32 = 0, walk_steps_hybrid = 0;
  int wave_max = 0, wave_sum = 0, wave_rows_max = 0;
  for (int g = 0; g < N; ++g) {
    if (g % 64 == 0 && g) {
      walk_waves += 1; walk_steps_max += wave_max; walk_steps_balanced += (wave_sum + 63) / 64; walk_rows_max += wave_rows_max;
      if (wave_max > 32) { walk_waves_over32 += 1; walk_steps_hybrid += wave_max; } else walk_steps_hybrid += (wave_sum + 63) / 64;
      wave_max = wave_sum = wave_rows_max = 0;
    }
    const Projected pr = project_gaussian(f, means + 3 * (size_t)g, scales + 3 * (size_t)g, rots + 4 * (size_t)g);
    if (!pr.visible) continue;
    const float rgb[3] = {0.5f, 0.5f, 0.5f};
    SplatRec r = make_record(pr, opac[g], rgb);
    const BinRange br = bin_range(r, W, H, pr.rminx, pr.rminy, pr.rmaxx, pr.rmaxy, 0.f);
    const float thr = alpha_threshold_log2(r.op);
    {
      const int wt = (br.x1 - br.x0) * (br.y1 - br.y0);
      const int ncb = wt > 0 ? ((br.x1 - 1) / 4 - br.x0 / 4 + 1) * ((br.y1 - 1) / 4 - br.y0 / 4 + 1) : 0;
      if (wt > 0 && ncb <= 6) {   // BIG_WALK: larger walks are cooperative already
        wave_max = wt > wave_max ? wt : wave_max; wave_sum += wt; walk_tiles += wt;
        wave_rows_max = (br.y1 - br.y0) > wave_rows_max ? (br.y1 - br.y0) : wave_rows_max;
      }
    }
    uint32_t depth_bits; memcpy(&depth_bits, &r.depth, 4);
    int idx = -1;
    for (int ty = br.y0; ty < br.y1; ++ty)
      for (int tx = br.x0; tx < br.x1; ++tx)
        if (bin_test(r, thr, tx, ty, W, H, 0.f)) {
          ++D_all;
          if (tx >= tx0 && tx < tx1 && ty >= ty0 && ty < ty1) {
            if (idx < 0) { idx = (int)recs.size(); recs.push_back(r); }
            lists[(size_t)(ty - ty0) * wx + (tx - tx0)].push_back(((unsigned long long)depth_bits << 32) | (unsigned)idx);
          }
        }
  }
  (void)TX8; (void)walk_hits;
  // statistics
  double n_tiles = 0, sumL = 0, sumK = 0, hits = 0, dense16 = 0;
  double sp16 = 0, sp32 = 0, sp64 = 0, spInf = 0;        // per-lane sparse phase-1 steps with batch B
  double q16 = 0, q64 = 0;                                // quadrant lists (4x4), lockstep max over quadrants
  double strip16 = 0;                                     // row-pair strips (8x2)
  double dead_entries = 0, live_entries = 0, behind = 0;  // entries < kmax with no accepted pixel; entries >= kmax
  double task_nonzero = 0, task_total = 0;                // (entry, row-pair) tasks of phase 2
  double hist_hits[65] = {0};
  double live16_batches = 0, all16_batches = 0;           // batches (of 16) after dropping dead entries
  double fstrip64[4] = {0, 0, 0, 0}, fstrip32[4] = {0, 0, 0, 0};   // forward strip-list steps (see below)
  // round 5:
  double fwd_steps = 0, fwd_steps_no_accept = 0;   // strip-list steps of the forward (32-entry halves, bbox lists) / those in
                                                   // which NO lane of the wave blends (a wave-uniform early-out could skip the blend)
  double row_task_nonzero = 0, row_task_total = 0; // (entry, single pixel row) tasks of phase 2 with at least one blended pixel
  double p2_groups_zero = 0, p2_groups = 0;        // phase 2's 4-pixel groups (per batch of 16: 4 groups x 64 lanes) that are zero in EVERY lane
  double sp16_ahead = 0;                           // phase-1 steps with batches of 16 when an idle lane may take ONE hit of the next batch early
  for (int ty = 0; ty < wy; ++ty)
    for (int tx = 0; tx < wx; ++tx) {
      auto& l = lists[(size_t)ty * wx + tx];
      std::sort(l.begin(), l.end());
      const int L = (int)l.size();
      if (!L) continue;
      n_tiles += 1; sumL += L;
      std::vector<unsigned long long> emask(L, 0ull);  // per entry: accepted-pixel mask
      int kmax = 0;
      for (int p = 0; p < 64; ++p) {
        const int px = (tx0 + tx) * 8 + (p & 7), py = (ty0 + ty) * 8 + (p >> 3);
        if (px >= W || py >= H) continue;
        PixelFwd ps; pixel_fwd_init(ps, true);
        for (int k = 0; k < L; ++k) {
          const SplatRec& r = recs[(unsigned)(l[k] & 0xffffffffull)];
          const SplatEval ev = eval_splat(r.mx, r.my, r.qa, r.qb, r.qc, r.op, (float)px, (float)py);
          const unsigned before = ps.last;
          pixel_fwd_step(ps, ev, r.depth, r.r, r.g, r.b, (unsigned)k);
          if (ps.last != before) emask[k] |= 1ull << p;
          if (!(ps.T > 0.f)) break;
        }
        kmax = std::max(kmax, (int)ps.last);
      }
      sumK += kmax; behind += L - kmax;
      dense16 += (kmax + 15) / 16 * 16;
      int pixhits[64] = {0};
      int nlive = 0;
      for (int k = 0; k < kmax; ++k) {
        const int c = __builtin_popcountll(emask[k]);
        hits += c; hist_hits[c] += 1;
        if (c) { ++live_entries; ++nlive; } else ++dead_entries;
        for (int rp = 0; rp < 4; ++rp) { task_total += 1; if ((emask[k] >> (16 * rp)) & 0xffffull) task_nonzero += 1; }
        for (int p = 0; p < 64; ++p) pixhits[p] += (emask[k] >> p) & 1;
      }
      all16_batches += (kmax + 15) / 16; live16_batches += (nlive + 15) / 16;
      for (int k = 0; k < kmax; ++k)
        for (int r = 0; r < 8; ++r) { row_task_total += 1; if ((emask[k] >> (8 * r)) & 0xffull) row_task_nonzero += 1; }
      // phase 2: lane (entry e, quarter q) of a 16-entry batch handles pixels 16 q + I; group g = steps I in [4g, 4g+4)
      for (int b0 = 0; b0 < kmax; b0 += 16)
        for (int grp = 0; grp < 4; ++grp) {
          bool any = false;
          for (int k = b0; k < std::min(kmax, b0 + 16) && !any; ++k)
            for (int q = 0; q < 4 && !any; ++q)
              if ((emask[k] >> (16 * q + 4 * grp)) & 0xfull) any = true;
          p2_groups += 1; if (!any) p2_groups_zero += 1;
        }
      {   // one-hit-ahead phase 1: walk the batches back to front like the kernel does
        int ahead[64] = {0};
        for (int b0 = ((kmax - 1) / 16) * 16; b0 >= 0; b0 -= 16) {
          int cnt[64] = {0}, nxt[64] = {0};
          for (int k = b0; k < std::min(kmax, b0 + 16); ++k) for (int p = 0; p < 64; ++p) cnt[p] += (emask[k] >> p) & 1;
          if (b0 >= 16) for (int k = b0 - 16; k < b0; ++k) for (int p = 0; p < 64; ++p) nxt[p] += (emask[k] >> p) & 1;
          int steps = 0;
          for (int p = 0; p < 64; ++p) steps = std::max(steps, cnt[p] - ahead[p]);
          sp16_ahead += steps;
          for (int p = 0; p < 64; ++p) { const int mine = cnt[p] - ahead[p]; ahead[p] = (mine < steps && nxt[p] > 0) ? 1 : 0; }
        }
      }
      spInf += *std::max_element(pixhits, pixhits + 64);
      auto sparse = [&](int B) {
        double steps = 0;
        for (int b0 = 0; b0 < kmax; b0 += B) {
          int cnt[64] = {0};
          for (int k = b0; k < std::min(kmax, b0 + B); ++k) for (int p = 0; p < 64; ++p) cnt[p] += (emask[k] >> p) & 1;
          steps += *std::max_element(cnt, cnt + 64);
        }
        return steps;
      };
      sp16 += sparse(16); sp32 += sparse(32); sp64 += sparse(64);
      auto quad = [&](int B, bool strips) {
        double steps = 0;
        for (int b0 = 0; b0 < kmax; b0 += B) {
          int cnt[4] = {0, 0, 0, 0};
          for (int k = b0; k < std::min(kmax, b0 + B); ++k)
            for (int q = 0; q < 4; ++q) {
              unsigned long long qm = 0;
              if (strips) qm = 0xffffull << (16 * q);
              else for (int y = 0; y < 4; ++y) qm |= 0xfull << (((q >> 1) * 4 + y) * 8 + (q & 1) * 4);
              if (emask[k] & qm) ++cnt[q];
            }
          steps += *std::max_element(cnt, cnt + 4);
        }
        return steps;
      };
      q16 += quad(16, false); q64 += quad(64, false); strip16 += quad(16, true);
      // FORWARD strip lists (batches of B entries over the whole list walked, i.e. up to kmax; four 8x2 strips in
      // lockstep): `bbox` = the product's test (the entry's alpha >= 1/255 y-extent reaches the strip's two rows),
      // `xy` = y-extent AND x-extent against the tile's 8 columns, `exact` = the alpha >= 1/255 ellipse reaches the
      // strip's rectangle (the binning's own test on a 8x2 rectangle), `ideal` = at least one pixel of the strip blended it
      auto reach = [&](const SplatRec& r, float x0, float x1, float y0, float y1) {   // ellipse vs pixel-centre rectangle
        const float thr = alpha_threshold_log2(r.op);
        // closest point of the rectangle to the centre in the metric of the conic: minimise over the rectangle by
        // coordinate descent on the convex quadratic (exact for axis-aligned boxes within a few rounds)
        float x = std::min(std::max(r.mx, x0), x1), y = std::min(std::max(r.my, y0), y1);
        for (int it = 0; it < 8; ++it) {
          // p2(dx, dy) = qa dx^2 + qb dx dy + qc dy^2 (negative definite): maximise
          const float dy = y - r.my;
          float dx = -r.qb * dy / (2.f * r.qa);
          x = std::min(std::max(r.mx + dx, x0), x1);
          dx = x - r.mx;
          float dyo = -r.qb * dx / (2.f * r.qc);
          y = std::min(std::max(r.my + dyo, y0), y1);
        }
        const float dx = x - r.mx, dy = y - r.my;
        return r.qa * dx * dx + r.qb * dx * dy + r.qc * dy * dy >= thr;
      };
      auto fwd = [&](int B, int mode) {
        double steps = 0;
        const float X0 = (float)((tx0 + tx) * 8), Y0 = (float)((ty0 + ty) * 8);
        for (int b0 = 0; b0 < kmax; b0 += B) {
          int cnt[4] = {0, 0, 0, 0};
          for (int k = b0; k < std::min(kmax, b0 + B); ++k) {
            const SplatRec& r = recs[(unsigned)(l[k] & 0xffffffffull)];
            for (int q = 0; q < 4; ++q) {
              const float y0 = Y0 + 2.f * q, y1 = y0 + 1.f;
              bool in;
              if (mode == 3) in = ((emask[k] >> (16 * q)) & 0xffffull) != 0;
              else {
                in = !(r.my + r.ey < y0) && !(r.my - r.ey > y1);
                if (mode >= 1) in = in && !(r.mx + r.ex < X0) && !(r.mx - r.ex > X0 + 7.f);
                if (mode == 2) in = in && reach(r, X0, X0 + 7.f, y0, y1);
              }
              if (in) ++cnt[q];
            }
          }
          steps += *std::max_element(cnt, cnt + 4);
        }
        return steps;
      };
      for (int m = 0; m < 4; ++m) { fstrip64[m] += fwd(64, m); fstrip32[m] += fwd(32, m); }
      {   // the forward as shipped (TRAIN: halves of 32 entries, y-extent strip lists in lockstep): in how many of its steps does
          // no lane of the wave accept its entry? (each strip is on ITS i-th entry of the half)
        const float Y0 = (float)((ty0 + ty) * 8);
        for (int b0 = 0; b0 < kmax; b0 += 32) {
          std::vector<int> lst[4];
          for (int k = b0; k < std::min(kmax, b0 + 32); ++k) {
            const SplatRec& r = recs[(unsigned)(l[k] & 0xffffffffull)];
            for (int q = 0; q < 4; ++q) {
              const float y0 = Y0 + 2.f * q, y1 = y0 + 1.f;
              if (!(r.my + r.ey < y0) && !(r.my - r.ey > y1)) lst[q].push_back(k);
            }
          }
          size_t ml = 0;
          for (int q = 0; q < 4; ++q) ml = std::max(ml, lst[q].size());
          for (size_t i = 0; i < ml; ++i) {
            bool any = false;
            for (int q = 0; q < 4; ++q)
              if (i < lst[q].size() && ((emask[lst[q][i]] >> (16 * q)) & 0xffffull)) any = true;
            fwd_steps += 1; if (!any) fwd_steps_no_accept += 1;
          }
        }
      }
    }
  double vals[] = {n_tiles, sumL, sumK, hits, dense16, sp16, sp32, sp64, spInf, q16, q64, strip16, dead_entries,
                   live_entries, behind, task_nonzero, task_total, (double)D_all, all16_batches, live16_batches,
                   fstrip64[0], fstrip64[1], fstrip64[2], fstrip64[3], fstrip32[0], fstrip32[1], fstrip32[2], fstrip32[3],
                   fwd_steps, fwd_steps_no_accept, row_task_nonzero, row_task_total, p2_groups_zero, p2_groups, sp16_ahead,
                   walk_waves, walk_steps_max, walk_steps_balanced, walk_tiles, walk_rows_max, walk_waves_over32, walk_steps_hybrid};
  int nv = (int)(sizeof(vals) / sizeof(vals[0]));
  for (int i = 0; i < nv && i < n_out; ++i) out[i] = vals[i];
  for (int i = 0; i < 65 && nv + i < n_out; ++i) out[nv + i] = hist_hits[i];
  return nv;
}
}
