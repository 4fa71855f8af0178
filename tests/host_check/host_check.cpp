// host_check.cpp -- CPU emulation of the HIP pipeline's DATA FLOW using the product's own math header
// (skyfall-gs_amd/csrc/raster_math.h compiled by g++). TEST INFRASTRUCTURE: lets the CPU test-suite
// (-m "not gpu") check, against the oracle, everything in the product that is not wave-level code:
// projection / radii / tile ranges, the opacity-aware 8x8 binning (must never drop a contributor),
// the (depth, duplicate) sort order, per-pixel compositing forward/backward math and the
// per-Gaussian chain rule. It is never loaded by the product.
#include <algorithm>
#include <cstring>
#include <vector>

#include "../../skyfall-gs_amd/csrc/raster_math.h"

using namespace sfgs;

extern "C" {

struct HcFrame {
  int32_t W, H;
  float tanfovx, tanfovy, kernel_size, scale_modifier;
  int32_t sh_degree, sh_coeffs, depth_mode;
  const float* subpix;
  const float* bg;
  const float* view;
  const float* proj;
  const float* campos;
};

// counters: [0]=D_eff [1]=D_ref [2]=N_vis [3]=max list
int hc_render(const HcFrame* hf, int32_t N, const float* means, const float* scales, const float* rots,
              const float* opac, const float* colors, const float* shs, float* out_color, float* out_depth,
              float* out_alpha, int32_t* radii, float* rec_out /* [N,12] or NULL */, const float* dL_dcolor,
              const float* dL_ddepth, const float* dL_dalpha, float* g_means3D, float* g_means2D, float* g_scales,
              float* g_rots, float* g_opac, float* g_colors, float* g_shs, int64_t* counters) {
  FrameParams f;
  f.W = hf->W; f.H = hf->H; f.tanfovx = hf->tanfovx; f.tanfovy = hf->tanfovy;
  f.kernel_size = hf->kernel_size; f.scale_modifier = hf->scale_modifier;
  f.sh_degree = hf->sh_degree; f.sh_coeffs = hf->sh_coeffs; f.depth_mode = hf->depth_mode;
  for (int i = 0; i < 16; ++i) { f.view[i] = hf->view[i]; f.proj[i] = hf->proj[i]; }
  for (int i = 0; i < 3; ++i) { f.campos[i] = hf->campos[i]; f.bg[i] = hf->bg[i]; }
  const int W = f.W, H = f.H;
  const size_t P = (size_t)W * H;
  const int TX8 = (W + 7) / 8, TY8 = (H + 7) / 8;
  float bound = 0.f;
  if (hf->subpix) for (size_t i = 0; i < 2 * P; ++i) bound = fmaxf(bound, fabsf(hf->subpix[i]));

  std::vector<SplatRec> rec(N);
  std::vector<BinRange> br(N);
  std::vector<uint32_t> dupoff(N + 1, 0);
  std::vector<std::vector<unsigned long long>> lists((size_t)TX8 * TY8);
  std::vector<uint32_t> dup_gauss;
  int64_t dref = 0, nvis = 0;
  // K1 + K3 (plan / scatter) in index order
  for (int g = 0; g < N; ++g) {
    const Projected pr = project_gaussian(f, means + 3 * (size_t)g, scales + 3 * (size_t)g, rots + 4 * (size_t)g);
    radii[g] = pr.radius;
    dupoff[g] = (uint32_t)dup_gauss.size();
    if (!pr.visible) continue;
    ++nvis;
    dref += (int64_t)(pr.rmaxx - pr.rminx) * (pr.rmaxy - pr.rminy);
    float rgb[3];
    if (colors) { rgb[0] = colors[3 * (size_t)g]; rgb[1] = colors[3 * (size_t)g + 1]; rgb[2] = colors[3 * (size_t)g + 2]; }
    else { unsigned m; float dir[3], len; sh_to_rgb(f.sh_degree, shs + 3 * (size_t)f.sh_coeffs * g, means + 3 * (size_t)g, f.campos, rgb, &m, dir, &len); }
    rec[g] = make_record(pr, opac[g], rgb);
    br[g] = bin_range(rec[g], W, H, pr.rminx, pr.rminy, pr.rmaxx, pr.rmaxy, bound);
    const float thr = alpha_threshold_log2(rec[g].op);
    uint32_t depth_bits; memcpy(&depth_bits, &rec[g].depth, 4);
    for (int ty = br[g].y0; ty < br[g].y1; ++ty)
      for (int tx = br[g].x0; tx < br[g].x1; ++tx)
        if (bin_test(rec[g], thr, tx, ty, W, H, bound)) {
          const uint32_t d = (uint32_t)dup_gauss.size();
          lists[(size_t)ty * TX8 + tx].push_back(((unsigned long long)depth_bits << 32) | d);
          dup_gauss.push_back((uint32_t)g);
        }
  }
  dupoff[N] = (uint32_t)dup_gauss.size();
  if (rec_out) for (int g = 0; g < N; ++g) if (radii[g] > 0) memcpy(rec_out + 12 * (size_t)g, &rec[g], 48);
  size_t maxlen = 0;
  for (auto& l : lists) { std::sort(l.begin(), l.end()); maxlen = std::max(maxlen, l.size()); }
  if (counters) { counters[0] = (int64_t)dup_gauss.size(); counters[1] = dref; counters[2] = nvis; counters[3] = (int64_t)maxlen; }

  // K5 composite
  std::vector<uint32_t> n_contrib(P, 0);
  std::vector<float> final_T(P, 1.f), dacc(P, 0.f);
  for (int py = 0; py < H; ++py)
    for (int px = 0; px < W; ++px) {
      const size_t pix = (size_t)py * W + px;
      float sx = (float)px, sy = (float)py;
      if (hf->subpix) { sx += hf->subpix[pix * 2]; sy += hf->subpix[pix * 2 + 1]; }
      const auto& l = lists[(size_t)(py / 8) * TX8 + px / 8];
      PixelFwd ps;
      pixel_fwd_init(ps, true);
      for (size_t k = 0; k < l.size(); ++k) {
        const SplatRec& r = rec[dup_gauss[(uint32_t)(l[k] & 0xffffffffull)]];
        const SplatEval ev = eval_splat(r.mx, r.my, r.qa, r.qb, r.qc, r.op, sx, sy);
        pixel_fwd_step(ps, ev, r.depth, r.r, r.g, r.b, (unsigned)k);
      }
      out_color[pix] = fmaf(pixel_fwd_final_T(ps), f.bg[0], ps.C0);
      out_color[P + pix] = fmaf(pixel_fwd_final_T(ps), f.bg[1], ps.C1);
      out_color[2 * P + pix] = fmaf(pixel_fwd_final_T(ps), f.bg[2], ps.C2);
      const float a = 1.0f - pixel_fwd_final_T(ps);
      out_alpha[pix] = a;
      out_depth[pix] = f.depth_mode == 0 ? ps.D / a : ps.D;
      n_contrib[pix] = ps.last; final_T[pix] = pixel_fwd_final_T(ps); dacc[pix] = ps.D;
    }
  if (!g_means3D) return 0;

  // composite backward: per-duplicate sums (float, pixel order inside the tile)
  std::vector<float> dupgrad(dup_gauss.size() * 12, 0.f);
  const float ddelx_dx = 0.5f * (float)W, ddely_dy = 0.5f * (float)H;
  for (int py = 0; py < H; ++py)
    for (int px = 0; px < W; ++px) {
      const size_t pix = (size_t)py * W + px;
      float sx = (float)px, sy = (float)py;
      if (hf->subpix) { sx += hf->subpix[pix * 2]; sy += hf->subpix[pix * 2 + 1]; }
      const auto& l = lists[(size_t)(py / 8) * TX8 + px / 8];
      const unsigned last = n_contrib[pix];
      PixelBwd ps;
      pixel_bwd_init(ps, last, final_T[pix], dacc[pix], dL_dcolor ? dL_dcolor[pix] : 0.f,
                     dL_dcolor ? dL_dcolor[P + pix] : 0.f, dL_dcolor ? dL_dcolor[2 * P + pix] : 0.f,
                     dL_ddepth ? dL_ddepth[pix] : 0.f, dL_dalpha ? dL_dalpha[pix] : 0.f, f.depth_mode, f.bg);
      for (int k = (int)last - 1; k >= 0; --k) {
        const uint32_t d = (uint32_t)(l[k] & 0xffffffffull);
        const SplatRec& r = rec[dup_gauss[d]];
        const SplatEval ev = eval_splat(r.mx, r.my, r.qa, r.qb, r.qc, r.op, sx, sy);
        if (!ev.ok) continue;
        float v[12];
        pixel_bwd_step(ps, ev, r.qa, r.qb, r.qc, r.op, r.depth, r.r, r.g, r.b, ddelx_dx, ddely_dy, v);
        for (int i = 0; i < 12; ++i) dupgrad[(size_t)d * 12 + i] += v[i];
      }
    }
  // preprocess backward
  for (int g = 0; g < N; ++g) {
    GaussGrads out;
    memset(&out, 0, sizeof(out));
    float* gsh = g_shs ? g_shs + 3 * (size_t)f.sh_coeffs * g : nullptr;
    if (gsh) memset(gsh, 0, sizeof(float) * 3 * f.sh_coeffs);
    if (radii[g] > 0) {
      float a[12] = {0};
      for (uint32_t d = dupoff[g]; d < dupoff[g + 1]; ++d)
        for (int i = 0; i < 12; ++i) a[i] += dupgrad[(size_t)d * 12 + i];
      Grad2D A;
      A.gmx = a[0]; A.gmy = a[1]; A.absx = a[2]; A.absy = a[3]; A.gA = a[4]; A.gB = a[5]; A.gC = a[6]; A.gop = a[7];
      A.grgb[0] = a[8]; A.grgb[1] = a[9]; A.grgb[2] = a[10]; A.gdepth = a[11];
      float gshl[48];
      preprocess_backward_one(f, means + 3 * (size_t)g, scales + 3 * (size_t)g, rots + 4 * (size_t)g, opac[g],
                              shs ? shs + 3 * (size_t)f.sh_coeffs * g : nullptr, A, out, gshl);
      if (gsh) memcpy(gsh, gshl, sizeof(float) * 3 * f.sh_coeffs);
    }
    for (int i = 0; i < 3; ++i) {
      g_means3D[3 * (size_t)g + i] = out.means3D[i];
      g_means2D[3 * (size_t)g + i] = out.means2D[i];
      g_scales[3 * (size_t)g + i] = out.scales[i];
      if (g_colors) g_colors[3 * (size_t)g + i] = out.rgb[i];
    }
    for (int i = 0; i < 4; ++i) g_rots[4 * (size_t)g + i] = out.rot[i];
    g_opac[g] = out.opacity;
  }
  return 0;
}

}  // extern "C"

// ---- test hooks on the product header's helpers (same signatures as the oracle's orc_test_*) -------------
extern "C" {
void hc_cov3d(const float* s, float mod, const float* q, float* cov6) { cov3d_of(s, mod, q, cov6); }
void hc_quat_to_R(const float* q, float* R9) { quat_to_rot(q, R9); }
void hc_sh(int deg, int M, const float* sh, const float* dir, float* rgb) {
  (void)M;
  const float zero[3] = {0.f, 0.f, 0.f};
  unsigned mask; float d[3], len;
  sh_to_rgb(deg, sh, dir, zero, rgb, &mask, d, &len);
}
void hc_project(const HcFrame* hf, const float* p, const float* s, const float* q, float* out4) {
  FrameParams f;
  f.W = hf->W; f.H = hf->H; f.tanfovx = hf->tanfovx; f.tanfovy = hf->tanfovy;
  f.kernel_size = hf->kernel_size; f.scale_modifier = hf->scale_modifier;
  f.sh_degree = hf->sh_degree; f.sh_coeffs = hf->sh_coeffs; f.depth_mode = hf->depth_mode;
  for (int i = 0; i < 16; ++i) { f.view[i] = hf->view[i]; f.proj[i] = hf->proj[i]; }
  for (int i = 0; i < 3; ++i) { f.campos[i] = hf->campos[i]; f.bg[i] = hf->bg[i]; }
  const Projected pr = project_gaussian(f, p, s, q);
  out4[0] = pr.visible ? pr.mx : 0.f; out4[1] = pr.visible ? pr.my : 0.f; out4[2] = pr.tz; out4[3] = (float)pr.radius;
}
}
