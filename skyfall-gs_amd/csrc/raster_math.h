// raster_math.h -- per-Gaussian math of the rasterizer hot path (pure functions, no memory traffic).
//
// Everything here is SFGS_HD (= __host__ __device__ under hipcc, nothing under g++) so that the
// same source is (a) inlined into the gfx950 kernels of raster_fwd.hip / raster_bwd.hip and (b)
// compiled by g++ into tests/host_check (CPU "host logic" tests, no GPU needed).
//
// Reference behaviour implemented (paths relative to the reference tree; [UPSTREAM] = public
// 3DGS / Mip-Splatting algorithm, SURVEY Appendix A -- the rasterizer source itself is an
// un-vendored submodule):
//   conventions        scene/cameras.py:62-73, utils/graphics_utils.py:106-126
//   quaternion -> R    utils/general_utils.py:78-99
//   Sigma packing      utils/general_utils.py:64-76
//   SH basis           utils/sh_utils.py:57-112 ; +0.5 / clamp gaussian_renderer/__init__.py:116-117
//   near plane 0.2     scene/gaussian_model.py:276
//
// Every value that decides an INTEGER output (radius, tile rectangle, visibility) is computed as
// a fixed sequence of IEEE-754 float32 operations: this file must be compiled with
// -ffp-contract=off, and FMA is only used where spelled fmaf().
#pragma once
#include <math.h>
#include <stdint.h>

#ifdef __HIPCC__
#define SFGS_HD __host__ __device__ __forceinline__
#else
#define SFGS_HD static inline
#endif

namespace sfgs {

constexpr int TILE_REF = 16;  // tile edge of the visibility rule ([UPSTREAM] BLOCK_X/Y)
constexpr int TILE_BIN = 8;   // tile edge of OUR binning: one wave64 = one 8x8 pixel tile
constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;

struct FrameParams {  // host-side scalars + the three small camera tensors copied to kernel args
  int W, H;
  float tanfovx, tanfovy, kernel_size, scale_modifier;
  int sh_degree, sh_coeffs, depth_mode;
  float view[16], proj[16], campos[3], bg[3];
};

// 12-float record consumed by the compositing kernels (one 48-byte gather per list entry).
struct SplatRec {
  float mx, my;       // pixel-space mean
  float qa, qb;       // log2-domain conic: p2 = qa dx^2 + qb dx dy + qc dy^2 ; alpha = op * 2^p2
  float qc, op;       // op = opacity * mip coefficient
  float depth, r;     // view-space z ; colour r
  float g, b;         // colour g, b
  float ex, ey;       // half extents (pixels) of the region where alpha can reach 1/255 (-1: nowhere)
};

struct Projected {
  bool visible;
  int radius;
  int rminx, rminy, rmaxx, rmaxy;  // 16x16 tile rectangle [min,max)
  float mx, my;
  float a, b, c;      // filtered 2D covariance
  float a0, b0, c0;   // unfiltered
  float cA, cB, cC;   // conic
  float coef;
  float tx, ty, tz;
  float cov3d[6];
};

SFGS_HD int f2i_sat(float v) {
  if (v >= 2147483520.0f) return 2147483647;
  if (v <= -2147483648.0f) return (-2147483647 - 1);
  return (int)v;
}
SFGS_HD int imin(int a, int b) { return a < b ? a : b; }
SFGS_HD int imax(int a, int b) { return a > b ? a : b; }

SFGS_HD void quat_to_rot(const float* q, float R[9]) {
  const float r = q[0], x = q[1], y = q[2], z = q[3];
  R[0] = 1.f - 2.f * (y * y + z * z);
  R[1] = 2.f * (x * y - r * z);
  R[2] = 2.f * (x * z + r * y);
  R[3] = 2.f * (x * y + r * z);
  R[4] = 1.f - 2.f * (x * x + z * z);
  R[5] = 2.f * (y * z - r * x);
  R[6] = 2.f * (x * z - r * y);
  R[7] = 2.f * (y * z + r * x);
  R[8] = 1.f - 2.f * (x * x + y * y);
}

SFGS_HD void cov3d_of(const float* s, float mod, const float* q, float cov[6]) {
  float R[9];
  quat_to_rot(q, R);
  const float S0 = mod * s[0], S1 = mod * s[1], S2 = mod * s[2];
  const float M0 = R[0] * S0, M1 = R[1] * S1, M2 = R[2] * S2;
  const float M3 = R[3] * S0, M4 = R[4] * S1, M5 = R[5] * S2;
  const float M6 = R[6] * S0, M7 = R[7] * S1, M8 = R[8] * S2;
  cov[0] = M0 * M0 + M1 * M1 + M2 * M2;
  cov[1] = M0 * M3 + M1 * M4 + M2 * M5;
  cov[2] = M0 * M6 + M1 * M7 + M2 * M8;
  cov[3] = M3 * M3 + M4 * M4 + M5 * M5;
  cov[4] = M3 * M6 + M4 * M7 + M5 * M8;
  cov[5] = M6 * M6 + M7 * M7 + M8 * M8;
}

// rows of T = J * W2C (2x3), shared by forward and backward
struct Jac {
  float T0[3], T1[3];
  float ux, uy, fx, fy, x_mul, y_mul;
};

SFGS_HD Jac ewa_jacobian(const FrameParams& f, float tx, float ty, float tz) {
  Jac j;
  const float limx = 1.3f * f.tanfovx, limy = 1.3f * f.tanfovy;
  const float txtz = tx / tz, tytz = ty / tz;
  j.ux = fminf(limx, fmaxf(-limx, txtz)) * tz;
  j.uy = fminf(limy, fmaxf(-limy, tytz)) * tz;
  j.x_mul = (txtz < -limx || txtz > limx) ? 0.f : 1.f;
  j.y_mul = (tytz < -limy || tytz > limy) ? 0.f : 1.f;
  j.fx = (float)f.W / (2.0f * f.tanfovx);
  j.fy = (float)f.H / (2.0f * f.tanfovy);
  const float J00 = j.fx / tz, J02 = -(j.fx * j.ux) / (tz * tz);
  const float J11 = j.fy / tz, J12 = -(j.fy * j.uy) / (tz * tz);
  for (int k = 0; k < 3; ++k) {
    j.T0[k] = J00 * f.view[k * 4 + 0] + J02 * f.view[k * 4 + 2];
    j.T1[k] = J11 * f.view[k * 4 + 1] + J12 * f.view[k * 4 + 2];
  }
  return j;
}

// SURVEY Appendix A.2 steps 1-8. Returns visible=false for culled Gaussians (radius 0).
SFGS_HD Projected project_gaussian(const FrameParams& f, const float* p, const float* s, const float* q) {
  Projected o;
  o.visible = false;
  o.radius = 0;
  o.rminx = o.rminy = o.rmaxx = o.rmaxy = 0;
  const float* V = f.view;
  const float* PM = f.proj;
  const float tx = V[0] * p[0] + V[4] * p[1] + V[8] * p[2] + V[12];
  const float ty = V[1] * p[0] + V[5] * p[1] + V[9] * p[2] + V[13];
  const float tz = V[2] * p[0] + V[6] * p[1] + V[10] * p[2] + V[14];
  o.tx = tx; o.ty = ty; o.tz = tz;
  if (!(tz > 0.2f)) return o;

  const float hx = PM[0] * p[0] + PM[4] * p[1] + PM[8] * p[2] + PM[12];
  const float hy = PM[1] * p[0] + PM[5] * p[1] + PM[9] * p[2] + PM[13];
  const float hw = PM[3] * p[0] + PM[7] * p[1] + PM[11] * p[2] + PM[15];
  const float pw = 1.0f / (hw + 0.0000001f);
  const float ndcx = hx * pw, ndcy = hy * pw;

  cov3d_of(s, f.scale_modifier, q, o.cov3d);
  const float* c3 = o.cov3d;
  const Jac j = ewa_jacobian(f, tx, ty, tz);
  float v0[3], v1[3];
  v0[0] = c3[0] * j.T0[0] + c3[1] * j.T0[1] + c3[2] * j.T0[2];
  v0[1] = c3[1] * j.T0[0] + c3[3] * j.T0[1] + c3[4] * j.T0[2];
  v0[2] = c3[2] * j.T0[0] + c3[4] * j.T0[1] + c3[5] * j.T0[2];
  v1[0] = c3[0] * j.T1[0] + c3[1] * j.T1[1] + c3[2] * j.T1[2];
  v1[1] = c3[1] * j.T1[0] + c3[3] * j.T1[1] + c3[4] * j.T1[2];
  v1[2] = c3[2] * j.T1[0] + c3[4] * j.T1[1] + c3[5] * j.T1[2];
  const float a0 = j.T0[0] * v0[0] + j.T0[1] * v0[1] + j.T0[2] * v0[2];
  const float b0 = j.T0[0] * v1[0] + j.T0[1] * v1[1] + j.T0[2] * v1[2];
  const float c0 = j.T1[0] * v1[0] + j.T1[1] * v1[1] + j.T1[2] * v1[2];
  o.a0 = a0; o.b0 = b0; o.c0 = c0;

  // Mip-Splatting 2D filter + opacity compensation [UPSTREAM]
  const float ks = f.kernel_size;
  const float det0 = fmaxf(1e-6f, a0 * c0 - b0 * b0);
  const float a = a0 + ks, c = c0 + ks, b = b0;
  const float det1 = fmaxf(1e-6f, a * c - b * b);
  float coef = sqrtf(det0 / (det1 + 1e-6f) + 1e-6f);
  if (det0 <= 1e-6f || det1 <= 1e-6f) coef = 0.0f;
  o.a = a; o.b = b; o.c = c; o.coef = coef;

  const float det = a * c - b * b;
  if (det == 0.0f) return o;
  const float det_inv = 1.0f / det;
  o.cA = c * det_inv; o.cB = -b * det_inv; o.cC = a * det_inv;
  const float mid = 0.5f * (a + c);
  const float disc = sqrtf(fmaxf(0.1f, mid * mid - det));
  const float l1 = mid + disc, l2 = mid - disc;
  const float radf = ceilf(3.0f * sqrtf(fmaxf(l1, l2)));
  const float mx = ((ndcx + 1.0f) * (float)f.W - 1.0f) * 0.5f;
  const float my = ((ndcy + 1.0f) * (float)f.H - 1.0f) * 0.5f;
  // non-finite projections are culled (x - x == 0 only for finite x)
  if (!((mx - mx) == 0.0f && (my - my) == 0.0f && (radf - radf) == 0.0f)) return o;

  const int TX = (f.W + TILE_REF - 1) / TILE_REF, TY = (f.H + TILE_REF - 1) / TILE_REF;
  o.rminx = imin(TX, imax(0, f2i_sat((mx - radf) / (float)TILE_REF)));
  o.rminy = imin(TY, imax(0, f2i_sat((my - radf) / (float)TILE_REF)));
  o.rmaxx = imin(TX, imax(0, f2i_sat((mx + radf + (float)(TILE_REF - 1)) / (float)TILE_REF)));
  o.rmaxy = imin(TY, imax(0, f2i_sat((my + radf + (float)(TILE_REF - 1)) / (float)TILE_REF)));
  if ((o.rmaxx - o.rminx) * (o.rmaxy - o.rminy) == 0) return o;
  o.mx = mx; o.my = my;
  o.radius = f2i_sat(radf);
  o.visible = true;
  return o;
}

// ---- spherical harmonics (utils/sh_utils.py:57-112) ------------------------------------------
constexpr float SH0 = 0.28209479177387814f;
constexpr float SH1 = 0.4886025119029199f;
constexpr float SH2_0 = 1.0925484305920792f, SH2_1 = -1.0925484305920792f, SH2_2 = 0.31539156525252005f,
                SH2_3 = -1.0925484305920792f, SH2_4 = 0.5462742152960396f;
constexpr float SH3_0 = -0.5900435899266435f, SH3_1 = 2.890611442640554f, SH3_2 = -0.4570457994644658f,
                SH3_3 = 0.3731763325901154f, SH3_4 = -0.4570457994644658f, SH3_5 = 1.445305721320277f,
                SH3_6 = -0.5900435899266435f;
// degree 4 (utils/sh_utils.py:44-54,101-111): in the reference only the Python eval_sh knows it; here every SH path does
constexpr float SH4_0 = 2.5033429417967046f, SH4_1 = -1.7701307697799304f, SH4_2 = 0.9461746957575601f,
                SH4_3 = -0.6690465435572892f, SH4_4 = 0.10578554691520431f, SH4_5 = -0.6690465435572892f,
                SH4_6 = 0.47308734787878004f, SH4_7 = -1.7701307697799304f, SH4_8 = 0.6258357354491761f;
constexpr int SH_MAX_COEFFS = 25;

// basis[k] for k < (deg+1)^2 at unit direction (x,y,z); colour = sum_k basis[k] * sh[k]
SFGS_HD void sh_basis(int deg, float x, float y, float z, float* Bk) {
  Bk[0] = SH0;
  if (deg > 0) {
    Bk[1] = -SH1 * y; Bk[2] = SH1 * z; Bk[3] = -SH1 * x;
    if (deg > 1) {
      const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
      Bk[4] = SH2_0 * xy; Bk[5] = SH2_1 * yz; Bk[6] = SH2_2 * (2.0f * zz - xx - yy);
      Bk[7] = SH2_3 * xz; Bk[8] = SH2_4 * (xx - yy);
      if (deg > 2) {
        Bk[9] = SH3_0 * y * (3.0f * xx - yy);
        Bk[10] = SH3_1 * xy * z;
        Bk[11] = SH3_2 * y * (4.0f * zz - xx - yy);
        Bk[12] = SH3_3 * z * (2.0f * zz - 3.0f * xx - 3.0f * yy);
        Bk[13] = SH3_4 * x * (4.0f * zz - xx - yy);
        Bk[14] = SH3_5 * z * (xx - yy);
        Bk[15] = SH3_6 * x * (xx - 3.0f * yy);
        if (deg > 3) {
          Bk[16] = SH4_0 * xy * (xx - yy);
          Bk[17] = SH4_1 * yz * (3.f * xx - yy);
          Bk[18] = SH4_2 * xy * (7.f * zz - 1.f);
          Bk[19] = SH4_3 * yz * (7.f * zz - 3.f);
          Bk[20] = SH4_4 * (zz * (35.f * zz - 30.f) + 3.f);
          Bk[21] = SH4_5 * xz * (7.f * zz - 3.f);
          Bk[22] = SH4_6 * (xx - yy) * (7.f * zz - 1.f);
          Bk[23] = SH4_7 * xz * (xx - 3.f * yy);
          Bk[24] = SH4_8 * (xx * (xx - 3.f * yy) - yy * (3.f * xx - yy));
        }
      }
    }
  }
}

// d basis[k] / d(x,y,z) (direction treated as free variables; the normalisation Jacobian is applied
// by the caller)
SFGS_HD void sh_basis_grad(int deg, float x, float y, float z, float* dBx, float* dBy, float* dBz) {
  dBx[0] = dBy[0] = dBz[0] = 0.f;
  if (deg > 0) {
    dBx[1] = 0.f; dBy[1] = -SH1; dBz[1] = 0.f;
    dBx[2] = 0.f; dBy[2] = 0.f; dBz[2] = SH1;
    dBx[3] = -SH1; dBy[3] = 0.f; dBz[3] = 0.f;
    if (deg > 1) {
      dBx[4] = SH2_0 * y; dBy[4] = SH2_0 * x; dBz[4] = 0.f;
      dBx[5] = 0.f; dBy[5] = SH2_1 * z; dBz[5] = SH2_1 * y;
      dBx[6] = SH2_2 * -2.f * x; dBy[6] = SH2_2 * -2.f * y; dBz[6] = SH2_2 * 4.f * z;
      dBx[7] = SH2_3 * z; dBy[7] = 0.f; dBz[7] = SH2_3 * x;
      dBx[8] = SH2_4 * 2.f * x; dBy[8] = SH2_4 * -2.f * y; dBz[8] = 0.f;
      if (deg > 2) {
        const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
        dBx[9] = SH3_0 * 6.f * xy; dBy[9] = SH3_0 * 3.f * (xx - yy); dBz[9] = 0.f;
        dBx[10] = SH3_1 * yz; dBy[10] = SH3_1 * xz; dBz[10] = SH3_1 * xy;
        dBx[11] = SH3_2 * -2.f * xy; dBy[11] = SH3_2 * (4.f * zz - xx - 3.f * yy); dBz[11] = SH3_2 * 8.f * yz;
        dBx[12] = SH3_3 * -6.f * xz; dBy[12] = SH3_3 * -6.f * yz; dBz[12] = SH3_3 * 3.f * (2.f * zz - xx - yy);
        dBx[13] = SH3_4 * (4.f * zz - 3.f * xx - yy); dBy[13] = SH3_4 * -2.f * xy; dBz[13] = SH3_4 * 8.f * xz;
        dBx[14] = SH3_5 * 2.f * xz; dBy[14] = SH3_5 * -2.f * yz; dBz[14] = SH3_5 * (xx - yy);
        dBx[15] = SH3_6 * 3.f * (xx - yy); dBy[15] = SH3_6 * -6.f * xy; dBz[15] = 0.f;
        if (deg > 3) {
          dBx[16] = SH4_0 * y * (3.f * xx - yy); dBy[16] = SH4_0 * x * (xx - 3.f * yy); dBz[16] = 0.f;
          dBx[17] = SH4_1 * 6.f * xy * z; dBy[17] = SH4_1 * 3.f * z * (xx - yy); dBz[17] = SH4_1 * y * (3.f * xx - yy);
          dBx[18] = SH4_2 * y * (7.f * zz - 1.f); dBy[18] = SH4_2 * x * (7.f * zz - 1.f); dBz[18] = SH4_2 * 14.f * xy * z;
          dBx[19] = 0.f; dBy[19] = SH4_3 * z * (7.f * zz - 3.f); dBz[19] = SH4_3 * y * (21.f * zz - 3.f);
          dBx[20] = 0.f; dBy[20] = 0.f; dBz[20] = SH4_4 * z * (140.f * zz - 60.f);
          dBx[21] = SH4_5 * z * (7.f * zz - 3.f); dBy[21] = 0.f; dBz[21] = SH4_5 * x * (21.f * zz - 3.f);
          dBx[22] = SH4_6 * 2.f * x * (7.f * zz - 1.f); dBy[22] = SH4_6 * -2.f * y * (7.f * zz - 1.f);
          dBz[22] = SH4_6 * 14.f * z * (xx - yy);
          dBx[23] = SH4_7 * 3.f * z * (xx - yy); dBy[23] = SH4_7 * -6.f * xy * z; dBz[23] = SH4_7 * x * (xx - 3.f * yy);
          dBx[24] = SH4_8 * 4.f * x * (xx - 3.f * yy); dBy[24] = SH4_8 * 4.f * y * (yy - 3.f * xx); dBz[24] = 0.f;
        }
      }
    }
  }
}

// colour of one Gaussian from its SH coefficients, +0.5, clamp at 0 (gaussian_renderer/__init__.py:115-118,124-125).
// clamp_mask bit c set <=> channel c was clamped (its gradient is zero).
// Coefficient (k, c) sits at sh[k * sk + c * sc]: (sk, sc) = (3, 1) is the rasterizer's own [K,3] layout
// (`shs = pc.get_features`), (1, K) the CHANNEL-MAJOR [3,K] layout utils/sh_utils.py eval_sh takes (render()'s Python
// colour paths, :112-118). dir_in != nullptr: the view direction is GIVEN (eval_sh's `dirs` argument, used as it is --
// eval_sh does not normalise either); otherwise it is normalize(p - campos).
SFGS_HD void sh_to_rgb(int deg, const float* sh, const float* p, const float* campos, float rgb[3],
                       unsigned* clamp_mask, float dir[3], float* len_out, int sk = 3, int sc = 1,
                       const float* dir_in = nullptr) {
  float x, y, z, len = 1.f;
  if (dir_in) {
    x = dir_in[0]; y = dir_in[1]; z = dir_in[2];
  } else {
    const float dx = p[0] - campos[0], dy = p[1] - campos[1], dz = p[2] - campos[2];
    len = sqrtf(dx * dx + dy * dy + dz * dz);
    x = dx / len; y = dy / len; z = dz / len;
  }
  dir[0] = x; dir[1] = y; dir[2] = z;
  *len_out = len;
  float Bk[SH_MAX_COEFFS];
  sh_basis(deg, x, y, z, Bk);
  const int M = (deg + 1) * (deg + 1);
  unsigned mask = 0;
  for (int c = 0; c < 3; ++c) {
    float r = 0.f;
    for (int k = 0; k < M; ++k) r += Bk[k] * sh[k * sk + c * sc];
    r += 0.5f;
    if (r < 0.f) { mask |= 1u << c; r = 0.f; }
    rgb[c] = r;
  }
  *clamp_mask = mask;
}

// ---- opacity-aware binning -------------------------------------------------------------------
// A (Gaussian, 8x8 tile) pair is binned iff the tile lies inside the reference's 16x16-tile
// rectangle AND alpha = op * 2^p2 can reach 1/255 somewhere on the tile's sample rectangle. The
// second test is conservative (margins below), so the set of per-pixel contributors is exactly
// the reference's.
SFGS_HD float alpha_threshold_log2(float op) {
  // p2 >= thr  <=  op * 2^p2 >= 0.999/255 (margin 0.1 % in alpha + 2e-3 absolute in p2)
  return log2f(0.999f / (255.0f * op)) - 2e-3f;
}

SFGS_HD void fill_extents(SplatRec& r, float cov_a, float cov_c) {
  const float thr = alpha_threshold_log2(r.op);  // <= 0 when the splat can be seen at all
  if (r.op != r.op) {
    // NaN opacity: min(0.99, NaN) = 0.99 in the compositing rule (fminf semantics, as in the CUDA original), i.e.
    // such a splat blends with alpha 0.99 wherever power <= 0 -- keep every tile of the reference rectangle
    r.ex = INFINITY; r.ey = INFINITY;
    return;
  }
  if (!(r.op > 0.f) || thr > 0.f) { r.ex = -1.f; r.ey = -1.f; return; }
  // power = ln2 * p2 >= ln2 * thr  <=>  d^T conic d <= tau2 = -2 ln2 thr ; bbox = sqrt(tau2 * cov_xx)
  const float tau2 = -2.0f * LN2 * thr;
  r.ex = sqrtf(tau2 * cov_a) * 1.001f + 1e-3f;
  r.ey = sqrtf(tau2 * cov_c) * 1.001f + 1e-3f;
}

SFGS_HD float p2_at(const SplatRec& r, float dx, float dy) {
  return r.qa * dx * dx + r.qc * dy * dy + r.qb * dx * dy;
}

// sample rectangle [x0,x1] x [y0,y1] in pixel coordinates (already widened by the subpixel bound): can alpha reach
// 1/255 anywhere on it, i.e. is max p2 over the rectangle >= thr? p2 is a concave quadratic about the splat's centre,
// so with the centre outside the rectangle the maximum lies on an edge that FACES the centre (from any other point of
// the rectangle one can move towards the centre, uphill, and stay inside), and on an edge it is the 1-D stationary
// point clamped to the edge. The nearer vertical and the nearer horizontal edge are therefore all that has to be
// evaluated (an edge that does not face the centre only adds a smaller candidate): 2 clamped stationary points
// instead of 4 corners + 4 edges -- the tile tests are a fifth of preprocess_kernel's instructions.
// hc = -0.5f / r.qc and ha = -0.5f / r.qa are passed in, so that a caller testing many tiles of one splat divides once
// (tile_can_contribute below computes them itself: the same two divisions, the same result bit for bit)
SFGS_HD bool tile_can_contribute_h(const SplatRec& r, float thr, float x0, float x1, float y0, float y1, float hc, float ha) {
  const float dxl = r.mx - x1, dxh = r.mx - x0, dyl = r.my - y1, dyh = r.my - y0;
  if (dxl <= 0.f && dxh >= 0.f && dyl <= 0.f && dyh >= 0.f) return true;  // centre inside: p2 = 0
  float best;
  if (r.qa < 0.f && r.qc < 0.f) {
    const float dxv = fabsf(dxl) < fabsf(dxh) ? dxl : dxh;   // nearer vertical edge
    const float dyv = fabsf(dyl) < fabsf(dyh) ? dyl : dyh;   // nearer horizontal edge
    const float dy = fminf(dyh, fmaxf(dyl, r.qb * dxv * hc));
    const float dx = fminf(dxh, fmaxf(dxl, r.qb * dyv * ha));
    best = fmaxf(p2_at(r, dxv, dy), p2_at(r, dx, dyv));
  } else {
    // degenerate or NaN conic: all four corners and whatever edge maxima exist (NaN keeps the pair below)
    best = fmaxf(fmaxf(p2_at(r, dxl, dyl), p2_at(r, dxl, dyh)), fmaxf(p2_at(r, dxh, dyl), p2_at(r, dxh, dyh)));
    if (r.qc < 0.f) {
      float dy = fminf(dyh, fmaxf(dyl, r.qb * dxl * hc));
      best = fmaxf(best, p2_at(r, dxl, dy));
      dy = fminf(dyh, fmaxf(dyl, r.qb * dxh * hc));
      best = fmaxf(best, p2_at(r, dxh, dy));
    }
    if (r.qa < 0.f) {
      float dx = fminf(dxh, fmaxf(dxl, r.qb * dyl * ha));
      best = fmaxf(best, p2_at(r, dx, dyl));
      dx = fminf(dxh, fmaxf(dxl, r.qb * dyh * ha));
      best = fmaxf(best, p2_at(r, dx, dyh));
    }
  }
  return !(best < thr);  // NaN keeps the pair
}
SFGS_HD bool tile_can_contribute(const SplatRec& r, float thr, float x0, float x1, float y0, float y1) {
  return tile_can_contribute_h(r, thr, x0, x1, y0, y1, -0.5f / r.qc, -0.5f / r.qa);
}

struct BinRange { int x0, x1, y0, y1; };  // 8x8-tile index ranges [x0,x1) x [y0,y1)

// tile range to visit: reference rectangle (in 16-tiles) intersected with the alpha bbox
SFGS_HD BinRange bin_range(const SplatRec& r, int W, int H, int rminx, int rminy, int rmaxx, int rmaxy,
                           float bound) {
  BinRange br;
  const int TX8 = (W + TILE_BIN - 1) / TILE_BIN, TY8 = (H + TILE_BIN - 1) / TILE_BIN;
  br.x0 = rminx * 2; br.x1 = imin(rmaxx * 2, TX8);
  br.y0 = rminy * 2; br.y1 = imin(rmaxy * 2, TY8);
  if (r.ex < 0.f) { br.x1 = br.x0; br.y1 = br.y0; return br; }
  // pixels that can be reached: [mx - ex - bound, mx + ex + bound]
  const float lo_x = r.mx - r.ex - bound, hi_x = r.mx + r.ex + bound;
  const float lo_y = r.my - r.ey - bound, hi_y = r.my + r.ey + bound;
  const int tx_lo = f2i_sat(floorf(lo_x / (float)TILE_BIN));
  const int ty_lo = f2i_sat(floorf(lo_y / (float)TILE_BIN));
  const int tx_hi = imin(f2i_sat(floorf(hi_x / (float)TILE_BIN)), TX8 - 1);  // clamp before the +1 below
  const int ty_hi = imin(f2i_sat(floorf(hi_y / (float)TILE_BIN)), TY8 - 1);
  br.x0 = imax(br.x0, tx_lo); br.x1 = imin(br.x1, tx_hi + 1);
  br.y0 = imax(br.y0, ty_lo); br.y1 = imin(br.y1, ty_hi + 1);
  if (br.x1 < br.x0) br.x1 = br.x0;
  if (br.y1 < br.y0) br.y1 = br.y0;
  return br;
}

// the tile whose first pixel is (fx, fy) -- tile index x 8 as a float, exact below 2^24 -- on an image whose last pixel is
// (wm1, hm1): the same four bounds as bin_test below, bit for bit (integer-valued floats: the float min and the float
// + 7 are the integer ones), for a walk that steps fx / fy by 8.0f instead of converting tile indices in every iteration
SFGS_HD bool bin_test_at_h(const SplatRec& r, float thr, float fx, float fy, float wm1, float hm1, float bound, float hc,
                           float ha) {
  const float x0 = fx - bound;
  const float x1 = fminf(fx + (float)(TILE_BIN - 1), wm1) + bound;
  const float y0 = fy - bound;
  const float y1 = fminf(fy + (float)(TILE_BIN - 1), hm1) + bound;
  return tile_can_contribute_h(r, thr, x0, x1, y0, y1, hc, ha);
}
SFGS_HD bool bin_test_at(const SplatRec& r, float thr, float fx, float fy, float wm1, float hm1, float bound) {
  return bin_test_at_h(r, thr, fx, fy, wm1, hm1, bound, -0.5f / r.qc, -0.5f / r.qa);
}

SFGS_HD bool bin_test(const SplatRec& r, float thr, int tx, int ty, int W, int H, float bound) {
  const float x0 = (float)(tx * TILE_BIN) - bound;
  const float x1 = (float)imin(tx * TILE_BIN + TILE_BIN - 1, W - 1) + bound;
  const float y0 = (float)(ty * TILE_BIN) - bound;
  const float y1 = (float)imin(ty * TILE_BIN + TILE_BIN - 1, H - 1) + bound;
  return tile_can_contribute(r, thr, x0, x1, y0, y1);
}

// Build the compositing record of a visible Gaussian.
SFGS_HD SplatRec make_record(const Projected& pr, float opacity, const float rgb[3]) {
  SplatRec r;
  r.mx = pr.mx; r.my = pr.my;
  r.qa = -0.5f * LOG2E * pr.cA;
  r.qb = -LOG2E * pr.cB;
  r.qc = -0.5f * LOG2E * pr.cC;
  r.op = opacity * pr.coef;
  r.depth = pr.tz;
  r.r = rgb[0]; r.g = rgb[1]; r.b = rgb[2];
  fill_extents(r, pr.a, pr.c);
  return r;
}

// ---- per-pixel compositing math, shared by the kernels and tests/host_check --
SFGS_HD float fast_exp2(float x) {
#ifdef __HIP_DEVICE_COMPILE__
  return __builtin_amdgcn_exp2f(x);  // v_exp_f32
#else
  return exp2f(x);
#endif
}
SFGS_HD float fast_rcp(float x) {
#ifdef __HIP_DEVICE_COMPILE__
  return __builtin_amdgcn_rcpf(x);   // v_rcp_f32 (1 ulp)
#else
  return 1.0f / x;
#endif
}

struct SplatEval {
  float dx, dy, G, alpha;
  bool ok;  // passes the reference's per-pixel tests: power <= 0 and alpha >= 1/255
};

// the one expression both passes use to decide whether a splat touches a sample point
SFGS_HD SplatEval eval_splat(float mx, float my, float qa, float qb, float qc, float op, float sx, float sy) {
  SplatEval e;
  e.dx = mx - sx; e.dy = my - sy;
  const float p2 = fmaf(qa * e.dx, e.dx, fmaf(qc * e.dy, e.dy, (qb * e.dx) * e.dy));
  e.G = fast_exp2(p2);
  e.alpha = fminf(0.99f, op * e.G);
  e.ok = !(p2 > 0.f) && !(e.alpha < 1.0f / 255.0f);
  return e;
}

struct PixelFwd {
  // live transmittance, SIGNED: > 0 while the pixel can still accept splats; when the pixel saturates it becomes
  // -|T| (the transmittance after the last ACCEPTED splat, sign-flipped), which makes every later test_T negative,
  // i.e. "stop" again, without a separate saturated flag or a second T register. |T| is the reference's final T.
  float T;
  float C0, C1, C2, D;
  unsigned last; // 1-based list position of the last accepted splat
};

SFGS_HD void pixel_fwd_init(PixelFwd& s, bool inside) {
  s.T = inside ? 1.f : -1.f;   // pixels outside the image: saturated from the start, never written
  s.C0 = s.C1 = s.C2 = s.D = 0.f; s.last = 0;
}

SFGS_HD float pixel_fwd_final_T(const PixelFwd& s) { return fabsf(s.T); }

// k = 0-based list position. Branch-free restatement of SURVEY A.4:
//   skip if power > 0 or alpha < 1/255;  stop (splat NOT applied) if T (1 - alpha) < 1e-4;  else blend.
SFGS_HD void pixel_fwd_step(PixelFwd& s, const SplatEval& e, float depth, float r, float g, float b, unsigned k) {
  const float test_T = s.T * (1.0f - e.alpha);
  const bool acc = e.ok && !(test_T < 0.0001f);
  const float w = acc ? e.alpha * s.T : 0.f;
  s.C0 = fmaf(r, w, s.C0); s.C1 = fmaf(g, w, s.C1); s.C2 = fmaf(b, w, s.C2);
  s.D = fmaf(depth, w, s.D);
  s.last = acc ? k + 1 : s.last;
  s.T = acc ? test_T : (e.ok ? -fabsf(s.T) : s.T);
}

// Training-mode variant: instead of the last-contributor index, the pixel records WHICH entries it blended as a bit
// mask (bit = entry index within the current 32-entry half batch; the shift uses the low five bits of j). The
// backward walks exactly these bits, so it never re-tests a pair (raster_bwd.hip). Same arithmetic as pixel_fwd_step.
SFGS_HD void pixel_fwd_step_mask(PixelFwd& s, const SplatEval& e, float depth, float r, float g, float b, unsigned j,
                                 unsigned& mask) {
  const float test_T = s.T * (1.0f - e.alpha);
  const bool acc = e.ok && !(test_T < 0.0001f);
  const float w = acc ? e.alpha * s.T : 0.f;
  s.C0 = fmaf(r, w, s.C0); s.C1 = fmaf(g, w, s.C1); s.C2 = fmaf(b, w, s.C2);
  s.D = fmaf(depth, w, s.D);
  mask |= (acc ? 1u : 0u) << (j & 31u);
  s.T = acc ? test_T : (e.ok ? -fabsf(s.T) : s.T);
}

struct PixelBwd {
  float Tr, last_alpha;
  float A, q_prev;   // A = sum_ch accum_ch * g_ch, q_prev = sum_ch value_ch(previous splat) * g_ch  (see below)
  float gch[5];      // upstream gradients of the five accumulators: r, g, b, raw depth, alpha
  float bg_dot, T_final;
};

// upstream gradients of one pixel -> gradients of the raw accumulators (depth normalisation folded in)
SFGS_HD void pixel_bwd_init(PixelBwd& s, unsigned last, float T_final, float dacc, float gr, float gg, float gb,
                            float gdep, float galp, int depth_mode, const float bg[3]) {
  s.Tr = T_final; s.T_final = T_final; s.last_alpha = 0.f;
  s.A = 0.f; s.q_prev = 0.f;
  s.gch[0] = gr; s.gch[1] = gg; s.gch[2] = gb;
  if (depth_mode == 0) {  // depth = Dacc / a, a = 1 - T_final
    const float a = 1.0f - T_final;
    s.gch[3] = gdep / a;
    s.gch[4] = galp - gdep * dacc / (a * a);
  } else {
    s.gch[3] = gdep;
    s.gch[4] = galp;
  }
  if (last == 0) { for (int i = 0; i < 5; ++i) s.gch[i] = 0.f; }  // nothing hit (a = 0): no gradient
  s.bg_dot = bg[0] * s.gch[0] + bg[1] * s.gch[1] + bg[2] * s.gch[2];
}

// One contributing splat, back to front: advances the pixel's recurrences and returns the two scalars
// every gradient of this (pixel, splat) pair is built from:
//   u = G * dL/dalpha   and   w = alpha * T  (the blending weight).
// The reference-style recurrence keeps, per channel, accum_ch = "value behind this splat":
//   accum_ch <- last_alpha * last_value_ch + (1 - last_alpha) * accum_ch ,
//   dL/dalpha = T * sum_ch (value_ch - accum_ch) * g_ch  - (T_final / (1 - alpha)) * (bg . g_rgb).
// Only its projection on g is ever used, and the recurrence is linear, so it is carried as ONE scalar:
//   A = sum_ch accum_ch g_ch,  q = sum_ch value_ch g_ch :  A <- last_alpha * q_prev + (1 - last_alpha) * A.
// (Identical algebra, 5x fewer operations; no cancellation is introduced.)
template <bool HAS_BG = true>
SFGS_HD void pixel_bwd_scalars(PixelBwd& s, const SplatEval& e, float depth, float r, float g, float b, float& u,
                               float& w) {
  const float inv = fast_rcp(1.0f - e.alpha);
  s.Tr = s.Tr * inv;
  w = e.alpha * s.Tr;
  const float q = fmaf(r, s.gch[0], fmaf(g, s.gch[1], fmaf(b, s.gch[2], fmaf(depth, s.gch[3], s.gch[4]))));
  s.A = fmaf(s.last_alpha, s.q_prev - s.A, s.A);
  s.q_prev = q;
  s.last_alpha = e.alpha;
  float dL_dalpha;
  // HAS_BG = false: the caller knows bg . g_rgb == 0 (black background, the reference's default): the fma's addend
  // is -0 and the result is the rounded product -- same bits, two instructions fewer
  if constexpr (HAS_BG) dL_dalpha = fmaf(q - s.A, s.Tr, -(s.T_final * inv) * s.bg_dot);
  else dL_dalpha = (q - s.A) * s.Tr;
  u = e.G * dL_dalpha;
}

// Per-pixel partial sums v[12] (Grad2D order) of one pair from (u, w). The min(0.99, .) clamp is ignored
// in the derivative [UPSTREAM]. Conic in natural-log units: A = -2 ln2 qa, B = -ln2 qb, C = -2 ln2 qc.
SFGS_HD void pair_partials(const SplatEval& e, float u, float w, float qa, float qb, float qc, float op,
                           const float gch[5], float ddelx_dx, float ddely_dy, float v[12]) {
  const float cA = -2.0f * LN2 * qa, cB = -LN2 * qb, cC = -2.0f * LN2 * qc;
  const float ou = op * u;  // = dL/dG * G
  const float gx = -ou * (cA * e.dx + cB * e.dy) * ddelx_dx;
  const float gy = -ou * (cC * e.dy + cB * e.dx) * ddely_dy;
  v[0] = gx; v[1] = gy; v[2] = fabsf(gx); v[3] = fabsf(gy);
  v[4] = -0.5f * ou * e.dx * e.dx;
  v[5] = -ou * e.dx * e.dy;
  v[6] = -0.5f * ou * e.dy * e.dy;
  v[7] = u;
  v[8] = w * gch[0]; v[9] = w * gch[1]; v[10] = w * gch[2];
  v[11] = w * gch[3];
}

SFGS_HD void pixel_bwd_step(PixelBwd& s, const SplatEval& e, float qa, float qb, float qc, float op, float depth,
                            float r, float g, float b, float ddelx_dx, float ddely_dy, float v[12]) {
  float u, w;
  pixel_bwd_scalars(s, e, depth, r, g, b, u, w);
  pair_partials(e, u, w, qa, qb, qc, op, s.gch, ddelx_dx, ddely_dy, v);
}

// ---- backward of the per-Gaussian chain (second half) -------------------
// Per-Gaussian sums of the 2D gradients produced by the compositing backward.
struct Grad2D {
  float gmx, gmy;     // dL/dmean2D, already in NDC units (x 0.5 W, x 0.5 H)
  float absx, absy;   // sum |.| of the same
  float gA, gB, gC;   // dL/dconic
  float gop;          // dL/d(op) where op = opacity * coef
  float grgb[3];
  float gdepth;
};

struct GaussGrads {
  float means3D[3], means2D[3], scales[3], rot[4], opacity, rgb[3];
};

// sh / g_sh may be null (precomputed colours). g_sh [M_total,3] is fully written (zeros above the
// active degree).
// The per-duplicate records composite_bwd writes (round 4) hold the RAW sums over the tile's pixels, about the splat's
// mean: sum u dx, sum u dy, sum |u| |lx|, sum |u| |ly|, sum u dx^2, sum u dx dy, sum u dy^2, sum u, and the four
// colour / depth sums (u = G dL/dalpha, lx = cA dx + cB dy, ly = cC dy + cB dx). The factors that turn them into Grad2D
// -- op and the conic -- are the same for every duplicate of a Gaussian, so they are applied ONCE to the summed record
// here instead of once per (Gaussian, tile) pair in the compositing kernel (20 instructions per 16-entry batch there).
// The coefficients are formed exactly as make_record / composite_bwd form them (same float sequence).
struct GradSums {
  float x, y, ax, ay, xx, xy, yy, u, r, g, b, d;
};

SFGS_HD Grad2D grad2d_from_sums(const GradSums& S, const Projected& pr, float opacity, int W, int H) {
  const float qa = -0.5f * LOG2E * pr.cA, qb = -LOG2E * pr.cB, qc = -0.5f * LOG2E * pr.cC;   // make_record
  const float cA = -2.0f * LN2 * qa, cB = -LN2 * qb, cC = -2.0f * LN2 * qc;                   // the kernels' natural-log conic
  const float op = opacity * pr.coef;
  const float ddelx_dx = 0.5f * (float)W, ddely_dy = 0.5f * (float)H;
  const float sxm = -op * ddelx_dx, sym = -op * ddely_dy;
  Grad2D A;
  A.gmx = sxm * (cA * S.x + cB * S.y);
  A.gmy = sym * (cC * S.y + cB * S.x);
  A.absx = op * ddelx_dx * S.ax;
  A.absy = op * ddely_dy * S.ay;
  A.gA = -0.5f * op * S.xx;
  A.gB = -op * S.xy;
  A.gC = -0.5f * op * S.yy;
  A.gop = S.u;
  A.grgb[0] = S.r; A.grgb[1] = S.g; A.grgb[2] = S.b;
  A.gdepth = S.d;
  return A;
}

// sh_cm / dir_in / g_dir: the eval_sh-folded colour path (sh_to_rgb above): channel-major coefficients and coefficient
// gradients, the view direction given, its gradient returned in g_dir[3] instead of flowing into means3D.
SFGS_HD void preprocess_backward_pr(const FrameParams& f, const Projected& pr, const float* p, const float* s,
                                    const float* q, float opacity, const float* sh, const Grad2D& A, GaussGrads& out,
                                    float* g_sh, bool sh_cm = false, const float* dir_in = nullptr,
                                    float* g_dir = nullptr, const float* center = nullptr) {
  const float* V = f.view;
  const float* PM = f.proj;
  float gp[3] = {0.f, 0.f, 0.f};

  out.means2D[0] = A.gmx; out.means2D[1] = A.gmy;
  out.means2D[2] = sqrtf(A.absx * A.absx + A.absy * A.absy);

  // op = opacity * coef
  out.opacity = A.gop * pr.coef;
  const float gcoef = A.gop * opacity;

  // conic = inverse(cov): dL/d(a,b,c)
  const float a = pr.a, b = pr.b, c = pr.c;
  const float det = a * c - b * b;
  const float inv2 = 1.0f / (det * det);
  float ga = (-c * c * A.gA + b * c * A.gB - b * b * A.gC) * inv2;
  float gb = (2.f * b * c * A.gA - (a * c + b * b) * A.gB + 2.f * a * b * A.gC) * inv2;
  float gc = (-b * b * A.gA + a * b * A.gB - a * a * A.gC) * inv2;
  {  // coef = sqrt(det0 / (det1 + 1e-6) + 1e-6)
    const float det0r = pr.a0 * pr.c0 - pr.b0 * pr.b0;
    const float det1r = det;
    if (det0r > 1e-6f && det1r > 1e-6f && pr.coef > 0.f) {
      const float dcoef_dr = 0.5f / pr.coef;
      const float den = det1r + 1e-6f;
      const float gdet0 = gcoef * dcoef_dr / den;
      const float gdet1 = -gcoef * dcoef_dr * det0r / (den * den);
      ga += gdet0 * pr.c0 + gdet1 * c;
      gb += gdet0 * (-2.f * pr.b0) + gdet1 * (-2.f * b);
      gc += gdet0 * pr.a0 + gdet1 * a;
    }
  }
  // cov2D = T Sigma T^T
  const Jac j = ewa_jacobian(f, pr.tx, pr.ty, pr.tz);
  const float* c3 = pr.cov3d;
  const float S3[9] = {c3[0], c3[1], c3[2], c3[1], c3[3], c3[4], c3[2], c3[4], c3[5]};
  float v0[3], v1[3];
  for (int r = 0; r < 3; ++r) {
    v0[r] = S3[r * 3 + 0] * j.T0[0] + S3[r * 3 + 1] * j.T0[1] + S3[r * 3 + 2] * j.T0[2];
    v1[r] = S3[r * 3 + 0] * j.T1[0] + S3[r * 3 + 1] * j.T1[1] + S3[r * 3 + 2] * j.T1[2];
  }
  float Gm[9];
  for (int r = 0; r < 3; ++r)
    for (int k = 0; k < 3; ++k)
      Gm[r * 3 + k] = ga * j.T0[r] * j.T0[k] + gb * j.T0[r] * j.T1[k] + gc * j.T1[r] * j.T1[k];
  float gT0[3], gT1[3];
  for (int k = 0; k < 3; ++k) {
    gT0[k] = 2.f * ga * v0[k] + gb * v1[k];
    gT1[k] = 2.f * gc * v1[k] + gb * v0[k];
  }
  float gJ00 = 0.f, gJ02 = 0.f, gJ11 = 0.f, gJ12 = 0.f;
  for (int k = 0; k < 3; ++k) {
    gJ00 += gT0[k] * V[k * 4 + 0];
    gJ02 += gT0[k] * V[k * 4 + 2];
    gJ11 += gT1[k] * V[k * 4 + 1];
    gJ12 += gT1[k] * V[k * 4 + 2];
  }
  const float tz1 = 1.f / pr.tz, tz2 = tz1 * tz1, tz3 = tz2 * tz1;
  const float gtx = j.x_mul * (-j.fx * tz2) * gJ02;
  const float gty = j.y_mul * (-j.fy * tz2) * gJ12;
  float gtz = -j.fx * tz2 * gJ00 - j.fy * tz2 * gJ11 + (2.f * j.fx * j.ux) * tz3 * gJ02 +
              (2.f * j.fy * j.uy) * tz3 * gJ12;
  gtz += A.gdepth;
  for (int k = 0; k < 3; ++k) gp[k] = gtx * V[k * 4 + 0] + gty * V[k * 4 + 1] + gtz * V[k * 4 + 2];
  {  // mean2D (NDC) -> p through the perspective divide
    const float hx = PM[0] * p[0] + PM[4] * p[1] + PM[8] * p[2] + PM[12];
    const float hy = PM[1] * p[0] + PM[5] * p[1] + PM[9] * p[2] + PM[13];
    const float hw = PM[3] * p[0] + PM[7] * p[1] + PM[11] * p[2] + PM[15];
    const float pw = 1.0f / (hw + 0.0000001f);
    const float mul1 = hx * pw * pw, mul2 = hy * pw * pw;
    for (int k = 0; k < 3; ++k)
      gp[k] += (PM[k * 4 + 0] * pw - PM[k * 4 + 3] * mul1) * A.gmx + (PM[k * 4 + 1] * pw - PM[k * 4 + 3] * mul2) * A.gmy;
  }
  {  // Sigma = M M^T, M = R diag(mod s)
    float R[9];
    quat_to_rot(q, R);
    const float mod = f.scale_modifier;
    const float Sv[3] = {mod * s[0], mod * s[1], mod * s[2]};
    float gR[9];
    for (int jj = 0; jj < 3; ++jj) {
      float gS = 0.f;
      for (int r = 0; r < 3; ++r) {
        float gM = 0.f;
        for (int k = 0; k < 3; ++k) gM += (Gm[r * 3 + k] + Gm[k * 3 + r]) * (R[k * 3 + jj] * Sv[jj]);
        gS += gM * R[r * 3 + jj];
        gR[r * 3 + jj] = gM * Sv[jj];
      }
      out.scales[jj] = mod * gS;
    }
    const float r = q[0], x = q[1], y = q[2], z = q[3];
    out.rot[0] = 2.f * (-z * gR[1] + y * gR[2] + z * gR[3] - x * gR[5] - y * gR[6] + x * gR[7]);
    out.rot[1] = 2.f * (y * gR[1] + z * gR[2] + y * gR[3] - 2.f * x * gR[4] - r * gR[5] + z * gR[6] + r * gR[7] -
                        2.f * x * gR[8]);
    out.rot[2] = 2.f * (-2.f * y * gR[0] + x * gR[1] + r * gR[2] + x * gR[3] + z * gR[5] - r * gR[6] + z * gR[7] -
                        2.f * y * gR[8]);
    out.rot[3] = 2.f * (-2.f * z * gR[0] - r * gR[1] + x * gR[2] + r * gR[3] - 2.f * z * gR[4] + y * gR[5] +
                        x * gR[6] + y * gR[7]);
  }
  out.rgb[0] = A.grgb[0]; out.rgb[1] = A.grgb[1]; out.rgb[2] = A.grgb[2];
  if (sh) {
    float rgb[3], dir[3], len;
    unsigned mask;
    const int sk = sh_cm ? 1 : 3, sc = sh_cm ? f.sh_coeffs : 1;
    // center != nullptr (SfgsGaussians.sh_centers): the direction is normalize(p - center) instead of normalize(p - campos)
    sh_to_rgb(f.sh_degree, sh, p, center ? center : f.campos, rgb, &mask, dir, &len, sk, sc, dir_in);
    float Bk[SH_MAX_COEFFS], dBx[SH_MAX_COEFFS], dBy[SH_MAX_COEFFS], dBz[SH_MAX_COEFFS];
    sh_basis(f.sh_degree, dir[0], dir[1], dir[2], Bk);
    sh_basis_grad(f.sh_degree, dir[0], dir[1], dir[2], dBx, dBy, dBz);
    const int M = (f.sh_degree + 1) * (f.sh_degree + 1);
    float gd[3] = {0.f, 0.f, 0.f};
    for (int ch = 0; ch < 3; ++ch) {
      const float gr = ((mask >> ch) & 1u) ? 0.f : A.grgb[ch];
      for (int k = 0; k < f.sh_coeffs; ++k) g_sh[k * sk + ch * sc] = (k < M) ? Bk[k] * gr : 0.f;
      float ax = 0.f, ay = 0.f, az = 0.f;
      for (int k = 0; k < M; ++k) {
        const float cf = sh[k * sk + ch * sc];
        ax += dBx[k] * cf; ay += dBy[k] * cf; az += dBz[k] * cf;
      }
      gd[0] += ax * gr; gd[1] += ay * gr; gd[2] += az * gr;
    }
    if (dir_in) {   // the direction was an input of its own: its gradient goes back to the caller's graph
      g_dir[0] = gd[0]; g_dir[1] = gd[1]; g_dir[2] = gd[2];
    } else {
      const float dot = dir[0] * gd[0] + dir[1] * gd[1] + dir[2] * gd[2];
      gp[0] += (gd[0] - dir[0] * dot) / len;
      gp[1] += (gd[1] - dir[1] * dot) / len;
      gp[2] += (gd[2] - dir[2] * dot) / len;
    }
  }
  out.means3D[0] = gp[0]; out.means3D[1] = gp[1]; out.means3D[2] = gp[2];
}

// from finished 2D gradients (tests/host_check, the oracle-style callers)
SFGS_HD void preprocess_backward_one(const FrameParams& f, const float* p, const float* s, const float* q,
                                     float opacity, const float* sh, const Grad2D& A, GaussGrads& out,
                                     float* g_sh) {
  const Projected pr = project_gaussian(f, p, s, q);  // bit-identical to the forward
  preprocess_backward_pr(f, pr, p, s, q, opacity, sh, A, out, g_sh);
}

// from the summed per-duplicate records (preprocess_bwd_kernel)
SFGS_HD void preprocess_backward_sums(const FrameParams& f, const float* p, const float* s, const float* q,
                                      float opacity, const float* sh, const GradSums& S, GaussGrads& out,
                                      float* g_sh, bool sh_cm = false, const float* dir_in = nullptr,
                                      float* g_dir = nullptr, const float* center = nullptr) {
  const Projected pr = project_gaussian(f, p, s, q);  // bit-identical to the forward
  preprocess_backward_pr(f, pr, p, s, q, opacity, sh, grad2d_from_sums(S, pr, opacity, f.W, f.H), out, g_sh, sh_cm,
                         dir_in, g_dir, center);
}

}  // namespace sfgs
