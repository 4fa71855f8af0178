/*
 * sfgs_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement of the differentiable tile-based Gaussian rasterizer that the reference
 * calls as `diff_gauss.GaussianRasterizer` (gaussian_renderer/__init__.py:132-140). Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this file's library; the
 * product path (skyfall-gs_amd/) never does.
 *
 * PARITY UNPINNED. The source of diff_gauss is an un-vendored git submodule
 * (.gitmodules:1-3, github.com/jayin92/diff-gaussian-rasterization, pinned commit unknown: the
 * reference export has no .git) and the reference ships no tests or golden vectors (SURVEY 4,
 * 8c). This file therefore restates the PUBLISHED algorithm (3DGS EWA splatting + Mip-Splatting
 * 2D filter + depth/alpha accumulation) and pins every convention that IS in
 * the reference tree against it:
 *   - matrix conventions / transposition     scene/cameras.py:62-73, utils/graphics_utils.py:106-126
 *   - quaternion -> rotation                 utils/general_utils.py:78-99
 *   - Sigma = (R S)(R S)^T, 6-float packing  scene/gaussian_model.py:75-79, utils/general_utils.py:64-76,101-110
 *   - SH basis, +0.5, clamp at 0             utils/sh_utils.py:57-112, gaussian_renderer/__init__.py:116-117
 *   - near plane 0.2                         scene/gaussian_model.py:276
 *   - pixel / focal convention               scene/gaussian_model.py:279-286, scene/cameras.py:76-79
 *   - means2D.grad column contract           scene/gaussian_model.py:744-749
 * Those helpers are checked against golden vectors generated from the reference's own Python
 * (tests/golden/make_golden.py). Everything else is marked [UPSTREAM] = public algorithm.
 *
 * Arithmetic: float32 throughout the forward (as the CUDA extension), compiled with
 * -ffp-contract=off so that integer-deciding values (radius, tile rect) are a fixed sequence of
 * IEEE operations. Per-Gaussian gradient sums are accumulated in double.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define TILE 16

typedef struct OrcFrame {
  int32_t H, W;
  float tanfovx, tanfovy, kernel_size, scale_modifier;
  int32_t sh_degree, sh_coeffs, depth_mode;
  const float* subpix; /* [H,W,2] or NULL */
  const float* bg;     /* [3]  */
  const float* view;   /* [16] */
  const float* proj;   /* [16] */
  const float* campos; /* [3]  */
} OrcFrame;

typedef struct OrcGeom {
  float mx, my;          /* pixel-space mean */
  float ca, cb, cc;      /* conic (A,B,C): power = -0.5(A dx^2 + C dy^2) - B dx dy */
  float op;              /* opacity * coef */
  float coef;
  float depth;           /* view-space z */
  float rgb[3];
  uint8_t clamped[3];
  int32_t radius;
  int32_t rminx, rminy, rmaxx, rmaxy;
  float cov3d[6];
  /* values kept for the backward chain */
  float a0, b0, c0;      /* unfiltered 2D covariance */
  float tx, ty, tz;      /* view-space mean */
} OrcGeom;

typedef struct OrcState {
  int32_t N, W, H, TX, TY;
  OrcGeom* g;
  int64_t D;
  int64_t* tile_start;   /* [T+1] */
  uint32_t* list;        /* [D] sorted Gaussian ids, tile-major */
  uint32_t* n_contrib;   /* [P] */
  float* final_T;        /* [P] */
  float* dacc;           /* [P] raw accumulated depth */
} OrcState;

static inline int32_t f2i_sat(float v) {
  /* saturating float->int (what v_cvt_i32_f32 does); callers never pass NaN */
  if (v >= 2147483520.0f) return 2147483647;
  if (v <= -2147483648.0f) return (-2147483647 - 1);
  return (int32_t)v;
}
static inline int32_t imin(int32_t a, int32_t b) { return a < b ? a : b; }
static inline int32_t imax(int32_t a, int32_t b) { return a > b ? a : b; }

/* SH constants: utils/sh_utils.py:26-56 */
static const float SH_C0 = 0.28209479177387814f;
static const float SH_C1 = 0.4886025119029199f;
static const float SH_C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f,
                               -1.0925484305920792f, 0.5462742152960396f};
static const float SH_C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f,
                               0.3731763325901154f, -0.4570457994644658f, 1.445305721320277f,
                               -0.5900435899266435f};

/* quaternion (r,x,y,z) -> R, utils/general_utils.py:78-99 (the op receives an already
 * normalised quaternion, scene/gaussian_model.py:216-217; no normalisation here). */
static void quat_to_R(const float* q, float R[9]) {
  float r = q[0], x = q[1], y = q[2], z = q[3];
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

/* Sigma = M M^T with M = R diag(mod*s); packing xx,xy,xz,yy,yz,zz (utils/general_utils.py:64-76) */
static void cov3d_from_scale_rot(const float* s, float mod, const float* q, float cov[6]) {
  float R[9];
  quat_to_R(q, R);
  float S[3] = {mod * s[0], mod * s[1], mod * s[2]};
  float M[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) M[i * 3 + j] = R[i * 3 + j] * S[j];
  cov[0] = M[0] * M[0] + M[1] * M[1] + M[2] * M[2];
  cov[1] = M[0] * M[3] + M[1] * M[4] + M[2] * M[5];
  cov[2] = M[0] * M[6] + M[1] * M[7] + M[2] * M[8];
  cov[3] = M[3] * M[3] + M[4] * M[4] + M[5] * M[5];
  cov[4] = M[3] * M[6] + M[4] * M[7] + M[5] * M[8];
  cov[5] = M[6] * M[6] + M[7] * M[7] + M[8] * M[8];
}

/* SH -> RGB. utils/sh_utils.py:74-100 (degrees 0..3), +0.5 and clamp at 0 as
 * gaussian_renderer/__init__.py:116-117,124-125. shs layout [M,3] (coefficient-major). */
static void sh_to_rgb(int deg, int M, const float* sh, const float* p, const float* campos,
                      float rgb[3], uint8_t clamped[3]) {
  (void)M;
  float dx = p[0] - campos[0], dy = p[1] - campos[1], dz = p[2] - campos[2];
  float len = sqrtf(dx * dx + dy * dy + dz * dz);
  float x = dx / len, y = dy / len, z = dz / len;
  for (int c = 0; c < 3; ++c) {
    float r = SH_C0 * sh[0 * 3 + c];
    if (deg > 0) {
      r = r - SH_C1 * y * sh[1 * 3 + c] + SH_C1 * z * sh[2 * 3 + c] - SH_C1 * x * sh[3 * 3 + c];
      if (deg > 1) {
        float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
        r = r + SH_C2[0] * xy * sh[4 * 3 + c] + SH_C2[1] * yz * sh[5 * 3 + c] +
            SH_C2[2] * (2.0f * zz - xx - yy) * sh[6 * 3 + c] + SH_C2[3] * xz * sh[7 * 3 + c] +
            SH_C2[4] * (xx - yy) * sh[8 * 3 + c];
        if (deg > 2) {
          r = r + SH_C3[0] * y * (3.0f * xx - yy) * sh[9 * 3 + c] +
              SH_C3[1] * xy * z * sh[10 * 3 + c] +
              SH_C3[2] * y * (4.0f * zz - xx - yy) * sh[11 * 3 + c] +
              SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * sh[12 * 3 + c] +
              SH_C3[4] * x * (4.0f * zz - xx - yy) * sh[13 * 3 + c] +
              SH_C3[5] * z * (xx - yy) * sh[14 * 3 + c] +
              SH_C3[6] * x * (xx - 3.0f * yy) * sh[15 * 3 + c];
        }
      }
    }
    r += 0.5f;
    clamped[c] = (r < 0.0f);
    rgb[c] = r < 0.0f ? 0.0f : r;
  }
}

/* ---- A.2 preprocess ---------------------------------------------------------------------- */
static void preprocess_one(const OrcFrame* f, int TX, int TY, const float* p, const float* s,
                           const float* q, float opacity, const float* color, const float* sh,
                           OrcGeom* o) {
  const float* V = f->view;
  const float* PM = f->proj;
  memset(o, 0, sizeof(*o));
  /* p_view = [p,1] * viewmatrix (scene/cameras.py:62-64) */
  float tx = V[0] * p[0] + V[4] * p[1] + V[8] * p[2] + V[12];
  float ty = V[1] * p[0] + V[5] * p[1] + V[9] * p[2] + V[13];
  float tz = V[2] * p[0] + V[6] * p[1] + V[10] * p[2] + V[14];
  o->tx = tx; o->ty = ty; o->tz = tz;
  if (!(tz > 0.2f)) return; /* near cull, scene/gaussian_model.py:276 mirrors it */

  float hx = PM[0] * p[0] + PM[4] * p[1] + PM[8] * p[2] + PM[12];
  float hy = PM[1] * p[0] + PM[5] * p[1] + PM[9] * p[2] + PM[13];
  float hw = PM[3] * p[0] + PM[7] * p[1] + PM[11] * p[2] + PM[15];
  float pw = 1.0f / (hw + 0.0000001f);
  float ndcx = hx * pw, ndcy = hy * pw;

  cov3d_from_scale_rot(s, f->scale_modifier, q, o->cov3d);
  const float* c3 = o->cov3d;

  /* [UPSTREAM] EWA projection with the 1.3*tanfov guard band */
  float limx = 1.3f * f->tanfovx, limy = 1.3f * f->tanfovy;
  float txtz = tx / tz, tytz = ty / tz;
  float ux = fminf(limx, fmaxf(-limx, txtz)) * tz;
  float uy = fminf(limy, fmaxf(-limy, tytz)) * tz;
  float fx = (float)f->W / (2.0f * f->tanfovx); /* scene/cameras.py:76-79 */
  float fy = (float)f->H / (2.0f * f->tanfovy);
  float J00 = fx / tz, J02 = -(fx * ux) / (tz * tz);
  float J11 = fy / tz, J12 = -(fy * uy) / (tz * tz);
  float T0[3], T1[3];
  for (int b = 0; b < 3; ++b) {
    T0[b] = J00 * V[b * 4 + 0] + J02 * V[b * 4 + 2];
    T1[b] = J11 * V[b * 4 + 1] + J12 * V[b * 4 + 2];
  }
  float v0[3], v1[3];
  v0[0] = c3[0] * T0[0] + c3[1] * T0[1] + c3[2] * T0[2];
  v0[1] = c3[1] * T0[0] + c3[3] * T0[1] + c3[4] * T0[2];
  v0[2] = c3[2] * T0[0] + c3[4] * T0[1] + c3[5] * T0[2];
  v1[0] = c3[0] * T1[0] + c3[1] * T1[1] + c3[2] * T1[2];
  v1[1] = c3[1] * T1[0] + c3[3] * T1[1] + c3[4] * T1[2];
  v1[2] = c3[2] * T1[0] + c3[4] * T1[1] + c3[5] * T1[2];
  float a0 = T0[0] * v0[0] + T0[1] * v0[1] + T0[2] * v0[2];
  float b0 = T0[0] * v1[0] + T0[1] * v1[1] + T0[2] * v1[2];
  float c0 = T1[0] * v1[0] + T1[1] * v1[1] + T1[2] * v1[2];
  o->a0 = a0; o->b0 = b0; o->c0 = c0;

  /* [UPSTREAM] Mip-Splatting 2D filter with opacity compensation (kernel_size from
   * arguments/__init__.py:111 via gaussian_renderer/__init__.py:45) */
  float ks = f->kernel_size;
  float det0 = fmaxf(1e-6f, a0 * c0 - b0 * b0);
  float a = a0 + ks, c = c0 + ks, b = b0;
  float det1 = fmaxf(1e-6f, a * c - b * b);
  float coef = sqrtf(det0 / (det1 + 1e-6f) + 1e-6f);
  if (det0 <= 1e-6f || det1 <= 1e-6f) coef = 0.0f;

  float det = a * c - b * b;
  if (det == 0.0f) return;
  float det_inv = 1.0f / det;
  float cA = c * det_inv, cB = -b * det_inv, cC = a * det_inv;
  float mid = 0.5f * (a + c);
  float disc = sqrtf(fmaxf(0.1f, mid * mid - det));
  float l1 = mid + disc, l2 = mid - disc;
  float radf = ceilf(3.0f * sqrtf(fmaxf(l1, l2)));
  /* pixel coords of NDC: ((v+1)*S-1)/2  [UPSTREAM ndc2Pix] */
  float mx = ((ndcx + 1.0f) * (float)f->W - 1.0f) * 0.5f;
  float my = ((ndcy + 1.0f) * (float)f->H - 1.0f) * 0.5f;
  if (!(isfinite(mx) && isfinite(my) && isfinite(radf))) return; /* oracle decision: NaN/Inf culled */

  int32_t rminx = imin(TX, imax(0, f2i_sat((mx - radf) / (float)TILE)));
  int32_t rminy = imin(TY, imax(0, f2i_sat((my - radf) / (float)TILE)));
  int32_t rmaxx = imin(TX, imax(0, f2i_sat((mx + radf + (float)(TILE - 1)) / (float)TILE)));
  int32_t rmaxy = imin(TY, imax(0, f2i_sat((my + radf + (float)(TILE - 1)) / (float)TILE)));
  if ((rmaxx - rminx) * (rmaxy - rminy) == 0) return;

  if (color) {
    o->rgb[0] = color[0]; o->rgb[1] = color[1]; o->rgb[2] = color[2];
  } else {
    sh_to_rgb(f->sh_degree, f->sh_coeffs, sh, p, f->campos, o->rgb, o->clamped);
  }
  o->mx = mx; o->my = my;
  o->ca = cA; o->cb = cB; o->cc = cC;
  o->coef = coef;
  o->op = opacity * coef;
  o->depth = tz;
  o->radius = f2i_sat(radf);
  o->rminx = rminx; o->rminy = rminy; o->rmaxx = rmaxx; o->rmaxy = rmaxy;
}

typedef struct KeyId { uint32_t key; uint32_t id; } KeyId;
static int cmp_keyid(const void* pa, const void* pb) {
  const KeyId* a = (const KeyId*)pa;
  const KeyId* b = (const KeyId*)pb;
  if (a->key != b->key) return a->key < b->key ? -1 : 1;
  if (a->id != b->id) return a->id < b->id ? -1 : 1;
  return 0;
}

void orc_free(OrcState* st) {
  if (!st) return;
  free(st->g); free(st->tile_start); free(st->list); free(st->n_contrib); free(st->final_T);
  free(st->dacc); free(st);
}

/* ---- forward: A.2 - A.4 ------------------------------------------------------------------ */
/* orc_forward_rows: the same algorithm restricted to the 16x16 tile rows [row0, row1) -- every Gaussian is projected
 * (radii are complete), only the tiles of those rows get lists and pixels; the other pixels keep what the output arrays
 * held. Lets a test check two bands of a 16 M-Gaussian joint frame (BASELINE.json configs[4]) in seconds. D counts the
 * band's duplicates. orc_forward = all rows. */
OrcState* orc_forward_rows(const OrcFrame* f, int32_t N, const float* means3D, const float* scales,
                           const float* rots, const float* opac, const float* colors, const float* shs,
                           float* out_color, float* out_depth, float* out_alpha, int32_t* radii, int32_t row0,
                           int32_t row1);

OrcState* orc_forward(const OrcFrame* f, int32_t N, const float* means3D, const float* scales,
                      const float* rots, const float* opac, const float* colors, const float* shs,
                      float* out_color, float* out_depth, float* out_alpha, int32_t* radii) {
  return orc_forward_rows(f, N, means3D, scales, rots, opac, colors, shs, out_color, out_depth, out_alpha, radii, 0,
                          (f->H + TILE - 1) / TILE);
}

OrcState* orc_forward_rows(const OrcFrame* f, int32_t N, const float* means3D, const float* scales,
                           const float* rots, const float* opac, const float* colors, const float* shs,
                           float* out_color, float* out_depth, float* out_alpha, int32_t* radii, int32_t row0,
                           int32_t row1) {
  const int W = f->W, H = f->H;
  const int TX = (W + TILE - 1) / TILE, TY = (H + TILE - 1) / TILE;
  if (row0 < 0) row0 = 0;
  if (row1 > TY) row1 = TY;
  const int64_t T = (int64_t)TX * TY, P = (int64_t)W * H;
  OrcState* st = (OrcState*)calloc(1, sizeof(OrcState));
  st->N = N; st->W = W; st->H = H; st->TX = TX; st->TY = TY;
  st->g = (OrcGeom*)calloc((size_t)(N > 0 ? N : 1), sizeof(OrcGeom));
  st->tile_start = (int64_t*)calloc((size_t)T + 1, sizeof(int64_t));
  st->n_contrib = (uint32_t*)calloc((size_t)P, sizeof(uint32_t));
  st->final_T = (float*)calloc((size_t)P, sizeof(float));
  st->dacc = (float*)calloc((size_t)P, sizeof(float));

#pragma omp parallel for schedule(static)
  for (int32_t i = 0; i < N; ++i) {
    preprocess_one(f, TX, TY, means3D + 3 * (size_t)i, scales + 3 * (size_t)i, rots + 4 * (size_t)i,
                   opac[i], colors ? colors + 3 * (size_t)i : NULL,
                   shs ? shs + 3 * (size_t)f->sh_coeffs * i : NULL, &st->g[i]);
    radii[i] = st->g[i].radius;
  }

  /* A.3 binning: count, scan, fill in index order, sort each tile by (depth bits, id) */
  int64_t* cnt = (int64_t*)calloc((size_t)T + 1, sizeof(int64_t));
  for (int32_t i = 0; i < N; ++i) {
    const OrcGeom* g = &st->g[i];
    if (g->radius <= 0) continue;
    for (int y = imax(g->rminy, row0); y < imin(g->rmaxy, row1); ++y)
      for (int x = g->rminx; x < g->rmaxx; ++x) cnt[(int64_t)y * TX + x]++;
  }
  int64_t D = 0;
  for (int64_t t = 0; t < T; ++t) { st->tile_start[t] = D; D += cnt[t]; }
  st->tile_start[T] = D;
  st->D = D;
  KeyId* kv = (KeyId*)malloc(sizeof(KeyId) * (size_t)(D > 0 ? D : 1));
  memset(cnt, 0, sizeof(int64_t) * ((size_t)T + 1));
  for (int32_t i = 0; i < N; ++i) {
    const OrcGeom* g = &st->g[i];
    if (g->radius <= 0) continue;
    uint32_t kb; memcpy(&kb, &g->depth, 4);
    for (int y = imax(g->rminy, row0); y < imin(g->rmaxy, row1); ++y)
      for (int x = g->rminx; x < g->rmaxx; ++x) {
        int64_t t = (int64_t)y * TX + x;
        int64_t pos = st->tile_start[t] + cnt[t]++;
        kv[pos].key = kb; kv[pos].id = (uint32_t)i;
      }
  }
  free(cnt);
#pragma omp parallel for schedule(dynamic, 16)
  for (int64_t t = 0; t < T; ++t) {
    int64_t s = st->tile_start[t], e = st->tile_start[t + 1];
    if (e - s > 1) qsort(kv + s, (size_t)(e - s), sizeof(KeyId), cmp_keyid);
  }
  st->list = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)(D > 0 ? D : 1));
  for (int64_t k = 0; k < D; ++k) st->list[k] = kv[k].id;
  free(kv);

  /* A.4 composite */
  const float bg0 = f->bg[0], bg1 = f->bg[1], bg2 = f->bg[2];
#pragma omp parallel for schedule(dynamic, 4)
  for (int64_t t = (int64_t)row0 * TX; t < (int64_t)row1 * TX; ++t) {
    int tx0 = (int)(t % TX) * TILE, ty0 = (int)(t / TX) * TILE;
    int64_t s = st->tile_start[t], e = st->tile_start[t + 1];
    for (int py = ty0; py < ty0 + TILE && py < H; ++py)
      for (int px = tx0; px < tx0 + TILE && px < W; ++px) {
        int64_t pix = (int64_t)py * W + px;
        float sx = (float)px, sy = (float)py;
        if (f->subpix) { sx += f->subpix[pix * 2 + 0]; sy += f->subpix[pix * 2 + 1]; }
        float Tr = 1.0f, C0 = 0.f, C1 = 0.f, C2 = 0.f, Dacc = 0.f;
        uint32_t contributor = 0, last = 0;
        for (int64_t k = s; k < e; ++k) {
          contributor++;
          const OrcGeom* g = &st->g[st->list[k]];
          float dx = g->mx - sx, dy = g->my - sy;
          float power = -0.5f * (g->ca * dx * dx + g->cc * dy * dy) - g->cb * dx * dy;
          if (power > 0.0f) continue;
          float alpha = fminf(0.99f, g->op * expf(power));
          if (alpha < 1.0f / 255.0f) continue;
          float test_T = Tr * (1.0f - alpha);
          if (test_T < 0.0001f) break;
          float w = alpha * Tr;
          C0 += g->rgb[0] * w; C1 += g->rgb[1] * w; C2 += g->rgb[2] * w;
          Dacc += g->depth * w;
          Tr = test_T;
          last = contributor;
        }
        st->final_T[pix] = Tr;
        st->n_contrib[pix] = last;
        st->dacc[pix] = Dacc;
        out_color[0 * P + pix] = C0 + Tr * bg0;
        out_color[1 * P + pix] = C1 + Tr * bg1;
        out_color[2 * P + pix] = C2 + Tr * bg2;
        float a = 1.0f - Tr;
        out_alpha[pix] = a;
        out_depth[pix] = (f->depth_mode == 0) ? Dacc / a : Dacc;
      }
  }
  return st;
}

/* Test diagnostics (not part of the restated algorithm): how close does ANY per-pixel decision of the compositing loop
 * above come to its threshold? Re-walks the loop of orc_forward and returns the smallest relative distances
 *   out[0]: |alpha * 255 - 1|          over the pairs that reach the alpha test  (skip if alpha < 1/255)
 *   out[1]: |test_T / 1e-4 - 1|        over the pairs that reach the transmittance test (stop if test_T < 1e-4)
 *   out[2]: |power|                    over the pairs with |power| < 1e-3 (skip if power > 0)
 * A float32 implementation that evaluates alpha with another exponential (v_exp_f32 on 2^x vs expf) differs from this one
 * by a few ulps: a decision closer than that to its threshold may legitimately fall the other way, and the gradients of
 * the splat concerned then differ at the 1/255 level. tests/ uses this to tell such a flip from a defect. */
void orc_decision_margins(const OrcState* st, const OrcFrame* f, double out[3]) {
  const int W = st->W, H = st->H, TX = st->TX;
  const int64_t T = (int64_t)st->TX * st->TY;
  double m_alpha = 1e30, m_T = 1e30, m_pow = 1e30;
  for (int64_t t = 0; t < T; ++t) {
    int tx0 = (int)(t % TX) * TILE, ty0 = (int)(t / TX) * TILE;
    int64_t s = st->tile_start[t], e = st->tile_start[t + 1];
    for (int py = ty0; py < ty0 + TILE && py < H; ++py)
      for (int px = tx0; px < tx0 + TILE && px < W; ++px) {
        int64_t pix = (int64_t)py * W + px;
        float sx = (float)px, sy = (float)py;
        if (f->subpix) { sx += f->subpix[pix * 2 + 0]; sy += f->subpix[pix * 2 + 1]; }
        float Tr = 1.0f;
        for (int64_t k = s; k < e; ++k) {
          const OrcGeom* g = &st->g[st->list[k]];
          float dx = g->mx - sx, dy = g->my - sy;
          float power = -0.5f * (g->ca * dx * dx + g->cc * dy * dy) - g->cb * dx * dy;
          if (fabsf(power) < 1e-3f && fabs((double)power) < m_pow) m_pow = fabs((double)power);
          if (power > 0.0f) continue;
          float alpha = fminf(0.99f, g->op * expf(power));
          double da = fabs((double)alpha * 255.0 - 1.0);
          if (da < m_alpha) m_alpha = da;
          if (alpha < 1.0f / 255.0f) continue;
          float test_T = Tr * (1.0f - alpha);
          double dt = fabs((double)test_T / 1e-4 - 1.0);
          if (dt < m_T) m_T = dt;
          if (test_T < 0.0001f) break;
          Tr = test_T;
        }
      }
  }
  out[0] = m_alpha; out[1] = m_T; out[2] = m_pow;
}

/* introspection for tests */
void orc_get_counts(const OrcState* st, int64_t out[4]) {
  int64_t nvis = 0, maxlen = 0;
  for (int32_t i = 0; i < st->N; ++i) nvis += st->g[i].radius > 0;
  int64_t T = (int64_t)st->TX * st->TY;
  for (int64_t t = 0; t < T; ++t) {
    int64_t l = st->tile_start[t + 1] - st->tile_start[t];
    if (l > maxlen) maxlen = l;
  }
  out[0] = st->D; out[1] = nvis; out[2] = maxlen; out[3] = T;
}
void orc_get_tiles_touched(const OrcState* st, int32_t* out) {
  for (int32_t i = 0; i < st->N; ++i) {
    const OrcGeom* g = &st->g[i];
    out[i] = g->radius > 0 ? (g->rmaxx - g->rminx) * (g->rmaxy - g->rminy) : 0;
  }
}
void orc_get_n_contrib(const OrcState* st, uint32_t* out) {
  memcpy(out, st->n_contrib, sizeof(uint32_t) * (size_t)st->W * st->H);
}
void orc_get_tile_lists(const OrcState* st, int64_t* tile_start, uint32_t* list) {
  int64_t T = (int64_t)st->TX * st->TY;
  memcpy(tile_start, st->tile_start, sizeof(int64_t) * ((size_t)T + 1));
  memcpy(list, st->list, sizeof(uint32_t) * (size_t)st->D);
}
/* per-Gaussian 2D state: [N,12] = mx,my,A,B,C,op,depth,r,g,b,coef,radius */
void orc_get_geom(const OrcState* st, float* out) {
  for (int32_t i = 0; i < st->N; ++i) {
    const OrcGeom* g = &st->g[i];
    float* o = out + 12 * (size_t)i;
    o[0] = g->mx; o[1] = g->my; o[2] = g->ca; o[3] = g->cb; o[4] = g->cc; o[5] = g->op;
    o[6] = g->depth; o[7] = g->rgb[0]; o[8] = g->rgb[1]; o[9] = g->rgb[2]; o[10] = g->coef;
    o[11] = (float)g->radius;
  }
}

/* ---- backward: A.6 ----------------------------------------------------------------------- */
typedef struct Acc2D { double gmx, gmy, absx, absy, gA, gB, gC, gop, grgb[3], gdepth; } Acc2D;

static inline void atomic_add_d(double* p, double v) {
#pragma omp atomic
  *p += v;
}

int orc_backward(const OrcState* st, const OrcFrame* f, const float* means3D, const float* scales,
                 const float* rots, const float* opac, const float* colors, const float* shs,
                 const float* out_alpha, const float* dL_dcolor, const float* dL_ddepth,
                 const float* dL_dalpha, float* g_means3D, float* g_means2D, float* g_scales,
                 float* g_rots, float* g_opac, float* g_colors, float* g_shs) {
  (void)colors; (void)out_alpha;
  const int N = st->N, W = st->W, H = st->H, TX = st->TX, TY = st->TY;
  const int64_t T = (int64_t)TX * TY, P = (int64_t)W * H;
  Acc2D* acc = (Acc2D*)calloc((size_t)(N > 0 ? N : 1), sizeof(Acc2D));
  const float bg[3] = {f->bg[0], f->bg[1], f->bg[2]};
  const float ddelx_dx = 0.5f * (float)W, ddely_dy = 0.5f * (float)H;

#pragma omp parallel for schedule(dynamic, 4)
  for (int64_t t = 0; t < T; ++t) {
    int tx0 = (int)(t % TX) * TILE, ty0 = (int)(t / TX) * TILE;
    int64_t s = st->tile_start[t], e = st->tile_start[t + 1];
    int64_t L = e - s;
    if (L == 0) continue;
    Acc2D* loc = (Acc2D*)calloc((size_t)L, sizeof(Acc2D));
    for (int py = ty0; py < ty0 + TILE && py < H; ++py)
      for (int px = tx0; px < tx0 + TILE && px < W; ++px) {
        int64_t pix = (int64_t)py * W + px;
        uint32_t last = st->n_contrib[pix];
        if (last == 0) continue;
        float sx = (float)px, sy = (float)py;
        if (f->subpix) { sx += f->subpix[pix * 2 + 0]; sy += f->subpix[pix * 2 + 1]; }
        const float T_final = st->final_T[pix];
        /* upstream gradients of the 5 accumulated channels: rgb, raw depth, alpha */
        float gch[5];
        gch[0] = dL_dcolor ? dL_dcolor[0 * P + pix] : 0.f;
        gch[1] = dL_dcolor ? dL_dcolor[1 * P + pix] : 0.f;
        gch[2] = dL_dcolor ? dL_dcolor[2 * P + pix] : 0.f;
        float gdep = dL_ddepth ? dL_ddepth[pix] : 0.f;
        float galp = dL_dalpha ? dL_dalpha[pix] : 0.f;
        if (f->depth_mode == 0) {
          /* depth = Dacc / a, a = 1 - T_final: fold into raw accumulators (A.6) */
          float a = 1.0f - T_final;
          float Dacc = st->dacc[pix];
          gch[3] = gdep / a;
          gch[4] = galp - gdep * Dacc / (a * a);
        } else {
          gch[3] = gdep;
          gch[4] = galp;
        }
        const float bg_dot = bg[0] * gch[0] + bg[1] * gch[1] + bg[2] * gch[2];
        float Tr = T_final;
        float accum[5] = {0, 0, 0, 0, 0}, lastv[5] = {0, 0, 0, 0, 0};
        float last_alpha = 0.f;
        for (int64_t k = s + (int64_t)last - 1; k >= s; --k) {
          const uint32_t id = st->list[k];
          const OrcGeom* g = &st->g[id];
          float dx = g->mx - sx, dy = g->my - sy;
          float power = -0.5f * (g->ca * dx * dx + g->cc * dy * dy) - g->cb * dx * dy;
          if (power > 0.0f) continue;
          float G = expf(power);
          float alpha = fminf(0.99f, g->op * G);
          if (alpha < 1.0f / 255.0f) continue;
          Tr = Tr / (1.0f - alpha);
          const float w = alpha * Tr;
          float val[5] = {g->rgb[0], g->rgb[1], g->rgb[2], g->depth, 1.0f};
          float dL_dalpha_i = 0.f;
          Acc2D* A = &loc[k - s];
          for (int ch = 0; ch < 5; ++ch) {
            accum[ch] = last_alpha * lastv[ch] + (1.f - last_alpha) * accum[ch];
            lastv[ch] = val[ch];
            dL_dalpha_i += (val[ch] - accum[ch]) * gch[ch];
          }
          A->grgb[0] += (double)(w * gch[0]);
          A->grgb[1] += (double)(w * gch[1]);
          A->grgb[2] += (double)(w * gch[2]);
          A->gdepth += (double)(w * gch[3]);
          dL_dalpha_i *= Tr;
          last_alpha = alpha;
          dL_dalpha_i += (-T_final / (1.f - alpha)) * bg_dot;
          /* [UPSTREAM] the min(0.99,.) clamp is ignored in the derivative */
          const float dL_dG = g->op * dL_dalpha_i;
          const float gdx = G * dx, gdy = G * dy;
          const float dG_ddelx = -gdx * g->ca - gdy * g->cb;
          const float dG_ddely = -gdy * g->cc - gdx * g->cb;
          const float gx = dL_dG * dG_ddelx * ddelx_dx;
          const float gy = dL_dG * dG_ddely * ddely_dy;
          A->gmx += (double)gx; A->gmy += (double)gy;
          A->absx += (double)fabsf(gx); A->absy += (double)fabsf(gy);
          A->gA += (double)(-0.5f * gdx * dx * dL_dG);
          A->gB += (double)(-gdx * dy * dL_dG);
          A->gC += (double)(-0.5f * gdy * dy * dL_dG);
          A->gop += (double)(G * dL_dalpha_i);
        }
      }
    for (int64_t k = 0; k < L; ++k) {
      Acc2D* dst = &acc[st->list[s + k]];
      const Acc2D* src = &loc[k];
      atomic_add_d(&dst->gmx, src->gmx); atomic_add_d(&dst->gmy, src->gmy);
      atomic_add_d(&dst->absx, src->absx); atomic_add_d(&dst->absy, src->absy);
      atomic_add_d(&dst->gA, src->gA); atomic_add_d(&dst->gB, src->gB);
      atomic_add_d(&dst->gC, src->gC); atomic_add_d(&dst->gop, src->gop);
      atomic_add_d(&dst->grgb[0], src->grgb[0]); atomic_add_d(&dst->grgb[1], src->grgb[1]);
      atomic_add_d(&dst->grgb[2], src->grgb[2]); atomic_add_d(&dst->gdepth, src->gdepth);
    }
    free(loc);
  }

  /* per-Gaussian chain rule */
  const float* V = f->view;
  const float* PM = f->proj;
  const int M = f->sh_coeffs;
#pragma omp parallel for schedule(static)
  for (int32_t i = 0; i < N; ++i) {
    const OrcGeom* g = &st->g[i];
    const Acc2D* A = &acc[i];
    float* gm3 = g_means3D + 3 * (size_t)i;
    float* gm2 = g_means2D + 3 * (size_t)i;
    float* gs = g_scales + 3 * (size_t)i;
    float* gq = g_rots + 4 * (size_t)i;
    gm3[0] = gm3[1] = gm3[2] = 0.f;
    gm2[0] = gm2[1] = gm2[2] = 0.f;
    gs[0] = gs[1] = gs[2] = 0.f;
    gq[0] = gq[1] = gq[2] = gq[3] = 0.f;
    g_opac[i] = 0.f;
    if (g_colors) { g_colors[3 * (size_t)i] = g_colors[3 * (size_t)i + 1] = g_colors[3 * (size_t)i + 2] = 0.f; }
    if (g_shs) memset(g_shs + 3 * (size_t)M * i, 0, sizeof(float) * 3 * (size_t)M);
    if (g->radius <= 0) continue;

    const float* p = means3D + 3 * (size_t)i;
    const float* s = scales + 3 * (size_t)i;
    const float* q = rots + 4 * (size_t)i;
    float gA = (float)A->gA, gB = (float)A->gB, gC = (float)A->gC;
    float gop_hat = (float)A->gop;

    /* means2D.grad contract: scene/gaussian_model.py:744-749 (col 2 = abs magnitude) */
    gm2[0] = (float)A->gmx; gm2[1] = (float)A->gmy;
    gm2[2] = (float)sqrt(A->absx * A->absx + A->absy * A->absy);

    /* opacity' = opacity * coef */
    g_opac[i] = gop_hat * g->coef;
    float gcoef = gop_hat * opac[i];

    /* conic -> filtered cov (a,b,c) */
    float ks = f->kernel_size;
    float a = g->a0 + ks, b = g->b0, c = g->c0 + ks;
    float det = a * c - b * b;
    float inv2 = 1.0f / (det * det);
    float ga = (-c * c * gA + b * c * gB - b * b * gC) * inv2;
    float gb = (2.f * b * c * gA - (a * c + b * b) * gB + 2.f * a * b * gC) * inv2;
    float gc = (-b * b * gA + a * b * gB - a * a * gC) * inv2;
    /* coef -> (a0,b0,c0) */
    {
      float det0r = g->a0 * g->c0 - g->b0 * g->b0;
      float det1r = a * c - b * b;
      if (det0r > 1e-6f && det1r > 1e-6f && g->coef > 0.f) {
        float dcoef_dr = 0.5f / g->coef;
        float den = det1r + 1e-6f;
        float gdet0 = gcoef * dcoef_dr / den;
        float gdet1 = -gcoef * dcoef_dr * det0r / (den * den);
        ga += gdet0 * g->c0 + gdet1 * c;
        gb += gdet0 * (-2.f * g->b0) + gdet1 * (-2.f * b);
        gc += gdet0 * g->a0 + gdet1 * a;
      }
    }
    /* recompute T rows */
    float tx = g->tx, ty = g->ty, tz = g->tz;
    float limx = 1.3f * f->tanfovx, limy = 1.3f * f->tanfovy;
    float txtz = tx / tz, tytz = ty / tz;
    float ux = fminf(limx, fmaxf(-limx, txtz)) * tz;
    float uy = fminf(limy, fmaxf(-limy, tytz)) * tz;
    float x_mul = (txtz < -limx || txtz > limx) ? 0.f : 1.f;
    float y_mul = (tytz < -limy || tytz > limy) ? 0.f : 1.f;
    float fx = (float)W / (2.0f * f->tanfovx), fy = (float)H / (2.0f * f->tanfovy);
    float J00 = fx / tz, J02 = -(fx * ux) / (tz * tz), J11 = fy / tz, J12 = -(fy * uy) / (tz * tz);
    float T0[3], T1[3];
    for (int k = 0; k < 3; ++k) {
      T0[k] = J00 * V[k * 4 + 0] + J02 * V[k * 4 + 2];
      T1[k] = J11 * V[k * 4 + 1] + J12 * V[k * 4 + 2];
    }
    const float* c3 = g->cov3d;
    float S3[9] = {c3[0], c3[1], c3[2], c3[1], c3[3], c3[4], c3[2], c3[4], c3[5]};
    float v0[3], v1[3];
    for (int r = 0; r < 3; ++r) {
      v0[r] = S3[r * 3 + 0] * T0[0] + S3[r * 3 + 1] * T0[1] + S3[r * 3 + 2] * T0[2];
      v1[r] = S3[r * 3 + 0] * T1[0] + S3[r * 3 + 1] * T1[1] + S3[r * 3 + 2] * T1[2];
    }
    /* dL/dSigma3 (full matrix G): a0 = T0^T S T0, b0 = T0^T S T1, c0 = T1^T S T1 */
    float Gm[9];
    for (int r = 0; r < 3; ++r)
      for (int cc2 = 0; cc2 < 3; ++cc2)
        Gm[r * 3 + cc2] = ga * T0[r] * T0[cc2] + gb * T0[r] * T1[cc2] + gc * T1[r] * T1[cc2];
    float gT0[3], gT1[3];
    for (int k = 0; k < 3; ++k) {
      gT0[k] = 2.f * ga * v0[k] + gb * v1[k];
      gT1[k] = 2.f * gc * v1[k] + gb * v0[k];
    }
    float gJ00 = 0, gJ02 = 0, gJ11 = 0, gJ12 = 0;
    for (int k = 0; k < 3; ++k) {
      gJ00 += gT0[k] * V[k * 4 + 0];
      gJ02 += gT0[k] * V[k * 4 + 2];
      gJ11 += gT1[k] * V[k * 4 + 1];
      gJ12 += gT1[k] * V[k * 4 + 2];
    }
    /* [UPSTREAM] J -> t, with the clamp mask applied to tx,ty only (public 3DGS behaviour) */
    float tz1 = 1.f / tz, tz2 = tz1 * tz1, tz3 = tz2 * tz1;
    float gtx = x_mul * (-fx * tz2) * gJ02;
    float gty = y_mul * (-fy * tz2) * gJ12;
    float gtz = -fx * tz2 * gJ00 - fy * tz2 * gJ11 + (2.f * fx * ux) * tz3 * gJ02 +
                (2.f * fy * uy) * tz3 * gJ12;
    gtz += (float)A->gdepth; /* depth_i = tz */
    /* t = W2C p + tr  ->  dL/dp_k = sum_a gt_a * V[k*4+a] */
    float gp[3];
    for (int k = 0; k < 3; ++k) gp[k] = gtx * V[k * 4 + 0] + gty * V[k * 4 + 1] + gtz * V[k * 4 + 2];
    /* mean2D (NDC units) -> p through the perspective divide */
    {
      float hx = PM[0] * p[0] + PM[4] * p[1] + PM[8] * p[2] + PM[12];
      float hy = PM[1] * p[0] + PM[5] * p[1] + PM[9] * p[2] + PM[13];
      float hw = PM[3] * p[0] + PM[7] * p[1] + PM[11] * p[2] + PM[15];
      float pw = 1.0f / (hw + 0.0000001f);
      float mul1 = hx * pw * pw, mul2 = hy * pw * pw;
      for (int k = 0; k < 3; ++k) {
        gp[k] += (PM[k * 4 + 0] * pw - PM[k * 4 + 3] * mul1) * gm2[0] +
                 (PM[k * 4 + 1] * pw - PM[k * 4 + 3] * mul2) * gm2[1];
      }
    }
    /* Sigma3 = M M^T, M = R diag(mod*s):  dL/dM = (G + G^T) M */
    {
      float R[9];
      quat_to_R(q, R);
      float mod = f->scale_modifier;
      float Sv[3] = {mod * s[0], mod * s[1], mod * s[2]};
      float Mm[9], gM[9];
      for (int r = 0; r < 3; ++r)
        for (int j = 0; j < 3; ++j) Mm[r * 3 + j] = R[r * 3 + j] * Sv[j];
      for (int r = 0; r < 3; ++r)
        for (int j = 0; j < 3; ++j) {
          float acc2 = 0.f;
          for (int k = 0; k < 3; ++k) acc2 += (Gm[r * 3 + k] + Gm[k * 3 + r]) * Mm[k * 3 + j];
          gM[r * 3 + j] = acc2;
        }
      float gR[9];
      for (int j = 0; j < 3; ++j) {
        float gS = 0.f;
        for (int r = 0; r < 3; ++r) {
          gS += gM[r * 3 + j] * R[r * 3 + j];
          gR[r * 3 + j] = gM[r * 3 + j] * Sv[j];
        }
        gs[j] = mod * gS;
      }
      float r = q[0], x = q[1], y = q[2], z = q[3];
      gq[0] = 2.f * (-z * gR[1] + y * gR[2] + z * gR[3] - x * gR[5] - y * gR[6] + x * gR[7]);
      gq[1] = 2.f * (y * gR[1] + z * gR[2] + y * gR[3] - 2.f * x * gR[4] - r * gR[5] + z * gR[6] +
                     r * gR[7] - 2.f * x * gR[8]);
      gq[2] = 2.f * (-2.f * y * gR[0] + x * gR[1] + r * gR[2] + x * gR[3] + z * gR[5] - r * gR[6] +
                     z * gR[7] - 2.f * y * gR[8]);
      gq[3] = 2.f * (-2.f * z * gR[0] - r * gR[1] + x * gR[2] + r * gR[3] - 2.f * z * gR[4] +
                     y * gR[5] + x * gR[6] + y * gR[7]);
    }
    /* colour */
    if (g_colors) {
      g_colors[3 * (size_t)i + 0] = (float)A->grgb[0];
      g_colors[3 * (size_t)i + 1] = (float)A->grgb[1];
      g_colors[3 * (size_t)i + 2] = (float)A->grgb[2];
    } else if (g_shs) {
      /* SH backward: d rgb / d sh = basis; d rgb / d dir -> p */
      const float* sh = shs + 3 * (size_t)M * i;
      float* gsh = g_shs + 3 * (size_t)M * i;
      float ddx = p[0] - f->campos[0], ddy = p[1] - f->campos[1], ddz = p[2] - f->campos[2];
      float len = sqrtf(ddx * ddx + ddy * ddy + ddz * ddz);
      float x = ddx / len, y = ddy / len, z = ddz / len;
      float gdirx = 0.f, gdiry = 0.f, gdirz = 0.f;
      int deg = f->sh_degree;
      for (int ch = 0; ch < 3; ++ch) {
        float gr = g->clamped[ch] ? 0.f : (float)A->grgb[ch];
        float dRdx = 0.f, dRdy = 0.f, dRdz = 0.f;
        gsh[0 * 3 + ch] = SH_C0 * gr;
        if (deg > 0) {
          gsh[1 * 3 + ch] = -SH_C1 * y * gr;
          gsh[2 * 3 + ch] = SH_C1 * z * gr;
          gsh[3 * 3 + ch] = -SH_C1 * x * gr;
          dRdx = -SH_C1 * sh[3 * 3 + ch];
          dRdy = -SH_C1 * sh[1 * 3 + ch];
          dRdz = SH_C1 * sh[2 * 3 + ch];
          if (deg > 1) {
            float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            gsh[4 * 3 + ch] = SH_C2[0] * xy * gr;
            gsh[5 * 3 + ch] = SH_C2[1] * yz * gr;
            gsh[6 * 3 + ch] = SH_C2[2] * (2.f * zz - xx - yy) * gr;
            gsh[7 * 3 + ch] = SH_C2[3] * xz * gr;
            gsh[8 * 3 + ch] = SH_C2[4] * (xx - yy) * gr;
            dRdx += SH_C2[0] * y * sh[4 * 3 + ch] + SH_C2[2] * 2.f * -x * sh[6 * 3 + ch] +
                    SH_C2[3] * z * sh[7 * 3 + ch] + SH_C2[4] * 2.f * x * sh[8 * 3 + ch];
            dRdy += SH_C2[0] * x * sh[4 * 3 + ch] + SH_C2[1] * z * sh[5 * 3 + ch] +
                    SH_C2[2] * 2.f * -y * sh[6 * 3 + ch] + SH_C2[4] * 2.f * -y * sh[8 * 3 + ch];
            dRdz += SH_C2[1] * y * sh[5 * 3 + ch] + SH_C2[2] * 2.f * 2.f * z * sh[6 * 3 + ch] +
                    SH_C2[3] * x * sh[7 * 3 + ch];
            if (deg > 2) {
              gsh[9 * 3 + ch] = SH_C3[0] * y * (3.f * xx - yy) * gr;
              gsh[10 * 3 + ch] = SH_C3[1] * xy * z * gr;
              gsh[11 * 3 + ch] = SH_C3[2] * y * (4.f * zz - xx - yy) * gr;
              gsh[12 * 3 + ch] = SH_C3[3] * z * (2.f * zz - 3.f * xx - 3.f * yy) * gr;
              gsh[13 * 3 + ch] = SH_C3[4] * x * (4.f * zz - xx - yy) * gr;
              gsh[14 * 3 + ch] = SH_C3[5] * z * (xx - yy) * gr;
              gsh[15 * 3 + ch] = SH_C3[6] * x * (xx - 3.f * yy) * gr;
              dRdx += SH_C3[0] * sh[9 * 3 + ch] * 3.f * 2.f * xy + SH_C3[1] * sh[10 * 3 + ch] * yz +
                      SH_C3[2] * sh[11 * 3 + ch] * -2.f * xy +
                      SH_C3[3] * sh[12 * 3 + ch] * -3.f * 2.f * xz +
                      SH_C3[4] * sh[13 * 3 + ch] * (-3.f * xx + 4.f * zz - yy) +
                      SH_C3[5] * sh[14 * 3 + ch] * 2.f * xz +
                      SH_C3[6] * sh[15 * 3 + ch] * 3.f * (xx - yy);
              dRdy += SH_C3[0] * sh[9 * 3 + ch] * 3.f * (xx - yy) + SH_C3[1] * sh[10 * 3 + ch] * xz +
                      SH_C3[2] * sh[11 * 3 + ch] * (-3.f * yy + 4.f * zz - xx) +
                      SH_C3[3] * sh[12 * 3 + ch] * -3.f * 2.f * yz +
                      SH_C3[4] * sh[13 * 3 + ch] * -2.f * xy +
                      SH_C3[5] * sh[14 * 3 + ch] * -2.f * yz +
                      SH_C3[6] * sh[15 * 3 + ch] * -3.f * 2.f * xy;
              dRdz += SH_C3[1] * sh[10 * 3 + ch] * xy + SH_C3[2] * sh[11 * 3 + ch] * 4.f * 2.f * yz +
                      SH_C3[3] * sh[12 * 3 + ch] * 3.f * (2.f * zz - xx - yy) +
                      SH_C3[4] * sh[13 * 3 + ch] * 4.f * 2.f * xz +
                      SH_C3[5] * sh[14 * 3 + ch] * (xx - yy);
            }
          }
        }
        gdirx += dRdx * gr; gdiry += dRdy * gr; gdirz += dRdz * gr;
      }
      /* dir = d/|d|: J^T g = (g - dir (dir.g)) / |d| */
      float dot = x * gdirx + y * gdiry + z * gdirz;
      gp[0] += (gdirx - x * dot) / len;
      gp[1] += (gdiry - y * dot) / len;
      gp[2] += (gdirz - z * dot) / len;
    }
    gm3[0] = gp[0]; gm3[1] = gp[1]; gm3[2] = gp[2];
  }
  free(acc);
  return 0;
}

/* ---- fused_ssim oracle: utils/loss_utils.py:23-63 in C (double accumulation) -------------- */
static void ssim_window(float w[11]) {
  /* gaussian(11, 1.5): utils/loss_utils.py:23-25 (python floats = double, then float32 tensor) */
  double g[11], sum = 0.0;
  for (int x = 0; x < 11; ++x) { g[x] = exp(-(double)((x - 5) * (x - 5)) / (2.0 * 1.5 * 1.5)); sum += g[x]; }
  for (int x = 0; x < 11; ++x) w[x] = (float)((float)g[x] / (float)sum);
}

/* mean SSIM and optional gradient wrt img1. img [B,C,H,W]. */
double orc_ssim(const float* img1, const float* img2, int B, int C, int H, int W, float* ssim_map,
                float* grad_img1 /* may be NULL; dL/dmean = 1 */) {
  float w1[11];
  ssim_window(w1);
  const double C1 = 0.01 * 0.01, C2 = 0.03 * 0.03;
  const int64_t P = (int64_t)H * W, NP = (int64_t)B * C * P;
  double total = 0.0;
  double *dm_dmu1 = NULL, *dm_dsig1 = NULL, *dm_dsig12 = NULL;
  if (grad_img1) {
    dm_dmu1 = (double*)calloc((size_t)NP, sizeof(double));
    dm_dsig1 = (double*)calloc((size_t)NP, sizeof(double));
    dm_dsig12 = (double*)calloc((size_t)NP, sizeof(double));
  }
#pragma omp parallel for reduction(+ : total) schedule(static)
  for (int64_t pl = 0; pl < (int64_t)B * C; ++pl) {
    const float* a = img1 + pl * P;
    const float* b = img2 + pl * P;
    for (int y = 0; y < H; ++y)
      for (int x = 0; x < W; ++x) {
        double mu1 = 0, mu2 = 0, s11 = 0, s22 = 0, s12 = 0;
        for (int dy = -5; dy <= 5; ++dy) {
          int yy = y + dy;
          if (yy < 0 || yy >= H) continue;
          for (int dx = -5; dx <= 5; ++dx) {
            int xx = x + dx;
            if (xx < 0 || xx >= W) continue;
            double wv = (double)(w1[dy + 5] * w1[dx + 5]); /* 2D window is a float32 outer product */
            double p1 = a[(int64_t)yy * W + xx], p2 = b[(int64_t)yy * W + xx];
            mu1 += wv * p1; mu2 += wv * p2; s11 += wv * p1 * p1; s22 += wv * p2 * p2; s12 += wv * p1 * p2;
          }
        }
        double mu1sq = mu1 * mu1, mu2sq = mu2 * mu2, mu12 = mu1 * mu2;
        double sg1 = s11 - mu1sq, sg2 = s22 - mu2sq, sg12 = s12 - mu12;
        double A1 = 2 * mu12 + C1, A2 = 2 * sg12 + C2, B1 = mu1sq + mu2sq + C1, B2 = sg1 + sg2 + C2;
        double m = (A1 * A2) / (B1 * B2);
        total += m;
        int64_t idx = pl * P + (int64_t)y * W + x;
        if (ssim_map) ssim_map[idx] = (float)m;
        if (grad_img1) {
          /* partials wrt mu1, sigma1_sq, sigma12 treating (mu1, E[x^2], E[xy]) as the conv outputs */
          dm_dmu1[idx] = (2 * mu2 * A2) / (B1 * B2) - (2 * mu1 * A1 * A2) / (B1 * B1 * B2)
                         /* via sigma1_sq = s11 - mu1^2 and sigma12 = s12 - mu1 mu2 */
                         + (-2 * mu1) * (-(A1 * A2) / (B1 * B2 * B2)) + (-mu2) * (2 * A1 / (B1 * B2));
          dm_dsig1[idx] = -(A1 * A2) / (B1 * B2 * B2);
          dm_dsig12[idx] = 2 * A1 / (B1 * B2);
        }
      }
  }
  if (grad_img1) {
    const double scale = 1.0 / (double)NP;
#pragma omp parallel for schedule(static)
    for (int64_t pl = 0; pl < (int64_t)B * C; ++pl) {
      const float* a = img1 + pl * P;
      const float* b = img2 + pl * P;
      for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
          double acc1 = 0, acc2 = 0, acc3 = 0;
          for (int dy = -5; dy <= 5; ++dy) {
            int yy = y + dy;
            if (yy < 0 || yy >= H) continue;
            for (int dx = -5; dx <= 5; ++dx) {
              int xx = x + dx;
              if (xx < 0 || xx >= W) continue;
              double wv = (double)(w1[dy + 5] * w1[dx + 5]);
              int64_t j = pl * P + (int64_t)yy * W + xx;
              acc1 += wv * dm_dmu1[j]; acc2 += wv * dm_dsig1[j]; acc3 += wv * dm_dsig12[j];
            }
          }
          double p1 = a[(int64_t)y * W + x], p2 = b[(int64_t)y * W + x];
          grad_img1[pl * P + (int64_t)y * W + x] = (float)(scale * (acc1 + 2.0 * p1 * acc2 + p2 * acc3));
        }
    }
    free(dm_dmu1); free(dm_dsig1); free(dm_dsig12);
  }
  return total / (double)NP;
}

/* ---- simple_knn oracle: brute force mean squared distance to the 3 nearest other points ---- */
void orc_knn_dist2(const float* xyz, int32_t N, float* out) {
#pragma omp parallel for schedule(static)
  for (int32_t i = 0; i < N; ++i) {
    float best[3] = {INFINITY, INFINITY, INFINITY};
    float px = xyz[3 * (size_t)i], py = xyz[3 * (size_t)i + 1], pz = xyz[3 * (size_t)i + 2];
    for (int32_t j = 0; j < N; ++j) {
      if (j == i) continue;
      float dx = xyz[3 * (size_t)j] - px, dy = xyz[3 * (size_t)j + 1] - py, dz = xyz[3 * (size_t)j + 2] - pz;
      float d = dx * dx + dy * dy + dz * dz;
      if (d < best[2]) {
        if (d < best[0]) { best[2] = best[1]; best[1] = best[0]; best[0] = d; }
        else if (d < best[1]) { best[2] = best[1]; best[1] = d; }
        else best[2] = d;
      }
    }
    int cntv = 0; float sum = 0.f;
    for (int k = 0; k < 3; ++k) if (isfinite(best[k])) { sum += best[k]; cntv++; }
    out[i] = cntv ? sum / (float)cntv : 0.f;
  }
}

/* ---- test hooks: expose the helpers that the reference's own Python pins (tests/golden) ------------ */
void orc_test_cov3d(const float* s, float mod, const float* q, float* cov6) { cov3d_from_scale_rot(s, mod, q, cov6); }
void orc_test_quat_to_R(const float* q, float* R9) { quat_to_R(q, R9); }
void orc_test_sh(int deg, int M, const float* sh /* [M,3] */, const float* dir /* unit */, float* rgb) {
  /* sh_to_rgb normalises p - campos: pass campos = 0 and p = dir */
  const float zero[3] = {0.f, 0.f, 0.f};
  uint8_t cl[3];
  sh_to_rgb(deg, M, sh, dir, zero, rgb, cl);
}
/* pixel-space mean, depth and radius of one Gaussian: out = {mx, my, depth, radius} */
void orc_test_project(const OrcFrame* f, const float* p, const float* s, const float* q, float* out4) {
  OrcGeom g;
  const int TX = (f->W + TILE - 1) / TILE, TY = (f->H + TILE - 1) / TILE;
  const float col[3] = {0.f, 0.f, 0.f};
  preprocess_one(f, TX, TY, p, s, q, 1.0f, col, NULL, &g);
  out4[0] = g.mx; out4[1] = g.my; out4[2] = g.depth; out4[3] = (float)g.radius;
}
