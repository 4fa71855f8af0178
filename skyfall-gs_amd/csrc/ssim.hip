// ssim.hip -- fused SSIM forward/backward for gfx950. Replaces fused_ssim.fused_ssim(img1, img2)
// (reference call sites train.py:222,778; semantics == utils/loss_utils.py:23-63: 11x11 Gaussian window
// sigma 1.5, zero "same" padding, C1 = 0.01^2, C2 = 0.03^2, mean over all elements, gradient w.r.t. img1).
//
// One 32 x 22 output tile per 256-thread workgroup: the 42 x 32 input halos of both images are staged in LDS, the
// separable window runs as a horizontal pass into LDS followed by a vertical pass, both with REGISTER sliding windows
// (a thread produces 4 (3) neighbouring outputs from 14 (13) staged values: 3.5-4.3 LDS reads per output and moment
// instead of 11); 5 moments: mu1, mu2, E[x^2], E[y^2], E[xy]. Every output is the same ascending-k fma chain as a
// direct 11-tap sum. HBM traffic: forward reads 2 planes (the halo re-reads hit L2) and writes 3 partial-derivative
// maps (training) -- backward reads those 3 maps + 2 planes, writes 1. The mean is reduced without float atomics
// (per-block partials + a fixed-order final sum), so the loss is bit-reproducible.
#include "sfgs_internal.h"

namespace sfgs {

__constant__ float SSIM_W[11] = {1.028380124e-03f, 7.598758209e-03f, 3.600077331e-02f, 1.093606874e-01f,
                                 2.130055279e-01f, 2.660117149e-01f, 2.130055279e-01f, 1.093606874e-01f,
                                 3.600077331e-02f, 7.598758209e-03f, 1.028380124e-03f};
// tile of ST x STY = 32 x 22 outputs: the 42 x 32 staged halo gives the horizontal pass exactly 32 rows x 8 groups of
// 4 columns = 256 items, one per thread (a 32 x 32 tile has 336: a second round in which 176 threads idle at the barrier);
// the vertical pass is 32 columns x 8 groups of 3 rows (24 >= 22). 31.7 KB of LDS and <= 96 registers: five
// workgroups per CU (32 x 32: three).
// (SSIM_W[k] == SSIM_W[10 - k] bit for bit: window() reads entries 0..5 only)
constexpr int ST = 32, STY = 22, SHALO = 5, SIN = ST + 2 * SHALO, SINY = STY + 2 * SHALO;  // 42 x 32
constexpr int SPITCH = SIN + 2;                          // LDS row pitch of the staged inputs
constexpr int SQ = 4, SQV = 3;                           // outputs per thread: horizontal pass, vertical pass
static_assert(SINY * (ST / SQ) == 256 && ST * ((STY + SQV - 1) / SQV) == 256, "one item per thread in both passes");
constexpr float SSIM_C1 = 0.01f * 0.01f, SSIM_C2 = 0.03f * 0.03f;

__device__ __forceinline__ float block_sum_256(float v, float* smem) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
  const int tid = threadIdx.x;
  if ((tid & 63) == 0) smem[tid >> 6] = v;
  __syncthreads();
  return smem[0] + smem[1] + smem[2] + smem[3];
}

constexpr int SROUNDS = SINY / 4;   // staging rounds: wave w of the 4 stages halo row 4 r + w in round r, lane = column
static_assert(SINY % 4 == 0 && SIN <= 64 && SPITCH >= SIN, "a halo row is one wave's (partial) load");

// Global memory goes through buffer descriptors, one per image plane (wave-uniform): an access is descriptor + scalar
// byte offset (the row, or the tile origin) + ONE per-thread byte offset that every access of the thread shares, so
// the address arithmetic costs no VALU instruction -- flat addressing spent a 64-bit add per load and store, and these
// kernels are VALU-bound (profiles/r5_ssim_*.txt). A plane is at most 2^32 - 1 bytes (checked at the entry points).
using rsrc_t = __amdgpu_buffer_rsrc_t;
__device__ __forceinline__ rsrc_t plane_rsrc(const float* plane, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(plane), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float bload(rsrc_t r, uint32_t voff, uint32_t soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
__device__ __forceinline__ void bstore(float v, rsrc_t r, uint32_t voff, uint32_t soff) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), r, voff, soff, 0);
}

// Staging: wave w of the 4 stages halo row 4 r + w in round r, lane = column. Rows are wave-uniform, so a row's byte
// offset and validity are scalars; the column offset (clamped into the row: always a valid address) is the thread's
// one vector offset. Two v_cndmask per element apply the zero padding on the way to LDS -- instead of an integer
// division, four compares and a 64-bit address per element.
struct HaloLane { uint32_t xoff; bool xin; };
__device__ __forceinline__ HaloLane halo_lane(int lane, int x0, int W) {
  const int gx = x0 - SHALO + lane;
  return {(uint32_t)min(max(gx, 0), W - 1) * 4u, lane < SIN && gx >= 0 && gx < W};
}
// byte offset of image row (clamped) `y0 - SHALO + row` within its plane, and whether that row exists
__device__ __forceinline__ uint32_t halo_row(int row, int y0, int H, int W, bool& yin) {
  const int gy = y0 - SHALO + row;
  yin = gy >= 0 && gy < H;
  return (uint32_t)min(max(gy, 0), H - 1) * (uint32_t)W * 4u;
}

// Workgroup -> tile. The hardware deals workgroups to the 8 XCDs round-robin by linear id, and each XCD has its own L2:
// with tile = workgroup id, the four neighbours whose halos overlap a tile's (10 of its 32 staged rows, 10 of its 42
// columns) run on OTHER XCDs and every XCD pulls its own copy of the shared lines through the fabric -- 2.7x the
// algorithmic bytes for the backward kernel. So XCD k takes the k-th CONTIGUOUS eighth of the tiles (row-major within a
// plane), in order: what it has in flight at any time is a band of a few tile rows, whose halos meet in its L2.
// Returns the tile's index in plane-major, row-major order (also the index of its partial sum).
struct SsimTile { int index, plane, x0, y0; };
__device__ __forceinline__ SsimTile ssim_tile(int tiles_x, int tiles_y) {
  const int L = (int)xcd_remap(blockIdx.x, gridDim.x);
  const int per_plane = tiles_x * tiles_y;
  const int plane = L / per_plane, rem = L - plane * per_plane;
  const int by = rem / tiles_x, bx = rem - by * tiles_x;
  return {L, plane, bx * ST, by * STY};
}

// hz row of tap k of the vertical window that starts at row ly0 <= STY - 1: only the taps of the output rows that do not
// exist (22, 23) can pass the last staged row, and only those pay for the clamp
__device__ __forceinline__ int vrow(int ly0, int k) {
  return k <= SINY - STY ? ly0 + k : min(ly0 + k, SINY - 1);
}

// out[q] = sum_k W[k] * v[q + k], k ascending (the order of a direct 11-tap sum)
// The window's six distinct weights (it is symmetric) in VECTOR registers: v_fmac_f32 with a scalar-register
// weight measures 9-15 % slower over the whole forward kernel than with a vector-register one
// (profiles/r5_ssim_steps.txt), and the compiler keeps a __constant__ table in scalar registers unless told otherwise.
struct WindowWeights { float w[6]; };
__device__ __forceinline__ WindowWeights window_weights() {
  WindowWeights r;
#pragma unroll
  for (int k = 0; k < 6; ++k) asm volatile("v_mov_b32 %0, %1" : "=v"(r.w[k]) : "s"(SSIM_W[k]));
  return r;
}
template <int Q>
__device__ __forceinline__ void window(const WindowWeights& ww, const float (&v)[Q + 10], float (&out)[Q]) {
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    float a = 0.f;
#pragma unroll
    for (int k = 0; k < 11; ++k) a = fmaf(ww.w[k < 6 ? k : 10 - k], v[q + k], a);
    out[q] = a;
  }
}

__global__ void __launch_bounds__(256)
ssim_fwd_kernel(const float* __restrict__ img1, const float* __restrict__ img2, int H, int W, int tiles_x, int tiles_y,
                float* __restrict__ ssim_map, float* __restrict__ block_partials, float* __restrict__ dm_dmu1,
                float* __restrict__ dm_dsig1, float* __restrict__ dm_dsig12) {
  __shared__ float s1[SINY][SPITCH], s2[SINY][SPITCH];   // 11.3 KB of staged inputs
  __shared__ float hz[5][SINY][ST];                       // 20.5 KB of horizontal-pass results
  __shared__ float red[4];
  const SsimTile T = ssim_tile(tiles_x, tiles_y);
  const size_t poff = (size_t)T.plane * H * W;
  const uint32_t pbytes = (uint32_t)H * (uint32_t)W * 4u;
  const int x0 = T.x0, y0 = T.y0;
  const int tid = threadIdx.x;
  const WindowWeights ww = window_weights();
  // staging: every load is issued before the first LDS write, from an address that is always valid (clamped into the
  // image); the zero padding is applied when the value goes to LDS. (A load behind a per-lane condition compiles to
  // load / s_waitcnt vmcnt(0) / ds_write per element: 14 serialised round trips to HBM per tile, 57 % of the wave
  // time parked in waits -- profiles/r5_ssim_sq_counters_before.txt.)
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const HaloLane hl = halo_lane(lane, x0, W);
  float r1[SROUNDS], r2[SROUNDS];
  bool yin[SROUNDS];
  const rsrc_t b1 = plane_rsrc(img1 + poff, pbytes), b2 = plane_rsrc(img2 + poff, pbytes);
#pragma unroll
  for (int r = 0; r < SROUNDS; ++r) {
    const uint32_t row = halo_row(4 * r + wv, y0, H, W, yin[r]);
    r1[r] = bload(b1, hl.xoff, row);
    r2[r] = bload(b2, hl.xoff, row);
  }
  if (lane < SIN) {
#pragma unroll
    for (int r = 0; r < SROUNDS; ++r) {
      const bool in = hl.xin && yin[r];
      s1[4 * r + wv][lane] = in ? r1[r] : 0.f;
      s2[4 * r + wv][lane] = in ? r2[r] : 0.f;
    }
  }
  __syncthreads();
  {  // horizontal pass: thread = (staged row, group of SQ output columns)
    const int ly = tid / (ST / SQ), hx = (tid - ly * (ST / SQ)) * SQ;
    float a[SQ + 10], b[SQ + 10], t[SQ + 10], o[SQ];
#pragma unroll
    for (int k = 0; k < SQ + 10; ++k) { a[k] = s1[ly][hx + k]; b[k] = s2[ly][hx + k]; }
    window<SQ>(ww, a, o);
#pragma unroll
    for (int q = 0; q < SQ; ++q) hz[0][ly][hx + q] = o[q];
    window<SQ>(ww, b, o);
#pragma unroll
    for (int q = 0; q < SQ; ++q) hz[1][ly][hx + q] = o[q];
#pragma unroll
    for (int k = 0; k < SQ + 10; ++k) t[k] = a[k] * a[k];
    window<SQ>(ww, t, o);
#pragma unroll
    for (int q = 0; q < SQ; ++q) hz[2][ly][hx + q] = o[q];
#pragma unroll
    for (int k = 0; k < SQ + 10; ++k) t[k] = b[k] * b[k];
    window<SQ>(ww, t, o);
#pragma unroll
    for (int q = 0; q < SQ; ++q) hz[3][ly][hx + q] = o[q];
#pragma unroll
    for (int k = 0; k < SQ + 10; ++k) t[k] = a[k] * b[k];
    window<SQ>(ww, t, o);
#pragma unroll
    for (int q = 0; q < SQ; ++q) hz[4][ly][hx + q] = o[q];
  }
  __syncthreads();
  // vertical pass: thread = (column, group of SQV output rows); the last group's rows 22, 23 do not exist: their
  // windows read hz rows that were never staged (clamped below) and nothing is written for them
  const int lx = tid & (ST - 1), ly0 = (tid / ST) * SQV;
  float mo[5][SQV];
#pragma unroll
  for (int m = 0; m < 5; ++m) {
    float v[SQV + 10];
#pragma unroll
    for (int k = 0; k < SQV + 10; ++k) v[k] = hz[m][vrow(ly0, k)][lx];
    window<SQV>(ww, v, mo[m]);
  }
  const int gx = x0 + lx;
  const rsrc_t o0 = plane_rsrc(ssim_map + poff, pbytes), o1 = plane_rsrc(dm_dmu1 + poff, pbytes),
               o2 = plane_rsrc(dm_dsig1 + poff, pbytes), o3 = plane_rsrc(dm_dsig12 + poff, pbytes);
  const uint32_t vout = (uint32_t)(ly0 * W + lx) * 4u;   // the thread's first output, relative to the tile origin
  float vsum = 0.f;
#pragma unroll
  for (int q = 0; q < SQV; ++q) {
    const int gy = y0 + ly0 + q;
    const uint32_t sout = (uint32_t)((y0 + q) * W + x0) * 4u;   // tile origin + q rows: scalar
    if (gx < W && gy < H && ly0 + q < STY) {
      const float mu1 = mo[0][q], mu2 = mo[1][q], e11 = mo[2][q], e22 = mo[3][q], e12 = mo[4][q];
      const float mu1sq = mu1 * mu1, mu2sq = mu2 * mu2, mu12 = mu1 * mu2;
      const float sg1 = e11 - mu1sq, sg2 = e22 - mu2sq, sg12 = e12 - mu12;
      const float A1 = 2.f * mu12 + SSIM_C1, A2 = 2.f * sg12 + SSIM_C2;
      const float B1 = mu1sq + mu2sq + SSIM_C1, B2 = sg1 + sg2 + SSIM_C2;
      const float inv = 1.0f / (B1 * B2);
      const float val = A1 * A2 * inv;
      vsum += val;
      if (ssim_map) bstore(val, o0, vout, sout);
      if (dm_dmu1) {
        // partials w.r.t. the three convolution outputs that depend on img1: mu1, E[x^2], E[xy]
        // one division per pixel: 1 / B1 = B2 * inv and 1 / B2 = B1 * inv (an IEEE division is ten VALU instructions
        // and this kernel is VALU-bound)
        const float d_sig1 = -val * (B1 * inv);    // d/d sigma1_sq = -val / B2
        const float d_sig12 = 2.f * A1 * inv;      // d/d sigma12
        bstore(2.f * mu2 * A2 * inv - 2.f * mu1 * val * (B2 * inv) - 2.f * mu1 * d_sig1 - mu2 * d_sig12, o1, vout, sout);
        bstore(d_sig1, o2, vout, sout);
        bstore(d_sig12, o3, vout, sout);
      }
    }
  }
  const float bs = block_sum_256(vsum, red);
  if (tid == 0) block_partials[T.index] = bs;
}

// fixed-order final reduction: mean = sum(partials) / count
__global__ void __launch_bounds__(1024) ssim_mean_kernel(const float* __restrict__ partials, int n, float inv_count,
                                                         float* __restrict__ out) {
  __shared__ double sm[16];
  double acc = 0.0;
  for (int i = threadIdx.x; i < n; i += 1024) acc += (double)partials[i];
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int w = 0; w < 16; ++w) t += sm[w];
    out[0] = (float)(t * (double)inv_count);
  }
}

__global__ void __launch_bounds__(256)
ssim_bwd_kernel(const float* __restrict__ img1, const float* __restrict__ img2, int H, int W, int tiles_x, int tiles_y,
                const float* __restrict__ dm_dmu1, const float* __restrict__ dm_dsig1,
                const float* __restrict__ dm_dsig12, const float* __restrict__ dL_dmean, float inv_count,
                float* __restrict__ dL_dimg1) {
  __shared__ float s[3][SINY][SPITCH];   // 16.9 KB
  __shared__ float hz[3][SINY][ST];      // 12.3 KB
  const SsimTile T = ssim_tile(tiles_x, tiles_y);
  const size_t poff = (size_t)T.plane * H * W;
  const uint32_t pbytes = (uint32_t)H * (uint32_t)W * 4u;
  const int x0 = T.x0, y0 = T.y0;
  const int tid = threadIdx.x;
  const WindowWeights ww = window_weights();
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const HaloLane hl = halo_lane(lane, x0, W);
  float r0[SROUNDS], r1[SROUNDS], r2[SROUNDS];   // all 24 loads in flight before the first LDS write (see forward)
  bool yin[SROUNDS];
  const rsrc_t m0 = plane_rsrc(dm_dmu1 + poff, pbytes), m1 = plane_rsrc(dm_dsig1 + poff, pbytes),
               m2 = plane_rsrc(dm_dsig12 + poff, pbytes);
#pragma unroll
  for (int r = 0; r < SROUNDS; ++r) {
    const uint32_t row = halo_row(4 * r + wv, y0, H, W, yin[r]);
    r0[r] = bload(m0, hl.xoff, row);
    r1[r] = bload(m1, hl.xoff, row);
    r2[r] = bload(m2, hl.xoff, row);
  }
  // this thread's own pixels are needed last: their loads go out with the halos'
  const int lx = tid & (ST - 1), ly0 = (tid / ST) * SQV;
  const int gx = x0 + lx;
  // (pixels that do not exist load the plane's first pixel: unused)
  const rsrc_t b1 = plane_rsrc(img1 + poff, pbytes), b2 = plane_rsrc(img2 + poff, pbytes),
               og = plane_rsrc(dL_dimg1 + poff, pbytes);
  const uint32_t vout = (uint32_t)(ly0 * W + lx) * 4u;   // the thread's first pixel, relative to the tile origin
  float p1[SQV], p2[SQV];
#pragma unroll
  for (int q = 0; q < SQV; ++q) {
    const bool ok = gx < W && y0 + ly0 + q < H && ly0 + q < STY;
    const uint32_t sout = (uint32_t)((y0 + q) * W + x0) * 4u;
    p1[q] = bload(b1, ok ? vout : 0u, ok ? sout : 0u);
    p2[q] = bload(b2, ok ? vout : 0u, ok ? sout : 0u);
  }
  if (lane < SIN) {
#pragma unroll
    for (int r = 0; r < SROUNDS; ++r) {
      const bool in = hl.xin && yin[r];
      s[0][4 * r + wv][lane] = in ? r0[r] : 0.f;
      s[1][4 * r + wv][lane] = in ? r1[r] : 0.f;
      s[2][4 * r + wv][lane] = in ? r2[r] : 0.f;
    }
  }
  __syncthreads();
  {
    const int ly = tid / (ST / SQ), hx = (tid - ly * (ST / SQ)) * SQ;
#pragma unroll
    for (int m = 0; m < 3; ++m) {
      float v[SQ + 10], o[SQ];
#pragma unroll
      for (int k = 0; k < SQ + 10; ++k) v[k] = s[m][ly][hx + k];
      window<SQ>(ww, v, o);
#pragma unroll
      for (int q = 0; q < SQ; ++q) hz[m][ly][hx + q] = o[q];
    }
  }
  __syncthreads();
  float mo[3][SQV];
#pragma unroll
  for (int m = 0; m < 3; ++m) {
    float v[SQV + 10];
#pragma unroll
    for (int k = 0; k < SQV + 10; ++k) v[k] = hz[m][vrow(ly0, k)][lx];
    window<SQV>(ww, v, mo[m]);
  }
  const float scale = dL_dmean[0] * inv_count;
#pragma unroll
  for (int q = 0; q < SQV; ++q) {
    const int gy = y0 + ly0 + q;
    if (gx < W && gy < H && ly0 + q < STY)
      bstore(scale * (mo[0][q] + 2.f * p1[q] * mo[1][q] + p2[q] * mo[2][q]), og, vout, (uint32_t)((y0 + q) * W + x0) * 4u);
  }
}

}  // namespace sfgs

using namespace sfgs;

static inline size_t ssim_nblocks(int B, int C, int H, int W) {
  return (size_t)B * C * ((H + STY - 1) / STY) * ((W + ST - 1) / ST);
}

extern "C" size_t sfgs_ssim_scratch_bytes(int32_t B, int32_t C, int32_t H, int32_t W, int32_t with_grad) {
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0) return 0;
  size_t b = align_up(ssim_nblocks(B, C, H, W) * 4, 256);
  if (with_grad) b += 3 * align_up((size_t)B * C * H * W * 4, 256);
  return b;
}

extern "C" int sfgs_ssim_forward(const float* img1, const float* img2, int32_t B, int32_t C, int32_t H, int32_t W,
                                 float* ssim_map_or_null, float* ssim_mean, void* scratch, size_t scratch_sz,
                                 int32_t with_grad, void* stream_) {
  SFGS_REQUIRE(img1 && img2 && ssim_mean && scratch, SFGS_E_ARG, "NULL argument");
  SFGS_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, SFGS_E_ARG, "bad image shape [%d,%d,%d,%d]", B, C, H, W);
  SFGS_REQUIRE(ssim_nblocks(B, C, H, W) <= (size_t)INT32_MAX, SFGS_E_UNSUPPORTED, "more than 2^31 - 1 tiles of 32 x 22");
  SFGS_REQUIRE((int64_t)H * W < ((int64_t)1 << 30), SFGS_E_UNSUPPORTED, "an image plane of 2^30 pixels or more");
  SFGS_REQUIRE(scratch_sz >= sfgs_ssim_scratch_bytes(B, C, H, W, with_grad), SFGS_E_CAPACITY, "ssim scratch too small");
  hipStream_t stream = (hipStream_t)stream_;
  const size_t nblk = ssim_nblocks(B, C, H, W), plane = align_up((size_t)B * C * H * W * 4, 256);
  float* partials = (float*)scratch;
  char* maps = (char*)scratch + align_up(nblk * 4, 256);
  float* m0 = with_grad ? (float*)maps : nullptr;
  float* m1 = with_grad ? (float*)(maps + plane) : nullptr;
  float* m2 = with_grad ? (float*)(maps + 2 * plane) : nullptr;
  const int tiles_x = (W + ST - 1) / ST, tiles_y = (H + STY - 1) / STY;
  { ProfScope ps_(KID_SSIM_FWD, stream);
    hipLaunchKernelGGL(ssim_fwd_kernel, dim3((unsigned)nblk), dim3(256), 0, stream, img1, img2, H, W, tiles_x, tiles_y,
                       ssim_map_or_null, partials, m0, m1, m2); }
  SFGS_POST_LAUNCH("ssim_fwd", stream, 0);
  { ProfScope ps_(KID_SSIM_MEAN, stream);
    hipLaunchKernelGGL(ssim_mean_kernel, dim3(1), dim3(1024), 0, stream, partials, (int)nblk,
                       1.0f / (float)((double)B * C * H * W), ssim_mean); }
  SFGS_POST_LAUNCH("ssim_mean", stream, 0);
  return SFGS_OK;
}

extern "C" int sfgs_ssim_backward(const float* img1, const float* img2, int32_t B, int32_t C, int32_t H, int32_t W,
                                  const void* scratch, const float* dL_dmean, float* dL_dimg1, void* stream_) {
  SFGS_REQUIRE(img1 && img2 && scratch && dL_dmean && dL_dimg1, SFGS_E_ARG, "NULL argument");
  SFGS_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, SFGS_E_ARG, "bad image shape");
  SFGS_REQUIRE(ssim_nblocks(B, C, H, W) <= (size_t)INT32_MAX && (int64_t)H * W < ((int64_t)1 << 30),
               SFGS_E_UNSUPPORTED, "image too large");
  hipStream_t stream = (hipStream_t)stream_;
  const size_t nblk = ssim_nblocks(B, C, H, W), plane = align_up((size_t)B * C * H * W * 4, 256);
  const char* maps = (const char*)scratch + align_up(nblk * 4, 256);
  const int tiles_x = (W + ST - 1) / ST, tiles_y = (H + STY - 1) / STY;
  { ProfScope ps_(KID_SSIM_BWD, stream);
    hipLaunchKernelGGL(ssim_bwd_kernel, dim3((unsigned)nblk), dim3(256), 0, stream, img1, img2, H, W, tiles_x, tiles_y,
                       (const float*)maps,
                       (const float*)(maps + plane), (const float*)(maps + 2 * plane), dL_dmean,
                       1.0f / (float)((double)B * C * H * W), dL_dimg1); }
  SFGS_POST_LAUNCH("ssim_bwd", stream, 0);
  return SFGS_OK;
}
