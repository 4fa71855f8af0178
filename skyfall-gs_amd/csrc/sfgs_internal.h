// sfgs_internal.h -- blob layouts, launch helpers and error plumbing shared by the .hip files.
// Not part of the ABI (include/sfgs.h is).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/sfgs.h"
#include "raster_math.h"

namespace sfgs {

// ---- error plumbing --------------------------------------------------------------------------
void set_error(const char* fmt, ...);

#define SFGS_CHECK_HIP(expr)                                                              \
  do {                                                                                    \
    hipError_t e_ = (expr);                                                               \
    if (e_ != hipSuccess) {                                                               \
      sfgs::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
      return SFGS_E_HIP;                                                                  \
    }                                                                                     \
  } while (0)

#define SFGS_REQUIRE(cond, code, ...)  \
  do {                                 \
    if (!(cond)) {                     \
      sfgs::set_error(__VA_ARGS__);    \
      return (code);                   \
    }                                  \
  } while (0)

// ---- optional per-kernel profiler (api.cpp): HIP events on the launch stream ---------------------
enum KernelId {
  KID_SUBPIX = 0, KID_PREPROCESS, KID_BIN_COUNT, KID_BIN_RANK, KID_BIN_SCATTER, KID_PLAN_SCAN, KID_FINE_BIN, KID_SORT_SMALL, KID_SORT_REG_LONG, KID_SORT_LDS,
  KID_COMPOSITE_FWD, KID_COMPOSITE_BWD, KID_PREPROCESS_BWD, KID_SSIM_FWD, KID_SSIM_MEAN, KID_SSIM_BWD, KID_KNN, KID_PREPASS_FWD, KID_PREPASS_BWD, KID_FILTER3D, KID_DENSIFY_STATS, KID_ADAM, KID_SH_EVAL_FWD, KID_SH_EVAL_BWD, KID_COMPACT_SCAN, KID_COMPACT_GATHER, KID_DENSIFY,
  KID_COUNT
};
// process-wide route options (include/sfgs.h: sfgs_set_option; api.cpp): one relaxed atomic load per query
enum { OPT_SORT, OPT_PLAN_SCAN, OPT_BINNING, OPT_PREFILL, OPT_KNN, OPT_TILE_ORDER, OPT_COUNT };
enum { SORT_AUTO = 0, SORT_FUSED = 1, SORT_FUSED1024 = 2, SORT_SPLIT = 3, SORT_FUSED768 = 4 };
enum { PREFILL_AUTO = 0, PREFILL_ALWAYS = 1, PREFILL_NEVER = 2 };
int option(int which);
bool prof_enabled();
bool prof_selected(int id);
void* prof_begin(int id, hipStream_t stream);
void prof_end(void* token, hipStream_t stream);
struct ProfScope {
  void* tok; hipStream_t st;
  ProfScope(int id, hipStream_t s) : tok(prof_selected(id) ? prof_begin(id, s) : nullptr), st(s) {}
  ~ProfScope() { if (tok) prof_end(tok, st); }
};

// after a launch: always catch launch errors; in debug mode also synchronise (SfgsFrame.debug)
#define SFGS_POST_LAUNCH(name, stream, debug)                                      \
  do {                                                                             \
    hipError_t e_ = hipGetLastError();                                             \
    if (e_ == hipSuccess && (debug)) e_ = hipStreamSynchronize(stream);            \
    if (e_ != hipSuccess) {                                                        \
      sfgs::set_error("kernel %s failed: %s", name, hipGetErrorString(e_));        \
      return SFGS_E_HIP;                                                           \
    }                                                                              \
  } while (0)

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// ---- kernel-side view of SfgsFrame -------------------------------------------------------------
struct KFrame {
  int W, H;
  float tanfovx, tanfovy, kernel_size, scale_modifier;
  int sh_degree, sh_coeffs, depth_mode;
  int band0, band1;  // 8-pixel tile rows [band0, band1) that are binned / composited
  const float* view;
  const float* proj;
  const float* campos;
  const float* bg;
  const float* subpix;
};

static inline KFrame make_kframe(const SfgsFrame* f) {
  KFrame k;
  k.W = f->image_width; k.H = f->image_height;
  k.tanfovx = f->tanfovx; k.tanfovy = f->tanfovy;
  k.kernel_size = f->kernel_size; k.scale_modifier = f->scale_modifier;
  k.sh_degree = f->sh_degree; k.sh_coeffs = f->sh_coeffs; k.depth_mode = f->depth_mode;
  const int ty8 = (f->image_height + TILE_BIN - 1) / TILE_BIN;
  k.band0 = f->tile_row_end > 0 ? (f->tile_row_begin < 0 ? 0 : f->tile_row_begin) : 0;
  k.band1 = f->tile_row_end > 0 ? (f->tile_row_end > ty8 ? ty8 : f->tile_row_end) : ty8;
  k.view = f->viewmatrix; k.proj = f->projmatrix; k.campos = f->campos; k.bg = f->bg;
  k.subpix = f->subpixel_offset;
  return k;
}

__device__ __forceinline__ FrameParams load_frame(const KFrame& k) {
  FrameParams f;
  f.W = k.W; f.H = k.H;
  f.tanfovx = k.tanfovx; f.tanfovy = k.tanfovy;
  f.kernel_size = k.kernel_size; f.scale_modifier = k.scale_modifier;
  f.sh_degree = k.sh_degree; f.sh_coeffs = k.sh_coeffs; f.depth_mode = k.depth_mode;
#pragma unroll
  for (int i = 0; i < 16; ++i) { f.view[i] = k.view[i]; f.proj[i] = k.proj[i]; }
#pragma unroll
  for (int i = 0; i < 3; ++i) { f.campos[i] = k.campos[i]; f.bg[i] = k.bg[i]; }
  return f;
}

// ---- blob layouts -------------------------------------------------------------------------------
constexpr int PRE_BLOCK = 256;      // threads per block of the per-Gaussian kernels
// SfgsFrame.feedback: 8 uint64 words of caller-owned persistent device memory (late statistics of the previous frame)
enum { FB_VALID = 0, FB_LONG_TILES = 1, FB_MAX_LIST = 2, FB_PREFILLED = 3, FB_OVER_512 = 4 };   // FB_OVER_512: tiles with > 512 entries
// reserve `total` duplicate indices from pool `pool`; *fits = the pool still had room
constexpr int HDR_WORDS = 64;       // uint64 words at the head of the tiles blob

enum HeaderSlot {
  HDR_D_EFF = 0,      // binned (Gaussian, 8x8 tile) pairs; doubles as the duplicate-index allocator of K1
  HDR_D_REF = 1,      // sum of the reference's tiles_touched (16x16 rule)
  HDR_N_VIS = 2,      // count(radii > 0)
  HDR_MAX_LIST = 3,   // longest per-tile list
  HDR_OVERFLOW = 4,   // set by preprocess when dup_capacity / coarse_capacity is too small (redo the plan)
  HDR_SUBPIX_BOUND = 5,  // float bits of max |subpixel_offset| (0 when none)
  HDR_ITEM_ALLOC = 6,    // list slots the coarse bins' lists can take at most (total of the plan's per-bin bases)
  HDR_MAX_COARSE = 7,    // fullest coarse-bin SLAB: the most items appended directly to one bin (splats of more than BIG_WALK
                         // bins; one-pass binning) -- what coarse_capacity has to hold; the two-pass items live in BinsView::csr
  HDR_LONG_COUNT = 8,    // tiles whose list is too long for the register sort (> 512 entries)
  HDR_BIG_CHUNKS = 9,    // entries of BinsView::big_chunks (1024-record chunks of Gaussians with > BWD_BIG duplicates)
  HDR_BIG_COUNT = 10,    // entries of GeomView::big_list (splats whose binning walk is done by big_walk_kernel)
  HDR_PREFILLED = 11,    // backward: 1 when dupgrad_prefill_kernel zeroed the whole record array (composite_bwd then
                         // skips the entries behind a tile's last contributor), else 0
  HDR_CSR_CURSOR = 12,   // two-pass binning: items handed out of BinsView::csr so far (bin_rank_kernel: one atomic per 16 bins)
  HDR_MAX_BIN_ITEMS = 13, // fullest coarse bin counting ALL its items (slab + csr run): what the sort route is chosen from
  HDR_TILE_ORDER = 14    // bit 0 / 1: TilesView::tile_order holds this frame's longest-first tile order for composite_fwd / _bwd
};

// float4s per compositing record in global memory: 3 = packed 48-byte records; 4 = 64-byte stride (the fourth is never
// touched): every id -> record gather then falls into ONE 64-byte sector instead of 1.5 on average
constexpr int REC_F4 = 3;

struct GeomView {
  float4* rec;      // [N][REC_F4] float4: (mx, my, qa, qb) (qc, op, r, g) (b, depth, ex, ey) -- SplatRec with (r, g), (b, depth) paired
  uint2* dup;       // [N] (first duplicate index, duplicate count) of every Gaussian
  uint4* big_list;  // [N] work list of big_walk_kernel: (Gaussian id, tile range x0 | x1 << 16, y0 | y1 << 16, 0);
                    //     HDR_BIG_COUNT entries, written only for splats that reach more than BIG_WALK coarse bins
  uint32_t* block_items;  // [NB] two-pass binning: how many coarse items preprocess workgroup b wrote to BinsView::pairs
};
constexpr int PAIRS_PER_BLOCK = PRE_BLOCK * 6;   // a thread emits at most BIG_WALK (= 6) items itself (raster_fwd.hip)
// The binning's pair list (96 bytes of address space per Gaussian) lives in the BINS blob, on top of the per-tile item /
// list arrays that only the render stage writes (BinsView::pairs; ADVICE r3: it used to be part of this blob, which the
// caller keeps for the backward): 72 bytes per Gaussian here instead of 168.
static inline size_t pairs_bytes(int64_t N) {
  const int64_t NB = (N + PRE_BLOCK - 1) / PRE_BLOCK;
  return align_up((size_t)NB * PAIRS_PER_BLOCK * 16, 256);
}
static inline size_t geom_bytes(int64_t N) {
  const int64_t NB = (N + PRE_BLOCK - 1) / PRE_BLOCK;
  return align_up((size_t)N * 16 * REC_F4, 256) + align_up((size_t)N * 8, 256) + align_up((size_t)N * 16, 256) +
         align_up((size_t)NB * 4, 256);
}
static inline GeomView geom_view(void* base, int64_t N) {
  GeomView g;
  const int64_t NB = (N + PRE_BLOCK - 1) / PRE_BLOCK;
  char* p = (char*)base;
  g.rec = (float4*)p; p += align_up((size_t)N * 16 * REC_F4, 256);
  g.dup = (uint2*)p; p += align_up((size_t)N * 8, 256);
  g.big_list = (uint4*)p; p += align_up((size_t)N * 16, 256);
  g.block_items = (uint32_t*)p;
  (void)NB;
  return g;
}

// Coarse bins: 4x4 tiles of 8x8 pixels = 32x32 pixels. Binning is two-level so that the slow device-wide
// atomics (measured 26.7 G/s on MI355X, independent of scope / return / locality) are paid once per (Gaussian,
// coarse bin) pair instead of once per (Gaussian, tile) pair; the per-tile ranks come from LDS atomics.
constexpr int COARSE = 4;           // tiles per coarse-bin edge
constexpr int COARSE_TILES = COARSE * COARSE;
constexpr int CC_STRIDE = 32;       // coarse counters live 128 bytes apart: atomics on one cache line serialise
                                    // (~12 ns each), and a dense array would put 16-32 hot counters on one line

// Duplicate indices are handed out from DUP_POOLS independent ranges of the index space [0, dup_capacity): pool p owns
// [p R, (p + 1) R), R = dup_capacity / DUP_POOLS, with its own counter on its own cache line. ONE allocator word took a
// returning atomic per preprocess workgroup (7 813 at 2 M Gaussians) that queue at ~10 ns each in the memory-side atomic
// unit: 15 of the kernel's ~210 us. Workgroup b draws from pool b % DUP_POOLS (= its XCD), so the pools fill evenly; a pool
// that runs over sets the overflow flag like any other capacity. The index space is sparse (gaps between the pools):
// everything indexed by duplicate index is sized by dup_capacity.
constexpr int DUP_POOLS = 8, DP_STRIDE = 16;   // counters 128 bytes apart

struct TilesView {
  unsigned long long* hdr;  // [HDR_WORDS]
  unsigned long long* dup_pool;   // [DUP_POOLS * DP_STRIDE] duplicate indices handed out per pool
  uint32_t* coarse_count;   // [NCB * CC_STRIDE] one 128-byte line per coarse bin: word 0 = items appended (keeps counting
                            //   past capacity), word 1 = tile hits of those items (words 0-1 are ONE 64-bit atomic
                            //   counter), word 2 = the bin's first list slot (scanned by the plan from the hits)
                            //   word 3 = slots handed out to the bin's tiles so far (select_sort_kernel: one atomic
                            //   per tile), word 4 = the bin's longest tile list (reduced by list_stats), word 5 = the bin's
                            //   tiles with more than 512 entries (select_sort_kernel<1024>; summed by list_stats),
                            //   word 6 = first item of the bin's run in BinsView::csr, word 7 = items of that run (two-pass
                            //   binning, bin_rank_kernel; word 0 counts them too: word 0 - word 7 items sit in the bin's slab)
  uint2* tile_range;        // [T8] (first list slot, list length) of every 8x8 tile
  uint32_t* long_tiles;     // [T8] ids of the tiles with more than 512 entries (HDR_LONG_COUNT of them)
  uint32_t* block_nvis;     // [NB]
  unsigned long long* block_dref;  // [NB]
  size_t zero_bytes;        // bytes from the start of the blob that plan() must clear
  // two-pass binning without device atomics: per (scatter workgroup, coarse bin) the workgroup's items, their tile hits
  // and -- after the column scan -- the first slab rank of its run ([scatter_groups(N)][N_cb] each, fully rewritten per frame)
  uint32_t *sc_cnt, *sc_hits, *sc_base;
  // longest-first tile order of the compositing kernels (tile_order_kernel, SFGS_HINT_TILE_ORDER): [2][8][order_slots], the
  // forward's (by list length) and the backward's (by last contributor); 0xffffffff = no tile
  uint32_t* tile_order;
};
constexpr int SCATTER_BLOCKS = 32;   // preprocess workgroups per scatter workgroup
static inline int tiles8_x(int W) { return (W + TILE_BIN - 1) / TILE_BIN; }
static inline int tiles8_y(int H) { return (H + TILE_BIN - 1) / TILE_BIN; }
static inline int64_t tiles8(int W, int H) { return (int64_t)tiles8_x(W) * tiles8_y(H); }
static inline int coarse_x(int W) { return (tiles8_x(W) + COARSE - 1) / COARSE; }
static inline int coarse_y(int H) { return (tiles8_y(H) + COARSE - 1) / COARSE; }
static inline int64_t coarse_bins(int W, int H) { return (int64_t)coarse_x(W) * coarse_y(H); }
static inline int64_t pre_blocks(int64_t N) { return (N + PRE_BLOCK - 1) / PRE_BLOCK; }
static inline int64_t scatter_groups(int64_t N) { return (pre_blocks(N) + SCATTER_BLOCKS - 1) / SCATTER_BLOCKS; }
// tile slots per XCD of the compositing kernels' chunked mapping (composite_wave_role below): chunks of 8 x 4 blocks of 2 x 2 tiles,
// every eighth chunk to an XCD
static inline int64_t order_slots(int W, int H) {
  const int64_t SX = (tiles8_x(W) + 1) / 2, SY = (tiles8_y(H) + 1) / 2;
  const int64_t nch = ((SX + 7) / 8) * ((SY + 3) / 4);
  return (nch + 7) / 8 * 128;
}
static inline TilesView tiles_view(void* base, int W, int H, int64_t N, size_t* total) {
  TilesView t;
  const int64_t T8 = tiles8(W, H), NB = pre_blocks(N) + 1, NCB = coarse_bins(W, H);
  char* p = (char*)base;
  size_t off = 0;
  t.hdr = (unsigned long long*)(p + off); off += HDR_WORDS * 8;
  t.dup_pool = (unsigned long long*)(p + off); off += align_up((size_t)DUP_POOLS * DP_STRIDE * 8, 256);
  t.coarse_count = (uint32_t*)(p + off); off += align_up((size_t)NCB * CC_STRIDE * 4, 256);
  t.zero_bytes = off;
  t.tile_range = (uint2*)(p + off); off += align_up((size_t)T8 * 8, 256);
  t.long_tiles = (uint32_t*)(p + off); off += align_up((size_t)T8 * 4, 256);
  t.block_nvis = (uint32_t*)(p + off); off += align_up((size_t)NB * 4, 256);
  t.block_dref = (unsigned long long*)(p + off); off += align_up((size_t)NB * 8, 256);
  const size_t msz = align_up((size_t)scatter_groups(N) * NCB * 4, 256);
  t.sc_cnt = (uint32_t*)(p + off); off += msz;
  t.sc_hits = (uint32_t*)(p + off); off += msz;
  t.sc_base = (uint32_t*)(p + off); off += msz;
  t.tile_order = (uint32_t*)(p + off); off += align_up((size_t)(2 * 8 * order_slots(W, H)) * 4, 256);
  if (total) *total = off;
  return t;
}

// A Gaussian that covers thousands of tiles owns thousands of per-duplicate gradient records: preprocess_bwd would sum
// them with ONE wave while the rest of the chip idles (2 000 screen-filling splats: 11.7 ms). Such Gaussians (more than
// BWD_BIG duplicates) are listed by the forward in 1024-record chunks; the backward first reduces every chunk to its
// head record with one workgroup per chunk (dupgrad_reduce_kernel), preprocess_bwd then adds the few heads.
constexpr unsigned BWD_BIG = 2048, BWD_CHUNK = 1024;
static inline size_t big_chunk_capacity(int64_t D) { return (size_t)(D / 512 + 64); }

struct BinsView {
  uint4* slabs;          // [NCB][coarse_capacity] coarse items (Gaussian id, depth bits, first dup index, 16-bit tile mask)
                         //     appended DIRECTLY, with a device atomic per item: those of the splats reaching more than
                         //     BIG_WALK coarse bins (walked by a whole wave: preprocess_kernel / big_walk_kernel), every item of
                         //     the one-pass binning, a merged plan's. The two-pass binning's items are in `csr`.
  uint4* csr;            // two-pass binning: the coarse items sorted by bin, every bin's run contiguous and exactly sized
                         //     (coarse_count words 6, 7) -- no per-bin capacity, memory in proportion to the items. Behind
                         //     everything else in the blob (bins_view_csr: only the plan / render / export stages, which know
                         //     N, use it); capacity = the pair list's = PAIRS_PER_BLOCK per preprocess workgroup
  uint2* big_chunks;     // [D / 512 + 64] (Gaussian id, chunk index) of the chunks described above
  uint4* items;          // [D] per-tile segments, unsorted: (Gaussian id, depth bits, dup index, 0)
  uint32_t* sorted_id;   // [D] per-tile lists of Gaussian ids, front to back
  uint32_t* sorted_dup;  // [D] the matching duplicate indices
  uint4* pairs;          // = items: [NB][PAIRS_PER_BLOCK] two-pass binning (PLAN stage only): the coarse items of preprocess
                         //     workgroup b, densely from pairs[b][0] -- (Gaussian id, depth bits, first dup, tile mask |
                         //     coarse bin << 16) -- written with plain stores; bin_scatter_kernel moves them into the slabs
                         //     before anything of items / sorted_id / sorted_dup (render stage) is written. May reach
                         //     beyond sorted_dup: bins_bytes_plan() sizes the blob for it.
};
// render / backward / export / merge: what the per-tile arrays need
static inline size_t bins_bytes(int64_t D, int64_t NCB, int64_t coarse_cap) {
  return align_up((size_t)NCB * coarse_cap * 16, 256) + align_up(big_chunk_capacity(D) * 8, 256) +
         align_up((size_t)D * 16, 256) + 2 * align_up((size_t)D * 4, 256);
}
// the plan of N Gaussians (and therefore what sfgs_raster_sizes / sfgs_raster_scratch_layout report): room for the pair list
// ... with the binning's arrays: the pair list (on top of the lists) and, behind both, the bin-sorted items (BinsView::csr)
static inline size_t bins_csr_offset(int64_t D, int64_t NCB, int64_t coarse_cap, int64_t N) {
  const size_t lists = align_up((size_t)D * 16, 256) + 2 * align_up((size_t)D * 4, 256);
  return align_up((size_t)NCB * coarse_cap * 16, 256) + align_up(big_chunk_capacity(D) * 8, 256) +
         (lists > pairs_bytes(N) ? lists : pairs_bytes(N));
}
static inline size_t bins_bytes_plan(int64_t D, int64_t NCB, int64_t coarse_cap, int64_t N) {
  return bins_csr_offset(D, NCB, coarse_cap, N) + pairs_bytes(N);
}
static inline BinsView bins_view(void* base, int64_t D, int64_t NCB, int64_t coarse_cap) {
  BinsView b;
  char* p = (char*)base;
  b.slabs = (uint4*)p; p += align_up((size_t)NCB * coarse_cap * 16, 256);
  b.big_chunks = (uint2*)p; p += align_up(big_chunk_capacity(D) * 8, 256);
  b.items = (uint4*)p; p += align_up((size_t)D * 16, 256);
  b.sorted_id = (uint32_t*)p; p += align_up((size_t)D * 4, 256);
  b.sorted_dup = (uint32_t*)p;
  b.pairs = b.items;
  b.csr = nullptr;
  return b;
}
// the view of the stages that know N (plan, render, export): with the bin-sorted item array
static inline BinsView bins_view_csr(void* base, int64_t D, int64_t NCB, int64_t coarse_cap, int64_t N) {
  BinsView b = bins_view(base, D, NCB, coarse_cap);
  b.csr = (uint4*)((char*)base + bins_csr_offset(D, NCB, coarse_cap, N));
  return b;
}

#ifdef __HIPCC__
// A coarse bin's items as its consumers see them: the run in BinsView::csr first (two-pass binning), then what was appended
// directly to the bin's slab (clamped to the slab's capacity: an overflowing plan is flagged, never read past).
struct BinItems {
  const uint4* csr;   // items 0 .. nc-1
  const uint4* uni;   // items nc .. n-1, addressed uni[i] (the pointer is pre-shifted by -nc)
  unsigned nc, n;
  __device__ __forceinline__ uint4 operator[](unsigned i) const { return *(i < nc ? csr + i : uni + i); }
};
__device__ __forceinline__ BinItems bin_items(const uint32_t* __restrict__ line, const uint4* __restrict__ csr,
                                              const uint4* __restrict__ slabs, size_t cb, unsigned coarse_capacity) {
  BinItems b;
  b.nc = csr ? line[7] : 0u;
  const unsigned direct = line[0] - b.nc;
  b.n = b.nc + (direct < coarse_capacity ? direct : coarse_capacity);
  b.csr = csr + line[6];
  b.uni = slabs + cb * (size_t)coarse_capacity - b.nc;
  return b;
}
#endif

constexpr int LIST_ALIGN = 64;  // every tile list starts on a multiple of 64 slots (fine_bin_kernel)

struct ImageView {
  uint32_t* n_contrib;  // [P] list position (1-based) of the last contributor
  float* final_T;       // [P]
  float* dacc;          // [P] un-normalised accumulated depth
  uint2* hitmask;       // [slot capacity] word (first slot of the tile + 64 b + lane) = which of the 64 entries of the
                        //   tile's b-th batch pixel `lane` BLENDED (bit j of .x: entry j, bit j of .y: entry 32 + j)
  uint32_t* tile_kmax;  // [T8] list position (1-based) of the tile's LAST contributor = max of n_contrib over its pixels
  uint16_t* tile_dead;  // [T8] min(65535, list entries behind the last contributor)
};
static inline size_t image_bytes(int W, int H, int64_t D) {
  return 3 * align_up((size_t)W * H * 4, 256) + align_up((size_t)D * 8, 256) +
         align_up((size_t)((W + TILE_BIN - 1) / TILE_BIN) * ((H + TILE_BIN - 1) / TILE_BIN) * 4, 256) +
         align_up((size_t)((W + TILE_BIN - 1) / TILE_BIN) * ((H + TILE_BIN - 1) / TILE_BIN) * 2 + 16, 256);
}
static inline ImageView image_view(void* base, int W, int H, int64_t D) {
  ImageView v;
  char* p = (char*)base;
  const size_t plane = align_up((size_t)W * H * 4, 256);
  v.n_contrib = (uint32_t*)p; v.final_T = (float*)(p + plane); v.dacc = (float*)(p + 2 * plane);
  v.hitmask = (uint2*)(p + 3 * plane);
  v.tile_kmax = (uint32_t*)(p + 3 * plane + align_up((size_t)D * 8, 256));
  v.tile_dead = (uint16_t*)((char*)v.tile_kmax +
                            align_up((size_t)((W + TILE_BIN - 1) / TILE_BIN) * ((H + TILE_BIN - 1) / TILE_BIN) * 4, 256));
  return v;
}

// Backward: do the dead entries make up more than a quarter of the frame's duplicates? (Measured break-even on near-nadir city
// views, profiles/r6_live_flags_ab.txt: 18 % dead -> flags +1.5 %, 23 % -> equal, 33 % -> -2 %.) Then the frame runs in LIVE-FLAG mode:
// dupgrad_prefill_kernel clears one byte per duplicate index, composite_bwd sets the byte of every record it writes and
// writes nothing for the dead entries, and the readers (dupgrad_reduce_kernel, preprocess_bwd) fetch only flagged records.
__host__ __device__ inline bool prefill_wanted(unsigned long long dead, unsigned long long n_dup) { return dead * 4ull > n_dup; }

// per-duplicate gradient record. DG_F4 = 3: 48 bytes, floats in Grad2D order. 4: 64 bytes, float 3 k + r of the
// record at position 4 r + k (every fourth float is padding): lane (entry, row r) of composite_bwd then owns one whole
// 16-byte quarter and the four lanes of an entry write a full 64-byte sector with ONE store instruction
constexpr int DG_F4 = 3;
constexpr int DUPGRAD_FLOATS = 4 * DG_F4;
// the backward's scratch blob: [D records][D live flags, one byte each][one line of zeros (what a dead record reads as)]
static inline size_t dupgrad_flags_offset(int64_t D) { return align_up((size_t)D * DUPGRAD_FLOATS * 4, 256); }
static inline size_t dupgrad_zero_offset(int64_t D) { return dupgrad_flags_offset(D) + align_up((size_t)D, 256); }
static inline size_t dupgrad_bytes(int64_t D) { return dupgrad_zero_offset(D) + 256; }

// pools in use for a frame of NB preprocess workgroups: small frames (few workgroups: their totals would not spread evenly
// over the pools) draw from one pool that owns the whole index space
__host__ __device__ inline unsigned dup_pools_used(long long NB) { return NB >= 512 ? (unsigned)DUP_POOLS : 1u; }
__device__ __forceinline__ unsigned long long dup_alloc(unsigned long long* __restrict__ dup_pool, unsigned npools,
                                                        unsigned chooser, unsigned total,
                                                        unsigned long long dup_capacity, bool* fits) {
  const unsigned pool = chooser % npools;
  const unsigned long long R = dup_capacity / npools;
  const unsigned long long old = atomicAdd(&dup_pool[pool * DP_STRIDE], (unsigned long long)total);
  *fits = old + total <= R;
  return pool * R + old;
}

// ---- XCD-aware block remap (bijective; guide T1) -------------------------------------------------
// Hardware places workgroup b on XCD b % 8; give every XCD a contiguous range of logical blocks so
// that neighbouring tiles (which gather the same splat records) share an L2.
__device__ __forceinline__ unsigned xcd_remap(unsigned b, unsigned nblk) {
  const unsigned xcd = b & 7u, idx = b >> 3;
  const unsigned q = nblk >> 3, r = nblk & 7u;
  const unsigned base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

// Waves per workgroup of the two compositing kernels (compile-time constant; 4 = one 16x16
// super-tile per workgroup). With 1 every 8x8 tile is a workgroup of its own: the four waves of a super-tile never
// synchronise anyway, and a one-wave workgroup gives its LDS and wave slot back the moment ITS list is done instead of
// when the longest of four lists is (2: half a super-tile). Workgroups stay XCD-contiguous in super-tile order.
constexpr int CWG_WAVES = 1;   // composite_fwd (round 6, with the chunked XCD mapping: -2 ... -3.5 % against 4 on the headline and on
                               // every regime, profiles/r6_wg_waves_chunks_ab.txt); composite_bwd keeps four, see BWG_WAVES
// composite_bwd's own value (round 4 experiment: 8 = 4x2 tiles, 16 = 4x4 tiles = a coarse bin per workgroup, so that the
// tiles that gather the same records share a CU's L1; one-wave workgroups measured +3.8 %: profiles/r4_bwd_lds18_ab_not_kept.txt)
constexpr int BWG_WAVES = 4;   // (1 and 2 measured again in round 6: +6 ... +7 % on the headline, nothing gained on the skewed regimes)
static_assert(CWG_WAVES == 4 || CWG_WAVES == 2 || CWG_WAVES == 1, "compositing workgroups: 4, 2 or 1 of a super-tile's tiles");
static_assert(BWG_WAVES == 16 || BWG_WAVES == 8 || BWG_WAVES == 4 || BWG_WAVES == 2 || BWG_WAVES == 1, "composite_bwd workgroups");
// A workgroup's tiles come from a BLOCK of CBLK x CBLK tiles (2 x 2 = the 16x16-pixel super-tile; 4 x 4 for workgroups of
// 8 / 16 waves); a block is split over CBLK^2 / WAVES workgroups.
template <int WAVES> constexpr int composite_block_edge() { return WAVES > 4 ? 4 : 2; }
// Which block of tiles a compositing workgroup takes. Hardware places workgroup b on XCD b % 8. Until round 6 every XCD owned
// ONE contiguous eighth of the image (xcd_remap): perfect for the L2, and a load-balance disaster on any frame that is not
// uniform -- the wave timeline of an orbit view of a city at 25 degrees elevation (tools/timeline.py, profiles/r6_*) has one XCD
// finish composite_fwd after 13 us and another after 299 us. Now the image is cut into CHUNKS of 8 x 4 blocks (128 x 64 pixels
// with 2 x 2-tile blocks; 32 workgroups that run back to back on ONE XCD and share its L2 like before) and the chunks are
// dealt to the XCDs round-robin in row-major order: every XCD gets every eighth chunk, spread over the whole frame.
constexpr int CHUNK_BX = 8, CHUNK_BY = 4;   // (order_slots above spells these out: it is needed before this point)
static_assert(CHUNK_BX == 8 && CHUNK_BY == 4, "order_slots(): 8 x 4 blocks of 2 x 2 tiles = 128 tiles per chunk");
static inline unsigned composite_grid(int SX, int SY, int per_block) {   // workgroups to launch (padded: surplus ones return at once)
  const unsigned nch = (unsigned)((SX + CHUNK_BX - 1) / CHUNK_BX) * (unsigned)((SY + CHUNK_BY - 1) / CHUNK_BY);
  return (nch + 7u) / 8u * 8u * (unsigned)(CHUNK_BX * CHUNK_BY * per_block);
}
#ifdef __HIPCC__
// block (sbx, sby), wave-in-block `wave` and wave-in-workgroup `lw` (its slice of the workgroup's LDS) of the calling
// wave; all wave-uniform: the tile, its list range and every loop bound derived from them become SGPRs. false: a surplus
// workgroup of the padded grid (nothing to do).
template <int WAVES = CWG_WAVES>
__device__ __forceinline__ bool composite_wave_role(int SX, int SY, int& sbx, int& sby, int& wave, int& lw) {
  constexpr int E = composite_block_edge<WAVES>();
  constexpr unsigned PER = E * E / WAVES;   // workgroups per block
  constexpr unsigned CS = CHUNK_BX * CHUNK_BY * PER;
  const unsigned b = blockIdx.x, idx = b >> 3;
  const unsigned c = (idx / CS) * 8u + (b & 7u), within = idx % CS;   // chunk, position inside it
  const unsigned SXc = (unsigned)(SX + CHUNK_BX - 1) / CHUNK_BX;
  const unsigned st = within / PER;
  sbx = (int)((c % SXc) * CHUNK_BX + st % CHUNK_BX);
  sby = (int)((c / SXc) * CHUNK_BY + st / CHUNK_BX);
  lw = WAVES == 1 ? 0 : __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  wave = (int)(within % PER) * WAVES + lw;
  return sbx < SX && sby < SY;
}

// With a tile ORDER (TilesView::tile_order, one list per XCD: the tiles of the XCD's chunks, longest list first) the workgroup
// on XCD x at position p of that XCD takes order[x * P + p] instead; a 4-wave workgroup takes four consecutive entries
// (lists of similar length: the workgroup's LDS is released when all four are done). Returns the tile or 0xffffffff.
template <int WAVES>
__device__ __forceinline__ unsigned ordered_tile(const uint32_t* __restrict__ order, unsigned P, int& lw) {
  static_assert(WAVES <= 4 && composite_block_edge<WAVES>() == 2, "order_slots() counts tiles of 2 x 2-tile blocks: grid x WAVES == 8 P");
  const unsigned b = blockIdx.x;
  lw = WAVES == 1 ? 0 : __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const unsigned p = (b >> 3) * (unsigned)WAVES + (unsigned)lw;
  return p < P ? order[(size_t)(b & 7u) * P + p] : 0xffffffffu;
}

// the same idea in one dimension (select_sort_kernel: workgroups = rows of four tiles of the coarse bins, 4 per bin): chunks of
// CS consecutive workgroups, dealt to the XCDs round-robin; the last < 8 CS workgroups keep their own index. Bijective.
__device__ __forceinline__ unsigned xcd_chunked(unsigned b, unsigned n, unsigned CS) {
  const unsigned nfull = n / (8u * CS) * (8u * CS);
  if (b >= nfull) return b;
  const unsigned idx = b >> 3;
  return ((idx / CS) * 8u + (b & 7u)) * CS + idx % CS;
}
#endif

// ---- [N,3] rows (means, scales, colours and their gradients): ONE 12-byte access per thread instead of three dword accesses
// 12 bytes apart (round 4; cf. profiles/r4_bwd_store3_ab.txt: partial scattered accesses cost per instruction, not per byte)
typedef float sfgs_v3f __attribute__((ext_vector_type(3), aligned(4)));
__device__ __forceinline__ void load3(const float* __restrict__ src, float (&dst)[3]) {
  const sfgs_v3f v = *reinterpret_cast<const sfgs_v3f*>(src);
  dst[0] = v.x; dst[1] = v.y; dst[2] = v.z;
}
__device__ __forceinline__ void store3(float* __restrict__ dst, float a, float b, float c) {
  sfgs_v3f v; v.x = a; v.y = b; v.z = c;
  *reinterpret_cast<sfgs_v3f*>(dst) = v;
}

// ---- per-Gaussian SH coefficient rows: 3K floats, 16-byte vector accesses when the row size allows -----------
template <int CNT>
__device__ __forceinline__ void load_row(const float* __restrict__ src, float (&dst)[CNT]) {
  if constexpr (CNT % 4 == 0) {
#pragma unroll
    for (int i = 0; i < CNT / 4; ++i) {
      const float4 v = reinterpret_cast<const float4*>(src)[i];
      dst[4 * i] = v.x; dst[4 * i + 1] = v.y; dst[4 * i + 2] = v.z; dst[4 * i + 3] = v.w;
    }
  } else if constexpr (CNT % 3 == 0) {   // 3 K floats with K = 1, 9, 25: 12-byte accesses instead of dwords a row apart
#pragma unroll
    for (int i = 0; i < CNT / 3; ++i) {
      const sfgs_v3f v = *reinterpret_cast<const sfgs_v3f*>(src + 3 * i);   // (sizeof(sfgs_v3f) is 16: index in floats)
      dst[3 * i] = v.x; dst[3 * i + 1] = v.y; dst[3 * i + 2] = v.z;
    }
  } else {
#pragma unroll
    for (int i = 0; i < CNT; ++i) dst[i] = src[i];
  }
}
template <int CNT>
__device__ __forceinline__ void store_row(float* __restrict__ dst, const float (&src)[CNT]) {
  if constexpr (CNT % 4 == 0) {
#pragma unroll
    for (int i = 0; i < CNT / 4; ++i)
      reinterpret_cast<float4*>(dst)[i] = make_float4(src[4 * i], src[4 * i + 1], src[4 * i + 2], src[4 * i + 3]);
  } else if constexpr (CNT % 3 == 0) {
#pragma unroll
    for (int i = 0; i < CNT / 3; ++i) {
      sfgs_v3f v; v.x = src[3 * i]; v.y = src[3 * i + 1]; v.z = src[3 * i + 2];
      *reinterpret_cast<sfgs_v3f*>(dst + 3 * i) = v;
    }
  } else {
#pragma unroll
    for (int i = 0; i < CNT; ++i) dst[i] = src[i];
  }
}

// A Gaussian's K coefficient rows (coefficient-major: 3 floats per coefficient) from ONE array [N,K,3] or, split storage
// (SfgsGaussians.shs_rest), coefficient 0 from `first` [N,1,3] and the others from `rest` [N,K-1,3]: the same K 12-byte
// accesses either way -- only two wave-uniform base pointers differ, no branch. preprocess uses this form for both
// layouts; preprocess_bwd only in its split instantiations (a branch between the float4-grouped and the 3-float-grouped
// form of the row cost its UNSPLIT path 20 % at 16 coefficients, this form 10 %: profiles/r4_split_sh_rows_ab.txt).
template <int K>
__device__ __forceinline__ void load_sh_rows(const float* __restrict__ first, const float* __restrict__ rest, size_t g,
                                             float (&dst)[3 * K]) {
  const float* r0 = first + (rest ? 3 : 3 * K) * g;
  const float* rk = rest ? rest + (size_t)(3 * K - 3) * g - 3 : r0;
  load3(r0, reinterpret_cast<float(&)[3]>(dst[0]));
#pragma unroll
  for (int k = 1; k < K; ++k) load3(rk + 3 * k, reinterpret_cast<float(&)[3]>(dst[3 * k]));
}
template <int K>
__device__ __forceinline__ void store_sh_rows(float* __restrict__ first, float* __restrict__ rest, size_t g,
                                              const float (&src)[3 * K]) {
  float* r0 = first + (rest ? 3 : 3 * K) * g;
  float* rk = rest ? rest + (size_t)(3 * K - 3) * g - 3 : r0;
  store3(r0, src[0], src[1], src[2]);
#pragma unroll
  for (int k = 1; k < K; ++k) store3(rk + 3 * k, src[3 * k], src[3 * k + 1], src[3 * k + 2]);
}

// Launch `KERNEL<K, DEG>` for the runtime (sh_coeffs, sh_degree) pair; K = 0 is the colors_precomp path.
#define SFGS_DISPATCH_SH(K_RT, DEG_RT, LAUNCH)                                   \
  do {                                                                           \
    const int k_ = (K_RT), d_ = (DEG_RT);                                        \
    if (k_ == 0) { LAUNCH(0, 0); }                                               \
    else if (k_ == 1) { LAUNCH(1, 0); }                                          \
    else if (k_ == 4) { if (d_ == 0) { LAUNCH(4, 0); } else { LAUNCH(4, 1); } }  \
    else if (k_ == 9) { if (d_ == 0) { LAUNCH(9, 0); } else if (d_ == 1) { LAUNCH(9, 1); } else { LAUNCH(9, 2); } } \
    else if (k_ == 16) { if (d_ == 0) { LAUNCH(16, 0); } else if (d_ == 1) { LAUNCH(16, 1); } else if (d_ == 2) { LAUNCH(16, 2); } else { LAUNCH(16, 3); } } \
    else { if (d_ == 0) { LAUNCH(25, 0); } else if (d_ == 1) { LAUNCH(25, 1); } else if (d_ == 2) { LAUNCH(25, 2); } else if (d_ == 3) { LAUNCH(25, 3); } else { LAUNCH(25, 4); } } \
  } while (0)

// ---- small wave / block primitives ---------------------------------------------------------------
__device__ __forceinline__ unsigned lane_id() { return __lane_id(); }

__device__ __forceinline__ unsigned wave_incl_scan_u32(unsigned v) {
  const unsigned lane = lane_id();
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const unsigned t = __shfl_up(v, d);
    if (lane >= (unsigned)d) v += t;
  }
  return v;
}

// exclusive scan across a block of NT threads; smem must hold NT/64 + 1 words. Returns the
// exclusive prefix of `v`; *total = block sum (all threads).
template <int NT>
__device__ __forceinline__ unsigned block_excl_scan_u32(unsigned v, unsigned* total, unsigned* smem) {
  const unsigned lane = lane_id(), wave = threadIdx.x >> 6;
  const unsigned incl = wave_incl_scan_u32(v);
  if (lane == 63) smem[wave] = incl;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned run = 0;
    for (int w = 0; w < NT / 64; ++w) { const unsigned t = smem[w]; smem[w] = run; run += t; }
    smem[NT / 64] = run;
  }
  __syncthreads();
  const unsigned res = smem[wave] + incl - v;
  *total = smem[NT / 64];
  __syncthreads();
  return res;
}

// composite_bwd.hip (a translation unit of its own: compiled with another scheduling strategy, see the Makefile)
void launch_composite_bwd(unsigned grid, hipStream_t stream, KFrame kf, int TX8, int TY8, int SX, int SY,
                          const uint2* tile_range, const uint32_t* sorted_id, const uint32_t* sorted_dup, const float4* rec,
                          const uint32_t* n_contrib, const float* final_T, const float* dacc, const float* dL_dcolor,
                          const float* dL_ddepth, const float* dL_dalpha, const uint2* hitmask, const uint32_t* tile_kmax,
                          float4* dupgrad, uint8_t* live, const unsigned long long* hdr, int not_prefilled,
                          const uint32_t* order, unsigned order_slots_per_xcd);

}  // namespace sfgs
