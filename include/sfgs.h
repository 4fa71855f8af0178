/*
 * sfgs.h -- C ABI of libsfgs.so, the MI355X (gfx950) Gaussian-splat rasterizer hot path.
 *
 * This header is the drop-in boundary. Everything here is plain C: device pointers, sizes and
 * a HIP stream handle passed as void*. No torch types, no C++ types. The Python packages
 * diff_gauss / fused_ssim / simple_knn (skyfall-gs_amd/) bind these symbols with ctypes; the
 * binding a maintainer of the reference would add is shown in INTEGRATION.md.
 *
 * Reference interface each entry point replaces (paths relative to the reference tree):
 *
 *   sfgs_raster_*      diff_gauss.GaussianRasterizer.__call__      gaussian_renderer/__init__.py:57,132-140
 *                      diff_gauss.GaussianRasterizationSettings    gaussian_renderer/__init__.py:40-55
 *                      autograd backward of that call              train.py:279,845
 *   sfgs_ssim_*        fused_ssim.fused_ssim(img1, img2)           train.py:42,222,778
 *   sfgs_knn_dist2     simple_knn._C.distCUDA2(points)             scene/gaussian_model.py:25,324
 *   sfgs_prepass_*     GaussianModel.get_*_with_3D_filter/get_rotation  scene/gaussian_model.py:207-249 (next row)
 *
 * Ownership: every buffer (inputs, outputs, gradients, scratch "blobs") is allocated by the
 * caller (PyTorch's caching allocator on the right device). The library never allocates or
 * frees device memory and keeps no device state between calls, so it is thread-safe per
 * stream by construction (the autograd backward runs on a different host thread).
 *
 * Errors: every function returns SFGS_OK (0) or a negative SfgsStatus. Nothing throws across
 * the ABI, nothing calls exit(). sfgs_last_error() returns a thread-local message.
 */
#ifndef SFGS_H
#define SFGS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SFGS_ABI_VERSION 18

typedef enum SfgsStatus {
  SFGS_OK = 0,
  SFGS_E_ARG = -1,         /* bad argument (null pointer, negative size, struct_size mismatch) */
  SFGS_E_HIP = -2,         /* a HIP runtime call or kernel launch failed                       */
  SFGS_E_CAPACITY = -3,    /* a caller-owned blob is smaller than the plan requires            */
  SFGS_E_UNSUPPORTED = -4  /* e.g. sh_degree > 4, image wider than 65535 tiles                 */
} SfgsStatus;

/* Depth output convention (SURVEY 8c "unpinned semantic decisions"). */
#define SFGS_DEPTH_NORMALISED 0 /* depth = sum(T a z) / (1 - T_final); NaN where nothing hit (default) */
#define SFGS_DEPTH_RAW 1        /* depth = sum(T a z)                                                  */

/* One camera frame = diff_gauss.GaussianRasterizationSettings (14 fields,
 * gaussian_renderer/__init__.py:40-55). Matrix pointers are DEVICE pointers to the 16 floats of
 * the reference's row-major *transposed* matrices (scene/cameras.py:62-73):
 * p_view = [p,1] * viewmatrix, p_hom = [p,1] * projmatrix. */
typedef struct SfgsFrame {
  uint32_t struct_size;          /* = sizeof(SfgsFrame), ABI versioning                         */
  int32_t image_height;
  int32_t image_width;
  float tanfovx;
  float tanfovy;
  float kernel_size;             /* Mip-Splatting 2D filter variance (arguments/__init__.py:111) */
  float scale_modifier;
  int32_t sh_degree;             /* active SH degree 0..4 (sh_coeffs 1 / 4 / 9 / 16 / 25 stored)     */
  int32_t sh_coeffs;             /* coefficients stored per Gaussian in `shs` (max_degree+1)^2   */
  int32_t prefiltered;           /* accepted, ignored (reference always passes False)            */
  int32_t debug;                 /* !=0: synchronise + check after every launch                  */
  int32_t depth_mode;            /* SFGS_DEPTH_*                                                 */
  int32_t tile_row_begin;        /* band rendering (multi-GPU joint render): only 8-pixel  */
  int32_t tile_row_end;          /* tile rows [begin, end) are binned/composited; end <= 0 = all rows.
                                    Pixels outside the band are left untouched in the outputs.      */
  const float* subpixel_offset;  /* device [H,W,2] or NULL (= zeros)                             */
  const float* bg;               /* device [3]                                                   */
  const float* viewmatrix;       /* device [16]                                                  */
  const float* projmatrix;       /* device [16]                                                  */
  const float* campos;           /* device [3]                                                   */
  uint32_t launch_hints;         /* SFGS_HINT_* bits, 0 = none (see below)                       */
  void* feedback;                /* optional: 64 bytes of DEVICE memory, caller-owned and persistent across the frames of
                                    one stream (zero it once). The render / backward stages leave the frame's late
                                    statistics there -- they only exist once those stages have run -- and the NEXT
                                    sfgs_raster_forward_plan reports them (SfgsRasterCounters.prev_*): the feedback the
                                    launch hints are chosen from.                                                */
} SfgsFrame;

/* Launch hints. A frame's optional kernels -- the huge-splat walk, the three long-list sorts, the backward's dead-entry
 * prefill and chunk pre-reduction -- each cost ~5 us of queue time even when they find nothing to do, and whether they
 * have work is only known on the device. The caller may assert what it learnt from the previous frames:
 *   NO_HUGE_SPLATS  plan: do not launch the huge-splat walk. If SfgsRasterCounters.num_huge_splats comes back != 0 the
 *                   frame was binned WITHOUT those splats: redo plan + render without the hint.
 *   FEW_LONG_LISTS  render: ONE catch-all kernel sorts every list longer than 512 entries instead of three size-class
 *                   kernels. Always correct; slower when many lists are long.
 *   SHORT_LISTS     render: the caller expects per-tile lists of a few hundred entries at most and coarse bins of a few
 *                   thousand items (what the previous frame's plan / feedback reported): fine binning and the sort of the
 *                   short lists run as ONE kernel that never writes per-tile items to memory (select_sort_kernel). Always
 *                   correct -- both routes build bit-identical lists --; slower than the two-kernel route when lists
 *                   or bins are long.
 *   MEDIUM_LISTS    render: as SHORT_LISTS, for frames whose lists reach 513 .. 1 024 entries (what the feedback of the
 *                   previous frame reported): the same kernel with twice the list capacity (48 KB of LDS per workgroup,
 *                   the 16-key register network for the lists beyond 512). Always correct; implies the fused route.
 *   LISTS_768       render, with MEDIUM_LISTS: the fused kernel with room for 768 instead of 1 024 entries per list (36 KB of LDS
 *                   per workgroup: four instead of three workgroups per CU) -- for frames whose lists exceed 512 but (almost)
 *                   never 768 entries; the few longer ones take the long-list kernels. Always correct.
 *   TILE_ORDER      render: the caller expects tile lists of very different lengths (the previous frame's longest list
 *                   several times its mean: a city seen from above, a few facades edge-on). The compositing kernels then
 *                   take their tiles longest list first within each XCD's share of the image (two launches of a small
 *                   ordering kernel) instead of in image order, so that the kernels' last stretch is filled with SHORT
 *                   lists. Always correct: tiles are independent, the results are the same bits in any order.
 *   NO_PREFILL      backward: do not launch the dead-entry prefill kernel (its decision then reads "no"). Always correct.
 *   NO_BIG_CHUNKS   backward: do not launch the chunk pre-reduction. Only valid when THIS frame's plan reported
 *                   num_big_chunks == 0. */
#define SFGS_HINT_NO_HUGE_SPLATS 1u
#define SFGS_HINT_FEW_LONG_LISTS 2u
#define SFGS_HINT_NO_PREFILL 4u
#define SFGS_HINT_NO_BIG_CHUNKS 8u
#define SFGS_HINT_SHORT_LISTS 16u
#define SFGS_HINT_MEDIUM_LISTS 32u
#define SFGS_HINT_TILE_ORDER 64u
#define SFGS_HINT_LISTS_768 128u

/* Per-Gaussian inputs = keyword arguments of GaussianRasterizer.__call__
 * (gaussian_renderer/__init__.py:132-140). All float32, contiguous, device memory.
 * Exactly one of colors_precomp / shs is non-NULL. cov3Ds_precomp must be NULL: that path is
 * dead in the reference (scales.float() at :138 dereferences None when it would be used). */
typedef struct SfgsGaussians {
  uint32_t struct_size;
  int32_t count;                 /* N                                              */
  const float* means3D;          /* [N,3]                                          */
  const float* scales;           /* [N,3]  (already 3D-filtered, activated)        */
  const float* rotations;        /* [N,4]  (w,x,y,z), normalised by the caller     */
  const float* opacities;        /* [N,1]  (already 3D-filter compensated)         */
  const float* colors_precomp;   /* [N,3] or NULL                                  */
  const float* shs;              /* [N,sh_coeffs,3] or NULL                        */
  /* RAW-PARAMETER MODE (the activations + Mip-Splatting 3D filter folded into preprocess and
   * preprocess_bwd). When filter_3D is non-NULL, `scales`, `rotations` and `opacities` above are the model's RAW
   * parameters -- _scaling, _rotation, _opacity of scene/gaussian_model.py -- and the library evaluates
   * get_scaling_with_3D_filter (:207-213), get_opacity_with_3D_filter (:237-249) and get_rotation (:216-217) itself, with
   * the same float sequence as sfgs_prepass_forward (the frame equals, bit for bit, the one rendered from that function's
   * outputs). The backward then writes the gradients of the RAW parameters (sfgs_prepass_backward's results) into
   * SfgsGaussianGrads.scales / rotations / opacities. */
  const void* filter_3D;         /* [N,1] float32 or float64, or NULL = activated inputs (the mode above this comment) */
  int32_t raw_f64_mask;          /* bit 0: filter_3D is float64; bit 1: `opacities` points to float64 raw opacities
                                    (the reference's _opacity after its first reset_opacity, :483-501); 0 when
                                    filter_3D is NULL */
  /* EVAL_SH-FOLDED COLOUR PATH (ABI 12; SURVEY 8f row 1, last part). render()'s Python colour paths -- the appearance
   * MLP's toned coefficients and `convert_SHs_python` -- compute
   *     colors_precomp = clamp_min(eval_sh(active_sh_degree, sh[N,3,K], dirs[N,3]) + 0.5, 0)
   * (gaussian_renderer/__init__.py:112-118,121-125; utils/sh_utils.py:57-112) with torch kernels and hand the result over
   * as colors_precomp. When sh_dirs is non-NULL, `shs` above is that CHANNEL-MAJOR [N,3,sh_coeffs] coefficient tensor and
   * (shs_channel_major = 1) and sh_dirs the `dirs` argument ([N,3], used as given: eval_sh does not normalise it either), and preprocess /
   * preprocess_bwd evaluate the expression themselves (degrees 0-4): no N x 3 intermediate, no eval_sh launch. The
   * backward writes SfgsGaussianGrads.shs channel-major and the direction gradient to SfgsGaussianGrads.sh_dirs (it does
   * NOT flow into grads.means3D: `dirs` is an input of its own, the caller's graph carries it on). */
  const float* sh_dirs;          /* [N,3] or NULL = `shs` is [N,sh_coeffs,3] and the direction is normalize(means3D - campos) */
  int32_t shs_channel_major;     /* with sh_dirs: 1 = `shs` is eval_sh's [N,3,sh_coeffs]; 0 = it is [N,sh_coeffs,3] (what
                                    render()'s convert_SHs_python path passes to eval_sh is a transposed VIEW of the
                                    model's [N,K,3] features: hand over the features themselves). 0 without sh_dirs */
  /* SPLIT SH STORAGE (ABI 13). The reference's model keeps the SH coefficients as two parameters, _features_dc [N,1,3]
   * and _features_rest [N,K-1,3], and concatenates them on every get_features call (scene/gaussian_model.py:227-231; two
   * calls per render(): gaussian_renderer/__init__.py:114,121-122,127). When shs_rest is non-NULL, `shs` above holds
   * coefficient 0 only ([N,1,3]) and shs_rest coefficients 1 .. sh_coeffs-1 ([N,sh_coeffs-1,3]); preprocess reads the two
   * arrays, the backward writes SfgsGaussianGrads.shs ([N,1,3]) and .shs_rest: no concatenated copy, no split of its
   * gradient. Needs sh_coeffs > 1 and shs_channel_major == 0; with or without sh_dirs. */
  const float* shs_rest;         /* [N,sh_coeffs-1,3] or NULL = `shs` holds all sh_coeffs coefficients */
  /* DIRECTIONS FROM CENTRES (ABI 15). render()'s Python colour paths compute eval_sh's `dirs` as
   *     dir_pp = xyz - camera_center.repeat(N, 1);  dirs = dir_pp / dir_pp.norm(dim=1, keepdim=True)
   * (gaussian_renderer/__init__.py:114-115, :122-123): six elementwise / reduction launches and eight more on the way
   * back. sh_centers, an alternative to sh_dirs, hands over the subtrahend ([N,3], as render() built it -- any values)
   * instead of the finished directions: preprocess evaluates normalize(means3D - sh_centers) itself (the float sequence
   * of the in-kernel SH path's normalize(means3D - campos)) and preprocess_bwd adds the direction's gradient to
   * grads.means3D; grads.sh_dirs stays NULL. Everything else as with sh_dirs (shs layout, shs_channel_major, shs_rest). */
  const float* sh_centers;       /* [N,3] or NULL */
} SfgsGaussians;

/* Gradient outputs of the backward pass (all device, float32, fully overwritten). */
typedef struct SfgsGaussianGrads {
  uint32_t struct_size;
  float* means3D;        /* [N,3] */
  float* means2D;        /* [N,3]: cols 0:2 = dL/dmean2D in NDC units, col 2 = abs-grad magnitude
                            (scene/gaussian_model.py:744-749)                                    */
  float* scales;         /* [N,3] */
  float* rotations;      /* [N,4] */
  float* opacities;      /* [N,1]; raw-parameter mode with raw_f64_mask bit 1: [N,1] float64 (cast the pointer) */
  float* colors_precomp; /* [N,3] or NULL */
  float* shs;            /* [N,sh_coeffs,3] or NULL ([N,3,sh_coeffs] with SfgsGaussians.shs_channel_major) */
  float* sh_dirs;        /* [N,3]; non-NULL exactly when SfgsGaussians.sh_dirs is (NULL with sh_centers) */
  float* shs_rest;       /* [N,sh_coeffs-1,3]; non-NULL exactly when SfgsGaussians.shs_rest is (`shs` is then [N,1,3]) */
} SfgsGaussianGrads;

/* Sizes (bytes) of the caller-owned scratch blobs. */
typedef struct SfgsRasterSizes {
  uint32_t struct_size;
  size_t geom_bytes;     /* f(N):   per-Gaussian 2D records, duplicate offsets (72 bytes per Gaussian)   */
  size_t tiles_bytes;    /* f(W,H,N): counters, per-tile counts/offsets, per-block scan partials, the
                            binning radix pass's [workgroup][coarse bin] matrices                 */
  size_t bins_bytes;     /* f(D, coarse_capacity, N): coarse-bin slabs, per-tile duplicates, sorted lists; during
                            the plan the binning's pair list (96 bytes per Gaussian) lies on top of the per-tile
                            arrays, which only the render stage writes */
  size_t image_bytes;    /* f(W,H,D): per-pixel last contributor, final T, raw depth, and one 8-byte
                            blended-entries mask per pixel and 64-entry list batch (for backward) */
  size_t dupgrad_bytes;  /* f(D):   per-duplicate 2D gradient records + flags (backward only)    */
  int64_t coarse_bins;   /* number of 32x32-pixel coarse bins of this image (informational)      */
} SfgsRasterSizes;

/* Counters produced by the plan stage (host copy). */
typedef struct SfgsRasterCounters {
  int64_t num_duplicates;      /* D_eff: (Gaussian, 8x8 tile) pairs binned (opacity-aware test)             */
  int64_t num_duplicates_ref;  /* D:     sum of tiles_touched by the reference's 3-sigma 16x16 rule         */
  int64_t num_visible;         /* N_vis: count(radii > 0)                                                   */
  int64_t max_tile_list;       /* longest per-tile list (valid when read after sfgs_raster_forward_render)  */
  int64_t overflow;            /* != 0: dup_capacity or coarse_capacity was too small: redo the plan with
                                  dup_capacity >= sfgs_raster_slot_capacity(W, H, num_duplicates) and
                                  coarse_capacity >= max_coarse_bin. If both ALREADY hold, one of the 8 duplicate-index
                                  pools (each owns dup_capacity / 8 indices; a workgroup draws from pool id % 8) ran
                                  over because a few workgroups own most of the frame's duplicates: double
                                  dup_capacity. A caller should also treat max_coarse_bin > coarse_capacity as
                                  overflow (the library ORs it in; belt and braces)                         */
  int64_t max_coarse_bin;      /* most items appended DIRECTLY to one 32x32-pixel coarse bin's slab: those of the splats
                                  that reach more than 6 coarse bins (a whole wave walks them) or, with one-pass binning,
                                  every item. ABI 14: the items of the default two-pass binning no longer count -- they are
                                  stored bin-sorted and exactly sized, without a per-bin capacity -- so this is 0 for a
                                  frame of small splats, however skewed (a distant view: the scene in a few bins) */
  int64_t num_huge_splats;     /* splats reaching >= 64 coarse bins (their walk is a kernel of its own)        */
  int64_t num_big_chunks;      /* 1024-record chunks of Gaussians with > 2048 duplicates (backward pre-reduction) */
  int64_t prev_valid;          /* != 0: the prev_* fields below were read from SfgsFrame.feedback              */
  int64_t prev_long_tiles;     /* previous frame: tiles with lists longer than 512 entries                     */
  int64_t prev_max_tile_list;  /* previous frame: longest list                                                 */
  int64_t prev_prefilled;      /* previous frame's backward: the prefill kernel ran and chose to fill          */
  int64_t prev_tiles_over_512; /* previous frame: tiles with more than 512 entries, whichever route sorted them
                                  (prev_long_tiles counts the tiles LEFT to the long-list kernels: under MEDIUM_LISTS
                                  only those beyond 1 024): MEDIUM_LISTS pays when they are a sizeable part of the frame */
  int64_t max_bin_items;       /* ALL items of the fullest coarse bin (slab + bin-sorted run): what SFGS_HINT_SHORT_LISTS is
                                  chosen from. Filled by sfgs_raster_counters_decode / sfgs_raster_read_counters; 0 from
                                  sfgs_raster_read_counters_pinned (its 64 bytes end before it) */
} SfgsRasterCounters;

int sfgs_abi_version(void);
const char* sfgs_last_error(void);

/* Process-wide ROUTE options (ABI 16; tests and A/B runs -- never needed for correctness: every route builds bit-identical
 * results, tests/test_gpu_raster.py). They replace the getenv() calls the library used to make on every render: the
 * environment is read ONCE, when the library is loaded (SFGS_SORT, SFGS_PLAN_SCAN, SFGS_BINNING, SFGS_PREFILL, SFGS_KNN, SFGS_TILE_ORDER: same
 * values),
 * and changed afterwards only through this call. Thread-safe: one atomic word per option; a render running on another
 * thread sees the old or the new value, never a mixture.
 *   key "sort"       "auto" (the frame's launch hints decide) | "fused" | "fused768" | "fused1024" | "split"
 *   key "plan_scan"  "fused" (the plan's epilogues ride in the scatter launch) | "separate"
 *   key "binning"    "auto" (two-pass below 65 536 coarse bins) | "direct"
 *   key "prefill"    "auto" (dead-entry prefill decided per frame on the device) | "always" | "never"
 *   key "knn"        "auto" (spatial search above 4 096 points) | "brute" (the exact all-pairs kernel at every size)
 *   key "tile_order" "auto" (SFGS_HINT_TILE_ORDER decides) | "always" | "never": longest-first tile order of the compositing kernels
 * sfgs_set_option returns SFGS_E_ARG for an unknown key or value; sfgs_get_option returns the current value's name
 * (a string constant) or NULL for an unknown key. */
int sfgs_set_option(const char* key, const char* value);
const char* sfgs_get_option(const char* key);

/* Optional per-kernel timing (bench.py's roofline leg): when enabled, every kernel launch of the
 * library is bracketed by HIP events recorded on the launch stream. sfgs_profile_collect waits for
 * the recorded events, returns per-kernel summed milliseconds and launch counts (arrays of
 * sfgs_profile_kernel_count() entries) and clears the record. Off by default; costs nothing then. */
int sfgs_profile_enable(int32_t on);
int sfgs_profile_select(uint64_t kernel_mask); /* bit i = time kernel id i (default: all); events cost ~4 us each */
int sfgs_profile_kernel_count(void);
const char* sfgs_profile_kernel_name(int32_t id);
int sfgs_profile_collect(double* ms_sum, int64_t* launches, int32_t n);

/* Box probe (ABI 17; bench.py, before its timed region -- not on the rendering path): a fixed FP32 multiply-add loop at 8 waves
 * per SIMD on every CU, timed with HIP events on `stream` (synchronous: returns when it has run, ~0.1 s). *valu_tflops = what
 * the box sustains on plain v_fma_f32; *sclk_mhz_effective = the shader clock that rate implies at one wave64 FMA per two cycles
 * per SIMD. Boxes of one pool run the same binary 3-8 % apart: a bench line carries these so that rounds can be compared. */
int sfgs_box_probe(double* valu_tflops, double* sclk_mhz_effective, void* stream);

/* Blob sizes for N Gaussians, a W x H image, a list-slot capacity D (bins, image, dupgrad) and a per-coarse-bin
 * item capacity (bins). Every 8x8 tile's list starts on a multiple of 64 slots and every 32x32-pixel coarse bin's lists
 * are placed from the bin's tile-hit total (an upper bound, scanned by the plan: no allocator), so a frame with
 * num_duplicates (Gaussian, tile) pairs needs D >= sfgs_raster_slot_capacity(W, H, num_duplicates) = num_duplicates +
 * 1088 * coarse bins (index space only: unused slots are never touched). */
int sfgs_raster_sizes(int32_t N, int32_t W, int32_t H, int64_t D, int64_t coarse_capacity,
                      SfgsRasterSizes* out);
int64_t sfgs_raster_slot_capacity(int32_t W, int32_t H, int64_t num_duplicates);

/* Forward, stage 1 ("plan"): preprocess every Gaussian (cull, EWA projection, 2D mip filter,
 * radius, SH->RGB), write radii[N] (int32) and bin every Gaussian COARSELY: one 16-byte item per
 * (Gaussian, 32x32-pixel coarse bin) holding the mask of the 8x8 tiles it can contribute to. The items
 * are sorted by bin into an exactly sized array inside `bins` (one radix pass; memory in proportion to
 * the items, whatever the frame's skew -- ABI 14); only the items of splats that reach more than 6
 * coarse bins are appended to their bin's slab (coarse_capacity items per bin) with an atomic each.
 * dup_capacity = duplicate indices / list slots. Neither count is known beforehand: if the counters
 * report overflow, call again with a bins blob sized for counters.num_duplicates /
 * counters.max_coarse_bin. Asynchronous on `stream`.
 * counters_pinned_host_128 (optional): 128 bytes of PINNED host memory (hipHostMalloc / torch pin_memory) that the
 * plan's last kernel fills with the frame's counters. Record an event right after this call, enqueue
 * sfgs_raster_forward_render, THEN wait for the event and sfgs_raster_counters_decode the buffer: the capacity
 * check overlaps with the render stage instead of draining the stream (max_tile_list is produced by the render
 * stage and reads 0 there; sfgs_raster_read_counters after the render returns it). */
int sfgs_raster_forward_plan(const SfgsFrame* frame, const SfgsGaussians* g, int32_t* radii,
                             void* geom, size_t geom_bytes, void* tiles, size_t tiles_bytes,
                             void* bins, size_t bins_bytes, int64_t dup_capacity,
                             int64_t coarse_capacity, void* counters_pinned_host_128, void* stream);
int sfgs_raster_counters_decode(const void* host_128, SfgsRasterCounters* out); /* host only, no GPU work */

/* Copies the plan counters to the host. SYNCHRONISES `stream` (the one host sync of the forward,
 * as in the reference, where the duplicate total sizes the sort buffers). */
int sfgs_raster_read_counters(const void* tiles, SfgsRasterCounters* out, void* stream);
/* Same, through a caller-owned PINNED host buffer of >= 64 bytes (no pageable staging copy: a few
 * microseconds less per frame; the library itself owns no host or device memory). */
int sfgs_raster_read_counters_pinned(const void* tiles, void* pinned_host_64, SfgsRasterCounters* out,
                                     void* stream);

/* Forward, stage 2 ("render"): expand the coarse items into per-tile lists (LDS-ranked), sort each
 * list by (depth, Gaussian index), alpha-composite front to back. Outputs: out_color[3,H,W],
 * out_depth[1,H,W], out_alpha[1,H,W]. num_duplicates = counters.num_duplicates of the plan that
 * filled `bins` (same capacities), or -1 when the caller enqueues the render BEFORE reading the
 * counters (no mid-frame host sync): an overflowing plan is memory-safe -- it drops the items that
 * did not fit -- so the caller may read the counters after the render and redo both stages on
 * overflow. `image` may be NULL when no backward will follow. A plan is SINGLE-USE: the render stage consumes
 * header words of the `tiles` blob (long-list count, longest list) that only a new plan resets; rendering the same plan
 * twice is undefined. Asynchronous. */
int sfgs_raster_forward_render(const SfgsFrame* frame, int32_t N, const void* geom, void* tiles,
                               void* bins, size_t bins_bytes, int64_t dup_capacity,
                               int64_t coarse_capacity, int64_t num_duplicates, float* out_color,
                               float* out_depth, float* out_alpha, void* image, size_t image_bytes,
                               void* stream);

/* Backward of the two calls above. dL_dcolor[3,H,W], dL_ddepth[1,H,W], dL_dalpha[1,H,W] may each
 * be NULL (= zeros). Needs the forward's blobs (geom, tiles, bins with the same capacities, image)
 * and radii unchanged. `dupgrad` is scratch: a 48-byte record and (ABI 18) a one-byte "record written" flag for each
 * duplicate INDEX of the plan, plus one line of zeros -- the indices are handed out from 8 disjoint ranges of
 * [0, dup_capacity), so it spans the capacity, not the count (ask sfgs_raster_sizes(N, W, H, dup_capacity, ..).dupgrad_bytes,
 * do not compute it; the gaps are never touched; nothing in it need survive the call). Every gradient tensor in `grads` is
 * fully overwritten. Deterministic (no float atomics). Asynchronous. Writes one word of the `tiles` header: whether this
 * frame's dead list entries (behind their tile's last contributor) are skipped through the flags or receive zero records
 * one by one; decided per frame on the device, sfgs_set_option("prefill", "always" | "never") forces either path for
 * tests: the gradients are bit-identical. */
int sfgs_raster_backward(const SfgsFrame* frame, const SfgsGaussians* g, const int32_t* radii,
                         const void* geom, const void* tiles, const void* bins, int64_t dup_capacity,
                         int64_t coarse_capacity, int64_t num_duplicates, const void* image, const float* dL_dcolor,
                         const float* dL_ddepth, const float* dL_dalpha, void* dupgrad,
                         size_t dupgrad_bytes, const SfgsGaussianGrads* grads, void* stream);

/* ---- One call per stage on ONE scratch allocation (ABI 11) -----------------------------------------------------------
 * The four blobs above as consecutive 256-byte aligned regions of a single caller-owned allocation: the caller asks for
 * the layout once per (N, W, H, capacities) and hands the library one pointer per frame; the library carves it up
 * (still allocates nothing, keeps no state). `with_image` != 0: the frame will be differentiated (image blob included).
 * slot_overhead: sfgs_raster_slot_capacity(W, H, d) == d + slot_overhead (saves the caller a call per frame). */
typedef struct SfgsScratchLayout {
  uint32_t struct_size;
  uint32_t reserved;
  size_t geom_offset, tiles_offset, bins_offset, image_offset;
  size_t total_bytes;      /* size of the allocation (through `bins` without an image, through `image` with one) */
  size_t dupgrad_bytes;    /* separate scratch of the backward (not part of the allocation)                       */
  int64_t coarse_bins;
  int64_t slot_overhead;
} SfgsScratchLayout;
int sfgs_raster_scratch_layout(int32_t N, int32_t W, int32_t H, int64_t dup_capacity, int64_t coarse_capacity,
                               int32_t with_image, SfgsScratchLayout* out);

/* sfgs_raster_forward_plan, hipEventRecord(plan_done_event, stream), sfgs_raster_forward_render(num_duplicates = -1) as
 * ONE call: replaces the autograd forward behind gaussian_renderer/__init__.py:132-140 with a single host -> library
 * transition per frame. plan_done_event: a hipEvent_t (NULL: none); the caller waits on it, decodes
 * counters_pinned_host_128 and redoes the call with larger capacities if the plan overflowed -- the render stage is
 * already running meanwhile. out_* as in sfgs_raster_forward_render. */
int sfgs_raster_forward(const SfgsFrame* frame, const SfgsGaussians* g, int32_t* radii, void* scratch,
                        size_t scratch_bytes, int64_t dup_capacity, int64_t coarse_capacity, int32_t with_image,
                        void* counters_pinned_host_128, void* plan_done_event, float* out_color, float* out_depth,
                        float* out_alpha, void* stream);
/* sfgs_raster_backward on the same allocation (laid out with with_image != 0). */
int sfgs_raster_backward_scratch(const SfgsFrame* frame, const SfgsGaussians* g, const int32_t* radii,
                                 const void* scratch, size_t scratch_bytes, int64_t dup_capacity,
                                 int64_t coarse_capacity, int64_t num_duplicates, const float* dL_dcolor,
                                 const float* dL_ddepth, const float* dL_dalpha, void* dupgrad, size_t dupgrad_bytes,
                                 const SfgsGaussianGrads* grads, void* stream);

/* fused_ssim: mean SSIM (11x11 Gaussian window, sigma 1.5, zero "same" padding, C1=0.01^2,
 * C2=0.03^2 == utils/loss_utils.py:23-63) of img1,img2 [B,C,H,W] float32.
 * forward: writes the mean into *ssim_mean (device float), the per-element map into ssim_map when
 * non-NULL and, when with_grad != 0, the three partial-derivative maps the backward consumes into
 * `scratch` (sfgs_ssim_scratch_bytes). The mean is reduced in a fixed order (bit-reproducible run to run).
 * Numerics: value and gradient are TOLERANCE-equal to the reference formula (utils/loss_utils.py:33-63; tests: 2e-6 on the
 * mean, 5e-5 relative on the gradient), not bit-equal to it nor across ABI 15 -> 16: since ABI 16 the derivative maps use
 * 1/B1 = B2 * inv and 1/B2 = B1 * inv with one reciprocal of B1 * B2 where the reference divides twice, and the 32 x 22
 * tiling sums the mean's partials in a different (fixed) order.
 * backward: dL_dimg1 = *dL_dmean (device float) * d(mean ssim)/d(img1); img2 gets no gradient
 * (it is the ground truth, train.py:222).
 * Limits (SFGS_E_UNSUPPORTED): H * W < 2^30 pixels per plane, at most 2^31 - 1 tiles of 32 x 22 pixels in all. */
size_t sfgs_ssim_scratch_bytes(int32_t B, int32_t C, int32_t H, int32_t W, int32_t with_grad);
int sfgs_ssim_forward(const float* img1, const float* img2, int32_t B, int32_t C, int32_t H,
                      int32_t W, float* ssim_map_or_null, float* ssim_mean, void* scratch,
                      size_t scratch_bytes, int32_t with_grad, void* stream);
int sfgs_ssim_backward(const float* img1, const float* img2, int32_t B, int32_t C, int32_t H,
                       int32_t W, const void* scratch, const float* dL_dmean, float* dL_dimg1,
                       void* stream);

/* Joint render with the Gaussians sharded over ranks (SURVEY 8e "all-gather the preprocessed 2D records"; the reference
 * has no multi-scene render). Every rank runs sfgs_raster_forward_plan on ITS Gaussians for the whole frame, then:
 *   sfgs_raster_plan_export  copies the plan's per-Gaussian compositing records rec_out[N][12 floats], the per-coarse-bin
 *                            item counts count_out[coarse_bins] and the items items_out[coarse_bins][export_capacity]
 *                            (16 bytes each) into caller-owned device buffers with FIXED strides, ready for an
 *                            all-gather (export_capacity >= the fullest coarse bin of any rank: all-reduce MAX of
 *                            SfgsRasterCounters.max_coarse_bin);
 *   sfgs_raster_plan_merge   builds ONE plan (geom / tiles / bins blobs sized by sfgs_raster_sizes for the total N, the
 *                            summed duplicate count and coarse_capacity >= parts x export_capacity) from the gathered
 *                            parts; Gaussian ids become positions in the concatenation (part order), so the per-tile
 *                            order -- ascending (depth bits, id) -- is the single-process one and
 *                            sfgs_raster_forward_render (num_duplicates = -1, any tile-row band) produces the same pixels
 *                            bit for bit. Forward only: a merged plan carries no duplicate indices for a backward.
 * Up to 16 parts. Asynchronous on `stream`. `bins` of sfgs_raster_plan_export is the blob the plan was given, in full
 * (sfgs_raster_sizes().bins_bytes: the bin-sorted items it reads lie at the blob's end; the call takes no size to check). */
int sfgs_raster_plan_export(const SfgsFrame* frame, int32_t N, const void* geom, const void* tiles, const void* bins,
                            int64_t dup_capacity, int64_t coarse_capacity, int64_t export_capacity, float* rec_out,
                            uint32_t* count_out, void* items_out, void* stream);
int sfgs_raster_plan_merge(const SfgsFrame* frame, int32_t parts, const int32_t* part_N, const float* const* part_rec,
                           const uint32_t* const* part_count, const void* const* part_items, int64_t export_capacity,
                           void* geom, size_t geom_bytes, void* tiles, size_t tiles_bytes, void* bins, size_t bins_bytes,
                           int64_t dup_capacity, int64_t coarse_capacity, void* stream);

/* simple_knn distCUDA2: out[i] = mean squared distance from point i to its 3 nearest other
 * points (index-excluded; fewer than 3 others: mean over those that exist). Exact for every input (brute force up to
 * 32 768 points; above, a Z-curve counting sort with box-pruned search -- the layout of the reference's simple-knn).
 * scratch: sfgs_knn_scratch_bytes(N) bytes, 256-byte aligned (0 / NULL for small N). Non-finite points get 0 and are
 * nobody's neighbour. */
size_t sfgs_knn_scratch_bytes(int32_t N);
int sfgs_knn_dist2(const float* xyz, int32_t N, float* out, void* scratch, size_t scratch_bytes,
                   void* stream);

/* Fused per-Gaussian pre-pass of render(): activations + Mip-Splatting 3D filter, i.e.
 *   scales    = sqrt(exp(scaling_raw)^2 + filter3d^2)        GaussianModel.get_scaling_with_3D_filter  scene/gaussian_model.py:207-213
 *   opacities = sigmoid(opacity_raw) * sqrt(prod s^2 / prod (s^2 + filter3d^2))  get_opacity_with_3D_filter  :237-249
 *   rotations = normalize(rotation_raw)                       get_rotation                               :216-217
 * filter3d is [N,1] float64 (training) or float32 (after load_ply); opacity_raw is float32 until the reference's first
 * reset_opacity (scene/gaussian_model.py:483-501), float64 afterwards (the reset divides by a float64 coefficient).
 * f64_mask: bit 0 = filter3d is float64, bit 1 = opacity_raw (and, in the backward, g_opacity_raw) is float64; torch's
 * type promotion is followed in every combination. Outputs float32.
 * backward: gradients w.r.t. the three raw tensors given gradients of the three outputs (each may be NULL = 0). */
int sfgs_prepass_forward(int32_t N, const float* scaling_raw, const void* opacity_raw,
                         const float* rotation_raw, const void* filter3d, int32_t f64_mask,
                         float* scales, float* opacities, float* rotations, void* stream);
int sfgs_prepass_backward(int32_t N, const float* scaling_raw, const void* opacity_raw,
                          const float* rotation_raw, const void* filter3d, int32_t f64_mask,
                          const float* g_scales, const float* g_opacities, const float* g_rotations,
                          float* g_scaling_raw, void* g_opacity_raw, float* g_rotation_raw, void* stream);

/* GaussianModel.compute_3D_filter (scene/gaussian_model.py:255-308): filter_out[N] (float64) =
 * (smallest camera-space depth at which any camera sees the point with a 15 % screen margin) / max_focal *
 * sqrt(0.2); points no camera sees get the largest seen depth. cams: device [C][18] float64 =
 * R[9] (row-major, used as xyz @ R like the reference), T[3], focal_x, focal_y, cx_ori, cy_ori, width, height. */
size_t sfgs_filter3d_scratch_bytes(int32_t N);
int sfgs_filter3d(const float* xyz, int32_t N, const double* cams, int32_t C, double max_focal,
                  double* filter_out, void* scratch, size_t scratch_bytes, void* stream);

/* GaussianModel.add_densification_stats (scene/gaussian_model.py:744-749), in place, no host
 * sync: for every i with update_filter[i] != 0 (bool/uint8 [N]):
 *   accum[i] += hypot(grad[i,0], grad[i,1]); accum_abs[i] += |grad[i,2]|; accum_abs_max[i] = max(., |grad[i,2]|);
 *   denom[i] += 1.   viewspace_grad is means2D.grad [N,3]; accum_abs_max may be NULL. */
int sfgs_densify_stats(int32_t N, const float* viewspace_grad, const unsigned char* update_filter,
                       float* xyz_gradient_accum, float* xyz_gradient_accum_abs,
                       float* xyz_gradient_accum_abs_max_or_null, float* denom, void* stream);

/* One Adam step over a list of float32 tensors in one launch. Replaces the per-iteration
 * `gaussians.optimizer.step()` (train.py:339,906) of `torch.optim.Adam(l, lr=0.0, eps=1e-15)`
 * (scene/gaussian_model.py:382); arithmetic = torch's default ("foreach") CUDA path, operation by operation:
 *   g' = grad + weight_decay*param (if weight_decay != 0);  m += (1-b1)*(g'-m);  v = v*b2 + (1-b2)*g'*g';
 *   param += neg_step_size * (m / (sqrt(v)/bias_correction2_sqrt + eps))
 * The host computes, in double like torch does, neg_step_size = -lr/(1-b1^t) and bias_correction2_sqrt =
 * sqrt(1-b2^t) for the tensor's own step count t (already incremented). `tensors` is a HOST array; all pointers are
 * device pointers to contiguous float32 (or, with SFGS_ADAM_F64, float64) storage of `count` elements, updated in place. */
#define SFGS_ADAM_F64 1u   /* flags: param, grad and both moments are float64 (the reference's `_opacity` group after
                              reset_opacity, scene/gaussian_model.py:483-501); the update then runs in float64 like torch's */
typedef struct SfgsAdamTensor {
  void* param;
  const void* grad;
  void* exp_avg;
  void* exp_avg_sq;
  int64_t count;
  double neg_step_size;           /* the scalars as torch forms them (Python floats = double); the float32 path */
  double one_minus_beta1;         /* rounds them to float, the float64 path uses them as they are             */
  double beta2;
  double one_minus_beta2;
  double bias_correction2_sqrt;
  double eps;
  double weight_decay;
  uint32_t flags;                 /* SFGS_ADAM_* */
  uint32_t reserved;
} SfgsAdamTensor;
int sfgs_adam_step(const SfgsAdamTensor* tensors, int32_t count, void* stream);

/* utils/sh_utils.py:57-112 eval_sh: out[n,c] = sum_{k < (deg+1)^2} basis_k(dirs[n]) * sh[n,c,k], the reference's
 * channel-major layout sh[N,3,K] with K >= (deg+1)^2 stored coefficients (used by render() at
 * gaussian_renderer/__init__.py:115,124). No normalisation of dirs, no +0.5, no clamp -- like the reference function.
 * backward: g_sh[N,3,K] fully written (zeros beyond the active degree), g_dirs[N,3] optional. */
int sfgs_sh_eval_forward(int32_t N, int32_t deg, int32_t K, const float* sh, const float* dirs, float* out,
                         void* stream);
int sfgs_sh_eval_backward(int32_t N, int32_t deg, int32_t K, const float* sh, const float* dirs, const float* g_out,
                          float* g_sh, float* g_dirs_or_null, void* stream);

/* Row compaction of many tensors by ONE keep-mask (replaces the 26 `tensor[mask]` operations of
 * GaussianModel.prune_points/_prune_optimizer, scene/gaussian_model.py:563-603). keep[N]: bool/uint8, nonzero = the
 * row survives. sfgs_compact_plan scans the mask into caller-owned scratch and, if kept_out != NULL, synchronises the
 * stream once to report the number of surviving rows (the caller allocates dst tensors of that many rows).
 * sfgs_compact_rows then copies row i of every src to row rank(i) of its dst in one launch; src/dst are device
 * pointers to contiguous [N, row_bytes] / [kept, row_bytes] storage, `tensors` is a HOST array. */
typedef struct SfgsCompactTensor {
  const void* src;
  void* dst;
  int64_t row_bytes;
} SfgsCompactTensor;
size_t sfgs_compact_scratch_bytes(int64_t N);
int sfgs_compact_plan(const unsigned char* keep, int64_t N, void* scratch, size_t scratch_bytes, int64_t* kept_out,
                      void* stream);
int sfgs_compact_rows(const unsigned char* keep, int64_t N, const void* scratch, const SfgsCompactTensor* tensors,
                      int32_t count, void* stream);

/* GaussianModel.densify_and_prune (scene/gaussian_model.py:603-742) as a handful of launches.
 *
 * sfgs_select_kth: exact order statistics of N non-negative floats by radix select, for ANY N (torch.quantile refuses
 * more than 16 M elements and the reference then falls back to Q = 0.99, :716-723). rank_dev: DEVICE float r (as
 * torch.quantile forms it: q * (N - 1)); out2[0] = value at rank floor(r), out2[1] = value at rank ceil(r) (device);
 * the caller interpolates like torch (lerp). Asynchronous.
 *
 * sfgs_densify_decide: per Gaussian, from grad_norm[N] = |xyz_gradient_accum / denom|, grad_abs[N] (the abs column),
 * the ACTIVATED scaling[N,3] / opacity[N] (float32, or float64 after the reference's reset_opacity) and the thresholds:
 * clone = selected and max(scaling) <= dense_threshold, split = selected and max(scaling) > dense_threshold with
 * selected = grad_norm >= max_grad or grad_abs >= Q (Q read from Q_dev when non-NULL, else Q_host), and which rows
 * survive the final prune (opacity < min_opacity, or -- when use_big_threshold -- max(scaling) > big_threshold; the
 * screen-size term never fires in the reference because densification_postfix resets max_radii2D first). Fills
 * `scratch` (sfgs_densify_scratch_bytes) and returns, with ONE host synchronisation, totals_out[5] = rows kept of
 * {originals, clones, children per child index} and the raw {clone, split} counts.
 * sfgs_densify_gather: writes every dst tensor [totals[0] + totals[1] + 2 * totals[2] rows] in the reference's final
 * order [surviving originals | clones | first children | second children]; zero_new_rows: clones / children get zeros
 * (Adam moments). sfgs_densify_children then overwrites the child rows of xyz (parent + R(q) * sample) and of the raw
 * scaling (log(scaling / 1.6)); samples[2 * totals[4], 3] = std * z in the layout of the reference's `samples` (:666),
 * or -- unit_samples != 0 -- just z ~ N(0, 1), which the kernel multiplies by the parent's scaling itself (no mask
 * gather on the host side).
 * sfgs_densify_masks: the decisions as byte masks (tests / diagnostics). */
typedef struct SfgsDensifyTensor {
  const void* src;
  void* dst;
  int64_t row_bytes;       /* multiple of 4 */
  int32_t zero_new_rows;
  int32_t reserved;
} SfgsDensifyTensor;
size_t sfgs_select_scratch_bytes(void);
int sfgs_select_kth(const float* values, int64_t N, const float* rank_dev, float* out2, void* scratch,
                    size_t scratch_bytes, void* stream);
size_t sfgs_densify_scratch_bytes(int64_t N);
int sfgs_densify_decide(int64_t N, const float* grad_norm, const float* grad_abs, const float* scaling,
                        const void* opacity, int32_t opacity_is_f64, const float* Q_dev, float Q_host, float max_grad,
                        double min_opacity, float dense_threshold, float big_threshold, int32_t use_big_threshold,
                        void* scratch, size_t scratch_bytes, int64_t totals_out[5], void* stream);
int sfgs_densify_masks(int64_t N, const void* scratch, unsigned char* clone_out, unsigned char* split_out,
                       unsigned char* keep_out, void* stream);
int sfgs_densify_gather(int64_t N, const void* scratch, const int64_t totals[5], const SfgsDensifyTensor* tensors,
                        int32_t count, void* stream);
int sfgs_densify_children(int64_t N, const void* scratch, const int64_t totals[5], const float* xyz,
                          const float* rotation_raw, const float* scaling, const float* samples, int32_t unit_samples,
                          float* xyz_out, float* scaling_raw_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SFGS_H */
