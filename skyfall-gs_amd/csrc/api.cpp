// api.cpp -- error plumbing, version entry points and the optional per-kernel profiler of libsfgs.so
// (include/sfgs.h).
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>

#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <mutex>
#include <vector>

#include "../../include/sfgs.h"
#include "sfgs_internal.h"

namespace sfgs {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// ---- route options (sfgs_set_option) ----------------------------------------------------------------
struct OptSpec { const char* key; const char* env; const char* values[6]; };
static const OptSpec kOpts[OPT_COUNT] = {
    {"sort", "SFGS_SORT", {"auto", "fused", "fused1024", "split", "fused768", nullptr}},
    {"plan_scan", "SFGS_PLAN_SCAN", {"fused", "separate", nullptr}},
    {"binning", "SFGS_BINNING", {"auto", "direct", nullptr}},
    {"prefill", "SFGS_PREFILL", {"auto", "always", "never", nullptr}},
    {"knn", "SFGS_KNN", {"auto", "brute", nullptr}},
    {"tile_order", "SFGS_TILE_ORDER", {"auto", "always", "never", nullptr}}};
static std::atomic<int> g_opt[OPT_COUNT];
static int opt_value(int which, const char* v) {
  for (int i = 0; v && kOpts[which].values[i]; ++i)
    if (!strcmp(v, kOpts[which].values[i])) return i;
  return -1;
}
// the environment is read once, at load time (a value the option does not know leaves the default)
static const bool g_opt_init = [] {
  for (int k = 0; k < OPT_COUNT; ++k) {
    const int v = opt_value(k, getenv(kOpts[k].env));
    g_opt[k].store(v < 0 ? 0 : v, std::memory_order_relaxed);
  }
  return true;
}();
int option(int which) { return g_opt[which].load(std::memory_order_relaxed); }

// ---- profiler: HIP events recorded on the launch stream around every kernel ---------------------
static const char* const kKernelNames[KID_COUNT] = {
    "subpix_bound", "preprocess", "bin_count", "bin_rank", "bin_scatter", "plan_scan", "fine_bin", "sort_tiles_small", "sort_tiles_reg_long",
    "sort_tiles_lds", "composite_fwd", "composite_bwd", "preprocess_bwd", "ssim_fwd", "ssim_mean", "ssim_bwd",
    "knn_dist2", "prepass_fwd", "prepass_bwd", "filter3d", "densify_stats", "adam", "sh_eval_fwd", "sh_eval_bwd", "compact_scan", "compact_gather", "densify"};

struct ProfRec { int id; hipEvent_t a, b; };
static std::mutex g_prof_mu;
static bool g_prof_on = false;
static unsigned long long g_prof_mask = ~0ull;
static std::vector<ProfRec> g_prof_recs;
static std::vector<hipEvent_t> g_prof_pool;

bool prof_enabled() { return g_prof_on; }
bool prof_selected(int id) { return g_prof_on && ((g_prof_mask >> id) & 1ull); }

static hipEvent_t prof_get_event() {
  if (!g_prof_pool.empty()) { hipEvent_t e = g_prof_pool.back(); g_prof_pool.pop_back(); return e; }
  hipEvent_t e = nullptr;
  if (hipEventCreate(&e) != hipSuccess) return nullptr;
  return e;
}

void* prof_begin(int id, hipStream_t stream) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  if (!g_prof_on) return nullptr;
  ProfRec r{id, prof_get_event(), prof_get_event()};
  if (!r.a || !r.b) return nullptr;
  (void)hipEventRecord(r.a, stream);
  g_prof_recs.push_back(r);
  return (void*)(uintptr_t)g_prof_recs.size();  // index + 1
}

void prof_end(void* token, hipStream_t stream) {
  if (!token) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  const size_t idx = (size_t)(uintptr_t)token - 1;
  if (idx < g_prof_recs.size()) (void)hipEventRecord(g_prof_recs[idx].b, stream);
}
}  // namespace sfgs

using namespace sfgs;

extern "C" int sfgs_abi_version(void) { return SFGS_ABI_VERSION; }
extern "C" const char* sfgs_last_error(void) { return sfgs::g_err; }

extern "C" int sfgs_set_option(const char* key, const char* value) {
  SFGS_REQUIRE(key && value, SFGS_E_ARG, "sfgs_set_option: NULL key / value");
  for (int k = 0; k < OPT_COUNT; ++k)
    if (!strcmp(key, kOpts[k].key)) {
      const int v = opt_value(k, value);
      SFGS_REQUIRE(v >= 0, SFGS_E_ARG, "sfgs_set_option: option \"%s\" has no value \"%s\"", key, value);
      g_opt[k].store(v, std::memory_order_relaxed);
      return SFGS_OK;
    }
  SFGS_REQUIRE(false, SFGS_E_ARG, "sfgs_set_option: unknown option \"%s\"", key);
  return SFGS_E_ARG;
}
extern "C" const char* sfgs_get_option(const char* key) {
  for (int k = 0; key && k < OPT_COUNT; ++k)
    if (!strcmp(key, kOpts[k].key)) return kOpts[k].values[option(k)];
  return nullptr;
}

extern "C" int sfgs_profile_kernel_count(void) { return KID_COUNT; }
extern "C" const char* sfgs_profile_kernel_name(int32_t id) {
  return (id >= 0 && id < KID_COUNT) ? kKernelNames[id] : "";
}

extern "C" int sfgs_profile_enable(int32_t on) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  for (auto& r : g_prof_recs) { g_prof_pool.push_back(r.a); g_prof_pool.push_back(r.b); }
  g_prof_recs.clear();
  g_prof_on = on != 0;
  return SFGS_OK;
}

extern "C" int sfgs_profile_select(uint64_t kernel_mask) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  g_prof_mask = kernel_mask;
  return SFGS_OK;
}

extern "C" int sfgs_profile_collect(double* ms_sum, int64_t* launches, int32_t n) {
  SFGS_REQUIRE(ms_sum && launches && n >= KID_COUNT, SFGS_E_ARG, "profile arrays must hold %d entries", KID_COUNT);
  std::lock_guard<std::mutex> lk(g_prof_mu);
  for (int i = 0; i < n; ++i) { ms_sum[i] = 0.0; launches[i] = 0; }
  for (auto& r : g_prof_recs) {
    SFGS_CHECK_HIP(hipEventSynchronize(r.b));
    float ms = 0.f;
    SFGS_CHECK_HIP(hipEventElapsedTime(&ms, r.a, r.b));
    ms_sum[r.id] += (double)ms;
    launches[r.id] += 1;
  }
  for (auto& r : g_prof_recs) { g_prof_pool.push_back(r.a); g_prof_pool.push_back(r.b); }
  g_prof_recs.clear();
  return SFGS_OK;
}
