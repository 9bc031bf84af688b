// common.hip — C-ABI plumbing shared by every entry point: version + thread-local error string.
#include <stdarg.h>
#include <atomic>
#include <mutex>
#include <string.h>

#include "gt_common.h"

static thread_local char g_err[512] = "";

void gt_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int gt_version(void) { return 100; /* 0.1.0 */ }
extern "C" const char* gt_last_error(void) { return g_err; }

// ---- opt-in launch profiler ----------------------------------------------------------------------
// HIP events recorded on the launch stream around selected entry points (bench.py's roofline
// object needs per-kernel durations measured inside the timed region, also when the kernels are
// enqueued by the composite layer calls).  Off by default; guarded by a mutex; never synchronises.
#include <mutex>
#include <vector>

namespace {
struct ProfRecord {
  const char* name;
  hipEvent_t e0, e1;
  int64_t dims[6];
};
std::mutex g_prof_mu;
std::vector<ProfRecord> g_prof;
std::vector<hipEvent_t> g_pool;  // events are created when profiling is switched on, not per launch
size_t g_pool_next = 0;
unsigned g_prof_mask = 0;
constexpr size_t POOL_EVENTS = 16384;
}  // namespace

unsigned gt_prof_mask() { return g_prof_mask; }

int64_t gt_prof_begin(const char* name, hipStream_t stream, const int64_t* dims, int ndims) {
  ProfRecord r{};
  r.name = name;
  for (int i = 0; i < 6; ++i) r.dims[i] = i < ndims ? dims[i] : 0;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  if (g_pool_next + 2 > g_pool.size()) return -1;  // pool exhausted: stop recording, never allocate here
  r.e0 = g_pool[g_pool_next++];
  r.e1 = g_pool[g_pool_next++];
  (void)hipEventRecord(r.e0, stream);
  g_prof.push_back(r);
  return (int64_t)g_prof.size() - 1;
}

void gt_prof_end(int64_t id, hipStream_t stream) {
  if (id < 0) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  if ((size_t)id < g_prof.size()) (void)hipEventRecord(g_prof[id].e1, stream);
}

// ---- named runtime options ----------------------------------------------------------------------------------------------------
namespace {
std::atomic<int> g_options[GT_OPT_COUNT];   // zero-initialised = every default
const char* const OPTION_NAMES[GT_OPT_COUNT] = {"attn_f32_exact", "bnstats_rows_kernel", "lin_ring"};
int option_id(const char* name) {
  if (name)
    for (int i = 0; i < GT_OPT_COUNT; ++i)
      if (strcmp(name, OPTION_NAMES[i]) == 0) return i;
  return -1;
}
}  // namespace
int gt_opt(int id) { return g_options[id].load(std::memory_order_relaxed); }
extern "C" int gt_option_set(const char* name, int value) {
  const int id = option_id(name);
  if (id < 0) { gt_set_error("gt_option_set: unknown option '%s'", name ? name : "(null)"); return GT_ERR_INVALID_ARG; }
  return g_options[id].exchange(value < 0 ? 0 : value);
}
extern "C" int gt_option_get(const char* name) {
  const int id = option_id(name);
  if (id < 0) { gt_set_error("gt_option_get: unknown option '%s'", name ? name : "(null)"); return GT_ERR_INVALID_ARG; }
  return gt_opt(id);
}

extern "C" int gt_profile_enable(unsigned mask) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  if (mask) {
    g_prof.clear();
    g_pool_next = 0;
    while (g_pool.size() < POOL_EVENTS) {
      hipEvent_t e;
      if (hipEventCreate(&e) != hipSuccess) break;
      g_pool.push_back(e);
    }
  }
  g_prof_mask = mask;
  return GT_OK;
}

extern "C" int gt_profile_resume(unsigned mask) {  // toggle recording without clearing the records
  std::lock_guard<std::mutex> lk(g_prof_mu);
  g_prof_mask = mask;
  return GT_OK;
}

extern "C" int64_t gt_profile_count(void) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  return (int64_t)g_prof.size();
}

extern "C" int gt_profile_get(int64_t i, char* name_out, int64_t name_cap, float* ms, int64_t* dims6) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  if (i < 0 || (size_t)i >= g_prof.size() || !name_out || !ms || !dims6) return GT_ERR_INVALID_ARG;
  const ProfRecord& r = g_prof[i];
  snprintf(name_out, (size_t)name_cap, "%s", r.name);
  if (hipEventElapsedTime(ms, r.e0, r.e1) != hipSuccess) return GT_ERR_LAUNCH;  // caller synchronises first
  for (int k = 0; k < 6; ++k) dims6[k] = r.dims[k];
  return GT_OK;
}


// ---- side streams of the fused path: created here so that they can carry a HIP priority ---------------------------
// level: -1 = the highest priority the device offers, 0 = default, +1 = the lowest.  Returns the stream or NULL.
extern "C" void* gt_stream_create(int level) {
  int least = 0, greatest = 0;   // numerically: greatest priority <= least priority
  (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
  const int prio = level < 0 ? greatest : (level > 0 ? least : 0);
  hipStream_t st = nullptr;
  if (hipStreamCreateWithPriority(&st, hipStreamNonBlocking, prio) != hipSuccess) return nullptr;
  return st;
}
extern "C" void gt_stream_destroy(void* stream) {
  if (stream) (void)hipStreamDestroy((hipStream_t)stream);
}
extern "C" int gt_stream_priority_range(int* least, int* greatest) {
  return hipDeviceGetStreamPriorityRange(least, greatest) == hipSuccess ? GT_OK : GT_ERR_LAUNCH;
}

// ---- events for cross-stream dependencies between entry points -------------------------------------
extern "C" void* gt_event_create(void) {
  hipEvent_t ev = nullptr;
  if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) return nullptr;
  return ev;
}
extern "C" void gt_event_destroy(void* event) {
  if (event) (void)hipEventDestroy((hipEvent_t)event);
}
extern "C" int gt_event_record(void* event, gt_stream_t stream) {
  GT_CHECK_ARG(event, "null event");
  if (hipEventRecord((hipEvent_t)event, (hipStream_t)stream) != hipSuccess) { gt_set_error("gt_event_record failed"); return GT_ERR_LAUNCH; }
  return GT_OK;
}
extern "C" int gt_stream_wait_event(gt_stream_t stream, void* event) {
  GT_CHECK_ARG(event, "null event");
  if (hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)event, 0) != hipSuccess) { gt_set_error("gt_stream_wait_event failed"); return GT_ERR_LAUNCH; }
  return GT_OK;
}
