// libmfx: error reporting, versioning and the hipEvent-based per-class kernel timer.
#include <stdarg.h>

#include <mutex>
#include <vector>

#include "mfx_internal.h"

namespace mfx {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

struct TimedSpan {
  int cls;
  hipEvent_t start, stop;
};
static std::mutex g_tmutex;
static std::vector<TimedSpan> g_spans;
static bool g_timing = false;

ScopedTimer::ScopedTimer(int c, hipStream_t s) : cls(c), stream(s), slot(nullptr) {
  if (!g_timing) return;
  TimedSpan* sp = new TimedSpan{c, nullptr, nullptr};
  if (hipEventCreate(&sp->start) != hipSuccess || hipEventCreate(&sp->stop) != hipSuccess) {
    delete sp;
    return;
  }
  (void)hipEventRecord(sp->start, stream);
  slot = sp;
}

ScopedTimer::~ScopedTimer() {
  if (!slot) return;
  TimedSpan* sp = static_cast<TimedSpan*>(slot);
  (void)hipEventRecord(sp->stop, stream);
  std::lock_guard<std::mutex> lock(g_tmutex);
  g_spans.push_back(*sp);
  delete sp;
}

}  // namespace mfx

extern "C" {

const char* mfx_last_error(void) { return mfx::g_err; }
int mfx_version(void) { return MFX_VERSION; }

int mfx_timing_enable(int enable) {
  mfx::g_timing = enable != 0;
  return MFX_OK;
}

int mfx_timing_reset(void) {
  std::lock_guard<std::mutex> lock(mfx::g_tmutex);
  for (auto& sp : mfx::g_spans) {
    (void)hipEventDestroy(sp.start);
    (void)hipEventDestroy(sp.stop);
  }
  mfx::g_spans.clear();
  return MFX_OK;
}

int mfx_timing_read(int cls, double* total_ms, int64_t* launches) {
  std::lock_guard<std::mutex> lock(mfx::g_tmutex);
  double tot = 0.0;
  int64_t cnt = 0;
  for (auto& sp : mfx::g_spans) {
    if (sp.cls != cls) continue;
    MFX_CHECK_HIP(hipEventSynchronize(sp.stop));
    float ms = 0.f;
    MFX_CHECK_HIP(hipEventElapsedTime(&ms, sp.start, sp.stop));
    tot += ms;
    ++cnt;
  }
  if (total_ms) *total_ms = tot;
  if (launches) *launches = cnt;
  return MFX_OK;
}

}  // extern "C"
