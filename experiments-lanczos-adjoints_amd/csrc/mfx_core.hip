// libmfx: error reporting, versioning and the hipEvent-based per-class kernel timer.
#include <stdarg.h>
#include <stdlib.h>

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

// ------------------------------------------------------------------------------------------------
// fills / copies as kernels (see mfx_internal.h)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_zero_words(uint32_t* __restrict__ p, size_t nwords) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < nwords) p[i] = 0u;
}
__global__ __launch_bounds__(256) void k_copy_rows_words(uint32_t* __restrict__ dst, size_t dst_pitch_w,
                                                         const uint32_t* __restrict__ src, size_t src_pitch_w, size_t width_w) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x, r = blockIdx.y;
  if (i < width_w) dst[r * dst_pitch_w + i] = src[r * src_pitch_w + i];
}

int zero_async(void* dst, size_t bytes, hipStream_t stream) {
  if (bytes == 0) return MFX_OK;
  MFX_REQUIRE(bytes % 4 == 0 && reinterpret_cast<uintptr_t>(dst) % 4 == 0, MFX_ERR_INVALID, "zero_async: 4-byte granularity");
  const size_t nw = bytes / 4;
  k_zero_words<<<(unsigned)((nw + 255) / 256), 256, 0, stream>>>(static_cast<uint32_t*>(dst), nw);
  MFX_CHECK_LAUNCH();
  return MFX_OK;
}

int copy_rows_async(void* dst, size_t dst_pitch, const void* src, size_t src_pitch, size_t width, size_t rows,
                    hipStream_t stream) {
  if (width == 0 || rows == 0) return MFX_OK;
  MFX_REQUIRE(width % 4 == 0 && dst_pitch % 4 == 0 && src_pitch % 4 == 0 && rows <= 65535, MFX_ERR_INVALID,
              "copy_rows_async: 4-byte granularity, at most 65535 rows");
  const size_t ww = width / 4;
  k_copy_rows_words<<<dim3((unsigned)((ww + 255) / 256), (unsigned)rows), 256, 0, stream>>>(
      static_cast<uint32_t*>(dst), dst_pitch / 4, static_cast<const uint32_t*>(src), src_pitch / 4, ww);
  MFX_CHECK_LAUNCH();
  return MFX_OK;
}

// ------------------------------------------------------------------------------------------------
// hipGraph cache (see mfx_internal.h)
// ------------------------------------------------------------------------------------------------
struct GraphEntry {
  std::string key;
  hipGraphExec_t exec;  // null until the second call with this key
  bool failed;          // capture or instantiation failed once: always eager
  uint64_t last_use;
  hipEvent_t done;      // recorded behind every replay: an exec is destroyed only after its last launch has finished
};
static void destroy_entry(GraphEntry& g) {
  if (g.exec) {
    if (g.done) (void)hipEventSynchronize(g.done);  // the host may be many replays ahead of the GPU
    (void)hipGraphExecDestroy(g.exec);
  }
  if (g.done) (void)hipEventDestroy(g.done);
  g.exec = nullptr;
  g.done = nullptr;
}
static std::mutex g_gmutex;
static std::vector<GraphEntry> g_graphs;
static uint64_t g_gclock = 0;
static int64_t g_captured = 0, g_replayed = 0;
// captures run on a private stream of the current device: the caller's stream may be the legacy default stream
static hipStream_t g_capture_stream[16] = {};
constexpr size_t kMaxGraphs = 32;

static bool graphs_enabled() {
  static const bool on = [] { const char* e = getenv("MFX_GRAPHS"); return !e || atoi(e) != 0; }();
  return on;
}

int run_graphed(bool eligible, const GraphKey& key, hipStream_t stream, const std::function<int(hipStream_t)>& fn) {
  if (!eligible || g_timing || !graphs_enabled()) return fn(stream);
  // A graph belongs to the device its kernels were captured for, and the capture stream is created on the CURRENT device: replay only
  // when the caller's stream lives on the current device (a caller that passes a stream of another device without making that
  // device current gets eager launches, which HIP routes by the stream).  The Python wrappers enter torch.cuda.device(dev).
  int device = -1, stream_device = -1;
  if (hipGetDevice(&device) != hipSuccess || device < 0 || device >= 16) return fn(stream);
  if (stream != nullptr && (hipStreamGetDevice(stream, &stream_device) != hipSuccess || stream_device != device)) {
    (void)hipGetLastError();
    return fn(stream);
  }
  std::string full = key.bytes;
  full.append(reinterpret_cast<const char*>(&device), sizeof(device));
  std::lock_guard<std::mutex> lock(g_gmutex);
  GraphEntry* e = nullptr;
  for (auto& g : g_graphs)
    if (g.key == full) e = &g;
  if (!e) {  // first sighting: run eagerly, remember the key
    if (g_graphs.size() >= kMaxGraphs) {
      size_t old = 0;
      for (size_t i = 1; i < g_graphs.size(); ++i)
        if (g_graphs[i].last_use < g_graphs[old].last_use) old = i;
      destroy_entry(g_graphs[old]);
      g_graphs.erase(g_graphs.begin() + (long)old);
    }
    g_graphs.push_back(GraphEntry{full, nullptr, false, ++g_gclock, nullptr});
    return fn(stream);
  }
  e->last_use = ++g_gclock;
  if (e->failed) return fn(stream);
  if (!e->exec) {
    hipStream_t& cap = g_capture_stream[device];
    if (!cap && hipStreamCreateWithFlags(&cap, hipStreamNonBlocking) != hipSuccess) {
      e->failed = true;
      (void)hipGetLastError();
      return fn(stream);
    }
    if (hipStreamBeginCapture(cap, hipStreamCaptureModeRelaxed) != hipSuccess) {
      e->failed = true;
      (void)hipGetLastError();
      return fn(stream);
    }
    const int rc = fn(cap);
    hipGraph_t graph = nullptr;
    const hipError_t ec = hipStreamEndCapture(cap, &graph);
    if (rc != MFX_OK) {  // the driver refused its arguments: nothing ran, nothing to replay
      if (graph) (void)hipGraphDestroy(graph);
      (void)hipGetLastError();
      e->failed = true;
      return rc;
    }
    hipGraphExec_t exec = nullptr;
    if (ec != hipSuccess || !graph || hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess) {
      if (graph) (void)hipGraphDestroy(graph);
      (void)hipGetLastError();
      e->failed = true;
      return fn(stream);
    }
    (void)hipGraphDestroy(graph);
    e->exec = exec;
    if (hipEventCreateWithFlags(&e->done, hipEventDisableTiming) != hipSuccess) {
      e->done = nullptr;
      (void)hipGetLastError();
    }
    ++g_captured;
  }
  MFX_CHECK_HIP(hipGraphLaunch(e->exec, stream));
  if (e->done) (void)hipEventRecord(e->done, stream);
  ++g_replayed;
  return MFX_OK;
}

}  // namespace mfx

extern "C" {

const char* mfx_last_error(void) { return mfx::g_err; }
int mfx_version(void) { return MFX_VERSION; }

int mfx_timing_enable(int enable) {
  mfx::g_timing = enable != 0;
  return MFX_OK;
}

int mfx_graph_stats(int64_t* captured, int64_t* replayed) {
  std::lock_guard<std::mutex> lock(mfx::g_gmutex);
  if (captured) *captured = mfx::g_captured;
  if (replayed) *replayed = mfx::g_replayed;
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
