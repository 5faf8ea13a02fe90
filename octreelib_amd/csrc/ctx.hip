// Context, error plumbing, device buffers, kernel timers, small device utilities.
#include <chrono>
#include <cstdarg>
#include <cstdlib>

#include <algorithm>

#include "common.h"

int octl_set_error(octl_ctx* ctx, int code, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  if (ctx) ctx->err = buf;
  return code;
}

std::atomic<uint64_t> g_octl_host_syncs{0};

namespace {
struct OptName { const char* name; int64_t OctlOptions::*field; };
const OptName kOptions[] = {
    {"NO_BUCKET_BUILD", &OctlOptions::no_bucket_build},     {"BUCKET_POINTS", &OctlOptions::bucket_points},
    {"SYNC_GEOM", &OctlOptions::sync_geom},                 {"NO_GEOM_HINT", &OctlOptions::no_geom_hint},
    {"NO_EXACT_DIGITS", &OctlOptions::no_exact_digits},     {"NO_FAST_ORDER", &OctlOptions::no_fast_order},
    {"NO_BUCKET_HISTORY", &OctlOptions::no_bucket_history}, {"NO_CUBE_FAST", &OctlOptions::no_cube_fast},
    {"NO_CUBE_PREFIX", &OctlOptions::no_cube_prefix},       {"CUBE_PREFIX_MIN", &OctlOptions::cube_prefix_min},
    {"NO_INCREMENTAL", &OctlOptions::no_incremental},       {"ROUTE_SELF_SENDRECV", &OctlOptions::route_self_sendrecv},
    {"TRACE_BUILD", &OctlOptions::trace_build},             {"SCAN", &OctlOptions::scan_mode},
    {"NO_FUSED_TABLES", &OctlOptions::no_fused_tables},     {"NO_SPIN_WAIT", &OctlOptions::no_spin_wait},
    {"NO_SPEC_FINISH", &OctlOptions::no_spec_finish},       {"RANSAC_WAVES", &OctlOptions::ransac_waves},
    {"GEOM_MARGIN", &OctlOptions::geom_margin},             {"NO_RANSAC_PRESCREEN", &OctlOptions::no_ransac_prescreen},
};
}  // namespace

int64_t* octl_option_field(OctlOptions& o, const char* name) {
  if (!name) return nullptr;
  if (std::strncmp(name, "OCTL_", 5) == 0) name += 5;
  for (const OptName& e : kOptions)
    if (std::strcmp(e.name, name) == 0) return &(o.*(e.field));
  return nullptr;
}

// the environment is read HERE and nowhere else on a build's path: once per context
static void options_from_environment(OctlOptions& o) {
  char var[64];
  for (const OptName& e : kOptions) {
    std::snprintf(var, sizeof(var), "OCTL_%s", e.name);
    const char* v = getenv(var);
    if (!v) continue;
    char* end = nullptr;
    const long long x = std::strtoll(v, &end, 10);
    // ("OCTL_SCAN=1pass" / "3pass" read as 1 / 3; a switch that is set to something that is not a number is on)
    o.*(e.field) = (end != v) ? (int64_t)x : 1;
  }
}

extern "C" int octl_debug_set_option(octl_ctx* ctx, const char* name, int64_t value) {
  if (!ctx) return OCTL_E_INVALID;
  int64_t* f = octl_option_field(ctx->opt, name);
  if (!f) return octl_set_error(ctx, OCTL_E_INVALID, "no such option: %s", name ? name : "(null)");
  *f = value;
  return OCTL_OK;
}

std::atomic<uint64_t> g_octl_launches{0};

extern "C" int octl_debug_launches(uint64_t* count) {
  if (!count) return OCTL_E_INVALID;
  *count = g_octl_launches.load(std::memory_order_relaxed);
  return OCTL_OK;
}

std::atomic<uint64_t> g_octl_spec_held{0}, g_octl_spec_missed{0};

extern "C" int octl_debug_spec_finish(uint64_t* held, uint64_t* missed) {
  if (!held || !missed) return OCTL_E_INVALID;
  *held = g_octl_spec_held.load(std::memory_order_relaxed);
  *missed = g_octl_spec_missed.load(std::memory_order_relaxed);
  return OCTL_OK;
}

extern "C" int octl_debug_host_syncs(uint64_t* count) {
  if (!count) return OCTL_E_INVALID;
  *count = g_octl_host_syncs.load(std::memory_order_relaxed);
  return OCTL_OK;
}

// a polite spin: the x86 PAUSE / the arm YIELD hint where there is one, nothing elsewhere
static inline void octl_cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
  __builtin_ia32_pause();
#elif defined(__aarch64__) || defined(__arm__)
  __asm__ __volatile__("yield");
#else
  std::atomic_signal_fence(std::memory_order_seq_cst);
#endif
}

uint32_t octl_wait_next_seq(octl_ctx* ctx) {
  if (++ctx->wait_seq == 0) ++ctx->wait_seq;
  return ctx->wait_seq;
}

// a RANSAC launch met a block larger than it was promised (ransac.hip: k_block_prepare): reported by the next wait
static int mirror_check_violation(octl_ctx* ctx) {
  volatile uint32_t* m = static_cast<volatile uint32_t*>(ctx->small_host);
  if (!m[MIRROR_RS_VIOLATION]) return OCTL_OK;
  m[MIRROR_RS_VIOLATION] = 0;
  return octl_set_error(ctx, OCTL_E_STATE,
                        "RANSAC: a block is larger than the launch's size bound (the kernels for larger blocks were not "
                        "launched, its mask is stale): the block table changed without resetting the bound");
}

int octl_wait_mirror_flags(octl_ctx* ctx, const int* words, int n_words, uint32_t seq, int64_t budget_us) {
  volatile uint32_t* m = static_cast<volatile uint32_t*>(ctx->small_host);
  auto all_there = [&]() {
    for (int i = 0; i < n_words; ++i)
      if (m[words[i]] != seq) return false;
    return true;
  };
  if (!ctx->opt.no_spin_wait && budget_us > 0) {
    g_octl_host_syncs.fetch_add(1, std::memory_order_relaxed);
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
      for (int spin = 0; spin < 64; ++spin) {
        if (all_there()) {
          std::atomic_thread_fence(std::memory_order_acquire);
          return mirror_check_violation(ctx);
        }
        octl_cpu_relax();
      }
      if (std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() > budget_us)
        break;
    }
    g_octl_host_syncs.fetch_sub(1, std::memory_order_relaxed);  // (the synchronisation below counts itself)
  }
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  if (!all_there())
    return octl_set_error(ctx, OCTL_E_HIP, "a kernel did not publish its results (mirror flag %d)", words[0]);
  return mirror_check_violation(ctx);
}

bool octl_ctx_side_stream(octl_ctx* ctx) {
  if (ctx->self_stream) return true;
  hipStream_t st = nullptr;
  hipEvent_t gate = nullptr, done = nullptr;
  if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess &&
      hipEventCreateWithFlags(&gate, hipEventDisableTiming) == hipSuccess &&
      hipEventCreateWithFlags(&done, hipEventDisableTiming) == hipSuccess) {
    ctx->self_stream = st;
    ctx->self_gate = gate;
    ctx->self_done = done;
    return true;
  }
  if (done) (void)hipEventDestroy(done);
  if (gate) (void)hipEventDestroy(gate);
  if (st) (void)hipStreamDestroy(st);
  (void)hipGetLastError();
  return false;
}

int octl_ctx_cus(octl_ctx* ctx) {
  if (ctx->cus <= 0) {
    hipDeviceProp_t prop;
    ctx->cus = (hipGetDeviceProperties(&prop, ctx->device) == hipSuccess && prop.multiProcessorCount > 0)
                   ? prop.multiProcessorCount
                   : 256;
  }
  return ctx->cus;
}

// Test hook (octl_debug_fail_alloc): the n-th GROWTH of a device buffer from the moment the hook is armed fails the
// way a failed hipMalloc does.  Every device allocation of the library goes through devbuf_reserve, so sweeping n
// over an operation visits every allocation-failure path it has.
static std::atomic<int64_t> g_fail_alloc_at{0};    // 0: disarmed
static std::atomic<int64_t> g_alloc_growths{0};    // growths since the hook was last armed

extern "C" int octl_debug_fail_alloc(int64_t nth, int64_t* seen) {
  if (seen) *seen = g_alloc_growths.load(std::memory_order_relaxed);
  g_alloc_growths.store(0, std::memory_order_relaxed);
  g_fail_alloc_at.store(nth > 0 ? nth : 0, std::memory_order_relaxed);
  return OCTL_OK;
}

int devbuf_reserve(octl_ctx* ctx, DevBuf& b, size_t bytes, int keep) {
  if (bytes <= b.cap) return OCTL_OK;
  {
    const int64_t k = g_alloc_growths.fetch_add(1, std::memory_order_relaxed) + 1;
    const int64_t at = g_fail_alloc_at.load(std::memory_order_relaxed);
    if (at > 0 && k == at)
      return octl_set_error(ctx, OCTL_E_NOMEM, "hipMalloc(%zu) failed: injected by octl_debug_fail_alloc (growth %lld)",
                            bytes, (long long)k);
  }
  size_t want = bytes + bytes / 4 + 256;  // grow with slack so level loops rarely realloc
  // a buffer that is appended to (the point store: one pose after the other) doubles, so that P appends cost
  // O(log P) reallocations and copies instead of one for every 25 % of growth (64 poses: 19 -> 6)
  if (keep && b.p && b.cap) want = std::max(want, 2 * b.cap);
  void* np = nullptr;
  hipError_t e = hipSuccess;
  // the smallest pooled block that fits without wasting more than the request again
  int best = -1;
  for (int i = 0; i < (int)ctx->pool.size(); ++i)
    if (ctx->pool[i].cap >= bytes && ctx->pool[i].cap <= 2 * want &&
        (best < 0 || ctx->pool[i].cap < ctx->pool[best].cap))
      best = i;
  if (best >= 0) {
    np = ctx->pool[best].p;
    want = ctx->pool[best].cap;
    ctx->pool_bytes -= want;
    ctx->pool.erase(ctx->pool.begin() + best);
  } else {
    e = hipMalloc(&np, want);
    if (e != hipSuccess && !ctx->pool.empty()) {  // give the pool back to the allocator and try again
      for (auto& pb : ctx->pool) (void)hipFree(pb.p);
      ctx->pool.clear();
      ctx->pool_bytes = 0;
      e = hipMalloc(&np, want);
    }
  }
  if (e != hipSuccess)
    return octl_set_error(ctx, OCTL_E_NOMEM, "hipMalloc(%zu) failed: %s", want,
                          hipGetErrorString(e));
  if (keep && b.p && b.cap) {
    e = hipMemcpyAsync(np, b.p, b.cap, hipMemcpyDeviceToDevice, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
      (void)hipFree(np);
      return octl_set_error(ctx, OCTL_E_HIP, "devbuf copy failed: %s", hipGetErrorString(e));
    }
  } else if (b.p) {
    // the old block may still be referenced by queued kernels
    (void)hipStreamSynchronize(ctx->stream);
  }
  if (b.p) devbuf_release(ctx, b);  // (the stream has been synchronised above)
  b.p = np;
  b.cap = want;
  return OCTL_OK;
}

void devbuf_release(octl_ctx* ctx, DevBuf& b) {
  if (!b.p) return;
  // Only an over-full pool (> 48 GiB or 2048 blocks parked) goes back to the allocator.  Small blocks are
  // parked too: hipFree synchronises the WHOLE device - a 400 KB scratch block that grew in the middle of a
  // scan made the compute stream wait 3 ms for the copy stream's upload of the next scan.
  if (ctx && ctx->pool_bytes + b.cap <= ((size_t)48 << 30) && ctx->pool.size() < 2048) {
    ctx->pool.push_back(b);
    ctx->pool_bytes += b.cap;
  } else {
    (void)hipFree(b.p);
  }
  b.p = nullptr;
  b.cap = 0;
}

int pin_region_wait(octl_ctx* ctx, int r) {
  // (a copy out of the region that finished long ago - the usual case - costs no wait)
  if (ctx->pin_event[r] && hipEventQuery(ctx->pin_event[r]) != hipSuccess) {
    (void)hipGetLastError();  // (hipErrorNotReady is not an error here; keep it out of the launch checks)
    HIP_TRY(ctx, hipEventSynchronize(ctx->pin_event[r]));
  }
  return OCTL_OK;
}

int pin_region_mark(octl_ctx* ctx, int r) {
  if (!ctx->pin_event[r])
    HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->pin_event[r], hipEventDisableTiming));
  HIP_TRY(ctx, hipEventRecord(ctx->pin_event[r], ctx->stream));
  return OCTL_OK;
}

namespace {
__global__ __launch_bounds__(256) void k_copy_u64(const uint64_t* __restrict__ src, uint64_t* __restrict__ dst, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
}  // namespace

int octl_copy_from_pinned(octl_ctx* ctx, void* dst_dev, const void* src_pinned, size_t bytes) {
  const size_t n = bytes / 8;
  if (n == 0) return OCTL_OK;
  const unsigned grid = (unsigned)std::min<size_t>(64, (n + 255) / 256);
  OCTL_LAUNCH(k_copy_u64, dim3(grid), dim3(256), 0, ctx->stream, static_cast<const uint64_t*>(src_pinned),
                     static_cast<uint64_t*>(dst_dev), n);
  HIP_TRY(ctx, hipGetLastError());
  return OCTL_OK;
}

void devbuf_free(DevBuf& b) {
  if (b.p) (void)hipFree(b.p);
  b.p = nullptr;
  b.cap = 0;
}

KTimer::KTimer(octl_ctx* c, const char* name) : ctx(c) {
  if (!ctx->profiling) return;
  if (ctx->profiling == 2 && std::strcmp(name, "ransac") != 0) return;
  PendingEvent pe;
  pe.name = name;
  for (hipEvent_t* ev : {&pe.start, &pe.stop}) {
    if (!ctx->event_pool.empty()) {
      *ev = ctx->event_pool.back();
      ctx->event_pool.pop_back();
    } else if (hipEventCreate(ev) != hipSuccess) {
      return;
    }
  }
  (void)hipEventRecord(pe.start, ctx->stream);
  idx = (int)ctx->pending.size();
  ctx->pending.push_back(pe);
}

KTimer::~KTimer() {
  if (idx >= 0) (void)hipEventRecord(ctx->pending[idx].stop, ctx->stream);
}

int octl_collect_timings(octl_ctx* ctx) {
  if (ctx->pending.empty()) return OCTL_OK;
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  for (auto& pe : ctx->pending) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, pe.start, pe.stop) == hipSuccess) {
      auto& t = ctx->timings[pe.name];
      t.ms += ms;
      t.launches += 1;
    }
    ctx->event_pool.push_back(pe.start);
    ctx->event_pool.push_back(pe.stop);
  }
  ctx->pending.clear();
  return OCTL_OK;
}

int ctx_wait_uploads(octl_ctx* ctx, const void* p, size_t bytes) {
  const char* a = static_cast<const char*>(p);
  for (auto& u : ctx->uploads) {
    if (a && (a + bytes <= u.dst || u.dst + u.bytes <= a)) continue;  // another buffer's upload
    HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, u.done, 0));
  }
  return OCTL_OK;
}

extern "C" {

int octl_abi_version(void) { return OCTL_ABI_VERSION; }

int octl_device_count(int* count) {
  if (!count) return OCTL_E_INVALID;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  *count = (e == hipSuccess) ? n : 0;
  return e == hipSuccess ? OCTL_OK : OCTL_E_HIP;
}

int octl_ctx_create(int device_id, octl_ctx** out) {
  if (!out) return OCTL_E_INVALID;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device_id < 0 || device_id >= n)
    return OCTL_E_HIP;
  if (hipSetDevice(device_id) != hipSuccess) return OCTL_E_HIP;
  octl_ctx* ctx = new octl_ctx();
  ctx->device = device_id;
  options_from_environment(ctx->opt);
  if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
    delete ctx;
    return OCTL_E_HIP;
  }
  if (hipMalloc(&ctx->small.p, 4096) != hipSuccess ||
      hipHostMalloc(&ctx->small_host, 4096, hipHostMallocDefault) != hipSuccess ||
      hipHostMalloc(&ctx->pinned, OCTL_PINNED_BYTES, hipHostMallocDefault) != hipSuccess) {
    if (ctx->small.p) (void)hipFree(ctx->small.p);
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return OCTL_E_NOMEM;
  }
  ctx->small.cap = 4096;
  // (the mirror carries the sequence flags the host polls: a recycled page must not hold a word that equals the first
  //  sequence number a context waits for)
  std::memset(ctx->small_host, 0, 4096);
  (void)hipMemsetAsync(ctx->small.p, 0, 4096, ctx->stream);
  (void)hipStreamSynchronize(ctx->stream);
  *out = ctx;
  return OCTL_OK;
}

int octl_comm_destroy(octl_ctx* ctx);

void octl_ctx_destroy(octl_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  if (ctx->comm) (void)octl_comm_destroy(ctx);
  for (auto& pe : ctx->pending) {
    (void)hipEventDestroy(pe.start);
    (void)hipEventDestroy(pe.stop);
  }
  for (auto ev : ctx->event_pool) (void)hipEventDestroy(ev);
  if (ctx->handoff_event) (void)hipEventDestroy(ctx->handoff_event);
  for (auto ev : ctx->pin_event)
    if (ev) (void)hipEventDestroy(ev);
  for (auto& pb : ctx->pool) (void)hipFree(pb.p);
  ctx->pool.clear();
  for (auto& ub : ctx->user_blocks) (void)hipFree(ub.first);  // (blocks the caller never gave back)
  ctx->user_blocks.clear();
  for (auto& b : ctx->scan_tmp) devbuf_free(b);
  devbuf_free(ctx->hyp_dev);
  devbuf_free(ctx->rs_counters);
  devbuf_free(ctx->scan_status);
  devbuf_free(ctx->small);
  devbuf_free(ctx->routed_xyz);
  devbuf_free(ctx->routed_gidx);
  for (DevBuf* b : {&ctx->rt_hist, &ctx->rt_counts, &ctx->rt_matrix, &ctx->rt_send_xyz,
                    &ctx->rt_send_gidx})
    devbuf_free(*b);
  if (ctx->small_host) (void)hipHostFree(ctx->small_host);
  if (ctx->pinned) (void)hipHostFree(ctx->pinned);
  if (ctx->self_stream) {
    (void)hipStreamSynchronize(ctx->self_stream);
    (void)hipStreamDestroy(ctx->self_stream);
    if (ctx->self_gate) (void)hipEventDestroy(ctx->self_gate);
    if (ctx->self_done) (void)hipEventDestroy(ctx->self_done);
  }
  if (ctx->copy_stream) {
    (void)hipStreamSynchronize(ctx->copy_stream);
    (void)hipStreamDestroy(ctx->copy_stream);
  }
  if (ctx->copy_gate) (void)hipEventDestroy(ctx->copy_gate);
  for (auto& u : ctx->uploads) (void)hipEventDestroy(u.done);
  (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

const char* octl_last_error(const octl_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int octl_ctx_sync(octl_ctx* ctx) {
  if (!ctx) return OCTL_E_INVALID;
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return mirror_check_violation(ctx);
}

// ---- asynchronous host feed ----------------------------------------------------------------------------------
// The reference copies every scan to the device inside evaluate() and waits for it (cuda_ransac.py:57-67); a
// SLAM loop that hands scan i+1 over while scan i is being fitted hides the 240 MB / 4.9 ms of PCIe behind
// the 5 ms of compute.  Page-locked host memory makes the copy a pure DMA the host does not wait for.
int octl_host_alloc(octl_ctx* ctx, int64_t bytes, void** p) {
  if (!ctx || !p || bytes < 0) return OCTL_E_INVALID;
  (void)hipSetDevice(ctx->device);
  hipError_t e = hipHostMalloc(p, (size_t)(bytes > 0 ? bytes : 1), hipHostMallocDefault);
  if (e != hipSuccess)
    return octl_set_error(ctx, OCTL_E_NOMEM, "hipHostMalloc(%lld) failed: %s", (long long)bytes, hipGetErrorString(e));
  return OCTL_OK;
}

int octl_host_free(octl_ctx* ctx, void* p) {
  if (!ctx) return OCTL_E_INVALID;
  if (ctx->copy_stream) (void)hipStreamSynchronize(ctx->copy_stream);  // a copy out of it may be in flight
  if (p) HIP_TRY(ctx, hipHostFree(p));
  return OCTL_OK;
}

int octl_dev_upload_async(octl_ctx* ctx, void* dptr, const void* src, int64_t bytes) {
  if (!ctx || bytes < 0 || (bytes > 0 && (!dptr || !src))) return OCTL_E_INVALID;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (!ctx->copy_stream) {
    HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
    HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->copy_gate, hipEventDisableTiming));
  }
  if (bytes == 0) return OCTL_OK;
  // the destination may still be read by compute work that was enqueued before this call
  HIP_TRY(ctx, hipEventRecord(ctx->copy_gate, ctx->stream));
  HIP_TRY(ctx, hipStreamWaitEvent(ctx->copy_stream, ctx->copy_gate, 0));
  HIP_TRY(ctx, hipMemcpyAsync(dptr, src, (size_t)bytes, hipMemcpyHostToDevice, ctx->copy_stream));
  // finished uploads leave the list (their events are reused)
  hipEvent_t ev = nullptr;
  for (size_t i = 0; i < ctx->uploads.size();) {
    if (hipEventQuery(ctx->uploads[i].done) == hipSuccess) {
      if (!ev) ev = ctx->uploads[i].done; else (void)hipEventDestroy(ctx->uploads[i].done);
      ctx->uploads.erase(ctx->uploads.begin() + (long)i);
    } else {
      (void)hipGetLastError();  // (hipErrorNotReady is not an error here)
      ++i;
    }
  }
  if (!ev) HIP_TRY(ctx, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  HIP_TRY(ctx, hipEventRecord(ev, ctx->copy_stream));
  ctx->uploads.push_back(octl_ctx::Upload{static_cast<const char*>(dptr), (size_t)bytes, ev});
  return OCTL_OK;
}

int octl_ctx_sync_uploads(octl_ctx* ctx) {
  if (!ctx) return OCTL_E_INVALID;
  if (ctx->copy_stream) HIP_TRY(ctx, hipStreamSynchronize(ctx->copy_stream));
  return OCTL_OK;
}

int octl_ctx_set_profiling(octl_ctx* ctx, int enabled) {
  if (!ctx) return OCTL_E_INVALID;
  OCTL_TRY(octl_collect_timings(ctx));
  ctx->profiling = enabled == 2 ? 2 : (enabled != 0 ? 1 : 0);
  ctx->timings.clear();
  return OCTL_OK;
}

int octl_ctx_get_timings(octl_ctx* ctx, char* names, int name_stride, float* ms,
                         int64_t* launches, int cap, int* n) {
  if (!ctx || !n) return OCTL_E_INVALID;
  OCTL_TRY(octl_collect_timings(ctx));
  *n = (int)ctx->timings.size();
  int i = 0;
  for (auto& kv : ctx->timings) {
    if (i >= cap) break;
    if (names && name_stride > 0) {
      std::snprintf(names + (size_t)i * name_stride, name_stride, "%s", kv.first.c_str());
    }
    if (ms) ms[i] = (float)kv.second.ms;
    if (launches) launches[i] = kv.second.launches;
    ++i;
  }
  return OCTL_OK;
}

int octl_dev_alloc(octl_ctx* ctx, int64_t bytes, void** dptr) {
  if (!ctx || !dptr || bytes < 0) return OCTL_E_INVALID;
  (void)hipSetDevice(ctx->device);
  DevBuf b;  // (from the context's pool when a parked block fits)
  OCTL_TRY(devbuf_reserve(ctx, b, (size_t)(bytes > 0 ? bytes : 1)));
  ctx->user_blocks[b.p] = b.cap;
  *dptr = b.p;
  return OCTL_OK;
}

int octl_dev_free(octl_ctx* ctx, void* dptr) {
  if (!ctx) return OCTL_E_INVALID;
  if (!dptr) return OCTL_OK;
  // uploads into the block and the compute work that may read it must have finished; other uploads carry on
  const char* a = static_cast<const char*>(dptr);
  auto it = ctx->user_blocks.find(dptr);
  const size_t cap = it != ctx->user_blocks.end() ? it->second : 0;
  for (auto& u : ctx->uploads)
    if (!cap || !(a + cap <= u.dst || u.dst + u.bytes <= a)) (void)hipEventSynchronize(u.done);
  (void)hipStreamSynchronize(ctx->stream);
  if (it == ctx->user_blocks.end()) {
    HIP_TRY(ctx, hipFree(dptr));
    return OCTL_OK;
  }
  DevBuf b{dptr, cap};
  ctx->user_blocks.erase(it);
  devbuf_release(ctx, b);
  return OCTL_OK;
}

int octl_dev_upload(octl_ctx* ctx, void* dptr, const void* src, int64_t bytes) {
  if (!ctx || (bytes > 0 && (!dptr || !src))) return OCTL_E_INVALID;
  HIP_TRY(ctx, hipMemcpyAsync(dptr, src, (size_t)bytes, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return OCTL_OK;
}

int octl_dev_download(octl_ctx* ctx, void* dst, const void* dptr, int64_t bytes) {
  if (!ctx || (bytes > 0 && (!dptr || !dst))) return OCTL_E_INVALID;
  // (the source may be the target of an octl_dev_upload_async that is still in flight on the copy stream)
  if (bytes > 0) OCTL_TRY(ctx_wait_uploads(ctx, dptr, (size_t)bytes));
  HIP_TRY(ctx, hipMemcpyAsync(dst, dptr, (size_t)bytes, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return OCTL_OK;
}

int octl_dev_copy_bandwidth(octl_ctx* ctx, int64_t bytes, int iters, double* bytes_per_s) {
  if (!ctx || !bytes_per_s || bytes <= 0 || iters <= 0) return OCTL_E_INVALID;
  void *a = nullptr, *b = nullptr;
  HIP_TRY(ctx, hipMalloc(&a, (size_t)bytes));
  if (hipMalloc(&b, (size_t)bytes) != hipSuccess) {
    (void)hipFree(a);
    return octl_set_error(ctx, OCTL_E_NOMEM, "hipMalloc failed");
  }
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  (void)hipMemsetAsync(a, 1, (size_t)bytes, ctx->stream);
  (void)hipMemcpyAsync(b, a, (size_t)bytes, hipMemcpyDeviceToDevice, ctx->stream);
  (void)hipEventRecord(e0, ctx->stream);
  for (int i = 0; i < iters; ++i)
    (void)hipMemcpyAsync(b, a, (size_t)bytes, hipMemcpyDeviceToDevice, ctx->stream);
  (void)hipEventRecord(e1, ctx->stream);
  hipError_t e = hipStreamSynchronize(ctx->stream);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  (void)hipFree(a);
  (void)hipFree(b);
  if (e != hipSuccess) return octl_set_error(ctx, OCTL_E_HIP, "copy failed");
  // a copy reads and writes every byte
  *bytes_per_s = 2.0 * (double)bytes * iters / (ms * 1e-3);
  return OCTL_OK;
}

}  // extern "C"
