// Shared host-side infrastructure of liboctree_hip.so: context, device buffers, error
// plumbing, per-kernel hipEvent timers.  gfx950 only.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/octreelib_hip.h"

// Every place where the host waits for the device is counted (octl_debug_host_syncs): the number of
// round trips per step is part of what bench.py reports.
#include <atomic>
extern std::atomic<uint64_t> g_octl_host_syncs;
static inline hipError_t octl_counted_stream_sync(hipStream_t s) {
  g_octl_host_syncs.fetch_add(1, std::memory_order_relaxed);
  return hipStreamSynchronize(s);
}
static inline hipError_t octl_counted_event_sync(hipEvent_t e) {
  g_octl_host_syncs.fetch_add(1, std::memory_order_relaxed);
  return hipEventSynchronize(e);
}
#define hipStreamSynchronize(s) octl_counted_stream_sync(s)
#define hipEventSynchronize(e) octl_counted_event_sync(e)
// ... and every kernel launch and asynchronous fill (a fill is a kernel of the runtime's): octl_debug_launches -
// launches per step is the other figure a small scan lives by
extern std::atomic<uint64_t> g_octl_launches;
// speculative launches of k_bucket_finish that did the work / that the host had to repeat (octl_debug_spec_finish)
extern std::atomic<uint64_t> g_octl_spec_held, g_octl_spec_missed;
// (the project's own launch macro around the PUBLIC hipLaunchKernelGGL - not a redefinition of the runtime's macro
//  through its internals: a ROCm update that renames those would break every translation unit)
#define OCTL_LAUNCH(...)                                                      \
  do {                                                                        \
    g_octl_launches.fetch_add(1, std::memory_order_relaxed);                  \
    hipLaunchKernelGGL(__VA_ARGS__);                                          \
  } while (0)
#define hipMemsetAsync(...) (g_octl_launches.fetch_add(1, std::memory_order_relaxed), hipMemsetAsync(__VA_ARGS__))

// Compile-time experiment switches that change RESULTS (ablations, duplicated work) or add instrumentation never
// belong in a shipped library: they only compile when the variant build script defines OCTL_EXPERIMENTS
// (tools/build_variant.sh -> build/variants/NAME.so, loaded through OCTREELIB_AMD_LIB; the Makefile never does).
#if !defined(OCTL_EXPERIMENTS) && \
    (defined(PS_DUP_KEYS) || defined(PS_DUP_STORE) || defined(RS_COUNTS) || defined(BB_STAMPS))
#error "PS_DUP_* / RS_COUNTS / BB_STAMPS are experiments: build them with tools/build_variant.sh (-DOCTL_EXPERIMENTS)"
#endif

#define OCTL_WAVE 64
#define OCTL_PINNED_BYTES (256 * 1024)

struct KernelTiming {
  double ms = 0.0;
  int64_t launches = 0;
};

struct PendingEvent {
  std::string name;
  hipEvent_t start, stop;
};

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  template <typename T>
  T* as() const {
    return reinterpret_cast<T*>(p);
  }
};

// Diagnostic switches of a context (tests and A/B runs compare code paths with them).  Read from the environment
// ONCE, when the context is created (OCTL_<NAME>=value), or set afterwards with octl_debug_set_option; nothing on a
// build's path calls getenv.  All default to 0 = the shipped behaviour.
struct OctlOptions {
  int64_t no_bucket_build = 0;     // OCTL_NO_BUCKET_BUILD: every build through the level-synchronous path
  int64_t bucket_points = 0;       // OCTL_BUCKET_POINTS: average points per bucket (0: 2560; 1280 up to 400 k points)
  int64_t sync_geom = 0;           // OCTL_SYNC_GEOM: the host-side form of the key geometry on small clouds too
  int64_t no_geom_hint = 0;        // OCTL_NO_GEOM_HINT: no geometry carried over from the context's previous build
  int64_t no_exact_digits = 0;     // OCTL_NO_EXACT_DIGITS: child digits level by level, never six at once
  int64_t no_fast_order = 0;       // OCTL_NO_FAST_ORDER: the listing order always from order.hip
  int64_t no_bucket_history = 0;   // OCTL_NO_BUCKET_HISTORY: a subdivide over a previous scheme through the general path
  int64_t no_cube_fast = 0;        // OCTL_NO_CUBE_FAST: a fresh single cube through key generation
  int64_t no_cube_prefix = 0;      // OCTL_NO_CUBE_PREFIX: no prefix partition of big single cubes
  int64_t cube_prefix_min = 0;     // OCTL_CUBE_PREFIX_MIN: points from which a single cube is prefix-partitioned (0: 4 Mi)
  int64_t no_incremental = 0;      // OCTL_NO_INCREMENTAL: late poses by re-placement of every stored point
  int64_t route_self_sendrecv = 0; // OCTL_ROUTE_SELF_SENDRECV: a rank's own part through RCCL too (1-rank rehearsal)
  int64_t trace_build = 0;         // OCTL_TRACE_BUILD: host wall time between the phases of forest_build on stderr
  int64_t scan_mode = 0;           // OCTL_SCAN: 1 = single-pass scan always, 3 = three-kernel scan always
  int64_t no_fused_tables = 0;     // OCTL_NO_FUSED_TABLES: the small table chains as separate launches (A/B)
  int64_t no_spin_wait = 0;        // OCTL_NO_SPIN_WAIT: every host wait is a hipStreamSynchronize (A/B)
  int64_t ransac_waves = 0;        // OCTL_RANSAC_WAVES: waves per block of the H > 256 RANSAC instances (0: the library's policy; 1, 2, 4)
  int64_t geom_margin = 0;         // OCTL_GEOM_MARGIN: voxels of slack around the true box in a single-pass / hinted key geometry (0: one voxel; negative: none)
  int64_t no_ransac_prescreen = 0; // OCTL_NO_RANSAC_PRESCREEN: every hypothesis of a leaf through the exact plane fit (no approximate prescreen)
  int64_t no_spec_finish = 0;      // OCTL_NO_SPEC_FINISH: k_bucket_finish only behind the host's look at the totals (A/B)
};
// name (without the OCTL_ prefix or with it) -> field; nullptr when there is no such switch
int64_t* octl_option_field(OctlOptions& o, const char* name);

struct octl_ctx {
  int device = 0;
  OctlOptions opt;
  hipStream_t stream = nullptr;
  std::string err;
  int profiling = 0;  // 0 off, 1 every timed region, 2 the RANSAC kernel only (an event pair costs ~10 us of pipeline)
  std::map<std::string, KernelTiming> timings;
  std::vector<PendingEvent> pending;
  std::vector<hipEvent_t> event_pool;
  // scratch for scans / sorts (grown on demand, reused across calls)
  DevBuf scan_tmp[3];
  DevBuf scan_status;            // single-pass scan: per-tile status words
  uint32_t scan_epoch = 0;
  int cus = 0;      // compute units of the device (queried once: hipGetDeviceProperties is not free)
  DevBuf small;     // 4 KiB of device scalars (counters, flags)
  void* small_host = nullptr;  // pinned mirror
  void* pinned = nullptr;      // pinned staging for small uploads (OCTL_PINNED_BYTES)
  // one event per staging region ([0,128K) build, [128K,192K) hypothesis table, [192K,256K) pose
  // epochs): recorded after the region's H2D copy, waited for before the region is written again
  hipEvent_t pin_event[3] = {nullptr, nullptr, nullptr};
  hipEvent_t handoff_event = nullptr;  // orders a buffer handed to another context behind this stream
  // asynchronous host feed (octl_dev_upload_async): a second stream for host-to-device copies, so that the
  // upload of scan i+1 overlaps the build and fit of scan i; copy_gate orders a copy behind the compute work
  // enqueued before it (its destination may still be read), copy_done orders compute behind the copies
  hipStream_t copy_stream = nullptr;
  hipEvent_t copy_gate = nullptr;
  // The context's SIDE stream (octl_ctx_side_stream): work that may run next to the context's stream between two
  // events - self_gate: what it reads is complete on the context's stream; self_done: the side work is.
  // route.hip: a rank's own part of the all-to-all, next to the RCCL transfers; ransac.hip: the instances for the
  // larger blocks of a split launch, next to the one-wave instance.
  hipStream_t self_stream = nullptr;
  hipEvent_t self_gate = nullptr, self_done = nullptr;
  struct Upload { const char* dst; size_t bytes; hipEvent_t done; };
  std::vector<Upload> uploads;  // copies that may still be in flight, each with the event recorded behind it
  // device blocks handed back by destroyed forests, kept for the next one: a fresh Grid per scan
  // otherwise pays ~10 ms of hipMalloc / hipFree per build for its dozen large buffers
  std::vector<DevBuf> pool;
  size_t pool_bytes = 0;
  // blocks handed out by octl_dev_alloc (pointer -> capacity): they come from and go back to the pool too - a
  // hipMalloc / hipFree pair of a 240 MB scan buffer costs milliseconds and synchronises the device
  std::map<void*, size_t> user_blocks;
  // key geometry (padded voxel box, bucket width) formed from the TRUE box of the last bucket build on this context:
  // the next build of a cloud that was taken in place starts its histogram pass under this geometry and finds its own
  // true box in the same pass (bucket_build.hip: hinted geometry); opaque here
  unsigned char geom_hint[192] = {0};
  bool geom_hint_valid = false;
  bool geom_hint_staged = false;  // forest_build's first launch has put the hint into the scalar block already
  struct octl_forest* pending_mask_forest = nullptr;  // the forest whose apply_mask totals are still in flight (one per context: they share the mirror's words)
  uint64_t geom_hint_want = 0;
  bool geom_hint_two_pass = false;  // the hint is the geometry of a TWO-pass build (> 4096 buckets, host-side form)
  // the last build found a sparse scene (more than 4096 buckets): the next one skips the single-pass attempt
  bool geom_sparse = false;
  // the last bucket build had buckets beyond 4096 points (a skewed scene): the next one launches the chunk kernels
  // beside k_bucket_build right away instead of learning about them from the totals (two launches an even scene saves)
  bool had_chunks = false;
  // node / voxel / block counts of the context's previous bucket build: the tables of a speculative k_bucket_finish
  // are sized from them (0: no bucket build yet)
  int64_t spec_points = 0, spec_nodes = 0, spec_vox = 0, spec_blocks = 0;
  // the hypothesis table of the last octl_forest_ransac_all on this context (CudaRansac draws it once per object,
  // cuda_ransac.py:39-41, and a loop over scans hands the same one over for every scan - to a fresh forest each
  // time): kept per CONTEXT so that it is uploaded once
  DevBuf hyp_dev;
  // launch counters of ransac.hip's preparation: two sets used alternately (a launch zeroes the next one's)
  DevBuf rs_counters;
  int rs_parity = 0;
  uint32_t wait_seq = 0;   // sequence numbers of the mirror flags (octl_wait_mirror_flags)
  bool rs_counters_dirty = false;
  std::vector<double> hyp_host;
  // RCCL (route.hip)
  void* comm = nullptr;
  int n_ranks = 1, rank = 0;
  // routed cloud of the last octl_route_points + its scratch (kept: hipMalloc per step costs ms)
  DevBuf routed_xyz, routed_gidx;
  bool routed_taken = false;  // a forest took routed_xyz over: nothing to hand out until the next route
  DevBuf rt_hist, rt_counts, rt_matrix, rt_send_xyz, rt_send_gidx;
  int64_t routed_n = 0;
};

int octl_set_error(octl_ctx* ctx, int code, const char* fmt, ...);
// Host waits for a few scalars, not for the stream: the kernel that produces them writes them into the pinned mirror
// (ctx->small_host) and then the launch's sequence number into flag word(s) of that mirror; the host polls the flags
// for up to budget_us and only then falls back to hipStreamSynchronize.  (A stream synchronisation returns 15-25 us
// after the kernel has finished - interrupt, wake-up - which is a tenth of a 100 k-point scan.)  Counted as a host
// wait like a synchronisation.  Words of the mirror: MIRROR_FLAG_BUILD (bucket totals), MIRROR_FLAG_MASK0/1
// (apply_mask: kept points, surviving blocks), MIRROR_MASK_TOTALS (the two totals themselves).
// Layout of the 4 KB mirror (words): [0, MIRROR_COPY_WORDS) targets of small device-to-host copies and the scalar
// block a build mirrors; MIRROR_BBOX_WORD.. the voxel box read back by the host-side geometry; then the words above,
// which no copy may reach.
// MIRROR_RS_VIOLATION: set by ransac.hip's preparation kernel when a block is larger than the launch was told any
// block could be (max_block: the instances for larger blocks are not launched then) - every host wait checks it.
enum { MIRROR_COPY_WORDS = 512, MIRROR_BBOX_WORD = 896, MIRROR_RS_VIOLATION = 990, MIRROR_MASK_TOTALS = 992,
       MIRROR_FLAG_BUILD = 1000, MIRROR_FLAG_MASK0 = 1001, MIRROR_FLAG_MASK1 = 1002 };
static_assert(MIRROR_COPY_WORDS <= MIRROR_BBOX_WORD && MIRROR_BBOX_WORD + 8 <= MIRROR_RS_VIOLATION &&
              MIRROR_RS_VIOLATION < MIRROR_MASK_TOTALS &&
              MIRROR_MASK_TOTALS + 2 <= MIRROR_FLAG_BUILD && MIRROR_FLAG_MASK1 < 1024, "mirror layout");
uint32_t octl_wait_next_seq(octl_ctx* ctx);
int octl_wait_mirror_flags(octl_ctx* ctx, const int* words, int n_words, uint32_t seq, int64_t budget_us);
// device side: publish `seq` behind everything this thread (and, through the barriers in front of the call, its
// workgroup) has written to the mirror
__device__ __forceinline__ void mirror_publish(uint32_t* mirror, int word, uint32_t seq) {
  __threadfence_system();
  __hip_atomic_store(&mirror[word], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// the compute stream waits (on the device, no host wait) for the uploads enqueued so far whose destination
// overlaps [p, p + bytes); p == nullptr: for all of them
int ctx_wait_uploads(octl_ctx* ctx, const void* p = nullptr, size_t bytes = 0);

#define HIP_TRY(ctx, expr)                                                                 \
  do {                                                                                     \
    hipError_t _e = (expr);                                                                \
    if (_e != hipSuccess)                                                                  \
      return octl_set_error((ctx), OCTL_E_HIP, "%s failed: %s (%s:%d)", #expr,             \
                            hipGetErrorString(_e), __FILE__, __LINE__);                    \
  } while (0)

#define OCTL_TRY(expr)        \
  do {                        \
    int _r = (expr);          \
    if (_r != OCTL_OK) return _r; \
  } while (0)

// staging regions of ctx->pinned: wait until the previous asynchronous copy out of region r has
// finished / mark a new one as in flight
int pin_region_wait(octl_ctx* ctx, int r);
int pin_region_mark(octl_ctx* ctx, int r);
// copy `bytes` (a multiple of 8) from page-locked HOST memory to the device with a kernel on the context's
// stream: a small hipMemcpyAsync goes through the DMA queue and waits there behind a large upload that is in
// flight on the copy stream (measured: 2.9 ms per scan of the asynchronous feed)
int octl_copy_from_pinned(octl_ctx* ctx, void* dst_dev, const void* src_pinned, size_t bytes);

// grow-only device buffer; contents are NOT preserved unless keep != 0
int devbuf_reserve(octl_ctx* ctx, DevBuf& b, size_t bytes, int keep = 0);
void devbuf_free(DevBuf& b);
// hand a block back to the context's pool (the stream must have finished with it) instead of freeing it
void devbuf_release(octl_ctx* ctx, DevBuf& b);

// RAII timer: records hipEvents around the launches in its scope when profiling is on
struct KTimer {
  octl_ctx* ctx;
  int idx = -1;
  KTimer(octl_ctx* c, const char* name);
  ~KTimer();
};
// fold the pending event pairs into ctx->timings (synchronises the stream)
int octl_collect_timings(octl_ctx* ctx);

static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ---- device-wide primitives (scan.hip, radix_sort.hip) ------------------------------------
// exclusive prefix sum of n uint32 (in may equal out); if total_dev != nullptr the grand total
// (uint32) is written there.  n up to 2^31.
// compute units of the context's device (cached)
int octl_ctx_cus(octl_ctx* ctx);
// the side stream and its two events, created on first use; false when they cannot be had (not an error: the
// caller uses the context's stream)
bool octl_ctx_side_stream(octl_ctx* ctx);
int octl_exclusive_scan_u32(octl_ctx* ctx, const uint32_t* in, uint32_t* out, int64_t n,
                            uint32_t* total_dev);
// status words + a fresh epoch for one look-back chain of up to `tiles` workgroups (lookback.h)
int octl_scan_status_acquire(octl_ctx* ctx, int64_t tiles, uint64_t** status, uint32_t* epoch);
// stable LSD radix sort of (key u64, value u32) pairs on bits [0, key_bits).  keys[0]/vals[0]
// hold the input; keys[1]/vals[1] are ping-pong space of the same size.  *result (0 or 1)
// tells which pair of buffers holds the sorted output.
int octl_radix_sort_u64_u32(octl_ctx* ctx, uint64_t* keys[2], uint32_t* vals[2], int64_t n,
                            int key_bits, DevBuf& hist_scratch, int* result);
