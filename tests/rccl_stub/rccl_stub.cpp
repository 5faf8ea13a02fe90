// TEST INFRASTRUCTURE ONLY - never loaded by the product unless OCTL_RCCL_LIBRARY points at it.
//
// A stand-in for the ten RCCL entry points csrc/route.hip resolves with dlsym, for R > 1 ranks that are
// separate PROCESSES sharing ONE GPU - which real RCCL refuses (two ranks of a communicator on the same
// device) - so that the multi-rank half of the routing (count matrix, send / receive offsets, the grouped
// transfers, the grow-agreement all-reduce, the collective error exits) runs on a one-GPU test box.
//
// Transport: a POSIX shared-memory segment named after the unique id.  Every collective synchronises the
// caller's stream, stages through the segment with device<->host copies, and meets the other ranks at
// process-shared barriers (sense counters in the segment, spin + sleep, 120 s time-out -> ncclSystemError,
// so a rank that never arrives fails the test instead of hanging the box).  Point-to-point calls are
// recorded between ncclGroupStart / ncclGroupEnd and executed at the end of the group: all sends are
// staged, barrier, all receives are copied out, barrier - the semantics the grouped all-to-all relies on.
// Nothing here is tuned; sizes are bounded by the arena (OCTL_STUB_ARENA_MB per rank, default 64).
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

namespace {

constexpr int MAX_RANKS = 16;
constexpr int MAX_MSGS = 64;  // messages per (source rank, group)

struct Msg {
  int32_t dst;
  uint64_t offset, bytes;
};

struct Header {
  std::atomic<uint32_t> ready;      // rank 0 has initialised the segment
  std::atomic<uint32_t> arrived;    // barrier: ranks that reached the current phase
  std::atomic<uint32_t> phase;      // barrier: generation
  std::atomic<uint32_t> failed;     // some rank gave up: everybody returns an error
  uint32_t n_ranks;
  uint64_t arena_bytes;             // per rank
  uint32_t n_msgs[MAX_RANKS];
  Msg msgs[MAX_RANKS][MAX_MSGS];
};

struct Comm {
  Header* h = nullptr;
  char* arena = nullptr;  // n_ranks arenas behind the header
  size_t map_bytes = 0;
  int rank = 0, n_ranks = 1;
  char name[64];
};

struct P2P {
  bool send;
  void* ptr;
  size_t bytes;
  int peer;
  Comm* comm;
  hipStream_t stream;
};
thread_local int g_group_depth = 0;
thread_local std::vector<P2P> g_ops;
thread_local Comm* g_last_comm = nullptr;  // a group without operations still meets the other ranks' barriers

size_t type_size(ncclDataType_t t) {
  switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclFloat16: case ncclBfloat16: return 2;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    default: return 8;  // ncclInt64, ncclUint64, ncclFloat64
  }
}

bool barrier(Comm* c) {
  Header* h = c->h;
  if (h->failed.load()) return false;
  const uint32_t gen = h->phase.load();
  if (h->arrived.fetch_add(1) + 1 == (uint32_t)c->n_ranks) {
    h->arrived.store(0);
    h->phase.store(gen + 1);
    return true;
  }
  const auto t0 = std::chrono::steady_clock::now();
  while (h->phase.load() == gen) {
    if (h->failed.load()) return false;
    if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120)) {
      h->failed.store(1);
      return false;
    }
    std::this_thread::sleep_for(std::chrono::microseconds(50));
  }
  return true;
}

char* arena_of(Comm* c, int rank) { return c->arena + (size_t)rank * c->h->arena_bytes; }

ncclResult_t run_group(std::vector<P2P>& ops) {
  Comm* c = ops.empty() ? g_last_comm : ops[0].comm;
  if (!c) return ncclSuccess;
  if (!ops.empty() && hipStreamSynchronize(ops[0].stream) != hipSuccess) return ncclUnhandledCudaError;
  // 1. stage every send in this rank's arena, publish the descriptors
  uint64_t off = 0;
  uint32_t nm = 0;
  for (auto& o : ops) {
    if (!o.send) continue;
    if (nm >= MAX_MSGS || off + o.bytes > c->h->arena_bytes) {
      c->h->failed.store(1);
      return ncclInvalidUsage;
    }
    if (hipMemcpy(arena_of(c, c->rank) + off, o.ptr, o.bytes, hipMemcpyDeviceToHost) != hipSuccess)
      return ncclUnhandledCudaError;
    c->h->msgs[c->rank][nm] = Msg{o.peer, off, o.bytes};
    off += (o.bytes + 63) & ~(uint64_t)63;
    ++nm;
  }
  c->h->n_msgs[c->rank] = nm;
  if (!barrier(c)) return ncclSystemError;
  // 2. receives: the k-th receive from peer p matches the k-th message p addressed to this rank
  uint32_t next[MAX_RANKS] = {0};
  for (auto& o : ops) {
    if (o.send) continue;
    const int p = o.peer;
    uint32_t k = next[p];
    while (k < c->h->n_msgs[p] && c->h->msgs[p][k].dst != c->rank) ++k;
    if (k >= c->h->n_msgs[p] || c->h->msgs[p][k].bytes != o.bytes) {
      c->h->failed.store(1);
      return ncclInvalidUsage;  // unmatched receive: real RCCL would hang
    }
    next[p] = k + 1;
    if (hipMemcpy(o.ptr, arena_of(c, p) + c->h->msgs[p][k].offset, o.bytes, hipMemcpyHostToDevice) != hipSuccess)
      return ncclUnhandledCudaError;
  }
  // every message addressed to this rank must have been received (an unmatched send hangs real RCCL)
  for (int p = 0; p < c->n_ranks; ++p) {
    uint32_t k = next[p];
    while (k < c->h->n_msgs[p] && c->h->msgs[p][k].dst != c->rank) ++k;
    if (k < c->h->n_msgs[p]) {
      c->h->failed.store(1);
      return ncclInvalidUsage;
    }
  }
  if (!barrier(c)) return ncclSystemError;  // (the arenas may be overwritten again)
  return ncclSuccess;
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
  std::memset(id, 0, sizeof(*id));
  const auto now = std::chrono::steady_clock::now().time_since_epoch().count();
  std::snprintf(id->internal, sizeof(id->internal), "/octlstub_%d_%llx", (int)getpid(), (unsigned long long)now);
  return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int n_ranks, ncclUniqueId id, int rank) {
  if (n_ranks < 1 || n_ranks > MAX_RANKS || rank < 0 || rank >= n_ranks) return ncclInvalidArgument;
  Comm* c = new Comm();
  c->rank = rank;
  c->n_ranks = n_ranks;
  std::snprintf(c->name, sizeof(c->name), "%s", id.internal);
  const char* mb = getenv("OCTL_STUB_ARENA_MB");
  const uint64_t arena = (uint64_t)(mb ? atoi(mb) : 64) << 20;
  c->map_bytes = ((sizeof(Header) + 4095) & ~(size_t)4095) + (size_t)n_ranks * arena;
  int fd = -1;
  if (rank == 0) {
    fd = shm_open(c->name, O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, (off_t)c->map_bytes) != 0) return ncclSystemError;
  } else {
    for (int tries = 0; tries < 120000 && fd < 0; ++tries) {  // up to ~120 s
      fd = shm_open(c->name, O_RDWR, 0600);
      struct stat sb;
      if (fd >= 0 && (fstat(fd, &sb) != 0 || (size_t)sb.st_size < c->map_bytes)) {
        close(fd);
        fd = -1;
      }
      if (fd < 0) std::this_thread::sleep_for(std::chrono::milliseconds(1));
    }
    if (fd < 0) return ncclSystemError;
  }
  void* p = mmap(nullptr, c->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) return ncclSystemError;
  c->h = static_cast<Header*>(p);
  c->arena = static_cast<char*>(p) + ((sizeof(Header) + 4095) & ~(size_t)4095);
  if (rank == 0) {
    c->h->n_ranks = (uint32_t)n_ranks;
    c->h->arena_bytes = arena;
    c->h->arrived.store(0);
    c->h->phase.store(0);
    c->h->failed.store(0);
    c->h->ready.store(1);
  } else {
    for (int tries = 0; tries < 120000 && !c->h->ready.load(); ++tries)
      std::this_thread::sleep_for(std::chrono::milliseconds(1));
    if (!c->h->ready.load()) return ncclSystemError;
  }
  if (!barrier(c)) return ncclSystemError;
  if (rank == 0) shm_unlink(c->name);  // (every rank has it mapped: the name can go)
  *comm = reinterpret_cast<ncclComm_t>(c);
  g_last_comm = c;
  return ncclSuccess;
}

ncclResult_t ncclCommCount(const ncclComm_t comm, int* count) {
  const Comm* c = reinterpret_cast<const Comm*>(comm);
  if (!c || !count) return ncclInvalidArgument;
  *count = c->n_ranks;
  return ncclSuccess;
}

ncclResult_t ncclCommUserRank(const ncclComm_t comm, int* rank) {
  const Comm* c = reinterpret_cast<const Comm*>(comm);
  if (!c || !rank) return ncclInvalidArgument;
  *rank = c->rank;
  return ncclSuccess;
}

ncclResult_t ncclGetVersion(int* version) {
  if (!version) return ncclInvalidArgument;
  *version = 0;   // (the stand-in, not a release of RCCL)
  return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
  Comm* c = reinterpret_cast<Comm*>(comm);
  if (!c) return ncclSuccess;
  if (g_last_comm == c) g_last_comm = nullptr;
  if (c->h) munmap(c->h, c->map_bytes);
  delete c;
  return ncclSuccess;
}

ncclResult_t ncclAllGather(const void* sendbuff, void* recvbuff, size_t sendcount, ncclDataType_t datatype,
                           ncclComm_t comm, hipStream_t stream) {
  Comm* c = reinterpret_cast<Comm*>(comm);
  const size_t bytes = sendcount * type_size(datatype);
  if (bytes * c->n_ranks > c->h->arena_bytes) return ncclInvalidUsage;
  if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
  // rank r's contribution lives at offset r * bytes of rank 0's arena
  if (hipMemcpy(arena_of(c, 0) + (size_t)c->rank * bytes, sendbuff, bytes, hipMemcpyDeviceToHost) != hipSuccess)
    return ncclUnhandledCudaError;
  if (!barrier(c)) return ncclSystemError;
  if (hipMemcpy(recvbuff, arena_of(c, 0), bytes * c->n_ranks, hipMemcpyHostToDevice) != hipSuccess)
    return ncclUnhandledCudaError;
  if (!barrier(c)) return ncclSystemError;
  return ncclSuccess;
}

ncclResult_t ncclAllReduce(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t datatype,
                           ncclRedOp_t op, ncclComm_t comm, hipStream_t stream) {
  Comm* c = reinterpret_cast<Comm*>(comm);
  if (datatype != ncclInt64 || (op != ncclSum && op != ncclMax)) return ncclInvalidArgument;  // what route.hip uses
  const size_t bytes = count * 8;
  if (bytes * c->n_ranks > c->h->arena_bytes) return ncclInvalidUsage;
  if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
  if (hipMemcpy(arena_of(c, 0) + (size_t)c->rank * bytes, sendbuff, bytes, hipMemcpyDeviceToHost) != hipSuccess)
    return ncclUnhandledCudaError;
  if (!barrier(c)) return ncclSystemError;
  std::vector<int64_t> acc(count);
  const int64_t* all = reinterpret_cast<const int64_t*>(arena_of(c, 0));
  for (size_t i = 0; i < count; ++i) {
    int64_t v = all[i];
    for (int r = 1; r < c->n_ranks; ++r) {
      const int64_t w = all[(size_t)r * count + i];
      v = op == ncclSum ? v + w : (w > v ? w : v);
    }
    acc[i] = v;
  }
  if (!barrier(c)) return ncclSystemError;  // (everybody has read the contributions)
  if (hipMemcpy(recvbuff, acc.data(), bytes, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
  return ncclSuccess;
}

ncclResult_t ncclGroupStart() {
  ++g_group_depth;
  return ncclSuccess;
}

ncclResult_t ncclGroupEnd() {
  if (g_group_depth <= 0) return ncclInvalidUsage;
  if (--g_group_depth > 0) return ncclSuccess;
  std::vector<P2P> ops;
  ops.swap(g_ops);
  return run_group(ops);
}

ncclResult_t ncclSend(const void* sendbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm,
                      hipStream_t stream) {
  Comm* c = reinterpret_cast<Comm*>(comm);
  if (peer < 0 || peer >= c->n_ranks) return ncclInvalidArgument;
  g_ops.push_back(P2P{true, const_cast<void*>(sendbuff), count * type_size(datatype), peer, c, stream});
  if (g_group_depth > 0) return ncclSuccess;
  std::vector<P2P> ops;
  ops.swap(g_ops);
  return run_group(ops);
}

ncclResult_t ncclRecv(void* recvbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm,
                      hipStream_t stream) {
  Comm* c = reinterpret_cast<Comm*>(comm);
  if (peer < 0 || peer >= c->n_ranks) return ncclInvalidArgument;
  g_ops.push_back(P2P{false, recvbuff, count * type_size(datatype), peer, c, stream});
  if (g_group_depth > 0) return ncclSuccess;
  std::vector<P2P> ops;
  ops.swap(g_ops);
  return run_group(ops);
}

const char* ncclGetErrorString(ncclResult_t r) {
  switch (r) {
    case ncclSuccess: return "success (rccl stub)";
    case ncclSystemError: return "rccl stub: a rank did not arrive (time-out) or the segment could not be set up";
    case ncclInvalidUsage: return "rccl stub: unmatched send / receive or a message larger than the arena";
    case ncclInvalidArgument: return "rccl stub: invalid argument";
    default: return "rccl stub: HIP error";
  }
}

}  // extern "C"
