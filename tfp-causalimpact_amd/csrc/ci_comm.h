// ci_comm.h -- chain gather / diagnostics collectives behind the C-ABI (ci_comm_*), no PyTorch.
//
// Included at the end of ci_api.hip (it needs ci_session / ci_ll_session, fail(), DevBuf).
// The reference has no communication at all (one chain, one process); chains
// are independent, so the fit itself never communicates.  After the fit:
//   * all-gather of per-chain result blocks that are still RESIDENT IN HBM (pooled summaries),
//   * all-reduce of the diagnostics' partial sums (split-R-hat / ESS: a few hundred doubles).
// Two transports behind the same entry points:
//   CI_COMM_RCCL  librccl (dlopen'ed on first use: a single-GPU fit never loads it): one rank per
//                 GPU, ncclAllGather / ncclAllReduce on device buffers over xGMI;
//   CI_COMM_HOST  a POSIX shared-memory segment on one node: for ranks that SHARE a device (RCCL
//                 refuses two ranks on one GPU: "Duplicate GPU detected") and for GPU-less tests of
//                 the launcher; device-resident blocks are staged through the host.
#pragma once
#include <dlfcn.h>
#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <atomic>

namespace {

// ---- the slice of the RCCL API this file uses (rccl.h: ncclUniqueId is 128 opaque bytes) ----
struct RcclId { char internal[CI_COMM_ID_BYTES]; };
typedef void* RcclComm;
enum { RCCL_SUM = 0, RCCL_MAX = 2 };
enum { RCCL_INT8 = 0, RCCL_FLOAT32 = 7, RCCL_FLOAT64 = 8 };
struct RcclApi {
  void* handle = nullptr;
  int (*GetUniqueId)(RcclId*) = nullptr;
  int (*CommInitRank)(RcclComm*, int, RcclId, int) = nullptr;
  int (*CommDestroy)(RcclComm) = nullptr;
  int (*CommAbort)(RcclComm) = nullptr;
  int (*CommCount)(RcclComm, int*) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, RcclComm, hipStream_t) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, RcclComm, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  const char* (*GetLastError)(RcclComm) = nullptr;
};
RcclApi g_rccl;
std::mutex g_rccl_mu;

int rccl_load() {
  std::lock_guard<std::mutex> lk(g_rccl_mu);
  if (g_rccl.handle) return 0;
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void* h = nullptr;
  for (const char* n : names) {
    h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
    if (h) break;
  }
  if (!h) return fail("cannot load librccl: %s", dlerror());
  RcclApi a;
  a.handle = h;
#define CI_SYM(field, name)                                                  \
  *(void**)(&a.field) = dlsym(h, name);                                      \
  if (!a.field) { dlclose(h); return fail("librccl has no %s", name); }
  CI_SYM(GetUniqueId, "ncclGetUniqueId")
  CI_SYM(CommInitRank, "ncclCommInitRank")
  CI_SYM(CommDestroy, "ncclCommDestroy")
  CI_SYM(CommCount, "ncclCommCount")
  CI_SYM(AllReduce, "ncclAllReduce")
  CI_SYM(AllGather, "ncclAllGather")
  CI_SYM(GetErrorString, "ncclGetErrorString")
#undef CI_SYM
  *(void**)(&a.GetLastError) = dlsym(h, "ncclGetLastError");   // optional
  *(void**)(&a.CommAbort) = dlsym(h, "ncclCommAbort");         // optional
  g_rccl = a;
  return 0;
}

#define RCCL_TRY(comm, expr)                                                            \
  do {                                                                                  \
    const int r_ = (expr);                                                              \
    if (r_ != 0) {                                                                      \
      const char* last_ = g_rccl.GetLastError ? g_rccl.GetLastError(comm) : "";         \
      return fail("%s failed: %s %s", #expr, g_rccl.GetErrorString(r_), last_ ? last_ : ""); \
    }                                                                                   \
  } while (0)

// Every collective is bounded: a rank that died (or an RCCL ring that never forms) must surface as
// an error on the survivors, never as a hang.  $CI_COMM_TIMEOUT_S (default 300 s) bounds one
// collective on either transport.
double comm_timeout_s() {
  const char* e = getenv("CI_COMM_TIMEOUT_S");
  const double v = e ? atof(e) : 0.0;
  return v > 0.0 ? v : 300.0;
}
double mono_seconds() {
  timespec t;
  clock_gettime(CLOCK_MONOTONIC, &t);
  return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}
// hipStreamSynchronize with a deadline (hipStreamQuery polling): an RCCL collective whose peers
// never arrive leaves its kernel spinning on the stream for ever.  The caller queues NOTHING that
// touches caller-owned host memory behind the collective before this returns 0 (a
// device-to-host copy into pageable memory is host-synchronous in HIP -- queued behind a stuck
// collective it would block inside hipMemcpyAsync, before this loop is ever reached, and after a
// time-out the stream would still hold a copy into memory the caller may have freed): results are
// copied out AFTER the collective has completed, when nothing on the stream can wait for a peer.
int stream_wait_raw(hipStream_t st, const char* what, double limit) {
  const double t0 = mono_seconds();
  for (unsigned spin = 0;; ++spin) {
    const hipError_t q = hipStreamQuery(st);
    if (q == hipSuccess) return 0;
    if (q != hipErrorNotReady) return fail("%s: %s", what, hipGetErrorString(q));
    if (spin < 2000) { sched_yield(); continue; }
    if (mono_seconds() - t0 > limit)
      return fail("ci_comm (rccl transport): %s did not complete within %.0f s (a rank died or the "
                  "ring never formed)", what, limit);
    usleep(100);
  }
}

// ---- host transport: one shared-memory segment per communicator ------------------------------
constexpr size_t HOST_SLOT_BYTES = (size_t)4 << 20;       // staging area per rank
struct HostHeader {
  std::atomic<int> ready;        // 1 once rank 0 has initialised the header
  std::atomic<int> attached;     // ranks that have mapped the segment
  std::atomic<int> arrived;      // barrier: arrivals of the current generation
  std::atomic<int> generation;
  int world;
  int pad[11];
};
static_assert(sizeof(HostHeader) == 64, "header is one cache line");

}  // namespace

struct ci_comm {
  int transport = CI_COMM_RCCL, rank = 0, world = 1, device = 0, ranks_seen = 0;
  // RCCL
  RcclComm nc = nullptr;
  hipStream_t stream = nullptr;
  bool dead = false;             // a collective timed out: the communicator was aborted
  double timeout_s = 0.0;        // bound of one collective (ci_comm_set_timeout; 0: $CI_COMM_TIMEOUT_S / 300 s)
  DevBuf<unsigned char> send, recv;
  // host
  HostHeader* hdr = nullptr;
  unsigned char* slots = nullptr;
  size_t map_bytes = 0;
  char shm_name[80] = {0};
};

namespace {

// Bounded wait for the collective queued on c->stream.  On a time-out the communicator is ABORTED
// (ncclCommAbort makes the spinning kernel exit; without it the kernel would keep a CU busy and the
// stream could never be destroyed), the stream is given a few seconds to drain, and the comm is
// marked dead: every later collective on it fails at once instead of queueing behind the wreck.
double comm_limit(const ci_comm* c) { return c->timeout_s > 0.0 ? c->timeout_s : comm_timeout_s(); }
int stream_wait(ci_comm* c, const char* what) {
  if (stream_wait_raw(c->stream, what, comm_limit(c)) == 0) return 0;
  const std::string keep = g_err;
  c->dead = true;
  if (c->nc && g_rccl.CommAbort) {
    (void)g_rccl.CommAbort(c->nc);
    c->nc = nullptr;
    (void)stream_wait_raw(c->stream, "drain after ncclCommAbort", 5.0);
  }
  g_err = keep + (g_rccl.CommAbort ? "; communicator aborted" : "; librccl has no ncclCommAbort");
  return 1;
}
int comm_alive(const ci_comm* c) {
  if (c->dead) return fail("ci_comm (rccl transport): the communicator was aborted after a time-out");
  return 0;
}

void host_name(const uint8_t* id, char* out, size_t n) {
  static const char* hx = "0123456789abcdef";
  size_t o = (size_t)snprintf(out, n, "/ci_comm_");
  for (int i = 0; i < 16 && o + 2 < n; ++i) { out[o++] = hx[id[i] >> 4]; out[o++] = hx[id[i] & 15]; }
  out[o] = 0;
}

int host_wait(const std::atomic<int>& v, int want_at_least, double timeout_s, const char* what) {
  timespec t0;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  for (unsigned spin = 0;; ++spin) {
    if (v.load(std::memory_order_acquire) >= want_at_least) return 0;
    if ((spin & 63) == 63) {
      timespec t1;
      clock_gettime(CLOCK_MONOTONIC, &t1);
      if ((double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec) > timeout_s)
        return fail("ci_comm (host transport): timed out waiting for %s", what);
      usleep(50);
    } else {
      sched_yield();
    }
  }
}

// Sense-reversing barrier on the shared header.
int host_barrier(ci_comm* c) {
  HostHeader* h = c->hdr;
  const int gen = h->generation.load(std::memory_order_acquire);
  if (h->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == c->world) {
    h->arrived.store(0, std::memory_order_relaxed);
    h->generation.fetch_add(1, std::memory_order_release);
    return 0;
  }
  timespec t0;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  for (unsigned spin = 0;; ++spin) {
    if (h->generation.load(std::memory_order_acquire) != gen) return 0;
    if ((spin & 63) == 63) {
      timespec t1;
      clock_gettime(CLOCK_MONOTONIC, &t1);
      if ((double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec) > comm_limit(c))
        return fail("ci_comm (host transport): barrier timed out after %.1f s (a rank died?)",
                    comm_limit(c));
      usleep(50);
    } else {
      sched_yield();
    }
  }
}

int host_attach(ci_comm* c, const uint8_t* id) {
  host_name(id, c->shm_name, sizeof(c->shm_name));
  c->map_bytes = sizeof(HostHeader) + (size_t)c->world * HOST_SLOT_BYTES;
  int fd = -1;
  if (c->rank == 0) {
    fd = shm_open(c->shm_name, O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd < 0) return fail("shm_open(%s) failed: %s", c->shm_name, strerror(errno));
    if (ftruncate(fd, (off_t)c->map_bytes) != 0) {
      close(fd); shm_unlink(c->shm_name);
      return fail("ftruncate(%s, %zu) failed: %s", c->shm_name, c->map_bytes, strerror(errno));
    }
  } else {
    timespec t0;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (;;) {
      fd = shm_open(c->shm_name, O_RDWR, 0600);
      if (fd >= 0) {
        struct stat st;
        if (fstat(fd, &st) == 0 && (size_t)st.st_size >= c->map_bytes) break;
        close(fd);
        fd = -1;
      }
      timespec t1;
      clock_gettime(CLOCK_MONOTONIC, &t1);
      if ((double)(t1.tv_sec - t0.tv_sec) > 120.0)
        return fail("ci_comm (host transport): rank 0 never created %s", c->shm_name);
      usleep(200);
    }
  }
  void* p = mmap(nullptr, c->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) {
    if (c->rank == 0) shm_unlink(c->shm_name);
    return fail("mmap(%s) failed: %s", c->shm_name, strerror(errno));
  }
  c->hdr = (HostHeader*)p;
  c->slots = (unsigned char*)p + sizeof(HostHeader);
  if (c->rank == 0) {
    c->hdr->attached.store(0);
    c->hdr->arrived.store(0);
    c->hdr->generation.store(0);
    c->hdr->world = c->world;
    c->hdr->ready.store(1, std::memory_order_release);
  } else {
    if (host_wait(c->hdr->ready, 1, 120.0, "rank 0's header")) return 1;
    if (c->hdr->world != c->world)
      return fail("ci_comm: world size mismatch (%d here, %d at rank 0)", c->world, c->hdr->world);
  }
  c->hdr->attached.fetch_add(1, std::memory_order_acq_rel);
  if (host_wait(c->hdr->attached, c->world, 120.0, "all ranks to attach")) return 1;
  c->ranks_seen = c->hdr->attached.load();
  // everybody has the mapping: the name can go (the memory lives until the last munmap)
  if (c->rank == 0) shm_unlink(c->shm_name);
  return 0;
}

// All-gather of `bytes` per rank through the slots, in chunks of HOST_SLOT_BYTES.
int host_all_gather(ci_comm* c, const unsigned char* send, unsigned char* recv, size_t bytes) {
  for (size_t off = 0; off < bytes || off == 0; off += HOST_SLOT_BYTES) {
    const size_t n = std::min(HOST_SLOT_BYTES, bytes - off);
    if (n) memcpy(c->slots + (size_t)c->rank * HOST_SLOT_BYTES, send + off, n);
    if (host_barrier(c)) return 1;
    for (int r = 0; r < c->world; ++r)
      if (n) memcpy(recv + (size_t)r * bytes + off, c->slots + (size_t)r * HOST_SLOT_BYTES, n);
    if (host_barrier(c)) return 1;
    if (bytes == 0) break;
  }
  return 0;
}

int host_all_reduce(ci_comm* c, double* v, size_t n, int op) {
  const size_t per = HOST_SLOT_BYTES / sizeof(double);
  for (size_t off = 0; off < n || off == 0; off += per) {
    const size_t m = std::min(per, n - off);
    if (m) memcpy(c->slots + (size_t)c->rank * HOST_SLOT_BYTES, v + off, m * sizeof(double));
    if (host_barrier(c)) return 1;
    for (size_t i = 0; i < m; ++i) {        // rank order: every rank forms the same bits
      double acc = ((const double*)c->slots)[i];
      for (int r = 1; r < c->world; ++r) {
        const double x = ((const double*)(c->slots + (size_t)r * HOST_SLOT_BYTES))[i];
        acc = op == CI_COMM_MAX ? (x > acc ? x : acc) : acc + x;
      }
      v[off + i] = acc;
    }
    if (host_barrier(c)) return 1;
    if (n == 0) break;
  }
  return 0;
}

int comm_scratch(ci_comm* c, size_t send_bytes, size_t recv_bytes) {
  if (c->send.n < send_bytes) { c->send.release(); HIP_TRY(c->send.alloc(send_bytes)); }
  if (c->recv.n < recv_bytes) { c->recv.release(); HIP_TRY(c->recv.alloc(recv_bytes)); }
  return 0;
}

// All-gather of a device-resident block of `count` floats from every rank into host `recv`
// [world, count].
int gather_device_floats(ci_comm* c, const float* dev, size_t count, float* recv, int device) {
  if (count == 0) return 0;      // (the host side hands in a zero-initialised array)
  if (!dev || !recv) return fail("ci_comm: nothing resident to gather / recv is NULL");
  HIP_TRY(hipSetDevice(device));
  const size_t bytes = count * sizeof(float);
  if (c->transport == CI_COMM_RCCL) {
    if (device != c->device) return fail("ci_comm: session is on device %d, communicator on %d", device, c->device);
    if (comm_alive(c) || comm_scratch(c, 0, bytes * c->world)) return 1;
    RCCL_TRY(c->nc, g_rccl.AllGather(dev, c->recv.p, count, RCCL_FLOAT32, c->nc, c->stream));
    if (stream_wait(c, "ncclAllGather of a resident result array")) return 1;
    HIP_TRY(hipMemcpy(recv, c->recv.p, bytes * c->world, hipMemcpyDeviceToHost));   // collective done
    return 0;
  }
  std::vector<unsigned char> mine(bytes);
  HIP_TRY(hipMemcpy(mine.data(), dev, bytes, hipMemcpyDeviceToHost));
  return host_all_gather(c, mine.data(), (unsigned char*)recv, bytes);
}

}  // namespace

extern "C" {

int ci_comm_unique_id(int32_t transport, uint8_t* id) {
  if (!id) return fail("id is NULL");
  memset(id, 0, CI_COMM_ID_BYTES);
  if (transport == CI_COMM_RCCL) {
    if (rccl_load()) return 1;
    RcclId u;
    RCCL_TRY(nullptr, g_rccl.GetUniqueId(&u));
    memcpy(id, u.internal, CI_COMM_ID_BYTES);
    return 0;
  }
  if (transport != CI_COMM_HOST) return fail("unknown transport %d", transport);
  const int fd = open("/dev/urandom", O_RDONLY);
  if (fd < 0 || read(fd, id, 16) != 16) {
    if (fd >= 0) close(fd);
    return fail("cannot read /dev/urandom");
  }
  close(fd);
  return 0;
}

int ci_comm_destroy(ci_comm* c);

int ci_comm_create(int32_t transport, const uint8_t* id, int32_t rank, int32_t world, int32_t device,
                   ci_comm** out) {
  if (!id || !out) return fail("NULL argument");
  if (world < 1 || rank < 0 || rank >= world) return fail("need 0 <= rank < world, got %d / %d", rank, world);
  ci_comm* c = new ci_comm();
  c->transport = transport; c->rank = rank; c->world = world; c->device = device;
  int rc = 0;
  if (transport == CI_COMM_RCCL) {
    rc = rccl_load();
    if (!rc) {
      auto init = [&]() -> int {
        HIP_TRY(hipSetDevice(device));
        HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
        RcclId u;
        memcpy(u.internal, id, CI_COMM_ID_BYTES);
        RCCL_TRY(nullptr, g_rccl.CommInitRank(&c->nc, world, u, rank));
        RCCL_TRY(c->nc, g_rccl.CommCount(c->nc, &c->ranks_seen));
        return 0;
      };
      rc = init();
    }
  } else if (transport == CI_COMM_HOST) {
    rc = host_attach(c, id);
  } else {
    rc = fail("unknown transport %d", transport);
  }
  if (rc) {
    const std::string keep = g_err;
    ci_comm_destroy(c);
    g_err = keep;
    return 1;
  }
  *out = c;
  return 0;
}

int ci_comm_info(const ci_comm* c, int32_t* rank, int32_t* world, int32_t* ranks_seen) {
  if (!c) return fail("comm is NULL");
  if (rank) *rank = c->rank;
  if (world) *world = c->world;
  if (ranks_seen) *ranks_seen = c->ranks_seen;
  return 0;
}

int ci_comm_set_timeout(ci_comm* c, double seconds) {
  if (!c) return fail("comm is NULL");
  c->timeout_s = seconds > 0.0 ? seconds : 0.0;
  return 0;
}

int ci_comm_all_reduce(ci_comm* c, double* values, int64_t n, int32_t op) {
  if (!c || (n > 0 && !values)) return fail("NULL argument");
  if (n < 0) return fail("n must be >= 0");
  if (op != CI_COMM_SUM && op != CI_COMM_MAX) return fail("op must be CI_COMM_SUM or CI_COMM_MAX");
  if (c->transport == CI_COMM_HOST) return host_all_reduce(c, values, (size_t)n, op);
  if (n == 0) return 0;
  HIP_TRY(hipSetDevice(c->device));
  const size_t bytes = (size_t)n * sizeof(double);
  if (comm_alive(c) || comm_scratch(c, bytes, bytes)) return 1;
  // the upload completes before anything can wait for a peer (nothing is queued ahead of it)
  HIP_TRY(hipMemcpyAsync(c->send.p, values, bytes, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  RCCL_TRY(c->nc, g_rccl.AllReduce(c->send.p, c->recv.p, (size_t)n, RCCL_FLOAT64,
                                   op == CI_COMM_MAX ? RCCL_MAX : RCCL_SUM, c->nc, c->stream));
  if (stream_wait(c, "ncclAllReduce")) return 1;
  HIP_TRY(hipMemcpy(values, c->recv.p, bytes, hipMemcpyDeviceToHost));               // collective done
  return 0;
}

int ci_comm_barrier(ci_comm* c) {
  if (!c) return fail("comm is NULL");
  if (c->transport == CI_COMM_HOST) return host_barrier(c);
  double one = 1.0;
  return ci_comm_all_reduce(c, &one, 1, CI_COMM_SUM);
}

int ci_comm_all_gather(ci_comm* c, const void* send, void* recv, int64_t bytes) {
  if (!c || (bytes > 0 && (!send || !recv))) return fail("NULL argument");
  if (bytes < 0) return fail("bytes must be >= 0");
  if (c->transport == CI_COMM_HOST)
    return host_all_gather(c, (const unsigned char*)send, (unsigned char*)recv, (size_t)bytes);
  if (bytes == 0) return 0;
  HIP_TRY(hipSetDevice(c->device));
  if (comm_alive(c) || comm_scratch(c, (size_t)bytes, (size_t)bytes * c->world)) return 1;
  HIP_TRY(hipMemcpyAsync(c->send.p, send, (size_t)bytes, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  RCCL_TRY(c->nc, g_rccl.AllGather(c->send.p, c->recv.p, (size_t)bytes, RCCL_INT8, c->nc, c->stream));
  if (stream_wait(c, "ncclAllGather")) return 1;
  HIP_TRY(hipMemcpy(recv, c->recv.p, (size_t)bytes * c->world, hipMemcpyDeviceToHost));   // collective done
  return 0;
}

int ci_comm_session_all_gather(ci_comm* c, ci_session* s, int32_t field, float* recv) {
  if (!c || !s) return fail("NULL argument");
  if (!s->ran) return fail("ci_comm_session_all_gather needs a finished ci_session_run");
  const DevBuf<float>* b = nullptr;
  switch (field) {
    case CI_FIELD_OBSERVATION_NOISE_SCALE: b = &s->o_obs; break;
    case CI_FIELD_LEVEL_SCALE: b = &s->o_lscale; break;
    case CI_FIELD_SLOPE_SCALE: b = &s->o_sscale; break;
    case CI_FIELD_SEASONAL_DRIFT_SCALES: b = &s->o_drift; break;
    case CI_FIELD_WEIGHTS: b = &s->o_w; break;
    case CI_FIELD_LEVEL: b = &s->o_level; break;
    case CI_FIELD_SLOPE: b = &s->o_slope; break;
    case CI_FIELD_SEASONAL_LEVELS: b = &s->o_seasonal; break;
    case CI_FIELD_POSTERIOR_MEANS: b = &s->o_pm; break;
    case CI_FIELD_POSTERIOR_TRAJECTORIES: b = &s->o_traj; break;
    default: return fail("unknown field %d", field);
  }
  return gather_device_floats(c, b->p, b->n, recv, s->pb.device);
}

int ci_comm_ll_session_all_gather(ci_comm* c, ci_ll_session* s, int32_t field, float* recv) {
  if (!c || !s) return fail("NULL argument");
  if (!s->h_ran) return fail("ci_comm_ll_session_all_gather needs a finished ci_ll_session_hmc_run");
  const DevBuf<float>* b = nullptr;
  switch (field) {
    case CI_FIELD_OBSERVATION_NOISE_SCALE: b = &s->h_obs; break;
    case CI_FIELD_LEVEL_SCALE: b = &s->h_lscale; break;
    case CI_FIELD_SLOPE_SCALE: b = &s->h_sscale; break;
    case CI_FIELD_SEASONAL_DRIFT_SCALES: b = &s->h_drift; break;
    case CI_FIELD_SEASONAL_LEVELS: b = &s->h_seasonal; break;
    case CI_FIELD_WEIGHTS: b = &s->h_w; break;
    case CI_FIELD_LEVEL: b = &s->h_level; break;
    case CI_FIELD_SLOPE: b = &s->h_slope; break;
    case CI_FIELD_POSTERIOR_MEANS: b = &s->h_pm; break;
    case CI_FIELD_POSTERIOR_TRAJECTORIES: b = &s->h_traj; break;
    default: return fail("field %d is not part of an HMC fit", field);
  }
  return gather_device_floats(c, b->p, b->n, recv, s->device);
}

int ci_comm_destroy(ci_comm* c) {
  if (!c) return 0;
  if (c->transport == CI_COMM_RCCL) {
    (void)hipSetDevice(c->device);
    //  a communicator that timed out and could NOT be aborted (librccl without
    // ncclCommAbort) still has its kernel spinning on c->stream: ncclCommDestroy, the buffer frees
    // and hipStreamDestroy all synchronise with it -- the hang the time-out bounded would move here.
    // Leak the three instead; the process is about to report the failure anyway.
    const bool wedged = c->dead && c->nc != nullptr;
    if (!wedged) {
      if (c->nc && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->nc);
      c->send.release();
      c->recv.release();
      if (c->stream) (void)hipStreamDestroy(c->stream);
    }
  } else if (c->hdr) {
    munmap((void*)c->hdr, c->map_bytes);
  }
  delete c;
  return 0;
}

}  // extern "C"
