// comm.hip -- the data-parallel exchange step of the videovec path: ONE fp32 all-reduce (sum) of the flat [dW | db]
// gradient buffer per iteration, between the backward pass and the update.  The reference has nothing
// to mirror (single device, src/caffe/common.cpp:127-145); this is the north star's "RCCL all-reduce of the projection
// gradients over xGMI ... on a second HIP stream".
//
//   VV_COMM_RCCL   librccl, one process per GPU.  The library is dlopen'ed (the copy a host framework such as PyTorch has
//                  already loaded is reused, so the process keeps ONE collective runtime); rank 0 writes the
//                  ncclUniqueId to id_path (+ ".tmp" and rename), the other ranks wait for the file.
//   VV_COMM_SHM    a host staged all-reduce through a POSIX shared-memory object named after id_path: every rank copies
//                  its buffer out, all ranks add the world's buffers in rank order (bit-identical results on every
//                  rank), copy back.  For tests of the N > 1 path on a box with ONE device (RCCL refuses two ranks on
//                  one device); never the benchmark's transport.
//   VV_COMM_PEER   one-shot DIRECT exchange over peer mappings (hipIpc): every rank's gradient / parameter buffers are mapped
//                  into every other rank, a reduce-scatter is ONE kernel in which rank r reads shard r of all N buffers
//                  (N - 1 of them over xGMI, all seven links at once) and adds them in rank order, an all-gather ONE
//                  kernel that pulls the N - 1 foreign shards; ranks meet at flag words in host-coherent shared memory
//                  (a one-wave kernel signals and polls: the host is not in the loop).  Two hops of 1/N of the bytes
//                  each instead of the ring's 2 (N - 1) steps; bit for bit the VV_COMM_SHM sums.  Works between
//                  processes on ONE device too (the mappings are then local), which is how it is tested here.
//
// The collective runs on a communication stream owned by the context; events join it with the compute stream, the
// host never blocks (RCCL and PEER transports).
#include <cerrno>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <atomic>
#include <vector>
#include <algorithm>

#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include "vv_comm.h"

namespace vv {

// ---- the handful of RCCL entry points this path needs (rccl.h: ncclGetUniqueId :187, ncclCommInitRank :220,
// ncclAllReduce, ncclCommDestroy, ncclGetErrorString); declared here so that building the library does not need the
// RCCL headers and loading it does not need librccl unless a communicator is created
struct NcclUniqueId { char internal[128]; };
typedef void* NcclComm;
typedef int (*fn_GetUniqueId)(NcclUniqueId*);
typedef int (*fn_CommInitRank)(NcclComm*, int, NcclUniqueId, int);
typedef int (*fn_AllReduce)(const void*, void*, size_t, int /*dtype*/, int /*op*/, NcclComm, hipStream_t);
typedef int (*fn_ReduceScatter)(const void*, void*, size_t, int /*dtype*/, int /*op*/, NcclComm, hipStream_t);   // rccl.h: ncclReduceScatter
typedef int (*fn_AllGather)(const void*, void*, size_t, int /*dtype*/, NcclComm, hipStream_t);                   // ncclAllGather
typedef int (*fn_Group)(void);                                                                                  // ncclGroupStart / ncclGroupEnd
typedef int (*fn_CommDestroy)(NcclComm);
typedef const char* (*fn_GetErrorString)(int);
enum { kNcclFloat32 = 7, kNcclInt8 = 0, kNcclSum = 0 };

// Who wrote a rendezvous object: the writer's pid and its start time (field 22 of /proc/<pid>/stat).  Rank 0 stays
// inside comm_create until every rank has joined, so a reader accepts an id file / a shared-memory header only while
// its writer is ALIVE: what a crashed or finished earlier run left behind under the same name is ignored (its writer
// is gone, or the pid now belongs to a process with another start time).  All ranks of a communicator share one node
// (one /proc); where /proc cannot be read the check degrades to "accept".
static uint64_t proc_starttime(int pid) {
  char path[64];
  snprintf(path, sizeof(path), "/proc/%d/stat", pid);
  FILE* f = fopen(path, "r");
  if (!f) return 0;
  char buf[1024];
  const size_t n = fread(buf, 1, sizeof(buf) - 1, f);
  fclose(f);
  buf[n] = 0;
  const char* p = strrchr(buf, ')');           // the command name may hold spaces and parentheses
  if (!p) return 0;
  int field = 2;
  for (++p; *p; ++p)
    if (*p == ' ' && ++field == 22) return strtoull(p + 1, nullptr, 10);
  return 0;
}
static bool proc_readable() { return proc_starttime((int)getpid()) != 0; }
static bool writer_alive(int pid, uint64_t start) {
  if (!proc_readable()) return true;
  return pid > 0 && start != 0 && proc_starttime(pid) == start;
}

struct IdFile {
  uint64_t magic;                               // "VVRCID01"
  int32_t pid; int32_t pad_;
  uint64_t start;
  char id[128];
};
static constexpr uint64_t kIdMagic = 0x5656524349443031ull;

struct ShmHdr {
  uint64_t magic;
  int32_t world; int32_t pid;                   // pid, start: the creating rank 0 (see proc_starttime)
  uint64_t start;
  uint64_t n_floats;
  alignas(64) std::atomic<int64_t> arrive;     // monotonic arrival counter (barrier generations of `world` arrivals)
};
static constexpr uint64_t kShmMagic = 0x5656434f4d4d3031ull;   // "VVCOMM01"

static constexpr int kPeerMax = 16;               // ranks of a direct exchange (one node)
static constexpr int kPeerFlagStride = 16;        // words between two ranks' flags (64 bytes)
struct PeerReg { unsigned char* base = nullptr; size_t size = 0; unsigned char* peer[kPeerMax] = {}; };
struct PeerSlot { hipIpcMemHandle_t h; uint64_t size; };          // one rank's entry of the handle table in the shared object
static_assert(sizeof(PeerSlot) <= 256, "handle table slot");

struct Comm {
  int world = 1, rank = 0, transport = VV_COMM_RCCL;
  hipStream_t stream = nullptr;                 // communication stream
  hipStream_t cur = nullptr;                    // where the collectives are queued: the communication stream, or the caller's (comm_use_stream)
  hipEvent_t ev_ready = nullptr, ev_done = nullptr;
  // RCCL
  void* dl = nullptr; NcclComm nccl = nullptr;
  fn_AllReduce AllReduce = nullptr; fn_CommDestroy CommDestroy = nullptr; fn_GetErrorString ErrStr = nullptr;
  fn_ReduceScatter ReduceScatter = nullptr; fn_AllGather AllGather = nullptr; fn_Group GroupStart = nullptr, GroupEnd = nullptr;
  // shared-memory stub
  ShmHdr* shm = nullptr; size_t shm_bytes = 0; std::string shm_name; float* stage = nullptr; size_t stage_floats = 0;
  int64_t barriers = 0;
  // direct peer exchange
  std::vector<PeerReg> regs;                     // buffers mapped so far (the allocation that holds them, in every rank)
  uint32_t* flags_host = nullptr; uint32_t* flags_dev = nullptr;   // world words, kPeerFlagStride apart (shared, host coherent)
  uint32_t* status_host = nullptr; uint32_t* status_dev = nullptr; // != 0: a wait gave up (a rank is missing)
  uint32_t* xcd_count = nullptr;                 // device word: the meeting kernel's workgroups 1..7 count themselves in
  uint32_t seq = 0;                              // meeting points so far (every rank counts the same ones)
  uint32_t* fail_dev = nullptr;     // device word: != 0 once a meeting gave up -- the exchange kernels queued behind it return at once, the update is skipped
  double peer_timeout_s = 120.0;
  std::string err;
};

static size_t shm_hdr_bytes() { return 4096; }

static bool shm_barrier(Comm* c, double timeout_s) {
  const int64_t target = (++c->barriers) * c->world;
  c->shm->arrive.fetch_add(1, std::memory_order_acq_rel);
  const auto t0 = std::chrono::steady_clock::now();
  unsigned spins = 0;
  while (c->shm->arrive.load(std::memory_order_acquire) < target) {
    if (++spins > 1000) std::this_thread::sleep_for(std::chrono::microseconds(50));
    if ((spins & 4095) == 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s) return false;
  }
  return true;
}

const char* comm_error(Comm* c) { return c ? c->err.c_str() : "no communicator"; }
const uint32_t* comm_fail_flag(Comm* c) { return c ? c->fail_dev : nullptr; }
bool comm_failed(Comm* c) {
  if (!c || !c->status_host || !*(volatile uint32_t*)c->status_host) return false;
  c->err = "a rank did not reach the exchange in time (direct peer transport)";
  return true;
}
int comm_world(Comm* c) { return c ? c->world : 1; }
int comm_rank(Comm* c) { return c ? c->rank : 0; }

void comm_destroy(Comm* c) {
  if (!c) return;
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->nccl && c->CommDestroy) c->CommDestroy(c->nccl);
  if (c->transport == VV_COMM_PEER && c->shm) {
    // nobody unmaps or frees what a peer may still be reading: the ranks meet once more (briefly: a rank that died does not hold the others)
    if (c->flags_host && c->world > 1) (void)shm_barrier(c, 5.0);
    for (auto& r : c->regs)
      for (int k = 0; k < c->world; ++k)
        if (k != c->rank && r.peer[k]) (void)hipIpcCloseMemHandle(r.peer[k]);
    c->regs.clear();
    if (c->flags_host) (void)hipHostUnregister(c->flags_host);
    if (c->status_host) (void)hipHostFree(c->status_host);
    if (c->fail_dev) (void)hipFree(c->fail_dev);
    if (c->xcd_count) (void)hipFree(c->xcd_count);
  }
  if (c->shm) { munmap((void*)c->shm, c->shm_bytes); if (c->rank == 0) shm_unlink(c->shm_name.c_str()); }
  if (c->stage) (void)hipHostFree(c->stage);
  if (c->ev_ready) (void)hipEventDestroy(c->ev_ready);
  if (c->ev_done) (void)hipEventDestroy(c->ev_done);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  // the dlopen handle is kept: unloading a collective runtime with live device state is not safe
  delete c;
}

static std::string shm_name_of(const char* id_path) {
  std::string n = "/vvcomm_";
  for (const char* p = id_path; *p; ++p) n += (*p == '/' || *p == '.') ? '_' : *p;
  if (n.size() > 200) n = "/vvcomm_" + n.substr(n.size() - 180);
  return n;
}

Comm* comm_create(int world, int rank, const char* id_path, int transport, size_t n_floats, std::string* err) {
  Comm* c = new Comm();
  c->world = world; c->rank = rank; c->transport = transport;
  auto fail = [&](const std::string& m) { if (err) *err = m; comm_destroy(c); return (Comm*)nullptr; };
  // the communication stream gets the highest priority: its short kernels (the chunks of the overlapped update) run beside
  // the compute stream's large ones and must not queue behind them
  int prio_lo = 0, prio_hi = 0;
  (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
  if (hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, prio_hi) != hipSuccess) return fail("hipStreamCreate failed");
  c->cur = c->stream;
  if (hipEventCreateWithFlags(&c->ev_ready, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&c->ev_done, hipEventDisableTiming) != hipSuccess) return fail("hipEventCreate failed");
  const double timeout_s = getenv("VV_COMM_TIMEOUT") ? atof(getenv("VV_COMM_TIMEOUT")) : 120.0;
  if (transport == VV_COMM_RCCL) {
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", nullptr};
    for (int i = 0; names[i] && !c->dl; ++i) c->dl = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
    if (!c->dl) { const char* e = dlerror(); return fail(std::string("cannot load librccl: ") + (e ? e : "?")); }
    auto GetUniqueId = (fn_GetUniqueId)dlsym(c->dl, "ncclGetUniqueId");
    auto CommInitRank = (fn_CommInitRank)dlsym(c->dl, "ncclCommInitRank");
    c->AllReduce = (fn_AllReduce)dlsym(c->dl, "ncclAllReduce");
    c->CommDestroy = (fn_CommDestroy)dlsym(c->dl, "ncclCommDestroy");
    c->ErrStr = (fn_GetErrorString)dlsym(c->dl, "ncclGetErrorString");
    c->ReduceScatter = (fn_ReduceScatter)dlsym(c->dl, "ncclReduceScatter");       // (the sharded schedule's; their absence is reported when it is asked for)
    c->AllGather = (fn_AllGather)dlsym(c->dl, "ncclAllGather");
    c->GroupStart = (fn_Group)dlsym(c->dl, "ncclGroupStart");
    c->GroupEnd = (fn_Group)dlsym(c->dl, "ncclGroupEnd");
    if (!GetUniqueId || !CommInitRank || !c->AllReduce || !c->CommDestroy) return fail("librccl lacks the expected entry points");
    NcclUniqueId id;
    memset(&id, 0, sizeof(id));
    if (rank == 0) {
      const int rc = GetUniqueId(&id);
      if (rc != 0) return fail(std::string("ncclGetUniqueId: ") + (c->ErrStr ? c->ErrStr(rc) : "error"));
      if (world > 1) {
        IdFile rec;
        memset(&rec, 0, sizeof(rec));
        rec.magic = kIdMagic; rec.pid = (int32_t)getpid(); rec.start = proc_starttime(rec.pid);
        memcpy(rec.id, &id, sizeof(id));
        (void)unlink(id_path);                   // whatever an earlier run left under this name
        const std::string tmp = std::string(id_path) + ".tmp";
        FILE* f = fopen(tmp.c_str(), "wb");
        if (!f || fwrite(&rec, sizeof(rec), 1, f) != 1) { if (f) fclose(f); return fail("cannot write " + tmp); }
        fclose(f);
        if (rename(tmp.c_str(), id_path) != 0) return fail(std::string("cannot rename to ") + id_path);
      }
    } else {
      const auto t0 = std::chrono::steady_clock::now();
      for (;;) {
        IdFile rec;
        FILE* f = fopen(id_path, "rb");
        if (f) {
          const size_t n = fread(&rec, sizeof(rec), 1, f);
          fclose(f);
          // a file whose writer is not alive is a leftover: keep waiting for this launch's rank 0 to replace it
          if (n == 1 && rec.magic == kIdMagic && writer_alive(rec.pid, rec.start)) { memcpy(&id, rec.id, sizeof(id)); break; }
        }
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s)
          return fail(std::string("timed out waiting for the communicator id file ") + id_path);
        std::this_thread::sleep_for(std::chrono::milliseconds(5));
      }
    }
    const int rc = CommInitRank(&c->nccl, world, id, rank);
    if (rank == 0 && world > 1) (void)unlink(id_path);      // every rank has joined (or the attempt failed): the file has done its job
    if (rc != 0) return fail(std::string("ncclCommInitRank: ") + (c->ErrStr ? c->ErrStr(rc) : "error"));
    // One small collective now, waited for: the library sets up its channels and connections lazily at the first call, which
    // can take seconds -- not something to happen under a forward GEMM that is waiting at its gates (api.hip, overlap).
    {
      float* warm = nullptr;
      if (hipMalloc((void**)&warm, 4096) != hipSuccess || hipMemsetAsync(warm, 0, 4096, c->stream) != hipSuccess) return fail("hipMalloc failed");
      const int rw = c->AllReduce(warm, warm, 1024, kNcclFloat32, kNcclSum, c->nccl, c->stream);
      const hipError_t es = hipStreamSynchronize(c->stream);
      (void)hipFree(warm);
      if (rw != 0 || es != hipSuccess) return fail(std::string("the communicator's first all-reduce failed: ") + (rw != 0 && c->ErrStr ? c->ErrStr(rw) : hipGetErrorString(es)));
    }
  } else if (transport == VV_COMM_SHM || transport == VV_COMM_PEER) {
    if (transport == VV_COMM_PEER && world > kPeerMax) return fail("the direct peer exchange takes at most 16 ranks");
    c->shm_name = shm_name_of(id_path);
    // VV_COMM_SHM: the header and one gradient-sized slab per rank.  VV_COMM_PEER: the header, a page of flag words, a page of handle slots.
    c->shm_bytes = transport == VV_COMM_SHM ? shm_hdr_bytes() + (size_t)world * n_floats * sizeof(float) : shm_hdr_bytes() + 2 * 4096;
    int fd = -1;
    if (rank == 0) {
      shm_unlink(c->shm_name.c_str());
      fd = shm_open(c->shm_name.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
      if (fd < 0 || ftruncate(fd, (off_t)c->shm_bytes) != 0) { if (fd >= 0) close(fd); return fail("cannot create the shared-memory object " + c->shm_name); }
    } else {
      const auto t0 = std::chrono::steady_clock::now();
      for (;;) {
        fd = shm_open(c->shm_name.c_str(), O_RDWR, 0600);
        struct stat st;
        if (fd >= 0 && fstat(fd, &st) == 0 && (size_t)st.st_size >= c->shm_bytes) break;
        if (fd >= 0) { close(fd); fd = -1; }
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s) return fail("timed out waiting for " + c->shm_name);
        std::this_thread::sleep_for(std::chrono::milliseconds(5));
      }
    }
    void* mem = mmap(nullptr, c->shm_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (mem == MAP_FAILED) return fail("mmap of " + c->shm_name + " failed");
    c->shm = (ShmHdr*)mem;
    if (rank == 0) {
      c->shm->world = world; c->shm->n_floats = n_floats; c->shm->arrive.store(0);
      c->shm->pid = (int32_t)getpid(); c->shm->start = proc_starttime((int)getpid());
      std::atomic_thread_fence(std::memory_order_release);
      c->shm->magic = kShmMagic;
    } else {
      // accept the object only once its header is complete AND its creator is alive; an object of an earlier run (opened
      // before this launch's rank 0 unlinked and re-created the name) is dropped and the name opened again
      const auto t0 = std::chrono::steady_clock::now();
      for (;;) {
        // (checked on every turn, whatever branch it takes: a stale object that stays in place because this launch's rank 0 never
        // arrives must end in the time-out, not in a hot loop of re-opens)
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s) return fail("timed out waiting for " + c->shm_name + " to be initialised by a live rank 0");
        volatile ShmHdr* h = (volatile ShmHdr*)c->shm;
        if (h->magic == kShmMagic) {
          std::atomic_thread_fence(std::memory_order_acquire);
          if (writer_alive(h->pid, h->start)) break;
          std::this_thread::sleep_for(std::chrono::milliseconds(2));
          munmap((void*)c->shm, c->shm_bytes); c->shm = nullptr;
          for (;;) {
            const int fd2 = shm_open(c->shm_name.c_str(), O_RDWR, 0600);
            struct stat st;
            if (fd2 >= 0 && fstat(fd2, &st) == 0 && (size_t)st.st_size >= c->shm_bytes) {
              void* m2 = mmap(nullptr, c->shm_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd2, 0);
              close(fd2);
              if (m2 != MAP_FAILED) { c->shm = (ShmHdr*)m2; break; }
            } else if (fd2 >= 0) close(fd2);
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s) return fail("timed out waiting for " + c->shm_name);
            std::this_thread::sleep_for(std::chrono::milliseconds(5));
          }
          continue;
        }
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s) return fail("shared-memory object never initialised");
        std::this_thread::sleep_for(std::chrono::milliseconds(1));
      }
      if (c->shm->world != world || c->shm->n_floats != n_floats) return fail("shared-memory object belongs to a different job shape");
    }
    if (transport == VV_COMM_SHM) {
      c->stage_floats = n_floats;
      if (hipHostMalloc((void**)&c->stage, n_floats * sizeof(float), hipHostMallocDefault) != hipSuccess) return fail("hipHostMalloc failed");
    } else {
      // the flag page, seen by this rank's device: system-scope atomics of the meeting kernel go to host memory every rank maps
      c->flags_host = (uint32_t*)((unsigned char*)c->shm + shm_hdr_bytes());
      if (rank == 0) memset(c->flags_host, 0, 4096);
      if (hipHostRegister(c->flags_host, 4096, hipHostRegisterMapped | hipHostRegisterPortable) != hipSuccess) { c->flags_host = nullptr; return fail("hipHostRegister of the flag page failed"); }
      if (hipHostGetDevicePointer((void**)&c->flags_dev, c->flags_host, 0) != hipSuccess) return fail("hipHostGetDevicePointer failed");
      if (hipHostMalloc((void**)&c->status_host, 64, hipHostMallocMapped) != hipSuccess) return fail("hipHostMalloc failed");
      c->status_host[0] = 0;
      if (hipHostGetDevicePointer((void**)&c->status_dev, c->status_host, 0) != hipSuccess) return fail("hipHostGetDevicePointer failed");
      if (hipMalloc((void**)&c->xcd_count, 64) != hipSuccess || hipMemset(c->xcd_count, 0, 64) != hipSuccess) return fail("hipMalloc failed");
      if (hipMalloc((void**)&c->fail_dev, 64) != hipSuccess || hipMemset(c->fail_dev, 0, 64) != hipSuccess) return fail("hipMalloc failed");
      // (the same default as every other wait of the transports: a rank held up by a long host pause -- first-use kernel load, snapshot
      // I/O -- must not make its peers give up first)
      c->peer_timeout_s = getenv("VV_COMM_TIMEOUT") ? atof(getenv("VV_COMM_TIMEOUT")) : 120.0;
    }
    if (!shm_barrier(c, timeout_s)) return fail("ranks did not all arrive");
    if (rank == 0) shm_unlink(c->shm_name.c_str());     // every rank holds its mapping: the name has done its job (nothing is left behind, crash or not)
  } else {
    return fail("unknown transport");
  }
  return c;
}


// ---- VV_COMM_PEER: the meeting point, the one-shot reduce and the one-shot gather ---------------------------------------------
// Meeting point `seq`, EIGHT one-wave workgroups (one per XCD: workgroup -> XCD is blockIdx mod 8).  Every one of them first runs a
// system-scope release fence -- a write-back of ITS XCD's L2 -- so that whatever the stream's earlier kernels wrote is at the memory side
// (where a peer's read over xGMI is served) without relying on the scope the runtime gives a kernel's end-of-kernel release; workgroup 0
// waits for the other seven (a device counter), raises this rank's flag to seq (release, system scope), and lane r waits for rank r's
// flag.  One wave polls: it occupies nothing a peer process on the same device needs in order to get there.  A wait that outlasts
// `timeout_ticks` (100 MHz) gives up and sets *status (host-visible: every later entry point fails) and *fail_dev (device memory: the
// reduce / gather kernels and the update queued behind this meeting on the stream return at once instead of working on a peer's
// incomplete data).  Neither word is ever cleared: a failed exchange is FATAL for the context -- the ranks' parameters can no longer be
// assumed equal; the job restarts from a snapshot.
__global__ void __launch_bounds__(64) k_peer_meet(uint32_t* flags, uint32_t* xcd_count, int world, int rank, uint32_t seq, unsigned long long timeout_ticks, uint32_t* status, uint32_t* fail_dev) {
  const int t = threadIdx.x;
  __atomic_thread_fence(__ATOMIC_RELEASE);                       // (system scope: buffer_wbl2 sc0 sc1)
  if (blockIdx.x != 0) {
    if (t == 0) __hip_atomic_fetch_add(xcd_count, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  const unsigned long long t0 = wall_clock64();
  if (t == 0) {
    const uint32_t want = (gridDim.x - 1) * seq;                 // seq counts the meetings: every one adds gridDim.x - 1
    unsigned spins = 0;
    while ((int32_t)(__hip_atomic_load(xcd_count, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) - want) < 0) {
      __builtin_amdgcn_s_sleep(2);
      if ((++spins & 255u) == 0 && wall_clock64() - t0 > timeout_ticks) { __hip_atomic_store(fail_dev, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __hip_atomic_store(status, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
    }
    __hip_atomic_store(flags + (size_t)rank * kPeerFlagStride, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  if (t < world && t != rank) {
    unsigned spins = 0;
    while ((int32_t)(__hip_atomic_load(flags + (size_t)t * kPeerFlagStride, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - seq) < 0) {
      __builtin_amdgcn_s_sleep(8);
      if ((++spins & 255u) == 0 && wall_clock64() - t0 > timeout_ticks) { __hip_atomic_store(fail_dev, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __hip_atomic_store(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
    }
  }
}

// What a peer wrote is read past this device's caches: system-scope loads (sc0 sc1), 8 bytes per lane -- a line of a peer's buffer
// that an earlier step left in this device's L2 is never served again, whatever scope the kernel's start-of-kernel acquire had.
__device__ __forceinline__ float2 ld_peer8(const float* p) {
  const unsigned long long v = __hip_atomic_load((const unsigned long long*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  return make_float2(__uint_as_float((unsigned)v), __uint_as_float((unsigned)(v >> 32)));
}
__device__ __forceinline__ float ld_peer4(const float* p) {
  return __uint_as_float(__hip_atomic_load((const unsigned*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
}

struct PeerSrc { const float* p[kPeerMax]; };
// out[i] = ((src0[i] + src1[i]) + src2[i]) + ...   -- rank order, the order of the shared-memory transport's sums
template <int WORLD>      // 0: any world (run-time loop)
__global__ void __launch_bounds__(256) k_peer_reduce(PeerSrc src, float* __restrict__ out, size_t n, int world_rt, int vec, const uint32_t* fail_dev) {
  if (__hip_atomic_load(fail_dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;      // the meeting in front gave up: a peer's data is not there
  const int world = WORLD ? WORLD : world_rt;
  const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x, nth = (size_t)gridDim.x * 256;
  const size_t n2 = vec ? n / 2 : 0;
  for (size_t i = tid; i < n2; i += nth) {
    float2 v[WORLD ? WORLD : 1];
    if (WORLD) {
#pragma unroll
      for (int r = 0; r < WORLD; ++r) v[r] = ld_peer8(src.p[r] + 2 * i);       // all the peers' loads in flight together
      float2 a = v[0];
#pragma unroll
      for (int r = 1; r < WORLD; ++r) { a.x += v[r].x; a.y += v[r].y; }
      ((float2*)out)[i] = a;
    } else {
      float2 a = ld_peer8(src.p[0] + 2 * i);
      for (int r = 1; r < world; ++r) { const float2 b = ld_peer8(src.p[r] + 2 * i); a.x += b.x; a.y += b.y; }
      ((float2*)out)[i] = a;
    }
  }
  for (size_t i = n2 * 2 + tid; i < n; i += nth) {
    float a = ld_peer4(src.p[0] + i);
    for (int r = 1; r < world; ++r) a += ld_peer4(src.p[r] + i);
    out[i] = a;
  }
}

constexpr int kGatherBufs = 4;
struct GatherArgs {
  int n, world, rank, blocks_per_piece;
  const uint32_t* fail_dev;
  unsigned char* dst[kGatherBufs];                       // this rank's buffers
  const unsigned char* src[kGatherBufs][kPeerMax];       // the same buffers in every rank (peer mappings)
  size_t stride[kGatherBufs], total[kGatherBufs];        // rank r owns bytes [r stride, min((r + 1) stride, total))
};
// piece (buffer i, peer r) = blockIdx.y: the peer's own bytes of buffer i pulled into this rank's copy
__global__ void __launch_bounds__(256) k_peer_gather(GatherArgs g) {
  if (__hip_atomic_load(g.fail_dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;    // the meeting in front gave up
  const int i = blockIdx.y / g.world, r = blockIdx.y % g.world;
  if (r == g.rank) return;
  const size_t lo = (size_t)r * g.stride[i];
  if (lo >= g.total[i]) return;
  const size_t nb = (g.total[i] - lo < g.stride[i]) ? g.total[i] - lo : g.stride[i];
  const unsigned char* s = g.src[i][r] + lo;
  unsigned char* d = g.dst[i] + lo;
  const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x, nth = (size_t)gridDim.x * 256;
  size_t body = 0;
  if ((((uintptr_t)s | (uintptr_t)d) & 7) == 0) {
    body = nb / 8;
    size_t k = tid;
    for (; k + 3 * nth < body; k += 4 * nth) {           // four loads in flight per lane
      unsigned long long v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = __hip_atomic_load((const unsigned long long*)s + k + j * nth, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#pragma unroll
      for (int j = 0; j < 4; ++j) ((unsigned long long*)d)[k + j * nth] = v[j];
    }
    for (; k < body; k += nth) ((unsigned long long*)d)[k] = __hip_atomic_load((const unsigned long long*)s + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    body *= 8;
  }
  for (size_t k = body + tid; k < nb; k += nth)
    d[k] = (unsigned char)(__hip_atomic_load((const unsigned*)((uintptr_t)(s + k) & ~(uintptr_t)3), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) >> (8 * ((uintptr_t)(s + k) & 3)));
}

// The allocation that holds `ptr`, mapped in every rank (found, or exchanged now: a COLLECTIVE the first time a buffer is used --
// every rank runs the same sequence of collectives on the same buffers, so every rank gets here together).
static PeerReg* peer_reg(Comm* c, const void* ptr) {
  for (auto& r : c->regs)
    if ((const unsigned char*)ptr >= r.base && (const unsigned char*)ptr < r.base + r.size) return &r;
  PeerReg reg;
  void* base = nullptr; size_t size = 0;
  if (hipMemGetAddressRange((hipDeviceptr_t*)&base, &size, (hipDeviceptr_t)ptr) != hipSuccess) { c->err = "hipMemGetAddressRange failed (the buffer is not a device allocation)"; return nullptr; }
  reg.base = (unsigned char*)base; reg.size = size; reg.peer[c->rank] = reg.base;
  if (c->world > 1) {
    PeerSlot* table = (PeerSlot*)((unsigned char*)c->shm + shm_hdr_bytes() + 4096);
    PeerSlot* mine = (PeerSlot*)((unsigned char*)table + (size_t)c->rank * 256);
    if (hipIpcGetMemHandle(&mine->h, base) != hipSuccess) { c->err = "hipIpcGetMemHandle failed"; return nullptr; }
    mine->size = size;
    const double timeout_s = getenv("VV_COMM_TIMEOUT") ? atof(getenv("VV_COMM_TIMEOUT")) : 120.0;
    if (!shm_barrier(c, timeout_s)) { c->err = "mapping a buffer into the peers: a rank is missing"; return nullptr; }
    for (int k = 0; k < c->world; ++k) {
      if (k == c->rank) continue;
      const PeerSlot* theirs = (const PeerSlot*)((const unsigned char*)table + (size_t)k * 256);
      if (theirs->size != size) { c->err = "a peer's buffer has another size"; return nullptr; }
      void* p = nullptr;
      const hipError_t e = hipIpcOpenMemHandle(&p, theirs->h, hipIpcMemLazyEnablePeerAccess);
      if (e != hipSuccess) { c->err = std::string("hipIpcOpenMemHandle: ") + hipGetErrorString(e); return nullptr; }
      reg.peer[k] = (unsigned char*)p;
    }
    if (!shm_barrier(c, timeout_s)) { c->err = "mapping a buffer into the peers: a rank is missing"; return nullptr; }   // the table is free again
  }
  c->regs.push_back(reg);
  return &c->regs.back();
}

static int peer_meet(Comm* c) {
  if (c->status_host && *(volatile uint32_t*)c->status_host) { c->err = "a rank did not reach the exchange in time (direct peer transport)"; return -1; }
  if (c->world == 1) return 0;
  ++c->seq;
  k_peer_meet<<<8, 64, 0, c->cur>>>(c->flags_dev, c->xcd_count, c->world, c->rank, c->seq, (unsigned long long)(c->peer_timeout_s * 1e8), c->status_dev, c->fail_dev);
  return hipGetLastError() == hipSuccess ? 0 : (c->err = "launch of the meeting kernel failed", -1);
}

// rank r's range [lo, lo + n) of `buf` (floats) <- the sum over the ranks of that range, in rank order
static int peer_reduce_range(Comm* c, float* buf, size_t lo, size_t n) {
  PeerReg* reg = peer_reg(c, buf);
  if (!reg) return -1;
  if (n == 0 || c->world == 1) return 0;
  const size_t off = (unsigned char*)(buf + lo) - reg->base;
  PeerSrc src;
  for (int k = 0; k < c->world; ++k) src.p[k] = (const float*)(reg->peer[k] + off);
  const int vec = (off & 7) == 0;
  const size_t work = vec ? (n + 1) / 2 : n;
  const int blocks = (int)std::min<size_t>(1024, std::max<size_t>(1, (work + 255) / 256));
  switch (c->world) {
    case 2: k_peer_reduce<2><<<blocks, 256, 0, c->cur>>>(src, buf + lo, n, c->world, vec, c->fail_dev); break;
    case 4: k_peer_reduce<4><<<blocks, 256, 0, c->cur>>>(src, buf + lo, n, c->world, vec, c->fail_dev); break;
    case 8: k_peer_reduce<8><<<blocks, 256, 0, c->cur>>>(src, buf + lo, n, c->world, vec, c->fail_dev); break;
    default: k_peer_reduce<0><<<blocks, 256, 0, c->cur>>>(src, buf + lo, n, c->world, vec, c->fail_dev); break;
  }
  return hipGetLastError() == hipSuccess ? 0 : (c->err = "launch of the reduce kernel failed", -1);
}

// buffers i < n: every rank's own bytes [r stride_i, ...) of `total_i` pulled from their owners
static int peer_gather(Comm* c, void* const* bufs, const size_t* stride, const size_t* total, int n) {
  if (n > kGatherBufs) { c->err = "too many buffers in one all-gather"; return -1; }
  GatherArgs g;
  memset(&g, 0, sizeof(g));
  g.n = n; g.world = c->world; g.rank = c->rank; g.fail_dev = c->fail_dev;
  size_t largest = 0;
  for (int i = 0; i < n; ++i) {
    PeerReg* reg = peer_reg(c, bufs[i]);
    if (!reg) return -1;
    const size_t off = (unsigned char*)bufs[i] - reg->base;
    g.dst[i] = (unsigned char*)bufs[i];
    for (int k = 0; k < c->world; ++k) g.src[i][k] = reg->peer[k] + off;
    g.stride[i] = stride[i]; g.total[i] = total[i];
    largest = std::max(largest, stride[i]);
  }
  if (c->world == 1 || largest == 0) return 0;
  const int bx = (int)std::min<size_t>(64, std::max<size_t>(1, (largest / 32 + 255) / 256));
  k_peer_gather<<<dim3(bx, n * c->world), 256, 0, c->cur>>>(g);
  return hipGetLastError() == hipSuccess ? 0 : (c->err = "launch of the gather kernel failed", -1);
}

// all-reduce(sum) of buf[0 .. n) in place.  `after`: the collective starts once this event (recorded on the compute
// stream) has completed; on return ev_done (comm_done_event) marks its end on the communication stream.
// The collective queued directly on the caller's stream (RCCL only): no second stream, no events.  For the synchronous
// schedule, where nothing runs beside the all-reduce anyway, this saves the two stream joins (~6 us of stream time each).
// Returns 1 when the transport cannot do that (the caller then uses comm_allreduce), 0 on success, -1 on error.
static int peer_allreduce(Comm* c, float* buf, size_t off, size_t n);
int comm_allreduce_inline(Comm* c, float* buf, size_t n, hipStream_t stream) {
  if (c->transport == VV_COMM_PEER) {           // the direct exchange's kernels queued on the caller's stream
    hipStream_t keep = c->cur;
    c->cur = stream;
    const int rc = peer_allreduce(c, buf, 0, n);
    c->cur = keep;
    return rc;
  }
  if (c->transport != VV_COMM_RCCL) return 1;
  const int rc = c->AllReduce(buf, buf, n, kNcclFloat32, kNcclSum, c->nccl, stream);
  if (rc != 0) { c->err = std::string("ncclAllReduce: ") + (c->ErrStr ? c->ErrStr(rc) : "error"); return -1; }
  return 0;
}

// VV_COMM_PEER: reduce-scatter + all-gather of the range, both direct: rank r sums slice r (a multiple of four floats; the last one may
// be short), the ranks meet, every rank pulls the other slices; a third meeting keeps a fast rank's NEXT gradients out of a buffer a slow
// rank is still pulling from
static int peer_allreduce(Comm* c, float* buf, size_t off, size_t n) {
  if (!peer_reg(c, buf)) return -1;
  const size_t per = ((n + c->world - 1) / c->world + 3) / 4 * 4;
  const size_t lo = std::min(n, (size_t)c->rank * per), hi = std::min(n, lo + per);
  void* bufs[1] = {buf + off};
  const size_t stride[1] = {per * sizeof(float)}, total[1] = {n * sizeof(float)};
  if (peer_meet(c) || peer_reduce_range(c, buf, off + lo, hi - lo) || peer_meet(c) || peer_gather(c, bufs, stride, total, 1) || peer_meet(c)) return -1;
  return 0;
}

int comm_allreduce(Comm* c, float* buf, size_t off, size_t n, hipEvent_t after) {
  if (after && hipStreamWaitEvent(c->cur, after, 0) != hipSuccess) { c->err = "hipStreamWaitEvent failed"; return -1; }
  if (c->transport == VV_COMM_RCCL) {
    const int rc = c->AllReduce(buf + off, buf + off, n, kNcclFloat32, kNcclSum, c->nccl, c->cur);
    if (rc != 0) { c->err = std::string("ncclAllReduce: ") + (c->ErrStr ? c->ErrStr(rc) : "error"); return -1; }
  } else if (c->transport == VV_COMM_PEER) {
    if (peer_allreduce(c, buf, off, n)) return -1;
  } else {
    const double timeout_s = getenv("VV_COMM_TIMEOUT") ? atof(getenv("VV_COMM_TIMEOUT")) : 120.0;
    float* slabs = (float*)((unsigned char*)c->shm + shm_hdr_bytes());
    float* mine = slabs + (size_t)c->rank * c->stage_floats + off;
    if (hipMemcpyAsync(c->stage + off, buf + off, n * sizeof(float), hipMemcpyDeviceToHost, c->cur) != hipSuccess ||
        hipStreamSynchronize(c->cur) != hipSuccess) { c->err = "device-to-host copy failed"; return -1; }
    memcpy(mine, c->stage + off, n * sizeof(float));
    if (!shm_barrier(c, timeout_s)) { c->err = "all-reduce barrier timed out (a rank is missing)"; return -1; }
    float* out = c->stage + off;
    for (size_t i = 0; i < n; ++i) out[i] = slabs[off + i];                                   // rank 0 first, then in rank order
    for (int r = 1; r < c->world; ++r) {
      const float* s = slabs + (size_t)r * c->stage_floats + off;
      for (size_t i = 0; i < n; ++i) out[i] += s[i];
    }
    if (!shm_barrier(c, timeout_s)) { c->err = "all-reduce barrier timed out (a rank is missing)"; return -1; }
    if (hipMemcpyAsync(buf + off, out, n * sizeof(float), hipMemcpyHostToDevice, c->cur) != hipSuccess) { c->err = "host-to-device copy failed"; return -1; }
  }
  if (hipEventRecord(c->ev_done, c->cur) != hipSuccess) { c->err = "hipEventRecord failed"; return -1; }
  return 0;
}

// ---- the sharded update's two collectives
int comm_reduce_scatter(Comm* c, float* buf, size_t shard, hipEvent_t after) {
  if (after && hipStreamWaitEvent(c->cur, after, 0) != hipSuccess) { c->err = "hipStreamWaitEvent failed"; return -1; }
  if (c->transport == VV_COMM_RCCL) {
    if (!c->ReduceScatter) { c->err = "librccl has no ncclReduceScatter"; return -1; }
    const int rc = c->ReduceScatter(buf, buf + (size_t)c->rank * shard, shard, kNcclFloat32, kNcclSum, c->nccl, c->cur);   // in place: recv = send + rank * count
    if (rc != 0) { c->err = std::string("ncclReduceScatter: ") + (c->ErrStr ? c->ErrStr(rc) : "error"); return -1; }
    return 0;
  }
  if (c->transport == VV_COMM_PEER) {
    // every rank's buffer is complete (meeting), then ONE kernel: shard `rank` of all the buffers, summed in rank order, into this
    // rank's.  Nobody writes what a peer reads: rank r's shard of rank k's buffer is read by r alone and written by nobody but k's NEXT
    // backward pass -- which follows k's next forward pass, which follows the all-gather every rank enters only after this kernel.
    if (!peer_reg(c, buf)) return -1;
    if (peer_meet(c) || peer_reduce_range(c, buf, (size_t)c->rank * shard, shard)) return -1;
    return 0;
  }
  // shared-memory stand-in: every rank publishes its whole buffer, then sums ITS shard over the ranks in rank order -- the
  // order of comm_allreduce's sums, so the shard is bit for bit what the all-reduce would have left there
  const double timeout_s = getenv("VV_COMM_TIMEOUT") ? atof(getenv("VV_COMM_TIMEOUT")) : 120.0;
  const size_t n = shard * c->world;
  if (n > c->stage_floats) { c->err = "reduce-scatter larger than the communicator's buffer"; return -1; }
  float* slabs = (float*)((unsigned char*)c->shm + shm_hdr_bytes());
  if (hipMemcpyAsync(c->stage, buf, n * sizeof(float), hipMemcpyDeviceToHost, c->cur) != hipSuccess ||
      hipStreamSynchronize(c->cur) != hipSuccess) { c->err = "device-to-host copy failed"; return -1; }
  memcpy(slabs + (size_t)c->rank * c->stage_floats, c->stage, n * sizeof(float));
  if (!shm_barrier(c, timeout_s)) { c->err = "reduce-scatter barrier timed out (a rank is missing)"; return -1; }
  const size_t off = (size_t)c->rank * shard;
  float* out = c->stage + off;
  for (size_t i = 0; i < shard; ++i) out[i] = slabs[off + i];
  for (int r = 1; r < c->world; ++r) {
    const float* s = slabs + (size_t)r * c->stage_floats + off;
    for (size_t i = 0; i < shard; ++i) out[i] += s[i];
  }
  if (!shm_barrier(c, timeout_s)) { c->err = "reduce-scatter barrier timed out (a rank is missing)"; return -1; }
  if (hipMemcpyAsync(buf + off, out, shard * sizeof(float), hipMemcpyHostToDevice, c->cur) != hipSuccess) { c->err = "host-to-device copy failed"; return -1; }
  return 0;
}

int comm_allgather(Comm* c, void* const* bufs, const size_t* shard_bytes, int n, int fence_after) {
  if (c->transport == VV_COMM_RCCL) {
    if (!c->AllGather) { c->err = "librccl has no ncclAllGather"; return -1; }
    if (n > 1 && c->GroupStart) c->GroupStart();
    int rc = 0;
    for (int i = 0; i < n && rc == 0; ++i) {
      unsigned char* b = (unsigned char*)bufs[i];
      rc = c->AllGather(b + (size_t)c->rank * shard_bytes[i], b, shard_bytes[i], kNcclInt8, c->nccl, c->cur);            // in place: send = recv + rank * count
    }
    if (n > 1 && c->GroupEnd) { const int rg = c->GroupEnd(); if (rc == 0) rc = rg; }
    if (rc != 0) { c->err = std::string("ncclAllGather: ") + (c->ErrStr ? c->ErrStr(rc) : "error"); return -1; }
    return 0;
  }
  if (c->transport == VV_COMM_PEER) {
    // every rank's shard is final (meeting), then ONE kernel pulls the foreign shards of all n buffers.  `fence_after`: a closing
    // meeting, for a gather that no later collective orders against the owners' next writes.
    size_t total[kGatherBufs];
    if (n > kGatherBufs) { c->err = "too many buffers in one all-gather"; return -1; }
    for (int i = 0; i < n; ++i) { total[i] = shard_bytes[i] * c->world; if (!peer_reg(c, bufs[i])) return -1; }
    if (peer_meet(c) || peer_gather(c, bufs, shard_bytes, total, n)) return -1;
    if (fence_after && peer_meet(c)) return -1;
    return 0;
  }
  const double timeout_s = getenv("VV_COMM_TIMEOUT") ? atof(getenv("VV_COMM_TIMEOUT")) : 120.0;
  unsigned char* slabs = (unsigned char*)c->shm + shm_hdr_bytes();
  const size_t slab_bytes = c->stage_floats * sizeof(float);
  for (int i = 0; i < n; ++i) {
    const size_t sb = shard_bytes[i];
    if (sb > slab_bytes) { c->err = "all-gather shard larger than the communicator's buffer"; return -1; }
    unsigned char* b = (unsigned char*)bufs[i];
    if (hipMemcpyAsync(c->stage, b + (size_t)c->rank * sb, sb, hipMemcpyDeviceToHost, c->cur) != hipSuccess ||
        hipStreamSynchronize(c->cur) != hipSuccess) { c->err = "device-to-host copy failed"; return -1; }
    memcpy(slabs + (size_t)c->rank * slab_bytes, c->stage, sb);
    if (!shm_barrier(c, timeout_s)) { c->err = "all-gather barrier timed out (a rank is missing)"; return -1; }
    for (int r = 0; r < c->world; ++r) {
      if (r == c->rank) continue;
      // (pageable source: the copy is staged by the runtime before the call returns for sizes like these; the barrier below keeps the
      // slab intact until every rank has issued its copies and synchronised)
      if (hipMemcpyAsync(b + (size_t)r * sb, slabs + (size_t)r * slab_bytes, sb, hipMemcpyHostToDevice, c->cur) != hipSuccess) { c->err = "host-to-device copy failed"; return -1; }
    }
    if (hipStreamSynchronize(c->cur) != hipSuccess) { c->err = "host-to-device copy failed"; return -1; }
    if (!shm_barrier(c, timeout_s)) { c->err = "all-gather barrier timed out (a rank is missing)"; return -1; }
  }
  return 0;
}

hipEvent_t comm_done_event(Comm* c) { return c->ev_done; }
hipStream_t comm_stream(Comm* c) { return c->stream; }
void comm_use_stream(Comm* c, hipStream_t s) { c->cur = s ? s : c->stream; }
int comm_record_done(Comm* c) {
  if (hipEventRecord(c->ev_done, c->stream) != hipSuccess) { c->err = "hipEventRecord failed"; return -1; }
  return 0;
}

}  // namespace vv
