// domain.hip -- the two live GraviT schedulers as native loops over device-resident queues:
//
//   Tracer<ImageScheduler>::operator()    src/gvt/render/algorithm/ImageTracer.h:127-269
//   Tracer<DomainScheduler>::operator()   src/gvt/render/algorithm/DomainTracer.h:185-349, SendRays :370-496
//   asynchronous Domain tracer            src/gvt/render/tracer/Domain/DomainTracer.cpp:109-192, vote: core/comm/vote/vote.cpp:47-152
//
// What the reference does one adapter call at a time (pick the fullest queue, trace it, shuffle the moved rays) runs here in ROUNDS:
// all non-empty local queues of a rank go through ONE merged launch chain (closest hit -> shade -> any hit, trace.hip
// wave_trace_chain) followed by ONE shuffle of all moved rays, with ray counts kept in device memory; the host reads the queue
// sizes back ONCE per round.  A ray list's order carries no meaning in the reference and every ray owns its RNG stream, so the image
// is the same (bit for bit where a pixel receives one deposit, within float-add reordering elsewhere).
//
// Under the Domain scheduler the ray exchange (MPI in the reference) is RCCL point-to-point on a dedicated communication stream:
//   per round and peer  1. an "announce" of fixed size: {rays, bytes, this rank's total outgoing rays, its local pending rays,
//                          rays per destination queue} -- SendRays' count exchange (:397-415) and the termination gather/scatter
//                          (:337-349) in one message; built on the device, exchanged in one ncclGroup, read back with the queue sizes
//                          in the round's single host synchronisation;
//                       2. the payload in the reference's wire format, per queue [int32 queueId][int32 nRays][Ray x nRays] with the
//                          80-byte Ray image (:441-455), sent while the next round's local chain runs; unpacked on the compute stream
//                          behind an event.
// Termination: every rank sees every rank's ballot (no local rays, nothing outgoing) in the same exchange and commits when all do --
// the two-phase vote of the asynchronous tracer (propose / collect / commit) folded into the announce, with the guarantee the
// reference's vote lacks: rays in flight are counted, because a round's announce is built after the previous payload was unpacked.
// GVT_HIP_FRAME_BSP reproduces Tracer<DomainScheduler>: trace until the local queues are dry, then exchange.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <atomic>
#include <condition_variable>
#include <memory>
#include <mutex>

#include <chrono>
#include <thread>

#include "gvt_internal.h"

int shuffle_async(gvt_hip_top *T, gvt_hip_queue *q_in, size_t n_ub, const int *from_arr, int from, gvt_hip_queue *const *queues, const uint8_t *keep_mask,
                  gvt_hip_fb *fb, unsigned *d_overflow, const void *d_qdesc);
int shuffle_exact(gvt_hip_top *T, gvt_hip_queue *q_in, const int *from_arr, int from, gvt_hip_queue *const *queues, gvt_hip_fb *fb);
int camera_one_instance_async(gvt_hip_top *T, const gvt_hip_camera *cam, int tile, gvt_hip_queue *q, gvt_hip_fb *fb, unsigned *d_overflow, unsigned *d_moved_count);
int camera_filter_async(gvt_hip_top *T, const gvt_hip_camera *cam, int tile, gvt_hip_queue *const *queues, const uint8_t *keep_mask, unsigned *d_overflow,
                        size_t first, size_t count, bool rect = false);
size_t camera_instance_bound(gvt_hip_top *T, const gvt_hip_camera *cam, int tile, size_t inst);

// ------------------------------------------------------------------------------------------------
// RCCL, resolved at first use (a single-GPU process never loads it)
// ------------------------------------------------------------------------------------------------
namespace {
struct Rccl {
  void *lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Reduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
  ncclResult_t (*CommAbort)(ncclComm_t) = nullptr; // optional
};
Rccl g_rccl;
std::mutex g_rccl_mu;

int rccl_load() {
  std::lock_guard<std::mutex> lk(g_rccl_mu);
  if (g_rccl.lib) return 0;
  void *h = nullptr;
  if (const char *e = getenv("GVT_HIP_RCCL_LIB")) { // a particular build of the library (or, in tests/, a stand-in with the same entry points)
    if (*e && !(h = dlopen(e, RTLD_NOW | RTLD_LOCAL))) { set_error("GVT_HIP_RCCL_LIB=%s: %s", e, dlerror()); return GVT_HIP_ERR_DEVICE; }
  }
  if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL); // the copy already in the process (e.g. PyTorch's) or the system one
  if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
  if (!h) { set_error("RCCL not found: %s", dlerror()); return GVT_HIP_ERR_DEVICE; }
#define GVT_SYM(field, name)                                                                  \
  *(void **)(&g_rccl.field) = dlsym(h, name);                                                 \
  if (!g_rccl.field) { set_error("RCCL symbol %s missing", name); dlclose(h); return GVT_HIP_ERR_DEVICE; }
  GVT_SYM(GetUniqueId, "ncclGetUniqueId") GVT_SYM(CommInitRank, "ncclCommInitRank") GVT_SYM(CommDestroy, "ncclCommDestroy")
  GVT_SYM(GroupStart, "ncclGroupStart") GVT_SYM(GroupEnd, "ncclGroupEnd") GVT_SYM(Send, "ncclSend") GVT_SYM(Recv, "ncclRecv")
  GVT_SYM(Reduce, "ncclReduce") GVT_SYM(GetErrorString, "ncclGetErrorString") GVT_SYM(CommCount, "ncclCommCount")
#undef GVT_SYM
  *(void **)(&g_rccl.CommAbort) = dlsym(h, "ncclCommAbort");
  g_rccl.lib = h;
  return 0;
}
#define NCCLCHK(expr)                                                                                      \
  do {                                                                                                     \
    ncclResult_t _r = (expr);                                                                              \
    if (_r != ncclSuccess) {                                                                               \
      set_error("%s failed: %s (%s:%d)", #expr, g_rccl.GetErrorString(_r), __FILE__, __LINE__);            \
      return GVT_HIP_ERR_DEVICE;                                                                           \
    }                                                                                                      \
  } while (0)

// the in-process transport's receive side: every message a rank takes in one group, copied (or added, float by float) by ONE launch
#define HUB_ITEMS 16
struct HubItem { unsigned *dst; const unsigned *src; unsigned long long words; int accumulate; };
struct HubBatch { int n; HubItem it[HUB_ITEMS]; };
__global__ __launch_bounds__(256) void k_hub_gather(const HubBatch B) {
  const HubItem I = B.it[blockIdx.y];
  for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < I.words; i += (unsigned long long)gridDim.x * blockDim.x) {
    if (I.accumulate) ((float *)I.dst)[i] += ((const float *)I.src)[i];
    else I.dst[i] = I.src[i];
  }
}
} // namespace

// ------------------------------------------------------------------------------------------------
// In-process transport: the ranks are threads of one process (each with its own context) sharing one device.  Same call pattern as
// the RCCL transport -- grouped point-to-point sends / receives ordered on the ranks' communication streams -- with the transfer done
// by device-to-device copies.  Several ranks per node is how the reference is usually run (mpirun -np P); on a one-GPU box it is
// also the only way to run the Domain scheduler's multi-rank control flow on the device.
// ------------------------------------------------------------------------------------------------
struct gvt_hip_hub {
  int world = 0;
  struct Slot {
    const void *ptr = nullptr;
    size_t bytes = 0;
    hipEvent_t ready = nullptr;  // recorded by the sender: the buffer is complete
    hipEvent_t copied = nullptr; // recorded by the receiver: the buffer has been read
    std::atomic<int> state{ 0 }; // 0 empty, 1 posted (ptr / bytes / ready valid), 2 consumed (copied valid)
  };
  std::unique_ptr<Slot[]> slots; // [dst * world + src]; a slot has ONE writer per state: the sender for 0 -> 1 and 2 -> 0, the receiver for 1 -> 2
  // ONE event pair per RANK and group, not per message: a rank records `sent[r]` once when its outgoing buffers are complete and `taken[r]` once
  // behind the one kernel that gathers everything it receives (the slots point at them).  An 8-rank announce was 7 records + 7 copies + 7 records
  // per rank and tick; it is 1 + 1 kernel + 1.
  std::unique_ptr<hipEvent_t[]> sent, taken;
  std::atomic<bool> aborted{ false };
};
namespace {
// The ranks are threads with a core each: a slot's state is awaited with loads (a condition variable costs a futex wake-up and the
// scheduler's latency, 10-50 us, several times per exchange), yielding once the wait gets long; false: deadline passed or hub aborted.
bool hub_await(gvt_hip_hub *H, std::atomic<int> &state, int want, int deadline_ms) {
  const auto t0 = std::chrono::steady_clock::now();
  for (unsigned spins = 0;; spins++) {
    if (state.load(std::memory_order_acquire) == want) return true;
    if (H->aborted.load(std::memory_order_relaxed)) return false;
    if ((spins & 1023u) == 1023u) {
      if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(deadline_ms)) return false;
      if (spins > (1u << 16)) std::this_thread::yield();
    }
  }
}
} // namespace

struct gvt_hip_comm {
  int rank = 0, world = 1;
  hipStream_t stream = nullptr; // communication stream
  ncclComm_t nccl = nullptr;
  gvt_hip_hub *hub = nullptr;
  struct Op { int send; void *ptr; size_t bytes; int peer; int accumulate; };
  std::vector<Op> ops; // the open group
  int count = 1;          // ranks the transport itself reports (ncclCommCount / the hub's world)
  int deadline_ms = 20000; // every blocking point of an exchange gives up after this long (GVT_HIP_ERR_TIMEOUT)
  bool dead = false;      // a deadline passed: the communicator was aborted
  // RCCL connects two ranks at their first send / receive (all pairs in a frame's first announce): that exchange alone may take as long
  // as connect_ms (GVT_HIP_CONNECT_TIMEOUT_MS), whatever the deadline of the steady state
  bool connected = false;
  int connect_ms = 180000;
  hipEvent_t ev_wait = nullptr;
  double ms_host_wait = 0.0; // host time spent in bounded waits since the frame began
  uint64_t groups = 0;       // transport groups issued since the frame began
  int cus = 0;               // compute units reserved for `stream` (knob comm_cus at creation)
};

extern "C" gvt_hip_hub *gvt_hip_hub_create(int world) {
  if (world < 1) { set_error("hub_create: world < 1"); return nullptr; }
  gvt_hip_hub *H = new gvt_hip_hub();
  H->world = world;
  H->slots.reset(new gvt_hip_hub::Slot[(size_t)world * world]);
  H->sent.reset(new hipEvent_t[world]());
  H->taken.reset(new hipEvent_t[world]());
  return H;
}
extern "C" void gvt_hip_hub_abort(gvt_hip_hub *H) { // every rank waiting in an exchange sees it (a rank failed)
  if (H) H->aborted.store(true);
}
extern "C" void gvt_hip_hub_destroy(gvt_hip_hub *H) {
  if (!H) return;
  for (int r = 0; r < H->world; r++) { if (H->sent[r]) hipEventDestroy(H->sent[r]); if (H->taken[r]) hipEventDestroy(H->taken[r]); }
  delete H;
}

extern "C" int gvt_hip_comm_unique_id(unsigned char id[128]) {
  if (!id) { set_error("comm_unique_id: null"); return GVT_HIP_ERR_INVALID; }
  int rc = rccl_load();
  if (rc) return rc;
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
  ncclUniqueId u;
  NCCLCHK(g_rccl.GetUniqueId(&u));
  std::memcpy(id, &u, 128);
  return 0;
}

static gvt_hip_comm *comm_new(int rank, int world) {
  if (ensure_init()) return nullptr;
  if (world < 1 || rank < 0 || rank >= world) { set_error("comm_create: bad rank %d of %d", rank, world); return nullptr; }
  gvt_hip_comm *K = new gvt_hip_comm();
  K->rank = rank; K->world = world;
  if (hipStreamCreateWithFlags(&K->stream, hipStreamNonBlocking) != hipSuccess) { set_error("comm_create: stream"); delete K; return nullptr; }
  if (hipEventCreateWithFlags(&K->ev_wait, hipEventDisableTiming) != hipSuccess) { set_error("comm_create: event"); hipStreamDestroy(K->stream); delete K; return nullptr; }
  if (const char *e = getenv("GVT_HIP_EXCHANGE_TIMEOUT_MS")) { const int v = atoi(e); if (v > 0) K->deadline_ms = v; }
  if (const char *e = getenv("GVT_HIP_CONNECT_TIMEOUT_MS")) { const int v = atoi(e); if (v > 0) K->connect_ms = v; }
  K->count = world;
  return K;
}
// Knob comm_cus = k > 0: the communicator's own stream may only use the device's first k compute units (CU mask) and the calling context's stream the others
// (measured, profiles/r06_comm_cus.txt: 8 or 16 masked CUs cost a frame 16-28 % -- the persistent grids lose their even spread --, k = 32 costs 5.6 %: the first 32
// bits are about one whole XCD, not one CU of every XCD as this comment used to assume) -- its persistent traversal grids are
// then sized for those (Ctx::n_cu).  A payload that moves on the communicator's stream (payload_overlap_kb) then finds CUs that no
// persistent wave holds; without the reservation an RCCL kernel beside k_trace waits for a block of it to leave.  Off by default.
static int comm_reserve_cus(gvt_hip_comm *K) {
  Ctx &C = gctx();
  const int k = C.comm_cus;
  if (k <= 0) return 0;
  hipDeviceProp_t prop;
  HIPCHK(hipGetDeviceProperties(&prop, C.device));
  const int total = prop.multiProcessorCount;
  if (k > total / 2) { set_error("comm_cus = %d: more than half of the device's %d compute units", k, total); return GVT_HIP_ERR_INVALID; }
  if (C.cu_reserved && C.cu_reserved != k) { set_error("comm_cus = %d, but this context's stream already leaves %d compute units free", k, C.cu_reserved); return GVT_HIP_ERR_INVALID; }
  const int words = (total + 31) / 32;
  std::vector<uint32_t> m_comm(words, 0u), m_comp(words, 0u);
  // which k compute units?  GVT_HIP_CU_MASK_LAYOUT=1: bit b belongs to XCD b / (total / 8) -- the k are taken k / 8 from every XCD; default (0): the first k bits
  // (right if the mask's bits go round the XCDs).  profiles/r06_comm_cus.txt has both measured
  static const int layout = getenv("GVT_HIP_CU_MASK_LAYOUT") ? atoi(getenv("GVT_HIP_CU_MASK_LAYOUT")) : 0;
  const int per_xcd = total / 8 > 0 ? total / 8 : total;
  for (int b = 0; b < total; b++) {
    const bool comm = layout == 1 ? (b % per_xcd) < (k + 7) / 8 : b < k;
    (comm ? m_comm : m_comp)[b >> 5] |= 1u << (b & 31);
  }
  hipStream_t s = nullptr;
  HIPCHK(hipExtStreamCreateWithCUMask(&s, (uint32_t)words, m_comm.data()));
  hipStreamDestroy(K->stream);
  K->stream = s;
  if (!C.cu_reserved) {
    hipStream_t cs = nullptr;
    HIPCHK(hipExtStreamCreateWithCUMask(&cs, (uint32_t)words, m_comp.data()));
    HIPCHK(hipStreamSynchronize(C.own_stream));
    if (C.stream == C.own_stream) C.stream = cs;
    hipStreamDestroy(C.own_stream);
    C.own_stream = cs;
    C.cu_reserved = k;
    C.n_cu = total - k;
  }
  K->cus = k;
  return 0;
}
extern "C" int gvt_hip_comm_reserved_cus(const gvt_hip_comm *K) { return K ? K->cus : 0; }

extern "C" gvt_hip_comm *gvt_hip_comm_create(const unsigned char id[128], int rank, int world) {
  if (!id) { set_error("comm_create: null id"); return nullptr; }
  if (rccl_load()) return nullptr;
  gvt_hip_comm *K = comm_new(rank, world);
  if (!K) return nullptr;
  if (comm_reserve_cus(K)) { hipEventDestroy(K->ev_wait); hipStreamDestroy(K->stream); delete K; return nullptr; }
  ncclUniqueId u;
  std::memcpy(&u, id, 128);
  ncclResult_t r = g_rccl.CommInitRank(&K->nccl, world, u, rank);
  if (r != ncclSuccess) { set_error("ncclCommInitRank failed: %s", g_rccl.GetErrorString(r)); hipEventDestroy(K->ev_wait); hipStreamDestroy(K->stream); delete K; return nullptr; }
  int cnt = -1;
  r = g_rccl.CommCount(K->nccl, &cnt);
  if (r != ncclSuccess || cnt != world) { // the communicator RCCL built is not the one the launcher asked for
    set_error("gvt_hip_comm_create: ncclCommCount = %d, expected the world size %d (%s)", cnt, world, r != ncclSuccess ? g_rccl.GetErrorString(r) : "ok");
    g_rccl.CommDestroy(K->nccl); hipEventDestroy(K->ev_wait); hipStreamDestroy(K->stream); delete K; return nullptr;
  }
  K->count = cnt;
  return K;
}
extern "C" gvt_hip_comm *gvt_hip_comm_create_local(gvt_hip_hub *hub, int rank) {
  if (!hub) { set_error("comm_create_local: null hub"); return nullptr; }
  gvt_hip_comm *K = comm_new(rank, hub->world);
  if (K && comm_reserve_cus(K)) { hipEventDestroy(K->ev_wait); hipStreamDestroy(K->stream); delete K; return nullptr; }
  if (K) K->hub = hub;
  return K;
}
extern "C" void gvt_hip_comm_destroy(gvt_hip_comm *K) {
  if (!K) return;
  if (!K->dead) hipStreamSynchronize(K->stream); // (a dead communicator's stream may hold an exchange that never completes)
  if (K->nccl) g_rccl.CommDestroy(K->nccl);
  if (K->ev_wait) hipEventDestroy(K->ev_wait);
  if (!K->dead) hipStreamDestroy(K->stream);
  delete K;
}
extern "C" int gvt_hip_comm_count(const gvt_hip_comm *K) { return K ? K->count : 1; }
extern "C" int gvt_hip_comm_set_deadline_ms(gvt_hip_comm *K, int ms) {
  if (!K || ms <= 0) { set_error("comm_set_deadline_ms: null communicator or ms <= 0"); return GVT_HIP_ERR_INVALID; }
  K->deadline_ms = ms;
  return 0;
}
extern "C" int gvt_hip_comm_rank(const gvt_hip_comm *K) { return K ? K->rank : 0; }
extern "C" int gvt_hip_comm_world(const gvt_hip_comm *K) { return K ? K->world : 1; }

namespace {
// A deadline passed: nothing this rank still has in flight on the communicator can be trusted to complete.  The RCCL communicator is
// aborted (its kernels leave the stream), the in-process hub wakes every rank; the caller returns GVT_HIP_ERR_TIMEOUT and the process
// is expected to report and exit non-zero -- never to re-exec itself.
int comm_timed_out(gvt_hip_comm *K, const char *what, const char *diag, int waited_ms = 0) {
  K->dead = true;
  if (K->hub) gvt_hip_hub_abort(K->hub);
  else if (K->nccl && g_rccl.CommAbort) { g_rccl.CommAbort(K->nccl); K->nccl = nullptr; }
  set_error("ray exchange: rank %d of %d waited more than %d ms for %s%s%s -- a peer has stopped taking part; the communicator is aborted", K->rank, K->world,
            waited_ms ? waited_ms : K->deadline_ms, what, diag && *diag ? "; " : "", diag ? diag : "");
  return GVT_HIP_ERR_TIMEOUT;
}
// host wait for an event with the communicator's deadline (hipEventSynchronize would wait for ever on an exchange a peer never joins).
// all_pairs: the awaited work sent to and received from EVERY peer (an announce exchange): only then is the communicator known to be
// connected and the long allowance of RCCL's first connections over (a composite or an image-split frame touches some pairs only).
int bounded_event_wait(gvt_hip_comm *K, hipEvent_t ev, const char *what, const char *diag = nullptr, bool all_pairs = false) {
  const auto t0 = std::chrono::steady_clock::now();
  const int limit_ms = (K->nccl && !K->connected) ? std::max(K->deadline_ms, K->connect_ms) : K->deadline_ms;
  int rc = 0;
  for (unsigned spins = 0;; spins++) {
    const hipError_t e = hipEventQuery(ev);
    if (e == hipSuccess) { if (all_pairs) K->connected = true; break; }
    if (e != hipErrorNotReady) { set_error("hipEventQuery while waiting for %s: %s", what, hipGetErrorString(e)); rc = GVT_HIP_ERR_DEVICE; break; }
    if ((spins & 63u) == 63u) {
      if (K->hub && K->hub->aborted.load()) { set_error("hub: aborted while waiting for %s", what); K->dead = true; rc = GVT_HIP_ERR_TIMEOUT; break; }
      if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(limit_ms)) { rc = comm_timed_out(K, what, diag, limit_ms); break; }
      if (spins > 4096u) std::this_thread::yield();
    }
  }
  K->ms_host_wait += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  return rc;
}
// the same for a word of pinned host memory that a kernel behind the exchange releases (k_publish): waited for with loads
int bounded_flag_wait(gvt_hip_comm *K, const unsigned *flag, unsigned want, hipStream_t st, const char *what, const char *diag, bool all_pairs) {
  const auto t0 = std::chrono::steady_clock::now();
  const int limit_ms = (K->nccl && !K->connected) ? std::max(K->deadline_ms, K->connect_ms) : K->deadline_ms;
  int rc = 0;
  for (unsigned spins = 0;; spins++) {
    if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == want) { if (all_pairs) K->connected = true; break; }
    if ((spins & 1023u) == 1023u) {
      if (K->hub && K->hub->aborted.load()) { set_error("hub: aborted while waiting for %s", what); K->dead = true; rc = GVT_HIP_ERR_TIMEOUT; break; }
      if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(limit_ms)) { rc = comm_timed_out(K, what, diag, limit_ms); break; }
      if ((spins & 0xffffu) == 0xffffu) { // a stream that ended (or faulted) without the word: not a peer's doing
        const hipError_t e = hipStreamQuery(st);
        if (e != hipErrorNotReady && __atomic_load_n(flag, __ATOMIC_ACQUIRE) != want) {
          set_error("%s: the stream %s without publishing report %u", what, e == hipSuccess ? "finished" : hipGetErrorString(e), want);
          rc = GVT_HIP_ERR_DEVICE; break;
        }
        std::this_thread::yield();
      }
    }
  }
  K->ms_host_wait += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  return rc;
}
int bounded_stream_wait(gvt_hip_comm *K, hipStream_t st, const char *what, const char *diag = nullptr) {
  HIPCHK(hipEventRecord(K->ev_wait, st));
  return bounded_event_wait(K, K->ev_wait, what, diag);
}
void comm_group_begin(gvt_hip_comm *K) { K->ops.clear(); }
void comm_send(gvt_hip_comm *K, const void *p, size_t bytes, int peer) { K->ops.push_back({ 1, (void *)p, bytes, peer, 0 }); }
void comm_recv(gvt_hip_comm *K, void *p, size_t bytes, int peer, int accumulate = 0) { K->ops.push_back({ 0, p, bytes, peer, accumulate }); }

int hub_group_end(gvt_hip_comm *K) {
  gvt_hip_hub *H = K->hub;
  const int W = H->world, me = K->rank;
  auto gone = [&](const char *what) -> int {
    if (H->aborted.load()) { set_error("hub: aborted"); K->dead = true; return GVT_HIP_ERR_TIMEOUT; }
    return comm_timed_out(K, what, nullptr);
  };
  bool any_send = false, any_recv = false;
  for (auto &o : K->ops) { any_send = any_send || o.send; any_recv = any_recv || !o.send; }
  if (!H->sent[me]) { HIPCHK(hipEventCreateWithFlags(&H->sent[me], hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&H->taken[me], hipEventDisableTiming)); }
  // 1. post every send behind ONE event (the slots are free: the previous group waited for their consumption)
  if (any_send) HIPCHK(hipEventRecord(H->sent[me], K->stream));
  for (auto &o : K->ops) {
    if (!o.send) continue;
    gvt_hip_hub::Slot &s = H->slots[(size_t)o.peer * W + me];
    if (!hub_await(H, s.state, 0, K->deadline_ms)) return gone("a free send slot (in-process transport)");
    s.ready = H->sent[me];
    s.ptr = o.ptr; s.bytes = o.bytes;
    s.state.store(1, std::memory_order_release);
  }
  // 2. every receive: wait for the matching send, then ONE kernel gathers them all (word copies, or float adds for a reduce) behind the senders' events
  HubBatch B;
  B.n = 0;
  unsigned long long max_words = 0;
  auto flush = [&]() -> int {
    if (!B.n) return 0;
    const unsigned gx = (unsigned)std::min<unsigned long long>(4096ull, (max_words + 255ull) / 256ull);
    k_hub_gather<<<dim3(gx ? gx : 1u, (unsigned)B.n), 256, 0, K->stream>>>(B);
    HIPCHK(hipGetLastError());
    B.n = 0; max_words = 0;
    return 0;
  };
  for (auto &o : K->ops) {
    if (o.send) continue;
    gvt_hip_hub::Slot &s = H->slots[(size_t)me * W + o.peer];
    if (!hub_await(H, s.state, 1, K->deadline_ms)) return gone("a peer's matching send (in-process transport)");
    if (s.bytes != o.bytes) { set_error("hub: rank %d expects %zu bytes from %d, which sends %zu", me, o.bytes, o.peer, s.bytes); H->aborted.store(true); return GVT_HIP_ERR_INVALID; }
    if (o.bytes & 3u) { set_error("hub: message of %zu bytes is not a multiple of 4", o.bytes); H->aborted.store(true); return GVT_HIP_ERR_INVALID; }
    HIPCHK(hipStreamWaitEvent(K->stream, s.ready, 0));
    if (o.bytes) {
      if (B.n == HUB_ITEMS || o.accumulate) { int rc = flush(); if (rc) return rc; } // (adds into one destination from several peers: one launch each, in rank order)
      B.it[B.n++] = HubItem{ (unsigned *)o.ptr, (const unsigned *)s.ptr, (unsigned long long)(o.bytes / 4), o.accumulate };
      max_words = std::max<unsigned long long>(max_words, o.bytes / 4);
      if (o.accumulate) { int rc = flush(); if (rc) return rc; }
    }
  }
  { int rc = flush(); if (rc) return rc; }
  if (any_recv) HIPCHK(hipEventRecord(H->taken[me], K->stream));
  for (auto &o : K->ops) {
    if (o.send) continue;
    gvt_hip_hub::Slot &s = H->slots[(size_t)me * W + o.peer];
    s.copied = H->taken[me];
    s.state.store(2, std::memory_order_release);
  }
  // 3. own sends consumed: later work on this stream must not overwrite a buffer that is still being read
  for (auto &o : K->ops) {
    if (!o.send) continue;
    gvt_hip_hub::Slot &s = H->slots[(size_t)o.peer * W + me];
    if (!hub_await(H, s.state, 2, K->deadline_ms)) return gone("a peer to take a send (in-process transport)");
    HIPCHK(hipStreamWaitEvent(K->stream, s.copied, 0));
    s.state.store(0, std::memory_order_release);
  }
  K->ops.clear();
  return 0;
}

int comm_group_end(gvt_hip_comm *K) {
  if (K->dead) { set_error("ray exchange: the communicator of rank %d was aborted after a deadline passed", K->rank); return GVT_HIP_ERR_TIMEOUT; }
  if (!K->ops.empty()) K->groups++;
  if (K->hub) return hub_group_end(K);
  if (K->ops.empty()) return 0;
  NCCLCHK(g_rccl.GroupStart());
  for (auto &o : K->ops) {
    if (!o.bytes) continue;
    if (o.send) NCCLCHK(g_rccl.Send(o.ptr, o.bytes, ncclUint8, o.peer, K->nccl, K->stream));
    else NCCLCHK(g_rccl.Recv(o.ptr, o.bytes, ncclUint8, o.peer, K->nccl, K->stream));
  }
  NCCLCHK(g_rccl.GroupEnd());
  K->ops.clear();
  return 0;
}

// sum of every rank's float buffer on `root`, in place (IceTComposite::composite, composite/IceTComposite.cpp:84-101)
int comm_reduce_sum(gvt_hip_comm *K, float *buf, size_t n_floats, int root) {
  if (K->world == 1) return 0;
  if (!K->hub) {
    NCCLCHK(g_rccl.Reduce(buf, buf, n_floats, ncclFloat, ncclSum, root, K->nccl, K->stream));
    return 0;
  }
  comm_group_begin(K);
  if (K->rank == root) { for (int p = 0; p < K->world; p++) if (p != root) comm_recv(K, buf, n_floats * 4, p, 1); }
  else comm_send(K, buf, n_floats * 4, root);
  return comm_group_end(K);
}
} // namespace

// Loop-back check of the transport on THIS rank: a grouped send-to-self / receive-from-self of `bytes` bytes and an in-place reduce,
// through exactly the calls the frame loop makes (ncclSend / ncclRecv / ncclReduce on the communication stream, or the in-process
// transport's copies).  Lets a one-GPU machine exercise the RCCL entry points' signatures, datatypes and stream ordering.
extern "C" int gvt_hip_comm_selftest(gvt_hip_comm *K, size_t bytes) {
  if (!K || bytes < 16) { set_error("comm_selftest: null communicator or fewer than 16 bytes"); return GVT_HIP_ERR_INVALID; }
  bytes &= ~(size_t)15;
  unsigned char *a = nullptr, *b = nullptr;
  HIPCHK(hipMalloc((void **)&a, bytes));
  HIPCHK(hipMalloc((void **)&b, bytes));
  std::vector<unsigned char> h(bytes), g(bytes, 0);
  for (size_t i = 0; i < bytes; i++) h[i] = (unsigned char)(i * 131u + 7u);
  int rc = 0;
  hipError_t e = hipMemcpyAsync(a, h.data(), bytes, hipMemcpyHostToDevice, K->stream);
  if (e == hipSuccess) e = hipMemsetAsync(b, 0, bytes, K->stream);
  if (e != hipSuccess) rc = GVT_HIP_ERR_DEVICE;
  if (!rc) {
    comm_group_begin(K);
    comm_send(K, a, bytes, K->rank);
    comm_recv(K, b, bytes, K->rank);
    rc = comm_group_end(K);
  }
  if (!rc && hipMemcpyAsync(g.data(), b, bytes, hipMemcpyDeviceToHost, K->stream) != hipSuccess) rc = GVT_HIP_ERR_DEVICE;
  if (!rc && hipStreamSynchronize(K->stream) != hipSuccess) rc = GVT_HIP_ERR_DEVICE;
  if (!rc && std::memcmp(g.data(), h.data(), bytes) != 0) { set_error("comm_selftest: the loop-back payload differs"); rc = GVT_HIP_ERR_DEVICE; }
  if (!rc && K->nccl) { // ncclReduce in place over one rank: the identity
    ncclResult_t r = g_rccl.Reduce(b, b, bytes / 4, ncclFloat, ncclSum, 0, K->nccl, K->stream);
    if (r != ncclSuccess) { set_error("comm_selftest: ncclReduce failed: %s", g_rccl.GetErrorString(r)); rc = GVT_HIP_ERR_DEVICE; }
    if (!rc && (hipMemcpyAsync(g.data(), b, bytes, hipMemcpyDeviceToHost, K->stream) != hipSuccess || hipStreamSynchronize(K->stream) != hipSuccess)) rc = GVT_HIP_ERR_DEVICE;
  }
  hipFree(a); hipFree(b);
  if (rc == GVT_HIP_ERR_DEVICE && !*gvt_hip_last_error()) set_error("comm_selftest: HIP error");
  return rc;
}

// ------------------------------------------------------------------------------------------------
// device helpers of the scheduler loop
// ------------------------------------------------------------------------------------------------
namespace {
#define ANN_HEAD 11 // announce row: {rays to you, bytes to you, my total outgoing rays, my local pending rays, the bounding rectangle of the
                    // pixels my framebuffer holds deposits in (x0, y0, x1, y1; for the composite), my error code (0: none -- a rank whose
                    // local work failed still takes part in the exchange, so that EVERY rank leaves the frame, with the same error),
                    // my exchange number (ranks out of step are a protocol error, not a hang), bytes of payload carried INLINE behind this
                    // row (0: none, or too large -- it follows in the payload exchange), rays per queue [n_inst]}
                    // The message a rank sends a peer per tick is this row followed by a fixed inline area (knob inline_kb, the same on every
                    // rank): a tick whose pairs all have at most that much to say is ONE exchange (one ncclGroup), the rays travel with the
                    // count exchange of SendRays (DomainTracer.h:397-415) instead of behind it (:433-463).
#define REPORT_TAIL 15 // words behind the queue sizes in the round's report (the last one: the sequence number of a polled report); [12]: the speculative part
                       // enqueued behind this report's exchange is VALID (k_publish decides, host and device read the same word); [13]: rays the speculative
                       // chain in front of this report traced (0: none ran)
// Device words of a tracer's speculation (gvt_hip_tracer::d_spec): [0] k_round_report: this rank announced a payload beyond the inline area, or an error (the
// next tick needs the host); [1] k_publish: the speculative part behind this exchange is valid; [2] k_spec_begin: rays of the speculative chain
#define SPEC_SLOW 0
#define SPEC_VALID 1
#define SPEC_N 2

// dword w (0..19) of ray `ray`'s 80-byte wire image (actor/Ray.h:68-96; words 16..19: the stream word and the known-miss list, inside Ray::data)
__device__ inline unsigned wire_word(const RayPlanes &q, unsigned ray, unsigned w) {
  if (w < 16) { const float4 *pl = w < 4 ? q.p0 : w < 8 ? q.p1 : w < 12 ? q.p2 : q.p3; return ((const unsigned *)(pl + ray))[w & 3]; }
  if (w == 16) return q.p4[ray];
  return q.p5[3 * (size_t)ray + (w - 17)];
}

// sizes[i] = *count_ptr[i]; the announce row for every peer; totals (2 x u64) and flags copied next to them so that ONE device-to-host
// copy carries everything the host needs from a round
// scan_fb (exchanging ranks): the launch has many blocks; every block first folds its share of the framebuffer into the rectangle of the
// pixels that hold a deposit (every deposit adds 1.0 to alpha, IceTComposite::localAdd; deposits only ever add pixels within a frame,
// so the rectangle accumulates from k_zero_totals on), takes a ticket, and the LAST block to arrive writes the report -- one launch where
// there were three (k_bbox_init, k_fb_bbox, k_round_report).
__global__ __launch_bounds__(256) void k_round_report(unsigned *const *__restrict__ count_ptr, const int *__restrict__ owner, int n_inst, int rank, int world,
                               unsigned *sizes, int *__restrict__ ann /* [world][ANN_HEAD + n_inst] */, unsigned *counters,
                               unsigned *overflow /* [0] flags, [4..7] rectangle, [10] this kernel's ticket */, unsigned *tail /* sizes + n_inst: tot[4], ovf trav, ovf queue, bbox[4] */,
                               int *bbox, int chain_end = 0, const unsigned char *__restrict__ chain_mask = nullptr,
                               unsigned *host_report = nullptr, unsigned host_seq = 0u, int err_code = 0, int tick = 0, const float4 *__restrict__ scan_fb = nullptr,
                               int fb_w = 0, int fb_h = 0, int msg_ints = 0 /* stride of the announce rows (row + inline area), ints */, int inl_bytes = 0,
                               const QueueDesc *__restrict__ qd = nullptr, unsigned *spec = nullptr /* SPEC_* words */, int speculative = 0) {
  __shared__ unsigned long long sh_out, sh_local;
  if (speculative && !spec[SPEC_VALID]) return; // enqueued ahead of the host's knowledge and the exchange in front called for the host: nothing happened
  if (speculative && scan_fb && spec[SPEC_N] == 0u) { // the speculative round held no ray: no new deposit, the rectangle stands -- one block reports, nobody scans
    if (blockIdx.x) return;
    scan_fb = nullptr;
  }
  if (scan_fb) {
    __shared__ int sh_bb[4];
    __shared__ unsigned sh_ticket;
    if (threadIdx.x == 0) { sh_bb[0] = fb_w; sh_bb[1] = fb_h; sh_bb[2] = 0; sh_bb[3] = 0; }
    __syncthreads();
    const unsigned n_pix = (unsigned)(fb_w * fb_h);
    int x0 = fb_w, y0 = fb_h, x1 = 0, y1 = 0;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n_pix; i += gridDim.x * blockDim.x)
      if (scan_fb[i].w > 0.f) { const int x = (int)(i % (unsigned)fb_w), y = (int)(i / (unsigned)fb_w); x0 = min(x0, x); y0 = min(y0, y); x1 = max(x1, x + 1); y1 = max(y1, y + 1); }
    if (x1 > x0) { atomicMin(&sh_bb[0], x0); atomicMin(&sh_bb[1], y0); atomicMax(&sh_bb[2], x1); atomicMax(&sh_bb[3], y1); }
    __syncthreads();
    if (threadIdx.x == 0) {
      if (sh_bb[2] > sh_bb[0]) { atomicMin(&bbox[0], sh_bb[0]); atomicMin(&bbox[1], sh_bb[1]); atomicMax(&bbox[2], sh_bb[2]); atomicMax(&bbox[3], sh_bb[3]); }
      __threadfence();
      sh_ticket = atomicAdd(&overflow[10], 1u);
    }
    __syncthreads();
    if (sh_ticket != gridDim.x - 1) return; // not the last block
    __threadfence();
    if (threadIdx.x == 0) overflow[10] = 0u; // for the next report
  }
  if (threadIdx.x == 0) { sh_out = 0; sh_local = 0; }
  if (chain_end) { // the launch chain in front left its k_wave_end to this kernel: last pass's shadow rays into the frame total, traced queues cleared
    if (threadIdx.x == 0) { unsigned long long *tot = (unsigned long long *)(counters + 16); tot[1] += counters[1]; }
    for (int i = threadIdx.x; i < n_inst; i += blockDim.x)
      if (chain_mask[i]) *count_ptr[i] = 0u;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < n_inst; i += blockDim.x) {
    const unsigned s = *count_ptr[i];
    sizes[i] = s;
    if (owner[i] == rank) atomicAdd(&sh_local, (unsigned long long)s); else atomicAdd(&sh_out, (unsigned long long)s);
  }
  __syncthreads();
  const int row = ANN_HEAD + n_inst;
  if (msg_ints < row) msg_ints = row;
  int bb[4];
  for (int k = 0; k < 4; k++) bb[k] = __hip_atomic_load(&bbox[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  for (int p = threadIdx.x; p < world; p += blockDim.x) {
    unsigned rays = 0, queues = 0;
    for (int i = 0; i < n_inst; i++) {
      const unsigned s = (owner[i] == p && p != rank) ? sizes[i] : 0u;
      ann[p * msg_ints + ANN_HEAD + i] = (int)s;
      rays += s; queues += s ? 1u : 0u;
    }
    const unsigned bytes = rays * 80u + queues * 8u; // SendRays: packed rays + {queue number, ray count} per queue (:397-407)
    ann[p * msg_ints + 0] = (int)rays;
    ann[p * msg_ints + 1] = (int)bytes;
    ann[p * msg_ints + 2] = (int)sh_out;
    ann[p * msg_ints + 3] = (int)sh_local;
    for (int k = 0; k < 4; k++) ann[p * msg_ints + 4 + k] = bb[k];
    ann[p * msg_ints + 8] = err_code;
    ann[p * msg_ints + 9] = tick;
    ann[p * msg_ints + 10] = (bytes && bytes <= (unsigned)inl_bytes && !err_code) ? (int)bytes : 0;
  }
  if (inl_bytes > 0 && !err_code) {
    // a peer's whole payload of this tick fits the inline area: packed here, in the reference's wire format, behind the row it is announced
    // in; the queues it came from are cleared (the sender's q.second.clear(), :455) -- `sizes` keeps what they held, the host applies
    // the same rule to it
    __syncthreads();
    for (int p = 0; p < world; p++) {
      const unsigned bytes = (unsigned)ann[p * msg_ints + 10];
      if (!bytes) continue;
      unsigned *buf = (unsigned *)(ann + (size_t)p * msg_ints + row);
      unsigned off = 0;
      for (int i = 0; i < n_inst; i++) {
        const unsigned n = (unsigned)ann[p * msg_ints + ANN_HEAD + i];
        if (!n) continue;
        const RayPlanes q = make_planes(qd[i].planes, qd[i].cap);
        for (unsigned l = threadIdx.x; l < 2u + 20u * n; l += blockDim.x) buf[off + l] = l < 2 ? (l == 0 ? (unsigned)i : n) : wire_word(q, (l - 2) / 20u, (l - 2) % 20u);
        if (threadIdx.x == 0) *count_ptr[i] = 0u;
        off += 2u + 20u * n;
      }
    }
  }
  if (threadIdx.x == 0) {
    tail[0] = counters[16]; tail[1] = counters[17]; tail[2] = counters[18]; tail[3] = counters[19];
    tail[4] = counters[8]; tail[5] = overflow[0];
    for (int k = 0; k < 4; k++) tail[6 + k] = (unsigned)bb[k];
    tail[10] = counters[9]; // packets handed over by k_packet
    tail[11] = counters[20] + counters[3]; // closest-hit rays parked for k_long_closest so far this frame (earlier launches + the last one)
    tail[12] = 0u;
    tail[13] = (speculative && spec) ? spec[SPEC_N] : 0u;
    counters[0] = 0u;       // the work counter of the next small chain (k_finish starts from 0 without a memset in front)
    if (spec) { // does the tick behind this one need the host?  (a pair's payload beyond the inline area, or this rank's error)
      unsigned slow = err_code ? 1u : 0u;
      for (int p = 0; p < world; p++) if (p != rank && ann[p * msg_ints + 1] > 0 && ann[p * msg_ints + 10] == 0) slow = 1u;
      spec[SPEC_SLOW] = slow;
    }
  }
  if (host_report) { // one rank: the report goes straight into the host's pinned copy, the sequence word last -- the host polls it
    __syncthreads(); // instead of a device-to-host copy + a stream synchronisation (an interrupt and a wake-up per round)
    for (int i = threadIdx.x; i < n_inst; i += blockDim.x) host_report[i] = sizes[i];
    for (int k = threadIdx.x; k < REPORT_TAIL - 1; k += blockDim.x) host_report[n_inst + k] = tail[k];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) { __hip_atomic_store(host_report + n_inst + REPORT_TAIL - 1, host_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
  }
}

// wire image of one queue: [int32 queueId][int32 nRays][nRays x 80-byte Ray] (DomainTracer.h:441-455); one thread per dword.  ALL queues
// of a tick -- every peer, every queue -- go through ONE launch: the items travel by value in the kernel's arguments (<= WIRE_MAX per
// launch), a thread finds its item by its dword number.
#define WIRE_MAX 48
struct WireItem {
  float4 *planes;            // the queue
  unsigned long long cap;
  unsigned *count;
  unsigned *buf;             // the item's wire image (header first)
  unsigned n;                // rays
  int qid;
  unsigned dword0;           // number of this item's first dword in the launch
  unsigned prior;            // unpack: rays of earlier items of this launch that go to the same queue
};
struct WireBatch {
  int n_items;
  unsigned total;            // dwords of the launch
  WireItem it[WIRE_MAX];
};
__device__ inline int wire_item_of(const WireBatch &B, unsigned t) {
  int lo = 0, hi = B.n_items - 1;
  while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (B.it[mid].dword0 <= t) lo = mid; else hi = mid - 1; }
  return lo;
}
// pack: also the sender's q.second.clear() (:455) -- the queue's count word is reset by the item's first thread (nothing reads it here: n comes from the host)
__global__ __launch_bounds__(256) void k_pack_all(const WireBatch B) {
  const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= B.total) return;
  const WireItem &I = B.it[wire_item_of(B, t)];
  const unsigned l = t - I.dword0;
  if (l == 0) *I.count = 0u;
  if (l < 2) { I.buf[l] = l == 0 ? (unsigned)I.qid : I.n; return; }
  const unsigned d = l - 2;
  I.buf[l] = wire_word(make_planes(I.planes, I.cap), d / 20u, d % 20u); // (the known-miss list travels with the ray, bytes 68..79)
}
// unpack: appended behind the queue's current rays.  Every thread reads the queue's count word as its base; the LAST block of the
// launch to finish (ticket) -- every other block has read its bases by then -- advances the count words.
__global__ __launch_bounds__(256) void k_unpack_all(const WireBatch B, unsigned *__restrict__ err, unsigned *ticket) {
  __shared__ unsigned sh_ticket;
  const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < B.total) {
    const WireItem &I = B.it[wire_item_of(B, t)];
    const unsigned l = t - I.dword0;
    if (l < 2) { if (I.buf[l] != (l == 0 ? (unsigned)I.qid : I.n)) atomicOr(err, 2u); } // in-band header disagrees with the announce
    else {
      const unsigned d = l - 2, ray = d / 20u, w = d % 20u;
      const unsigned long long slot = (unsigned long long)__hip_atomic_load(I.count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + I.prior + ray;
      if (slot >= I.cap) atomicOr(err, 1u);
      else {
        const RayPlanes q = make_planes(I.planes, I.cap);
        const unsigned v = I.buf[l];
        if (w < 16) { float4 *pl = w < 4 ? q.p0 : w < 8 ? q.p1 : w < 12 ? q.p2 : q.p3; ((unsigned *)(pl + slot))[w & 3] = v; }
        else if (w == 16) q.p4[slot] = v;
        else q.p5[3 * (size_t)slot + (w - 17)] = v;
      }
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) { __threadfence(); sh_ticket = atomicAdd(ticket, 1u); }
  __syncthreads();
  if (sh_ticket != gridDim.x - 1) return;
  __threadfence();
  for (int k = threadIdx.x; k < B.n_items; k += blockDim.x) atomicAdd(B.it[k].count, B.it[k].n);
  if (threadIdx.x == 0) *ticket = 0u;
}
// rectangle of the framebuffer <-> contiguous buffer (float4 pixels); ADD: buffer added into the framebuffer
template <bool ADD> __global__ __launch_bounds__(256) void k_rect(float4 *__restrict__ fb, int W, int x0, int y0, int w, int h, float4 *__restrict__ buf) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (unsigned)(w * h)) return;
  const size_t px = (size_t)(y0 + (int)(i / (unsigned)w)) * W + x0 + (int)(i % (unsigned)w);
  if (ADD) { float4 a = fb[px]; const float4 b = buf[i]; a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; fb[px] = a; }
  else buf[i] = fb[px];
}
// behind an announce exchange, ONE launch of `world` blocks: block `rank` gathers the peers' rows (without their inline areas) and this rank's
// report into the host's pinned mirror and releases the sequence word last -- the host polls it (no device-to-host copy, no event, no interrupt
// between the exchange and the vote); block p != rank appends the payload peer p sent INLINE behind its row to this rank's queues, where it
// arrived (no separate unpack launch, and the append overlaps the host's wake-up).  Slots are deterministic: a queue's incoming rays lie behind
// its current ones in peer order; every thread reads the count words as its base and the LAST block to finish advances them (ticket).
__global__ __launch_bounds__(256) void k_publish(const int *__restrict__ ann_in, int msg_ints, int row, int world, int rank, const unsigned *__restrict__ report, int n_rep,
                                                  int *h_ann, unsigned *h_report, unsigned seq, const QueueDesc *__restrict__ qd, int n_inst, unsigned *__restrict__ err,
                                                  unsigned *ticket, unsigned *spec = nullptr, const int *__restrict__ owner = nullptr, unsigned spec_max = 0u,
                                                  WaveSeg *__restrict__ spec_segs = nullptr) {
  const int p = (int)blockIdx.x;
  if (p == rank) {
    for (int k = threadIdx.x; k < world * row; k += blockDim.x) h_ann[k] = ann_in[(size_t)(k / row) * msg_ints + (k % row)];
    for (int k = threadIdx.x; k < n_rep; k += blockDim.x) h_report[k] = report[k];
    if (spec) {
      // Is the speculative part enqueued behind this exchange (the next tick's small round + report) valid?  Not if this tick needs the host: a payload beyond
      // the inline area out (k_round_report's word) or in, an error anywhere, or more local rays than one k_finish launch takes.  The rays the next round
      // holds are known exactly: this rank's own queues as reported + what the peers sent inline (the other blocks of this launch append them).
      __syncthreads();
      if (threadIdx.x == 0) {
        unsigned bad = spec[SPEC_SLOW];
        unsigned long long rays = 0ull;
        for (int pp = 0; pp < world && !bad; pp++) {
          if (pp == rank) continue;
          const int *a = ann_in + (size_t)pp * msg_ints;
          if (a[8] != 0 || (a[1] > 0 && a[10] == 0)) bad = 1u;
          if (a[10] > 0) rays += (unsigned)a[0];
        }
        for (int i = 0; i < n_inst && !bad; i++) if (owner[i] == rank) rays += report[i];
        if (rays > spec_max) bad = 1u;
        const unsigned valid = bad ? 0u : 1u;
        spec[SPEC_VALID] = valid;
        h_report[n_inst + 12] = valid; // (n_rep = n_inst + REPORT_TAIL - 1: the word lies inside the copied range; this store comes after the copy)
      }
    }
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(h_report + n_rep, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  } else if (qd && ann_in[(size_t)p * msg_ints + 10] > 0) {
    const int *head = ann_in + (size_t)p * msg_ints;
    const unsigned *buf = (const unsigned *)(head + row);
    unsigned off = 0;
    for (int i = 0; i < n_inst; i++) {
      const unsigned n = (unsigned)head[ANN_HEAD + i];
      if (!n) continue;
      unsigned prior = 0; // rays of earlier peers' inline payloads bound for the same queue
      for (int pp = 0; pp < p; pp++)
        if (pp != rank && ann_in[(size_t)pp * msg_ints + 10] > 0) prior += (unsigned)ann_in[(size_t)pp * msg_ints + ANN_HEAD + i];
      const QueueDesc Q = qd[i];
      const unsigned long long base = (unsigned long long)__hip_atomic_load(Q.count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + prior;
      const RayPlanes q = make_planes(Q.planes, Q.cap);
      if (threadIdx.x < 2 && buf[off + threadIdx.x] != (threadIdx.x == 0 ? (unsigned)i : n)) atomicOr(err, 2u); // in-band header disagrees with the row
      if (base + n > Q.cap) { if (threadIdx.x == 0) atomicOr(err, 1u); }
      else
        for (unsigned l = threadIdx.x; l < 20u * n; l += blockDim.x) {
          const unsigned ray = l / 20u, w = l % 20u, v = buf[off + 2u + l];
          const unsigned long long slot = base + ray;
          if (w < 16) { float4 *pl = w < 4 ? q.p0 : w < 8 ? q.p1 : w < 12 ? q.p2 : q.p3; ((unsigned *)(pl + slot))[w & 3] = v; }
          else if (w == 16) q.p4[slot] = v;
          else q.p5[3 * (size_t)slot + (w - 17)] = v;
        }
      off += 2u + 20u * n;
    }
  }
  if (!qd) return;
  __shared__ unsigned sh_ticket;
  __syncthreads();
  if (threadIdx.x == 0) { __threadfence(); sh_ticket = atomicAdd(ticket, 1u); }
  __syncthreads();
  if (sh_ticket != gridDim.x - 1) return;
  __threadfence();
  for (int i = threadIdx.x; i < n_inst; i += blockDim.x) {
    unsigned tot = 0;
    for (int pp = 0; pp < world; pp++)
      if (pp != rank && ann_in[(size_t)pp * msg_ints + 10] > 0) tot += (unsigned)ann_in[(size_t)pp * msg_ints + ANN_HEAD + i];
    if (tot) atomicAdd(qd[i].count, tot);
  }
  if (threadIdx.x == 0) *ticket = 0u;
  if (spec) {
    // Head of the next tick's SPECULATIVE part (k_finish + k_round_report are enqueued behind this launch before the host has read it): if block `rank` found it
    // valid -- every block took its ticket after its work, this is the last one -- the segment table of the small round is filled from the count words of this
    // rank's queues, which are then cleared: the rays now belong to the launch, exactly what the host-driven round does with its host-known sizes (finish_round)
    __syncthreads(); // (the count words above are final)
    if (threadIdx.x == 0) {
      unsigned run = 0u;
      if (spec[SPEC_VALID]) {
        int k = 0;
        for (int i = 0; i < n_inst; i++) {
          if (owner[i] != rank) continue;
          const unsigned n = __hip_atomic_load(qd[i].count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          WaveSeg sg;
          sg.planes = qd[i].planes; sg.cap = qd[i].cap; sg.begin = run; sg.n = n; sg.inst = i; sg.pad = 0;
          spec_segs[k++] = sg;
          run += n;
          *qd[i].count = 0u;
        }
      }
      spec[SPEC_N] = run;
    }
  }
}
// start of a frame: ray totals, flags, the work counter of the first small chain, the deposit rectangle (empty) and the kernels' tickets
// a frame's resets in one launch: every queue.clear() (count words) and the frame's totals / flags / deposit rectangle
__global__ void k_zero_totals(unsigned *c, unsigned *ovf, int fb_w, int fb_h, unsigned *const *__restrict__ count_ptr = nullptr, int n_inst = 0) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_inst; i += gridDim.x * blockDim.x) *count_ptr[i] = 0u;
  if (blockIdx.x) return;
  if (threadIdx.x < 4) c[16 + threadIdx.x] = 0u;
  if (threadIdx.x == 4) { *ovf = 0u; c[9] = 0u; c[0] = 0u; ovf[10] = 0u; ovf[11] = 0u; ovf[12] = 0u; c[3] = 0u; c[20] = 0u; }
  if (threadIdx.x == 5) { int *bb = (int *)(ovf + 4); bb[0] = fb_w; bb[1] = fb_h; bb[2] = 0; bb[3] = 0; }
}
} // namespace

// ------------------------------------------------------------------------------------------------
// the tracer object: Tracer<ImageScheduler> / Tracer<DomainScheduler> for one rank
// ------------------------------------------------------------------------------------------------
struct gvt_hip_tracer {
  gvt_hip_top *top = nullptr;
  size_t n_inst = 0;
  std::vector<gvt_hip_mesh *> meshes;
  std::vector<float> m, minv, normi;
  std::vector<gvt_hip_light> lights;
  int normal_mode = 0;
  gvt_hip_camera cam{};
  gvt_hip_fb *fb = nullptr;
  std::vector<gvt_hip_queue *> queues; // one per instance: local work (owned) or outgoing rays (not owned)
  gvt_hip_queue *q_moved = nullptr;
  std::vector<int> owner;
  std::vector<uint8_t> owned;
  gvt_hip_comm *comm = nullptr; // borrowed
  int rank = 0, world = 1;
  bool all_quad = true; // every local mesh carries the quad-per-ray layouts
  // device tables
  WaveInst *d_insts = nullptr;
  // per-round tables in ONE pinned block / ONE device block (a single host-to-device copy per round): segments, queue descriptors, mask
  void *d_round = nullptr, *h_round = nullptr;
  size_t round_bytes = 0;
  std::vector<unsigned char> round_uploaded; // what d_round holds (a round whose tables equal the last upload needs no copy)
  WaveSeg *d_segs = nullptr, *h_segs = nullptr;
  QueueDesc *d_qdesc = nullptr, *h_qdesc = nullptr;
  unsigned **d_count_ptr = nullptr;
  int *d_owner = nullptr;
  unsigned char *d_mask = nullptr, *h_mask = nullptr;
  unsigned *d_report = nullptr, *h_report = nullptr; // sizes[n_inst] + tail[REPORT_TAIL]; the tail's last word: sequence number of a polled report
  unsigned report_seq = 0;
  int *d_ann_out = nullptr, *d_ann_in = nullptr; // [world][msg_ints]: the announce row (ANN_HEAD + n_inst ints) + the inline payload area
  int *h_ann_in = nullptr;                       // [world][ANN_HEAD + n_inst]: the rows only (k_publish)
  size_t inl_bytes = 0, msg_ints = 0;            // inline area per pair (knob inline_kb when the tables were laid out), message length in ints
  double tick_us[8] = { 0 }; uint64_t tick_n[8] = { 0 }; double comp_us = 0; // GVT_HIP_SPEC_TRACE: host time from one announce's arrival to the next, by tick number
  uint64_t n_spec_enqueued = 0, n_spec_valid = 0, n_exchanges = 0; // since creation (GVT_HIP_SPEC_TRACE prints them when the tracer goes)
  size_t spec_recent_rays = 0;                   // rays of this rank's latest small round (sizes the speculative k_finish grid)
  bool sizes_exact = true;                       // no launch chain has run since the host last read the queue sizes
  bool spec_enqueued_last = false;               // a speculative part was enqueued behind the last announce exchange
  unsigned *d_spec = nullptr;                    // SPEC_* words of the speculative part of a tick
  WaveSeg *d_spec_segs = nullptr;                // its segment table (filled by k_spec_begin from the count words)
  bool layout_agreed = false;                    // the ranks have compared their message layout since the tables were last laid out (tracer_handshake)
  int *d_hs = nullptr, *h_hs = nullptr;          // [world + 1][4]: this rank's layout word quadruple, then every peer's
  unsigned *d_overflow = nullptr;
  std::vector<void *> send_buf, recv_buf;
  std::vector<size_t> send_cap, recv_cap;
  hipEvent_t ev_compute = nullptr, ev_report = nullptr, ev_pack = nullptr, ev_recv = nullptr, ev_comm = nullptr;
  hipEvent_t ev_chain0 = nullptr, ev_ann0 = nullptr, ev_pay0 = nullptr, ev_comp0 = nullptr; // phase timing (multi-rank frames)
  bool chain_timed = false, payload_timed = false;
  uint64_t last_local_pending = 0;
  std::vector<size_t> present; // host-known queue sizes as of the last report
  int long_cur = 0;            // the parking threshold this tracer's frames run with (0: Knobs::long_steps), adapted frame by frame (long_auto)
  // finish_auto (one rank, several instances): small rounds through k_finish or through per-hop merged chains, decided by timing
  uint64_t frame_no = 0;
  // ... and hops (rays without a hit taken on into the next local instance inside the merged launches, Knobs::hop_local) or the next round: a frame's ROUTE =
  // k_finish (0 / 1) + 2 x hops (0 never, 1 while the launch still hands out rays, 2 always)
  int fin_choice = -1;         // -1: probing (frames take the routes in turn), else the route
  int fin_limit = 0;           // the frame in progress: rounds of at most this many rays go through k_finish
  int hop_now = 0;             // the frame in progress: hops (TraceParams::hop)
  bool fin_eligible = false;   // the frame in progress had a round small enough for k_finish
  unsigned fin_probe = 0;
  double fin_best[6] = { 1e30, 1e30, 1e30, 1e30, 1e30, 1e30 }; // fastest frame seen with each route, ms
  int fin_n[6] = { 0, 0, 0, 0, 0, 0 };
  bool surfaces = false;       // every mesh of the scene is a surface by the builder's statistic (gvt_hip_mesh::packet_ok): the guess for hops where they are not timed
};

extern "C" void gvt_hip_tracer_destroy(gvt_hip_tracer *R) {
  if (!R) return;
  if (getenv("GVT_HIP_SPEC_TRACE") && R->n_exchanges)
    fprintf(stderr, "[tracer rank %d] %llu announce exchanges, speculative part enqueued behind %llu, valid %llu\n", R->rank, (unsigned long long)R->n_exchanges,
            (unsigned long long)R->n_spec_enqueued, (unsigned long long)R->n_spec_valid);
  if (getenv("GVT_HIP_SPEC_TRACE") && R->tick_n[0]) {
    fprintf(stderr, "[tracer rank %d] mean us from frame start / previous announce to the announce of tick:", R->rank);
    for (int k = 0; k < 8 && R->tick_n[k]; k++) fprintf(stderr, " %d: %.1f", k, R->tick_us[k] / (double)R->tick_n[k]);
    fprintf(stderr, "; last announce to frame end %.1f\n", R->comp_us / (double)R->tick_n[0]);
  }
  if (gctx().ready) hipStreamSynchronize(gctx().stream);
  if (R->comm && !R->comm->dead) hipStreamSynchronize(R->comm->stream);
  for (auto q : R->queues) gvt_hip_queue_destroy(q);
  gvt_hip_queue_destroy(R->q_moved);
  hipFree(R->d_insts); hipFree(R->d_round); hipHostFree(R->h_round); hipFree(R->d_count_ptr); hipFree(R->d_owner);
  hipFree(R->d_ann_out); hipFree(R->d_ann_in); hipHostFree(R->h_ann_in); // (d_report / h_report live behind the announces)
  hipFree(R->d_overflow); hipFree(R->d_hs); hipHostFree(R->h_hs); hipFree(R->d_spec); hipFree(R->d_spec_segs);
  for (void *p : R->send_buf) hipFree(p);
  for (void *p : R->recv_buf) hipFree(p);
  for (hipEvent_t e : { R->ev_compute, R->ev_report, R->ev_pack, R->ev_recv, R->ev_comm, R->ev_chain0, R->ev_ann0, R->ev_pay0, R->ev_comp0 }) if (e) hipEventDestroy(e);
  delete R;
}

static int tracer_alloc_tables(gvt_hip_tracer *R) {
  const size_t n = R->n_inst ? R->n_inst : 1, W = (size_t)R->world;
  const size_t row = ANN_HEAD + R->n_inst;
  hipFree(R->d_ann_out); hipFree(R->d_ann_in); hipHostFree(R->h_ann_in);
  R->d_ann_out = R->d_ann_in = R->h_ann_in = nullptr;
  R->inl_bytes = W > 1 ? ((size_t)gctx().inline_kb << 10) : 0;
  R->msg_ints = row + R->inl_bytes / 4;
  // the peers' messages as received and, behind them, this rank's report: one device block; the pinned mirror holds the rows and the report
  const size_t dview = sizeof(int) * W * R->msg_ints + sizeof(unsigned) * (n + REPORT_TAIL), hview = sizeof(int) * W * row + sizeof(unsigned) * (n + REPORT_TAIL);
  HIPCHK(hipMalloc((void **)&R->d_ann_out, sizeof(int) * W * R->msg_ints));
  HIPCHK(hipMalloc((void **)&R->d_ann_in, dview));
  HIPCHK(hipHostMalloc((void **)&R->h_ann_in, hview, hipHostMallocDefault));
  HIPCHK(hipMemset(R->d_ann_out, 0, sizeof(int) * W * R->msg_ints));
  HIPCHK(hipMemset(R->d_ann_in, 0, dview));
  std::memset(R->h_ann_in, 0, hview);
  R->d_report = (unsigned *)(R->d_ann_in + W * R->msg_ints);
  R->h_report = (unsigned *)(R->h_ann_in + W * row);
  R->report_seq = 0;
  R->layout_agreed = false;
  hipFree(R->d_hs); hipHostFree(R->h_hs);
  R->d_hs = R->h_hs = nullptr;
  if (W > 1) {
    HIPCHK(hipMalloc((void **)&R->d_hs, sizeof(int) * 4 * (W + 1)));
    HIPCHK(hipHostMalloc((void **)&R->h_hs, sizeof(int) * 4 * (W + 1), hipHostMallocDefault));
  }
  HIPCHK(hipStreamSynchronize(nullptr)); // the memsets above run on the null stream and may return before they are done; the frame's streams do not wait for it (non-blocking)
  for (void *p : R->send_buf) hipFree(p);
  for (void *p : R->recv_buf) hipFree(p);
  R->send_buf.assign(W, nullptr); R->recv_buf.assign(W, nullptr);
  R->send_cap.assign(W, 0); R->recv_cap.assign(W, 0);
  return 0;
}

extern "C" gvt_hip_tracer *gvt_hip_tracer_create(gvt_hip_top *T, gvt_hip_mesh *const *meshes, const float *m, const float *minv, const float *normi,
                                                 size_t n_inst, const gvt_hip_light *lights, size_t n_lights, int normal_mode, const gvt_hip_camera *cam,
                                                 gvt_hip_fb *fb) {
  if (ensure_init()) return nullptr;
  if (!T || !cam || !fb || (n_inst && (!meshes || !m || !minv || !normi)) || T->n != n_inst || (n_lights && !lights) || n_lights > 64) {
    set_error("tracer_create: null or inconsistent argument");
    return nullptr;
  }
  Ctx &C = gctx();
  gvt_hip_tracer *R = new gvt_hip_tracer();
  R->top = T; R->n_inst = n_inst; R->normal_mode = normal_mode; R->cam = *cam; R->fb = fb;
  R->meshes.assign(meshes, meshes + n_inst);
  R->m.assign(m, m + 16 * n_inst); R->minv.assign(minv, minv + 16 * n_inst); R->normi.assign(normi, normi + 9 * n_inst);
  R->lights.assign(lights, lights + n_lights);
  R->owner.assign(n_inst, 0); R->owned.assign(n_inst, 1); R->present.assign(n_inst, 0);
  bool ok = true;
  for (size_t i = 0; i < n_inst && ok; i++) { R->queues.push_back(gvt_hip_queue_create(0)); ok = R->queues.back() != nullptr; }
  R->q_moved = gvt_hip_queue_create(0);
  ok = ok && R->q_moved;
  const size_t n1 = n_inst ? n_inst : 1;
  std::vector<WaveInst> insts(n1);
  std::vector<unsigned *> cptr(n1, nullptr);
  for (size_t i = 0; i < n_inst && ok; i++) {
    gvt_hip_mesh *M = meshes[i];
    WaveInst &I = insts[i];
    std::memset(&I, 0, sizeof I);
    std::memcpy(I.minv.m, minv + 16 * i, 64);
    std::memcpy(I.normi.n, normi + 9 * i, 36);
    cptr[i] = R->queues[i]->d_count;
    if (!M) continue; // an instance whose data lives on another rank (Domain scheduler): never traced here
    if (!M->d_nodes4 && M->nNodes) ok = build_nodes4(M) == 0; // the merged kernels traverse the 4-wide layout
    // the cluster layout for k_finish (small rounds: several instances or ranks); knob finish_clusters = 0: the plain 4-wide nodes
    if (ok && C.finish_clusters && C.finish_rays > 0 && n_inst > 1 && M->d_nodes4) ok = build_nodes4c(M) == 0; // (once per mesh, under its own lock)
    I.nodes4 = M->d_nodes4; I.tris = M->d_tri; I.nodes4q = M->d_nodes4q; I.trisq = M->d_triq;
    I.nodes4c = C.finish_clusters ? M->d_nodes4c : nullptr; I.root_entry4c = M->root_entry4c;
    if (M->nNodes && !(M->d_nodes4q && M->d_triq)) R->all_quad = false;
    I.mv.slots = M->d_tri; I.mv.slot_of = M->d_slot_of; I.mv.verts = M->d_verts; I.mv.tris = M->d_tris; I.mv.normals = M->d_normals; I.mv.vcolors = M->d_vcolors;
    I.mv.materials = M->d_materials; I.mv.n_mat = (unsigned)M->nMat; I.mv.face_mat = M->d_face_mat; I.mv.mat = M->mesh_mat;
  }
  R->round_bytes = (sizeof(WaveSeg) + sizeof(QueueDesc)) * n1 + ((n1 + 15) & ~(size_t)15);
  ok = ok && hipMalloc((void **)&R->d_insts, sizeof(WaveInst) * n1) == hipSuccess && hipMalloc(&R->d_round, R->round_bytes) == hipSuccess &&
       hipHostMalloc(&R->h_round, R->round_bytes, hipHostMallocDefault) == hipSuccess &&
       hipMalloc((void **)&R->d_count_ptr, sizeof(unsigned *) * n1) == hipSuccess && hipMalloc((void **)&R->d_owner, sizeof(int) * n1) == hipSuccess &&
       hipMalloc((void **)&R->d_overflow, 64) == hipSuccess && hipMalloc((void **)&R->d_spec, 64) == hipSuccess &&
       hipMalloc((void **)&R->d_spec_segs, sizeof(WaveSeg) * n1) == hipSuccess;
  if (ok) {
    R->h_segs = (WaveSeg *)R->h_round; R->d_segs = (WaveSeg *)R->d_round;
    R->h_qdesc = (QueueDesc *)(R->h_segs + n1); R->d_qdesc = (QueueDesc *)(R->d_segs + n1);
    R->h_mask = (unsigned char *)(R->h_qdesc + n1); R->d_mask = (unsigned char *)(R->d_qdesc + n1);
    ok = hipMemcpy(R->d_insts, insts.data(), sizeof(WaveInst) * n1, hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(R->d_count_ptr, cptr.data(), sizeof(unsigned *) * n1, hipMemcpyHostToDevice) == hipSuccess &&
         hipMemset(R->d_owner, 0, sizeof(int) * n1) == hipSuccess && hipMemset(R->d_overflow, 0, 64) == hipSuccess && hipMemset(R->d_spec, 0, 64) == hipSuccess;
  }
  for (hipEvent_t *e : { &R->ev_pack })
    ok = ok && hipEventCreateWithFlags(e, hipEventDisableTiming) == hipSuccess;
  for (hipEvent_t *e : { &R->ev_compute, &R->ev_report, &R->ev_recv, &R->ev_comm, &R->ev_chain0, &R->ev_ann0, &R->ev_pay0, &R->ev_comp0 }) // timed: the frame's phase breakdown
    ok = ok && hipEventCreate(e) == hipSuccess;
  if (ok) ok = tracer_alloc_tables(R) == 0;
  if (!ok) { if (!*gvt_hip_last_error()) set_error("tracer_create: device allocation failed"); gvt_hip_tracer_destroy(R); return nullptr; }
  (void)C;
  return R;
}

extern "C" int gvt_hip_tracer_set_camera(gvt_hip_tracer *R, const gvt_hip_camera *cam) {
  if (!R || !cam) { set_error("tracer_set_camera: null"); return GVT_HIP_ERR_INVALID; }
  R->cam = *cam;
  return 0;
}

// mpiInstanceMap (DomainTracer.h:115-144): owner[i] = rank that holds instance i's data.  comm == NULL: one rank (Image scheduler).
extern "C" int gvt_hip_tracer_set_domains(gvt_hip_tracer *R, const int32_t *owner, gvt_hip_comm *comm) {
  if (!R) { set_error("tracer_set_domains: null"); return GVT_HIP_ERR_INVALID; }
  R->comm = comm;
  R->rank = comm ? comm->rank : 0;
  R->world = comm ? comm->world : 1;
  for (size_t i = 0; i < R->n_inst; i++) {
    const int o = owner ? owner[i] : 0;
    if (o < 0 || o >= R->world) { set_error("tracer_set_domains: owner[%zu] = %d of %d ranks", i, o, R->world); return GVT_HIP_ERR_INVALID; }
    R->owner[i] = o;
    R->owned[i] = o == R->rank ? 1 : 0;
    if (R->owned[i] && !R->meshes[i]) { set_error("tracer_set_domains: instance %zu is owned by this rank but has no mesh", i); return GVT_HIP_ERR_INVALID; }
  }
  if (R->n_inst) HIPCHK(hipMemcpy(R->d_owner, R->owner.data(), sizeof(int) * R->n_inst, hipMemcpyHostToDevice));
  return tracer_alloc_tables(R);
}

namespace {
// the host's view after a round: queue sizes, frame totals, error flags
struct Report {
  uint64_t rays_closest, rays_any;
};

// segments + queue descriptors + mask (one pinned block) to the device -- skipped when they equal what the device already holds
int round_tables_upload(gvt_hip_tracer *R, hipStream_t st) {
  if (R->round_uploaded.size() != R->round_bytes || std::memcmp(R->round_uploaded.data(), R->h_round, R->round_bytes) != 0) {
    HIPCHK(hipMemcpyAsync(R->d_round, R->h_round, R->round_bytes, hipMemcpyHostToDevice, st));
    R->round_uploaded.assign((const unsigned char *)R->h_round, (const unsigned char *)R->h_round + R->round_bytes);
  }
  return 0;
}

int grow(void **buf, size_t *cap, size_t bytes) {
  if (bytes <= *cap) return 0;
  if (*buf) HIPCHK(hipFree(*buf));
  *buf = nullptr; *cap = 0;
  const size_t want = bytes + bytes / 4 + 4096;
  HIPCHK(hipMalloc(buf, want));
  *cap = want;
  return 0;
}

// Capacity a speculative part needs WITHOUT any reallocation (nothing may move while it is in flight): this rank's own queues take what the peers send inline
// behind the next exchange, the outgoing ones what one k_finish launch can emit.  The host-driven paths reserve the same minima, so this normally holds.
static size_t spec_emit_bound(const gvt_hip_tracer *R) {
  const int passes = R->cam.depth > 1 ? R->cam.depth : 1;
  return (size_t)gctx().finish_rays * (size_t)(1 + (int)R->lights.size() * passes);
}
static bool spec_room(const gvt_hip_tracer *R) {
  const size_t inline_room = (size_t)(R->world - 1) * (R->inl_bytes / 80), emit = spec_emit_bound(R);
  for (size_t i = 0; i < R->n_inst; i++) {
    const gvt_hip_queue *Q = R->queues[i];
    if (Q->cap < (R->owned[i] ? 2 * inline_room : emit)) return false;
    if (R->h_qdesc[i].planes != Q->d_planes || R->h_qdesc[i].cap != Q->cap) return false; // (the device's table must be the current one: no upload from here)
  }
  return R->round_uploaded.size() == R->round_bytes && std::memcmp(R->round_uploaded.data(), R->h_round, R->round_bytes) == 0;
}

// (1) of a round: the merged launch chain over this rank's non-empty queues + the shuffle of everything that moved.  Host-known
// sizes in R->present; on return they are stale until the next report.  extra_in[i]: rays about to be appended to queue i by a
// pending unpack (room is reserved for them too).
int local_chain(gvt_hip_tracer *R, const std::vector<size_t> *extra_in, uint64_t *chains, bool fresh_from_camera, bool count_on_device = false,
                bool pass0_begun = false, bool defer_end = false) {
  Ctx &C = gctx();
  const size_t nI = R->n_inst;
  const int nL = (int)R->lights.size();
  const int passes = R->cam.depth > 1 ? R->cam.depth : 1;
  size_t N = 0;
  int n_seg = 0;
  for (size_t i = 0; i < nI; i++) {
    R->h_mask[i] = 0;
    if (!R->owned[i] || !R->present[i]) continue;
    WaveSeg &sg = R->h_segs[n_seg++];
    sg.planes = R->queues[i]->d_planes; sg.cap = R->queues[i]->cap; sg.begin = (unsigned)N; sg.n = (unsigned)R->present[i]; sg.inst = (int)i; sg.pad = 0;
    R->h_mask[i] = 1;
    N += R->present[i];
  }
  if (!N) return 0;
  const size_t bound = N * (size_t)(1 + nL * passes); // rays one round can emit: a forward or nL shadow rays per pass and input ray
  if (bound >= 0xffffffffull) { set_error("round: %zu rays exceed the 32-bit slot counters", bound); return GVT_HIP_ERR_INVALID; }
  int rc = queue_reserve(R->q_moved, bound);
  if (rc) return rc;
  // Every destination gets room for ALL rays the round can move (288 GB of HBM: worst-case room beats a read-back per shuffle) -- as
  // long as that fits the round's memory budget.  With hundreds of domains it does not (n_inst x bound x 68 bytes: 86 GB at 256
  // domains and 1080p): the round then reserves only what is known to arrive and shuffles with exact growth (two synchronisations).
  size_t growth = 0;
  for (size_t i = 0; i < nI; i++) {
    const size_t stay = R->h_mask[i] ? 0 : R->present[i], want = stay + bound + (extra_in ? (*extra_in)[i] : 0);
    if (want > R->queues[i]->cap) growth += (want - R->queues[i]->cap) * GVT_QUEUE_BYTES_PER_RAY;
  }
  const bool exact = growth > ((size_t)(C.round_room_mb > 0 ? C.round_room_mb : 1) << 20) && !count_on_device;
  const size_t inline_room = (R->world > 1 && R->owned.size()) ? (size_t)(R->world - 1) * (R->inl_bytes / 80) : 0; // what k_publish may append behind this tick's exchange
  const size_t spec_emit = (R->world > 1 && C.spec_ticks && R->inl_bytes) ? spec_emit_bound(R) : 0;                 // ... and what a speculative round behind it may emit (spec_room)
  for (size_t i = 0; i < nI; i++) {
    const size_t stay = R->h_mask[i] ? 0 : R->present[i];
    if ((rc = queue_reserve(R->queues[i], stay + std::max(exact ? (size_t)0 : bound, R->owned[i] ? (size_t)0 : spec_emit) + (extra_in ? (*extra_in)[i] : 0) + (R->owned[i] ? 2 * inline_room : 0)))) return rc;
  }
  R->sizes_exact = false;
  for (int k = 0; k < n_seg; k++) { // (a reserve above may have moved a traced queue)
    gvt_hip_queue *q = R->queues[R->h_segs[k].inst];
    R->h_segs[k].planes = q->d_planes; R->h_segs[k].cap = q->cap;
  }
  hipStream_t st = C.stream;
  for (size_t i = 0; i < nI; i++) { // the shuffle's destinations
    gvt_hip_queue *Q = R->queues[i];
    R->h_qdesc[i].planes = Q->d_planes; R->h_qdesc[i].cap = Q->cap; R->h_qdesc[i].count = Q->d_count; R->h_qdesc[i].keep = 1u;
  }
  // segments + descriptors + mask in one copy -- skipped when they equal what the device already holds (steady-state frames of a
  // one-queue scene: same pointers, same bound)
  for (size_t k = (size_t)n_seg; k < (nI ? nI : 1); k++) std::memset(&R->h_segs[k], 0, sizeof(WaveSeg));
  if ((rc = round_tables_upload(R, st))) return rc;
  int *d_from = (int *)scratch_get(17, sizeof(int) * bound);
  if (!d_from) return GVT_HIP_ERR_DEVICE;
  TraceParams P{};
  P.normal_mode = R->normal_mode; P.seed = 0; P.n_lights = nL; P.update_in_place = 0; P.carried_rng = 1;
  P.sink = TermSink{};
  if (C.term_sink) {
    P.sink.top = R->top->dev(); P.sink.from = -1;
    P.sink.fb = R->fb->d_rgba; P.sink.n_pix = (unsigned)(R->fb->w * R->fb->h);
  }
  // several instances on this rank: the merged launches take a ray that leaves one of them without a hit on into the next themselves (the frame's route:
  // gvt_hip_tracer_frame; knob hop_local; not with the known-miss shortcut, whose list is kept by the shuffle kernels)
  P.hop = R->hop_now == 1 ? 1 : R->hop_now == 2 ? 2 : 0;
  P.hop_owner = R->world > 1 ? R->d_owner : nullptr; P.hop_rank = R->rank;
  WaveSet W{ R->d_segs, R->d_insts, n_seg, R->all_quad ? 1 : 0, (int)nI };
  if (C.finish_rays > 0 && N <= (size_t)C.finish_rays && P.sink.fb && !count_on_device && !exact) R->fin_eligible = true;
  if (R->fin_limit > 0 && N <= (size_t)R->fin_limit && P.sink.fb && !count_on_device && !exact) {
    // a small round: ONE launch follows every ray to its end on this rank (finish_kernel.inc); what remains are rays in other ranks' queues
    if ((rc = finish_round(W, N, P, R->lights.data(), R->d_qdesc, R->d_owner, R->world > 1 ? R->rank : -1, R->d_overflow, R->d_count_ptr, R->d_mask))) return rc; // (+ the traced queues' clear())
    if (chains) (*chains)++;
    return 0;
  }
  WaveSingle one{};
  // (a frame that hops keeps to the merged lane kernels: a round with ONE non-empty queue -- the camera outside the scene: every ray enters the first slab -- would
  //  otherwise take the single-mesh kernels, which know no other instance, and leave all its misses to a second chain: the hall in 8 slabs 4.78 -> 4.50 ms)
  if (n_seg == 1 && R->meshes[R->h_segs[0].inst] && C.wave_single && !P.hop) {
    const int i0 = R->h_segs[0].inst;
    one.planes = make_planes(R->h_segs[0].planes, R->h_segs[0].cap);
    if (nI == 1) one.planes.p5 = nullptr; // no other instance a ray could have missed: the list is neither read nor written
    one.mesh = R->meshes[i0]; one.inst = i0;
    one.coherent = (fresh_from_camera && C.camera_tile == 8) ? 1 : 0;
    one.n_dev = count_on_device ? R->queues[i0]->d_count : nullptr; // present[i0] is then only the bound (the whole camera list)
    one.pass0_begun = pass0_begun ? 1 : 0;
    std::memcpy(one.minv.m, R->minv.data() + 16 * (size_t)i0, 64);
    std::memcpy(one.normi.n, R->normi.data() + 9 * (size_t)i0, 36);
  }
  const bool single = one.mesh != nullptr;
  // several queues straight from the camera filter: the segments' lengths and beginnings from the count words (in the chain's first kernel)
  const unsigned *n_dev0 = (count_on_device && !single) ? C.d_counters + 22 : nullptr;
  // several queues of camera rays in tile order, every traced mesh packet-friendly (or packets forced): the merged closest-hit launch walks packets too
  // (launches of a few hundred thousand rays over small meshes lose with packets -- 4 K waves, each a long serial walk: bunny.conf 0.243 -> 0.254 ms, the
  // 8-bunny grid 0.368 -> 0.407 -- where the hall cut into 8 slabs, 4.2 M camera rays, gains: 7.57 -> 7.25 ms; hence packet_min_rays)
  bool multi_packets = !single && !P.hop && fresh_from_camera && C.camera_tile == 8 && (C.packet == 2 || (C.packet == 1 && N >= (size_t)C.packet_min_rays));
  for (int k = 0; k < n_seg && multi_packets; k++) {
    const gvt_hip_mesh *Mk = R->meshes[R->h_segs[k].inst];
    multi_packets = Mk && Mk->d_nodes4 && (C.packet == 2 || Mk->packet_ok);
  }
  // every mesh of the round a plain LAMBERT one (no vertex colours, no per-face materials): with depth 1 and no area light the merged chain shades with k_shade's lean instantiation too
  bool simple_meshes = !single;
  for (int k = 0; k < n_seg && simple_meshes; k++) {
    const gvt_hip_mesh *Mk = R->meshes[R->h_segs[k].inst];
    simple_meshes = Mk && !Mk->d_vcolors && !Mk->d_face_mat && Mk->mesh_mat.type == 0;
  }
  if ((rc = wave_trace_chain(W, N, passes, R->q_moved, d_from, P, R->lights.data(), single ? &one : nullptr, R->d_count_ptr, R->d_mask, (int)nI, defer_end && single, n_dev0, multi_packets, simple_meshes))) return rc;
  // one instance in the whole scene and the terminal rule applied inside the kernels: nothing can have moved
  if (exact) {
    if ((rc = shuffle_exact(R->top, R->q_moved, single ? nullptr : d_from, single ? one.inst : -1, R->queues.data(), R->fb))) return rc;
  } else if (!(nI == 1 && P.sink.fb) &&
             (rc = shuffle_async(R->top, R->q_moved, bound, single ? nullptr : d_from, single ? one.inst : -1, R->queues.data(), nullptr, R->fb, R->d_overflow, R->d_qdesc))) return rc;
  if (chains) (*chains)++;
  return 0;
}

// (3)+(4)+(5): report kernel -> [announce exchange] -> ONE device-to-host copy -> ONE host synchronisation
// skip_kernel: the speculative part enqueued behind the previous exchange was valid -- this tick's round and its k_round_report have already run on the device.
// allow_spec: enqueue the NEXT tick's speculative part behind this exchange; *spec_valid: it was enqueued and the exchange left it valid.
int round_report(gvt_hip_tracer *R, bool exchange, uint64_t *syncs, bool chain_end = false, int err_code = 0, int tick = 0, gvt_hip_frame_stats *S = nullptr,
                 bool skip_kernel = false, bool allow_spec = false, bool *spec_valid = nullptr) {
  Ctx &C = gctx();
  if (spec_valid) *spec_valid = false;
  const size_t nI = R->n_inst, row = ANN_HEAD + nI;
  hipStream_t st = C.stream;
  int *d_bbox = (int *)(R->d_overflow + 4);
  const bool poll = !(exchange && R->world > 1) && C.report_poll;
  const bool timing = C.frame_timing != 0; // the per-phase breakdown of gvt_hip_frame_stats costs five event calls per tick: on request only
  int rc_wait = 0;
  // exchanging ranks: the rectangle this rank's deposits lie in (for the composite, carried by the announce) is folded into the same
  // launch -- many blocks scan the framebuffer, the last one to finish writes the report
  const bool scan = exchange && R->world > 1;
  const unsigned n_blk = scan ? (unsigned)std::min<size_t>(1024, ((size_t)R->fb->w * R->fb->h + 1023) / 1024) : 1u;
  if (scan && R->inl_bytes && !skip_kernel) { // the inline pack reads the outgoing queues through the descriptor table: current? (a reserve may have moved one since the last chain)
    // ... and k_publish appends what arrives inline to the owned queues: room for every peer's full inline area, beyond what they hold (a tick
    // in which a chain ran has it already: local_chain reserves the same slack on top of the round's bound) -- twice over, and the outgoing queues room for
    // one k_finish launch's emissions: what a speculative part behind this exchange needs without reallocating (spec_room)
    const size_t inline_room = (size_t)(R->world - 1) * (R->inl_bytes / 80);
    for (size_t i = 0; i < nI; i++)
      if (R->owned[i]) { int rc_q = queue_reserve(R->queues[i], R->queues[i]->size + inline_room); if (rc_q) return rc_q; }
    // (what a speculative part behind this exchange needs beyond that -- spec_room -- is reserved by local_chain, or here when no chain has run since the last
    //  report: only then are the host's sizes the device's, and a reallocation copies what the host knows of)
    if (C.spec_ticks && R->sizes_exact) {
      const size_t emit = spec_emit_bound(R);
      for (size_t i = 0; i < nI; i++) {
        int rc_q = queue_reserve(R->queues[i], R->queues[i]->size + (R->owned[i] ? 2 * inline_room : emit));
        if (rc_q) return rc_q;
      }
    }
    for (size_t i = 0; i < nI; i++) { gvt_hip_queue *Q = R->queues[i]; R->h_qdesc[i].planes = Q->d_planes; R->h_qdesc[i].cap = Q->cap; R->h_qdesc[i].count = Q->d_count; R->h_qdesc[i].keep = 1u; }
    int rc_up = round_tables_upload(R, st);
    if (rc_up) return rc_up;
  }
  if (!skip_kernel) {
    k_round_report<<<n_blk, 256, 0, st>>>(R->d_count_ptr, R->d_owner, (int)nI, R->rank, R->world, R->d_report, R->d_ann_out, C.d_counters, R->d_overflow,
                                          R->d_report + nI, d_bbox, chain_end ? 1 : 0, R->d_mask, poll ? R->h_report : nullptr, poll ? ++R->report_seq : 0u, err_code, tick,
                                          scan ? (const float4 *)R->fb->d_rgba : nullptr, R->fb->w, R->fb->h, (int)R->msg_ints, scan ? (int)R->inl_bytes : 0, R->d_qdesc,
                                          scan ? R->d_spec : nullptr, 0);
    HIPCHK(hipGetLastError());
  }
  if (exchange && R->world > 1) {
    gvt_hip_comm *K = R->comm;
    const size_t msg = sizeof(int) * R->msg_ints;
    if (K->stream != st || timing) HIPCHK(hipEventRecord(R->ev_compute, st)); // (also the timing event at the end of this tick's local chain)
    if (K->stream != st) HIPCHK(hipStreamWaitEvent(K->stream, R->ev_compute, 0));
    if (timing) HIPCHK(hipEventRecord(R->ev_ann0, K->stream));
    comm_group_begin(K);
    for (int p = 0; p < R->world; p++) {
      if (p == R->rank) continue;
      comm_send(K, R->d_ann_out + (size_t)p * R->msg_ints, msg, p);
      comm_recv(K, R->d_ann_in + (size_t)p * R->msg_ints, msg, p);
    }
    int rc = comm_group_end(K);
    if (rc) return rc;
    // behind the exchange ONE small kernel gathers the peers' rows and this rank's report into the pinned mirror and releases the
    // sequence word; the host waits for that word with loads (bounded: a peer that never joins the exchange must not hang this rank)
    const unsigned seq = ++R->report_seq;
    // The NEXT tick's speculative part goes behind this exchange, before the host has seen its result: k_spec_begin (segments from the count words) -> k_finish
    // (the small round) -> k_round_report (the next announce's rows + inline pack).  k_publish decides on the device whether it is valid -- no payload beyond the
    // inline area either way, no error, at most finish_rays local rays -- and tells the host in the report; a void part leaves no trace.  The sequence of
    // EXCHANGES is untouched (the next one is posted by the host as before): this is local to the rank, the peers cannot tell.
    const bool spec_now = allow_spec && C.spec_ticks && !timing && K->stream == st && R->inl_bytes > 0 && C.finish_rays > 0 && C.term_sink && !err_code &&
                          finish_lights_resident(R->lights.data(), (int)R->lights.size()) && spec_room(R);
    k_publish<<<(unsigned)R->world, 256, 0, K->stream>>>(R->d_ann_in, (int)R->msg_ints, (int)row, R->world, R->rank, R->d_report, (int)(nI + REPORT_TAIL - 1), R->h_ann_in, R->h_report,
                                                         seq, R->inl_bytes ? R->d_qdesc : nullptr, (int)nI, R->d_overflow, R->d_overflow + 12,
                                                         spec_now ? R->d_spec : nullptr, R->d_owner, (unsigned)(C.finish_rays > 0 ? C.finish_rays : 0), R->d_spec_segs);
    HIPCHK(hipGetLastError());
    if (spec_now) {
      int n_own = 0;
      for (size_t i = 0; i < nI; i++) n_own += R->owned[i] ? 1 : 0;
      TraceParams P{};
      P.normal_mode = R->normal_mode; P.seed = 0; P.n_lights = (int)R->lights.size(); P.update_in_place = 0; P.carried_rng = 1;
      P.sink = TermSink{};
      P.sink.top = R->top->dev(); P.sink.from = -1; P.sink.fb = R->fb->d_rgba; P.sink.n_pix = (unsigned)(R->fb->w * R->fb->h);
      WaveSet W{ R->d_spec_segs, R->d_insts, n_own, R->all_quad ? 1 : 0, (int)nI };
      // (the launch's grid: persistent waves pull rays from a counter, any size is correct; sized for twice what this rank's rounds have held lately, so that
      //  ranks sharing one device -- in-process ranks -- do not each claim all of it for a few hundred rays)
      const size_t grid_rays = std::min<size_t>((size_t)C.finish_rays, std::max<size_t>(1024, 2 * R->spec_recent_rays));
      int rc_s = n_own ? finish_round(W, grid_rays, P, R->lights.data(), R->d_qdesc, R->d_owner, R->rank, R->d_overflow, nullptr, nullptr, R->d_spec) : 0;
      if (rc_s) return rc_s;
      k_round_report<<<n_blk, 256, 0, st>>>(R->d_count_ptr, R->d_owner, (int)nI, R->rank, R->world, R->d_report, R->d_ann_out, C.d_counters, R->d_overflow,
                                            R->d_report + nI, d_bbox, 0, R->d_mask, nullptr, 0u, 0, tick + 1, (const float4 *)R->fb->d_rgba, R->fb->w, R->fb->h,
                                            (int)R->msg_ints, (int)R->inl_bytes, R->d_qdesc, R->d_spec, 1);
      HIPCHK(hipGetLastError());
    }
    R->spec_enqueued_last = spec_now;
    R->n_exchanges++; R->n_spec_enqueued += spec_now ? 1 : 0;
    if (timing || K->stream != st) HIPCHK(hipEventRecord(R->ev_report, K->stream));
    // The host leaves the wait below when block `rank` of k_publish has released the sequence word; the blocks that append the INLINE payloads to the owned
    // queues and advance their count words may still run.  On the compute stream the next chain is ordered behind them anyway; with the exchange on the
    // communicator's own stream (knob comm_stream / GVT_HIP_COMM_STREAM) the compute stream is told to wait for the whole of k_publish.
    if (K->stream != st) HIPCHK(hipStreamWaitEvent(st, R->ev_report, 0));
    {
      char diag[256];
      std::snprintf(diag, sizeof diag, "exchange %d, this rank's announce: %zu local rays pending, error word %d; last report: %u rays in the first queue", tick,
                    (size_t)R->last_local_pending, err_code, nI ? R->h_report[0] : 0u);
      if ((rc_wait = bounded_flag_wait(K, R->h_report + nI + REPORT_TAIL - 1, seq, K->stream, "the announce exchange", diag, true))) return rc_wait;
    }
    if (S && timing) { // phase times of this tick
      float ms = 0.f;
      hipEventSynchronize(R->ev_report);
      if (R->chain_timed && hipEventElapsedTime(&ms, R->ev_chain0, R->ev_compute) == hipSuccess) S->ms_chain += ms;
      if (hipEventElapsedTime(&ms, R->ev_ann0, R->ev_report) == hipSuccess) S->ms_announce += ms;
      if (R->payload_timed && hipEventElapsedTime(&ms, R->ev_pay0, R->ev_recv) == hipSuccess) S->ms_payload += ms;
      R->chain_timed = false; R->payload_timed = false;
    }
  } else if (poll) {
    // k_round_report wrote the report into pinned host memory and released the sequence word last: wait for it with loads (no
    // interrupt, no wake-up latency); a device that stops answering is caught by the stream synchronisation behind the deadline
    volatile unsigned *flag = R->h_report + nI + REPORT_TAIL - 1;
    const auto t0 = std::chrono::steady_clock::now();
    bool seen = false;
    for (unsigned spins = 0;; spins++) {
      if (__atomic_load_n((const unsigned *)flag, __ATOMIC_ACQUIRE) == R->report_seq) { seen = true; break; }
      if ((spins & 1023u) == 1023u && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(200)) break;
    }
    if (!seen) {
      HIPCHK(hipStreamSynchronize(st));
      if (__atomic_load_n((const unsigned *)flag, __ATOMIC_ACQUIRE) != R->report_seq) { set_error("round report: the device finished without publishing report %u", R->report_seq); return GVT_HIP_ERR_DEVICE; }
    }
  } else {
    HIPCHK(hipMemcpyAsync(R->h_report, R->d_report, sizeof(unsigned) * (nI + REPORT_TAIL - 1), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
  }
  if (syncs) (*syncs)++;
  if (spec_valid && exchange && R->world > 1) { *spec_valid = R->spec_enqueued_last && R->h_report[nI + 12] != 0u; R->n_spec_valid += *spec_valid ? 1 : 0; }
  for (size_t i = 0; i < nI; i++) { R->present[i] = R->h_report[i]; R->queues[i]->size = R->h_report[i]; }
  R->sizes_exact = !(spec_valid && *spec_valid); // (a valid speculative part is a chain in flight)
  R->q_moved->size = 0;
  const unsigned *tail = R->h_report + nI;
  if (tail[4]) { set_error("BVH traversal stack overflow: results of this frame are incomplete"); return GVT_HIP_ERR_DEVICE; }
  if (tail[5] & 1u) { set_error("a ray queue ran out of room inside a round (internal reservation error)"); return GVT_HIP_ERR_CAPACITY; }
  if (tail[5] & 2u) { set_error("ray exchange: an in-band queue header disagrees with the announced counts"); return GVT_HIP_ERR_DEVICE; }
  return 0;
}
} // namespace

// The announce message of a tick has a FIXED length on every rank: row (ANN_HEAD + n_inst ints) + the inline area (knob inline_kb of the rank's context when its
// tables were laid out).  Ranks that disagree would post sends and receives of different sizes: the in-process transport reports that, RCCL hangs until the
// deadline or corrupts the row.  So the first frame behind every (re)layout starts with ONE exchange of 16 bytes per pair -- {magic, inline bytes,
// instances, ABI revision}, a size that cannot disagree -- and a rank that finds a peer laid out differently leaves with GVT_HIP_ERR_INVALID (every rank sees
// the same disagreement: nobody is left inside the frame's exchanges).  Under RCCL this is also where the pairs connect (the long allowance of bounded_event_wait).
static int tracer_handshake(gvt_hip_tracer *R) {
  gvt_hip_comm *K = R->comm;
  const int W = R->world;
  if (!K || W < 2 || R->layout_agreed) return 0;
  if (K->dead) { set_error("ray exchange: the communicator of rank %d was aborted after a deadline passed", K->rank); return GVT_HIP_ERR_TIMEOUT; }
  int *mine = R->h_hs;
  mine[0] = 0x47565448; mine[1] = (int)R->inl_bytes; mine[2] = (int)R->n_inst; mine[3] = GVT_HIP_ABI_VERSION;
  HIPCHK(hipMemcpyAsync(R->d_hs, mine, 16, hipMemcpyHostToDevice, K->stream));
  comm_group_begin(K);
  for (int p = 0; p < W; p++) {
    if (p == R->rank) continue;
    comm_send(K, R->d_hs, 16, p);
    comm_recv(K, R->d_hs + 4 * (p + 1), 16, p);
  }
  int rc = comm_group_end(K);
  if (rc) return rc;
  HIPCHK(hipMemcpyAsync(R->h_hs + 4, R->d_hs + 4, sizeof(int) * 4 * W, hipMemcpyDeviceToHost, K->stream));
  HIPCHK(hipEventRecord(K->ev_wait, K->stream));
  if ((rc = bounded_event_wait(K, K->ev_wait, "the layout handshake of the frame's first exchange", nullptr, true))) return rc;
  for (int p = 0; p < W; p++) {
    if (p == R->rank) continue;
    const int *h = R->h_hs + 4 * (p + 1);
    if (h[0] != mine[0] || h[1] != mine[1] || h[2] != mine[2] || h[3] != mine[3]) {
      set_error("ray exchange: rank %d is laid out differently from rank %d: inline area %d vs %d bytes (knob inline_kb must be the same on every rank), %d vs %d instances, "
                "ABI revision %d vs %d%s", p, R->rank, h[1], mine[1], h[2], mine[2], h[3], mine[3], h[0] != mine[0] ? " (no handshake word: another protocol)" : "");
      return GVT_HIP_ERR_INVALID;
    }
  }
  R->layout_agreed = true;
  return 0;
}

extern "C" int gvt_hip_tracer_frame(gvt_hip_tracer *R, int flags, gvt_hip_frame_stats *out) {
  if (ensure_init()) return GVT_HIP_ERR_NODEVICE;
  if (!R) { set_error("tracer_frame: null"); return GVT_HIP_ERR_INVALID; }
  Ctx &C = gctx();
  hipStream_t st = C.stream;
  const size_t nI = R->n_inst, row = ANN_HEAD + nI;
  const bool bsp = (flags & GVT_HIP_FRAME_BSP) != 0;
  // Tracer<ImageScheduler> on several ranks (ImageTracer.h:111-125): the scene is replicated, every rank takes a contiguous portion of
  // the camera's ray list and traces it to the end on its own; no ray ever changes rank; the frame ends with the composite.
  const bool image_split = (flags & GVT_HIP_FRAME_IMAGE) != 0 && R->world > 1;
  int rc0 = 0;
  if (image_split)
    for (size_t i = 0; i < nI; i++)
      if (!R->meshes[i]) { set_error("tracer_frame: GVT_HIP_FRAME_IMAGE needs every instance's mesh on every rank (instance %zu has none)", i); return GVT_HIP_ERR_INVALID; }
  if (R->world > 1 && R->inl_bytes != ((size_t)C.inline_kb << 10)) { // the knob moved since the tables were laid out (every rank must move it alike)
    HIPCHK(hipStreamSynchronize(st));
    if ((rc0 = tracer_alloc_tables(R))) return rc0;
  }
  if ((rc0 = tracer_handshake(R))) return rc0;
  if (R->comm) { R->comm->ms_host_wait = 0.0; R->comm->groups = 0; } // (the frame's own counts: the once-per-layout handshake is not one of its ticks)
  const int world_saved = R->world;
  std::vector<uint8_t> owned_saved = R->owned;
  struct Restore { // whatever path leaves this function: the tracer is a rank of its communicator again
    gvt_hip_tracer *R; int world; const std::vector<uint8_t> *owned;
    ~Restore() { R->world = world; R->owned = *owned; }
  } restore{ R, world_saved, &owned_saved };
  gvt_hip_frame_stats S{};
  int rc;
  // closest hit: the node-step count after which a ray is parked for a whole wave (Knobs::long_steps) suits the benchmark's density; a
  // sparser scene has more rays beyond it (1 M-triangle soup: 4 % of the rays, k_long_closest 0.25 ms of a 1.27 ms frame).  long_auto:
  // the tracer raises the threshold from frame to frame while more than 0.3 % of a frame's closest-hit rays were parked, and lets it
  // fall back towards the knob when fewer than 0.05 % were.  Results never depend on it.
  // small rounds: k_finish, or per-hop chains -- on one rank with several instances whichever a few timed frames say is faster here (finish_auto)
  const auto frame_t0 = std::chrono::steady_clock::now();
  // ... and whether a ray that leaves an instance without a hit goes on into the next local one inside the launch -- hops: never, early (the closest-hit launch only,
  // before its drain) or always: the hall cut into 8 slabs 11.5 -> 4.3 ms with "always", the soup tiles 1.46 -> 1.34 with "early" and 1.75 with "always" (their grazing
  // rays stretch a launch's tail), bunny.conf best without -- timed like finish_auto on one rank (profiles/r06_hops.txt), by the meshes' kind on several
  if (R->frame_no == 0) {
    R->surfaces = true;
    for (size_t i = 0; i < nI; i++) if (R->meshes[i] && !R->meshes[i]->packet_ok) R->surfaces = false;
  }
  size_t n_mine = 0; // (a rank with one instance has nowhere to hop to)
  for (size_t i = 0; i < nI && n_mine < 2; i++) n_mine += (R->world == 1 || image_split || R->owned[i]) ? 1 : 0;
  const bool hop_ok = C.hop_local > 0 && C.term_sink && !C.skip_known && n_mine > 1;
  const bool one_rank_many = R->world == 1 && nI > 1;
  unsigned allowed = 0; // bit v: route v (v = k_finish + 2 x hop mode) may be taken
  for (int v = 0; v < 6; v++) {
    const bool f = (v & 1) != 0;
    const int h = v >> 1; // 0 never, 1 early, 2 always
    if (f ? C.finish_rays <= 0 : (C.finish_rays > 0 && !(C.finish_auto && one_rank_many))) continue;
    if (!hop_ok) { if (h) continue; }
    else if (C.hop_local == 2) { if (h != 2) continue; }
    else if (C.hop_local == 3) { if (h != 1) continue; }
    else if (!one_rank_many && h != (R->surfaces ? 2 : 0)) continue; // (several ranks: not timed)
    allowed |= 1u << v;
  }
  int n_allowed = 0, route_list[6] = { 0, 0, 0, 0, 0, 0 };
  for (int v = 0; v < 6; v++) if ((allowed >> v) & 1u) route_list[n_allowed++] = v;
  const bool fin_auto = n_allowed > 1;
  // (the first frames and the un-timed case: k_finish where it is on, hops where the meshes are surfaces)
  int fin_variant = route_list[n_allowed - 1];
  {
    const int h0 = !hop_ok ? 0 : C.hop_local == 2 ? 2 : C.hop_local == 3 ? 1 : R->surfaces ? 2 : 0, f0 = C.finish_rays > 0 ? 1 : 0;
    for (int k = 0; k < n_allowed; k++) if ((route_list[k] >> 1) == h0 && (route_list[k] & 1) == f0) fin_variant = route_list[k];
  }
  if (fin_auto && R->frame_no >= 2) {
    if ((R->frame_no & 2047u) == 0 || (R->fin_choice >= 0 && !((allowed >> R->fin_choice) & 1u))) { // look again now and then (or the knobs have moved)
      R->fin_choice = -1;
      for (int v = 0; v < 6; v++) { R->fin_n[v] = 0; R->fin_best[v] = 1e30; }
    }
    if (R->fin_choice >= 0) fin_variant = R->fin_choice;
    else { // probing: the allowed route with the fewest timed frames so far (the guess above first among equals)
      int pick = fin_variant;
      for (int k = 0; k < n_allowed; k++) if (R->fin_n[route_list[k]] < R->fin_n[pick]) pick = route_list[k];
      fin_variant = pick;
    }
  }
  R->fin_limit = (fin_variant & 1) ? C.finish_rays : 0;
  R->hop_now = fin_variant >> 1;
  R->fin_eligible = false;
  struct LongOverride { Ctx &C; ~LongOverride() { C.long_steps_override = 0; } } long_override{ C };
  C.long_steps_override = (C.long_auto && C.long_steps > 0 && R->long_cur > C.long_steps) ? R->long_cur : 0;
  // clearBuffer + generateRays + FilterRaysLocally / shuffleDropRays (ImageTracer.h:137-146, DomainTracer.h:148-183, 204-211)
  // One instance, one rank, terminal rule inside the kernels: the first (and only) launch chain takes its ray count from the queue's
  // count word on the device -- the camera filter needs no read-back and the frame has ONE host synchronisation.
  const bool one_shot = nI == 1 && R->world == 1 && C.term_sink && C.wave_single && R->meshes[0] && C.first_round_async;
  // ... and in eight launches: the camera filter's two kernels also clear the framebuffer and do the resets
  // (k_zero_totals, k_wave_pass_begin), the round's report does k_wave_end's work
  const bool lean = one_shot && C.lean_frame && R->fb->w == R->cam.width && R->fb->h == R->cam.height; // (the lean filter clears the camera's pixels only)
  for (size_t i = 0; i < nI; i++) R->queues[i]->size = 0;
  R->q_moved->size = 0;
  R->sizes_exact = true; R->spec_enqueued_last = false;
  bool first_on_device = false; // the queues hold the camera's rays, their sizes are on the device only (R->present: bounds)
  if (!lean) {
    if ((rc = gvt_hip_fb_clear(R->fb))) return rc;
    k_zero_totals<<<(unsigned)std::max<size_t>(1, (nI + 255) / 256), 256, 0, st>>>(C.d_counters, R->d_overflow, R->fb->w, R->fb->h, R->d_count_ptr, (int)nI); // + every queue.clear()
  }
  if (one_shot) {
    const size_t n_cam = (size_t)R->cam.width * R->cam.height * R->cam.samples * R->cam.samples;
    const int passes0 = R->cam.depth > 1 ? R->cam.depth : 1;
    if ((rc = queue_reserve(R->queues[0], n_cam * (size_t)(1 + (int)R->lights.size() * passes0)))) return rc; // what local_chain will ask for: no move later
    if (lean) { if ((rc = camera_one_instance_async(R->top, &R->cam, C.camera_tile, R->queues[0], R->fb, R->d_overflow, R->q_moved->d_count))) return rc; }
    else if ((rc = camera_filter_async(R->top, &R->cam, C.camera_tile, R->queues.data(), nullptr, R->d_overflow, 0, 0))) return rc;
    R->present[0] = n_cam;
    R->queues[0]->size = n_cam; // bound of what the device holds (a reallocation would copy at least that)
    if ((rc = local_chain(R, nullptr, &S.chains, true, true, lean, lean))) return rc;
    if ((rc = round_report(R, false, &S.host_syncs, lean))) return rc;
  } else if (image_split) {
    const size_t n_cam = (size_t)R->cam.width * R->cam.height * R->cam.samples * R->cam.samples;
    const size_t portion = n_cam / (size_t)R->world, first = (size_t)R->rank * portion;
    const size_t count = (R->rank + 1 == R->world) ? n_cam - first : portion; // "tack on any odd rays to last proc" (:116)
    for (size_t i = 0; i < nI; i++) if ((rc = queue_reserve(R->queues[i], count))) return rc;
    if ((rc = camera_filter_async(R->top, &R->cam, C.camera_tile, R->queues.data(), nullptr, R->d_overflow, first, count))) return rc;
    std::fill(R->owned.begin(), R->owned.end(), (uint8_t)1);
    R->world = 1; // the rounds below are a one-rank frame
    if ((rc = round_report(R, false, &S.host_syncs))) return rc;
  } else {
    // A few instances: the first chain is launched straight behind the camera filter -- every queue that can receive camera rays gets
    // room for the positions of the film rectangle its box projects onto, the filter advances the count words on the device only, the
    // chain's segments take their lengths from them (k_seg_begins); the host learns the sizes from the chain's report.  One host
    // synchronisation less per frame (bunny.conf 0.265 -> 0.24 ms).  More instances, or rectangles that add up to several films: the
    // filter's own read-back, as before.
    const uint8_t *keep = R->world > 1 ? R->owned.data() : nullptr;
    const size_t n_cam = (size_t)R->cam.width * R->cam.height * R->cam.samples * R->cam.samples;
    bool on_device = C.first_round_async && C.term_sink && nI <= 16;
    std::vector<size_t> room(nI, 0);
    if (on_device) {
      size_t sum = 0;
      for (size_t i = 0; i < nI; i++) {
        if (keep && !keep[i]) continue;
        if (!R->meshes[i]) { on_device = false; break; }
        room[i] = camera_instance_bound(R->top, &R->cam, C.camera_tile, i);
        sum += room[i];
      }
      if (sum > 2 * n_cam) on_device = false;
    }
    if (on_device) {
      for (size_t i = 0; i < nI; i++) if (room[i] && (rc = queue_reserve(R->queues[i], room[i]))) return rc;
      if ((rc = camera_filter_async(R->top, &R->cam, C.camera_tile, R->queues.data(), keep, R->d_overflow, 0, 0, true))) return rc;
      for (size_t i = 0; i < nI; i++) { R->present[i] = room[i]; R->queues[i]->size = room[i]; } // bounds, until the first report
      first_on_device = true;
    } else {
      if ((rc = gvt_hip_camera_filter(R->top, &R->cam, C.camera_tile, R->queues.data(), keep))) return rc;
      S.host_syncs++;
      for (size_t i = 0; i < nI; i++) R->present[i] = R->queues[i]->size;
    }
  }
  std::vector<size_t> incoming(nI, 0);
  bool payload_pending = false, payload_cross = false; // ... and whether it moves on another stream than the compute stream
  std::vector<int> pending_ann; // the announces the payload in flight was posted from

  // ALL queues of a tick's payload through ONE pack / unpack launch (the items by value in the kernel arguments, <= WIRE_MAX per launch)
  auto wire_flush = [&](WireBatch &B, bool unpack) -> int {
    if (!B.n_items) return 0;
    if (unpack) k_unpack_all<<<(B.total + 255) / 256, 256, 0, st>>>(B, R->d_overflow, R->d_overflow + 11);
    else k_pack_all<<<(B.total + 255) / 256, 256, 0, st>>>(B);
    HIPCHK(hipGetLastError());
    B.n_items = 0; B.total = 0;
    return 0;
  };
  auto wire_add = [&](WireBatch &B, bool unpack, gvt_hip_queue *q, void *buf, unsigned n, int qid) -> int {
    if (B.n_items == WIRE_MAX || (unsigned long long)B.total + 2ull + 20ull * n > 0xf0000000ull) { int rc_ = wire_flush(B, unpack); if (rc_) return rc_; }
    WireItem &I = B.it[B.n_items];
    I.planes = q->d_planes; I.cap = q->cap; I.count = q->d_count; I.buf = (unsigned *)buf; I.n = n; I.qid = qid; I.dword0 = B.total; I.prior = 0u;
    if (unpack) for (int k = 0; k < B.n_items; k++) if (B.it[k].qid == qid) I.prior += B.it[k].n;
    B.n_items++; B.total += 2u + 20u * n;
    return 0;
  };
  auto unpack_pending = [&]() -> int { // (2): append what the last exchange delivered, behind its event
    if (!payload_pending) return 0;
    if (payload_cross) HIPCHK(hipStreamWaitEvent(st, R->ev_recv, 0)); // (the payload moved on the communicator's own stream)
    WireBatch B;
    B.n_items = 0; B.total = 0;
    int rc_;
    for (int p = 0; p < R->world; p++) {
      if (p == R->rank) continue;
      const int *a = pending_ann.data() + (size_t)p * row;
      if (a[10]) continue; // arrived inside the announce: k_publish appended it where it arrived
      char *src = (char *)R->recv_buf[p];
      size_t off = 0;
      for (size_t i = 0; i < nI; i++) {
        const unsigned n = (unsigned)a[ANN_HEAD + i];
        if (!n) continue;
        if ((rc_ = wire_add(B, true, R->queues[i], src + off, n, (int)i))) return rc_;
        off += 8 + 80ull * n;
      }
    }
    if ((rc_ = wire_flush(B, true))) return rc_;
    payload_pending = false;
    for (size_t i = 0; i < nI; i++) { R->present[i] += incoming[i]; R->queues[i]->size = R->present[i]; incoming[i] = 0; } // exact: announced per queue
    return 0;
  };

  int local_err = 0;
  std::string local_msg;
  // after an announce exchange: did every rank arrive at the same exchange number without an error?
  auto peers_ok = [&](int tick_) -> int {
    int bad_rank = -1, bad_code = 0;
    for (int p = 0; p < R->world; p++) {
      if (p == R->rank) continue;
      const int *a = R->h_ann_in + (size_t)p * row;
      if (a[9] != tick_) { set_error("ray exchange: rank %d is at exchange %d while rank %d is at %d (ranks out of step)", p, a[9], R->rank, tick_); return GVT_HIP_ERR_DEVICE; }
      if (a[8] && bad_rank < 0) { bad_rank = p; bad_code = a[8]; }
    }
    if (local_err) { set_error("%s", local_msg.c_str()); return local_err; } // (the peers leave with GVT_HIP_ERR_PEER)
    if (bad_rank >= 0) { set_error("ray exchange: rank %d reported error %d in its announce of exchange %d; this frame is abandoned on every rank", bad_rank, bad_code, tick_); return GVT_HIP_ERR_PEER; }
    return 0;
  };
  // Under several ranks a failure of this rank's LOCAL work (a reservation, a launch, a capacity error) must not leave the peers inside
  // an exchange this rank never joins: the error is remembered, the rank still takes part in the tick's announce with its error word
  // set, and every rank leaves the frame after that exchange -- the failing one with its own error, the others with GVT_HIP_ERR_PEER.
  // (A rank that cannot even run the exchange -- a device fault, a dead process -- is what the deadlines are for.)
  auto failed = [&](int rc_) { if (rc_ && !local_err) { local_err = rc_; local_msg = gvt_hip_last_error(); } return rc_ != 0; };
  const bool multi = world_saved > 1;
  // The exchanges of a frame are issued on the COMPUTE stream itself (the communicator's own stream is bound to it for the frame): a tick is
  // a short dependent sequence -- chain, report, announce, copy, pack, payload, unpack -- and every hop between two streams costs an event
  // record, a wait and the queues' hand-over latency on the device (toy two-rank frame 914 -> 670 us, profiles/r04_tick_floor.txt).  What is
  // given up is the overlap of a SMALL payload with the next chain; a payload of a megabyte or more (knob payload_overlap_kb, sent +
  // received) still moves on the communicator's own stream while the next chain runs.  GVT_HIP_COMM_STREAM=1: everything on that stream, as in round 3.
  struct StreamBind { gvt_hip_comm *K; hipStream_t saved; ~StreamBind() { if (K) K->stream = saved; } } stream_bind{ nullptr, nullptr };
  static const bool env_comm_stream = getenv("GVT_HIP_COMM_STREAM") != nullptr;
  const bool own_comm_stream = env_comm_stream || C.comm_stream != 0;
  const size_t payload_overlap_min = (size_t)C.payload_overlap_kb << 10;
  if (R->comm && !own_comm_stream) { stream_bind.K = R->comm; stream_bind.saved = R->comm->stream; R->comm->stream = st; }
  static const bool spec_trace = getenv("GVT_HIP_SPEC_TRACE") != nullptr;
  auto t_mark = frame_t0;
  bool last_round_small = false; // this tick's host-driven round held at most finish_rays rays (one k_finish launch)
  bool spec_valid = false; // the speculative part enqueued behind the last exchange was valid: the device has already run this tick's round and k_round_report
  int tick = 0;
  for (;; tick++) {
    // (1) local work: one merged chain (asynchronous ticks), or chains until the local queues are dry (BSP rounds / one rank)
    if (multi && R->world > 1 && C.frame_timing) { HIPCHK(hipEventRecord(R->ev_chain0, st)); R->chain_timed = true; }
    if (C.inject_fail_tick == tick && multi) { set_error("injected failure at exchange %d (test knob inject_fail_tick)", tick); failed(GVT_HIP_ERR_CAPACITY); }
    if (R->world == 1 || bsp) {
      if (!local_err && failed(unpack_pending()) && !multi) return local_err;
      for (; !local_err;) {
        bool any = false;
        for (size_t i = 0; i < nI; i++) any = any || (R->owned[i] && R->present[i]);
        if (!any) break;
        if (failed(local_chain(R, nullptr, &S.chains, S.chains == 0 && tick == 0, first_on_device && S.chains == 0)) || failed(round_report(R, false, &S.host_syncs))) { if (!multi) return local_err; break; }
      }
      if (R->world == 1) {
        if (image_split) { // back among the ranks: one exchange, so that rank 0 learns every rank's deposit rectangle
          R->world = world_saved; R->owned = owned_saved;
          if ((rc = round_report(R, true, &S.host_syncs, false, local_err, tick, &S))) return rc;
          if ((rc = peers_ok(tick))) return rc;
        } else if (local_err) return local_err;
        break;
      }
    } else if (spec_valid) {
      // (nothing to launch: k_spec_begin -> k_finish -> k_round_report of this tick sit behind the last exchange; the exchange in front of them left them valid)
    } else if (!local_err) {
      bool have_local = false;
      for (size_t i = 0; i < nI; i++) have_local = have_local || (R->owned[i] && R->present[i]);
      {
        size_t n_loc = 0;
        for (size_t i = 0; i < nI; i++) if (R->owned[i]) n_loc += R->present[i] + incoming[i];
        last_round_small = n_loc <= (size_t)(C.finish_rays > 0 ? C.finish_rays : 0);
        if (last_round_small) R->spec_recent_rays = n_loc;
      }
      if (!have_local || !payload_cross) failed(unpack_pending()); // nothing to overlap a transfer with (or it is here already: inline / compute stream): take what arrived first
      if (!local_err) failed(local_chain(R, &incoming, &S.chains, S.chains == 0 && tick == 0, first_on_device && S.chains == 0));
      if (!local_err) failed(unpack_pending());
    }
    // (3)-(5) sizes + announce exchange (carrying this rank's error word), one bounded synchronisation
    { uint64_t lp = 0; for (size_t i = 0; i < nI; i++) if (R->owned[i]) lp += R->present[i]; R->last_local_pending = lp; }
    {
      const bool was_spec = spec_valid;
      // the next tick's speculative part: asynchronous ticks only, and not where a test makes that tick fail on the host (inject_fail_tick)
      // ... and only behind a tick whose own round was small: a big round (the camera's, a large payload's) usually sends a big payload, the part would be void
      const bool allow_spec = multi && R->world > 1 && !bsp && !local_err && C.inject_fail_tick != tick + 1 && stream_bind.K != nullptr && (was_spec || last_round_small);
      if ((rc = round_report(R, true, &S.host_syncs, false, local_err, tick, &S, was_spec, allow_spec, &spec_valid))) return rc;
      if (was_spec && R->h_report[nI + 13]) { S.chains++; R->spec_recent_rays = R->h_report[nI + 13]; }
      if (spec_trace) { const auto now = std::chrono::steady_clock::now(); if (tick < 8) { R->tick_us[tick] += std::chrono::duration<double, std::micro>(now - t_mark).count(); R->tick_n[tick]++; } t_mark = now; } // (the speculative round traced rays: a launch chain like the host-driven ones)
    }
    S.rounds++;
    if ((rc = peers_ok(tick))) return rc;
    // (6) the vote: every rank reads the same ballots
    uint64_t not_done = 0;
    for (size_t i = 0; i < nI; i++) not_done += R->present[i];
    for (int p = 0; p < R->world; p++)
      if (p != R->rank) not_done += (uint64_t)(unsigned)R->h_ann_in[(size_t)p * row + 2] + (uint64_t)(unsigned)R->h_ann_in[(size_t)p * row + 3];
    if (!not_done) break;
    // (7) payload: pack what is outgoing, post sends / receives; it moves while the next local chain runs
    gvt_hip_comm *K = R->comm;
    bool any_traffic = false;
    std::vector<size_t> bytes_out(R->world, 0), bytes_in(R->world, 0);
    WireBatch PB;
    PB.n_items = 0; PB.total = 0;
    for (int p = 0; p < R->world; p++) {
      if (p == R->rank) continue;
      for (size_t i = 0; i < nI; i++) if (R->owner[i] == p && R->present[i]) bytes_out[p] += 8 + 80 * R->present[i];
      bytes_in[p] = (size_t)(unsigned)R->h_ann_in[(size_t)p * row + 1];
      const size_t inl_in = (size_t)(unsigned)R->h_ann_in[(size_t)p * row + 10];
      if (inl_in) { // this peer's rays are already here: k_publish has appended them to their queues on the device
        if (inl_in != bytes_in[p] || inl_in > R->inl_bytes) { set_error("ray exchange: rank %d announces %zu inline bytes of %zu (inline area %zu)", p, inl_in, bytes_in[p], R->inl_bytes); return GVT_HIP_ERR_DEVICE; }
        bytes_in[p] = 0;
        for (size_t i = 0; i < nI; i++) {
          const unsigned n_in = (unsigned)R->h_ann_in[(size_t)p * row + ANN_HEAD + i];
          if (n_in) { R->present[i] += n_in; R->queues[i]->size = R->present[i]; }
        }
      }
      if (bytes_out[p] && bytes_out[p] <= R->inl_bytes) { // k_round_report applied the same rule: these rays left inside the announce, their queues are cleared
        S.bytes_sent += bytes_out[p];
        for (size_t i = 0; i < nI; i++) {
          if (R->owner[i] != p || !R->present[i]) continue;
          S.rays_sent += R->present[i]; S.rays_inline += R->present[i];
          R->present[i] = 0; R->queues[i]->size = 0;
        }
        bytes_out[p] = 0;
      }
      // (a failure from here on -- after the announce promised these bytes -- cannot be reported before the peers have posted their
      // matching operations: it ends this rank's frame at once and the peers run into the exchange deadline)
      if ((rc = grow(&R->send_buf[p], &R->send_cap[p], bytes_out[p]))) return rc;
      if ((rc = grow(&R->recv_buf[p], &R->recv_cap[p], bytes_in[p]))) return rc;
      S.bytes_sent += bytes_out[p];
      size_t off = 0;
      for (size_t i = 0; i < nI; i++) {
        if (R->owner[i] != p || !R->present[i]) continue;
        gvt_hip_queue *q = R->queues[i];
        const unsigned n = (unsigned)R->present[i];
        if ((rc = wire_add(PB, false, q, (char *)R->send_buf[p] + off, n, (int)i))) return rc;
        off += 8 + 80ull * n;
        S.rays_sent += n;
        R->present[i] = 0; q->size = 0; // the sender's q.second.clear() (:455): the count word is reset by k_pack_all
      }
      any_traffic = any_traffic || bytes_out[p] || bytes_in[p];
    }
    if ((rc = wire_flush(PB, false))) return rc;
    pending_ann.assign(R->h_ann_in, R->h_ann_in + (size_t)R->world * row);
    for (int p = 0; p < R->world; p++)
      if (p != R->rank && !pending_ann[(size_t)p * row + 10]) // (inline payloads are in their queues already)
        for (size_t i = 0; i < nI; i++) incoming[i] += (unsigned)pending_ann[(size_t)p * row + ANN_HEAD + i];
    // room for what arrives in the payload exchange, reserved now (the unpack kernel is launched in the next tick)
    if (any_traffic)
      for (size_t i = 0; i < nI; i++)
        if (incoming[i] && (rc = queue_reserve(R->queues[i], R->present[i] + incoming[i]))) return rc;
    payload_cross = false;
    if (any_traffic) {
      // A payload of a megabyte or more moves on the communicator's OWN stream, while the next tick's local chain runs on the compute
      // stream (ordered by two events); smaller ones -- every late tick of a frame -- stay on the compute stream (StreamBind above).
      size_t bytes_total = 0;
      for (int p = 0; p < R->world; p++) bytes_total += bytes_out[p] + bytes_in[p];
      const bool overlap = stream_bind.K && stream_bind.saved && bytes_total >= payload_overlap_min;
      if (overlap) K->stream = stream_bind.saved;
      struct Rebind { gvt_hip_comm *K; hipStream_t s; ~Rebind() { if (K) K->stream = s; } } rebind{ overlap ? K : nullptr, st };
      if (K->stream != st) { HIPCHK(hipEventRecord(R->ev_pack, st)); HIPCHK(hipStreamWaitEvent(K->stream, R->ev_pack, 0)); }
      if (C.frame_timing) { HIPCHK(hipEventRecord(R->ev_pay0, K->stream)); R->payload_timed = true; }
      comm_group_begin(K); // sizes are known on both sides from the announces: no size handshake on the wire
      for (int p = 0; p < R->world; p++) {
        if (p == R->rank) continue;
        if (bytes_out[p]) comm_send(K, R->send_buf[p], bytes_out[p], p);
        if (bytes_in[p]) comm_recv(K, R->recv_buf[p], bytes_in[p], p);
      }
      if ((rc = comm_group_end(K))) return rc;
      payload_cross = K->stream != st;
      if (payload_cross || C.frame_timing) HIPCHK(hipEventRecord(R->ev_recv, K->stream));
      payload_pending = true;
    }
  }
  // IceTComposite::composite (composite/IceTComposite.cpp:84-101): the float framebuffers summed on rank 0.  A rank's deposits lie
  // where its domains project, so every rank sends rank 0 only the bounding rectangle of the pixels it wrote to -- known to both
  // sides from the last round's announce, no extra handshake -- and rank 0 adds the rectangles: about one frame of traffic in
  // total, spread over the links of all peers, instead of a reduce of whole W x H x 16-byte frames.  (GVT_HIP_FRAME_FULL_REDUCE:
  // ncclReduce of the whole frame.)  Same sums in rank order; exact wherever one rank writes a pixel.
  if (R->world > 1 && !(flags & GVT_HIP_FRAME_NO_COMPOSITE)) {
    gvt_hip_comm *K = R->comm;
    const int W = R->fb->w;
    if (C.frame_timing) HIPCHK(hipEventRecord(R->ev_comp0, st));
    if (flags & GVT_HIP_FRAME_FULL_REDUCE) {
      if (K->stream != st) { HIPCHK(hipEventRecord(R->ev_compute, st)); HIPCHK(hipStreamWaitEvent(K->stream, R->ev_compute, 0)); }
      if ((rc = comm_reduce_sum(K, R->fb->d_rgba, (size_t)R->fb->w * R->fb->h * 4, 0))) return rc;
    } else if (R->rank != 0) {
      const int *bb = (const int *)(R->h_report + nI + 6);
      const int w = bb[2] - bb[0], h = bb[3] - bb[1];
      if (w > 0 && h > 0) {
        const size_t bytes = (size_t)w * h * 16;
        if ((rc = grow(&R->send_buf[0], &R->send_cap[0], bytes))) return rc;
        k_rect<false><<<(unsigned)(((size_t)w * h + 255) / 256), 256, 0, st>>>((float4 *)R->fb->d_rgba, W, bb[0], bb[1], w, h, (float4 *)R->send_buf[0]);
        HIPCHK(hipGetLastError());
        if (K->stream != st) { HIPCHK(hipEventRecord(R->ev_compute, st)); HIPCHK(hipStreamWaitEvent(K->stream, R->ev_compute, 0)); }
        comm_group_begin(K);
        comm_send(K, R->send_buf[0], bytes, 0);
        if ((rc = comm_group_end(K))) return rc;
      }
    } else {
      if (K->stream != st) { HIPCHK(hipEventRecord(R->ev_compute, st)); HIPCHK(hipStreamWaitEvent(K->stream, R->ev_compute, 0)); }
      comm_group_begin(K);
      for (int p = 1; p < R->world; p++) {
        const int *bb = R->h_ann_in + (size_t)p * row + 4;
        const int w = bb[2] - bb[0], h = bb[3] - bb[1];
        if (w <= 0 || h <= 0) continue;
        const size_t bytes = (size_t)w * h * 16;
        if ((rc = grow(&R->recv_buf[p], &R->recv_cap[p], bytes))) return rc;
        comm_recv(K, R->recv_buf[p], bytes, p);
      }
      if ((rc = comm_group_end(K))) return rc;
      for (int p = 1; p < R->world; p++) {
        const int *bb = R->h_ann_in + (size_t)p * row + 4;
        const int w = bb[2] - bb[0], h = bb[3] - bb[1];
        if (w <= 0 || h <= 0) continue;
        k_rect<true><<<(unsigned)(((size_t)w * h + 255) / 256), 256, 0, K->stream>>>((float4 *)R->fb->d_rgba, W, bb[0], bb[1], w, h, (float4 *)R->recv_buf[p]);
      }
      HIPCHK(hipGetLastError());
    }
    HIPCHK(hipEventRecord(R->ev_comm, K->stream));
    if (K->stream != st) HIPCHK(hipStreamWaitEvent(st, R->ev_comm, 0));
    if ((rc = bounded_event_wait(K, R->ev_comm, "the framebuffer composite"))) return rc;
    S.host_syncs++;
    if (C.frame_timing) { float ms = 0.f; if (hipEventElapsedTime(&ms, R->ev_comp0, R->ev_comm) == hipSuccess) S.ms_composite += ms; }
  }
  if (R->comm) { S.ms_host_wait = R->comm->ms_host_wait; S.exchanges = R->comm->groups; }
  if (spec_trace && multi) R->comp_us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_mark).count();
  const unsigned *tail = R->h_report + nI;
  S.rays_closest = (uint64_t)tail[0] | ((uint64_t)tail[1] << 32);
  S.rays_any = (uint64_t)tail[2] | ((uint64_t)tail[3] << 32);
  S.packets_bailed = tail[10];
  if (C.long_auto && C.long_steps > 0 && S.rays_closest >= 65536) {
    const double frac = (double)tail[11] / (double)S.rays_closest;
    int cur = std::max(R->long_cur, C.long_steps);
    if (frac > 0.01) cur = std::min(1024, cur + cur / 2);
    else if (frac > 0.003) cur = std::min(1024, cur + cur / 4);
    else if (frac < 0.0005) cur = std::max(C.long_steps, cur - cur / 5);
    R->long_cur = cur;
  }
  C.stats.rays_closest += S.rays_closest;
  C.stats.rays_any += S.rays_any;
  if (fin_auto && R->frame_no >= 2 && R->fin_choice < 0) { // a probing frame: its time counts
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - frame_t0).count();
    R->fin_best[fin_variant] = std::min(R->fin_best[fin_variant], ms);
    R->fin_n[fin_variant]++;
    R->fin_probe++;
    // once every route has been timed once: a route 30 % behind the best needs no second look (a short run does not spend its frames on them)
    bool once = true;
    double lead = 1e30;
    for (int k = 0; k < n_allowed; k++) { once = once && R->fin_n[route_list[k]] >= 1; lead = std::min(lead, R->fin_best[route_list[k]]); }
    if (once) for (int k = 0; k < n_allowed; k++) if (R->fin_best[route_list[k]] > 1.3 * lead) R->fin_n[route_list[k]] = std::max(R->fin_n[route_list[k]], 3);
    bool all = true;
    int best = route_list[0];
    for (int k = 0; k < n_allowed; k++) {
      all = all && R->fin_n[route_list[k]] >= 3;
      if (R->fin_best[route_list[k]] < R->fin_best[best] || (R->fin_best[route_list[k]] == R->fin_best[best] && route_list[k] > best)) best = route_list[k];
    }
    if (all) {
      R->fin_choice = best;
      if (getenv("GVT_HIP_ROUTE_TRACE")) {
        fprintf(stderr, "[tracer] frame %llu: route settled -- k_finish %d, hops %s; fastest frame per route timed, ms:", (unsigned long long)R->frame_no, best & 1,
                (best >> 1) == 0 ? "never" : (best >> 1) == 1 ? "early" : "always");
        for (int k = 0; k < n_allowed; k++) fprintf(stderr, " [k_finish %d hops %d] %.3f", route_list[k] & 1, route_list[k] >> 1, R->fin_best[route_list[k]]);
        fprintf(stderr, "\n");
      }
    }
  }
  R->frame_no++;
  if (out) *out = S;
  return 0;
}
