// TEST DOUBLE, not part of the product: a stand-in for librccl.so that lets SEVERAL PROCESSES ON ONE GPU run the library's RCCL leg
// (gravit_amd/csrc/domain.hip: ncclCommInitRank, grouped ncclSend / ncclRecv, ncclReduce) -- RCCL itself refuses two ranks on one device,
// and a GPU box of this pool has one.  Loaded through GVT_HIP_RCCL_LIB by tests/test_gpu_multiproc.py only.
//
// Same entry points and types as <rccl/rccl.h> (compiled against it, so a signature drift fails the build).  Semantics kept from the real
// library: a communicator of `world` ranks built from a 128-byte id shared out of band; point-to-point messages matched per ordered pair in
// issue order; every receive must find a send of exactly its size (a mismatch is an error here, a hang or corruption there); sends and
// receives of one group progress together (sends never wait for the peer); the reduce sums floats onto the root.  NOT kept: asynchrony --
// every call here synchronises the stream and moves the bytes through files under $TMPDIR before it returns (a legal, slow
// implementation: nothing in the caller may depend on an exchange being still in flight), so this says nothing about speed or overlap.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <string>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>
#include <vector>

namespace {
struct Op { int send; void *ptr; size_t bytes; int peer; hipStream_t stream; };
struct FakeComm {
  std::string dir;
  int rank = 0, world = 1;
  std::vector<unsigned long long> seq_out, seq_in; // messages sent to / received from every peer so far
  unsigned long long seq_red = 0;
  bool failed = false;
};
thread_local int g_depth = 0;
thread_local std::vector<std::pair<FakeComm *, Op>> g_ops;

int timeout_s() { const char *e = getenv("FAKE_RCCL_TIMEOUT_S"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 120; }
bool exists(const std::string &p) { struct stat st; return stat(p.c_str(), &st) == 0; }
bool wait_for(const std::string &p) {
  const auto t0 = std::chrono::steady_clock::now();
  for (unsigned spins = 0; !exists(p); spins++) {
    if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(timeout_s())) return false;
    if (spins > 2000) std::this_thread::sleep_for(std::chrono::microseconds(50));
  }
  return true;
}
bool write_file(const std::string &path, const void *data, size_t bytes) { // complete before it is visible: written under another name, then renamed
  const std::string tmp = path + ".part";
  FILE *f = std::fopen(tmp.c_str(), "wb");
  if (!f) return false;
  const bool ok = bytes == 0 || std::fwrite(data, 1, bytes, f) == bytes;
  std::fclose(f);
  return ok && std::rename(tmp.c_str(), path.c_str()) == 0;
}
long read_file(const std::string &path, std::vector<unsigned char> &out) {
  FILE *f = std::fopen(path.c_str(), "rb");
  if (!f) return -1;
  std::fseek(f, 0, SEEK_END);
  const long n = std::ftell(f);
  std::fseek(f, 0, SEEK_SET);
  out.resize((size_t)n);
  const bool ok = n == 0 || std::fread(out.data(), 1, (size_t)n, f) == (size_t)n;
  std::fclose(f);
  return ok ? n : -1;
}
std::string msg_path(const FakeComm *K, int src, int dst, unsigned long long seq) {
  char b[96];
  std::snprintf(b, sizeof b, "/p2p_%d_%d_%llu", src, dst, seq);
  return K->dir + b;
}
ncclResult_t fail(FakeComm *K, const char *fmt, long a = 0, long b = 0, long c = 0) {
  std::fprintf(stderr, "fake_rccl: rank %d of %d: ", K ? K->rank : -1, K ? K->world : -1);
  std::fprintf(stderr, fmt, a, b, c);
  std::fprintf(stderr, "\n");
  if (K) { K->failed = true; write_file(K->dir + "/FAILED", "x", 1); }
  return ncclInternalError;
}

ncclResult_t run(std::vector<std::pair<FakeComm *, Op>> &ops) {
  // every send of the group first (a send never waits for its peer), then every receive
  std::vector<unsigned char> host;
  for (auto &co : ops) {
    FakeComm *K = co.first; Op &o = co.second;
    if (!o.send) continue;
    if (hipStreamSynchronize(o.stream) != hipSuccess) return fail(K, "stream synchronisation before a send failed");
    host.resize(o.bytes);
    if (o.bytes && hipMemcpy(host.data(), o.ptr, o.bytes, hipMemcpyDeviceToHost) != hipSuccess) return fail(K, "device-to-host copy of a %ld-byte send failed", (long)o.bytes);
    if (!write_file(msg_path(K, K->rank, o.peer, K->seq_out[o.peer]++), host.data(), o.bytes)) return fail(K, "cannot write a message for rank %ld", o.peer);
  }
  for (auto &co : ops) {
    FakeComm *K = co.first; Op &o = co.second;
    if (o.send) continue;
    const std::string p = msg_path(K, o.peer, K->rank, K->seq_in[o.peer]++);
    if (!wait_for(p)) return fail(K, "no matching send from rank %ld for a receive of %ld bytes (message %ld of that pair)", o.peer, (long)o.bytes, (long)K->seq_in[o.peer] - 1);
    const long n = read_file(p, host);
    if (n < 0 || (size_t)n != o.bytes) return fail(K, "a receive of %ld bytes from rank %ld met a send of %ld bytes", (long)o.bytes, o.peer, n);
    std::remove(p.c_str());
    if (hipStreamSynchronize(o.stream) != hipSuccess) return fail(K, "stream synchronisation before a receive failed");
    if (o.bytes && hipMemcpy(o.ptr, host.data(), o.bytes, hipMemcpyHostToDevice) != hipSuccess) return fail(K, "host-to-device copy of a %ld-byte receive failed", (long)o.bytes);
  }
  ops.clear();
  return ncclSuccess;
}
size_t type_bytes(ncclDataType_t t) { return (t == ncclUint8 || t == ncclInt8) ? 1 : (t == ncclFloat16 || t == ncclBfloat16) ? 2 : (t == ncclFloat64 || t == ncclInt64 || t == ncclUint64) ? 8 : 4; }
} // namespace

extern "C" {
ncclResult_t ncclGetUniqueId(ncclUniqueId *id) {
  std::memset(id, 0, sizeof *id);
  unsigned char r[8] = { 0 };
  FILE *f = std::fopen("/dev/urandom", "rb");
  if (f) { if (std::fread(r, 1, 8, f) != 8) r[0] = 1; std::fclose(f); }
  std::snprintf(id->internal, sizeof id->internal, "fakerccl_%d_%02x%02x%02x%02x%02x%02x%02x%02x", (int)getpid(), r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7]);
  return ncclSuccess;
}
ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank) {
  if (!comm || nranks < 1 || rank < 0 || rank >= nranks || std::strncmp(id.internal, "fakerccl_", 9) != 0) return ncclInvalidArgument;
  FakeComm *K = new FakeComm();
  const char *tmp = getenv("TMPDIR");
  K->dir = std::string(tmp && *tmp ? tmp : "/tmp") + "/" + std::string(id.internal, strnlen(id.internal, sizeof id.internal));
  K->rank = rank; K->world = nranks;
  K->seq_out.assign(nranks, 0); K->seq_in.assign(nranks, 0);
  mkdir(K->dir.c_str(), 0700);
  char b[64];
  std::snprintf(b, sizeof b, "/rank_%d", rank);
  if (!write_file(K->dir + b, "x", 1)) { delete K; return ncclSystemError; }
  for (int r = 0; r < nranks; r++) { // every rank of the communicator has arrived (the real call is collective too)
    std::snprintf(b, sizeof b, "/rank_%d", r);
    if (!wait_for(K->dir + b)) { fail(K, "rank %ld never joined the communicator", r); delete K; return ncclInternalError; }
  }
  *comm = (ncclComm_t)K;
  return ncclSuccess;
}
ncclResult_t ncclCommCount(const ncclComm_t comm, int *count) { if (!comm || !count) return ncclInvalidArgument; *count = ((FakeComm *)comm)->world; return ncclSuccess; }
ncclResult_t ncclCommDestroy(ncclComm_t comm) { delete (FakeComm *)comm; return ncclSuccess; } // (the directory is the test's to remove: a peer may still be reading)
ncclResult_t ncclCommAbort(ncclComm_t comm) { if (comm) write_file(((FakeComm *)comm)->dir + "/FAILED", "x", 1); delete (FakeComm *)comm; return ncclSuccess; }
const char *ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : r == ncclInvalidArgument ? "invalid argument" : r == ncclSystemError ? "system error" : "internal error (fake_rccl: see stderr)"; }
ncclResult_t ncclGroupStart() { g_depth++; return ncclSuccess; }
ncclResult_t ncclGroupEnd() {
  if (g_depth <= 0) return ncclInvalidUsage;
  if (--g_depth) return ncclSuccess;
  return run(g_ops);
}
ncclResult_t ncclSend(const void *buf, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t stream) {
  FakeComm *K = (FakeComm *)comm;
  if (!K || peer < 0 || peer >= K->world) return ncclInvalidArgument;
  g_ops.push_back({ K, Op{ 1, (void *)buf, count * type_bytes(t), peer, stream } });
  return g_depth ? ncclSuccess : run(g_ops);
}
ncclResult_t ncclRecv(void *buf, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t stream) {
  FakeComm *K = (FakeComm *)comm;
  if (!K || peer < 0 || peer >= K->world) return ncclInvalidArgument;
  g_ops.push_back({ K, Op{ 0, buf, count * type_bytes(t), peer, stream } });
  return g_depth ? ncclSuccess : run(g_ops);
}
// float sums onto the root, peers added in rank order (the real library's order is its own; the caller may not depend on it)
ncclResult_t ncclReduce(const void *sendbuf, void *recvbuf, size_t count, ncclDataType_t t, ncclRedOp_t op, int root, ncclComm_t comm, hipStream_t stream) {
  FakeComm *K = (FakeComm *)comm;
  if (!K || t != ncclFloat || op != ncclSum || root < 0 || root >= K->world) return ncclInvalidArgument;
  const size_t bytes = count * 4;
  const unsigned long long seq = K->seq_red++;
  if (hipStreamSynchronize(stream) != hipSuccess) return fail(K, "stream synchronisation before a reduce failed");
  std::vector<float> mine(count);
  if (bytes && hipMemcpy(mine.data(), sendbuf, bytes, hipMemcpyDeviceToHost) != hipSuccess) return fail(K, "device-to-host copy of a reduce failed");
  char b[96];
  if (K->rank != root) {
    std::snprintf(b, sizeof b, "/red_%llu_%d", seq, K->rank);
    return write_file(K->dir + b, mine.data(), bytes) ? ncclSuccess : fail(K, "cannot write a reduce contribution");
  }
  std::vector<unsigned char> in;
  for (int p = 0; p < K->world; p++) {
    if (p == root) continue;
    std::snprintf(b, sizeof b, "/red_%llu_%d", seq, p);
    if (!wait_for(K->dir + b)) return fail(K, "rank %ld never contributed to reduce %ld", p, (long)seq);
    const long n = read_file(K->dir + b, in);
    if (n < 0 || (size_t)n != bytes) return fail(K, "reduce %ld: rank %ld contributed %ld bytes", (long)seq, p, n);
    std::remove((K->dir + b).c_str());
    const float *f = (const float *)in.data();
    for (size_t i = 0; i < count; i++) mine[i] += f[i];
  }
  if (bytes && hipMemcpy(recvbuf, mine.data(), bytes, hipMemcpyHostToDevice) != hipSuccess) return fail(K, "host-to-device copy of a reduce failed");
  return ncclSuccess;
}
}
