// fake_rccl.cc — a TEST DOUBLE for the eight RCCL entry points libfcp_hip.so binds at run time (fcp_shard.hip: FCP_RCCL_PATH),
// so that the native sharded step — fcp_shard_step_run's grouped ncclSend / ncclRecv schedule, batch-slice order, ring reuse,
// finalize / concat behind it — can run with SEVERAL RANKS AS PROCESSES ON ONE GPU.  Real RCCL refuses two ranks on one device
// and the boxes of this pool have one GPU; `tests/test_0_gpu_shard_ranks.py::test_native_sharded_step_over_rccl_two_gpus`
// remains the test over real RCCL on real hardware.  Test infrastructure only: it is never loaded unless FCP_RCCL_PATH names it.
//
// Transport: one POSIX shared-memory file per communicator (its name travels inside the 128-byte unique id), holding a
// world x world matrix of single-message mailboxes.  ncclSend / ncclRecv are queued between ncclGroupStart / ncclGroupEnd, as
// RCCL queues them; ncclGroupEnd (or a lone call) then
//   1. synchronises the stream (everything the caller enqueued before the collective has produced its data),
//   2. moves every queued message through its pair's mailbox in chunks of at most one slot (device -> host on the sending
//      side, host -> device on the receiving side), all operations progressing together: no operation waits for another,
//      so the exchange cannot deadlock whatever the message sizes.
// It BLOCKS the host where RCCL would only enqueue; ordering on the stream is what a real collective guarantees, timing is not
// what this double is for.  Message sizes must match between the two sides (checked): a schedule that pairs the wrong slices
// fails loudly.  Build: hipcc -shared -fPIC tests/native/fake_rccl.cc -o tests/native/libfake_rccl.so
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

namespace {

constexpr uint64_t kMagic = 0x46435046414b4552ull; // "FCPFAKER"
constexpr double kTimeoutSeconds = 120.0;          // a peer that never arrives: fail instead of hanging the box

struct Mailbox {
  std::atomic<uint64_t> written; // messages published by the sender
  std::atomic<uint64_t> read;    // messages consumed by the receiver
  uint64_t bytes;                // size of the message in flight
  uint64_t pad_;
};

struct Header {
  std::atomic<uint64_t> magic;
  std::atomic<uint32_t> arrived;  // ranks that have mapped the file (ncclCommInitRank is a collective)
  std::atomic<uint32_t> departed; // ranks that have destroyed their communicator
  uint32_t world;
  uint32_t pad_;
  uint64_t slot_bytes;
};

size_t slot_capacity() {
  const char *e = std::getenv("FCP_FAKE_RCCL_SLOT_BYTES");
  return e ? (size_t)std::atoll(e) : (size_t)4 << 20;
}

struct Op {
  bool send;
  const void *src;
  void *dst;
  size_t bytes;
  int peer;
  hipStream_t stream;
};

} // namespace

struct ncclComm {
  int rank = 0, world = 1;
  char name[64] = {0};
  char *base = nullptr;
  size_t map_bytes = 0;
  Header *hdr = nullptr;
  Mailbox *box(int src, int dst) { return reinterpret_cast<Mailbox *>(base + 4096) + (size_t)src * world + dst; }
  char *data(int src, int dst) { return base + 4096 + 4096 * ((sizeof(Mailbox) * (size_t)world * world + 4095) / 4096) + ((size_t)src * world + dst) * hdr->slot_bytes; }
};
typedef struct ncclComm *ncclComm_t;
typedef struct {
  char internal[128];
} ncclUniqueId;

namespace {

thread_local int g_depth = 0;
thread_local std::vector<std::pair<ncclComm_t, Op>> g_queue;
thread_local char g_error[256] = "no error";

enum { kSuccess = 0, kUnhandledCudaError = 1, kSystemError = 2, kInternalError = 3, kInvalidArgument = 4 };

int fail(int code, const char *what) {
  std::snprintf(g_error, sizeof(g_error), "fake_rccl: %s", what);
  std::fprintf(stderr, "%s\n", g_error);
  return code;
}

template <typename F> bool wait_until(F cond) {
  const auto t0 = std::chrono::steady_clock::now();
  int spins = 0;
  while (!cond()) {
    if (++spins > 2000) std::this_thread::sleep_for(std::chrono::microseconds(50));
    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > kTimeoutSeconds) return false;
  }
  return true;
}

size_t type_size(int datatype) {
  switch (datatype) { // ncclDataType_t of rccl.h
  case 0: case 1: return 1;               // int8 / uint8
  case 2: case 3: case 7: return 4;       // int32 / uint32 / float32
  case 4: case 5: case 8: return 8;       // int64 / uint64 / float64
  case 6: case 9: return 2;               // float16 / bfloat16
  default: return 0;
  }
}

int flush() {
  // A progress engine over every queued operation: a message travels in chunks of at most one mailbox slot, the first
  // chunk announcing the message's total size (checked against what the receiver expects).  No operation ever blocks the
  // others — each pass moves whatever can move — so any schedule RCCL would complete completes here too, whatever the
  // message sizes (SHARD's 30.7 MB slices through 4 MiB slots), and a schedule that pairs the wrong slices fails loudly.
  std::vector<std::pair<ncclComm_t, Op>> ops;
  ops.swap(g_queue);
  if (ops.empty()) return kSuccess;
  for (auto &o : ops)
    if (hipStreamSynchronize(o.second.stream) != hipSuccess) return fail(kUnhandledCudaError, "hipStreamSynchronize before the exchange");
  std::vector<size_t> done(ops.size(), 0);
  std::vector<char> finished(ops.size(), 0);
  // several operations of one group towards the same peer share that pair's mailbox: they take turns in queue order
  auto earlier_unfinished = [&](size_t i) {
    for (size_t j = 0; j < i; ++j)
      if (!finished[j] && ops[j].first == ops[i].first && ops[j].second.send == ops[i].second.send && ops[j].second.peer == ops[i].second.peer) return true;
    return false;
  };
  size_t left = ops.size();
  const auto t0 = std::chrono::steady_clock::now();
  int idle = 0;
  while (left) {
    bool moved = false;
    for (size_t i = 0; i < ops.size(); ++i) {
      if (finished[i] || earlier_unfinished(i)) continue;
      ncclComm_t c = ops[i].first;
      Op &o = ops[i].second;
      const size_t slot = c->hdr->slot_bytes;
      if (o.send) {
        Mailbox *b = c->box(c->rank, o.peer);
        if (b->written.load(std::memory_order_acquire) != b->read.load(std::memory_order_acquire)) continue; // previous chunk not consumed yet
        const size_t n = std::min(slot, o.bytes - done[i]);
        if (n && hipMemcpy(c->data(c->rank, o.peer), static_cast<const char *>(o.src) + done[i], n, hipMemcpyDeviceToHost) != hipSuccess)
          return fail(kUnhandledCudaError, "device -> host copy of a send");
        b->bytes = o.bytes; // every chunk carries the message's total size
        b->written.fetch_add(1, std::memory_order_release);
        done[i] += n;
        moved = true;
        if (done[i] == o.bytes) finished[i] = 1, --left;
      } else {
        Mailbox *b = c->box(o.peer, c->rank);
        if (b->written.load(std::memory_order_acquire) == b->read.load(std::memory_order_acquire)) continue; // nothing there yet
        if (b->bytes != o.bytes) {
          char msg[160];
          std::snprintf(msg, sizeof(msg), "rank %d expects %zu bytes from rank %d, which sent %llu: the two sides disagree about the slices",
                        c->rank, o.bytes, o.peer, (unsigned long long)b->bytes);
          return fail(kInvalidArgument, msg);
        }
        const size_t n = std::min(slot, o.bytes - done[i]);
        if (n && hipMemcpy(static_cast<char *>(o.dst) + done[i], c->data(o.peer, c->rank), n, hipMemcpyHostToDevice) != hipSuccess)
          return fail(kUnhandledCudaError, "host -> device copy of a receive");
        b->read.fetch_add(1, std::memory_order_release);
        done[i] += n;
        moved = true;
        if (done[i] == o.bytes) finished[i] = 1, --left;
      }
    }
    if (moved) {
      idle = 0;
      continue;
    }
    if (++idle > 2000) std::this_thread::sleep_for(std::chrono::microseconds(50));
    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > kTimeoutSeconds)
      return fail(kSystemError, "timeout: a peer never sent, or never consumed, a message");
  }
  return kSuccess;
}

} // namespace

extern "C" {

int ncclGetUniqueId(ncclUniqueId *id) {
  if (!id) return fail(kInvalidArgument, "null id");
  std::memset(id->internal, 0, sizeof(id->internal));
  const auto now = std::chrono::steady_clock::now().time_since_epoch().count();
  std::snprintf(id->internal, 64, "/fcp_fake_rccl_%d_%llx", (int)getpid(), (unsigned long long)now);
  return kSuccess;
}

int ncclCommInitRank(ncclComm_t *out, int world, ncclUniqueId id, int rank) {
  if (!out || world < 1 || rank < 0 || rank >= world || id.internal[0] != '/') return fail(kInvalidArgument, "bad communicator arguments");
  ncclComm *c = new ncclComm();
  c->rank = rank;
  c->world = world;
  std::memcpy(c->name, id.internal, 63);
  const size_t slot = slot_capacity();
  const size_t box_bytes = 4096 * ((sizeof(Mailbox) * (size_t)world * world + 4095) / 4096);
  c->map_bytes = 4096 + box_bytes + (size_t)world * world * slot;
  const int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
  if (fd < 0 || ftruncate(fd, (off_t)c->map_bytes) != 0) { // every rank sets the same size; a fresh file reads as zeros
    if (fd >= 0) close(fd);
    delete c;
    return fail(kSystemError, "shm_open / ftruncate");
  }
  void *p = mmap(nullptr, c->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) {
    delete c;
    return fail(kSystemError, "mmap");
  }
  c->base = static_cast<char *>(p);
  c->hdr = reinterpret_cast<Header *>(p);
  uint64_t zero = 0;
  if (c->hdr->magic.compare_exchange_strong(zero, kMagic)) {
    c->hdr->world = (uint32_t)world;
    c->hdr->slot_bytes = slot;
  }
  c->hdr->arrived.fetch_add(1);
  if (!wait_until([&] { return c->hdr->arrived.load() >= (uint32_t)world && c->hdr->slot_bytes == slot && c->hdr->world == (uint32_t)world; })) {
    munmap(p, c->map_bytes);
    delete c;
    return fail(kSystemError, "timeout: not every rank reached ncclCommInitRank (or the ranks disagree about the world)");
  }
  *out = c;
  return kSuccess;
}

int ncclCommDestroy(ncclComm_t c) {
  if (!c) return kSuccess;
  if (c->hdr->departed.fetch_add(1) + 1 == (uint32_t)c->world) shm_unlink(c->name); // the last one out removes the name
  munmap(c->base, c->map_bytes);
  delete c;
  return kSuccess;
}

int ncclGroupStart(void) {
  ++g_depth;
  return kSuccess;
}

int ncclGroupEnd(void) {
  if (g_depth <= 0) return fail(kInvalidArgument, "ncclGroupEnd without ncclGroupStart");
  if (--g_depth > 0) return kSuccess;
  return flush();
}

int ncclSend(const void *buf, size_t count, int datatype, int peer, ncclComm_t c, hipStream_t stream) {
  const size_t ts = type_size(datatype);
  if (!c || !ts || peer < 0 || peer >= c->world || (count && !buf)) return fail(kInvalidArgument, "bad ncclSend arguments");
  g_queue.push_back({c, Op{true, buf, nullptr, count * ts, peer, stream}});
  return g_depth > 0 ? kSuccess : flush();
}

int ncclRecv(void *buf, size_t count, int datatype, int peer, ncclComm_t c, hipStream_t stream) {
  const size_t ts = type_size(datatype);
  if (!c || !ts || peer < 0 || peer >= c->world || (count && !buf)) return fail(kInvalidArgument, "bad ncclRecv arguments");
  g_queue.push_back({c, Op{false, nullptr, buf, count * ts, peer, stream}});
  return g_depth > 0 ? kSuccess : flush();
}

const char *ncclGetErrorString(int) { return g_error; }

} // extern "C"
