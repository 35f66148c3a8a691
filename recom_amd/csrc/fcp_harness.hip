// fcp_harness.hip — native measurement harness (libfcp_harness.so).
//
// Mirrors the timing protocol of the reference's C++ harness
// `benchmark_multi_thread` (examples/cc/recom_examples.patch:98-263: shared
// model, `serve_workers` host threads, warm-up run, N timed `Session::Run`s on
// shared inputs, average latency per thread + throughput), with the TF session
// replaced by direct calls through the C ABI of libfcp_hip.so.  Host-side
// Python overhead is therefore not part of any measured number.
//
// The arena allocator plays the role of TF's allocator: a ring of pre-allocated
// arenas, sized so that consecutive requests never write the same bytes while
// they could still sit in the 256 MiB Infinity Cache.
// Under FCP_ORDER_INPUTS_READY (kernels launched without the queue's barrier bit) nothing FORMALLY orders request k + ring
// against request k, which wrote the same arena.  The harness relies on what the queue does in practice: it takes its
// packets in order and — measured, scripts/probes/any_order_probe.hip — does not start a kernel before the one in front of
// it has ended, barrier bit or not; request k + 6 follows five whole requests behind request k.  The harness reads no result of such a run (verify_resident runs in stream order).  A
// caller that consumes results must not copy this: it gives every in-flight request an arena nothing else writes
// (the promise FCP_ORDER_INPUTS_READY states in fcp_hip.h), e.g. by waiting for request k's completion event before it
// hands arena k out again.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "../../include/fcp_hip.h"

namespace {

struct Ring {
  std::vector<void *> bufs;
  size_t bytes = 0;
  size_t next = 0;
};

void *ring_alloc(void *ctx, size_t bytes) {
  Ring *r = static_cast<Ring *>(ctx);
  if (bytes > r->bytes || r->bufs.empty()) return nullptr;
  void *p = r->bufs[r->next];
  r->next = (r->next + 1) % r->bufs.size();
  return p;
}

#define H_TRY(expr)                                                                    \
  do {                                                                                 \
    hipError_t e_ = (expr);                                                            \
    if (e_ != hipSuccess) {                                                            \
      std::fprintf(stderr, "fcp_harness: %s: %s\n", #expr, hipGetErrorString(e_));     \
      return FCP_ERR_HIP;                                                              \
    }                                                                                  \
  } while (0)

// the calling thread on the harness' device for the length of a call (kernels launched here, events, hipMalloc)
struct OnDevice {
  int prev = -1;
  explicit OnDevice(int dev) {
    int cur = -1;
    if (hipGetDevice(&cur) == hipSuccess && cur != dev && hipSetDevice(dev) == hipSuccess) prev = cur;
  }
  ~OnDevice() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
};

} // namespace

struct fcp_harness {
  fcp_plan_t *plan = nullptr;
  std::vector<fcp_process_args_t> variants;
  std::vector<Ring> rings;
  std::vector<hipStream_t> streams;
  std::vector<int> status;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  long issued = 0; // requests issued so far per worker (variant rotation)
  int device = 0;  // the device the streams and arenas were created on: a new host thread starts on device 0, so every
                   // worker thread selects it first (rank r of a multi-GPU node serves device r)

  void issue(int t, long begin, int count) {
    const int nv = (int)variants.size();
    for (long k = begin; k < begin + count; ++k) {
      fcp_process_args_t a = variants[(size_t)((k + t) % nv)];
      a.stream = streams[t];
      a.malloc_buff = ring_alloc;
      a.malloc_buff_ctx = &rings[t];
      a.malloc_temp = nullptr;
      a.malloc_temp_ctx = nullptr;
      const int rc = fcp_process_feature_columns(plan, &a, nullptr);
      if (rc) {
        status[t] = rc;
        return;
      }
    }
  }
};

extern "C" {

// variants[n_variants]: request descriptors that are cycled through (their
// stream / allocator fields are overwritten; the arrays they point to must stay
// alive).  arena_ring: pre-allocated output arenas per worker.  n_threads:
// serve_workers — host threads sharing the plan, one stream each.
int fcp_harness_create(fcp_plan_t *plan, const fcp_process_args_t *variants, int n_variants, int arena_ring,
                       int n_threads, fcp_harness **out) {
  if (!plan || !variants || !out || n_variants < 1 || n_threads < 1 || arena_ring < 1)
    return FCP_ERR_INVALID_ARGUMENT;
  int64_t arena_bytes = 0;
  for (int v = 0; v < n_variants; ++v) {
    int64_t b = 0;
    int rc = fcp_plan_arena_bytes(plan, variants[v].concated_shapes, variants[v].symbols, &b);
    if (rc) return rc;
    arena_bytes = std::max(arena_bytes, b);
  }
  fcp_harness *h = new fcp_harness();
  h->plan = plan;
  H_TRY(hipGetDevice(&h->device));
  h->variants.assign(variants, variants + n_variants);
  h->rings.resize(n_threads);
  h->streams.resize(n_threads);
  h->status.assign(n_threads, FCP_OK);
  for (int t = 0; t < n_threads; ++t) {
    H_TRY(hipStreamCreateWithFlags(&h->streams[t], hipStreamNonBlocking));
    h->rings[t].bytes = (size_t)arena_bytes;
    for (int i = 0; i < arena_ring; ++i) {
      void *p = nullptr;
      H_TRY(hipMalloc(&p, (size_t)std::max<int64_t>(arena_bytes, 256)));
      h->rings[t].bufs.push_back(p);
    }
  }
  H_TRY(hipEventCreate(&h->e0));
  H_TRY(hipEventCreate(&h->e1));
  *out = h;
  return FCP_OK;
}

// Issues exactly `steps` requests per worker and waits for them.
//   wall_ms   host wall time (issue + final stream synchronisation);
//   dev_ms    HIP-event time over worker 0's requests, recorded on the stream
//             the kernels are launched on;
//   iter_ms   optional float[steps]: per-request device time of worker 0 from
//             one HIP event pair per request (perturbs throughput slightly — use
//             a separate call for latency percentiles).
int fcp_harness_run(fcp_harness *h, int steps, double *wall_ms, float *dev_ms, float *iter_ms) {
  if (!h || steps < 1) return FCP_ERR_INVALID_ARGUMENT;
  OnDevice on(h->device);
  const int n_threads = (int)h->streams.size();
  const long begin = h->issued;
  std::vector<hipEvent_t> ev;
  if (iter_ms) {
    ev.resize(2 * (size_t)steps);
    for (auto &e : ev) H_TRY(hipEventCreate(&e));
  }
  const auto t0 = std::chrono::steady_clock::now();
  H_TRY(hipEventRecord(h->e0, h->streams[0]));
  std::vector<std::thread> workers;
  for (int t = 1; t < n_threads; ++t) workers.emplace_back([h, t, begin, steps] {
      (void)hipSetDevice(h->device);
      h->issue(t, begin, steps);
    });
  hipError_t worker0 = hipSuccess; // no early return while the other workers run: a joinable std::thread must be joined
  if (iter_ms) {
    for (int k = 0; k < steps && worker0 == hipSuccess; ++k) {
      worker0 = hipEventRecord(ev[2 * k], h->streams[0]);
      h->issue(0, begin + k, 1);
      if (worker0 == hipSuccess) worker0 = hipEventRecord(ev[2 * k + 1], h->streams[0]);
    }
  } else {
    h->issue(0, begin, steps);
  }
  for (auto &w : workers) w.join();
  H_TRY(worker0);
  H_TRY(hipEventRecord(h->e1, h->streams[0]));
  // (round 6 polled the end of the region with hipStreamQuery for a few runs: it bought nothing against hipStreamSynchronize's
  // wake-up — the bracket is the first launch out of an idle queue and the caller's device synchronisation — and is suspected
  // of leaving the stream of a host-bound loop in a slower mode: profiles/HISTORY.md, round 6, models E / F)
  for (int t = 0; t < n_threads; ++t) H_TRY(hipStreamSynchronize(h->streams[t]));
  const auto t1 = std::chrono::steady_clock::now();
  h->issued += steps;
  for (int t = 0; t < n_threads; ++t)
    if (h->status[t]) return h->status[t];
  if (wall_ms) *wall_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
  if (dev_ms) H_TRY(hipEventElapsedTime(dev_ms, h->e0, h->e1));
  if (iter_ms) {
    for (int k = 0; k < steps; ++k) H_TRY(hipEventElapsedTime(&iter_ms[k], ev[2 * k], ev[2 * k + 1]));
    for (auto &e : ev) (void)hipEventDestroy(e);
  }
  return FCP_OK;
}

// Same protocol, but `group` consecutive requests of worker 0 are captured ONCE into a HIP
// graph (fcp_process_feature_columns is capture-safe when its descriptors are cached: it then
// only enqueues kernels) and the graph is replayed steps / group times.  For fixed-shape
// models (one-hot columns: S2, DLRM); a request with new shapes needs a new capture.
int fcp_harness_run_graph(fcp_harness *h, int steps, int group, double *wall_ms, float *dev_ms) {
  if (!h || group < 1 || steps < group || steps % group) return FCP_ERR_INVALID_ARGUMENT;
  OnDevice on(h->device);
  hipStream_t s = h->streams[0];
  h->issue(0, 0, group); // descriptors of the captured variants are now resident
  H_TRY(hipStreamSynchronize(s));
  if (h->status[0]) return h->status[0];
  h->rings[0].next = 0;
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  H_TRY(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  h->issue(0, 0, group);
  H_TRY(hipStreamEndCapture(s, &graph));
  if (h->status[0]) return h->status[0];
  H_TRY(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
  H_TRY(hipGraphLaunch(exec, s)); // warm-up replay
  H_TRY(hipStreamSynchronize(s));
  const auto t0 = std::chrono::steady_clock::now();
  H_TRY(hipEventRecord(h->e0, s));
  for (int k = 0; k < steps / group; ++k) H_TRY(hipGraphLaunch(exec, s));
  H_TRY(hipEventRecord(h->e1, s));
  H_TRY(hipStreamSynchronize(s));
  const auto t1 = std::chrono::steady_clock::now();
  if (wall_ms) *wall_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
  if (dev_ms) H_TRY(hipEventElapsedTime(dev_ms, h->e0, h->e1));
  (void)hipGraphExecDestroy(exec);
  (void)hipGraphDestroy(graph);
  return FCP_OK;
}

// One host thread, ONE caller stream — what a TensorFlow op sees — over a plan with private streams
// (fcp_plan_set_private_streams, set by the caller of this function): request k is issued on the caller's stream, its
// kernels run on a plan-owned lane, and the consumer of request k - (depth - 1) — fcp_result_wait on the caller's stream
// followed by a small kernel that READS the finished arena there, standing for Addons>ConcatOutputs and the layers
// behind it — is enqueued right after it, so `depth` requests are in flight.  Timed like fcp_harness_run: events on the
// caller's stream (which has waited for every result by the end) and the host clock around the loop.
__global__ void fcp_consume_probe_kernel(const float *arena, size_t n_floats, float *sink) {
  // first and last 256 floats of the result: a reader on the caller's stream, ordered behind the wait
  float acc = 0.f;
  const size_t i = threadIdx.x;
  if (i < n_floats) acc += arena[i] + arena[n_floats - 1 - i];
  if (acc == 12345.678f) *sink = acc;
}

// `threads` host threads (1 = the loop described above) issue `steps` requests EACH on the one caller stream — the
// reference's serving protocol: serve_workers threads share ONE Session, hence one compute stream
// (examples/cc/recom_examples.patch:193-216).  Every thread keeps `depth` of its own requests in flight (its own arena ring)
// and enqueues their consumers on the caller's stream; the host time of a request (event record, lane wait, launch,
// registry, consumer) is split between the threads.  With the plan's private streams OFF the same loop is "stream order +
// the same consumer": the figure a private-stream number has to beat.
int fcp_harness_run_private_threads(fcp_harness *h, int steps, int depth, int threads, double *wall_ms, float *dev_ms) {
  if (!h || steps < 1 || depth < 1 || threads < 1 || threads > (int)h->rings.size() || depth > (int)h->rings[0].bufs.size())
    return FCP_ERR_INVALID_ARGUMENT;
  OnDevice on(h->device);
  hipStream_t caller = h->streams[0];
  static float *sink = nullptr;
  if (!sink) H_TRY(hipMalloc(&sink, sizeof(float)));
  const int nv = (int)h->variants.size();
  static const bool reader = [] { // tuning aid: FCP_HARNESS_NO_READER=1 leaves the wait alone on the caller's stream
    const char *v = std::getenv("FCP_HARNESS_NO_READER");
    return !(v && std::atoi(v) != 0);
  }();
  static const bool no_wait = [] { // diagnostic: FCP_HARNESS_NO_WAIT=1 drops the consumer altogether (final sync of the lanes only)
    const char *v = std::getenv("FCP_HARNESS_NO_WAIT");
    return v && std::atoi(v) != 0;
  }();
  static const bool stats = std::getenv("FCP_HARNESS_STATS") != nullptr; // diagnostic: host time inside the two calls
  auto consume = [&](const std::pair<void *, int64_t> &res) -> int {
    if (no_wait) return FCP_OK;
    int rc = fcp_result_wait(res.first, caller);
    if (rc || !reader) return rc;
    hipLaunchKernelGGL(fcp_consume_probe_kernel, dim3(1), dim3(256), 0, caller, static_cast<const float *>(res.first),
                       (size_t)(res.second / 4), sink);
    return FCP_OK;
  };
  const long begin = h->issued;
  auto serve = [&](int t) -> int {
    std::vector<std::pair<void *, int64_t>> pending; // results not consumed yet, oldest first
    size_t head = 0;
    double ns_process = 0, ns_consume = 0;
    auto now = [] { return std::chrono::steady_clock::now(); };
    for (long k = begin; k < begin + steps; ++k) {
      fcp_process_args_t a = h->variants[(size_t)((k * threads + t) % nv)];
      a.stream = caller;
      a.malloc_buff = ring_alloc;
      a.malloc_buff_ctx = &h->rings[t];
      a.malloc_temp = nullptr;
      a.malloc_temp_ctx = nullptr;
      fcp_process_result_t res{};
      const auto p0 = now();
      int rc = fcp_process_feature_columns(h->plan, &a, &res);
      if (rc) return rc;
      const auto p1 = now();
      pending.emplace_back(res.buffer, res.buffer_bytes);
      if ((int)(pending.size() - head) >= depth) {
        rc = consume(pending[head++]);
        if (rc) return rc;
      }
      if (stats) {
        ns_process += std::chrono::duration<double, std::nano>(p1 - p0).count();
        ns_consume += std::chrono::duration<double, std::nano>(now() - p1).count();
      }
    }
    if (stats)
      std::fprintf(stderr, "fcp_harness_run_private: thread %d of %d, depth %d, host time per request: process call %.2f us, consumer (wait + reader) %.2f us\n",
                   t, threads, depth, ns_process / steps / 1e3, ns_consume / steps / 1e3);
    while (head < pending.size()) {
      const int rc = consume(pending[head++]);
      if (rc) return rc;
    }
    return FCP_OK;
  };
  const auto t0 = std::chrono::steady_clock::now();
  H_TRY(hipEventRecord(h->e0, caller));
  std::vector<int> rcs((size_t)threads, FCP_OK);
  std::vector<std::thread> workers;
  for (int t = 1; t < threads; ++t) workers.emplace_back([&, t] {
      (void)hipSetDevice(h->device);
      rcs[(size_t)t] = serve(t);
    });
  rcs[0] = serve(0);
  for (auto &w : workers) w.join();
  for (int rc : rcs)
    if (rc) return rc;
  if (no_wait) H_TRY(hipDeviceSynchronize());
  H_TRY(hipEventRecord(h->e1, caller));
  H_TRY(hipStreamSynchronize(caller));
  const auto t1 = std::chrono::steady_clock::now();
  h->issued += steps;
  if (wall_ms) *wall_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
  if (dev_ms) H_TRY(hipEventElapsedTime(dev_ms, h->e0, h->e1));
  return FCP_OK;
}

int fcp_harness_run_private(fcp_harness *h, int steps, int depth, double *wall_ms, float *dev_ms) {
  return fcp_harness_run_private_threads(h, steps, depth, 1, wall_ms, dev_ms);
}

// worker `t`'s stream (what fcp_harness_run_private issues on is worker 0's): for probes that need the caller's stream
void *fcp_harness_stream(fcp_harness *h, int t) { return (h && t >= 0 && t < (int)h->streams.size()) ? h->streams[t] : nullptr; }

int fcp_harness_destroy(fcp_harness *h) {
  if (!h) return FCP_OK;
  OnDevice on(h->device);
  for (size_t t = 0; t < h->streams.size(); ++t) {
    (void)hipStreamSynchronize(h->streams[t]);
    for (void *p : h->rings[t].bufs) (void)hipFree(p);
    (void)hipStreamDestroy(h->streams[t]);
  }
  if (h->e0) (void)hipEventDestroy(h->e0);
  if (h->e1) (void)hipEventDestroy(h->e1);
  delete h;
  return FCP_OK;
}

// Device-side copy bandwidth probe (float4 copy of `bytes` bytes, `iters` times): the "measured copy
// peak" the roofline is also quoted against.  One element per thread — the shape that reaches the
// highest rate on MI355X (6.0-6.25 TB/s read+write; a grid-stride loop over the same buffers only
// reaches 4.6-4.95 TB/s, scripts/probes/copy_variants.hip, profiles/HISTORY.md, round 1).
__global__ void __launch_bounds__(256) fcp_copy_probe_kernel(const float4 *__restrict__ src,
                                                             float4 *__restrict__ dst, size_t n) {
  typedef float __attribute__((ext_vector_type(4))) f4;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n)
    __builtin_nontemporal_store(__builtin_nontemporal_load(reinterpret_cast<const f4 *>(src) + i),
                                reinterpret_cast<f4 *>(dst) + i);
}

int fcp_harness_copy_probe(size_t bytes, int iters, float *ms_per_iter) {
  void *a = nullptr, *b = nullptr;
  H_TRY(hipMalloc(&a, bytes));
  H_TRY(hipMalloc(&b, bytes));
  H_TRY(hipMemset(a, 1, bytes));
  hipEvent_t e0, e1;
  H_TRY(hipEventCreate(&e0));
  H_TRY(hipEventCreate(&e1));
  const size_t n = bytes / 16;
  hipLaunchKernelGGL(fcp_copy_probe_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, (const float4 *)a, (float4 *)b, n);
  H_TRY(hipDeviceSynchronize());
  H_TRY(hipEventRecord(e0, 0));
  for (int i = 0; i < iters; ++i)
    hipLaunchKernelGGL(fcp_copy_probe_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, (const float4 *)a, (float4 *)b, n);
  H_TRY(hipEventRecord(e1, 0));
  H_TRY(hipDeviceSynchronize());
  float ms = 0;
  H_TRY(hipEventElapsedTime(&ms, e0, e1));
  if (ms_per_iter) *ms_per_iter = ms / iters;
  (void)hipFree(a);
  (void)hipFree(b);
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return FCP_OK;
}


// ---- pure read / pure write / chunked-write probes ------------------------------
typedef float __attribute__((ext_vector_type(4))) probe_f4;

__global__ void __launch_bounds__(256) fcp_read_probe_kernel(const probe_f4 *__restrict__ src, float *sink, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  probe_f4 acc = {0, 0, 0, 0};
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + 3 * stride < n; i += 4 * stride) {
    const probe_f4 a = __builtin_nontemporal_load(src + i), b = __builtin_nontemporal_load(src + i + stride),
                   c = __builtin_nontemporal_load(src + i + 2 * stride), d = __builtin_nontemporal_load(src + i + 3 * stride);
    acc += a + b + c + d;
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) *sink = acc.x;
}

__global__ void __launch_bounds__(256) fcp_write_probe_kernel(probe_f4 *__restrict__ dst, size_t n, int nt) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const probe_f4 v = {1.f, 2.f, 3.f, (float)threadIdx.x};
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    if (nt) __builtin_nontemporal_store(v, dst + i); else dst[i] = v;
  }
}

// our store shape: a wave writes `chunk` contiguous bytes of one row, a block 4*R rows
// of a [rows, width_bytes] matrix; blocks cover (row tile, chunk index).
__global__ void __launch_bounds__(256) fcp_chunk_write_probe_kernel(char *__restrict__ dst, int rows, int width_bytes,
                                                                    int R, int nchunks) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int chunk = blockIdx.x % nchunks, tile = blockIdx.x / nchunks;
  const probe_f4 v = {1.f, 2.f, 3.f, (float)lane};
  const size_t col = (size_t)chunk * 1024 + lane * 16;
  if (col + 16 > (size_t)width_bytes) return;
  for (int r = 0; r < R; ++r) {
    const int b = (tile * 4 + wave) * R + r;
    if (b < rows) __builtin_nontemporal_store(v, reinterpret_cast<probe_f4 *>(dst + (size_t)b * width_bytes + col));
  }
}

// kind: 0 read, 1 write (default policy), 2 write (nt), 3 chunked nt write [rows=512*mult, width=120000 B]
// Random-gather read probe: the memory system's ceiling for THIS path's access shape.  Every group of
// row_bytes/16 lanes reads one pseudo-random, row_bytes-aligned row of a `bytes`-sized buffer (16 B per
// lane), `depth` independent rows in flight per lane, nothing is written.  Returns the useful bytes
// gathered per launch through *useful.
__global__ void __launch_bounds__(256) fcp_gather_probe_kernel(const probe_f4 *__restrict__ src, float *sink,
                                                               unsigned long long n_rows, int lanes_per_row,
                                                               int depth, int rounds) {
  const unsigned long long gid = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned long long grp = gid / lanes_per_row;
  const int part = (int)(gid % lanes_per_row);
  probe_f4 acc = {0, 0, 0, 0};
  unsigned long long x = grp * 0x9E3779B97F4A7C15ull + 12345;
  for (int r = 0; r < rounds; ++r) {
    probe_f4 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      x ^= x >> 29;
      x *= 0xBF58476D1CE4E5B9ull;
      x ^= x >> 32;
      const unsigned long long row = x % n_rows;
      v[k] = k < depth ? src[row * lanes_per_row + part] : probe_f4{0, 0, 0, 0};
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) acc += v[k];
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) *sink = acc.x;
}

int fcp_harness_gather_probe(size_t bytes, int row_bytes, int depth, int iters, float *ms_per_iter, double *useful) {
  if (row_bytes < 16 || row_bytes % 16 || depth < 1 || depth > 8) return FCP_ERR_INVALID_ARGUMENT;
  void *a = nullptr;
  float *sink = nullptr;
  H_TRY(hipMalloc(&a, bytes));
  H_TRY(hipMalloc(&sink, 4));
  H_TRY(hipMemset(a, 1, bytes));
  hipEvent_t e0, e1;
  H_TRY(hipEventCreate(&e0));
  H_TRY(hipEventCreate(&e1));
  const int lanes_per_row = row_bytes / 16, rounds = 4;
  const unsigned long long n_rows = bytes / row_bytes;
  const int blocks = 256 * 8 * 4; // 8 blocks per CU resident, four generations
  auto launch = [&]() {
    hipLaunchKernelGGL(fcp_gather_probe_kernel, dim3(blocks), dim3(256), 0, 0, (const probe_f4 *)a, sink, n_rows,
                       lanes_per_row, depth, rounds);
  };
  launch();
  H_TRY(hipDeviceSynchronize());
  H_TRY(hipEventRecord(e0, 0));
  for (int i = 0; i < iters; ++i) launch();
  H_TRY(hipEventRecord(e1, 0));
  H_TRY(hipDeviceSynchronize());
  float ms = 0;
  H_TRY(hipEventElapsedTime(&ms, e0, e1));
  if (ms_per_iter) *ms_per_iter = ms / iters;
  if (useful) *useful = (double)blocks * 256 * 16.0 * depth * rounds;
  (void)hipFree(a);
  (void)hipFree(sink);
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return FCP_OK;
}

int fcp_harness_bw_probe(int kind, size_t bytes, int iters, float *ms_per_iter) {
  void *a = nullptr;
  float *sink = nullptr;
  H_TRY(hipMalloc(&a, bytes));
  H_TRY(hipMalloc(&sink, 4));
  H_TRY(hipMemset(a, 1, bytes));
  hipEvent_t e0, e1;
  H_TRY(hipEventCreate(&e0));
  H_TRY(hipEventCreate(&e1));
  const size_t n = bytes / 16;
  const int width = 120000, rows = (int)(bytes / width), R = 4, nchunks = (width + 1023) / 1024;
  auto launch = [&]() {
    if (kind == 0) hipLaunchKernelGGL(fcp_read_probe_kernel, dim3(2048), dim3(256), 0, 0, (const probe_f4 *)a, sink, n);
    else if (kind == 1) hipLaunchKernelGGL(fcp_write_probe_kernel, dim3(2048), dim3(256), 0, 0, (probe_f4 *)a, n, 0);
    else if (kind == 2) hipLaunchKernelGGL(fcp_write_probe_kernel, dim3(2048), dim3(256), 0, 0, (probe_f4 *)a, n, 1);
    else hipLaunchKernelGGL(fcp_chunk_write_probe_kernel, dim3(nchunks * ((rows + 4 * R - 1) / (4 * R))), dim3(256), 0, 0,
                            (char *)a, rows, width, R, nchunks);
  };
  launch();
  H_TRY(hipDeviceSynchronize());
  H_TRY(hipEventRecord(e0, 0));
  for (int i = 0; i < iters; ++i) launch();
  H_TRY(hipEventRecord(e1, 0));
  H_TRY(hipDeviceSynchronize());
  float ms = 0;
  H_TRY(hipEventElapsedTime(&ms, e0, e1));
  if (ms_per_iter) *ms_per_iter = ms / iters;
  (void)hipFree(a);
  (void)hipFree(sink);
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return FCP_OK;
}

} // extern "C"
