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
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
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

} // namespace

struct fcp_harness {
  fcp_plan_t *plan = nullptr;
  std::vector<fcp_process_args_t> variants;
  std::vector<Ring> rings;
  std::vector<hipStream_t> streams;
  std::vector<int> status;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  long issued = 0; // requests issued so far per worker (variant rotation)

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
  for (int t = 1; t < n_threads; ++t) workers.emplace_back([h, t, begin, steps] { h->issue(t, begin, steps); });
  if (iter_ms) {
    for (int k = 0; k < steps; ++k) {
      H_TRY(hipEventRecord(ev[2 * k], h->streams[0]));
      h->issue(0, begin + k, 1);
      H_TRY(hipEventRecord(ev[2 * k + 1], h->streams[0]));
    }
  } else {
    h->issue(0, begin, steps);
  }
  for (auto &w : workers) w.join();
  H_TRY(hipEventRecord(h->e1, h->streams[0]));
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

int fcp_harness_destroy(fcp_harness *h) {
  if (!h) return FCP_OK;
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

// Device-side copy bandwidth probe (float4 copy of `bytes` bytes, `iters`
// times): the "measured copy peak" the roofline is also quoted against.
__global__ void fcp_copy_probe_kernel(const float4 *__restrict__ src, float4 *__restrict__ dst, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    dst[i] = src[i];
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
  hipLaunchKernelGGL(fcp_copy_probe_kernel, dim3(256 * 8), dim3(256), 0, 0, (const float4 *)a, (float4 *)b, n);
  H_TRY(hipDeviceSynchronize());
  H_TRY(hipEventRecord(e0, 0));
  for (int i = 0; i < iters; ++i)
    hipLaunchKernelGGL(fcp_copy_probe_kernel, dim3(256 * 8), dim3(256), 0, 0, (const float4 *)a, (float4 *)b, n);
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

} // extern "C"
