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

extern "C" {

// Runs `warmup` untimed + `steps` timed requests per worker thread.
//   variants[n_variants]  request descriptors that are cycled through (their
//                         stream / allocator fields are overwritten here);
//   arena_ring            number of pre-allocated output arenas per worker;
//   n_threads             serve_workers: host threads sharing the plan, one
//                         stream each;
//   wall_ms               host wall time of the timed region (all workers,
//                         including the final stream synchronisation);
//   dev_ms                HIP-event time over worker 0's timed region, taken on
//                         the stream the kernels are launched on;
//   iter_ms[steps]        optional: per-request device time of worker 0 from
//                         HIP event pairs, measured in a SEPARATE pass after the
//                         timed region (so the events do not perturb it).
int fcp_harness_run(fcp_plan_t *plan, const fcp_process_args_t *variants, int n_variants, int arena_ring,
                    int steps, int warmup, int n_threads, double *wall_ms, float *dev_ms, float *iter_ms) {
  if (!plan || !variants || n_variants < 1 || steps < 1 || n_threads < 1 || arena_ring < 1)
    return FCP_ERR_INVALID_ARGUMENT;
  int64_t arena_bytes = 0;
  for (int v = 0; v < n_variants; ++v) {
    int64_t b = 0;
    int rc = fcp_plan_arena_bytes(plan, variants[v].concated_shapes, variants[v].symbols, &b);
    if (rc) return rc;
    arena_bytes = std::max(arena_bytes, b);
  }
  std::vector<Ring> rings(n_threads);
  std::vector<hipStream_t> streams(n_threads);
  for (int t = 0; t < n_threads; ++t) {
    H_TRY(hipStreamCreateWithFlags(&streams[t], hipStreamNonBlocking));
    rings[t].bytes = (size_t)arena_bytes;
    for (int i = 0; i < arena_ring; ++i) {
      void *p = nullptr;
      H_TRY(hipMalloc(&p, (size_t)std::max<int64_t>(arena_bytes, 256)));
      rings[t].bufs.push_back(p);
    }
  }
  hipEvent_t e0, e1;
  H_TRY(hipEventCreate(&e0));
  H_TRY(hipEventCreate(&e1));

  std::vector<int> status(n_threads, FCP_OK);
  auto issue = [&](int t, int begin, int count) {
    for (int k = begin; k < begin + count; ++k) {
      fcp_process_args_t a = variants[(k + t) % n_variants];
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
  };

  // warm-up (binds tables, fills the descriptor cache) — untimed
  for (int t = 0; t < n_threads; ++t) issue(t, 0, std::max(warmup, 1));
  for (int t = 0; t < n_threads; ++t) H_TRY(hipStreamSynchronize(streams[t]));
  for (int t = 0; t < n_threads; ++t)
    if (status[t]) return status[t];

  // timed region: exactly `steps` requests per worker
  const auto t0 = std::chrono::steady_clock::now();
  H_TRY(hipEventRecord(e0, streams[0]));
  if (n_threads == 1) {
    issue(0, warmup, steps);
  } else {
    std::vector<std::thread> workers;
    for (int t = 1; t < n_threads; ++t) workers.emplace_back(issue, t, warmup, steps);
    issue(0, warmup, steps);
    for (auto &w : workers) w.join();
  }
  H_TRY(hipEventRecord(e1, streams[0]));
  for (int t = 0; t < n_threads; ++t) H_TRY(hipStreamSynchronize(streams[t]));
  const auto t1 = std::chrono::steady_clock::now();
  for (int t = 0; t < n_threads; ++t)
    if (status[t]) return status[t];
  if (wall_ms) *wall_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
  if (dev_ms) H_TRY(hipEventElapsedTime(dev_ms, e0, e1));

  // separate pass: per-request device latency (p50 / p95 material)
  if (iter_ms) {
    std::vector<hipEvent_t> ev(2 * (size_t)steps);
    for (auto &e : ev) H_TRY(hipEventCreate(&e));
    for (int k = 0; k < steps; ++k) {
      H_TRY(hipEventRecord(ev[2 * k], streams[0]));
      issue(0, warmup + k, 1);
      H_TRY(hipEventRecord(ev[2 * k + 1], streams[0]));
    }
    H_TRY(hipStreamSynchronize(streams[0]));
    for (int k = 0; k < steps; ++k) H_TRY(hipEventElapsedTime(&iter_ms[k], ev[2 * k], ev[2 * k + 1]));
    for (auto &e : ev) (void)hipEventDestroy(e);
    if (status[0]) return status[0];
  }

  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  for (int t = 0; t < n_threads; ++t) {
    for (void *p : rings[t].bufs) (void)hipFree(p);
    (void)hipStreamDestroy(streams[t]);
  }
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
