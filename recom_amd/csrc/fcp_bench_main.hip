// fcp_bench — torch-free native driver for the S2-style workload (used for
// rocprofv3 PMC passes and kernel ablations; bench.py remains the contract).
//
// Counterpart of the reference's `benchmark_multi_thread` CLI
// (examples/cc/recom_examples.patch:98-263): builds the synthetic model (same
// closed-form tables and column recipe as recom_amd/synth.py model_s2), keeps
// `--requests` request blobs resident in HBM and drives libfcp_harness.
//
//   fcp_bench [--columns 1000] [--batch 512] [--vocab 1000000] [--steps 300]
//             [--warmup 50] [--threads 1] [--requests 16] [--ring 6] [--verify 1]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <thread>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/fcp_hip.h"
#include "numa_util.h"

struct fcp_harness;
extern "C" int fcp_harness_create(fcp_plan_t *, const fcp_process_args_t *, int, int, int, fcp_harness **);
extern "C" int fcp_harness_run(fcp_harness *, int, double *, float *, float *);
extern "C" int fcp_harness_destroy(fcp_harness *);
extern "C" int fcp_harness_copy_probe(size_t, int, float *);
#if defined(FCP_STAMPS)
extern "C" int fcp_debug_read_stamps(fcp_plan_t *, unsigned long long *, int);
#endif
extern "C" int fcp_harness_bw_probe(int, size_t, int, float *);
extern "C" int fcp_harness_gather_probe(size_t, int, int, int, float *, double *);

#define CHECK_HIP(e)                                                                        \
  do {                                                                                      \
    hipError_t e_ = (e);                                                                    \
    if (e_ != hipSuccess) {                                                                 \
      std::fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #e, hipGetErrorString(e_)); \
      std::exit(1);                                                                         \
    }                                                                                       \
  } while (0)
#define CHECK_FCP(e)                                                                                \
  do {                                                                                              \
    int s_ = (e);                                                                                   \
    if (s_) {                                                                                       \
      std::fprintf(stderr, "%s:%d %s: %s (%s)\n", __FILE__, __LINE__, #e, fcp_status_string(s_),   \
                   fcp_last_error());                                                               \
      std::exit(1);                                                                                 \
    }                                                                                               \
  } while (0)

// table[t][r][e] — identical to recom_amd/synth.py hash_rows()
__host__ __device__ inline float hash_elem(uint32_t seed, uint64_t r, uint32_t e) {
  uint64_t u = (r * 2654435761ull + (uint64_t)e * 40503ull + ((seed * 7919u + 12345u) & 0xFFFFFFFFull)) & 0xFFFFFFFFull;
  u ^= u >> 15;
  u = (u * 0x2C1B3C6Dull) & 0xFFFFFFFFull;
  u ^= u >> 12;
  u = (u * 0x297A2D39ull) & 0xFFFFFFFFull;
  u ^= u >> 15;
  return (float)(u >> 8) * 1.1920928955078125e-07f - 1.0f; // 2^-23
}

struct BigArg { char bytes[640]; };
__global__ void noop_big_kernel(BigArg a) { if (a.bytes[0] == 77) __builtin_trap(); }
__global__ void noop_kernel(int *p) { if (p) *p = 1; }

__global__ void fill_table(float *t, uint32_t seed, uint64_t vocab, uint32_t dim) {
  const uint64_t n = vocab * dim;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
    t[i] = hash_elem(seed, i / dim, (uint32_t)(i % dim));
}

static uint64_t splitmix(uint64_t &s) {
  uint64_t z = (s += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

static int bucketize_host(const std::vector<float> &b, float v) {
  int l = 0, r = (int)b.size() - 1;
  while (l <= r) {
    int mid = (l + r) >> 1;
    if (v < b[mid]) r = mid - 1; else l = mid + 1;
  }
  return r + 1;
}

static void *plain_alloc(void *ctx, size_t bytes) {
  void **slot = static_cast<void **>(ctx);
  if (hipMalloc(slot, bytes) != hipSuccess) return nullptr;
  return *slot;
}

int main(int argc, char **argv) {
  int columns = 1000, batch = 512, steps = 300, warmup = 50, threads = 1, requests = 16, ring = 6, verify = 1;
  int bucketize_every = 10; // every N-th column is bucketize-f32 sourced (0 = none)
  int slab = 0;             // 1: all tables carved from ONE hipMalloc (as under TF's BFC allocator)
  int h2d = 0;              // 1: PCIe-inclusive loop: stage (pack + H2D) every request, then process
  int narrow = 0;           // with --h2d: ship int64 ids as int32 (fcp_stager_stage_narrow)
  int zero_copy = std::getenv("FCP_STAGER_ZERO_COPY") ? 1 : 0; // with --h2d: no copy, the kernel reads the pinned ring over PCIe
  int fixed_dim = 0;        // 0: dims cycle 8/16/32/64 (S2); D: every column has dim D (E/F-like models: --dim 8)
  int pack_threads = 8;
  int stager_depth = 4;     // with --h2d: slots of the stager's pinned ring
  int copy_kernel = 1;      // with --h2d: the stager's copies are kernels reading the pinned ring (the default); 0: hipMemcpyAsync (SDMA)
  long vocab = 1000000;
  for (int i = 1; i + 1 < argc; i += 2) {
    std::string k = argv[i];
    long v = std::atol(argv[i + 1]);
    if (k == "--columns") columns = (int)v;
    else if (k == "--batch") batch = (int)v;
    else if (k == "--vocab") vocab = v;
    else if (k == "--steps") steps = (int)v;
    else if (k == "--warmup") warmup = (int)v;
    else if (k == "--threads") threads = (int)v;
    else if (k == "--requests") requests = (int)v;
    else if (k == "--ring") ring = (int)v;
    else if (k == "--verify") verify = (int)v;
    else if (k == "--bucketize-every") bucketize_every = (int)v;
    else if (k == "--slab") slab = (int)v;
    else if (k == "--h2d") h2d = (int)v;
    else if (k == "--narrow") narrow = (int)v;
    else if (k == "--zero-copy") zero_copy = (int)v;
    else if (k == "--dim") fixed_dim = (int)v;
    else if (k == "--pack-threads") pack_threads = (int)v;
    else if (k == "--stager-depth") stager_depth = (int)v;
    else if (k == "--copy-kernel") copy_kernel = (int)v;
    else if (k == "--bw-probe") {
      const char *names[4] = {"read", "write", "write-nt", "chunked-write-nt"};
      for (int kind = 0; kind < 4; ++kind) {
        float ms = 0;
        const size_t bytes = (size_t)v << 20;
        fcp_harness_bw_probe(kind, bytes, 20, &ms);
        std::printf("%s probe %ld MiB: %.1f GB/s  (%.2f us)\n", names[kind], v, bytes / (ms * 1e-3) / 1e9, ms * 1e3);
      }
      return 0;
    }
    else if (k == "--gather-probe") {
      // random row gathers from a v-MiB buffer: useful GB/s by row size and rows in flight per lane
      for (int row_bytes : {32, 64, 128, 256}) {
        for (int depth : {4, 8}) {
          float ms = 0;
          double useful = 0;
          fcp_harness_gather_probe((size_t)v << 20, row_bytes, depth, 10, &ms, &useful);
          std::printf("gather probe %ld MiB, %3d-byte rows, %d in flight: %.1f GB/s useful (%.2f us per launch)\n", v, row_bytes,
                      depth, useful / (ms * 1e-3) / 1e9, ms * 1e3);
        }
      }
      return 0;
    }
    else if (k == "--api-probe") {
      // host-side cost of the HIP calls on the request path (us per call)
      hipStream_t s1, s2;
      hipEvent_t e1, e2;
      CHECK_HIP(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
      CHECK_HIP(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
      CHECK_HIP(hipEventCreateWithFlags(&e1, hipEventDisableTiming));
      CHECK_HIP(hipEventCreateWithFlags(&e2, hipEventDisableTiming));
      const int n = (int)v;
      auto now = [] { return std::chrono::steady_clock::now(); };
      auto us = [&](std::chrono::steady_clock::time_point a) { return std::chrono::duration<double, std::micro>(now() - a).count() / n; };
      {
        // the same launch through the module API with a pre-resolved function handle and a raw argument
        // buffer of the size of FcpLaunch (what fcp_launch_fused would pass)
        hipFunction_t f = nullptr;
        CHECK_HIP(hipGetFuncBySymbol(&f, reinterpret_cast<const void *>(noop_big_kernel)));
        BigArg arg;
        std::memset(&arg, 0, sizeof(arg));
        size_t arg_size = sizeof(arg);
        void *extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &arg, HIP_LAUNCH_PARAM_BUFFER_SIZE, &arg_size, HIP_LAUNCH_PARAM_END};
        auto t = now();
        for (int i = 0; i < n; ++i) hipLaunchKernelGGL(noop_big_kernel, dim3(1), dim3(64), 0, s1, arg);
        std::printf("hipLaunchKernelGGL, 640-byte argument    %.2f us\n", us(t));
        CHECK_HIP(hipStreamSynchronize(s1));
        t = now();
        for (int i = 0; i < n; ++i) CHECK_HIP(hipModuleLaunchKernel(f, 1, 1, 1, 64, 1, 1, 0, s1, nullptr, extra));
        std::printf("hipModuleLaunchKernel, same argument     %.2f us\n", us(t));
        CHECK_HIP(hipStreamSynchronize(s1));
      }
      auto t = now();
      for (int i = 0; i < n; ++i) hipLaunchKernelGGL(noop_kernel, dim3(1), dim3(64), 0, s1, nullptr);
      std::printf("kernel launch          %.2f us\n", us(t));
      CHECK_HIP(hipDeviceSynchronize());
      t = now();
      for (int i = 0; i < n; ++i) CHECK_HIP(hipEventRecord(e1, s1));
      std::printf("hipEventRecord         %.2f us\n", us(t));
      CHECK_HIP(hipDeviceSynchronize());
      t = now();
      for (int i = 0; i < n; ++i) CHECK_HIP(hipStreamWaitEvent(s2, e1, 0));
      std::printf("hipStreamWaitEvent     %.2f us\n", us(t));
      CHECK_HIP(hipDeviceSynchronize());
      t = now();
      for (int i = 0; i < n; ++i) (void)hipEventQuery(e1);
      std::printf("hipEventQuery          %.2f us\n", us(t));
      t = now();
      for (int i = 0; i < n; ++i) CHECK_HIP(hipEventSynchronize(e1));
      std::printf("hipEventSynchronize    %.2f us\n", us(t));
      t = now();
      for (int i = 0; i < n; ++i) {
        hipLaunchKernelGGL(noop_kernel, dim3(1), dim3(64), 0, s2, nullptr);
        CHECK_HIP(hipEventRecord(e2, s2));
        CHECK_HIP(hipStreamWaitEvent(s1, e2, 0));
        hipLaunchKernelGGL(noop_kernel, dim3(1), dim3(64), 0, s1, nullptr);
        CHECK_HIP(hipEventRecord(e1, s1));
      }
      std::printf("upload+record+wait+launch+record  %.2f us (host issue)\n", us(t));
      CHECK_HIP(hipDeviceSynchronize());
      std::printf("  ... including drain  %.2f us\n", us(t));
      return 0;
    }
    else if (k == "--copy-probe") {
      float ms = 0;
      fcp_harness_copy_probe((size_t)v << 20, 20, &ms);
      std::printf("copy probe %ld MiB: %.1f GB/s (read+write)\n", v, 2.0 * ((size_t)v << 20) / (ms * 1e-3) / 1e9);
      return 0;
    }
    else { std::fprintf(stderr, "unknown flag %s\n", k.c_str()); return 2; }
  }
  const int dims[4] = {8, 16, 32, 64};
  std::vector<float> boundaries(100);
  for (int i = 0; i < 100; ++i) boundaries[i] = 5.0f * i;

  // ---- plan (model_s2 of recom_amd/synth.py) ----------------------------------
  std::vector<fcp_column_desc_t> cols(columns);
  std::vector<int32_t> ranks(columns, 1), esz(columns);
  for (int c = 0; c < columns; ++c) {
    fcp_column_desc_t &d = cols[c];
    std::memset(&d, 0, sizeof(d));
    const bool bkt = bucketize_every > 0 && c % bucketize_every == 0;
    d.form = FCP_FORM_GATHER;
    d.dim = fixed_dim > 0 ? fixed_dim : dims[c % 4];
    d.vocab = vocab;
    d.id_source = bkt ? FCP_IDS_F32_BUCKETIZE : FCP_IDS_I64;
    d.table_input = c;
    d.ids_input = c;
    d.seg_input = -1;
    d.seg_stride = 1;
    d.rows_source = FCP_ROWS_FROM_IDS;
    d.n_boundaries = bkt ? 100 : 0;
    d.boundaries = bkt ? boundaries.data() : nullptr;
    d.concat_group = 0;
    d.concat_slot = c;
    esz[c] = bkt ? 4 : 8;
  }
  fcp_plan_desc_t pd;
  std::memset(&pd, 0, sizeof(pd));
  pd.abi_version = FCP_ABI_VERSION;
  pd.n_columns = columns;
  pd.columns = cols.data();
  pd.n_host_inputs = columns;
  pd.host_input_ranks = ranks.data();
  pd.host_input_elem_sizes = esz.data();
  pd.n_device_inputs = columns;
  pd.n_groups = 1;
  pd.layout = FCP_LAYOUT_CONCAT;
  pd.shard_world = 1;
  fcp_plan_t *plan = nullptr;
  CHECK_FCP(fcp_plan_create(&pd, &plan));

  // ---- tables -------------------------------------------------------------------
  std::vector<void *> tables(columns);
  double table_bytes = 0;
  char *slab_base = nullptr;
  size_t slab_off = 0;
  if (slab) {
    size_t total = 0;
    for (int c = 0; c < columns; ++c) total += ((size_t)vocab * cols[c].dim * 4 + 255) / 256 * 256;
    CHECK_HIP(hipMalloc(reinterpret_cast<void **>(&slab_base), total));
  }
  for (int c = 0; c < columns; ++c) {
    const size_t bytes = (size_t)vocab * cols[c].dim * 4;
    if (slab) {
      tables[c] = slab_base + slab_off;
      slab_off += (bytes + 255) / 256 * 256;
    } else {
      CHECK_HIP(hipMalloc(&tables[c], bytes));
    }
    hipLaunchKernelGGL(fill_table, dim3(2048), dim3(256), 0, 0, (float *)tables[c], (uint32_t)(1000 + c),
                       (uint64_t)vocab, (uint32_t)cols[c].dim);
    table_bytes += bytes;
  }
  CHECK_HIP(hipDeviceSynchronize());

  // ---- resident requests ------------------------------------------------------------
  std::vector<std::vector<char>> blobs(requests);
  std::vector<std::vector<int32_t>> offs(requests), shps(requests);
  std::vector<void *> d_blobs(requests);
  std::vector<fcp_process_args_t> variants(requests);
  uint64_t rng = 0x1234;
  for (int v = 0; v < requests; ++v) {
    offs[v].resize(columns);
    shps[v].assign(columns, batch);
    size_t size = 0;
    for (int c = 0; c < columns; ++c) {
      offs[v][c] = (int32_t)size;
      size += (size_t)batch * esz[c];
    }
    blobs[v].resize(size);
    for (int c = 0; c < columns; ++c) {
      char *p = blobs[v].data() + offs[v][c];
      for (int b = 0; b < batch; ++b) {
        if (esz[c] == 4) {
          float x = -5.0f + 505.0f * (float)((splitmix(rng) >> 40) * (1.0 / 16777216.0));
          std::memcpy(p + 4 * b, &x, 4);
        } else {
          int64_t id = (int64_t)(splitmix(rng) % (uint64_t)vocab);
          std::memcpy(p + 8 * b, &id, 8);
        }
      }
    }
    CHECK_HIP(hipMalloc(&d_blobs[v], size));
    CHECK_HIP(hipMemcpy(d_blobs[v], blobs[v].data(), size, hipMemcpyHostToDevice));
    fcp_process_args_t &a = variants[v];
    std::memset(&a, 0, sizeof(a));
    a.concated_inputs = d_blobs[v];
    a.concated_bytes = (int64_t)size;
    a.concated_offsets = offs[v].data();
    a.concated_shapes = shps[v].data();
    a.input_ptrs = tables.data();
  }

  // ---- verification of one request against the closed form (bit-exact) -----------------
  if (verify) {
    void *arena = nullptr;
    fcp_process_args_t a = variants[0];
    a.malloc_buff = plain_alloc;
    a.malloc_buff_ctx = &arena;
    std::vector<void *> gp(1);
    std::vector<int32_t> gs(2);
    fcp_process_result_t res;
    std::memset(&res, 0, sizeof(res));
    res.group_ptrs = gp.data();
    res.group_shapes = gs.data();
    CHECK_FCP(fcp_process_feature_columns(plan, &a, &res));
    CHECK_HIP(hipDeviceSynchronize());
    const size_t W = gs[1];
    std::vector<float> out((size_t)batch * W);
    CHECK_HIP(hipMemcpy(out.data(), gp[0], out.size() * 4, hipMemcpyDeviceToHost));
    size_t bad = 0, off = 0;
    for (int c = 0; c < columns; ++c) {
      const char *p = blobs[0].data() + offs[0][c];
      for (int b = 0; b < batch; ++b) {
        int64_t id;
        if (esz[c] == 4) {
          float x;
          std::memcpy(&x, p + 4 * b, 4);
          id = bucketize_host(boundaries, x);
        } else {
          std::memcpy(&id, p + 8 * b, 8);
        }
        for (int e = 0; e < cols[c].dim; ++e)
          if (out[(size_t)b * W + off + e] != hash_elem(1000 + c, (uint64_t)id, e)) ++bad;
      }
      off += cols[c].dim;
    }
    std::printf("verify: %zu mismatching elements of %zu\n", bad, out.size());
    CHECK_HIP(hipFree(arena));
    if (bad) return 3;
  }

  // ---- PCIe-inclusive loop: host tensors -> pinned ring -> H2D -> kernel ----
  if (h2d) {
    {
      // the driver thread packs too: keep it on the GPU's socket (what `numactl` would do for a server)
      cpu_set_t near;
      if (!std::getenv("FCP_STAGER_NO_PIN") && fcp::cpus_near_device(0, &near)) (void)sched_setaffinity(0, sizeof(near), &near);
    }
    std::vector<uint8_t> nflags(columns, 0);
    size_t shipped = blobs[0].size();
    if (narrow) {
      // same model, ids declared 4-byte: the stager converts while packing
      shipped = 0;
      for (int c = 0; c < columns; ++c) {
        if (cols[c].id_source == FCP_IDS_I64) {
          cols[c].id_source = FCP_IDS_I32;
          esz[c] = 4;
          nflags[c] = 1;
        }
        shipped += (size_t)batch * 4;
      }
      CHECK_FCP(fcp_plan_destroy(plan));
      CHECK_FCP(fcp_plan_create(&pd, &plan));
      for (int c = 0; c < columns; ++c)
        if (nflags[c]) esz[c] = 8; // host tensors stay int64
    }
    std::vector<std::vector<fcp_host_tensor_t>> host(requests, std::vector<fcp_host_tensor_t>(columns));
    std::vector<int64_t> dims(1, batch);
    for (int v = 0; v < requests; ++v)
      for (int c = 0; c < columns; ++c) host[v][c] = {blobs[v].data() + offs[v][c], esz[c], 1, dims.data()};
    int64_t arena_bytes = 0;
    CHECK_FCP(fcp_plan_arena_bytes(plan, shps[0].data(), nullptr, &arena_bytes));
    // `--threads N`: N serve workers (the reference harness' serve_workers), each with its own stager, stream and output
    // ring, staging and serving its own requests: one caller's stage + process calls are ~55 us of host time per
    // request, more than the 45 us the copy takes, so a single caller is host-bound; two share the copy engine.
    struct RingCtx { std::vector<void *> *r; size_t i; };
    struct Worker {
      fcp_stager_t *st = nullptr;
      hipStream_t stream = nullptr;
      std::vector<void *> ring;
      RingCtx rc{nullptr, 0};
      double host_stage_us = 0, host_process_us = 0; // host time inside the two calls
    };
    const int nw = threads < 1 ? 1 : threads;
    std::vector<Worker> W(nw);
    for (Worker &w : W) {
      CHECK_FCP(fcp_stager_create_ex(0, (int64_t)blobs[0].size() + 4096, columns, columns, stager_depth, pack_threads,
                                     zero_copy ? FCP_STAGER_ZERO_COPY : copy_kernel ? FCP_STAGER_COPY_KERNEL : FCP_STAGER_COPY_SDMA, &w.st));
      w.ring.resize(6);
      for (auto &p : w.ring) CHECK_HIP(hipMalloc(&p, (size_t)arena_bytes));
      w.rc.r = &w.ring;
      CHECK_HIP(hipStreamCreateWithFlags(&w.stream, hipStreamNonBlocking));
    }
    auto one = [&](Worker &w, int k) {
      const int v = k % requests;
      fcp_process_args_t a;
      std::memset(&a, 0, sizeof(a));
      const auto h0 = std::chrono::steady_clock::now();
      CHECK_FCP(fcp_stager_stage_narrow(w.st, host[v].data(), columns, narrow ? nflags.data() : nullptr, w.stream,
                                        &a.concated_inputs, &a.concated_bytes, &a.concated_offsets,
                                        &a.concated_shapes));
      const auto h1 = std::chrono::steady_clock::now();
      a.input_ptrs = tables.data();
      a.stream = w.stream;
      a.malloc_buff_ctx = &w.rc;
      a.malloc_buff = [](void *ctx, size_t) -> void * { auto *x = static_cast<RingCtx *>(ctx); return (*x->r)[x->i++ % x->r->size()]; };
      CHECK_FCP(fcp_process_feature_columns(plan, &a, nullptr));
      w.host_stage_us += std::chrono::duration<double, std::micro>(h1 - h0).count();
      w.host_process_us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - h1).count();
    };
    std::atomic<int> ready{0};
    std::atomic<bool> go{false};
    auto serve = [&](int wi) {
      Worker &w = W[wi];
      for (int k = 0; k < warmup + 1; ++k) one(w, k + wi);
      CHECK_HIP(hipStreamSynchronize(w.stream));
      w.host_stage_us = w.host_process_us = 0;
      ready.fetch_add(1);
      while (!go.load(std::memory_order_acquire)) std::this_thread::yield();
      for (int k = 0; k < steps; ++k) one(w, k + wi);
      CHECK_HIP(hipStreamSynchronize(w.stream));
    };
    std::vector<std::thread> pool;
    for (int wi = 1; wi < nw; ++wi) pool.emplace_back(serve, wi);
    // worker 0 runs on this thread (pinned near the GPU above); the others inherit the affinity
    {
      Worker &w = W[0];
      for (int k = 0; k < warmup + 1; ++k) one(w, k);
      CHECK_HIP(hipStreamSynchronize(w.stream));
      w.host_stage_us = w.host_process_us = 0;
      while (ready.load() < nw - 1) std::this_thread::yield();
    }
    const auto t0 = std::chrono::steady_clock::now();
    go.store(true, std::memory_order_release);
    for (int k = 0; k < steps; ++k) one(W[0], k);
    CHECK_HIP(hipStreamSynchronize(W[0].stream));
    for (auto &t : pool) t.join();
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / ((double)steps * nw);
    double hs = 0, hp = 0;
    for (Worker &w : W) {
      hs += w.host_stage_us / steps / nw;
      hp += w.host_process_us / steps / nw;
    }
    // single-request latency: stage + process + sync, nothing else in flight
    double lat = 0;
    for (int k = 0; k < 50; ++k) {
      const auto a0 = std::chrono::steady_clock::now();
      one(W[0], k);
      CHECK_HIP(hipStreamSynchronize(W[0].stream));
      lat += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - a0).count();
    }
    hipStream_t stream = W[0].stream;
    // the floor of this loop: what the box's host-to-device link does with a blob of this size (pinned memory,
    // back-to-back hipMemcpyAsync on one stream: the copies of a pipelined run follow each other the same way)
    double copy_us = 0;
    {
      void *hp_buf = nullptr, *dp_buf = nullptr;
      CHECK_HIP(hipHostMalloc(&hp_buf, shipped, hipHostMallocDefault));
      CHECK_HIP(hipMalloc(&dp_buf, shipped));
      std::memset(hp_buf, 1, shipped);
      for (int k = 0; k < 10; ++k) CHECK_HIP(hipMemcpyAsync(dp_buf, hp_buf, shipped, hipMemcpyHostToDevice, stream));
      CHECK_HIP(hipStreamSynchronize(stream));
      const auto c0 = std::chrono::steady_clock::now();
      for (int k = 0; k < 100; ++k) CHECK_HIP(hipMemcpyAsync(dp_buf, hp_buf, shipped, hipMemcpyHostToDevice, stream));
      CHECK_HIP(hipStreamSynchronize(stream));
      copy_us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - c0).count() / 100;
      CHECK_HIP(hipFree(dp_buf));
      CHECK_HIP(hipHostFree(hp_buf));
    }
    fcp_stager_stats_t sst;
    std::memset(&sst, 0, sizeof(sst));
    CHECK_FCP(fcp_stager_stats(W[0].st, &sst));
    std::printf("{\"pcie_inclusive\": true, \"zero_copy\": %d, \"copy_kernel\": %d, \"serve_workers\": %d, \"pack_threads\": %d, \"blob_MB\": %.2f, \"us_per_request_pipelined\": %.2f, "
                "\"us_latency_single\": %.2f, \"inferences_per_s\": %.0f, \"host_us_stage_call\": %.2f, \"host_us_process_call\": %.2f, "
                "\"h2d_copy_alone_us\": %.2f, \"h2d_GBs\": %.1f, \"stager_calls\": %lld, \"copy_calls\": %lld, \"copy_calls_over_1ms\": %lld, "
                "\"max_copy_call_us\": %.1f, \"zero_copy_fallback_switches\": %lld}\n",
                zero_copy, copy_kernel, nw, pack_threads, shipped / 1e6, us, lat / 50, batch / (us * 1e-6), hs, hp, copy_us, shipped / copy_us / 1e3,
                (long long)sst.calls, (long long)sst.copy_calls, (long long)sst.copy_calls_over_1ms, sst.max_copy_call_us,
                (long long)sst.fallback_switches);
    for (Worker &w : W) CHECK_FCP(fcp_stager_destroy(w.st));
    CHECK_FCP(fcp_plan_destroy(plan));
    return 0;
  }

  // ---- timed run ----------------------------------------------------------------------
  fcp_harness *h = nullptr;
  CHECK_FCP(fcp_harness_create(plan, variants.data(), requests, ring, threads, &h));
  double wall = 0;
  float dev = 0;
  CHECK_FCP(fcp_harness_run(h, warmup > 0 ? warmup : 1, &wall, &dev, nullptr));
  CHECK_FCP(fcp_harness_run(h, steps, &wall, &dev, nullptr));
  double width = 0, idb = 0;
  for (int c = 0; c < columns; ++c) {
    width += cols[c].dim;
    idb += (double)batch * esz[c] + (esz[c] == 4 ? 400.0 : 0.0);
  }
  const double alg = 2.0 * batch * width * 4 + idb;
  const double us = dev * 1e3 / steps;
  std::printf("{\"columns\": %d, \"batch\": %d, \"table_GB\": %.1f, \"steps\": %d, \"threads\": %d, "
              "\"wall_us_per_step\": %.3f, \"dev_us_per_step\": %.3f, \"alg_MB\": %.3f, \"alg_GBs\": %.1f, "
              "\"frac_of_8TBs\": %.4f}\n",
              columns, batch, table_bytes / 1e9, steps, threads, wall * 1e3 / (steps * threads), us, alg / 1e6,
              alg / (us * 1e-6) / 1e9, alg / (us * 1e-6) / 8e12);
#if defined(FCP_STAMPS)
  {
    // per-block timeline of the last launch (diagnostic build)
    const int nb = 8 * ((int)((width / 4 + 63) / 64 + 7) / 8) * ((batch + 15) / 16);
    std::vector<unsigned long long> st(8 * (size_t)nb);
    CHECK_FCP(fcp_debug_read_stamps(plan, st.data(), nb));
    unsigned long long t0 = ~0ull, t1 = 0;
    int live = 0;
    for (int b = 0; b < nb; ++b)
      if (st[8 * b + 3]) {
        ++live;
        t0 = std::min(t0, st[8 * b]);
        t1 = std::max(t1, st[8 * b + 3]);
      }
    std::printf("stamps: %d live blocks, kernel span %.2f us (100 MHz ticks)\n", live, (t1 - t0) / 100.0);
    // histogram of block begin / end times and mean phase durations per bin of begin time:
    // desc = phase 0; or = raw-id issue + first barrier; stage = boundary staging; raw = extra wait
    // until the raw ids have landed; conv = phase 1b; rows = phase 2 (reads + stores acknowledged)
    const unsigned long long bin = (t1 - t0) < 1600 ? 50 : 200; // ticks per histogram bin (0.5 or 2 us)
    const int nbins = (int)((t1 - t0) / bin) + 1;
    std::vector<int> begins(nbins, 0), ends(nbins, 0);
    std::vector<std::vector<double>> d(6, std::vector<double>(nbins, 0.0));
    for (int b = 0; b < nb; ++b) {
      const unsigned long long *o = &st[8 * (size_t)b];
      if (!o[3]) continue;
      const int bb = (int)((o[0] - t0) / bin), be = (int)((o[3] - t0) / bin);
      ++begins[bb];
      ++ends[be];
      d[0][bb] += (o[1] - o[0]) / 100.0;
      d[1][bb] += (o[4] - o[1]) / 100.0;
      d[2][bb] += (o[5] - o[4]) / 100.0;
      d[3][bb] += (o[6] - o[5]) / 100.0;
      d[4][bb] += (o[2] - o[6]) / 100.0;
      d[5][bb] += (o[3] - o[2]) / 100.0;
    }
    for (int i = 0; i < nbins; ++i) {
      if (!begins[i] && !ends[i]) continue;
      const double n = begins[i] ? begins[i] : 1;
      std::printf("  t=%5.1f us  begin %5d  end %5d   mean us: desc %.2f or %.2f stage %.2f raw %.2f conv %.2f rows+stores %.2f\n",
                  i * (bin / 100.0), begins[i], ends[i], d[0][i] / n, d[1][i] / n, d[2][i] / n, d[3][i] / n, d[4][i] / n,
                  d[5][i] / n);
    }
  }
#endif
  CHECK_FCP(fcp_harness_destroy(h));
  CHECK_FCP(fcp_plan_destroy(plan));
  return 0;
}
