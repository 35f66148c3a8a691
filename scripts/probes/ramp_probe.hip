// ramp_probe — what does the FRONT of a launch cost on this box?  (round 2, profiles/HISTORY.md section 4b)
//
// A grid shaped like the S2 dense launch (3776 blocks x 256 threads, 640-byte by-value argument,
// 8 blocks per CU resident) in which every block runs a chain of dependent loads and stamps
// s_memrealtime (100 MHz) between the hops:
//   t0 block start | t1 first kernel-argument dword usable | t2 load 1 (address from the argument)
//   | t3 load 2 (address from load 1) | t4 load 3 (address from load 2) | t5 16 random 1-KiB row reads
// Launched back to back on one stream like the bench; the stamps of the LAST launch are reported for the
// first generation of blocks (begin within 2 us of the first block) and for the rest.
// Run with HIP_FORCE_DEV_KERNARG=0 / 1 to see where the kernel arguments live.
//
//   hipcc --offload-arch=gfx950 -O2 scripts/probes/ramp_probe.hip -o build/ramp_probe
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(e)                                                                                   \
  do {                                                                                             \
    hipError_t e_ = (e);                                                                           \
    if (e_ != hipSuccess) {                                                                        \
      std::fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #e, hipGetErrorString(e_));        \
      std::exit(1);                                                                                \
    }                                                                                              \
  } while (0)

struct Arg {
  const uint32_t *a; // a[i] -> index into b
  const uint32_t *b; // b[i] -> index into c
  const uint32_t *c; // c[i] -> row index
  const float4 *rows; // big buffer of 1-KiB rows
  unsigned long long *stamps;
  float4 *sink;
  uint32_t n, nrows;
  char pad[640 - 56];
};

__global__ void __launch_bounds__(256) chain_kernel(const Arg A) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  const uint32_t n = A.n; // first use of the argument block
  asm volatile("s_waitcnt lgkmcnt(0)" ::"s"(n) : "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
  const uint32_t i = (blockIdx.x * 977u + threadIdx.x / 64) % n;
  const uint32_t x = A.a[i];
  asm volatile("s_waitcnt vmcnt(0)" ::"v"(x) : "memory");
  const unsigned long long t2 = __builtin_amdgcn_s_memrealtime();
  const uint32_t y = A.b[x];
  asm volatile("s_waitcnt vmcnt(0)" ::"v"(y) : "memory");
  const unsigned long long t3 = __builtin_amdgcn_s_memrealtime();
  const uint32_t z = A.c[y];
  asm volatile("s_waitcnt vmcnt(0)" ::"v"(z) : "memory");
  const unsigned long long t4 = __builtin_amdgcn_s_memrealtime();
  float4 acc = make_float4(0, 0, 0, 0);
  const uint32_t lane = threadIdx.x & 63;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const uint32_t row = (z * 2654435761u + r * 40503u + blockIdx.x * 7919u + (threadIdx.x >> 6) * 104729u) % A.nrows;
    const float4 v = A.rows[(size_t)row * 64 + lane];
    acc.x += v.x;
    acc.y += v.y;
    acc.z += v.z;
    acc.w += v.w;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::"v"(acc.x) : "memory");
  const unsigned long long t5 = __builtin_amdgcn_s_memrealtime();
  if (acc.x == 1234.5f) A.sink[threadIdx.x] = acc;
  if (threadIdx.x == 0) {
    unsigned long long *o = A.stamps + 8ull * blockIdx.x;
    o[0] = t0;
    o[1] = t1;
    o[2] = t2;
    o[3] = t3;
    o[4] = t4;
    o[5] = t5;
  }
}

__global__ void __launch_bounds__(256) empty_kernel(const Arg A) {
  if (A.n == 0xdeadbeefu) A.sink[0] = make_float4(0, 0, 0, 0);
}

int main(int argc, char **argv) {
  const int blocks = argc > 1 ? std::atoi(argv[1]) : 3776;
  const int launches = argc > 2 ? std::atoi(argv[2]) : 300;
  const uint32_t n = 1u << 20;
  const uint32_t nrows = 8u << 20; // 8 GiB of 1-KiB rows
  std::vector<uint32_t> h(n);
  uint32_t *a, *b, *c;
  float4 *rows, *sink;
  unsigned long long *stamps;
  CHECK(hipMalloc(&a, n * 4));
  CHECK(hipMalloc(&b, n * 4));
  CHECK(hipMalloc(&c, n * 4));
  CHECK(hipMalloc(&rows, (size_t)nrows * 1024));
  CHECK(hipMalloc(&sink, 4096));
  CHECK(hipMalloc(&stamps, 8ull * 8 * blocks));
  CHECK(hipMemset(rows, 0, (size_t)nrows * 1024));
  uint64_t s = 88172645463325252ull;
  for (uint32_t *p : {a, b, c}) {
    for (auto &v : h) {
      s ^= s << 13;
      s ^= s >> 7;
      s ^= s << 17;
      v = (uint32_t)(s % n);
    }
    CHECK(hipMemcpy(p, h.data(), n * 4, hipMemcpyHostToDevice));
  }
  Arg A;
  std::memset(&A, 0, sizeof(A));
  A.a = a;
  A.b = b;
  A.c = c;
  A.rows = rows;
  A.stamps = stamps;
  A.sink = sink;
  A.n = n;
  A.nrows = nrows;
  hipStream_t st;
  CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  for (int kind = 0; kind < 2; ++kind) {
    for (int i = 0; i < 50; ++i) {
      if (kind == 0) hipLaunchKernelGGL(empty_kernel, dim3(blocks), dim3(256), 0, st, A);
      else hipLaunchKernelGGL(chain_kernel, dim3(blocks), dim3(256), 0, st, A);
    }
    CHECK(hipStreamSynchronize(st));
    CHECK(hipEventRecord(e0, st));
    for (int i = 0; i < launches; ++i) {
      if (kind == 0) hipLaunchKernelGGL(empty_kernel, dim3(blocks), dim3(256), 0, st, A);
      else hipLaunchKernelGGL(chain_kernel, dim3(blocks), dim3(256), 0, st, A);
    }
    CHECK(hipEventRecord(e1, st));
    CHECK(hipStreamSynchronize(st));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::printf("%s kernel, %d blocks x 256, 640-byte argument: %.2f us per back-to-back launch\n",
                kind == 0 ? "empty" : "chain", blocks, ms * 1e3 / launches);
  }
  std::vector<unsigned long long> hs(8ull * blocks);
  CHECK(hipMemcpy(hs.data(), stamps, hs.size() * 8, hipMemcpyDeviceToHost));
  unsigned long long tmin = ~0ull, tmax = 0;
  for (int i = 0; i < blocks; ++i) {
    tmin = std::min(tmin, hs[8 * i]);
    tmax = std::max(tmax, hs[8 * i + 5]);
  }
  std::printf("last chain launch: first block start -> last block end %.2f us\n", (tmax - tmin) / 100.0);
  const char *names[5] = {"arg", "load1", "load2", "load3", "4 rows"};
  for (int gen = 0; gen < 2; ++gen) {
    double d[5] = {0, 0, 0, 0, 0}, start = 0;
    int cnt = 0;
    for (int i = 0; i < blocks; ++i) {
      const unsigned long long *o = &hs[8 * i];
      const bool first = o[0] - tmin < 200;
      if (first != (gen == 0)) continue;
      ++cnt;
      start += (o[0] - tmin) / 100.0;
      for (int k = 0; k < 5; ++k) d[k] += (o[k + 1] - o[k]) / 100.0;
    }
    if (!cnt) continue;
    std::printf("  %s: %5d blocks, mean start %.2f us; mean us per hop:", gen == 0 ? "first generation" : "later blocks   ", cnt,
                start / cnt);
    for (int k = 0; k < 5; ++k) std::printf("  %s %.2f", names[k], d[k] / cnt);
    std::printf("\n");
  }
  return 0;
}
