// gather_policy_probe.hip — round 6: does ANY cache policy of a vector load fetch less than a 128-byte line for a 32- or
// 64-byte table row?  S2's PMC record (profiles/r05_s2_pmc_fcp_bench.txt) shows one fabric read request per 32 / 64 / 128-byte
// row and two per 256-byte row, and the gather probe of fcp_harness reads 32-, 64- and 128-byte rows at the SAME row rate
// (45-48 G rows/s): rows of dims 8 / 16 pay for a whole line.  HBM3's own access granularity is 32 bytes; if some policy
// (nt / sc0 / sc1 bits, or memory allocated uncached) made the L2 ask for 32 or 64 bytes only, a quarter of S2's read
// traffic would disappear.  Random row gathers from an 8-GiB buffer, nothing written, `depth` rows in flight per lane.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/probes/gather_policy_probe.hip -o build/gather_policy_probe
//   ./build/gather_policy_probe [GiB]            (prints useful TB/s and G rows/s per (allocation, policy, row size))
//   rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --kernel-trace -d <dir> -- ./build/gather_policy_probe 8 1
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHECK(x)                                                                                   \
  do {                                                                                             \
    hipError_t e_ = (x);                                                                           \
    if (e_ != hipSuccess) {                                                                        \
      std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                                 \
      std::exit(1);                                                                                \
    }                                                                                              \
  } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) const f4 gf4;

// Four independent 16-byte loads and the wait for them in ONE asm statement: the compiler knows nothing about loads
// issued from inline asm — given one statement per load it used the "results" before they had landed and recycled their
// registers as ADDRESSES of later loads, which a late data return then overwrote (memory access fault).
#define LD4(POLICY)                                                                                                  \
  asm volatile("global_load_dwordx4 %0, %4, off" POLICY "\n\tglobal_load_dwordx4 %1, %5, off" POLICY                   \
               "\n\tglobal_load_dwordx4 %2, %6, off" POLICY "\n\tglobal_load_dwordx4 %3, %7, off" POLICY               \
               "\n\ts_waitcnt vmcnt(0)"                                                                              \
               : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3])                                                   \
               : "v"(g[0]), "v"(g[1]), "v"(g[2]), "v"(g[3])                                                           \
               : "memory")
template <int POL> __device__ __forceinline__ void ld4x16(f4 (&v)[4], gf4 *const (&g)[4]) {
  if (POL == 0) LD4("");
  if (POL == 1) LD4(" nt");
  if (POL == 2) LD4(" sc0");
  if (POL == 3) LD4(" sc1");
  if (POL == 4) LD4(" sc0 sc1");
  if (POL == 5) LD4(" sc0 sc1 nt");
  if (POL == 6) LD4(" sc1 nt");
}
static const char *kPol[7] = {"default", "nt", "sc0", "sc1", "sc0 sc1", "sc0 sc1 nt", "sc1 nt"};

template <int POL>
__global__ void __launch_bounds__(256) gather_kernel(const f4 *__restrict__ src, float *sink, unsigned long long n_rows,
                                                     int lanes_per_row, int rounds) {
  const unsigned long long gid = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned long long grp = gid / lanes_per_row;
  const int part = (int)(gid % lanes_per_row);
  f4 acc = {0, 0, 0, 0};
  unsigned long long x = grp * 0x9E3779B97F4A7C15ull + 12345;
  for (int r = 0; r < rounds; ++r) {
    f4 v[4];
    gf4 *g[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      x ^= x >> 29;
      x *= 0xBF58476D1CE4E5B9ull;
      x ^= x >> 32;
      const unsigned long long row = x % n_rows;
      g[k] = (gf4 *)(src + row * lanes_per_row + part);
    }
    ld4x16<POL>(v, g);
#pragma unroll
    for (int k = 0; k < 4; ++k) acc += v[k];
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) *sink = acc.x;
}

template <int POL> static void run(const char *alloc, const f4 *buf, float *sink, size_t bytes, int row_bytes, int iters) {
  const int lanes_per_row = row_bytes / 16, rounds = 4, blocks = 256 * 8 * 4;
  const unsigned long long n_rows = bytes / row_bytes;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(gather_kernel<POL>, dim3(blocks), dim3(256), 0, 0, buf, sink, n_rows, lanes_per_row, rounds);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0, 0));
  for (int i = 0; i < iters; ++i)
    hipLaunchKernelGGL(gather_kernel<POL>, dim3(blocks), dim3(256), 0, 0, buf, sink, n_rows, lanes_per_row, rounds);
  CHECK(hipEventRecord(e1, 0));
  CHECK(hipDeviceSynchronize());
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double useful = (double)blocks * 256 * 16.0 * 4 * rounds, rows = useful / row_bytes;
  std::printf("%-9s %-11s %3d-byte rows: %6.2f TB/s useful, %5.1f G rows/s, %7.2f us per launch\n", alloc, kPol[POL], row_bytes,
              useful / (ms / iters * 1e-3) / 1e12, rows / (ms / iters * 1e-3) / 1e9, ms / iters * 1e3);
  CHECK(hipEventDestroy(e0));
  CHECK(hipEventDestroy(e1));
}

int main(int argc, char **argv) {
  const size_t gib = argc > 1 ? (size_t)std::atol(argv[1]) : 8;
  const int iters = argc > 2 ? std::atoi(argv[2]) : 10;
  const size_t bytes = gib << 30;
  float *sink = nullptr;
  CHECK(hipMalloc(&sink, 4));
  for (int alloc = 0; alloc < 2; ++alloc) {
    void *buf = nullptr;
    if (alloc == 0) CHECK(hipMalloc(&buf, bytes));
    else if (hipExtMallocWithFlags(&buf, bytes, hipDeviceMallocUncached) != hipSuccess) {
      std::printf("hipExtMallocWithFlags(hipDeviceMallocUncached) failed: skipped\n");
      break;
    }
    CHECK(hipMemset(buf, 1, bytes));
    CHECK(hipDeviceSynchronize());
    const char *an = alloc == 0 ? "hipMalloc" : "uncached";
    for (int rb : {32, 64, 128}) {
      run<0>(an, (const f4 *)buf, sink, bytes, rb, iters);
      run<1>(an, (const f4 *)buf, sink, bytes, rb, iters);
      run<2>(an, (const f4 *)buf, sink, bytes, rb, iters);
      run<3>(an, (const f4 *)buf, sink, bytes, rb, iters);
      run<4>(an, (const f4 *)buf, sink, bytes, rb, iters);
      run<5>(an, (const f4 *)buf, sink, bytes, rb, iters);
      run<6>(an, (const f4 *)buf, sink, bytes, rb, iters);
    }
    CHECK(hipFree(buf));
  }
  return 0;
}
