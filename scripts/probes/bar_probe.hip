// Feasibility probe: can the host write fine-grained device memory directly (large BAR)?
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
#define CK(e) do { hipError_t r = (e); if (r != hipSuccess) { printf("%s -> %s\n", #e, hipGetErrorString(r)); return 1; } } while (0)
__global__ void check(const unsigned *p, int n, unsigned seed, int *bad) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
    if (p[i] != seed + i) atomicAdd(bad, 1);
}
int main() {
  const int n = 6144; // 24 KiB
  unsigned *d = nullptr;
  int *bad;
  CK(hipMalloc(&bad, 4));
  hipError_t e = hipExtMallocWithFlags((void **)&d, n * 4, hipDeviceMallocFinegrained);
  printf("hipExtMallocWithFlags(finegrained): %s ptr=%p\n", hipGetErrorString(e), (void *)d);
  if (e != hipSuccess) return 1;
  hipPointerAttribute_t at;
  if (hipPointerGetAttributes(&at, d) == hipSuccess) printf("type=%d isManaged=%d hostPtr=%p devPtr=%p\n", (int)at.type, at.isManaged, at.hostPointer, at.devicePointer);
  std::vector<unsigned> h(n);
  double best = 1e9;
  int total_bad = 0;
  for (int it = 0; it < 20; ++it) {
    for (int i = 0; i < n; ++i) h[i] = 1000u * it + i;
    auto t0 = std::chrono::steady_clock::now();
    memcpy(d, h.data(), n * 4); // host CPU stores straight into VRAM
    __builtin_ia32_sfence();
    auto t1 = std::chrono::steady_clock::now();
    best = std::min(best, std::chrono::duration<double, std::micro>(t1 - t0).count());
    CK(hipMemset(bad, 0, 4));
    hipLaunchKernelGGL(check, dim3(4), dim3(256), 0, 0, d, n, 1000u * it, bad);
    int hb = -1;
    CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
    total_bad += hb;
  }
  printf("host write 24 KiB into VRAM: best %.2f us; mismatches over 20 rounds: %d\n", best, total_bad);
  return 0;
}
