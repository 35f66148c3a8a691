// Which plain-copy shape reaches the highest read+write rate on this MI355X?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(e) do { hipError_t r = (e); if (r != hipSuccess) { printf("%s -> %s\n", #e, hipGetErrorString(r)); return 1; } } while (0)
typedef float __attribute__((ext_vector_type(4))) f4;

template <bool NT_LD, bool NT_ST, int UNROLL>
__global__ void __launch_bounds__(256) copy_k(const f4 *__restrict__ s, f4 *__restrict__ d, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + (UNROLL - 1) * stride < n; i += UNROLL * stride) {
    f4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) v[u] = NT_LD ? __builtin_nontemporal_load(s + i + u * stride) : s[i + u * stride];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) { if (NT_ST) __builtin_nontemporal_store(v[u], d + i + u * stride); else d[i + u * stride] = v[u]; }
  }
  for (; i < n; i += stride) d[i] = s[i];
}

// one float4 per thread, no loop
__global__ void __launch_bounds__(256) copy_flat(const f4 *__restrict__ s, f4 *__restrict__ d, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) d[i] = s[i];
}
// contiguous chunk per block (block copies 16 KiB contiguous), like the guide's streaming kernels
template <bool NT>
__global__ void __launch_bounds__(256) copy_chunk(const f4 *__restrict__ s, f4 *__restrict__ d, size_t n) {
  const size_t base = (size_t)blockIdx.x * 1024; // 1024 float4 = 16 KiB per block
  f4 v[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) { size_t i = base + u * 256 + threadIdx.x; v[u] = i < n ? (NT ? __builtin_nontemporal_load(s + i) : s[i]) : f4{0,0,0,0}; }
#pragma unroll
  for (int u = 0; u < 4; ++u) { size_t i = base + u * 256 + threadIdx.x; if (i < n) { if (NT) __builtin_nontemporal_store(v[u], d + i); else d[i] = v[u]; } }
}

int main(int argc, char **argv) {
  const size_t bytes = (size_t)(argc > 1 ? atol(argv[1]) : 2048) << 20;
  const size_t n = bytes / 16;
  f4 *a, *b;
  CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
  CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 2, bytes));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto run = [&](const char *name, auto launch) {
    launch(); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < 10; ++i) launch();
    CK(hipEventRecord(e1, 0)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-44s %.0f GB/s (read+write)\n", name, 2.0 * bytes / (ms / 10 * 1e-3) / 1e9);
    return 0;
  };
  for (int g : {1024, 2048, 4096, 8192, 16384}) {
    char nm[64]; snprintf(nm, 64, "grid-stride unroll4 default, %d blocks", g);
    run(nm, [&] { hipLaunchKernelGGL((copy_k<false, false, 4>), dim3(g), dim3(256), 0, 0, a, b, n); });
  }
  run("grid-stride unroll4 nt-ld nt-st, 2048 blocks", [&] { hipLaunchKernelGGL((copy_k<true, true, 4>), dim3(2048), dim3(256), 0, 0, a, b, n); });
  run("grid-stride unroll4 nt-st only, 2048 blocks", [&] { hipLaunchKernelGGL((copy_k<false, true, 4>), dim3(2048), dim3(256), 0, 0, a, b, n); });
  run("grid-stride unroll8 default, 2048 blocks", [&] { hipLaunchKernelGGL((copy_k<false, false, 8>), dim3(2048), dim3(256), 0, 0, a, b, n); });
  run("grid-stride unroll1 default, 4096 blocks", [&] { hipLaunchKernelGGL((copy_k<false, false, 1>), dim3(4096), dim3(256), 0, 0, a, b, n); });
  run("flat one float4 per thread", [&] { hipLaunchKernelGGL(copy_flat, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, a, b, n); });
  run("16 KiB chunk per block default", [&] { hipLaunchKernelGGL((copy_chunk<false>), dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, 0, a, b, n); });
  run("16 KiB chunk per block nt", [&] { hipLaunchKernelGGL((copy_chunk<true>), dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, 0, a, b, n); });
  run("hipMemcpyDtoD", [&] { (void)hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0); });
  return 0;
}
