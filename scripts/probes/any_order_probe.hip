// any_order_probe.hip — what does a launch WITHOUT the queue's barrier bit (hipExtAnyOrderLaunch) buy behind an ordinary
// launch on the same stream, and what do same-address agent-scope atomics cost?  (Round 5: the gated segment-offset pre-pass.)
//   hipcc --offload-arch=gfx950 -O2 scripts/probes/any_order_probe.hip -o /tmp/any_order_probe && /tmp/any_order_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void spin_kernel(unsigned long long ticks, unsigned long long *t) { // A: every block spins for `ticks` (100 MHz)
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0 && blockIdx.x == 0) t[0] = t0;
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(4);
  if (threadIdx.x == 0 && blockIdx.x == 0) t[1] = __builtin_amdgcn_s_memrealtime();
}
// A with a TAIL: block i spins between ticks / 2 (i = 0) and ticks (the last block): wave slots free up from ticks / 2 on
__global__ void tail_kernel(unsigned long long ticks, unsigned long long *t) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0 && blockIdx.x == 0) t[0] = t0;
  const unsigned long long mine = ticks / 2 + ticks / 2 * blockIdx.x / gridDim.x;
  while (__builtin_amdgcn_s_memrealtime() - t0 < mine) __builtin_amdgcn_s_sleep(4);
  if (threadIdx.x == 0 && blockIdx.x == gridDim.x - 1) t[1] = __builtin_amdgcn_s_memrealtime();
}
__global__ void stamp_kernel(unsigned long long *t) { // B: when did my first block start / my last block end?
  const unsigned long long now = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) atomicMin(t + 2, now);
  if (threadIdx.x == 0) atomicMax(t + 3, __builtin_amdgcn_s_memrealtime());
}
__global__ void atomic_kernel(unsigned *ctr, int scope_agent, int returning, unsigned *sink) {
  if (threadIdx.x == 0) {
    if (returning) {
      const unsigned v = scope_agent ? __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : atomicAdd(ctr, 1u);
      if (v == 0xffffffffu) *sink = v;
    } else {
      if (scope_agent) (void)__hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else (void)atomicAdd(ctr, 1u);
    }
  }
}

int main() {
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  unsigned long long *t;
  CK(hipHostMalloc(&t, 64, hipHostMallocMapped));
  unsigned long long *dt;
  CK(hipHostGetDevicePointer((void **)&dt, t, 0));
  const unsigned long long ticks = 2000; // 20 us
  for (int variant = 0; variant < 4; ++variant) {
    const bool a_any = variant == 3, b_any = variant >= 2, a_small = variant == 1;
    double acc_start = 0, acc_end = 0;
    const int reps = 50;
    for (int r = 0; r < reps + 5; ++r) {
      t[0] = t[1] = 0; t[2] = ~0ull; t[3] = 0;
      hipExtLaunchKernelGGL(spin_kernel, dim3(a_small ? 256 : 1024), dim3(256), 0, s, nullptr, nullptr, a_any ? (int)hipExtAnyOrderLaunch : 0, ticks, dt);
      hipExtLaunchKernelGGL(stamp_kernel, dim3(2048), dim3(256), 0, s, nullptr, nullptr, b_any ? (int)hipExtAnyOrderLaunch : 0, dt);
      CK(hipStreamSynchronize(s));
      if (r >= 5) { acc_start += ((double)t[2] - (double)t[0]) / 100.0; acc_end += ((double)t[3] - (double)t[1]) / 100.0; }
    }
    printf("A %s (20 us spin, %d blocks) -> B %s (2048 blocks): B's first block starts %.2f us after A's start; B's last block ends %.2f us after A's end\n",
           a_any ? "any-order" : "in order", a_small ? 256 : 1024, b_any ? "any-order" : "in order", acc_start / reps, acc_end / reps);
  }
  // A with a tail (2048 blocks = every wave slot of the GPU taken; they retire between 10 and 20 us): does B start into the tail?
  for (int b_any = 0; b_any < 2; ++b_any) {
    double acc_start = 0, acc_end = 0;
    const int reps = 50;
    for (int r = 0; r < reps + 5; ++r) {
      t[0] = t[1] = 0; t[2] = ~0ull; t[3] = 0;
      hipExtLaunchKernelGGL(tail_kernel, dim3(2048), dim3(256), 0, s, nullptr, nullptr, 0, ticks, dt);
      hipExtLaunchKernelGGL(stamp_kernel, dim3(2048), dim3(256), 0, s, nullptr, nullptr, b_any ? (int)hipExtAnyOrderLaunch : 0, dt);
      CK(hipStreamSynchronize(s));
      if (r >= 5) { acc_start += ((double)t[2] - (double)t[0]) / 100.0; acc_end += ((double)t[3] - (double)t[1]) / 100.0; }
    }
    printf("A in order with a TAIL (2048 blocks retire between 10 and 20 us) -> B %s: B's first block starts %.2f us after A's start; B's last block ends %.2f us after A's last block\n",
           b_any ? "any-order" : "in order", acc_start / reps, acc_end / reps);
  }
  // same-address atomics: 1024 blocks, one atomic each
  unsigned *ctr, *sink;
  CK(hipMalloc(&ctr, 256)); CK(hipMalloc(&sink, 4)); CK(hipMemset(ctr, 0, 256));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int blocks : {1024, 8192}) for (int agent = 0; agent < 2; ++agent) for (int ret = 0; ret < 2; ++ret) {
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(atomic_kernel, dim3(blocks), dim3(256), 0, s, ctr, agent, ret, sink);
    CK(hipEventRecord(e0, s));
    for (int w = 0; w < 20; ++w) hipLaunchKernelGGL(atomic_kernel, dim3(blocks), dim3(256), 0, s, ctr, agent, ret, sink);
    CK(hipEventRecord(e1, s));
    CK(hipStreamSynchronize(s));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%d blocks x one %s atomic add (%s) on ONE address: %.2f us per launch\n", blocks, agent ? "agent-scope" : "default (atomicAdd)", ret ? "returning" : "no return", ms * 1e3 / 20);
  }
  hipLaunchKernelGGL(stamp_kernel, dim3(1024), dim3(256), 0, s, dt);
  CK(hipEventRecord(e0, s));
  for (int w = 0; w < 20; ++w) hipLaunchKernelGGL(stamp_kernel, dim3(1024), dim3(256), 0, s, dt);
  CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
  float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("(reference: 1024 blocks, atomics to host memory only: %.2f us per launch)\n", ms * 1e3 / 20);
  return 0;
}
