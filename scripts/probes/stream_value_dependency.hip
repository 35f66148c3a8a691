#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#define CK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { printf("%s -> %s\n", #e, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void spin(unsigned long long ticks) { const unsigned long long t0 = __builtin_amdgcn_s_memrealtime(); while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(4); }
__global__ void tiny() {}
int main() {
  hipStream_t caller, lane[3];
  CK(hipStreamCreateWithFlags(&caller, hipStreamNonBlocking));
  for (auto &l : lane) CK(hipStreamCreateWithFlags(&l, hipStreamNonBlocking));
  hipEvent_t in[8], out[24];
  for (auto &e : in) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming | hipEventDisableSystemFence));
  for (auto &e : out) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming | hipEventDisableSystemFence));
  uint32_t *sig = nullptr;
  CK(hipExtMallocWithFlags((void **)&sig, 8, hipMallocSignalMemory));
  CK(hipMemset(sig, 0, 8));
  auto now = [] { return std::chrono::steady_clock::now(); };
  const int N = 3000;
  for (int mode = 0; mode < 2; ++mode) {
    double t_dep = 0;
    CK(hipDeviceSynchronize());
    auto t0 = now();
    for (int i = 0; i < N; ++i) {
      const int l = i % 3;
      auto a = now();
      if (mode == 0) {
        CK(hipEventRecord(in[i % 8], caller));
        CK(hipStreamWaitEvent(lane[l], in[i % 8], 0));
      } else {
        CK(hipStreamWriteValue32(caller, sig, (uint32_t)(i + 1), 0));
        CK(hipStreamWaitValue32(lane[l], sig, (uint32_t)(i + 1), hipStreamWaitValueGte, 0xffffffffu));
      }
      t_dep += std::chrono::duration<double, std::micro>(now() - a).count();
      hipExtLaunchKernelGGL(spin, dim3(1), dim3(64), 0, lane[l], nullptr, out[i % 24], 0, 2000ull);
      if (i >= 2) { CK(hipStreamWaitEvent(caller, out[(i - 2) % 24], 0)); hipLaunchKernelGGL(tiny, dim3(1), dim3(1), 0, caller); }
    }
    CK(hipDeviceSynchronize());
    const double us = std::chrono::duration<double, std::micro>(now() - t0).count();
    printf("%s: %.2f us per request wall, dependency calls %.2f us of host time per request\n", mode == 0 ? "event record + stream wait" : "write value + wait value ", us / N, t_dep / N);
  }
  return 0;
}
