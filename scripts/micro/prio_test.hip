// development aid: does a kernel on a high-priority stream get wave slots while a long multi-round kernel
// on another stream still has workgroups pending? (decides whether speculative overlap can pay)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
// the same with 123 VGPRs allocated: four workgroups per CU then own every vector register of the CU
__global__ __launch_bounds__(256) void spin_fat(long long cycles, long long* out) {
  asm volatile("v_mov_b32 v122, 0" ::: "v122");
  const long long t0 = wall_clock64();
  long long t = t0;
  while (t - t0 < cycles) t = wall_clock64();
  if (out && threadIdx.x == 0) out[blockIdx.x] = t;
}
__global__ __launch_bounds__(256) void spin(long long cycles, long long* out) {
  extern __shared__ char lds[];
  const long long t0 = wall_clock64();
  long long t = t0;
  while (t - t0 < cycles) t = wall_clock64();
  if (out && threadIdx.x == 0) out[blockIdx.x] = t;
}
int main() {
  int lo, hi; CK(hipDeviceGetStreamPriorityRange(&lo, &hi)); printf("priority range: least %d greatest %d\n", lo, hi);
  hipStream_t a, bn, bh;
  CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
  CK(hipStreamCreateWithPriority(&bn, hipStreamNonBlocking, lo));
  CK(hipStreamCreateWithPriority(&bh, hipStreamNonBlocking, hi));
  hipEvent_t e0, eA, eB, go; CK(hipEventCreate(&e0)); CK(hipEventCreate(&eA)); CK(hipEventCreate(&eB)); CK(hipEventCreateWithFlags(&go, hipEventDisableTiming));
  // wall_clock64 ticks at 100 MHz: 20 us = 2000 ticks
  const long long big = 2400, small = 1000;      // 24 us per big workgroup, 10 us small
  const size_t lds_big = 36 * 1024;              // 4 workgroups per CU, like the log-likelihood kernel
  for (int which = 0; which < 3; ++which) {
    hipStream_t b = which == 0 ? a : (which == 1 ? bn : bh);
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0, a));
      hipLaunchKernelGGL(spin, dim3(4), dim3(256), 0, a, 100, nullptr);          // stands for the close kernel
      CK(hipEventRecord(go, a));
      hipLaunchKernelGGL(spin_fat, dim3(2500), dim3(256), 0, a, big, nullptr);  // the long kernel (2.44 rounds)
      CK(hipEventRecord(eA, a));
      if (b != a) CK(hipStreamWaitEvent(b, go, 0));
      hipLaunchKernelGGL(spin, dim3(4), dim3(256), 0, b, small, nullptr);         // the step kernel
      hipLaunchKernelGGL(spin, dim3(628), dim3(256), 0, b, 300, nullptr);         // the update kernel (3 us)
      CK(hipEventRecord(eB, b));
      CK(hipDeviceSynchronize());
      float ta, tb; CK(hipEventElapsedTime(&ta, e0, eA)); CK(hipEventElapsedTime(&tb, e0, eB));
      printf("%s: long kernel done at %.1f us, small kernels done at %.1f us\n", which == 0 ? "same stream     " : (which == 1 ? "low-prio stream " : "high-prio stream"), 1e3 * ta, 1e3 * tb);
    }
  }
  return 0;
}
