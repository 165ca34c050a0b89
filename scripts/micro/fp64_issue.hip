// development aid: issue rate of dependent and independent v_fma_f64 / v_mul_f64 / v_rcp_f64 chains on gfx950 as a
// function of the wavefronts per SIMD (1, 2, 4, 8) and of the independent chains per wavefront (1, 2, 4).
// Prints cycles per instruction per SIMD (4.0 = the fp64 VALU is saturated).
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int ILP, int OP>
__global__ __launch_bounds__(256) void chain(int iters, double* out, long long* cyc) {
  double x[4];
  const double a = 1.0 + 1e-9 * threadIdx.x, b = 1e-12;
  for (int i = 0; i < 4; ++i) x[i] = 1.0 + 1e-3 * i + 1e-6 * threadIdx.x;
  const long long t0 = wall_clock64(), c0 = clock64();
  for (int k = 0; k < iters; ++k) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
#pragma unroll
      for (int i = 0; i < ILP; ++i) {
        if (OP == 0) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
        if (OP == 1) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x[i]) : "v"(a));
        if (OP == 2) asm volatile("v_rcp_f64_e32 %0, %0\n\ts_nop 0" : "+v"(x[i]));
        if (OP == 3) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x[i]) : "v"(b));
        if (OP == 4) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(*(float*)&x[i]) : "v"((float)a), "v"((float)b));
      }
    }
  }
  const long long t1 = wall_clock64(), c1 = clock64();
  double s = 0;
  for (int i = 0; i < 4; ++i) s += x[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) { cyc[2 * blockIdx.x] = t1 - t0; cyc[2 * blockIdx.x + 1] = c1 - c0; }
}

template <int ILP, int OP>
int run(const char* name, int wgs_per_cu, double* d_out, long long* d_cyc) {
  const int iters = 2000, blocks = 256 * wgs_per_cu;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((chain<ILP, OP>), dim3(blocks), dim3(256), 0, 0, iters, d_out, d_cyc);
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  // instructions per SIMD: wgs_per_cu waves per SIMD x iters x 16 x ILP
  const double inst = (double)wgs_per_cu * iters * 16 * ILP;
  long long h[2];
  CK(hipMemcpy(h, d_cyc, sizeof h, hipMemcpyDeviceToHost));
  const double ghz = (double)h[1] / (double)h[0] * 0.1;      // wall_clock64 ticks at 100 MHz
  printf("%-10s waves/SIMD %d  chains/wave %d : %.2f cycles per instruction per SIMD at 2.4 GHz (%.3f ms); block 0: %lld shader clocks in %lld x 10 ns = %.2f GHz -> %.2f shader cycles per instruction\n", name, wgs_per_cu, ILP,
         best * 1e-3 * 2.4e9 / inst, best, h[1], h[0], ghz, (double)h[1] / ((double)iters * 16 * ILP) / 1.0);
  return 0;
}

int main() {
  double* d_out; long long* d_cyc;
  CK(hipMalloc(&d_out, sizeof(double) * 256 * 8 * 256)); CK(hipMalloc(&d_cyc, sizeof(long long) * 256 * 8 * 2));
  for (int w : {1, 2, 4, 8}) {
    run<1, 0>("fma_f64", w, d_out, d_cyc); run<2, 0>("fma_f64", w, d_out, d_cyc); run<4, 0>("fma_f64", w, d_out, d_cyc);
  }
  for (int w : {1, 4}) { run<1, 1>("mul_f64", w, d_out, d_cyc); run<2, 1>("mul_f64", w, d_out, d_cyc); }
  for (int w : {1, 4}) { run<1, 3>("add_f64", w, d_out, d_cyc); run<2, 3>("add_f64", w, d_out, d_cyc); }
  for (int w : {1, 4}) { run<1, 2>("rcp_f64", w, d_out, d_cyc); run<2, 2>("rcp_f64", w, d_out, d_cyc); run<4, 2>("rcp_f64", w, d_out, d_cyc); }
  for (int w : {1, 4}) { run<1, 4>("fma_f32", w, d_out, d_cyc); run<2, 4>("fma_f32", w, d_out, d_cyc); }
  return 0;
}
