// development aid: what a random gather from a small LDS table costs per wave-instruction on gfx950, by instruction form --
// the log table of the log-likelihood kernel's cell (two 8-byte entries per cell at a random index):
//   0: ds_read2st64_b64 (two arrays, one instruction: what hipcc makes of tab[j], tab[N + j])
//   1: two ds_read_b64
//   2: one ds_read_b128 of an interleaved {1/c, log c} table
//   3: one ds_read_b64 (half the data: a lower bound)
// 16 wavefronts per CU (four workgroups of 256), every lane its own pseudo-random index, indices of 10 bits (1024 entries).
// build: hipcc -O3 --offload-arch=gfx950 -o lds_gather scripts/micro/lds_gather.hip ; run: ./lds_gather
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
template <int MODE>
__global__ __launch_bounds__(256, 4) void k(double* out, int iters, int entries_log2) {
  extern __shared__ double lds[];
  const int n = 1 << entries_log2;
  for (int i = threadIdx.x; i < 2 * n; i += 256) lds[i] = 1.0 + 1e-3 * i;
  __syncthreads();
  unsigned s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
  double acc0 = 0.0, acc1 = 0.0;
  const unsigned mask = (unsigned)(n - 1);
  for (int it = 0; it < iters; ++it) {
    unsigned a[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { s = s * 1664525u + 1013904223u; a[u] = (s >> 12) & mask; }
    double x[4], y[4];
    typedef double v2d __attribute__((ext_vector_type(2)));
    if (MODE == 0) {
      v2d r0, r1, r2, r3;
      if (entries_log2 == 10)
        asm volatile("ds_read2st64_b64 %0, %4 offset1:16\n\tds_read2st64_b64 %1, %5 offset1:16\n\tds_read2st64_b64 %2, %6 offset1:16\n\tds_read2st64_b64 %3, %7 offset1:16\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3) : "v"(a[0] << 3), "v"(a[1] << 3), "v"(a[2] << 3), "v"(a[3] << 3));
      else
        asm volatile("ds_read2st64_b64 %0, %4 offset1:4\n\tds_read2st64_b64 %1, %5 offset1:4\n\tds_read2st64_b64 %2, %6 offset1:4\n\tds_read2st64_b64 %3, %7 offset1:4\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3) : "v"(a[0] << 3), "v"(a[1] << 3), "v"(a[2] << 3), "v"(a[3] << 3));
      acc0 += r0.x + r1.x + r2.x + r3.x; acc1 += r0.y + r1.y + r2.y + r3.y;
    } else if (MODE == 1) {
      const unsigned o = 8u << entries_log2;
      asm volatile("ds_read_b64 %0, %8\n\tds_read_b64 %1, %9\n\tds_read_b64 %2, %10\n\tds_read_b64 %3, %11\n\tds_read_b64 %4, %12\n\tds_read_b64 %5, %13\n\tds_read_b64 %6, %14\n\tds_read_b64 %7, %15\n\ts_waitcnt lgkmcnt(0)"
                   : "=&v"(x[0]), "=&v"(y[0]), "=&v"(x[1]), "=&v"(y[1]), "=&v"(x[2]), "=&v"(y[2]), "=&v"(x[3]), "=&v"(y[3])
                   : "v"(a[0] << 3), "v"((a[0] << 3) + o), "v"(a[1] << 3), "v"((a[1] << 3) + o), "v"(a[2] << 3), "v"((a[2] << 3) + o), "v"(a[3] << 3), "v"((a[3] << 3) + o));
#pragma unroll
      for (int u = 0; u < 4; ++u) { acc0 += x[u]; acc1 += y[u]; }
    } else if (MODE == 2) {
      v2d r0, r1, r2, r3;
      asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %5\n\tds_read_b128 %2, %6\n\tds_read_b128 %3, %7\n\ts_waitcnt lgkmcnt(0)"
                   : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3) : "v"(a[0] << 4), "v"(a[1] << 4), "v"(a[2] << 4), "v"(a[3] << 4));
      acc0 += r0.x + r1.x + r2.x + r3.x; acc1 += r0.y + r1.y + r2.y + r3.y;
    } else {
      asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %5\n\tds_read_b64 %2, %6\n\tds_read_b64 %3, %7\n\ts_waitcnt lgkmcnt(0)"
                   : "=&v"(x[0]), "=&v"(x[1]), "=&v"(x[2]), "=&v"(x[3]) : "v"(a[0] << 3), "v"(a[1] << 3), "v"(a[2] << 3), "v"(a[3] << 3));
#pragma unroll
      for (int u = 0; u < 4; ++u) acc0 += x[u];
    }
  }
  out[(long)blockIdx.x * 256 + threadIdx.x] = acc0 + acc1;
}
template <int MODE> static double run(int iters, int elog2) {
  double* out; hipMalloc(&out, sizeof(double) * 1024 * 256);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const size_t lds = sizeof(double) * 2 * (1 << elog2);
  k<MODE><<<1024, 256, lds>>>(out, 10, elog2);
  hipEventRecord(e0);
  k<MODE><<<1024, 256, lds>>>(out, iters, elog2);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  hipFree(out);
  return ms;
}
int main() {
  const int iters = 4000;
  for (int elog2 : {8, 10}) {
    const double t[4] = {run<0>(iters, elog2), run<1>(iters, elog2), run<2>(iters, elog2), run<3>(iters, elog2)};
    const char* nm[4] = {"ds_read2st64_b64", "2 x ds_read_b64", "ds_read_b128 (interleaved)", "1 x ds_read_b64"};
    // per CU: 16 waves x iters x 4 gathers; ns per wave-gather per CU
    for (int m = 0; m < 4; ++m) printf("entries %4d  %-28s %8.3f ms  = %6.2f ns per wave-gather per CU (x 2.2 GHz = %5.1f cycles)\n", 1 << elog2, nm[m], t[m], t[m] * 1e6 / (16.0 * iters * 4), t[m] * 1e6 / (16.0 * iters * 4) * 2.2);
  }
  return 0;
}
