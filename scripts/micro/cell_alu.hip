// development aid: the arithmetic ceiling of the cell loop. The same cell_eval() as the log-likelihood kernel's row sweep
// (one table logarithm from LDS, one reciprocal, the two Stirling tails, seven accumulations, a renormalisation every four
// cells) on inputs made up in registers -- no count matrix, no per-gene work, no reduction -- at the kernel's occupancy
// (256 threads, 4 workgroups per CU). Prints cells per second and the HBM-roofline equivalent at 4 B per cell.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../../ppcseq_amd/csrc/ppcx_model.h"
using namespace ppcx;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ __launch_bounds__(256, 4) void cell_alu(const double* logtab, int iters, double* out) {
  __shared__ double tab[2 * kLogTabSize];
  for (int i = threadIdx.x; i < 2 * kLogTabSize; i += 256) tab[i] = logtab[i];
  __syncthreads();
  const int t = blockIdx.x * 256 + threadIdx.x;
  GeneParams<2> gp;
  gp.phi = 3.0 + 1e-3 * (t & 255); gp.invphi = 1.0 / gp.phi; gp.dlt = 0.01; gp.dps = 0.02; gp.sigma_raw = 0.0; gp.A = 0.0; gp.A1 = 0.0;
  gp.coef[0] = gp.coef[1] = 0.0;
  const double A = 0.7 + 1e-4 * (t & 1023);
  CellAcc<2> acc; acc.zero();
  int y = 40 + (t & 63);
  double e = 1.0 + 1e-3 * (t & 31);
  for (int k = 0; k < iters; ++k) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      (void)cell_eval<2>(y, e, A, gp, tab, acc);
      y = 8 + ((y * 5 + 1) & 1023);            // stays >= 8: a row-sweep cell for every lane
      e = e * 1.0000001 + 1e-9;
    }
    acc.renorm();
  }
  out[t] = acc.SA + acc.SL + acc.TL + acc.TD + acc.Px + acc.Sr + (double)acc.Pxe;
}
int main() {
  std::vector<double> tab(2 * kLogTabSize);
  fill_log_table(tab.data());
  double *d_tab, *d_out;
  const int blocks = 1024 * 8, iters = 500;
  CK(hipMalloc(&d_tab, sizeof(double) * tab.size())); CK(hipMalloc(&d_out, sizeof(double) * blocks * 256));
  CK(hipMemcpy(d_tab, tab.data(), sizeof(double) * tab.size(), hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(cell_alu, dim3(blocks), dim3(256), 0, 0, d_tab, iters, d_out);
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double cells = (double)blocks * 256 * iters * 4;
    printf("cell arithmetic alone: %.1f G cells/s (%.2f ms for %.3g cells); 4 B per cell => %.2f TB/s equivalent = %.1f %% of 8 TB/s\n",
           cells / ms / 1e6, ms, cells, cells * 4 / ms / 1e9, cells * 4 / ms / 1e9 / 8 * 100);
  }
  return 0;
}
