// development aid: what the gene kernel's loads and stores cost without its arithmetic. Grid and access pattern of
// ppcx_gene_kernel<2> at 8 chains of cfg3 (79 x 8 workgroups of 256 threads, one thread per gene and chain; every value an
// 8-byte element of its own vector of the chain, 64 consecutive genes per wavefront): NR vectors read in one burst, NW
// vectors written, for the byte counts of a leaf (22 read, 29 written: 410 B per gene) and for fractions of them.
// Prints us per launch and TB/s: the floor the kernel's 15.4 us is to be read against.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int NR, int NW>
__global__ __launch_bounds__(256) void traffic(const double* in, double* out, int G, long stride, long chain_stride) {
  const int g = blockIdx.x * 256 + threadIdx.x;
  if (g >= G) return;
  const double* a = in + blockIdx.y * chain_stride + g;
  double* o = out + blockIdx.y * chain_stride + g;
  double v[NR > 0 ? NR : 1];
#pragma unroll
  for (int k = 0; k < NR; ++k) v[k] = a[k * stride];
  double s = 0.0;
#pragma unroll
  for (int k = 0; k < NR; ++k) s += v[k];
#pragma unroll
  for (int k = 0; k < NW; ++k) o[k * stride] = s + k;
}
__global__ void empty_kernel() {}

template <int NR, int NW>
int run(const double* in, double* out, int G, int chains, long stride, long chain_stride) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const dim3 grid((G + 255) / 256, chains);
  float best = 1e30f;
  for (int rep = 0; rep < 5; ++rep) {
    for (int i = 0; i < 20; ++i) traffic<NR, NW><<<grid, 256>>>(in, out, G, stride, chain_stride);
    CK(hipEventRecord(e0));
    for (int i = 0; i < 200; ++i) traffic<NR, NW><<<grid, 256>>>(in, out, G, stride, chain_stride);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  const double us = 1e3 * best / 200, mb = 8.0 * (NR + NW) * G * chains / 1e6;
  printf("%d chain(s): read %2d + write %2d vectors (%3d B per gene, %5.1f MB per launch): %6.2f us per launch, %5.2f TB/s\n", chains, NR, NW, 8 * (NR + NW), mb, us, mb / us);
  return 0;
}
int main() {
  const int G = 20000, chains = 8;
  const long stride = 60416, vecs = 67, chain_stride = stride * vecs;      // V_COUNT vectors of Dpad doubles per chain
  double *in, *out;
  CK(hipMalloc(&in, sizeof(double) * chain_stride * chains)); CK(hipMalloc(&out, sizeof(double) * chain_stride * chains));
  CK(hipMemset(in, 0, sizeof(double) * chain_stride * chains)); CK(hipMemset(out, 0, sizeof(double) * chain_stride * chains));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 20; ++i) empty_kernel<<<dim3(79, 8), 256>>>();
  CK(hipEventRecord(e0));
  for (int i = 0; i < 200; ++i) empty_kernel<<<dim3(79, 8), 256>>>();
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("empty launch of the same grid: %.2f us\n", 1e3 * ms / 200);
  if (run<22, 29>(in, out, G, chains, stride, chain_stride)) return 1;     // a leaf
  if (run<22, 23>(in, out, G, chains, stride, chain_stride)) return 1;     // without the proposal copies
  if (run<22, 24>(in, out, G, chains, stride, chain_stride)) return 1;     // without the anticipated constants
  if (run<16, 20>(in, out, G, chains, stride, chain_stride)) return 1;
  if (run<11, 15>(in, out, G, chains, stride, chain_stride)) return 1;     // half
  if (run<22, 0>(in, out, G, chains, stride, chain_stride)) return 1;      // the reads alone
  if (run<1, 29>(in, out, G, chains, stride, chain_stride)) return 1;      // the writes alone
  for (int ch : {1, 3}) {                                                  // few chains: the launch's latency, not its bytes
    if (run<22, 29>(in, out, G, ch, stride, chain_stride)) return 1;
    if (run<12, 11>(in, out, G, ch, stride, chain_stride)) return 1;       // the coordinates' work alone
  }
  return 0;
}
