// development aid: what the instructions of the log-likelihood cell cost on gfx950 -- issue time per wavefront instruction
// and SIMD with 4 wavefronts per SIMD and 8 independent chains per wavefront (throughput, not latency), from the launch time,
// and the shader clock the chip sustained meanwhile (clock64 against wall_clock64 in block 0).
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
enum { FMA_VVV, FMA_VVS, FMA_VVC, FMAC, MUL_VV, MUL_VS, ADD_VV, CVT_U32, CVT_I32, FREXP_M, FREXP_E, RCP, MOV32, MOV64, ADD_U32, AND_OR, BFE, LSHL_ADD,
       CNDMASK, CMP, LSHL_ADD_U64, LDEXP, FMA_MIX2, NOPS };
template <int OP>
__global__ __launch_bounds__(256) void chain(int iters, double* out, long long* cyc, double sv) {
  double x[8]; int n[8];
  const double a = 1.0 + 1e-9 * threadIdx.x, b = 1e-12;
  for (int i = 0; i < 8; ++i) { x[i] = 1.0 + 1e-3 * i + 1e-6 * threadIdx.x; n[i] = threadIdx.x + i; }
  const long long t0 = wall_clock64(), c0 = clock64();
  for (int k = 0; k < iters; ++k) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (OP == FMA_VVV) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
        if (OP == FMA_VVS) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "s"(sv));
        if (OP == FMA_VVC) asm volatile("v_fma_f64 %0, %0, %1, 1.0" : "+v"(x[i]) : "v"(a));
        if (OP == FMAC) asm volatile("v_fmac_f64_e32 %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
        if (OP == MUL_VV) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x[i]) : "v"(a));
        if (OP == MUL_VS) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x[i]) : "s"(sv));
        if (OP == ADD_VV) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x[i]) : "v"(b));
        if (OP == CVT_U32) asm volatile("v_cvt_f64_u32_e32 %0, %1" : "=v"(x[i]) : "v"(n[i]));
        if (OP == CVT_I32) asm volatile("v_cvt_f64_i32_e32 %0, %1" : "=v"(x[i]) : "v"(n[i]));
        if (OP == FREXP_M) asm volatile("v_frexp_mant_f64_e32 %0, %0" : "+v"(x[i]));
        if (OP == FREXP_E) asm volatile("v_frexp_exp_i32_f64_e32 %0, %1" : "=v"(n[i]) : "v"(x[i]));
        if (OP == RCP) asm volatile("v_rcp_f64_e32 %0, %0\n\ts_nop 0" : "+v"(x[i]));
        if (OP == MOV32) asm volatile("v_mov_b32_e32 %0, %1" : "=v"(n[i]) : "v"(n[(i + 1) & 7]));
        if (OP == MOV64) asm volatile("v_mov_b64_e32 %0, %1" : "=v"(x[i]) : "v"(x[(i + 1) & 7]));
        if (OP == ADD_U32) asm volatile("v_add_u32_e32 %0, %0, %1" : "+v"(n[i]) : "v"(n[(i + 1) & 7]));
        if (OP == AND_OR) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(n[i]) : "v"(n[(i + 1) & 7]), "v"(n[(i + 2) & 7]));
        if (OP == BFE) asm volatile("v_bfe_u32 %0, %0, 12, 8" : "+v"(n[i]));
        if (OP == LSHL_ADD) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(n[i]) : "v"(n[(i + 1) & 7]));
        if (OP == CNDMASK) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(n[i]) : "v"(n[(i + 1) & 7]) : );
        if (OP == CMP) asm volatile("v_cmp_lt_i32_e32 vcc, 7, %0" : : "v"(n[i]) : "vcc");
        if (OP == LSHL_ADD_U64) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(x[i]) : "v"(a));
        if (OP == LDEXP) asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(x[i]) : "v"(n[i]));
        if (OP == FMA_MIX2) { if (i & 1) asm volatile("v_add_u32_e32 %0, %0, %1" : "+v"(n[i]) : "v"(n[(i + 1) & 7]));
                              else asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b)); }
      }
    }
  }
  const long long t1 = wall_clock64(), c1 = clock64();
  double s = 0;
  for (int i = 0; i < 8; ++i) s += x[i] + n[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) { cyc[2 * blockIdx.x] = t1 - t0; cyc[2 * blockIdx.x + 1] = c1 - c0; }
}
template <int OP>
int run(const char* name, double* d_out, long long* d_cyc) {
  const int iters = 1000, wgs_per_cu = 4, blocks = 256 * wgs_per_cu;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((chain<OP>), dim3(blocks), dim3(256), 0, 0, iters, d_out, d_cyc, 1.25);
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  const double inst = (double)wgs_per_cu * iters * 64;
  long long h[2];
  CK(hipMemcpy(h, d_cyc, sizeof h, hipMemcpyDeviceToHost));
  const double ghz = (double)h[1] / (double)h[0] * 0.1;
  printf("%-22s %.3f ns per instruction and SIMD = %.2f cycles at the %.2f GHz block 0 saw (%.2f at 2.4 GHz)\n", name, best * 1e6 / inst, best * 1e6 / inst * ghz, ghz, best * 1e6 / inst * 2.4);
  return 0;
}
int main() {
  double* d_out; long long* d_cyc;
  CK(hipMalloc(&d_out, sizeof(double) * 256 * 4 * 256)); CK(hipMalloc(&d_cyc, sizeof(long long) * 256 * 4 * 2));
#define R(op) run<op>(#op, d_out, d_cyc)
  R(FMA_VVV); R(FMA_VVS); R(FMA_VVC); R(FMAC); R(MUL_VV); R(MUL_VS); R(ADD_VV); R(CVT_U32); R(CVT_I32); R(FREXP_M); R(FREXP_E); R(RCP);
  R(MOV32); R(MOV64); R(ADD_U32); R(AND_OR); R(BFE); R(LSHL_ADD); R(CNDMASK); R(CMP); R(LSHL_ADD_U64); R(LDEXP); R(FMA_MIX2);
  return 0;
}
