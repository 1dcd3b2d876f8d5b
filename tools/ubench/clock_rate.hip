// What clock does the shader core run at under this load, and how many core cycles does a wave-instruction take?
// clock64() = s_memtime (core clock), wall_clock64() = s_memrealtime (constant 100 MHz).
#include <hip/hip_runtime.h>
#include <cstdio>
template <int OP>
__global__ __launch_bounds__(256) void k(double* out, long long* clk, double a, double b, int n)
{
  double x0 = threadIdx.x * 1e-3 + a, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  unsigned i0 = threadIdx.x, i1 = i0 + 1, i2 = i0 + 2, i3 = i0 + 3, i4 = i0 + 4, i5 = i0 + 5, i6 = i0 + 6, i7 = i0 + 7;
  const long long c0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < n; ++it)
  {
    if (OP == 0) { x0 = fma(x0, a, b); x1 = fma(x1, a, b); x2 = fma(x2, a, b); x3 = fma(x3, a, b); x4 = fma(x4, a, b); x5 = fma(x5, a, b); x6 = fma(x6, a, b); x7 = fma(x7, a, b); }
    if (OP == 1) { i0 = __umul24(i0, 14) + 3; i1 = __umul24(i1, 14) + 3; i2 = __umul24(i2, 14) + 3; i3 = __umul24(i3, 14) + 3; i4 = __umul24(i4, 14) + 3; i5 = __umul24(i5, 14) + 3; i6 = __umul24(i6, 14) + 3; i7 = __umul24(i7, 14) + 3; }
  }
  const long long c1 = clock64(), w1 = wall_clock64();
  out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + i0 + i1 + i2 + i3 + i4 + i5 + i6 + i7;
  if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
}
template <int OP>
void run(const char* name, int waves_per_simd, double* d, long long* dc)
{
  const int n = 200000;
  hipLaunchKernelGGL(k<OP>, dim3(256 * waves_per_simd), dim3(256), 0, 0, d, dc, 1.0000001, 1e-9, n);
  long long h[2]; (void)hipMemcpy(h, dc, sizeof(h), hipMemcpyDeviceToHost);
  const double mhz = (double)h[0] / ((double)h[1] / 100.0);  // core cycles per microsecond
  const double cyc_per_inst = (double)h[0] / ((double)n * 8 * waves_per_simd);
  printf("%-14s %d waves/SIMD: core clock %.0f MHz, %.2f core cycles per wave-instruction per SIMD\n", name, waves_per_simd, mhz, cyc_per_inst);
}
int main()
{
  double* d; long long* dc; (void)hipMalloc(&d, 256 * 8 * 256 * sizeof(double)); (void)hipMalloc(&dc, 16);
  run<0>("v_fma_f64", 8, d, dc); run<0>("v_fma_f64", 4, d, dc); run<0>("v_fma_f64", 1, d, dc);
  run<1>("v_mad_u32_u24", 8, d, dc); run<1>("v_mad_u32_u24", 1, d, dc);
  return 0;
}
