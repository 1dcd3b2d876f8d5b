// Per-instruction issue cost on gfx950 for the ops in the scoring loop: N independent chains per
// lane, 8 waves per SIMD, measured with hipEvents.  Prints cycles per wave-instruction per SIMD
// assuming 2.4 GHz.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define ITER 4096
template <int OP>
__global__ __launch_bounds__(256) void k(double* out, double a, double b, int n)
{
  double x0 = threadIdx.x * 1e-3 + a, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  int i0 = threadIdx.x, i1 = i0 + 1, i2 = i0 + 2, i3 = i0 + 3, i4 = i0 + 4, i5 = i0 + 5, i6 = i0 + 6, i7 = i0 + 7;
  for (int it = 0; it < n; ++it)
  {
    if (OP == 0) { x0 = fma(x0, a, b); x1 = fma(x1, a, b); x2 = fma(x2, a, b); x3 = fma(x3, a, b); x4 = fma(x4, a, b); x5 = fma(x5, a, b); x6 = fma(x6, a, b); x7 = fma(x7, a, b); }
    if (OP == 1) { x0 += b; x1 += b; x2 += b; x3 += b; x4 += b; x5 += b; x6 += b; x7 += b; }
    if (OP == 2) { i0 += (int)x0; i1 += (int)x1; i2 += (int)x2; i3 += (int)x3; i4 += (int)x4; i5 += (int)x5; i6 += (int)x6; i7 += (int)x7; x0 += 1; }
    if (OP == 3) { x0 = floor(x0 * a); x1 = floor(x1 * a); x2 = floor(x2 * a); x3 = floor(x3 * a); x4 = floor(x4 * a); x5 = floor(x5 * a); x6 = floor(x6 * a); x7 = floor(x7 * a); }
    if (OP == 4) { i0 = __umul24(i0, 14) + 3; i1 = __umul24(i1, 14) + 3; i2 = __umul24(i2, 14) + 3; i3 = __umul24(i3, 14) + 3; i4 = __umul24(i4, 14) + 3; i5 = __umul24(i5, 14) + 3; i6 = __umul24(i6, 14) + 3; i7 = __umul24(i7, 14) + 3; }
    if (OP == 5) { i0 = min((unsigned)i0 + 7u, 2001u); i1 = min((unsigned)i1 + 7u, 2001u); i2 = min((unsigned)i2 + 7u, 2001u); i3 = min((unsigned)i3 + 7u, 2001u); i4 = min((unsigned)i4 + 7u, 2001u); i5 = min((unsigned)i5 + 7u, 2001u); i6 = min((unsigned)i6 + 7u, 2001u); i7 = min((unsigned)i7 + 7u, 2001u); }
    if (OP == 7) { i0 = (int)(x0 + i0); i1 = (int)(x1 + i1); i2 = (int)(x2 + i2); i3 = (int)(x3 + i3); i4 = (int)(x4 + i4); i5 = (int)(x5 + i5); i6 = (int)(x6 + i6); i7 = (int)(x7 + i7); }
    if (OP == 8) { x0 = (double)i0 + x0; x1 = (double)i1 + x1; x2 = (double)i2 + x2; x3 = (double)i3 + x3; x4 = (double)i4 + x4; x5 = (double)i5 + x5; x6 = (double)i6 + x6; x7 = (double)i7 + x7; }
    if (OP == 6) { x0 = x0 * a; x1 = x1 * a; x2 = x2 * a; x3 = x3 * a; x4 = x4 * a; x5 = x5 * a; x6 = x6 * a; x7 = x7 * a; }
  }
  out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + i0 + i1 + i2 + i3 + i4 + i5 + i6 + i7;
}
template <int OP>
void run(const char* name, int ops_per_iter, double* d)
{
  const int blocks = 256 * 8;  // 8 blocks of 4 waves per CU = 8 waves per SIMD
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 1.0000001, 1e-9, 16);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 1.0000001, 1e-9, ITER);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  // wave-instructions per SIMD: 8 waves * ITER * ops
  double winst = 8.0 * ITER * ops_per_iter;
  double cycles = ms * 1e-3 * 2.4e9;
  printf("%-28s %8.3f ms  %6.2f cycles per wave-instruction per SIMD (incl. loop overhead)\n", name, ms, cycles / winst);
}
int main()
{
  double* d; hipMalloc(&d, 256 * 8 * 256 * sizeof(double));
  run<0>("v_fma_f64", 8, d);
  run<1>("v_add_f64", 8, d);
  run<6>("v_mul_f64", 8, d);
  run<2>("v_cvt_i32_f64 + v_add_u32 (16 instr)", 16, d);
  run<7>("cvt_f64_i32 + add_f64 + cvt_i32_f64 (24 instr)", 24, d);
  run<8>("cvt_f64_i32 + add_f64 (16 instr)", 16, d);
  run<3>("v_mul_f64 + v_floor_f64", 8, d);
  run<4>("v_mad_u32_u24 (mul24+add)", 8, d);
  run<5>("v_add_u32 + v_min_u32", 8, d);
  return 0;
}
