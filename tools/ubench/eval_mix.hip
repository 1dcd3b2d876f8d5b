// The instruction mix of one likelihood-field evaluation, without memory: which part of it sets the pace?
// variants: full mix (4 fma_f64, 2 cvt_i32_f64, 2 min_u32, 4 address ops, add_f64), no integer ops, no conversions...
#include <hip/hip_runtime.h>
#include <cstdio>
template <int V>
__global__ __launch_bounds__(256, 4) void k(double* out, double a, double b, double c, double s, int n, unsigned lim)
{
  double acc[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
  double bx[8], by[8];
  for (int u = 0; u < 8; ++u) { bx[u] = threadIdx.x * 0.37 + u; by[u] = threadIdx.x * 0.11 - u; }
  double qx = a, qy = b;
  for (int it = 0; it < n; ++it)
  {
#pragma unroll
    for (int u = 0; u < 8; ++u)
    {
      const double vx = fma(c, bx[u], fma(-s, by[u], qx));
      const double vy = fma(s, bx[u], fma(c, by[u], qy));
      if (V == 0 || V == 1)
      {
        unsigned iu = (unsigned)(int)vx, iv = (unsigned)(int)vy;
        if (V == 0)
        {
          iu = min(iu, lim); iv = min(iv, lim + 1);
          const unsigned off = (iu << 4) + (iv << 1) + __umul24(iv & ~7u, lim);
          acc[u] += (double)(off & 0xff8u) * 1e-9;   // + 2 ops standing in for the gather result
        }
        else
          acc[u] += (double)((iu ^ iv) & 0xff8u) * 1e-9;
      }
      else
        acc[u] += vx * 1e-9 + vy;  // V == 2: only the fp64 part
    }
    qx += 1e-3; qy -= 1e-3;
  }
  double t = 0; for (int u = 0; u < 8; ++u) t += acc[u];
  out[blockIdx.x * 256 + threadIdx.x] = t;
}
template <int V>
void run(const char* name, double* d)
{
  const int n = 4000;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k<V>, dim3(256 * 4), dim3(256), 0, 0, d, 1000.5, 900.25, 0.8, 0.6, 16, 2001u);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k<V>, dim3(256 * 4), dim3(256), 0, 0, d, 1000.5, 900.25, 0.8, 0.6, n, 2001u);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  // 4 waves per SIMD x n x 8 evaluations
  printf("%-44s %7.3f ms  %6.1f ns per wave-evaluation per SIMD\n", name, ms, ms * 1e6 / (4.0 * n * 8));
}
int main()
{
  double* d; (void)hipMalloc(&d, 256 * 4 * 256 * sizeof(double));
  run<0>("full mix (fma x4, cvt x2, min x2, addr, add)", d);
  run<1>("no clamp / address (fma x4, cvt x2, add)", d);
  run<2>("fp64 only (fma x4, mul, add x2)", d);
  return 0;
}
