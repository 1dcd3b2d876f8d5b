// Vector-memory gather issue cost on gfx950 by element width and address pattern (L1/L2 resident).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <cstdlib>
#define ITER 2048
template <typename T>
__global__ __launch_bounds__(256) void k(const T* __restrict__ tab, const unsigned* __restrict__ idx, unsigned mask, unsigned long long* out, int n)
{
  unsigned i0 = idx[threadIdx.x + blockIdx.x * 256], i1 = i0 * 7 + 1, i2 = i0 * 13 + 5, i3 = i0 * 29 + 11;
  unsigned long long acc = 0;
  for (int it = 0; it < n; ++it)
  {
    const T a = tab[i0 & mask], b = tab[i1 & mask], c = tab[i2 & mask], d = tab[i3 & mask];
    acc += (unsigned long long)a + (unsigned long long)b + (unsigned long long)c + (unsigned long long)d;
    i0 = i0 * 1664525u + (unsigned)a + 1013904223u; i1 = i1 * 1664525u + (unsigned)b + 1013904223u;
    i2 = i2 * 1664525u + (unsigned)c + 1013904223u; i3 = i3 * 1664525u + (unsigned)d + 1013904223u;
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}
template <typename T>
void run(const char* name, unsigned elems, bool same)
{
  const int blocks = 256 * 8;
  std::vector<T> h(elems, 0);
  std::vector<unsigned> hi(blocks * 256);
  for (auto& v : hi) v = same ? 12345u : (unsigned)rand();
  T* d; unsigned* di; unsigned long long* o;
  hipMalloc(&d, elems * sizeof(T)); hipMalloc(&di, hi.size() * 4); hipMalloc(&o, hi.size() * 8);
  hipMemcpy(d, h.data(), elems * sizeof(T), hipMemcpyHostToDevice);
  hipMemcpy(di, hi.data(), hi.size() * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k<T>, dim3(blocks), dim3(256), 0, 0, d, di, elems - 1, o, 8);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k<T>, dim3(blocks), dim3(256), 0, 0, d, di, elems - 1, o, ITER);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  double winst = 32.0 * ITER * 4;  // wave-level gather instructions per CU (8 blocks x 4 waves)
  printf("%-34s %8.3f ms  %6.1f cycles per 64-lane gather per CU\n", name, ms, ms * 1e-3 * 2.4e9 / winst);
  (void)hipFree(d); (void)hipFree(di); (void)hipFree(o);
}
int main()
{
  // note: with zero-filled tables every chain converges to the same index sequence unless idx differ;
  // "same" = all lanes identical addresses, "rand" = lanes differ (table zero => index LCG per lane)
  run<uint16_t>("u16 same address", 1 << 11, true);
  run<uint16_t>("u16 random in 4 KB", 1 << 11, false);
  run<uint16_t>("u16 random in 64 KB", 1 << 15, false);
  run<uint16_t>("u16 random in 8 MB", 1 << 22, false);
  run<uint32_t>("u32 same address", 1 << 10, true);
  run<uint32_t>("u32 random in 4 KB", 1 << 10, false);
  run<uint32_t>("u32 random in 64 KB", 1 << 14, false);
  run<uint32_t>("u32 random in 8 MB", 1 << 21, false);
  run<uint64_t>("u64 random in 4 KB", 1 << 9, false);
  run<uint8_t>("u8 random in 4 KB", 1 << 12, false);
  return 0;
}
