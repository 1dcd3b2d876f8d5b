// Kernel start-to-end of near-empty launches in the shapes the resample kernels use: how much of a short kernel's
// duration is the launch shape itself (block size, dynamic LDS, a final store to pinned host memory)?
// build: hipcc --offload-arch=gfx950 -O2 -o launch_cost launch_cost.hip ; run: ./launch_cost
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>

__global__ void k_touch(int* out, volatile int* host, int host_words)
{
  extern __shared__ int lds[];
  if (threadIdx.x == 0)
    lds[0] = blockIdx.x;
  __syncthreads();
  if (threadIdx.x == 0 && blockIdx.x == 0)
  {
    out[0] = lds[0];
    for (int k = 0; k < host_words; ++k)
      host[k] = k;
    if (host_words)
      __threadfence_system();
  }
}

static float timed(int blocks, int threads, size_t lds, int host_words, int* d_out, int* h_pin, hipStream_t s)
{
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  hipFuncSetAttribute(reinterpret_cast<const void*>(k_touch), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
  float best = 1e9f, sum = 0;
  const int reps = 50;
  for (int r = 0; r < reps + 5; ++r)
  {
    hipExtLaunchKernelGGL(k_touch, dim3(blocks), dim3(threads), lds, s, a, b, 0, d_out, h_pin, host_words);
    hipStreamSynchronize(s);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    if (r >= 5)
    {
      sum += ms;
      if (ms < best)
        best = ms;
    }
  }
  printf("blocks %3d threads %4d lds %6zu host_words %2d : avg %.2f us  min %.2f us\n", blocks, threads, lds, host_words,
         sum / reps * 1e3, best * 1e3);
  return sum / reps;
}

int main()
{
  int *d_out, *h_pin;
  hipMalloc(&d_out, 64);
  hipHostMalloc(&h_pin, 256, hipHostMallocDefault);
  hipStream_t s;
  hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  timed(1, 64, 0, 0, d_out, h_pin, s);
  timed(1, 1024, 0, 0, d_out, h_pin, s);
  timed(1, 1024, 147472, 0, d_out, h_pin, s);
  timed(24, 1024, 0, 0, d_out, h_pin, s);
  timed(24, 1024, 147472, 0, d_out, h_pin, s);
  timed(24, 1024, 65536, 0, d_out, h_pin, s);
  timed(24, 256, 0, 0, d_out, h_pin, s);
  timed(49, 256, 0, 0, d_out, h_pin, s);
  timed(24, 1024, 147472, 8, d_out, h_pin, s);
  timed(24, 1024, 147472, 24, d_out, h_pin, s);
  timed(1, 1024, 0, 8, d_out, h_pin, s);
  return 0;
}
