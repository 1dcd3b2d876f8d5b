// Host-side cost of the HIP calls the pipelined seam makes per chunk, with two busy streams.
// hipcc --offload-arch=gfx950 -O2 -o api_cost_probe.bin api_cost_probe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void k_small(double* x, int n) { int i = blockIdx.x * 256 + threadIdx.x; if (i < n) x[i] += 1.0; }
int main()
{
  const size_t bytes = 800000;
  char* d; hipMalloc((void**)&d, 4 * bytes);
  double* dx; hipMalloc((void**)&dx, 1 << 20);
  char* h; hipHostMalloc((void**)&h, 4 * bytes, hipHostMallocDefault); memset(h, 1, 4 * bytes);
  hipStream_t s, c; hipStreamCreateWithFlags(&s, hipStreamNonBlocking); hipStreamCreateWithFlags(&c, hipStreamNonBlocking);
  hipEvent_t ev[4]; for (auto& e : ev) hipEventCreateWithFlags(&e, hipEventDisableTiming);
  for (int rep = 0; rep < 6; ++rep)
  {
    double t[8] = { 0 };
    double t_all = now();
    for (int k = 0; k < 4; ++k)
    {
      double a = now(); hipMemcpyAsync(d + k * bytes, h + k * bytes, bytes, hipMemcpyHostToDevice, c);
      double b = now(); hipEventRecord(ev[k], c);
      double e0 = now(); hipStreamWaitEvent(s, ev[k], 0);
      double f = now(); hipLaunchKernelGGL(k_small, dim3(100), dim3(256), 0, s, dx, 25600);
      double g = now(); hipLaunchKernelGGL(k_small, dim3(1000), dim3(256), 0, s, dx, 256000);
      double hh = now();
      t[0] += b - a; t[1] += e0 - b; t[2] += f - e0; t[3] += g - f; t[4] += hh - g;
    }
    double issued = now();
    hipStreamSynchronize(s);
    double end = now();
    if (rep >= 2)
      printf("per chunk: memcpyAsync %.1f us, eventRecord %.1f, streamWaitEvent %.1f, launch %.1f, launch %.1f | 4 chunks issued in %.1f us, done after %.1f us\n",
             t[0] / 4, t[1] / 4, t[2] / 4, t[3] / 4, t[4] / 4, issued - t_all, end - t_all);
  }
  return 0;
}
