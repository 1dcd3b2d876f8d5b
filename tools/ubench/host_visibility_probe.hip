// When do a kernel's PLAIN stores to host memory become visible to the host?  The kernel stores a word, then keeps
// running for ~2 ms; the host polls the word meanwhile.  For hipHostMalloc (default), hipHostMalloc (coherent) and
// hipHostRegister'ed malloc memory.
//   hipcc --offload-arch=gfx950 -O2 -o host_visibility_probe host_visibility_probe.hip && ./host_visibility_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>

__global__ void k_store_then_linger(volatile unsigned* p, unsigned v, long long ticks)
{
  if (threadIdx.x == 0 && blockIdx.x == 0)
    p[0] = v;
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks)
    __builtin_amdgcn_s_sleep(8);
}

static void probe(const char* name, volatile unsigned* host, unsigned* dev_view)
{
  host[0] = 0;
  hipStream_t s;
  hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  const auto t0 = std::chrono::steady_clock::now();
  hipLaunchKernelGGL(k_store_then_linger, dim3(64), dim3(64), 0, s, dev_view, 77u, 200000ll);  // 2 ms at 100 MHz
  double seen_at = -1.0;
  for (;;)
  {
    const double t = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    if (host[0] == 77u)
    {
      seen_at = t;
      break;
    }
    if (t > 20000.0)
      break;
  }
  hipStreamSynchronize(s);
  const double done = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
  printf("%-34s store seen after %8.1f us; kernel done after %8.1f us -> %s\n", name, seen_at, done,
         seen_at >= 0 && seen_at < done - 500.0 ? "written through while the kernel runs" : "visible only at the end");
  hipStreamDestroy(s);
}

int main()
{
  unsigned* a = nullptr;
  hipHostMalloc(reinterpret_cast<void**>(&a), 4096, hipHostMallocDefault);
  probe("hipHostMalloc default", a, a);
  unsigned* b = nullptr;
  hipHostMalloc(reinterpret_cast<void**>(&b), 4096, hipHostMallocCoherent);
  probe("hipHostMalloc coherent", b, b);
  unsigned* c = static_cast<unsigned*>(aligned_alloc(4096, 1 << 20));
  hipHostRegister(c, 1 << 20, hipHostRegisterDefault);
  void* dv = nullptr;
  hipHostGetDevicePointer(&dv, c, 0);
  probe("malloc + hipHostRegister default", c, static_cast<unsigned*>(dv));
  return 0;
}
