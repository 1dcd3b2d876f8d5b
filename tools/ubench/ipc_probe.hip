// Feasibility probe for a kernel-driven exchange through IPC-mapped device memory (two processes, one or two GPUs):
//   ./ipc_probe A <dir> [dev]   allocates an uncached buffer, exports its IPC handle to <dir>/handle, then runs a
//                               kernel that waits (bounded) for B's flag and reports what it read
//   ./ipc_probe B <dir> [dev]   opens the handle and runs a kernel that writes a payload and the flag
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <unistd.h>
#define CK(x) do { hipError_t r_ = (x); if (r_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(r_)); return 2; } } while (0)

__global__ void k_wait(unsigned long long* buf, unsigned gen, long long* out)
{
  // buf[0] = flag, buf[8..8+63] payload
  long long spins = 0;
  const long long t0 = wall_clock64();
  while (__hip_atomic_load(&buf[0], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != gen)
  {
    __builtin_amdgcn_s_sleep(8);
    if (++spins > 40000000ll || wall_clock64() - t0 > 300000000ll)  // ~3 s of the 100 MHz clock
      break;
  }
  out[0] = spins;
  out[1] = (long long)buf[8 + 5];
  out[2] = wall_clock64() - t0;
  out[3] = (long long)__hip_atomic_load(&buf[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ void k_post(unsigned long long* buf, unsigned gen)
{
  if (threadIdx.x < 64)
    buf[8 + threadIdx.x] = 1000ull + threadIdx.x;
  __syncthreads();
  if (threadIdx.x == 0)
    __hip_atomic_store(&buf[0], (unsigned long long)gen, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

int main(int argc, char** argv)
{
  if (argc < 3) return 1;
  const std::string dir = argv[2];
  const int dev = argc > 3 ? atoi(argv[3]) : 0;
  CK(hipSetDevice(dev));
  if (argv[1][0] == 'A')
  {
    unsigned long long* buf = nullptr;
    hipError_t r = hipExtMallocWithFlags((void**)&buf, 4096, hipDeviceMallocUncached);
    printf("A: hipExtMallocWithFlags(uncached) -> %s\n", hipGetErrorString(r));
    if (r != hipSuccess)
      CK(hipExtMallocWithFlags((void**)&buf, 4096, hipDeviceMallocFinegrained));
    CK(hipMemset(buf, 0, 4096));
    hipIpcMemHandle_t h;
    CK(hipIpcGetMemHandle(&h, buf));
    FILE* f = fopen((dir + "/handle.tmp").c_str(), "wb");
    fwrite(&h, sizeof(h), 1, f);
    fclose(f);
    rename((dir + "/handle.tmp").c_str(), (dir + "/handle").c_str());
    long long* out;
    CK(hipHostMalloc((void**)&out, 64, 0));
    memset(out, 0, 64);
    hipLaunchKernelGGL(k_wait, dim3(1), dim3(1), 0, 0, buf, 7u, out);
    CK(hipDeviceSynchronize());
    printf("A: spins %lld payload[5] %lld waited %.3f ms flag %lld -> %s\n", out[0], out[1], out[2] / 1e5, out[3],
           (out[3] == 7 && out[1] == 1005) ? "OK" : "FAILED");
    return (out[3] == 7 && out[1] == 1005) ? 0 : 3;
  }
  // B
  hipIpcMemHandle_t h;
  for (int t = 0; t < 200; ++t)
  {
    FILE* f = fopen((dir + "/handle").c_str(), "rb");
    if (f) { size_t n = fread(&h, sizeof(h), 1, f); fclose(f); if (n == 1) break; }
    usleep(20000);
    if (t == 199) { printf("B: no handle\n"); return 4; }
  }
  void* p = nullptr;
  CK(hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess));
  usleep(300000);  // let A's kernel be spinning already
  hipLaunchKernelGGL(k_post, dim3(1), dim3(64), 0, 0, (unsigned long long*)p, 7u);
  CK(hipDeviceSynchronize());
  printf("B: posted\n");
  CK(hipIpcCloseMemHandle(p));
  return 0;
}
