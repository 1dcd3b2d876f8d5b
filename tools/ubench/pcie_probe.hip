// What a 3.2 MB host->device and a 0.8 MB device->host copy cost from pageable, registered and hipHostMalloc'ed
// memory, one call or four chunks, and what hipEventSynchronize / hipStreamSynchronize cost after them.
// hipcc --offload-arch=gfx950 -O2 -o pcie_probe pcie_probe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
  const size_t up = 3200000, down = 800000;
  void* d; hipMalloc(&d, up);
  hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  char* pageable = (char*)malloc(up + 64); memset(pageable, 1, up + 64);
  char* reg = (char*)malloc(up + 64); memset(reg, 1, up + 64);
  double t0 = now();
  hipError_t er = hipHostRegister(reg + 8, up, hipHostRegisterDefault);
  printf("hipHostRegister(unaligned +8, 3.2 MB): %s, %.0f us\n", hipGetErrorString(er), now() - t0);
  char* pinned; hipHostMalloc((void**)&pinned, up, hipHostMallocDefault); memset(pinned, 1, up);
  struct { const char* name; char* p; } src[3] = { { "pageable", pageable }, { "registered", reg + 8 }, { "hipHostMalloc", pinned } };
  for (auto& b : src)
    for (int chunks : { 1, 4 })
    {
      for (int rep = 0; rep < 6; ++rep)
      {
        double a = now();
        for (int c = 0; c < chunks; ++c)
          hipMemcpyAsync((char*)d + c * (up / chunks), b.p + c * (up / chunks), up / chunks, hipMemcpyHostToDevice, s);
        double issued = now();
        hipStreamSynchronize(s);
        double e = now();
        if (rep >= 3)
          printf("H2D %-14s chunks %d: issue %.1f us, total %.1f us (%.1f GB/s)\n", b.name, chunks, issued - a, e - a, up / (e - a) / 1e3);
      }
      for (int rep = 0; rep < 6; ++rep)
      {
        double a = now();
        for (int c = 0; c < chunks; ++c)
          hipMemcpyAsync(b.p + c * (down / chunks), (char*)d + c * (down / chunks), down / chunks, hipMemcpyDeviceToHost, s);
        double issued = now();
        hipStreamSynchronize(s);
        double e = now();
        if (rep >= 3)
          printf("D2H %-14s chunks %d: issue %.1f us, total %.1f us (%.1f GB/s)\n", b.name, chunks, issued - a, e - a, down / (e - a) / 1e3);
      }
    }
  // event round trip
  hipEvent_t ev; hipEventCreateWithFlags(&ev, hipEventDisableTiming);
  for (int rep = 0; rep < 5; ++rep)
  {
    double a = now();
    hipMemcpyAsync(d, pinned, 64, hipMemcpyHostToDevice, s);
    hipEventRecord(ev, s);
    hipEventSynchronize(ev);
    printf("64 B copy + event record + hipEventSynchronize: %.1f us\n", now() - a);
  }
  // host scatter of 100 k doubles at stride 4
  std::vector<double> aos(400000), w(100000, 1.0);
  for (int rep = 0; rep < 3; ++rep)
  {
    double a = now();
    for (int i = 0; i < 100000; ++i) aos[4 * (size_t)i + 3] = w[i];
    printf("host scatter of 100 k weights: %.1f us\n", now() - a);
  }
  hipHostUnregister(reg + 8);
  return 0;
}
