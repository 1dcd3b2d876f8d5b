// Zero-copy alternatives for the host-buffer seam: a kernel that writes 100 k weights (8 B at a 32 B stride) straight
// into registered host memory, one that writes them contiguously, one that reads the 3.2 MB of records; the latency
// of a cross-stream event dependency; hipMemcpy2DAsync as a strided scatter.
// hipcc --offload-arch=gfx950 -O2 -o zero_copy_probe.bin zero_copy_probe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void k_write_strided(double* aos, const double* w, int n) { int i = blockIdx.x * 256 + threadIdx.x; if (i < n) aos[4 * (size_t)i + 3] = w[i]; }
__global__ void k_write_dense(double* out, const double* w, int n) { int i = blockIdx.x * 256 + threadIdx.x; if (i < n) out[i] = w[i]; }
__global__ void k_write_records(double4* aos, const double* w, int n) { int i = blockIdx.x * 256 + threadIdx.x; if (i < n) aos[i] = make_double4(w[i], w[i], w[i], w[i]); }
__global__ void k_read_records(const double4* aos, double* x, int n) { int i = blockIdx.x * 256 + threadIdx.x; if (i < n) { double4 v = aos[i]; x[i] = v.x + v.y + v.z + v.w; } }
__global__ void k_spin(long long cycles) { long long t = clock64(); while (clock64() - t < cycles) {} }
int main()
{
  const int n = 100000;
  double *d_w, *d_x; hipMalloc(&d_w, n * 8); hipMalloc(&d_x, n * 8); hipMemset(d_w, 0, n * 8);
  hipStream_t s, s2; hipStreamCreateWithFlags(&s, hipStreamNonBlocking); hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
  char* raw = (char*)malloc((size_t)n * 32 + 64); memset(raw, 1, (size_t)n * 32 + 64);
  double* host = (double*)(raw + 8);
  hipHostRegister(host, (size_t)n * 32, hipHostRegisterDefault);
  double* dev_view; hipHostGetDevicePointer((void**)&dev_view, host, 0);
  printf("host %p device view %p\n", (void*)host, (void*)dev_view);
  const int nb = (n + 255) / 256;
  for (int rep = 0; rep < 5; ++rep)
  {
    double a = now(); hipLaunchKernelGGL(k_write_strided, dim3(nb), dim3(256), 0, s, dev_view, d_w, n); hipStreamSynchronize(s);
    double b = now(); hipLaunchKernelGGL(k_write_dense, dim3(nb), dim3(256), 0, s, dev_view, d_w, n); hipStreamSynchronize(s);
    double c = now(); hipLaunchKernelGGL(k_write_records, dim3(nb), dim3(256), 0, s, (double4*)dev_view, d_w, n); hipStreamSynchronize(s);
    double d = now(); hipLaunchKernelGGL(k_read_records, dim3(nb), dim3(256), 0, s, (const double4*)dev_view, d_x, n); hipStreamSynchronize(s);
    double e = now(); hipLaunchKernelGGL(k_write_dense, dim3(1), dim3(64), 0, s, d_x, d_w, 64); hipStreamSynchronize(s);
    double f = now();
    hipMemcpy2DAsync(host + 3, 32, d_w, 8, 8, n, hipMemcpyDeviceToHost, s); hipStreamSynchronize(s);
    double g = now();
    printf("zero-copy: strided 8 B writes %.1f us | dense 0.8 MB %.1f us | records 3.2 MB write %.1f us | records read %.1f us | "
           "(empty launch+sync %.1f us) | hipMemcpy2DAsync scatter %.1f us\n", b - a, c - b, d - c, e - d, f - e, g - f);
  }
  // cross-stream dependency: kernel on s2 after an event recorded on s behind a 30 us spin
  hipEvent_t ev; hipEventCreateWithFlags(&ev, hipEventDisableTiming);
  hipEvent_t t0, t1, t2; hipEventCreate(&t0); hipEventCreate(&t1); hipEventCreate(&t2);
  for (int rep = 0; rep < 4; ++rep)
  {
    hipEventRecord(t0, s);
    hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s, 60000ll);
    hipEventRecord(t1, s);
    hipEventRecord(ev, s);
    hipStreamWaitEvent(s2, ev, 0);
    hipLaunchKernelGGL(k_write_dense, dim3(1), dim3(64), 0, s2, d_x, d_w, 64);
    hipEventRecord(t2, s2);
    hipStreamSynchronize(s2); hipStreamSynchronize(s);
    float spin = 0, tot = 0; hipEventElapsedTime(&spin, t0, t1); hipEventElapsedTime(&tot, t0, t2);
    printf("spin %.1f us; dependent kernel on the other stream ends %.1f us after the spin\n", spin * 1e3, (tot - spin) * 1e3);
  }
  hipHostUnregister(host);
  return 0;
}
