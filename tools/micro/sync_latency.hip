// sync_latency.hip -- what one host round trip costs on this box, by mechanism (round 6, drop-in route):
//   hipcc --offload-arch=gfx950 -O2 -o sync_latency sync_latency.hip && ./sync_latency
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>
__global__ void fill(float* p, int n, float v) { for (int i = threadIdx.x + blockIdx.x * blockDim.x; i < n; i += blockDim.x * gridDim.x) p[i] = v + i; }
__global__ void fill_flag(float* p, int n, float v, volatile int* flag, int stamp) {
  for (int i = threadIdx.x; i < n; i += blockDim.x) p[i] = v + i;
  __syncthreads();
  if (threadIdx.x == 0) { __threadfence_system(); *flag = stamp; }
}
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  const int n = 4800 * 3, iters = 300;
  hipStream_t st; hipStreamCreate(&st);
  float *d, *hp, *hm; hipMalloc(&d, n * 4); hipHostMalloc(&hp, n * 4, hipHostMallocDefault); hipHostMalloc(&hm, n * 4, hipHostMallocMapped);
  int* flag; hipHostMalloc(&flag, 64, hipHostMallocMapped); *flag = 0;
  std::vector<float> pageable(n);
  unsigned char *dimg, *himg; const size_t ib = 400 * 3360; hipMalloc(&dimg, ib); hipHostMalloc(&himg, ib, hipHostMallocDefault);
  for (int rep = 0; rep < 2; rep++) {
    double t0 = now();
    for (int i = 0; i < iters; i++) { hipLaunchKernelGGL(fill, dim3(1), dim3(512), 0, st, d, n, 1.f); hipStreamSynchronize(st); }
    double a = (now() - t0) / iters; t0 = now();
    for (int i = 0; i < iters; i++) { hipLaunchKernelGGL(fill, dim3(1), dim3(512), 0, st, d, n, 1.f); hipMemcpyAsync(hp, d, n * 4, hipMemcpyDeviceToHost, st); hipStreamSynchronize(st); }
    double b = (now() - t0) / iters; t0 = now();
    for (int i = 0; i < iters; i++) { hipLaunchKernelGGL(fill, dim3(1), dim3(512), 0, st, d, n, 1.f); hipMemcpyAsync(pageable.data(), d, n * 4, hipMemcpyDeviceToHost, st); hipStreamSynchronize(st); }
    double c = (now() - t0) / iters; t0 = now();
    for (int i = 0; i < iters; i++) { hipLaunchKernelGGL(fill, dim3(1), dim3(512), 0, st, hm, n, 1.f); hipStreamSynchronize(st); }
    double e = (now() - t0) / iters; t0 = now();
    for (int i = 0; i < iters; i++) { hipLaunchKernelGGL(fill_flag, dim3(1), dim3(512), 0, st, hm, n, 1.f, flag, i + 1 + rep * iters); while (*(volatile int*)flag != i + 1 + rep * iters) {} }
    double f = (now() - t0) / iters; hipStreamSynchronize(st); t0 = now();
    for (int i = 0; i < iters; i++) { hipLaunchKernelGGL(fill, dim3(1), dim3(512), 0, st, d, n, 1.f); hipLaunchKernelGGL(fill, dim3(1), dim3(512), 0, st, d, n, 2.f); hipLaunchKernelGGL(fill, dim3(1), dim3(512), 0, st, hm, n, 1.f); hipStreamSynchronize(st); }
    double g = (now() - t0) / iters; t0 = now();
    for (int i = 0; i < iters; i++) { hipMemcpyAsync(dimg, himg, ib, hipMemcpyHostToDevice, st); hipStreamSynchronize(st); }
    double h = (now() - t0) / iters; t0 = now();
    for (int i = 0; i < iters; i++) { hipMemcpyAsync(dimg, himg, ib, hipMemcpyHostToDevice, st); hipLaunchKernelGGL(fill, dim3(1), dim3(512), 0, st, hm, n, 1.f); hipStreamSynchronize(st); }
    double k = (now() - t0) / iters; t0 = now();
    for (int i = 0; i < iters; i++) { hipMemcpyAsync(d, hp, 2336, hipMemcpyHostToDevice, st); hipLaunchKernelGGL(fill, dim3(1), dim3(512), 0, st, d, n, 1.f); hipMemcpyAsync(hp, d, 4000, hipMemcpyDeviceToHost, st); hipStreamSynchronize(st); }
    double l = (now() - t0) / iters;
    printf("rep %d: kernel+sync %.1f us | kernel+D2H(pinned 57KB)+sync %.1f | kernel+D2H(pageable)+sync %.1f | kernel->mapped host+sync %.1f | kernel->mapped+flag poll %.1f | 3 kernels(last->mapped)+sync %.1f | H2D 1.34MB pinned+sync %.1f | H2D 1.34MB + kernel->mapped + sync %.1f | H2D 2KB + kernel + D2H 4KB + sync %.1f\n",
           rep, a, b, c, e, f, g, h, k, l);
  }
  return 0;
}
