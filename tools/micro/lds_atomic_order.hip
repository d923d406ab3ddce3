// Does a same-address LDS atomic of one wave instruction hand out its return values in lane order? (undocumented; the feature
// kernel's ordered scatter checks its result and falls back, this only tells which path will run)  hipcc --offload-arch=gfx950
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
__global__ void k(const int* key, int* out, int rounds) {
  __shared__ unsigned cnt[256];
  for (int i = threadIdx.x; i < 256; i += blockDim.x) cnt[i] = 0;
  __syncthreads();
  for (int r = 0; r < rounds; r++) {
    const int c = key[r * blockDim.x + threadIdx.x];
    const unsigned old = __hip_atomic_fetch_add(&cnt[c >> 1], 1u << (16 * (c & 1)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    out[r * blockDim.x + threadIdx.x] = (int)((old >> (16 * (c & 1))) & 0xFFFF);
  }
}
int main() {
  const int NT = 64, R = 64, N = NT * R;
  int* hk = (int*)malloc(N * 4); int* ho = (int*)malloc(N * 4);
  int *dk, *dout; hipMalloc(&dk, N * 4); hipMalloc(&dout, N * 4);
  long bad = 0, groups = 0;
  for (int trial = 0; trial < 200; trial++) {
    srand(trial);
    const int nkeys = 1 + (trial % 7) * (trial % 7) * 10;  // 1 .. 361 distinct counters (pairs share a word)
    for (int i = 0; i < N; i++) hk[i] = rand() % (nkeys < 512 ? nkeys : 512);
    hipMemcpy(dk, hk, N * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(NT), 0, 0, dk, dout, R);
    hipMemcpy(ho, dout, N * 4, hipMemcpyDeviceToHost);
    // single wave: instruction r precedes r + 1; inside an instruction the expected order is by lane
    int cnt[512] = {0};
    for (int i = 0; i < N; i++) { if (ho[i] != cnt[hk[i]]) bad++; cnt[hk[i]]++; groups++; }
  }
  printf("lds atomic rtn order: %ld of %ld return values differ from (round, lane) order\n", bad, groups);
  return 0;
}
