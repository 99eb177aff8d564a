// atomic_rate.hip — how fast does one MI355X hand out queue slots?
// Every wave issues N returning atomicAdd()s (one lane, like a wave-aggregated
// queue push) on: one shared counter / one counter per 8 blocks / per block.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned *ctr, int n_ctr, int iters, unsigned *sink) {
  unsigned acc = 0;
  unsigned *c = ctr + (blockIdx.x % n_ctr) * 64;  // 256 B apart
  for (int i = 0; i < iters; i++) {
    unsigned v = 0;
    if ((threadIdx.x & 63) == 0) v = atomicAdd(c, 64u);
    acc += __shfl(v, 0, 64);
  }
  if (acc == 0xdeadbeef) sink[0] = acc;
}
int main() {
  unsigned *ctr, *sink; hipMalloc(&ctr, 1 << 20); hipMalloc(&sink, 64);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int blocks = 2048, iters = 64;
  for (int n_ctr : {1, 8, 64, 256, 2048}) {
    hipMemset(ctr, 0, 1 << 20);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, ctr, n_ctr, iters, sink);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, ctr, n_ctr, iters, sink);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double n = (double)blocks * 4 * iters;
    printf("counters=%5d  %8.0f atomics in %.3f ms  -> %.1f ns per atomic (aggregate), %.1f M/s\n",
           n_ctr, n, ms, ms * 1e6 / n, n / ms / 1e3);
  }
  return 0;
}
