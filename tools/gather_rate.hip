// gather_rate.hip — how fast does a CU of one MI355X serve the node reads of a
// per-lane tree walk?  Every lane reads records of R x 16 bytes at its own
// pseudo-random place of a table (3 400 x 64 B = the 10 000-sphere tree, L2
// resident), `iters` times; the next index depends on what was read, like a walk,
// and 5 waves per SIMD hide the latency.  Variants: bytes per record (16 / 32 / 64,
// i.e. 1 / 2 / 4 dwordx4 loads per lane and visit) and how many lanes of a wave
// read the SAME record (1 = all different, 64 = one record per wave).
#include <hip/hip_runtime.h>
#include <cstdio>
template <int R>
__global__ __launch_bounds__(256, 5) void k(const float4 *tab, int n_rec, int iters, int share,
                                            unsigned *sink) {
  unsigned idx = (blockIdx.x * 256u + threadIdx.x) / share * 2654435761u;
  unsigned acc = 0;
  for (int i = 0; i < iters; i++) {
    const float4 *p = tab + 4 * (size_t)(idx % (unsigned)n_rec);
    float4 v[R];
#pragma unroll
    for (int r = 0; r < R; r++) v[r] = p[r];
    unsigned h = 0;
#pragma unroll
    for (int r = 0; r < R; r++) h += __float_as_uint(v[r].x) ^ __float_as_uint(v[r].w);
    acc += h;
    idx = idx * 1664525u + 1013904223u + (h & 1u);
  }
  if (acc == 0xdeadbeef) sink[0] = acc;
}
template <int R> void run(const float4 *tab, int n_rec, unsigned *sink, int share) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int blocks = 1280, iters = 2000;
  hipLaunchKernelGGL(k<R>, dim3(blocks), dim3(256), 0, 0, tab, n_rec, iters, share, sink);
  hipDeviceSynchronize();
  hipEventRecord(a);
  hipLaunchKernelGGL(k<R>, dim3(blocks), dim3(256), 0, 0, tab, n_rec, iters, share, sink);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double visits = (double)blocks * 256 * iters;        // lane-visits
  const double per_cu_cycle = visits / 256 / (ms * 1e-3 * 2.4e9);
  printf("%2d B per record, %2d lanes per record: %.3f ms, %.2f G lane-visits/s, %.3f lane-visits per CU "
         "and cycle (%.2f cycles per wave-visit and CU)\n", 16 * R, share, ms, visits / ms / 1e6,
         per_cu_cycle, 64 / per_cu_cycle);
}
int main() {
  const int n_rec = 3400;
  float4 *tab; unsigned *sink;
  hipMalloc(&tab, (size_t)n_rec * 64); hipMalloc(&sink, 64);
  hipMemset(tab, 0x3c, (size_t)n_rec * 64);
  for (int share : {1, 4, 64}) {
    run<1>(tab, n_rec, sink, share);
    run<2>(tab, n_rec, sink, share);
    run<4>(tab, n_rec, sink, share);
  }
  return 0;
}
