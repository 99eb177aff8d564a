// valu_peak.hip — measures what one MI355X sustains for the vector-FP32
// instruction forms this library can and cannot use:
//   fma      v_fma_f32        (2 flop/lane/instr; forbidden on the exact path)
//   pk_fma   v_pk_fma_f32     (4 flop/lane/instr; forbidden on the exact path)
//   mul_add  v_mul_f32 + v_add_f32 alternating   (1 flop/lane/instr; allowed)
//   pk_mul_add  v_pk_mul_f32 + v_pk_add_f32      (2 flop/lane/instr; allowed)
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_peak tools/valu_peak.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));

#define ITERS 4096
#define CHAINS 8

template <int MODE> __global__ __launch_bounds__(256) void k(float *out, float s) {
  float x[CHAINS];
  f2 y[CHAINS];
  for (int c = 0; c < CHAINS; c++) {
    x[c] = threadIdx.x * 1e-3f + c;
    y[c] = (f2){x[c], x[c] + 0.5f};
  }
  const float m = 1.0f + s, a = s;
  const f2 m2 = {m, m}, a2 = {a, a};
  for (int it = 0; it < ITERS; it++) {
#pragma unroll
    for (int c = 0; c < CHAINS; c++) {
      if (MODE == 0) {
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[c]) : "v"(m), "v"(a));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[c]) : "v"(m), "v"(a));
      } else if (MODE == 1) {
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(y[c]) : "v"(m2), "v"(a2));
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(y[c]) : "v"(m2), "v"(a2));
      } else if (MODE == 2) {
        asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[c]) : "v"(m));
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[c]) : "v"(a));
      } else {
        asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(y[c]) : "v"(m2));
        asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(y[c]) : "v"(a2));
      }
    }
  }
  float r = 0;
  for (int c = 0; c < CHAINS; c++)
    r += x[c] + y[c].x + y[c].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int MODE> double run(const char *name, double flop_per_instr, float *d_out, int blocks) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d_out, 1e-9f);
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int r = 0; r < 5; r++) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d_out, 1e-9f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  double instr = (double)blocks * 256 * ITERS * CHAINS * 2;  // lane-instructions
  double tinstr = instr / (best * 1e-3) / 1e12;
  printf("%-11s blocks=%5d  %.3f ms  %.2f T lane-instr/s  %.2f TFLOP/s\n", name, blocks, best,
         tinstr, tinstr * flop_per_instr);
  return tinstr;
}

int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  printf("%s  CUs=%d  clock=%d MHz\n", p.name, p.multiProcessorCount, p.clockRate / 1000);
  float *d;
  hipMalloc(&d, sizeof(float) * 256 * 8192);
  for (int wps : {1, 2, 4, 8}) {   // waves per SIMD
    int blocks = p.multiProcessorCount * wps;  // 256 threads = 4 waves = 1 per SIMD
    printf("-- %d wave(s) per SIMD\n", wps);
    run<0>("fma", 2, d, blocks);
    run<1>("pk_fma", 4, d, blocks);
    run<2>("mul_add", 1, d, blocks);
    run<3>("pk_mul_add", 2, d, blocks);
  }
  hipFree(d);
  return 0;
}
