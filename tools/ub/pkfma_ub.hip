// microbenchmark: v_pk_fma_f32 throughput with and without op_sel broadcast
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void __launch_bounds__(64) k(float* out, long long* cyc, int iters) {
  f2 acc[16]; f2 a = {1.0001f + threadIdx.x * 1e-6f, 0.9999f}; f2 b[4];
  for (int i = 0; i < 16; ++i) acc[i] = f2{(float)i, (float)i + 0.5f};
  for (int i = 0; i < 4; ++i) b[i] = f2{1e-3f * i, 2e-3f * i};
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (MODE == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b[i & 3]));
      if (MODE == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc[i]) : "v"(a), "v"(b[i & 3]));
      if (MODE == 2) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc[i]) : "v"(a), "v"(b[i & 3]));
      if (MODE == 3) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i].x) : "v"(a.x), "v"(b[i & 3].x));
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0; for (int i = 0; i < 16; ++i) s += acc[i].x + acc[i].y;
  out[blockIdx.x * 64 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
  float* out; long long* cyc; hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 8);
  const int iters = 10000;
  for (int mode = 0; mode < 4; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      if (mode == 0) k<0><<<1, 64>>>(out, cyc, iters); if (mode == 1) k<1><<<1, 64>>>(out, cyc, iters);
      if (mode == 2) k<2><<<1, 64>>>(out, cyc, iters); if (mode == 3) k<3><<<1, 64>>>(out, cyc, iters);
      hipDeviceSynchronize();
    }
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("mode %d: %.2f memtime-ticks per instruction\n", mode, (double)c / (iters * 16.0));
  }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0); k<0><<<1, 64>>>(out, cyc, 1000000); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  printf("memtime rate: %.1f MHz (ticks %lld in %.3f ms)\n", c / (ms * 1e3), c, ms);
  return 0;
}
