// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access widths the engine uses (MI355X_MICROARCH.md, HBM
// section: the counter is exact for 16 B/lane streaming stores, reports half the bytes of 16 B/lane streaming reads, and is
// uncalibrated for other widths).  Each kernel reads N bytes once and writes N bytes once, with 4, 8 or 16 bytes per lane.
// Build: hipcc --offload-arch=gfx950 -O3 -o fetch_calib fetch_calib.hip ; run under rocprofv3 --kernel-trace --pmc FETCH_SIZE (then WRITE_SIZE).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <typename T>
__global__ void __launch_bounds__(256) copy_kernel(const T* __restrict__ in, T* __restrict__ out, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[i];
}
struct alignas(8) B8 { float a, b; };
struct alignas(16) B16 { float a, b, c, d; };

template <typename T>
void run(const char* name, size_t bytes) {
  T *a, *b;
  const size_t n = bytes / sizeof(T);
  if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&b, bytes) != hipSuccess) { printf("alloc failed\n"); exit(1); }
  (void)hipMemset(a, 1, bytes); (void)hipMemset(b, 0, bytes);
  (void)hipDeviceSynchronize();
  for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(copy_kernel<T>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, a, b, n);
  (void)hipDeviceSynchronize();
  printf("%s: %zu bytes read + %zu bytes written per launch\n", name, bytes, bytes);
  (void)hipFree(a); (void)hipFree(b);
}

int main() {
  const size_t bytes = (size_t)1 << 30;   // 1 GiB: four times the Infinity Cache
  run<float>("copy4 (4 B/lane)", bytes);
  run<B8>("copy8 (8 B/lane)", bytes);
  run<B16>("copy16 (16 B/lane)", bytes);
  return 0;
}
