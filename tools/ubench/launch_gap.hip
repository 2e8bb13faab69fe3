// What does the queue charge in front of a kernel that takes the whole CU?  Alternating launches (small, X) on one stream, X = a trivial
// kernel with (a) nothing special, (b) 140 KB of static LDS, (c) 512 registers per lane, (d) both; wall time per pair by HIP events.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void small_k(float* o) { o[blockIdx.x * 256 + threadIdx.x] = 1.f; }
#define BIGK(NAME, LDSKB, NV)                                                                       \
  __global__ __launch_bounds__(256, 1) __attribute__((amdgpu_num_vgpr(NV))) void NAME(float* o) {   \
    __shared__ float lds[LDSKB * 256];                                                              \
    lds[threadIdx.x] = threadIdx.x;                                                                 \
    __syncthreads();                                                                                \
    o[blockIdx.x * 256 + threadIdx.x] = lds[255 - threadIdx.x];                                     \
  }
BIGK(k_plain, 1, 64)
BIGK(k_lds140, 140, 64)
BIGK(k_lds64, 64, 64)
BIGK(k_reg512, 1, 512)
BIGK(k_both, 140, 512)
// a big code object: ~N x 64 KB of straight-line code behind a branch that is never taken
template <int N> __device__ __forceinline__ void bloat(float* o) {
  float v = o[0];
#pragma unroll
  for (int i = 0; i < N * 4000; ++i) asm volatile("v_fma_f32 %0, %0, %0, %0\n\tv_add_f32 %0, 1.0, %0" : "+v"(v));
  o[1] = v;
}
__global__ __launch_bounds__(256, 1) void k_code(float* o, int never) {
  if (never) bloat<4>(o);
  o[blockIdx.x * 256 + threadIdx.x] = 2.f;
}
__global__ __launch_bounds__(256, 1) void k_code_big(float* o, int never) {
  if (never) bloat<16>(o);
  o[blockIdx.x * 256 + threadIdx.x] = 2.f;
}
template <typename K> float run2(K k, float* o, int n, bool pair) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 2; ++w) {
    hipEventRecord(e0);
    for (int i = 0; i < n; ++i) {
      if (pair) hipLaunchKernelGGL(small_k, dim3(256), dim3(256), 0, 0, o);
      hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, 0, o, 0);
    }
    hipEventRecord(e1); hipEventSynchronize(e1);
  }
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms * 1000.f / n;
}
typedef void (*kfn)(float*);
float run(kfn k, float* o, int n, bool pair) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 2; ++w) {
    hipEventRecord(e0);
    for (int i = 0; i < n; ++i) {
      if (pair) hipLaunchKernelGGL(small_k, dim3(256), dim3(256), 0, 0, o);
      hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, 0, o);
    }
    hipEventRecord(e1); hipEventSynchronize(e1);
  }
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms * 1000.f / n;
}
int main() {
  float* o; hipMalloc(&o, 256 * 256 * 4);
  const int n = 2000;
  printf("us per (small + X) pair | per X alone\n");
  printf("plain            %6.2f | %6.2f\n", run(k_plain, o, n, true), run(k_plain, o, n, false));
  printf("140 KB LDS       %6.2f | %6.2f\n", run(k_lds140, o, n, true), run(k_lds140, o, n, false));
  printf("64 KB LDS        %6.2f | %6.2f\n", run(k_lds64, o, n, true), run(k_lds64, o, n, false));
  printf("512 registers    %6.2f | %6.2f\n", run(k_reg512, o, n, true), run(k_reg512, o, n, false));
  printf("both             %6.2f | %6.2f\n", run(k_both, o, n, true), run(k_both, o, n, false));
  printf("256 KB of code   %6.2f | %6.2f\n", run2(k_code, o, n, true), run2(k_code, o, n, false));
  printf("1 MB of code     %6.2f | %6.2f\n", run2(k_code_big, o, n, true), run2(k_code_big, o, n, false));
  return 0;
}
