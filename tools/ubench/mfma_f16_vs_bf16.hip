// Is v_mfma_f32_32x32x16_f16 slower than _bf16 on gfx950 under sustained load?  (Round 4: the f16 mode of the encoder runs 4 % behind the
// bf16 mode with IDENTICAL instruction counts per kernel.)  Register-operand MFMA loops, one wave per SIMD on every CU, ~100 ms per run so that
// the power management settles; operands = pseudo-random values ~U(-2, 2) (data toggling, unlike an all-ones loop), optionally rotated per
// iteration.  Prints TFLOP/s and the shader clock under load (s_memtime ticks / 100 MHz s_memrealtime).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/mfma_f16_vs_bf16.hip -o tools/ubench/mfma_f16_vs_bf16
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(8 * sizeof(_Float16)))) _Float16 f16x8;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <typename E> struct T;
template <> struct T<__bf16> { typedef bf16x8 V; static __device__ f32x16 mm(V a, V b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); } };
template <> struct T<_Float16> { typedef f16x8 V; static __device__ f32x16 mm(V a, V b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); } };

template <typename E, int NM>
__global__ __launch_bounds__(256, 1) void k(int iters, float scale, float* sink, unsigned long long* clk) {
  typedef typename T<E>::V V;
  unsigned s = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((float)(s >> 8) * (1.0f / 8388608.0f) - 1.0f) * scale; };
  V a[8], b[4];
  for (int q = 0; q < 8; ++q) for (int e = 0; e < 8; ++e) a[q][e] = (E)rnd();
  for (int q = 0; q < 4; ++q) for (int e = 0; e < 8; ++e) b[q][e] = (E)rnd();
  f32x16 acc[NM];
  for (int n = 0; n < NM; ++n) for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; it += 2) {                 // (static register indices only: a dynamic one sends the operands through scratch)
#pragma unroll
    for (int n = 0; n < NM; ++n) acc[n] = T<E>::mm(a[n & 7], b[n & 3], acc[n]);
#pragma unroll
    for (int n = 0; n < NM; ++n) acc[n] = T<E>::mm(a[(n + 3) & 7], b[(n + 1) & 3], acc[n]);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float z = 0.f;
  for (int n = 0; n < NM; ++n) z += acc[n][0] + acc[n][9];
  if (z == 12345.f) sink[threadIdx.x] = z;
  if (blockIdx.x == 17 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <typename E> void run(const char* name, float scale, float* sink, unsigned long long* clk) {
  const int NCU = 256, iters = 400000, NM = 12;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k<E, NM>), dim3(NCU), dim3(256), 0, 0, iters / 10, scale, sink, clk); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL((k<E, NM>), dim3(NCU), dim3(256), 0, 0, iters, scale, sink, clk);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  unsigned long long h[2]; CK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost));
  const double flops = (double)NCU * 4 * iters * NM * 32768.0;
  printf("  %-5s operands ~U(-%g, %g): %8.2f ms  %7.1f TFLOP/s (%.1f%% of 2.5 PF)  shader clock under load %.3f GHz\n", name, scale, scale, ms,
         flops / (ms * 1e-3) / 1e12, flops / (ms * 1e-3) / 2.5e15 * 100, (double)h[0] / (double)h[1] * 0.1);
}

int main() {
  float* sink; CK(hipMalloc(&sink, 4096 * 4));
  unsigned long long* clk; CK(hipMalloc(&clk, 16));
  for (int round = 0; round < 3; ++round) {
    run<__bf16>("bf16", 2.0f, sink, clk);
    run<_Float16>("f16", 2.0f, sink, clk);
  }
  run<__bf16>("bf16", 0.0f, sink, clk);
  run<_Float16>("f16", 0.0f, sink, clk);
  return 0;
}
