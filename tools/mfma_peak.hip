// Practical MFMA ceiling on gfx950 (experiment, not product): 32x32x16 bf16 MFMA loops with 1 or 2 waves per
// SIMD, with and without interleaved ds_read_b128 traffic (the gemm3 inner-loop mix), plus the shader clock
// under that load (s_memtime ticks vs the 100 MHz s_memrealtime).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mfma_peak.hip -o /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// NM independent MFMAs per iteration (NM accumulators), NR ds_read_b128 interleaved (one after each of the first NR)
template <int NM, int NR, int WAVES>
__global__ __launch_bounds__(WAVES * 64, 1) void k_mfma(int iters, float* sink, unsigned long long* clk) {
  __shared__ __attribute__((aligned(16))) char lds[65536];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 65536 / 4; i += blockDim.x) reinterpret_cast<float*>(lds)[i] = 1.0f;
  __syncthreads();
  f32x16 acc[NM];
#pragma unroll
  for (int n = 0; n < NM; ++n)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
  bf16x8 a[8], b;
#pragma unroll
  for (int e = 0; e < 8; ++e) b[e] = (__bf16)1.0f;
#pragma unroll
  for (int q = 0; q < 8; ++q) a[q] = b;
  const char* base = lds + (threadIdx.x >> 6) * 8192 + lane * 16;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int n = 0; n < NM; ++n) {
      acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[n & 7], b, acc[n], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (n < NR) a[n & 7] = *reinterpret_cast<const bf16x8*>(base + ((n + it) & 7) * 1024);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
#pragma unroll
  for (int n = 0; n < NM; ++n) s += acc[n][0] + acc[n][7];
  if (s == 12345.f) sink[threadIdx.x] = s;
  if (blockIdx.x == 17 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <typename F> float timeit(F&& f) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0)); f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms;
}

int main() {
  float* sink; CK(hipMalloc(&sink, 4096 * 4));
  unsigned long long* clk; CK(hipMalloc(&clk, 16));
  const int NCU = 256, iters = 20000;
#define RUN(NM, NR, W) { float ms = timeit([&] { hipLaunchKernelGGL((k_mfma<NM, NR, W>), dim3(NCU), dim3(W * 64), 0, 0, iters, sink, clk); }); \
    unsigned long long h[2]; CK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost)); \
    double flops = (double)NCU * W * iters * NM * 32768.0; \
    printf("  %2d MFMA + %2d ds_read_b128 per iter, %d waves/CU: %7.1f us  %7.1f TFLOP/s  (%.1f%% of 2.5 PF)  memtime/realtime = %.3f  cycles/MFMA/SIMD = %.1f (at 2.4 GHz)\n", \
           NM, NR, W, ms * 1e3, flops / (ms * 1e-3) / 1e12, flops / (ms * 1e-3) / 2.5e15 * 100, (double)h[0] / (double)h[1], ms * 1e-3 * 2.4e9 / ((double)iters * NM * (W / 4.0))); }
  printf("== v_mfma_f32_32x32x16_bf16, register operands, 256 workgroups (1 per CU)\n");
  RUN(12, 0, 4) RUN(12, 0, 8) RUN(12, 7, 4) RUN(12, 7, 8) RUN(16, 8, 4) RUN(12, 12, 4) RUN(4, 0, 4) RUN(4, 0, 8) RUN(4, 0, 16)
  return 0;
}
