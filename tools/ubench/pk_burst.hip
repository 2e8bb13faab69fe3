// Microbenchmark (gfx950), one wave per SIMD (4 waves / workgroup, 256 workgroups):
//  A. pure VALU bursts, no MFMA: cycles per v_pk_fma_f32 / v_pk_mul_f32 / v_fma_f32 / v_pk_fma_f16 in a long independent stream
//  B. G MFMAs back to back, then ONE burst of N packed FMAs (the "phase + burst" schedule): total cycles, i.e. what the burst adds
//  C. the fused MLP's W-fragment refill behind every MFMA with a lookahead of LA reads (s_waitcnt lgkmcnt(LA-1) in front of the MFMA)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int KIND> __global__ __launch_bounds__(256, 1) void burst(float* out, long long* ticks, int iters) {
  f32x2 p[16]; float f[16];
  for (int i = 0; i < 16; ++i) { f[i] = threadIdx.x * 0.01f + i; p[i] = f32x2{f[i], f[i] + 1.f}; }
  const float c = 0.999f; const f32x2 c2 = {c, c};
  const long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 64; ++j) {
      if constexpr (KIND == 0) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[j & 15]) : "v"(c2));
      else if constexpr (KIND == 1) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[j & 15]) : "v"(c2));
      else if constexpr (KIND == 2) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[j & 15]) : "v"(c));
      else if constexpr (KIND == 3) asm volatile("v_pk_fma_f16 %0, %0, %1, %1" : "+v"(f[j & 15]) : "v"(c));
      else if constexpr (KIND == 4) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[j & 15]) : "v"(c2));
      else if constexpr (KIND == 5) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[j & 3]) : "v"(c2));     // dependent every 4th
      else if constexpr (KIND == 6) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[j & 1]) : "v"(c2));     // dependent every 2nd
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += f[i] + p[i][0] + p[i][1];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

template <int G, int N, int KIND> __global__ __launch_bounds__(256, 1) void phase_burst(float* out, long long* ticks, int iters) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(threadIdx.x * 0.001f + e); b[e] = (__bf16)(e * 0.5f); }
  f32x2 p[16]; float f[16];
  for (int i = 0; i < 16; ++i) { f[i] = threadIdx.x * 0.01f + i; p[i] = f32x2{f[i], f[i] + 1.f}; }
  const float c = 0.999f; const f32x2 c2 = {c, c};
  const long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < G; ++m) {
      acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m & 3], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int j = 0; j < N; ++j) {
      if constexpr (KIND == 0) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[j & 15]) : "v"(c2));
      else asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[j & 15]) : "v"(c));
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  for (int i = 0; i < 16; ++i) s += f[i] + p[i][0] + p[i][1];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

// C: rolling W-fragment refill, LA reads in flight, + NV v_fma per gap
template <int LA, int NV> __global__ __launch_bounds__(256, 1) void refill(float* out, long long* ticks, int iters) {
  __shared__ __attribute__((aligned(16))) char lds[65536];
  for (int i = threadIdx.x; i < 65536 / 4; i += 256) reinterpret_cast<float*>(lds)[i] = i * 1e-6f;
  __syncthreads();
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  bf16x8 b; bf16x8 wf[4];
  for (int e = 0; e < 8; ++e) b[e] = (__bf16)(e * 0.5f);
  const unsigned base = (unsigned)(size_t)lds + (threadIdx.x & 63) * 16;
  for (int i = 0; i < 4; ++i) asm volatile("ds_read_b128 %0, %1 offset:0" : "=v"(wf[i]) : "v"(base + i * 1024));
  float f[16];
  for (int i = 0; i < 16; ++i) f[i] = threadIdx.x * 0.01f + i;
  const float c = 0.999f;
  const long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      // fragment (m & 3) was requested LA gaps ago (LA <= 4): the reads younger than it number LA-1
      if constexpr (LA == 4) asm volatile("s_waitcnt lgkmcnt(3)" ::: "memory");
      else if constexpr (LA == 3) asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
      else if constexpr (LA == 2) asm volatile("s_waitcnt lgkmcnt(1)" ::: "memory");
      else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[m % LA], b, acc[m & 3], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("ds_read_b128 %0, %1" : "=v"(wf[m % LA]) : "v"(base + ((it * 16 + m) & 63) * 1024) : "memory");
#pragma unroll
      for (int j = 0; j < NV; ++j) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[(m * NV + j) & 15]) : "v"(c));
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  for (int i = 0; i < 16; ++i) s += f[i];
  for (int i = 0; i < 4; ++i) s += (float)wf[i][0];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

static double mean_ticks(long long* ticks) {
  std::vector<long long> h(256);
  (void)hipMemcpy(h.data(), ticks, 256 * 8, hipMemcpyDeviceToHost);
  double m = 0; for (int i = 0; i < 256; ++i) m += h[i];
  return m / 256;
}
#define RUN(K, NAME, PER, ...) do { hipLaunchKernelGGL((K), dim3(256), dim3(256), 0, 0, out, ticks, 500); (void)hipDeviceSynchronize(); \
  hipLaunchKernelGGL((K), dim3(256), dim3(256), 0, 0, out, ticks, 500); (void)hipDeviceSynchronize(); \
  printf("%-52s %8.2f ticks per %s\n", NAME, mean_ticks(ticks) / (500.0 * (PER)), __VA_ARGS__); } while (0)

int main() {
  float* out; long long* ticks;
  (void)hipMalloc(&out, 256 * 256 * 4); (void)hipMalloc(&ticks, 256 * 8);
  RUN(burst<0>, "A pure burst v_pk_fma_f32 (16 chains)", 64, "instruction");
  RUN(burst<5>, "A pure burst v_pk_fma_f32 (4 chains)", 64, "instruction");
  RUN(burst<6>, "A pure burst v_pk_fma_f32 (2 chains)", 64, "instruction");
  RUN(burst<1>, "A pure burst v_pk_mul_f32", 64, "instruction");
  RUN(burst<4>, "A pure burst v_pk_add_f32", 64, "instruction");
  RUN(burst<2>, "A pure burst v_fma_f32", 64, "instruction");
  RUN(burst<3>, "A pure burst v_pk_fma_f16", 64, "instruction");
  RUN((phase_burst<16, 0, 0>), "B 16 MFMA, no burst", 1, "iteration");
  RUN((phase_burst<16, 16, 0>), "B 16 MFMA + 16 v_pk_fma_f32", 1, "iteration");
  RUN((phase_burst<16, 48, 0>), "B 16 MFMA + 48 v_pk_fma_f32", 1, "iteration");
  RUN((phase_burst<16, 96, 0>), "B 16 MFMA + 96 v_pk_fma_f32", 1, "iteration");
  RUN((phase_burst<16, 96, 1>), "B 16 MFMA + 96 v_fma_f32", 1, "iteration");
  RUN((phase_burst<4, 24, 0>), "B 4 MFMA + 24 v_pk_fma_f32", 1, "iteration");
  RUN((phase_burst<1, 6, 0>), "B 1 MFMA + 6 v_pk_fma_f32", 1, "iteration");
  RUN((refill<1, 0>), "C refill lookahead 1, 0 fma", 16, "MFMA");
  RUN((refill<2, 0>), "C refill lookahead 2, 0 fma", 16, "MFMA");
  RUN((refill<3, 0>), "C refill lookahead 3, 0 fma", 16, "MFMA");
  RUN((refill<4, 0>), "C refill lookahead 4, 0 fma", 16, "MFMA");
  RUN((refill<3, 3>), "C refill lookahead 3, 3 fma", 16, "MFMA");
  RUN((refill<3, 5>), "C refill lookahead 3, 5 fma", 16, "MFMA");
  RUN((refill<4, 5>), "C refill lookahead 4, 5 fma", 16, "MFMA");
  RUN((refill<3, 7>), "C refill lookahead 3, 7 fma", 16, "MFMA");
  return 0;
}
