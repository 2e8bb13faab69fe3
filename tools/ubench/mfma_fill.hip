// Microbenchmark (gfx950): cycles per v_mfma_f32_32x32x16_bf16 with K filler instructions of one kind in every MFMA gap,
// one wave per SIMD (4 waves / workgroup, 256 workgroups).  Usage: ./mfma_fill  -> table.  Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int KIND, int K> __device__ __forceinline__ void fill(float (&f)[16], f32x2 (&p)[8], float c, const char* lds, bf16x8 (&wf)[4], int idx) {
  // KIND 0: v_fma_f32 (independent chains), 1: v_pk_fma_f32, 2: ds_read_b128 + (K-1) v_fma, 3: v_cvt_pk_bf16, 4: s_nop
#pragma unroll
  for (int j = 0; j < K; ++j) {
    if constexpr (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[(idx * K + j) & 15]) : "v"(c));
    else if constexpr (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[(idx * K + j) & 7]) : "v"(f32x2{c, c}));
    else if constexpr (KIND == 2) {
      if (j == 0) asm volatile("ds_read_b128 %0, %1" : "=v"(wf[idx & 3]) : "v"((unsigned)(size_t)lds + (threadIdx.x & 63) * 16 + (idx & 3) * 1024));
      else asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[(idx * K + j) & 15]) : "v"(c));
    } else if constexpr (KIND == 3) asm volatile("v_fma_f32 %0, %0, %1, %1\n\tv_fma_f32 %0, %0, %1, %1" : "+v"(f[(idx * K + j) & 15]) : "v"(c));
    else if constexpr (KIND == 5) asm volatile("v_exp_f32 %0, %0" : "+v"(f[(idx * K + j) & 15]));
    else if constexpr (KIND == 6) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(f[(idx * K + j) & 15]) : "v"(c));
    else if constexpr (KIND == 7) { if (j & 1) asm volatile("v_exp_f32 %0, %0" : "+v"(f[(idx * K + j) & 15])); else asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[(idx * K + j) & 15]) : "v"(c)); }
    else if constexpr (KIND == 8) asm volatile("v_pk_fma_f16 %0, %0, %1, %1" : "+v"(f[(idx * K + j) & 15]) : "v"(c));
    else if constexpr (KIND == 9) asm volatile("v_pk_mul_f16 %0, %0, %1" : "+v"(f[(idx * K + j) & 15]) : "v"(c));
    else if constexpr (KIND == 10) asm volatile("v_pk_max_f16 %0, %0, %1" : "+v"(f[(idx * K + j) & 15]) : "v"(c));
    else if constexpr (KIND == 11) asm volatile("v_cvt_f32_f16 %0, %0" : "+v"(f[(idx * K + j) & 15]));
    else if constexpr (KIND == 12) asm volatile("v_rcp_f32 %0, %0" : "+v"(f[(idx * K + j) & 15]));
    else if constexpr (KIND == 13) asm volatile("s_mov_b32 s40, s41" ::: "s40");
    else if constexpr (KIND == 14) asm volatile("s_waitcnt lgkmcnt(0)");
    else if constexpr (KIND == 15) {          // the fused MLP's real gap: ds_read_b128 + satisfied counted wait + 2 scalar + (K-4) v_fma
      if (j == 0) asm volatile("ds_read_b128 %0, %1" : "=v"(wf[idx & 3]) : "v"((unsigned)(size_t)lds + (threadIdx.x & 63) * 16 + (idx & 3) * 1024));
      else if (j == 1) asm volatile("s_waitcnt lgkmcnt(1)");
      else if (j == 2 || j == 3) asm volatile("s_mov_b32 s40, s41" ::: "s40");
      else asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[(idx * K + j) & 15]) : "v"(c));
    } else if constexpr (KIND == 16) {        // sigmoid-form GELU value: mul, fma, mul, exp, add, rcp, mul  (7 per value; K values per gap)
      float& x = f[(idx * K + j) & 15];
      asm volatile("v_mul_f32 %0, %1, %1\n\tv_fma_f32 %0, %0, %2, %2\n\tv_mul_f32 %0, %0, %1\n\tv_exp_f32 %0, %0\n\tv_add_f32 %0, 1.0, %0\n\tv_rcp_f32 %0, %0\n\tv_mul_f32 %1, %1, %0"
                   : "=&v"(p[j & 7][0]), "+v"(x) : "v"(c));
    } else if constexpr (KIND == 17) {        // the same seven as two interleaved values (independent chains adjacent)
      float& x = f[(idx * K + j) & 7]; float& y = f[8 + ((idx * K + j) & 7)];
      asm volatile("v_mul_f32 %0, %2, %2\n\tv_mul_f32 %1, %3, %3\n\tv_fma_f32 %0, %0, %4, %4\n\tv_fma_f32 %1, %1, %4, %4\n\tv_mul_f32 %0, %0, %2\n\tv_mul_f32 %1, %1, %3\n\t"
                   "v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1\n\tv_add_f32 %0, 1.0, %0\n\tv_add_f32 %1, 1.0, %1\n\tv_rcp_f32 %0, %0\n\tv_rcp_f32 %1, %1\n\tv_mul_f32 %2, %2, %0\n\tv_mul_f32 %3, %3, %1"
                   : "=&v"(p[j & 7][0]), "=&v"(p[j & 7][1]), "+v"(x), "+v"(y) : "v"(c));
    }
    else asm volatile("s_nop 0");
  }
}

template <int KIND, int K> __global__ __launch_bounds__(256, 1) void bench(float* out, long long* ticks, int iters) {
  __shared__ __attribute__((aligned(16))) char lds[8192];
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  bf16x8 a, b; bf16x8 wf[4];
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(threadIdx.x * 0.001f + e); b[e] = (__bf16)(e * 0.5f); }
  for (int i = 0; i < 4; ++i) wf[i] = a;
  for (int i = threadIdx.x; i < 8192 / 4; i += 256) reinterpret_cast<float*>(lds)[i] = i;
  __syncthreads();
  float f[16]; f32x2 p[8];
  for (int i = 0; i < 16; ++i) f[i] = threadIdx.x * 0.01f + i;
  for (int i = 0; i < 8; ++i) p[i] = f32x2{f[i], f[i + 8]};
  const float c = 0.999f;
  const long long t0 = __builtin_amdgcn_s_memtime();
  const long long c0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, KIND == 2 ? wf[(m + 2) & 3] : b, acc[m & 3], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      fill<KIND, K>(f, p, c, lds, wf, m);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const long long c1 = __builtin_readcyclecounter();
  const long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  for (int i = 0; i < 16; ++i) s += f[i];
  for (int i = 0; i < 8; ++i) s += p[i][0] + p[i][1];
  for (int i = 0; i < 4; ++i) s += (float)wf[i][0];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) { ticks[blockIdx.x * 2] = t1 - t0; ticks[blockIdx.x * 2 + 1] = c1 - c0; }
}

template <int KIND, int K> void run(const char* name, float* out, long long* ticks) {
  const int iters = 2000;
  hipLaunchKernelGGL((bench<KIND, K>), dim3(256), dim3(256), 0, 0, out, ticks, iters);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL((bench<KIND, K>), dim3(256), dim3(256), 0, 0, out, ticks, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> h(512);
  hipMemcpy(h.data(), ticks, 512 * 8, hipMemcpyDeviceToHost);
  double mt = 0, mc = 0; for (int i = 0; i < 256; ++i) { mt += h[2 * i]; mc += h[2 * i + 1]; }
  const double n = 256.0 * iters * 16;
  printf("%-28s K=%d  memtime ticks/MFMA %6.1f  cyclecounter/MFMA %6.1f  ns/MFMA %6.2f\n", name, K, mt / n, mc / n, ms * 1e6 / (iters * 16.0));
}

int main() {
  float* out; long long* ticks;
  hipMalloc(&out, 256 * 256 * 4); hipMalloc(&ticks, 512 * 8);
  run<0, 0>("bare MFMA", out, ticks);
  run<0, 2>("v_fma_f32", out, ticks); run<0, 4>("v_fma_f32", out, ticks); run<0, 5>("v_fma_f32", out, ticks); run<0, 6>("v_fma_f32", out, ticks); run<0, 8>("v_fma_f32", out, ticks); run<0, 12>("v_fma_f32", out, ticks);
  run<1, 1>("v_pk_fma_f32", out, ticks); run<1, 2>("v_pk_fma_f32", out, ticks); run<1, 3>("v_pk_fma_f32", out, ticks); run<1, 4>("v_pk_fma_f32", out, ticks); run<1, 6>("v_pk_fma_f32", out, ticks);
  run<2, 1>("ds_read_b128 (+K-1 fma)", out, ticks); run<2, 3>("ds_read_b128 (+K-1 fma)", out, ticks); run<2, 5>("ds_read_b128 (+K-1 fma)", out, ticks); run<2, 7>("ds_read_b128 (+K-1 fma)", out, ticks);
  run<3, 2>("2x dependent v_fma pairs", out, ticks); run<3, 4>("2x dependent v_fma pairs", out, ticks);
  run<5, 1>("v_exp_f32", out, ticks); run<5, 2>("v_exp_f32", out, ticks); run<5, 3>("v_exp_f32", out, ticks); run<5, 4>("v_exp_f32", out, ticks); run<5, 6>("v_exp_f32", out, ticks); run<5, 8>("v_exp_f32", out, ticks);
  run<6, 4>("v_cvt_pk_bf16_f32", out, ticks); run<6, 6>("v_cvt_pk_bf16_f32", out, ticks);
  run<7, 4>("fma/exp alternating", out, ticks); run<7, 6>("fma/exp alternating", out, ticks); run<7, 8>("fma/exp alternating", out, ticks);
  run<4, 4>("s_nop 0", out, ticks); run<4, 8>("s_nop 0", out, ticks);
  // round 6: packed 16-bit VALU, conversions, reciprocal, scalar issue, the fused MLP's real gap mix, sigmoid-form GELU
  run<8, 2>("v_pk_fma_f16", out, ticks); run<8, 4>("v_pk_fma_f16", out, ticks); run<8, 6>("v_pk_fma_f16", out, ticks); run<8, 8>("v_pk_fma_f16", out, ticks);
  run<9, 4>("v_pk_mul_f16", out, ticks); run<9, 6>("v_pk_mul_f16", out, ticks); run<9, 8>("v_pk_mul_f16", out, ticks);
  run<10, 4>("v_pk_max_f16", out, ticks); run<10, 6>("v_pk_max_f16", out, ticks);
  run<11, 4>("v_cvt_f32_f16", out, ticks); run<11, 6>("v_cvt_f32_f16", out, ticks);
  run<12, 1>("v_rcp_f32", out, ticks); run<12, 2>("v_rcp_f32", out, ticks); run<12, 4>("v_rcp_f32", out, ticks);
  run<13, 2>("s_mov_b32", out, ticks); run<13, 4>("s_mov_b32", out, ticks); run<13, 8>("s_mov_b32", out, ticks);
  run<14, 1>("s_waitcnt lgkmcnt(0)", out, ticks); run<14, 2>("s_waitcnt lgkmcnt(0)", out, ticks);
  run<15, 4>("ds_read+wait+2 salu (+K-4 fma)", out, ticks); run<15, 6>("ds_read+wait+2 salu (+K-4 fma)", out, ticks); run<15, 8>("ds_read+wait+2 salu (+K-4 fma)", out, ticks); run<15, 9>("ds_read+wait+2 salu (+K-4 fma)", out, ticks); run<15, 10>("ds_read+wait+2 salu (+K-4 fma)", out, ticks);
  run<16, 1>("sigmoid gelu value (7 ops)", out, ticks);
  run<17, 1>("2 sigmoid gelu values (14 ops)", out, ticks);
  return 0;
}
