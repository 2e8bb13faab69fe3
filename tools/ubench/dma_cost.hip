// Microbenchmark (gfx950): what does the LDS-DMA weight ring cost beside an MFMA stream?  One wave per SIMD, 256 workgroups.
// Per "stage" of 16 MFMAs every wave issues NP pieces (global_load_lds_dwordx4, 1 KB each) from an L2-resident buffer,
// waits for the pieces of LAG stages ago (counted vmcnt) and optionally meets the other waves at a barrier; optionally a
// ds_read_b128 per MFMA from the ring.   Build: hipcc --offload-arch=gfx950 -O3 dma_cost.hip -o dma_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

// MODE bit 0: DMA pieces, bit 1: barrier, bit 2: ds_read per MFMA, bit 3: pieces spread over 4 gaps, bit 4: plain loads to VGPRs instead of DMA
template <int MODE> __global__ __launch_bounds__(256, 1) void bench(const char* wbuf, float* out, long long* ticks, int iters) {
  __shared__ __attribute__((aligned(16))) char ring[8 * 16384];
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  bf16x8 a, b, wf[4];
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(threadIdx.x * 0.001f + e); b[e] = (__bf16)(e * 0.5f); }
  for (int i = 0; i < 4; ++i) wf[i] = a;
  for (int i = threadIdx.x; i < 8 * 16384 / 4; i += 256) reinterpret_cast<float*>(ring)[i] = i;
  __syncthreads();
  const unsigned ring_lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)ring;
  float4 sink = {0, 0, 0, 0};
  const long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {               // one iteration = 8 stages (the whole ring once)
#pragma unroll
    for (int st = 0; st < 8; ++st) {
      const char* src = wbuf + ((size_t)((it * 8 + st) & 7) * 16384 + w * 4096 + lane * 16);
      const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(ring_lds + st * 16384 + w * 4096));
#pragma unroll
      for (int m = 0; m < 16; ++m) {
        if (m == 8) {
          if constexpr (MODE & 1) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
          if constexpr (MODE & 2) __builtin_amdgcn_s_barrier();
        }
        acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16((MODE & 4) ? wf[m & 3] : a, b, acc[m & 3], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (MODE & 4) wf[m & 3] = *reinterpret_cast<const bf16x8*>(ring + ((st + 1) & 7) * 16384 + (m & 3) * 4096 + (m >> 2) * 1024 + lane * 16);
        if constexpr ((MODE & 1) && !(MODE & 16)) {
          unsigned keep;
          if constexpr (MODE & 8) {
            if (m >= 8 && (m & 1) == 0) {
              asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                           : "=&s"(keep) : "v"(src + ((m - 8) >> 1) * 1024), "s"(dst + ((m - 8) >> 1) * 1024) : "memory");
            }
          } else if (m == 8) {
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\tglobal_load_lds_dwordx4 %1, off offset:1024\n\t"
                         "global_load_lds_dwordx4 %1, off offset:2048\n\tglobal_load_lds_dwordx4 %1, off offset:3072\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
          }
        }
        if constexpr ((MODE & 1) && (MODE & 16)) {
          if (m >= 8 && (m & 1) == 0) {
            f32x4v v;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(src + ((m - 8) >> 1) * 1024) : "memory");
            asm volatile("" :: "v"(v));
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const long long t1 = __builtin_amdgcn_s_memtime();
  float s = sink.x;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  for (int i = 0; i < 4; ++i) s += (float)wf[i][0];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

template <int MODE> void run(const char* name, const char* wbuf, float* out, long long* ticks) {
  const int iters = 300;
  for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL((bench<MODE>), dim3(256), dim3(256), 0, 0, wbuf, out, ticks, iters); hipDeviceSynchronize(); }
  std::vector<long long> h(256);
  hipMemcpy(h.data(), ticks, 256 * 8, hipMemcpyDeviceToHost);
  double mt = 0; for (int i = 0; i < 256; ++i) mt += h[i];
  printf("%-64s cycles/MFMA %6.1f\n", name, mt / (256.0 * iters * 128));
}

int main() {
  char* wbuf; float* out; long long* ticks;
  hipMalloc(&wbuf, 8 * 16384 + 65536); hipMemset(wbuf, 0, 8 * 16384 + 65536); hipMalloc(&out, 256 * 256 * 4); hipMalloc(&ticks, 256 * 8);
  run<0>("bare MFMA stream", wbuf, out, ticks);
  run<2>("+ barrier per 16", wbuf, out, ticks);
  run<4>("+ ds_read_b128 per MFMA", wbuf, out, ticks);
  run<6>("+ ds_read + barrier", wbuf, out, ticks);
  run<1>("+ 4 DMA pieces per 16 (burst)", wbuf, out, ticks);
  run<9>("+ 4 DMA pieces per 16 (spread)", wbuf, out, ticks);
  run<3>("+ DMA burst + barrier", wbuf, out, ticks);
  run<5>("+ DMA burst + ds_read", wbuf, out, ticks);
  run<7>("+ DMA burst + ds_read + barrier   (the ring)", wbuf, out, ticks);
  run<15>("+ DMA spread + ds_read + barrier", wbuf, out, ticks);
  run<17>("+ 4 plain global_load_dwordx4 per 16 (to VGPRs)", wbuf, out, ticks);
  run<23>("+ plain loads + ds_read + barrier", wbuf, out, ticks);
  return 0;
}
