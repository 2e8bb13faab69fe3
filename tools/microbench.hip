// Per-CU vector-memory microbenchmarks (experiment, not product): LDS-DMA read rate from an L2-resident
// buffer, and store rates for row-strided 8/16-byte-per-lane vs contiguous patterns, with 8 waves per CU.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/microbench.hip -o /tmp/microbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned u32x4;
typedef __attribute__((__vector_size__(2 * sizeof(unsigned)))) unsigned u32x2;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// mode 0: each wave issues PIECES glds (1 KB each) per iteration, waits so that LAG iterations stay in flight
template <int PIECES, int ROWB, bool BARRIER>
__global__ __launch_bounds__(512, 2) void k_glds(const char* __restrict__ w, size_t wbytes, int iters, float* sink) {
  constexpr int NSLOT = (150 / (8 * PIECES)) >= 3 ? 3 : 2;
  __shared__ __attribute__((aligned(16))) char smem[150 * 1024];          // 1 workgroup per CU, like the panel kernel
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  constexpr int LPR = ROWB / 16;
  const int row = lane / LPR, ch = lane % LPR;
  const size_t stage_bytes = (size_t)8 * PIECES * 1024;
  size_t off = 0;
  for (int it = 0; it < iters; ++it) {
    char* dst = smem + (it % NSLOT) * stage_bytes;
#pragma unroll
    for (int i = 0; i < PIECES; ++i) {
      const int piece = wv * PIECES + i;
      const char* g = w + off + (size_t)(piece * (64 / LPR) + row) * 768 + ch * 16;   // rows 768 B apart like W[n][384]
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                       (__attribute__((address_space(3))) void*)(dst + piece * 1024), 16, 0, 0);
    }
    if (PIECES == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else if (PIECES == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if (PIECES == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    if (BARRIER) __builtin_amdgcn_s_barrier();
    off += ROWB;                       // walk along K like the ring stages do
    if (off + (size_t)8 * PIECES * 64 / LPR * 768 + 768 > wbytes) off = 0;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) sink[blockIdx.x] = *reinterpret_cast<float*>(smem);
}

// mode 3: the LDS-DMA stream through BUFFER instructions (SGPR resource + 32-bit per-lane offset instead of a 64-bit per-lane address)
template <int PIECES>
__global__ __launch_bounds__(512, 2) void k_bufglds(const char* __restrict__ w, size_t wbytes, int iters, float* sink) {
  constexpr int NSLOT = (150 / (8 * PIECES)) >= 3 ? 3 : 2;
  __shared__ __attribute__((aligned(16))) char smem[150 * 1024];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(w), 0, (int)wbytes, 0x00020000);
  const size_t stage_bytes = (size_t)8 * PIECES * 1024;
  unsigned off = 0;
  for (int it = 0; it < iters; ++it) {
    char* dst = smem + (it % NSLOT) * stage_bytes;
#pragma unroll
    for (int i = 0; i < PIECES; ++i) {
      const int piece = wv * PIECES + i;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(dst + piece * 1024), 16, lane * 16, (int)(off + piece * 1024), 0, 0);
    }
    if (PIECES == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else if (PIECES == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if (PIECES == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    off += (unsigned)stage_bytes;
    if (off + stage_bytes > wbytes) off = 0;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) sink[blockIdx.x] = *reinterpret_cast<float*>(smem);
}

// the same with global_load_lds and fully contiguous 1 KB pieces (the blocked weight copies of the product kernels)
template <int PIECES, int THREADS>
__global__ __launch_bounds__(THREADS, 1) void k_glds_contig(const char* __restrict__ w, size_t wbytes, int iters, float* sink) {
  constexpr int NW = THREADS / 64;
  __shared__ __attribute__((aligned(16))) char smem[3 * NW * PIECES * 1024];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const size_t stage_bytes = (size_t)NW * PIECES * 1024;
  size_t off = 0;
  for (int it = 0; it < iters; ++it) {
    char* dst = smem + (it % 3) * stage_bytes;
#pragma unroll
    for (int i = 0; i < PIECES; ++i) {
      const int piece = wv * PIECES + i;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(w + off + piece * 1024 + lane * 16),
                                       (__attribute__((address_space(3))) void*)(dst + piece * 1024), 16, 0, 0);
    }
    if (PIECES == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else if (PIECES == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if (PIECES == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    off += stage_bytes;
    if (off + stage_bytes > wbytes) off = 0;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) sink[blockIdx.x] = *reinterpret_cast<float*>(smem);
}

// mode 2: the same L2-resident stream by ORDINARY loads (16 B/lane to VGPRs, UNR loads in flight per lane), contiguous 1 KB per wave
// instruction: is the ~37 GB/s per CU of the LDS-DMA path a property of global_load_lds or of the L2 -> CU path?
template <int UNR>
__global__ __launch_bounds__(512, 2) void k_plain(const char* __restrict__ w, size_t wbytes, int iters, float* sink) {
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  size_t off = (size_t)wv * UNR * 1024;
  u32x4 acc = {0u, 0u, 0u, 0u};
  for (int it = 0; it < iters; ++it) {
    u32x4 v[UNR];
#pragma unroll
    for (int i = 0; i < UNR; ++i) v[i] = *reinterpret_cast<const u32x4*>(w + off + i * 1024 + lane * 16);
#pragma unroll
    for (int i = 0; i < UNR; ++i) acc ^= v[i];
    off += (size_t)8 * UNR * 1024;
    if (off + (size_t)UNR * 1024 > wbytes) off = (size_t)wv * UNR * 1024;
  }
  if (acc[0] == 0x12345678u && acc[3] == 7u) sink[blockIdx.x] = 1.f;
}

// mode 1: stores.  STRIDED: lane (t = lane&31, half) writes BYTES at row t (row pitch `pitch`), + half*BYTES
//         contiguous: lane writes BYTES at base + lane*BYTES
template <int BYTES, bool STRIDED>
__global__ __launch_bounds__(512, 2) void k_store(char* __restrict__ out, size_t pitch, int iters, int per_iter) {
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const size_t wg_base = (size_t)blockIdx.x * 128 * pitch;                 // 128 rows per workgroup
  u32x4 v = {(unsigned)lane, (unsigned)wv, 3u, 4u};
  for (int it = 0; it < iters; ++it) {
#pragma unroll 4
    for (int i = 0; i < per_iter; ++i) {
      const int col = (it * per_iter + i) * 2 * BYTES;                      // advances along the row
      size_t a;
      if (STRIDED) a = wg_base + (size_t)((wv & 3) * 32 + (lane & 31)) * pitch + (col % (pitch - 64)) + (lane >> 5) * BYTES;
      else a = wg_base + (size_t)wv * 64 * BYTES * 64 + ((size_t)(it * per_iter + i) % 64) * 64 * BYTES + lane * BYTES;
      if (BYTES == 8) { u32x2 q = {v[0], v[1]}; *reinterpret_cast<u32x2*>(out + a) = q; }
      else *reinterpret_cast<u32x4*>(out + a) = v;
    }
  }
}

template <typename F> float timeit(F f, int reps = 5) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(a)); for (int i = 0; i < reps; ++i) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / reps;
}

int main() {
  const int NCU = 256;
  const size_t wbytes = 2u << 20;                       // 2 MB "weights": L2 resident
  char* w; CK(hipMalloc(&w, wbytes)); CK(hipMemset(w, 1, wbytes));
  float* sink; CK(hipMalloc(&sink, 4096 * 4));
  const size_t obytes = (size_t)1 << 30; char* out; CK(hipMalloc(&out, obytes));
  const double clk = 2.1e9;
  printf("== LDS-DMA (global_load_lds 16 B/lane) from a 2 MB L2-resident buffer, 256 workgroups x 8 waves, 1 WG/CU\n");
  const int iters = 4000;
#define GL(P, R, B) { float ms = timeit([&] { hipLaunchKernelGGL((k_glds<P, R, B>), dim3(NCU), dim3(512), 0, 0, w, wbytes, iters, sink); }); \
    double bytes = (double)iters * 8 * P * 1024; printf("  pieces/wave %d rowB %3d barrier %d: %7.1f us  %6.1f GB/s/CU  %5.1f B/clk/CU  chip %5.2f TB/s  (%.0f cyc/iter)\n", P, R, (int)B, ms * 1e3, bytes / (ms * 1e-3) / 1e9, bytes / (ms * 1e-3) / clk, bytes * NCU / (ms * 1e-3) / 1e12, ms * 1e-3 * clk / iters); }
  GL(1, 64, false) GL(2, 64, false) GL(2, 128, false) GL(2, 64, true) GL(2, 128, true) GL(4, 128, false) GL(4, 128, true) GL(8, 128, false) GL(8, 128, true)
  printf("== LDS-DMA by buffer_load ... lds (SGPR resource + 32-bit offsets), contiguous 1 KB pieces, 8 waves\n");
#define BG(P) { float ms = timeit([&] { hipLaunchKernelGGL((k_bufglds<P>), dim3(NCU), dim3(512), 0, 0, w, wbytes, iters, sink); }); \
    double bytes = (double)iters * 8 * P * 1024; printf("  pieces/wave %d: %7.1f us  %6.1f GB/s/CU  %5.1f B/clk/CU  (%.0f cyc/iter)\n", P, ms * 1e3, bytes / (ms * 1e-3) / 1e9, bytes / (ms * 1e-3) / clk, ms * 1e-3 * clk / iters); }
  BG(1) BG(2) BG(4) BG(8)
  printf("== LDS-DMA by global_load_lds, contiguous 1 KB pieces: 8 waves vs 4 waves (one per SIMD, the fused-MLP / gemm3 geometry)\n");
#define GC(P, T) { float ms = timeit([&] { hipLaunchKernelGGL((k_glds_contig<P, T>), dim3(NCU), dim3(T), 0, 0, w, wbytes, iters, sink); }); \
    double bytes = (double)iters * (T / 64) * P * 1024; printf("  %d waves, pieces/wave %d: %7.1f us  %6.1f GB/s/CU  %5.1f B/clk/CU  (%.0f cyc/iter, %.0f cyc per piece per wave)\n", T / 64, P, ms * 1e3, bytes / (ms * 1e-3) / 1e9, bytes / (ms * 1e-3) / clk, ms * 1e-3 * clk / iters, ms * 1e-3 * clk / iters / P); }
  GC(4, 512) GC(4, 256) GC(8, 256) GC(2, 256)
  printf("== ordinary global_load_dwordx4 from the same buffer (1 KB contiguous per wave instruction), 256 workgroups x 8 waves\n");
#define PL(U) { float ms = timeit([&] { hipLaunchKernelGGL((k_plain<U>), dim3(NCU), dim3(512), 0, 0, w, wbytes, iters, sink); }); \
    double bytes = (double)iters * 8 * U * 1024; printf("  loads in flight/lane %2d: %7.1f us  %6.1f GB/s/CU  %5.1f B/clk/CU  chip %5.2f TB/s\n", U, ms * 1e3, bytes / (ms * 1e-3) / 1e9, bytes / (ms * 1e-3) / clk, bytes * NCU / (ms * 1e-3) / 1e12); }
  PL(1) PL(2) PL(4) PL(8)
  printf("== stores, 256 workgroups x 8 waves; each store instruction = 64 lanes\n");
  const int sit = 200, per = 16;
#define ST(BY, STR, PITCH) { float ms = timeit([&] { hipLaunchKernelGGL((k_store<BY, STR>), dim3(NCU), dim3(512), 0, 0, out, (size_t)PITCH, sit, per); }); \
    double instr = (double)sit * per * 8; double bytes = instr * 64 * BY; printf("  %2d B/lane %-10s pitch %5d: %7.1f us  %6.1f GB/s/CU  %5.2f cyc per store instr per CU  chip %5.2f TB/s\n", BY, STR ? "row-strided" : "contiguous", (int)PITCH, ms * 1e3, bytes / (ms * 1e-3) / 1e9, ms * 1e-3 * clk / instr, bytes * NCU / (ms * 1e-3) / 1e12); }
  ST(8, true, 3072) ST(8, true, 2304) ST(8, false, 3072) ST(16, true, 1536) ST(16, false, 1536)
  return 0;
}
