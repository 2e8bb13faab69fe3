// K-streaming NT GEMM, third generation ("gemm3"): out = epilogue(X[M,K] . W[N,K]^T + bias), bf16 / f16
// operands, BOTH operands and the output in the fragment-blocked layout (common.hpp blk_off).
//
// Why (profiles/README.md, "library yardstick"): gemm2's 64x64 wave tiles need one LDS fragment read and
// 1.5 KB of DMA per MFMA; the vendor library's kernels for these shapes use 128x128 wave tiles (one wave
// per SIMD, 256 accumulators) and run 1.4-1.7x faster.  Same idea here, written for the blocked layout:
//   * tile 256 tokens x TN features (TN = 256, or 192 when N is only a multiple of 192), 4 waves = 2 token
//     halves x 2 feature halves, wave tile 128 x TN/2 = 4 x NT MFMA tiles (NT = TN/64): per k16 step
//     (4 + NT) fragment reads feed 4*NT MFMAs (0.5-0.58 reads/MFMA), 192-256 fp32 accumulators;
//   * stage = 32 k: X 16 KB + W TN*64 B, 4-slot ring filled by global_load_lds three stages ahead
//     (84-96 KB in flight per CU).  A blocked cell [32 rows][16 B] is 512 contiguous bytes in HBM and is
//     copied verbatim: a fragment read (32 rows x 16 B per half-wave) is one contiguous KB -> conflict-free
//     without swizzles, and every DMA lane-group reads 512 contiguous bytes (the fast TA case);
//   * one wave per SIMD means nothing else hides a wave's own latencies, so the barrier sits in the MIDDLE
//     of a stage: [read frags(s, k16=1)] [MFMAs k16=0] [wait stage s+1, barrier, DMA stage s+4 into the
//     slot just vacated, read frags(s+1, k16=0)] [MFMAs k16=1] — both fragment reads fly under MFMAs;
//   * MFMA issued swapped (A-operand = W rows): a lane owns 4 consecutive features of one token.
#include "common.hpp"
#include "kernels.hpp"
#include <type_traits>


#ifndef GEMM3_SUPER
#define GEMM3_SUPER 6                                     // row tiles per supertile (0: row-major order); configs[3] same box: 23.25 -> 23.57 k crops/s, qkv +3 %, fc1 +2 %
#endif
#ifndef GEMM3_NT
#define GEMM3_NT 3                                        // non-temporal: 1 = fp32 residual in / out (touched once), 2 = + the 16-bit outputs
#endif
namespace effocr {
namespace {

constexpr int G3RING = 4;

template <int I, int N, typename F> __device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

// JT = 32-token tiles per wave: 4 (256-token workgroup tile, the main launch) or 1 / 2 (64 / 128-token tiles
// for the leftover token tiles of the last, partially filled round of CUs — see gemm3_nt)
// FOLD: the LayerNorm between a residual producer and the next linear is folded into both (GemmArgs::stats / lnf; api.hip): for
// EPI_BIAS_RESID the kernel also writes the row as 16-bit operands + per-slice (sum, sum of squares); for the other epilogues it reads
// those and finishes the normalisation.  Compile-time: a wave-uniform runtime flag put a branch around every store group of the epilogue
// and the values held across them spilled next to the 256 accumulators.
template <typename E, int NT, int JT, int EPI, typename TO, bool FOLD = false>
__global__ __launch_bounds__(256, (JT == 2 ? 2 : 1)) void gemm3_kernel(GemmArgs g) {
  constexpr int TN = NT * 64, G3M = JT * 64;
  constexpr int XS = G3M * 64, WS = TN * 64, STAGE = XS + WS;          // bytes per 32-k stage
  constexpr int PX = XS / 1024, PW = WS / 1024, PP = (PX + PW) / 4;    // 1 KB DMA pieces: X, W, per wave
  static_assert((PX + PW) % 4 == 0, "pieces must split evenly over 4 waves");
  __shared__ __attribute__((aligned(16))) char smem[G3RING * STAGE];
  typedef typename Op16<E>::V8 V8;
  const int tid = threadIdx.x, lane = tid & 63, r31 = lane & 31, half = lane >> 5;
  const int wv = wave_id(), wm = wv & 1, wn = wv >> 1;
  const int ntn = g.N / TN;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  // tile order inside an XCD's run of consecutive ids: supertiles of GEMM3_SUPER row tiles, column-major inside — the ~32 workgroups an XCD
  // runs at a time then share GEMM3_SUPER activation tiles and 32 / GEMM3_SUPER weight tiles (row-major order: 2.7 and all 9-12 of them,
  // 5.8 MB against a 4 MB L2)
  int mt, nt;
  if (GEMM3_SUPER > 1) {
    const int mtiles = (int)gridDim.x / ntn;
    const int per = GEMM3_SUPER * ntn, grp = bid / per, rem = bid - grp * per;
    const int rows = mtiles - grp * GEMM3_SUPER < GEMM3_SUPER ? mtiles - grp * GEMM3_SUPER : GEMM3_SUPER;
    nt = rem / rows; mt = grp * GEMM3_SUPER + (rem - nt * rows);
  } else { mt = bid / ntn; nt = bid - mt * ntn; }
  const int m0 = mt * G3M, n0 = nt * TN;
  const int nst = g.K >> 5;
  const int kch = g.K >> 3;                                            // 16-byte chunks per operand row
  const int last_rb = (g.rows_alloc >> 5) - 1;                         // last addressable X row block

  // Folded LayerNorm (consumer side): the row statistics of this lane's JT tokens are requested FIRST — oldest in the in-order VM queue, so
  // they have landed when the ring's stage 0 has — and reduced to (rstd, -mean rstd) behind the first barrier.  (Loaded in the epilogue
  // they were four dependent HBM round trips per tile, one per token tile: fc1 908 -> 863 TFLOP/s.)
  constexpr bool LNF = EPI != EPI_BIAS_RESID && FOLD;
  f32x4 stv[JT][4];
  float rstd_j[JT], nmr_j[JT];
  if constexpr (LNF) {
#pragma unroll
    for (int j = 0; j < JT; ++j) {
      const int m = m0 + wm * JT * 32 + j * 32 + r31;
      const f32x4* st = reinterpret_cast<const f32x4*>(g.lnf_stats + (size_t)(m < g.M ? m : g.M - 1) * 16);
#pragma unroll
      for (int p2 = 0; p2 < 4; ++p2) stv[j][p2] = st[p2];               // slices 2 p2, 2 p2 + 1: (sum, sum of squares) each
    }
  }

  // per-lane DMA sources; piece q (1 KB = two adjacent 16-B chunk cells of one 32-row block) lands at q KB
  const char* src[PP];
#pragma unroll
  for (int i = 0; i < PP; ++i) {
    const int q = wv * PP + i;
    if (q < PX) {
      int rb = (m0 >> 5) + (q >> 1);
      rb = rb < last_rb ? rb : last_rb;                                // rows past the buffer: any valid block
      src[i] = static_cast<const char*>(g.X) + ((size_t)rb * kch + 2 * (q & 1)) * 512 + lane * 16;
    } else {
      const int qq = q - PX;
      src[i] = static_cast<const char*>(g.Wblk) + ((size_t)((n0 >> 5) + (qq >> 1)) * kch + 2 * (qq & 1)) * 512 + lane * 16;
    }
  }
  auto issue_piece = [&](int s, int i) {                   // DMA piece i of stage s (caller checks s < nst)
    char* dst = smem + (s & (G3RING - 1)) * STAGE + wv * PP * 1024;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[i] + (size_t)s * 2048),
                                     (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 0);
  };
#pragma unroll
  for (int s = 0; s < G3RING; ++s)
    if (s < nst) {
#pragma unroll
      for (int i = 0; i < PP; ++i) issue_piece(s, i);
    }
  // Steady state: the same piece as ONE inline-asm statement.  hipcc models __builtin_amdgcn_global_load_lds as an access to both
  // address spaces ("pending flat") and makes its NEXT LDS wait s_waitcnt lgkmcnt(0): a DMA piece placed between two MFMAs
  // drained the fragment reads issued around it (mlp_kernel.hpp: 70 of 73 LDS waits of the loop were full drains).  Completion
  // is counted by hand either way (vmcnt at the stage barrier).  M0 = the piece's LDS address (saved / restored: the register is
  // the compiler's), s_nop = the M0-write -> LDS-DMA wait state.
  const unsigned smem_lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  auto issue_piece_asm = [&](int s, int i) {
    const char* p = src[i] + (size_t)s * 2048;
    const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(smem_lds + (unsigned)((s & (G3RING - 1)) * STAGE) + (unsigned)((wv * PP + i) * 1024)));
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\t"
                 "s_mov_b32 m0, %2\n\t"
                 "s_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, off\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(p), "s"(dst) : "memory");
  };

  f32x16 acc[NT][JT];                                                  // [feature tile][token tile]
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < JT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int xo = (wm * JT * 4 + half) * 512 + r31 * 16;                // (row block wm*JT + j, chunk 2*c4 + half)
  const int wo = XS + (wn * NT * 4 + half) * 512 + r31 * 16;
  struct Frags { V8 w[NT]; V8 x[JT]; };
  constexpr int NF = NT + JT, NMM = NT * JT;                           // fragment reads / MFMAs per k16 step
  // fragment n of a k16 step: n < NT -> W tile n, else token tile n - NT
  auto load_one = [&](Frags& f, const char* st, int c4, auto N_) {
    constexpr int n = decltype(N_)::value;
    if constexpr (n < NT) f.w[n] = *reinterpret_cast<const V8*>(st + wo + (n * 4 + 2 * c4) * 512);
    else f.x[n - NT] = *reinterpret_cast<const V8*>(st + xo + ((n - NT) * 4 + 2 * c4) * 512);
  };
  // One k16 step: MFMA n = (feature tile n / JT, token tile n % JT); `between(n)` is issued in its shadow
  // (with one wave per SIMD, whatever sits between two MFMAs instead of under one idles the matrix pipe).
  auto mma_step = [&](const Frags& f, auto&& between) {
    static_for<0, NMM>([&](auto N_) {
      constexpr int n = decltype(N_)::value;
      constexpr int i = n / JT, j = n % JT;
      acc[i][j] = Op16<E>::mfma(f.w[i], f.x[j], acc[i][j]);
      __builtin_amdgcn_sched_barrier(0);
      between(N_);
      __builtin_amdgcn_sched_barrier(0);
    });
    // small tail tiles have fewer MFMAs than items to place: the rest goes after the last one
    constexpr int NITEM = (PP > NF ? PP : NF);
    static_for<NMM, NITEM>([&](auto N_) { between(N_); });
  };

  Frags fa, fb;
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * PP) : "memory");        // stage 0 (own pieces) ...
  __builtin_amdgcn_s_barrier();                                        // ... and everybody's
  asm volatile("" ::: "memory");
  static_for<0, NF>([&](auto N_) { load_one(fa, smem, 0, N_); });
  if constexpr (LNF) {
    const int np = g.K >> 7;                                           // 128-feature slices the producer cut the row into (<= 8)
    const float invk = 1.0f / (float)g.K;
#pragma unroll
    for (int j = 0; j < JT; ++j) {
      float S = 0.f, SS = 0.f;
#pragma unroll
      for (int p2 = 0; p2 < 4; ++p2) {
        if (2 * p2 < np) { S += stv[j][p2][0]; SS += stv[j][p2][1]; }
        if (2 * p2 + 1 < np) { S += stv[j][p2][2]; SS += stv[j][p2][3]; }
      }
      const float mean = S * invk;
      float var = SS * invk - mean * mean;
      var = var > 0.f ? var : 0.f;
      rstd_j[j] = 1.0f / sqrtf(var + g.lnf_eps);
      nmr_j[j] = -mean * rstd_j[j];
      asm volatile("" : "+v"(rstd_j[j]), "+v"(nmr_j[j]));              // pin the reduction HERE (sunk to its use in the epilogue, the 16 raw registers per token tile stayed live across the main loop and spilled)
    }
  }

  // One stage.  MORE: stage s+4 exists (DMA it), NEXT: stage s+1 exists (prefetch its fragments).  Both are
  // compile-time so that the steady state is ONE basic block: with a single wave per SIMD every scalar
  // branch between two MFMAs is a bubble in the matrix pipe (measured: 57% -> see profiles/README.md).
  auto stage = [&](int s, auto MORE, auto NEXT) {
    constexpr bool more = decltype(MORE)::value, next = decltype(NEXT)::value;
    const char* st = smem + (s & (G3RING - 1)) * STAGE;
    const char* stn = smem + ((s + 1) & (G3RING - 1)) * STAGE;
    // k16 step 0; the fragments of step 1 are read in its shadow
    mma_step(fa, [&](auto N_) {
      if constexpr (decltype(N_)::value < NF) load_one(fb, st, 1, N_);
    });
    // stage s+1 landed (own pieces; s+2, s+3 may stay in flight), every wave holds its stage-s fragments
    // in registers -> past the barrier slot s&3 is free for stage s+4
    if constexpr (more) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PP) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    // k16 step 1; in its shadow: DMA of stage s+4 and the fragments of stage s+1, step 0
    mma_step(fb, [&](auto N_) {
      constexpr int n = decltype(N_)::value;
      if constexpr (more && n < PP) issue_piece_asm(s + 4, n);
      if constexpr (next && n < NF) load_one(fa, stn, 0, N_);
    });
  };
  {
    int s = 0;
    for (; s + 4 < nst; ++s) stage(s, std::true_type{}, std::true_type{});
    for (; s + 1 < nst; ++s) stage(s, std::false_type{}, std::true_type{});
    stage(s, std::false_type{}, std::false_type{});
  }

  // ---- epilogue: lane = token (m0 + (wm*JT + j)*32 + r31), 4 consecutive features per (i, q): n = nb + 32 i + 8 q.
  // Addresses: ONE 64-bit base per token tile j and operand, everything else is a compile-time offset — a lane's features (i, q) sit in
  // fp32 chunk c4 + 8 i + 2 q and in 16-bit chunk c8 + 4 i + q (bytes 8 half .. 8 half + 7) of its row, chunks are 512 bytes apart.
  // (blk_off per (i, q) cost a 64-bit multiply-add each and, next to 256 accumulators, spilled the bias registers.)
  constexpr int CH = 16 / (int)sizeof(TO);                             // elements per 16-byte output chunk
  const int ns = n0 + wn * NT * 32;                                    // first feature of the wave's slice (a multiple of 32)
  const int nb = ns + 4 * half;
  f32x4 bv[NT][4];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int q = 0; q < 4; ++q) bv[i][q] = *reinterpret_cast<const f32x4*>(g.bias + nb + i * 32 + 8 * q);
  // LayerNorm folded into this linear (GemmArgs::lnf; wave-uniform): X holds the UN-normalised rows as 16-bit operands, W carries gamma,
  // bias carries W.beta, and the row statistics arrive as per-slice partial sums written by the producer of the rows (below):
  //     y = rstd (acc - mean s[n]) + bias[n],   s[n] = sum_k W'[n][k]
  constexpr bool lnf = EPI != EPI_BIAS_RESID && FOLD;
  constexpr bool do16 = EPI == EPI_BIAS_RESID && FOLD;                // producer: rows also as 16-bit operands ...
  constexpr bool dost = do16;                                          // ... + per-slice (sum, sum of squares) of every row
  f32x4 sv[NT][4];
  if constexpr (lnf) {
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q) sv[i][q] = *reinterpret_cast<const f32x4*>(g.lnf_s + nb + i * 32 + 8 * q);
  }
#pragma unroll
  for (int j = 0; j < JT; ++j) {
    const int m = m0 + wm * JT * 32 + j * 32 + r31;
    const bool ok = m < g.M;
    const int mr = ok ? m : g.M - 1;
    const int64_t rbi = mr >> 5;
    const int rl = (mr & 31) * 16;
    char* ob = static_cast<char*>(g.out) + (rbi * (g.N / CH) + ns / CH) * 512 + rl + (CH == 4 ? half * 512 : half * 8);
    constexpr int OI = CH == 4 ? 8 * 512 : 4 * 512, OQ = CH == 4 ? 2 * 512 : 512;   // byte steps of i and q
    float rstd = 1.f, nmr = 0.f;                                       // lnf: 1 / sqrt(var + eps), -mean * rstd of this lane's token
    if constexpr (lnf) { rstd = rstd_j[j]; nmr = nmr_j[j]; }
    if constexpr (EPI == EPI_BIAS_RESID) {
      const char* rp = reinterpret_cast<const char*>(g.resid) + (rbi * (g.N / 4) + ns / 4) * 512 + rl + half * 512;
      f32x4 rv[NT][4];
#pragma unroll
      for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          rv[i][q] = GEMM3_NT ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(rp + i * OI + q * OQ))
                              : *reinterpret_cast<const f32x4*>(rp + i * OI + q * OQ);
#pragma unroll
      for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[i][j][4 * q + e] += rv[i][q][e];
    }
    float psum = 0.f, psq = 0.f;                                       // producer: this lane's share of the slice's (sum, sum of squares)
    char* x16b = do16 ? static_cast<char*>(g.x16) + (rbi * (g.N / 8) + ns / 8) * 512 + rl + half * 8 : nullptr;
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      float v[16];
      if constexpr (lnf) {
        // as two-element vector FMAs (v_pk_fma_f32 with the row scalars duplicated): left to itself the GELU variant scalarised the 512
        // FMAs of a (i, j) tile pair — +22 % epilogue instructions, fc1 908 -> 840 TFLOP/s
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        const f32x2 r2 = {rstd, rstd}, n2 = {nmr, nmr};
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int e = 0; e < 4; e += 2) {
            const f32x2 a2 = {acc[i][j][4 * q + e], acc[i][j][4 * q + e + 1]};
            const f32x2 s2 = {sv[i][q][e], sv[i][q][e + 1]}, b2 = {bv[i][q][e], bv[i][q][e + 1]};
            const f32x2 y2 = __builtin_elementwise_fma(a2, r2, __builtin_elementwise_fma(n2, s2, b2));
            v[4 * q + e] = y2[0]; v[4 * q + e + 1] = y2[1];
          }
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int e = 0; e < 4; ++e) v[4 * q + e] = acc[i][j][4 * q + e] + bv[i][q][e];
      }
      if constexpr (EPI == EPI_BIAS_RESID) {
#pragma unroll
        for (int e = 0; e < 16; ++e) { psum += v[e]; psq = __builtin_fmaf(v[e], v[e], psq); }
      }
      if constexpr (EPI == EPI_BIAS_GELU) gelu_fold_n<E, 16>(v);         // (on the fp32 pre-activation; rounding it to 16 bits first, as the row-panel kernel's hand-over does, cost a convert + shift per value: 18 % of this epilogue's instructions)
      if (ok) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          char* p = ob + i * OI + q * OQ;
          if constexpr (sizeof(TO) == 4) {
            const f32x4 o = {v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
            if (GEMM3_NT) __builtin_nontemporal_store(o, reinterpret_cast<f32x4*>(p)); else *reinterpret_cast<f32x4*>(p) = o;
          } else {
            const u32x2 o2 = pack4<TO>(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
            if (GEMM3_NT & 2) __builtin_nontemporal_store(o2, reinterpret_cast<u32x2*>(p)); else *reinterpret_cast<u32x2*>(p) = o2;
          }
          if constexpr (EPI == EPI_BIAS_RESID) {
            if constexpr (do16)                                          // the new residual row as 16-bit operands of the next (LayerNorm-folded) linear
              *reinterpret_cast<u32x2*>(x16b + (i * 4 + q) * 512) = pack4<E>(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
          }
        }
      }
    }
    if constexpr (EPI == EPI_BIAS_RESID) {
      if constexpr (dost) {                                              // slice (column tile, wave half) of the row: both half-waves' shares, one writer
        psum += __shfl_xor(psum, 32, 64);
        psq += __shfl_xor(psq, 32, 64);
        if (ok && half == 0) {
          typedef float f32x2 __attribute__((ext_vector_type(2)));
          *reinterpret_cast<f32x2*>(g.stats + (size_t)mr * 16 + (nt * 2 + wn) * 2) = f32x2{psum, psq};
        }
      }
    }
  }
}


template <typename E, int NT, int JT>
int launch3_tile(int epi, const GemmArgs& g, hipStream_t s) {
  const int grid = ((g.M + JT * 64 - 1) / (JT * 64)) * (g.N / (NT * 64));
  const bool fold = g.lnf != 0 || g.stats != nullptr;
  if constexpr (NT != 4) {                                               // the folded LayerNorm is built for 256-wide tiles only (gemm3_nt checks)
    switch (epi) {
      case EPI_BIAS:       hipLaunchKernelGGL((gemm3_kernel<E, NT, JT, EPI_BIAS, E>), dim3(grid), dim3(256), 0, s, g); break;
      case EPI_BIAS_GELU:  hipLaunchKernelGGL((gemm3_kernel<E, NT, JT, EPI_BIAS_GELU, E>), dim3(grid), dim3(256), 0, s, g); break;
      case EPI_BIAS_RESID: hipLaunchKernelGGL((gemm3_kernel<E, NT, JT, EPI_BIAS_RESID, float>), dim3(grid), dim3(256), 0, s, g); break;
      default: return fail(EFFOCR_EINVAL, "gemm3: unknown epilogue");
    }
  } else
  switch (epi) {
    case EPI_BIAS:
      if (fold) hipLaunchKernelGGL((gemm3_kernel<E, NT, JT, EPI_BIAS, E, true>), dim3(grid), dim3(256), 0, s, g);
      else hipLaunchKernelGGL((gemm3_kernel<E, NT, JT, EPI_BIAS, E>), dim3(grid), dim3(256), 0, s, g);
      break;
    case EPI_BIAS_GELU:
      if (fold) hipLaunchKernelGGL((gemm3_kernel<E, NT, JT, EPI_BIAS_GELU, E, true>), dim3(grid), dim3(256), 0, s, g);
      else hipLaunchKernelGGL((gemm3_kernel<E, NT, JT, EPI_BIAS_GELU, E>), dim3(grid), dim3(256), 0, s, g);
      break;
    case EPI_BIAS_RESID:
      if (fold) hipLaunchKernelGGL((gemm3_kernel<E, NT, JT, EPI_BIAS_RESID, float, true>), dim3(grid), dim3(256), 0, s, g);
      else hipLaunchKernelGGL((gemm3_kernel<E, NT, JT, EPI_BIAS_RESID, float>), dim3(grid), dim3(256), 0, s, g);
      break;
    default: return fail(EFFOCR_EINVAL, "gemm3: unknown epilogue");
  }
  return check_launch("gemm3");
}

// One workgroup per CU (LDS) and ceil(M/256)*N/TN tiles rarely fill the last round of CUs.  The token tiles
// of that round are therefore run by a second launch with 64- or 128-token tiles (4x / 2x the workgroups, a
// fraction of the time each): rows [0, main_rows) with the big tile, the rest with the small one.
template <typename E, int NT>
int launch3(int epi, const GemmArgs& g, hipStream_t s) {
  const int ntn = g.N / (NT * 64), mtiles = (g.M + 255) / 256, slots = device_cus();
  const int full_rounds = (mtiles * ntn) / slots;
  int main_mt = g.no_tail_split ? mtiles : (full_rounds * slots) / ntn;   // whole token tiles inside the full rounds
  int tail_wgs = (mtiles - main_mt) * ntn;
  if (tail_wgs * 2 > slots) { main_mt = mtiles; tail_wgs = 0; }            // tail already fills most CUs
  int rc = EFFOCR_OK;
  if (main_mt > 0) {
    GemmArgs m = g;
    m.M = main_mt * 256 < g.M ? main_mt * 256 : g.M;
    if ((rc = launch3_tile<E, NT, 4>(epi, m, s))) return rc;
  }
  if (main_mt < mtiles) {
    GemmArgs t = g;
    const int64_t rb = (int64_t)main_mt * 8;                                // first row block of the tail
    const int osz = epi == EPI_BIAS_RESID ? 4 : 2;
    t.X = static_cast<const char*>(g.X) + rb * (g.K / 8) * 512;
    t.out = static_cast<char*>(g.out) + rb * (g.N * osz / 16) * 512;
    if (g.resid) t.resid = reinterpret_cast<const float*>(reinterpret_cast<const char*>(g.resid) + rb * (g.N / 4) * 512);
    if (g.stats) t.stats = g.stats + rb * 32 * 16;
    if (g.x16) t.x16 = static_cast<char*>(g.x16) + rb * (g.N / 8) * 512;
    if (g.lnf) t.lnf_stats = g.lnf_stats + rb * 32 * 16;
    t.M = g.M - main_mt * 256;
    t.rows_alloc = g.rows_alloc - main_mt * 256;
    rc = (tail_wgs * 4 <= slots) ? launch3_tile<E, NT, 1>(epi, t, s) : launch3_tile<E, NT, 2>(epi, t, s);
  }
  return rc;
}

}  // namespace

// LayerNorm folded between a gemm3 producer of D-wide rows and its consumer: the row statistics travel as <= 8 slice partials
bool gemm3_lnfold_supported(int D) { return D % 256 == 0 && D / 128 <= 8; }   // (256-wide tiles on both sides: ViT-B 768, ViT-L 1024)

bool gemm3_supported(int prec, int N, int K) {
  return (prec == PREC_BF16 || prec == PREC_FP16) && N > 0 && (N % 256 == 0 || N % 192 == 0) && K >= 128 && K % 32 == 0;
}

int gemm3_nt(int prec, int epi, const GemmArgs& g_in, hipStream_t s) {
  if (g_in.M <= 0) return EFFOCR_OK;
  if (!gemm3_supported(prec, g_in.N, g_in.K)) return fail(EFFOCR_EUNSUPPORTED, "gemm3: needs bf16/fp16, N % 192 == 0 or N % 256 == 0, K % 32 == 0, K >= 128");
  if (!g_in.Wblk || !g_in.blk_x || !g_in.blk_out) return fail(EFFOCR_EINVAL, "gemm3: operands and output must be fragment-blocked");
  GemmArgs g = g_in;
  if (g.rows_alloc <= 0) g.rows_alloc = ((g.M + 31) / 32) * 32;
  if (g.rows_alloc % 32 != 0 || g.rows_alloc < g.M) return fail(EFFOCR_EINVAL, "gemm3: rows_alloc must be a multiple of 32 covering M");
  if ((g.stats || g.x16) && (epi != EPI_BIAS_RESID || !g.stats || !g.x16 || !gemm3_lnfold_supported(g.N)))
    return fail(EFFOCR_EINVAL, "gemm3: row statistics / 16-bit row copy belong to the residual epilogue, both or neither, N in <= 8 slices");
  if (g.lnf && (epi == EPI_BIAS_RESID || !g.lnf_stats || !g.lnf_s || !gemm3_lnfold_supported(g.K) || g.N % 256 != 0))
    return fail(EFFOCR_EINVAL, "gemm3: folded LayerNorm needs the statistics, the column sums, a non-residual epilogue and K in <= 8 slices");
  const bool wide = g.N % 256 == 0;
  if (prec == PREC_BF16) return wide ? launch3<__bf16, 4>(epi, g, s) : launch3<__bf16, 3>(epi, g, s);
  return wide ? launch3<_Float16, 4>(epi, g, s) : launch3<_Float16, 3>(epi, g, s);
}

}  // namespace effocr
