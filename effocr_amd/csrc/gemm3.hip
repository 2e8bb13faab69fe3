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
//   * MFMA issued swapped (A-operand = W rows): a lane owns 4 consecutive features of one token;
//   * tiles are PERSISTENT on a CONTINUOUS ring (round 5): one workgroup per CU walks tiles b, b + G, ...; the last four stages of a tile
//     request the first four of the next one, the epilogue runs while they land, bias / column sums / row statistics come through LDS.
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

// -DGEMM3_TRACE (tools/ab_build.sh variant, never shipped): wave 0 of every workgroup records s_memrealtime (100 MHz) at the milestones
// of each of its first G3T_TILES tiles; tools/gemm3_timeline.py reads the last launch's table through effocr_debug_gemm3_stamps.
#ifdef GEMM3_TRACE
constexpr int G3T_WGS = 256, G3T_TILES = 48, G3T_N = 10;     // 0-6 milestones, 7 s_memtime at the tile top, 8 / 9 shader ticks spent in the stage vmcnt waits / barriers
__device__ unsigned long long g3_stamps[G3T_WGS * G3T_TILES * G3T_N];
#define G3_STAMP(k) tstamp[k] = __builtin_amdgcn_s_memrealtime();
#else
#define G3_STAMP(k)
#endif

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
__global__ __launch_bounds__(256, 1) void gemm3_kernel(GemmArgs g) {
  constexpr int TN = NT * 64, G3M = JT * 64;
  constexpr int XS = G3M * 64, WS = TN * 64, STAGE = XS + WS;          // bytes per 32-k stage
  constexpr int PX = XS / 1024, PW = WS / 1024, PP = (PX + PW) / 4;    // 1 KB DMA pieces: X, W, per wave
  static_assert((PX + PW) % 4 == 0, "pieces must split evenly over 4 waves");
  constexpr bool LNF = EPI != EPI_BIAS_RESID && FOLD;                  // consumer of a folded LayerNorm
  // LDS: the ring, then this tile's bias and column sums (1 KB each) and, for a LayerNorm consumer, the row statistics of its G3M tokens
  // (64 B each) — all of them DMA'd at the top of a tile and read in its epilogue: no vector-memory load sits between the main loop and
  // the first store, and nothing of it occupies registers across the main loop.
  constexpr int RINGB = G3RING * STAGE, BIAS_O = RINGB, LNS_O = RINGB + 1024, STAT_O = RINGB + 2048;
  __shared__ __attribute__((aligned(16))) char smem[RINGB + 2048 + (LNF ? G3M * 64 : 0)];
  typedef typename Op16<E>::V8 V8;
  const int tid = threadIdx.x, lane = tid & 63, r31 = lane & 31, half = lane >> 5;
  const int wv = wave_id(), wm = wv & 1, wn = wv >> 1;
  const int ntn = g.N / TN;
  const int mtiles = (g.M + G3M - 1) / G3M, ntiles = mtiles * ntn;
  const int nst = g.K >> 5;                                            // a multiple of 4, >= 8 (gemm3_supported)
  const int kch = g.K >> 3;                                            // 16-byte chunks per operand row
  const int last_rb = (g.rows_alloc >> 5) - 1;                         // last addressable X row block
  // PERSISTENT tiles on a CONTINUOUS ring.  The launch has one workgroup per CU (the ring leaves room for one); workgroup b runs tiles
  // b, b + G, b + 2G, ... and the last four stages of a tile already request the first four of the next one, so the ring never drains:
  // the epilogue of tile t runs while stages 0-3 of tile t + 1 land, its stores drain under the first stages of t + 1, and the first
  // k-step of a tile starts from a zero C operand instead of 256 zeroed accumulators.  (One tile per workgroup, measured by compiling
  // parts out: 7.8 us of turn-around per K = 768 tile — last store acknowledged, launch, zeroing, a cold ring — and a residual epilogue of
  // 27-36 us: 21 + 5 us of a 26 us qkv tile, 57 + 36 of a 97 us fc2 tile.)
  // Tile order: virtual id v = round * G + b walks the XCD's contiguous run of ids (xcd_remap), inside it supertiles of GEMM3_SUPER row
  // tiles, column-major — the ~32 workgroups an XCD runs at a time share GEMM3_SUPER activation tiles and 32 / GEMM3_SUPER weight tiles
  // (row-major order: 2.7 and all 9-12 of them, 5.8 MB against a 4 MB L2).
  auto tile_origin = [&](int v, int& m0_, int& n0_, int& nt_) {
    const int bid = xcd_remap(v, ntiles);
    int mt;
    if (GEMM3_SUPER > 1) {
      const int per = GEMM3_SUPER * ntn, grp = bid / per, rem = bid - grp * per;
      const int rows = mtiles - grp * GEMM3_SUPER < GEMM3_SUPER ? mtiles - grp * GEMM3_SUPER : GEMM3_SUPER;
      nt_ = rem / rows; mt = grp * GEMM3_SUPER + (rem - nt_ * rows);
    } else { mt = bid / ntn; nt_ = bid - mt * ntn; }
    m0_ = mt * G3M; n0_ = nt_ * TN;
  };
  int v = blockIdx.x;
  if (v >= ntiles) return;
  int m0, n0, nt;
  tile_origin(v, m0, n0, nt);

  // DMA sources: piece q (1 KB = two adjacent 16-B chunk cells of one 32-row block) lands at q KB.  A piece's address is wave-uniform
  // but for lane * 16: the bases live in SGPR pairs (the tile-to-tile set-up is scalar code, nothing of it occupies a vector register
  // next to the 256 accumulators) and ONE vector offset, lane * 16 + stage * 2048, serves all of a stage's pieces.
  const char* sb[PP];
  const unsigned lane16 = (unsigned)lane * 16u;
  auto set_src = [&](int m0_, int n0_) {
#pragma unroll
    for (int i = 0; i < PP; ++i) {
      const int q = wv * PP + i;
      if (q < PX) {
        int rb = (m0_ >> 5) + (q >> 1);
        rb = rb < last_rb ? rb : last_rb;                              // rows past the buffer: any valid block
        sb[i] = static_cast<const char*>(g.X) + ((size_t)rb * kch + 2 * (q & 1)) * 512;
      } else {
        const int qq = q - PX;
        sb[i] = static_cast<const char*>(g.Wblk) + ((size_t)((n0_ >> 5) + (qq >> 1)) * kch + 2 * (qq & 1)) * 512;
      }
    }
  };
  // One DMA piece as ONE inline-asm statement: 64 lanes x 16 B from base + voff to the LDS address in M0.  hipcc models
  // __builtin_amdgcn_global_load_lds as an access to both address spaces ("pending flat") and makes its NEXT LDS wait s_waitcnt
  // lgkmcnt(0): a DMA piece placed between two MFMAs drained the fragment reads issued around it (mlp_kernel.hpp: 70 of 73 LDS waits of
  // the loop were full drains), and any ordinary load waited for next to it gets vmcnt(0).  Completion is counted by hand (vmcnt at the
  // stage barriers).  M0 is saved / restored (the register is the compiler's), s_nop = the M0-write -> LDS-DMA wait state.
  const unsigned smem_lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  auto dma = [&](unsigned voff, const char* base, unsigned lds) {
    const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)lds);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\t"
                 "s_mov_b32 m0, %3\n\t"
                 "s_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %2\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(base), "s"(dst) : "memory");
  };
  // piece i of ring stage S (of the tile sb[] points at; voff = lane16 + (stage << 11))
  auto ring_piece = [&](unsigned voff, int S, int i) {
    dma(voff, sb[i], smem_lds + (unsigned)((S & (G3RING - 1)) * STAGE) + (unsigned)((wv * PP + i) * 1024));
  };

  f32x16 acc[NT][JT];                                                  // [feature tile][token tile]
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
  // ZERO: the first step of a tile accumulates onto a zero operand (the accumulators still hold the previous tile).
  auto mma_step = [&](const Frags& f, auto ZERO, auto&& between) {
    static_for<0, NMM>([&](auto N_) {
      constexpr int n = decltype(N_)::value;
      constexpr int i = n / JT, j = n % JT;
      if constexpr (decltype(ZERO)::value) {
        const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        acc[i][j] = Op16<E>::mfma(f.w[i], f.x[j], z);
      } else {
        acc[i][j] = Op16<E>::mfma(f.w[i], f.x[j], acc[i][j]);
      }
      __builtin_amdgcn_sched_barrier(0);
      between(N_);
      __builtin_amdgcn_sched_barrier(0);
    });
    // small tail tiles have fewer MFMAs than items to place: the rest goes after the last one
    constexpr int NITEM = (PP > NF ? PP : NF);
    static_for<NMM, NITEM>([&](auto N_) { between(N_); });
  };

  Frags fa, fb;
#ifdef GEMM3_TRACE
  unsigned long long tstamp[G3T_N] = {};
  int titer = 0;
#endif
  // One stage s of a tile; in its second half it requests ring stage s + 4 — of this tile, or (WRAP, the last four stages; sb[] has been
  // switched) stage s + 4 - nst of the next one.  WAIT: stage s + 1 is awaited by count — everything but the newest two stages' pieces
  // of the in-order VM queue has completed.  Stages 0-2 of a tile do not wait: their successors were requested before the previous
  // epilogue, which has seen them land (below), and a count would also wait for that epilogue's newest stores.  NEXT: stage s + 1
  // belongs to this tile (prefetch its fragments).  All compile-time so that the steady state is ONE basic block: with a single wave
  // per SIMD every scalar branch between two MFMAs is a bubble in the matrix pipe (measured: 57%, profiles/README.md).
  auto stage = [&](int s, auto WAIT, auto WRAP, auto NEXT, auto FIRST) {
    constexpr bool wait = decltype(WAIT)::value, wrap = decltype(WRAP)::value, next = decltype(NEXT)::value;
    const char* st = smem + (s & (G3RING - 1)) * STAGE;
    const char* stn = smem + ((s + 1) & (G3RING - 1)) * STAGE;
    // k16 step 0; the fragments of step 1 are read in its shadow
    mma_step(fa, FIRST, [&](auto N_) {
      if constexpr (decltype(N_)::value < NF) load_one(fb, st, 1, N_);
    });
    // stage s+1 landed (own pieces; s+2, s+3 may stay in flight), every wave holds its stage-s fragments
    // in registers -> past the barrier slot s&3 is free for stage s+4
#ifdef GEMM3_TRACE
    const unsigned long long w0_ = __builtin_amdgcn_s_memtime();
#endif
    if constexpr (wait) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PP) : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#ifdef GEMM3_TRACE
    const unsigned long long w1_ = __builtin_amdgcn_s_memtime();
#endif
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#ifdef GEMM3_TRACE
    const unsigned long long w2_ = __builtin_amdgcn_s_memtime();
    tstamp[8] += w1_ - w0_; tstamp[9] += w2_ - w1_;
#endif
    // k16 step 1; in its shadow: DMA of stage s+4 and the fragments of stage s+1, step 0
    const unsigned voff = lane16 + ((unsigned)(wrap ? s + 4 - nst : s + 4) << 11);
    mma_step(fb, std::false_type{}, [&](auto N_) {
      constexpr int n = decltype(N_)::value;
      if constexpr (n < PP) ring_piece(voff, s + 4, n);                // (nst % 4 == 0: the slot is (s + 4) & 3 either way)
      if constexpr (next && n < NF) load_one(fa, stn, 0, N_);
    });
  };
  constexpr std::true_type T_{};
  constexpr std::false_type F_{};

  // ---- first tile: fill the ring and see it land
  set_src(m0, n0);
#pragma unroll
  for (int S = 0; S < G3RING; ++S)
#pragma unroll
    for (int i = 0; i < PP; ++i) ring_piece(lane16 + ((unsigned)S << 11), S, i);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  for (;;) {                                                           // ---- one tile per iteration
    G3_STAMP(0)
#ifdef GEMM3_TRACE
    tstamp[7] = __builtin_amdgcn_s_memtime(); tstamp[8] = 0; tstamp[9] = 0;
#endif
    // every wave has seen its pieces of stages 0-3 land and has left the previous epilogue (bias / statistics areas are free)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    {
      const int bl = lane < TN / 4 ? lane : TN / 4 - 1;                // (TN = 192: the last lanes repeat the last 16 bytes)
      if (wv == 0) dma((unsigned)bl * 16u, reinterpret_cast<const char*>(g.bias + n0), smem_lds + BIAS_O);
      if constexpr (LNF) {
        if (wv == 1) dma((unsigned)bl * 16u, reinterpret_cast<const char*>(g.lnf_s + n0), smem_lds + LNS_O);
#pragma unroll
        for (int k = 0; k < JT; ++k) {                                 // 16 tokens x 64 B per piece
          const int p = wv * JT + k;
          int row = m0 + 16 * p;
          row = row < g.rows_alloc - 16 ? row : g.rows_alloc - 16;     // (rows past M are never stored; the buffer has rows_alloc rows)
          dma(lane16, reinterpret_cast<const char*>(g.lnf_stats + (size_t)row * 16), smem_lds + STAT_O + p * 1024);
        }
      }
    }
    G3_STAMP(1)
    static_for<0, NF>([&](auto N_) { load_one(fa, smem, 0, N_); });
    stage(0, F_, F_, T_, T_);
    stage(1, F_, F_, T_, F_);
    stage(2, F_, F_, T_, F_);
    G3_STAMP(2)
    int s = 3;
#ifdef GEMM3_TRACE
    stage(s, T_, F_, T_, F_); ++s;
    G3_STAMP(3)
#endif
    for (; s + 4 < nst; ++s) stage(s, T_, F_, T_, F_);
    // ---- the next tile (the last tile of a workgroup requests itself again and drains that before it exits: no branch here)
    const int vn = v + (int)gridDim.x;
    const bool has_next = vn < ntiles;
    int m0n, n0n, ntn_;
    tile_origin(has_next ? vn : v, m0n, n0n, ntn_);
    set_src(m0n, n0n);
    for (; s + 1 < nst; ++s) stage(s, T_, T_, T_, F_);
    stage(s, T_, T_, F_, F_);
    G3_STAMP(4)

    // ---- epilogue: lane = token (m0 + (wm*JT + j)*32 + r31), 4 consecutive features per (i, q): n = nb + 32 i + 8 q.
    // Addresses: ONE 64-bit base per token tile j and operand, everything else is a compile-time offset — a lane's features (i, q) sit in
    // fp32 chunk c4 + 8 i + 2 q and in 16-bit chunk c8 + 4 i + q (bytes 8 half .. 8 half + 7) of its row, chunks are 512 bytes apart.
    // (blk_off per (i, q) cost a 64-bit multiply-add each and, next to 256 accumulators, spilled the bias registers.)
    constexpr int CH = 16 / (int)sizeof(TO);                           // elements per 16-byte output chunk
    const int ns = n0 + wn * NT * 32;                                  // first feature of the wave's slice (a multiple of 32)
    const int nbl = (wn * NT * 32 + 4 * half) * 4;                     // byte offset of the lane's first feature in the LDS bias / sums
    f32x4 bv[NT][4];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q) bv[i][q] = *reinterpret_cast<const f32x4*>(smem + BIAS_O + nbl + (i * 32 + 8 * q) * 4);
    // LayerNorm folded into this linear (GemmArgs::lnf; wave-uniform): X holds the UN-normalised rows as 16-bit operands, W carries gamma,
    // bias carries W.beta, and the row statistics arrive as per-slice partial sums written by the producer of the rows (below):
    //     y = rstd (acc - mean s[n]) + bias[n],   s[n] = sum_k W'[n][k]
    constexpr bool lnf = LNF;
    constexpr bool do16 = EPI == EPI_BIAS_RESID && FOLD;              // producer: rows also as 16-bit operands ...
    constexpr bool dost = do16;                                        // ... + per-slice (sum, sum of squares) of every row
    f32x4 sv[NT][4];
    if constexpr (lnf) {
#pragma unroll
      for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) sv[i][q] = *reinterpret_cast<const f32x4*>(smem + LNS_O + nbl + (i * 32 + 8 * q) * 4);
    }
    // Residual epilogue: the fp32 residual tile is read in chunks c = (token tile j, feature tile i) of four 16-byte loads per lane, RD
    // chunks AHEAD and in FRONT of the stores of the chunk in hand.  (One token tile at a time — 16 loads, wait, add, 32 stores — put
    // every load behind the previous token tile's stores in the in-order VM queue: four times [store acknowledge + load latency].  resid
    // may alias out: a lane reads a value before the same lane overwrites it and nobody else touches it.)  The loads are younger than the
    // ring requests of the next tile's stages 1-3: consuming one has seen those land.
    constexpr int RD = 3, NC = JT * NT;
    f32x4 rv[RD][4];
    const char* rpj[JT];
    auto load_chunk = [&](int c) {
      const int j = c / NT, i = c % NT;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4* p = reinterpret_cast<const f32x4*>(rpj[j] + i * (8 * 512) + q * (2 * 512));
        rv[c % RD][q] = GEMM3_NT ? __builtin_nontemporal_load(p) : *p;
      }
    };
    if constexpr (EPI == EPI_BIAS_RESID) {
#pragma unroll
      for (int j = 0; j < JT; ++j) {
        const int m = m0 + wm * JT * 32 + j * 32 + r31;
        const int mr = m < g.M ? m : g.M - 1;
        rpj[j] = reinterpret_cast<const char*>(g.resid) + ((int64_t)(mr >> 5) * (g.N / 4) + ns / 4) * 512 + (mr & 31) * 16 + half * 512;
      }
#pragma unroll
      for (int c = 0; c < RD && c < NC; ++c) load_chunk(c);
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
      float rstd = 1.f, nmr = 0.f;                                     // lnf: 1 / sqrt(var + eps), -mean * rstd of this lane's token
      if constexpr (lnf) {
        const f32x4* st = reinterpret_cast<const f32x4*>(smem + STAT_O + (wm * JT * 32 + j * 32 + r31) * 64);
        const int np = g.K >> 7;                                       // 128-feature slices the producer cut the row into (<= 8)
        const float invk = 1.0f / (float)g.K;
        float S = 0.f, SS = 0.f;
#pragma unroll
        for (int p2 = 0; p2 < 4; ++p2) {                               // slices 2 p2, 2 p2 + 1: (sum, sum of squares) each
          const f32x4 t = st[p2];
          if (2 * p2 < np) { S += t[0]; SS += t[1]; }
          if (2 * p2 + 1 < np) { S += t[2]; SS += t[3]; }
        }
        const float mean = S * invk;
        float var = SS * invk - mean * mean;
        var = var > 0.f ? var : 0.f;
        rstd = 1.0f / sqrtf(var + g.lnf_eps);
        nmr = -mean * rstd;
      }
      float psum = 0.f, psq = 0.f;                                     // producer: this lane's share of the slice's (sum, sum of squares)
      char* x16b = do16 ? static_cast<char*>(g.x16) + (rbi * (g.N / 8) + ns / 8) * 512 + rl + half * 8 : nullptr;
#pragma unroll
      for (int i = 0; i < NT; ++i) {
        float v[16];
        if constexpr (EPI == EPI_BIAS_RESID) {
          const int c = j * NT + i;
#pragma unroll
          for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][j][4 * q + e] += rv[c % RD][q][e];
          if (c + RD < NC) load_chunk(c + RD);
        }
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
        if constexpr (EPI == EPI_BIAS_GELU) gelu_fold_n<E, 16>(v);       // (on the fp32 pre-activation; rounding it to 16 bits first, as the row-panel kernel's hand-over does, cost a convert + shift per value: 18 % of this epilogue's instructions)
        if constexpr (EPI != EPI_BIAS_RESID) {
          // no load of this epilogue is younger than the next tile's ring requests: see them land (own pieces) before the first store
          // joins the queue — from here to stage 3 of the next tile nothing is awaited by count
          if (j == 0 && i == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
#ifdef GEMM3_TRACE
        if (j == 0 && i == 0) { asm volatile("" :: "v"(v[0]), "v"(v[15])); G3_STAMP(5) }
#endif
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
              if constexpr (do16)                                        // the new residual row as 16-bit operands of the next (LayerNorm-folded) linear
                *reinterpret_cast<u32x2*>(x16b + (i * 4 + q) * 512) = pack4<E>(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
            }
          }
        }
      }
      if constexpr (EPI == EPI_BIAS_RESID) {
        if constexpr (dost) {                                            // slice (column tile, wave half) of the row: both half-waves' shares, one writer
          psum += __shfl_xor(psum, 32, 64);
          psq += __shfl_xor(psq, 32, 64);
          if (ok && half == 0) {
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            *reinterpret_cast<f32x2*>(g.stats + (size_t)mr * 16 + (nt * 2 + wn) * 2) = f32x2{psum, psq};
          }
        }
      }
    }
#ifdef GEMM3_TRACE
    G3_STAMP(6)
    if (wv == 0 && blockIdx.x < G3T_WGS && titer < G3T_TILES) {
#pragma unroll
      for (int k = 0; k < G3T_N; ++k) g3_stamps[((size_t)blockIdx.x * G3T_TILES + titer) * G3T_N + k] = tstamp[k];
    }
    ++titer;
#endif
    if (!has_next) break;
    v = vn; m0 = m0n; n0 = n0n; nt = ntn_;
  }                                                                    // tiles
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                     // the self-request of the last tile: nothing may land in LDS after the workgroup has gone
}


template <typename E, int NT, int JT>
int launch3_tile(int epi, const GemmArgs& g, hipStream_t s) {
  const int tiles = ((g.M + JT * 64 - 1) / (JT * 64)) * (g.N / (NT * 64)), slots = device_cus();
  const int grid = tiles < slots ? tiles : slots;                        // persistent: one workgroup per CU walks tiles b, b + grid, ...
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
  return (prec == PREC_BF16 || prec == PREC_FP16) && N > 0 && (N % 256 == 0 || N % 192 == 0) && K >= 256 && K % 128 == 0;
}

int gemm3_nt(int prec, int epi, const GemmArgs& g_in, hipStream_t s) {
  if (g_in.M <= 0) return EFFOCR_OK;
  if (!gemm3_supported(prec, g_in.N, g_in.K)) return fail(EFFOCR_EUNSUPPORTED, "gemm3: needs bf16/fp16, N % 192 == 0 or N % 256 == 0, K % 128 == 0, K >= 256");
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

#ifdef GEMM3_TRACE
extern "C" int effocr_debug_gemm3_stamps(unsigned long long* out, int n) {
  hipDeviceSynchronize();
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(effocr::g3_stamps), (size_t)n * sizeof(unsigned long long));
}
#endif
