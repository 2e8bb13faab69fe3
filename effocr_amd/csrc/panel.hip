// Row-panel NT GEMM for the K = embed-dim linears of the ViT encoder (qkv, proj, fc1) on gfx950:
//     out[m][n] = epilogue( sum_k A[m][k] * W[n][k] + bias[n] ),  A = x  or  A = LayerNorm(x)
//
// Why a second GEMM structure: with K = 384 a 128x128 tile has only 6 K-stages, and the K-streaming
// kernel (gemm.hip) pays one full memory latency per stage plus a cold prologue and an epilogue per
// tile (measured: 13.6 us per workgroup for 2.6 us of MFMA work).  Here a workgroup owns 128 token
// rows for the WHOLE output width:
//   * the A panel [128 x K] (bf16/f16, 96 KB at K=384) is loaded ONCE, in a single burst, and stays
//     in LDS; because the workgroup sees complete rows, the pre-norm LayerNorm of timm's Block
//     (norm1 before attn.qkv, norm2 before mlp.fc1) is fused into that load: fp32 residual stream in,
//     fp32 statistics, normalised 16-bit operands straight into LDS — the LayerNorm kernel, its
//     output buffer and one launch per linear disappear;
//   * W (L2-resident, <= 1.2 MB) streams through a 3-deep LDS ring of [128 n x 64 k] stages (full
//     128-byte rows) filled by global_load_lds (16 B/lane, no VGPR round trip) issued two stages ahead;
//     waits are counted (s_waitcnt vmcnt(N), N exact in the presence of the epilogue stores that
//     share the in-order VM counter) and the workgroup barrier is the raw s_barrier so the DMA stays
//     in flight across it.  The ring image is lane-linear, so the bank-conflict swizzle
//     (16-B chunk ^= (row>>1)&7) is applied to the per-lane SOURCE address and again on the read;
//   * 8 waves = 2 (token halves) x 4 (32-feature slices) sweep the output in 128-column steps (every
//     layer width is a multiple of 128: no partial steps).  A wave tile is 64 tokens x 32 features:
//     32 accumulator registers — small on purpose, rocprof/s_memtime showed the 64x64 variant living
//     on the register cliff (spills share the VM counter with the W DMA and drain it every stage);
//   * the MFMA is issued swapped (A-operand = W rows) like gemm.hip: a lane owns 4 consecutive
//     features of one token;
//   * the epilogue of sweep step t is DEFERRED into the stages of step t+1: at the step boundary the
//     accumulators (+bias) are parked as packed 16-bit values, and every following stage stores — or,
//     for mlp.fc1, unpacks/GELUs/packs/stores — two parked groups, hand-sliced between that stage's 8
//     MFMAs so that the VALU work issues in the MFMA shadow (s_memtime timeline before: stage body
//     1400-1700 cycles with the epilogue behind the MFMAs, 500-900 without).
// LDS: 128 x (2K+16) + 3 x 16 KB + 4K x 4 (bias) = 152 KB at K = 384 -> one workgroup (2 waves/SIMD) per CU.
#include "common.hpp"
#include "kernels.hpp"
#include "ln.hpp"
#include <type_traits>


namespace effocr {
namespace {

constexpr int PNT = 128;            // output columns per sweep step
constexpr int RING = 3;

// Two geometries, same code:
//   BMT = 128: 8 waves (2 token halves x 4 feature slices), ring stage [128 n x 64 k] = 16 KB (8 MFMAs per
//              wave per barrier), 152 KB LDS -> ONE workgroup per CU;
//   BMT =  64: 4 waves (4 feature slices), ring stage [128 n x 32 k] = 8 KB (4 MFMAs per wave per barrier),
//              79 KB LDS -> TWO independent workgroups per CU: one workgroup's panel load / LayerNorm /
//              barrier and DMA waits overlap the other's MFMAs, at the price of streaming W twice.
template <int BMT> struct Geo {
  static constexpr int WAVES = BMT / 16;               // 16 panel rows per wave in the prologue
  static constexpr int THREADS = WAVES * 64;
  static constexpr int KSTEPS = (BMT == 128) ? 4 : 2;  // k16 steps per ring stage
  static constexpr int ROWB = KSTEPS * 32;             // bytes per W row per stage
  static constexpr int CHUNKS = ROWB / 16;
  static constexpr int WSTAGE = PNT * ROWB;
  static constexpr int LPR = CHUNKS;                   // lanes per W row in a DMA piece
};

template <typename TO> __device__ __forceinline__ void pstore4(TO* p, float a, float b, float c, float d) {
  if constexpr (sizeof(TO) == 4) {
    f32x4 v = {a, b, c, d};
    *reinterpret_cast<f32x4*>(p) = v;
  } else {
    *reinterpret_cast<u32x2*>(p) = pack4<TO>(a, b, c, d);
  }
}

template <int KD> struct LnShape;                       // LayerNorm lane layout for a row of KD floats
template <> struct LnShape<384> { static constexpr int G = 32, V = 3; };
template <> struct LnShape<128> { static constexpr int G = 32, V = 1; };

template <int I, int N, typename F> __device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

// FULL: every output / residual row index of a panel is addressable (M % 128 == 0, or the buffers
// carry padding rows up to the next multiple of 128) -> branch-free stores.
template <typename E, int KD, int PRO, int EPI, typename TO, bool FULL, int BMT>
__global__ __launch_bounds__(Geo<BMT>::THREADS, 2) void panel_gemm_kernel(PanelArgs a) {
  typedef Geo<BMT> GEO;
  constexpr int PBM = BMT, WSTAGE = GEO::WSTAGE, NT = GEO::THREADS, KSTEPS = GEO::KSTEPS, NM = 2 * KSTEPS;
  constexpr int APITCH = KD * 2 + 16;                    // bytes per A-panel row (+16: conflict-free b128 reads)
  constexpr int NKS = KD / (16 * KSTEPS);                // ring stages per 128-column sweep step
  constexpr int NMAX = 4 * KD;                           // widest layer on this path: mlp.fc1
  constexpr int NG = 8;                                  // epilogue groups per wave per step (2 token tiles x 4)
  __shared__ __attribute__((aligned(16))) char smem[PBM * APITCH + RING * WSTAGE + NMAX * 4];
  char* sA = smem;
  char* sW = smem + PBM * APITCH;
  float* sBias = reinterpret_cast<float*>(smem + PBM * APITCH + RING * WSTAGE);   // bias via LDS: epilogue
                                                         // reads must not share the VM counter with the W DMA
  typedef typename Op16<E>::V8 V8;

  const int tid = threadIdx.x, lane = tid & 63, r31 = lane & 31, half = lane >> 5;
  const int wv = wave_id();
  const int wn = (BMT == 128) ? (wv >> 1) : wv, wm = (BMT == 128) ? (wv & 1) : 0;
  // Tail split: 1 workgroup/CU and ceil(M/PBM) panels rarely fill the last round of CUs, so the panels of
  // that round (index >= tail_first) are cut along N into tail_split workgroups of ~niter/tail_split sweep
  // steps each (each repeats the cheap prologue); they have the highest block ids = dispatched last.
  int panel = blockIdx.x, niter = a.N / PNT, it_lo = 0;
  if (a.tail_split > 1 && panel >= a.tail_first) {
    const int t = panel - a.tail_first;
    panel = a.tail_first + t / a.tail_split;
    const int part = t - (t / a.tail_split) * a.tail_split;
    it_lo = part * niter / a.tail_split;                // uneven parts are fine: [part*niter/split, (part+1)*niter/split)
    niter = (part + 1) * niter / a.tail_split - it_lo;
  }
  const int m0 = panel * PBM;
  const int nlo = it_lo * PNT;                           // first output column of this workgroup
  const int S = niter * NKS;
  const char* Wb = static_cast<const char*>(a.W) + (size_t)nlo * KD * 2;

  // ---- W ring fill: stage s = (sweep step it, k-stage ks); 2 x 1 KB DMA pieces per wave.  Lane-linear
  // LDS image -> the bank swizzle goes on the SOURCE address: 128-B rows: chunk ^= (row>>1)&7, 64-B rows:
  // chunk ^= (row>>2)&3 (both make the 16 rows of a ds_read_b128 lane group hit 16 distinct 16-B slots)
  uint32_t wsrc[2];                                      // per-lane source offset inside a stage's W block
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int p = (wv * 2 + i) * 64 + lane;
    const int row = p / GEO::LPR, ch = p % GEO::LPR;
    const int swz = (GEO::CHUNKS == 8) ? ((row >> 1) & 7) : ((row >> 2) & 3);
    wsrc[i] = (uint32_t)(row * KD * 2 + ((ch ^ swz) * 16));
  }
  auto issue_w = [&](int s, int slot) {
    if (s >= S) return;
    const int it = s / NKS, ks = s - it * NKS;
    const char* src = Wb + ((size_t)it * PNT * KD + ks * 16 * KSTEPS) * 2;
    char* dst = sW + slot * WSTAGE;
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + wsrc[i]),
                                       (__attribute__((address_space(3))) void*)(dst + (wv * 2 + i) * 1024), 16, 0, 0);
  };
  issue_w(0, 0);
  issue_w(1, 1);
  for (int n = tid; n < niter * PNT; n += NT) sBias[n] = a.bias[nlo + n];

  // ---- A panel
  if constexpr (PRO == PRO_LN) {
    if (a.blk_a) {
      // fragment-blocked fp32 x: lane = (token tl of this wave's 16 rows, quarter `part` of the row): the 16
      // token-lanes of a quarter read 256 contiguous bytes of one cell; statistics = lane-local sums over the
      // quarter + a 4-lane-group exchange; exact two-pass variance from the registers.
      constexpr int NQ = KD / 16;                        // float4 chunks per lane (KD/4 chunks / 4 parts)
      const int tl = lane & 15, part = lane >> 4;
      const int row = wv * 16 + tl;
      int m = m0 + row;
      m = m < a.M ? m : a.M - 1;
      const char* xb = static_cast<const char*>(a.A);
      f32x4 xv[NQ];
#pragma unroll
      for (int i = 0; i < NQ; ++i) xv[i] = *reinterpret_cast<const f32x4*>(xb + blk_off(m, part + 4 * i, KD / 4));
      float sum = 0.f;
#pragma unroll
      for (int i = 0; i < NQ; ++i) sum += (xv[i][0] + xv[i][1]) + (xv[i][2] + xv[i][3]);
      sum += __shfl_xor(sum, 16, 64);
      sum += __shfl_xor(sum, 32, 64);
      const float mean = sum * (1.0f / KD);
      float ss = 0.f;
#pragma unroll
      for (int i = 0; i < NQ; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d = xv[i][e] - mean; ss += d * d; }
      ss += __shfl_xor(ss, 16, 64);
      ss += __shfl_xor(ss, 32, 64);
      const float rstd = 1.0f / sqrtf(ss * (1.0f / KD) + a.eps);
#pragma unroll
      for (int i = 0; i < NQ; ++i) {
        const int c = part + 4 * i;
        const f32x4 gm = *reinterpret_cast<const f32x4*>(a.gamma + c * 4);
        const f32x4 bt = *reinterpret_cast<const f32x4*>(a.beta + c * 4);
        *reinterpret_cast<u32x2*>(sA + row * APITCH + c * 8) =
            pack4<E>((xv[i][0] - mean) * rstd * gm[0] + bt[0], (xv[i][1] - mean) * rstd * gm[1] + bt[1],
                     (xv[i][2] - mean) * rstd * gm[2] + bt[2], (xv[i][3] - mean) * rstd * gm[3] + bt[3]);
      }
    } else {
    constexpr int G = LnShape<KD>::G, V = LnShape<KD>::V, RPW = 64 / G;
    const float* X = static_cast<const float*>(a.A);
    const int sub = lane % G;
    constexpr int NP = 16 / RPW;                         // passes: this wave's 16 panel rows
    f32x4 xv[NP][V];
#pragma unroll
    for (int pass = 0; pass < NP; ++pass) {              // every load of the wave's 16 rows in flight at once
      int m = m0 + wv * 16 + pass * RPW + lane / G;
      m = m < a.M ? m : a.M - 1;
#pragma unroll
      for (int i = 0; i < V; ++i) xv[pass][i] = *reinterpret_cast<const f32x4*>(X + (int64_t)m * KD + (sub + G * i) * 4);
    }
#pragma unroll
    for (int pass = 0; pass < NP; ++pass) {
      const int row = wv * 16 + pass * RPW + lane / G;
      f32x4 y[V];
      ln_apply<G, V>(xv[pass], sub, a.gamma, a.beta, a.eps, y);
#pragma unroll
      for (int i = 0; i < V; ++i)
        *reinterpret_cast<u32x2*>(sA + row * APITCH + (sub + G * i) * 8) = pack4<E>(y[i][0], y[i][1], y[i][2], y[i][3]);
    }
    }
  } else {
    const E* X = static_cast<const E*>(a.A);
    constexpr int CPR = KD / 8;                          // 16-B chunks per row
    constexpr int NCH = PBM * CPR / NT;
    u32x4 v[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int id = tid + NT * i;
      int row = id / CPR, c = id - row * CPR;
      if (a.blk_a) { c = id / PBM; row = id - c * PBM; } // blocked A: consecutive threads = consecutive tokens of one cell
      int m = m0 + row;
      m = m < a.M ? m : a.M - 1;
      const char* p = a.blk_a ? reinterpret_cast<const char*>(X) + blk_off(m, c, CPR)
                              : reinterpret_cast<const char*>(X + (int64_t)m * a.lda + c * 8);
      v[i] = *reinterpret_cast<const u32x4*>(p);
    }
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const int id = tid + NT * i;
      int row = id / CPR, c = id - row * CPR;
      if (a.blk_a) { c = id / PBM; row = id - c * PBM; }
      *reinterpret_cast<u32x4*>(sA + row * APITCH + c * 16) = v[i];
    }
  }
  __syncthreads();                                       // panel visible; also drains stages 0,1 (prologue only)
  issue_w(2, 2);

  constexpr bool DEFER = (EPI != EPI_BIAS_RESID);        // residual loads would stall the W stream: see below
  constexpr int EPG = (NKS >= 2 * NG) ? 1 : ((NG + NKS - 1) / NKS < 2 ? 2 : (NG + NKS - 1) / NKS);   // deferred groups per stage
  static_assert(NKS * EPG >= NG && NKS % 2 == 0, "panel geometry");
  f32x16 acc[2];                                         // [token tile]: 32 features x 32 tokens each
  u32x2 donep[NG];                                       // parked step: bias added, packed to E (group = 4 values)
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
#pragma unroll
  for (int g = 0; g < NG; ++g) { donep[g][0] = 0u; donep[g][1] = 0u; }

  const int sw = (GEO::CHUNKS == 8) ? ((r31 >> 1) & 7) : ((r31 >> 2) & 3);
  const char* pa = sA + (wm * 64 + r31) * APITCH + half * 16;
  const int wrow = (wn * 32 + r31) * GEO::ROWB;
  TO* out = static_cast<TO*>(a.out);
  int mrow[2];
  bool mok[2];
  uint32_t prow[2];                                      // byte offset of this lane's two output rows at its first column
#pragma unroll                                           // (uniform base + 32-bit offset: launcher guarantees < 4 GB)
  for (int j = 0; j < 2; ++j) {
    const int m = m0 + wm * 64 + j * 32 + r31;
    mok[j] = FULL || m < a.M;
    mrow[j] = mok[j] ? m : a.M - 1;
    prow[j] = (uint32_t)(((int64_t)mrow[j] * a.ldo + wn * 32 + 4 * half) * (int64_t)sizeof(TO));
    if (a.blk_out) {
      // blocked: byte offset of (row, column wn*32 + 4*half); a column step of c elements moves (c / CH) cells
      constexpr int CH = 16 / (int)sizeof(TO);
      prow[j] = (uint32_t)(blk_off(mrow[j], (wn * 32 + 4 * half) / CH, (int)(a.ldo / CH)) + ((4 * half) % CH) * (int)sizeof(TO));
    }
  }
  const uint32_t colstep = a.blk_out ? 512u / (16 / (uint32_t)sizeof(TO)) * 1u : (uint32_t)sizeof(TO);   // bytes per column element step (in units of 1 element, for multiples of the chunk)
  prow[0] += (uint32_t)nlo * colstep;
  prow[1] += (uint32_t)nlo * colstep;

  // MFMA operand fragments.  W rows are the A-operand, tokens the B-operand.  W fragments of a whole
  // stage (4 k16 steps) are fetched one stage ahead (that frees the ring slot at the next barrier);
  // the token fragments come from the resident panel and are pipelined one k16 step ahead.
  struct WFrags { V8 w[KSTEPS]; };
  auto load_w = [&](WFrags& f, int slot) {
    const char* pw = sW + slot * WSTAGE + wrow;
#pragma unroll
    for (int c4 = 0; c4 < KSTEPS; ++c4) f.w[c4] = *reinterpret_cast<const V8*>(pw + (((2 * c4 + half) ^ sw) * 16));
  };
  auto load_x = [&](V8 (&x)[2], int k16) {               // k16 = k16-step index within K
#pragma unroll
    for (int j = 0; j < 2; ++j) x[j] = *reinterpret_cast<const V8*>(pa + j * 32 * APITCH + k16 * 32);
  };

  // group g = (token tile j = g>>2, feature quad q = g&3): 4 consecutive features of one token
  auto group_ptr = [&](int g, int n0) -> TO* {
    const int j = g >> 2, q = g & 3;
    // columns advance in multiples of 8 elements here: row-major 8*sizeof bytes; blocked (8/CH) cells of 512 B
    return reinterpret_cast<TO*>(reinterpret_cast<char*>(out) + (prow[j] + (uint32_t)(n0 + 8 * q) * colstep));
  };
  auto store_group = [&](int g, int n0, u32x2 v) {
    TO* p = group_ptr(g, n0);
    if constexpr (FULL) *reinterpret_cast<u32x2*>(p) = v;
    else if (mok[g >> 2]) *reinterpret_cast<u32x2*>(p) = v;
  };
  // immediate epilogue straight from the fp32 accumulators (residual path, and the last sweep step)
  auto epi_now = [&](int n0) {
    float v[NG * 4];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const int j = g >> 2, q = g & 3;
      const f32x4 bv = *reinterpret_cast<const f32x4*>(sBias + n0 + wn * 32 + 8 * q + 4 * half);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[g * 4 + e] = acc[j][4 * q + e] + bv[e];
    }
    if constexpr (EPI == EPI_BIAS_GELU) {
      // same argument rounding as the parked path below: every column's result is then independent of
      // which sweep step was the workgroup's last (tail split, batch size) -> bitwise batch invariance
#pragma unroll
      for (int e = 0; e < NG * 4; ++e) v[e] = (float)(E)v[e];
      gelu_erf_fast_n<NG * 4>(v);
    }
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      TO* p = group_ptr(g, n0);
      if constexpr (FULL) pstore4<TO>(p, v[g * 4], v[g * 4 + 1], v[g * 4 + 2], v[g * 4 + 3]);
      else if (mok[g >> 2]) pstore4<TO>(p, v[g * 4], v[g * 4 + 1], v[g * 4 + 2], v[g * 4 + 3]);
    }
  };
  // sweep-step boundary: acc + bias -> packed E (half the registers of an fp32 copy), acc = 0.
  // (mlp.fc1: GELU therefore sees its argument rounded to the operand type first; its result is
  // rounded to the same type anyway, so the extra error stays below one output ulp.)
  auto park = [&](int n0) {
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const int j = g >> 2, q = g & 3;
      const f32x4 bv = *reinterpret_cast<const f32x4*>(sBias + n0 + wn * 32 + 8 * q + 4 * half);
      donep[g] = pack4<E>(acc[j][4 * q] + bv[0], acc[j][4 * q + 1] + bv[1], acc[j][4 * q + 2] + bv[2], acc[j][4 * q + 3] + bv[3]);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  };

  WFrags wa, wb;
  V8 x0[2], x1[2];
  load_w(wa, 0);
  load_x(x0, 0);
  int s = 0, slot = 0;                                   // ring slot of stage s
  int pend_n0 = 0;

  // deferred-epilogue stores issued at stage position k of a steady-state sweep step (FULL only)
  auto stores_at = [](int k) constexpr { k = (k % NKS + NKS) % NKS; const int lo = k * EPG; return lo >= NG ? 0 : (lo + EPG > NG ? NG - lo : EPG); };

  // One ring stage.  Top: stage s+1 has landed for this wave's own DMA pieces (stage s+2 may stay in
  // flight — the VM counter is in order and also counts the deferred stores issued since, hence the
  // exact per-position count); past the barrier for everybody's, and every wave has finished reading
  // slot(s) (its W fragments were fetched a stage ago) so that slot takes stage s+3.
  auto stage = [&](auto KS, auto WITH_EPI, bool steady) {
    constexpr int ks = decltype(KS)::value;
    constexpr bool with_epi = decltype(WITH_EPI)::value;
    constexpr int extra = (FULL && DEFER && with_epi) ? stores_at(ks - 1) + stores_at(ks - 2) : 0;
    if (s + 2 >= S) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (extra > 0 && steady) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 + extra) : "memory");
    else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    issue_w(s + 3, slot);
    const int nslot = (slot + 1 == RING) ? 0 : slot + 1;
    WFrags& cur = (ks & 1) ? wb : wa;
    WFrags& nxt = (ks & 1) ? wa : wb;
    load_w(nxt, nslot);                                  // stage s+1 (garbage after the last stage: never consumed)
    // MFMA n of the stage = (k16 step c4 = n>>1, token tile j = n&1); token fragments ping-pong between
    // x0 (even steps) and x1 (odd steps), each fetched while the other one is being consumed
    auto mma1 = [&](int n) {
      const int c4 = n >> 1, j = n & 1;
      if (j == 0) {
        const int nk = (c4 == KSTEPS - 1) ? ((ks + 1) % NKS) * KSTEPS : ks * KSTEPS + c4 + 1;   // next k16 step
        if (c4 & 1) load_x(x0, nk); else load_x(x1, nk);
      }
      acc[j] = Op16<E>::mfma(cur.w[c4], (c4 & 1) ? x1[j] : x0[j], acc[j]);
    };
    constexpr int g0 = ks * EPG;
    constexpr int ng = (g0 >= NG) ? 0 : (g0 + EPG > NG ? NG - g0 : EPG);
    if constexpr (DEFER && with_epi && ng > 0 && EPI == EPI_BIAS_GELU) {
      // Deferred GELU of `ng` parked groups (NE elements), hand-sliced (10 slices) between the stage's NM
      // MFMAs: every slice is a handful of independent (packed) VALU ops that issue in the 32-cycle
      // shadow of the MFMA in front of it.  sched_barrier(0) pins the interleave (emits nothing).
      constexpr int NE = ng * 4;
      typedef __attribute__((__vector_size__(4 * sizeof(E)))) E E4;
      constexpr float c[7] = {4.1060451e-05f, -0.00051103633f, 0.00423542528f, -0.0251028568f, 0.111079332f, -0.375314877f, 1.12826843f};
      float x[NE], z[NE], t[NE], p[NE];
      auto slice = [&](auto I) {
        constexpr int i = decltype(I)::value;
        if constexpr (i == 0) {
#pragma unroll
          for (int gg = 0; gg < ng; ++gg) {
            const E4 h = __builtin_bit_cast(E4, donep[g0 + gg]);
#pragma unroll
            for (int e = 0; e < 4; ++e) x[gg * 4 + e] = (float)h[e];
          }
#pragma unroll
          for (int e = 0; e < NE; ++e) z[e] = __builtin_amdgcn_fmed3f(x[e] * 0.70710678118654752440f, -3.0f, 3.0f);
        } else if constexpr (i == 1) {
#pragma unroll
          for (int e = 0; e < NE; ++e) { t[e] = z[e] * z[e]; p[e] = fmaf(4.07419588e-08f, t[e], -1.94481757e-06f); }
        } else if constexpr (i <= 8) {
#pragma unroll
          for (int e = 0; e < NE; ++e) p[e] = fmaf(p[e], t[e], c[i - 2]);
        } else {
#pragma unroll
          for (int e = 0; e < NE; ++e) { const float h = 0.5f * x[e]; x[e] = fmaf(z[e] * p[e], h, h); }
#pragma unroll
          for (int gg = 0; gg < ng; ++gg) store_group(g0 + gg, pend_n0, pack4<E>(x[gg * 4], x[gg * 4 + 1], x[gg * 4 + 2], x[gg * 4 + 3]));
        }
      };
      static_for<0, NM>([&](auto N_) {
        constexpr int n = decltype(N_)::value;
        mma1(n);
        __builtin_amdgcn_sched_barrier(0);
        static_for<(n * 10) / NM, ((n + 1) * 10) / NM>([&](auto I) { slice(I); });
        __builtin_amdgcn_sched_barrier(0);
      });
    } else {
#pragma unroll
      for (int n = 0; n < NM; ++n) mma1(n);
      if constexpr (DEFER && with_epi && ng > 0) {
#pragma unroll
        for (int gg = 0; gg < ng; ++gg) store_group(g0 + gg, pend_n0, donep[g0 + gg]);
      }
    }
    ++s;
    slot = nslot;
  };

  for (int it = 0; it < niter; ++it) {
    const int n0 = it * PNT + wn * 32;                   // first column of this wave's 32-feature slice
    if (it == 0) static_for<0, NKS>([&](auto KS) { stage(KS, std::false_type{}, false); });
    else static_for<0, NKS>([&](auto KS) { stage(KS, std::true_type{}, it >= 2); });
    if constexpr (DEFER) {
      if (it == niter - 1) {
        epi_now(it * PNT);
      } else {
        park(it * PNT);
        pend_n0 = it * PNT;
      }
    } else {
      // residual epilogue (proj): global loads inside the pipelined loop would share the in-order VM
      // counter with the W DMA, so it stays a plain load-all / store-all block at the step boundary
      f32x4 rv[2][4];
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const char* rp = a.blk_out ? reinterpret_cast<const char*>(a.resid) + blk_off(mrow[j], (nlo + n0 + 8 * q + 4 * half) >> 2, (int)(a.ldr >> 2))
                                     : reinterpret_cast<const char*>(a.resid + (int64_t)mrow[j] * a.ldr + nlo + n0 + 8 * q + 4 * half);
          rv[j][q] = *reinterpret_cast<const f32x4*>(rp);
        }
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[j][4 * q + e] += rv[j][q][e];
      epi_now(it * PNT);
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    }
  }
}


template <typename E, int KD, bool FULL, int BMT>
int launch_panel(int pro, int epi, const PanelArgs& a_in, hipStream_t s) {
  PanelArgs a = a_in;
  const int npanels = (a.M + BMT - 1) / BMT, niter = a.N / PNT;
  const int slots = device_cus() * (BMT == 128 ? 1 : 2);    // resident workgroups (LDS-limited)
  const int tail = npanels % slots;
  a.tail_first = npanels;
  a.tail_split = 1;
  if (!a.no_tail_split && tail > 0) {
    int split = slots / tail;                            // as many parts as fit into one round of CUs (>= 1 sweep step each)
    if (split > niter) split = niter;
    if (split < 1) split = 1;
    a.tail_first = npanels - tail;
    a.tail_split = split;
  }
  const dim3 grid((unsigned)(a.tail_first + (npanels - a.tail_first) * a.tail_split)), blk(Geo<BMT>::THREADS);
#define EFFOCR_PANEL(P, EP, TOUT) hipLaunchKernelGGL((panel_gemm_kernel<E, KD, P, EP, TOUT, FULL, BMT>), grid, blk, 0, s, a)
  if (pro == PRO_LN) {
    switch (epi) {
      case EPI_BIAS: EFFOCR_PANEL(PRO_LN, EPI_BIAS, E); break;
      case EPI_BIAS_GELU: EFFOCR_PANEL(PRO_LN, EPI_BIAS_GELU, E); break;
      case EPI_BIAS_RESID: EFFOCR_PANEL(PRO_LN, EPI_BIAS_RESID, float); break;
      default: return fail(EFFOCR_EINVAL, "panel_gemm: unknown epilogue");
    }
  } else {
    switch (epi) {
      case EPI_BIAS: EFFOCR_PANEL(PRO_COPY, EPI_BIAS, E); break;
      case EPI_BIAS_GELU: EFFOCR_PANEL(PRO_COPY, EPI_BIAS_GELU, E); break;
      case EPI_BIAS_RESID: EFFOCR_PANEL(PRO_COPY, EPI_BIAS_RESID, float); break;
      default: return fail(EFFOCR_EINVAL, "panel_gemm: unknown epilogue");
    }
  }
#undef EFFOCR_PANEL
  return check_launch("panel_gemm");
}

template <typename E, int KD>
int launch_panel_full(int pro, int epi, const PanelArgs& a, hipStream_t s) {
  const bool full = (a.M % 128 == 0) || a.rows_padded;
  if (a.panel_rows == 64)
    return full ? launch_panel<E, KD, true, 64>(pro, epi, a, s) : launch_panel<E, KD, false, 64>(pro, epi, a, s);
  return full ? launch_panel<E, KD, true, 128>(pro, epi, a, s) : launch_panel<E, KD, false, 128>(pro, epi, a, s);
}

}  // namespace

// (panel_gemm_supported — the width predicate the forward also uses to pick the ViT-S class of kernels — lives in mlp.hip, which
// is part of BOTH builds; this translation unit is linked into the A/B build only: make AB=1)
int panel_gemm(int prec, int pro, int epi, const PanelArgs& a, hipStream_t s) {
  if (a.M <= 0) return EFFOCR_OK;
  if (!panel_gemm_supported(prec, a.N, a.K)) return fail(EFFOCR_EUNSUPPORTED, "panel_gemm: needs bf16/fp16, K in {128,384}, N % 128 == 0, N <= 4K");
  if (pro == PRO_COPY && (a.lda % 8) != 0) return fail(EFFOCR_EINVAL, "panel_gemm: A rows must be 16-byte aligned");
  if ((int64_t)(a.M + 128) * a.ldo * 4 >= ((int64_t)1 << 32)) return fail(EFFOCR_EUNSUPPORTED, "panel_gemm: output larger than 4 GB");
  if (prec == PREC_BF16) return a.K == 384 ? launch_panel_full<__bf16, 384>(pro, epi, a, s) : launch_panel_full<__bf16, 128>(pro, epi, a, s);
  return a.K == 384 ? launch_panel_full<_Float16, 384>(pro, epi, a, s) : launch_panel_full<_Float16, 128>(pro, epi, a, s);
}

}  // namespace effocr
