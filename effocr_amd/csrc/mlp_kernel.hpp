// Fused transformer MLP over the fragment-blocked residual stream (bf16 / f16 operands, gfx950):
//     x <- x + fc2(GELU(fc1(LayerNorm(x))))          (timm Block.mlp + norm2 + residual; models/encoders.py:58,63)
// The hidden activations [tokens, 4*D] never leave the CU: per block that removes 620 MB written + 620 MB read
// (ViT-S, 1024 crops) and one 310 MB pass over x — the MLP was 52 % of the forward and its two GEMMs were bound
// by exactly that traffic (profiles/README.md).
//
// One workgroup = 4 waves (one per SIMD, accumulators in AGPRs, gemm3.hip's regime) = a panel of 128 tokens;
// wave w owns tokens 32w..32w+31 from LayerNorm to the final store, so nothing is exchanged between waves:
//   prologue  LayerNorm of the wave's 32 rows, two lanes per row; lane (row r, half) loads exactly the fp32 chunks
//             that make up ITS MFMA B-operand fragments (k chunks 2t+half), so the normalised panel never exists
//             anywhere but in registers (D/16 fragments = 96 VGPRs at D = 384) — no LDS panel, no LDS reads for it;
//   chunk c   (128 hidden features; H/128 chunks):
//     phase A   hT[128 hid x 32 tok] = W1_c . xn^T          4 MFMA tiles x D/16 k-steps, W1 stages from the ring
//     GELU      bias + GELU on the accumulators, rounded to the operand type.  In the SWAPPED MFMA C-layout a
//               lane holds hidden {0-3, 8-11, 16-19, 24-27} (+4 for the upper half-wave) of its token: read as two
//               8-element vectors that IS a valid B-operand for phase B, provided W2's k index is permuted the
//               same way — done once on the host (fc2 weight copy "blocked + permuted", api.hip);
//     phase B   outT[D x 32 tok] += W2[:, c] . h_c           D/32 MFMA tiles x 8 k-steps, B-operand from registers
//   epilogue  x <- outT + bias2 + x (fp32, blocked).
// Weights stream through an 8-slot ring of 16 KB stages (32 cells = 4 row blocks x 64 k) by global_load_lds, six
// stages ahead (LDS holds nothing else but the biases); both weight copies are fragment-blocked so a stage is a
// verbatim copy of 512-byte HBM cells and every fragment read is conflict-free without swizzles.  As in gemm3 the
// barrier sits in the MIDDLE of a stage, so the first fragments of stage s+1 are read under stage s's last MFMAs.
// The bias + GELU hand-over of chunk c is a list of single scalar instructions (ChunkOps below) hosted in the issue slots behind the
// MFMAs of the two phases that follow its phase A, on one of two alternating register sets:
//     A(0) | park(0) | A(1)+gelu(0) | { B(c-1)+ops(c)[first part] | A(c+1)+ops(c)[rest] } ... | B(n-2)+ops(n-1)[first part] | ops(n-1)[rest] | B(n-1)
// (History, same-box A/B each: the round-2 form ran the GELU as 8-value bursts of packed FMAs behind single MFMAs — packed fp32 never
// overlaps the matrix pipe, tools/ubench/mfma_fill.hip; slicing THAT form over the gaps was 1-2 % slower, parking half a chunk under
// phase B's second k half 4 % slower; the scalar list is 7 % faster than the bursts.)
// FORMS of the body (round 6; launch_mlp picks by the call's row count, every boundary is parity-tested: tests/test_gpu_encoder.py
// test_embedding_across_the_kernel_selection_boundaries):
//   whole panels        128 tokens per workgroup, the whole MLP (the headline: 1576 panels per launch at 1024 crops)
//   split parts         the 128-token panels of a partially filled LAST round cut 2 / 4 / 6-way over the hidden chunks (PARTIAL), fp32
//                       partial sums + the reduction / LayerNorm launch; part 0 keeps the row (KEEP), the others start from zero
//   pair panels         64 tokens per workgroup, the two waves of a pair share a token tile: hidden features of a chunk, output tiles of the
//                       projection, both LayerNorms and the stores split between them, exchanged through LDS (PAIR; calls of 37-83 crops)
//   pair parts          pair panels whose hidden chunks are also cut over 6 / 3 / 2 workgroups + the reduction launch (PAIR && PARTIAL;
//                       calls of <= 13 / 27 / 36 crops and the class-token rows of the last block)
// (kernel template + launcher; instantiated per operand type / projection flag in mlp_*.hip so that the four
// variants compile in parallel: one translation unit took 200 s)
#pragma once
#include "common.hpp"
#include "kernels.hpp"
#include <type_traits>

#ifndef MLP_BARRIER_DRAIN
#define MLP_BARRIER_DRAIN 0
#endif
#ifndef MLP_FUSED_REDUCE_LN
#define MLP_FUSED_REDUCE_LN 1
#endif
#ifndef MLP_ROWS_LATE
#define MLP_ROWS_LATE 1                                  // rows of output group g+1 requested at the start of group g of the projection
#endif
#ifndef MLP_M0_ONCE
#define MLP_M0_ONCE 1                                    // LDS-DMA base (M0) written once per ring stage instead of saved / set / restored per piece
#endif
#ifndef MLP_GELU_BURST
#define MLP_GELU_BURST 2                                 // hand-over of a chunk: 0 = ~900 single instructions hosted in the MFMA gaps (rounds 3-5),
                                                         // 1 = park hosted + GELU as 8 packed-fp32 bursts under A(c+1), 2 = bias + GELU + round in 8 fused bursts under B(c-1)
#endif
#ifndef MLP_DIAG
#define MLP_DIAG 0                                       // timing ablations, WRONG results, never shipped (tools/ab_build.sh): 1 no hand-over ops,
#endif                                                   // 2 no W-fragment reads in the steady state, 4 no steady-state DMA, 8 no mid-stage barrier
#ifndef MLP_LN_PK
#define MLP_LN_PK 1                                      // LayerNorm arithmetic (norm2 and the second output) as packed fp32 pairs: nothing runs on the matrix pipe beside it, and a
#endif                                                   // v_pk_* issues like a scalar instruction (4.96 vs 4.7 cycles, tools/ubench/pk_burst.hip) for two values
#ifndef MLP_PRE_STAGES
#define MLP_PRE_STAGES 3                                 // ring stages requested in front of the prologue's wait (the rest behind its barrier)
#endif
#ifndef MLP_NT
#define MLP_NT 15                                        // non-temporal: 1 row loads, 2 attention-fragment loads, 4 row stores, 8 second-output stores
#endif

namespace effocr {
// the pair kernels (mlp_pair_kernel<E, 384, 1536, TNCW>, TNCW = 0 / 2 / 4 / 6) are instantiated in translation units of their own
// (mlp_bf16pair.hip, mlp_f16pair.hip: the fused-MLP objects compile in parallel; one unit with all 24 bodies took 5 minutes)
int mlp_pair_launch_bf16(const MlpArgs& a, int tncw, unsigned grid, hipStream_t s);
int mlp_pair_launch_f16(const MlpArgs& a, int tncw, unsigned grid, hipStream_t s);
namespace {

// -DMLP_STAMP (tools/ab_build.sh variant, never shipped): wave 0 of every whole-panel workgroup records s_memtime at the panel's
// milestones + its hardware id; tools/mlp_timeline.py reads the last launch's table through effocr_debug_mlp_stamps.
#ifdef MLP_STAMP
constexpr int MLP_STAMP_WGS = 2048, MLP_STAMP_N = 20;   // 0-11 milestones, 12 / 13 wait sums, 15 hardware id, 14 / 16 the 100 MHz real-time counter at entry / exit
__device__ unsigned long long mlp_stamps[MLP_STAMP_WGS * MLP_STAMP_N];
// branch-free (a branch would split the kernel's scheduling regions): every lane of wave 0..3 stores, the last writer's value stays
#ifndef MLP_STAMP_PART
#define MLP_STAMP_PART false                             // true: the split parts stamp instead of the whole panels (small calls)
#endif
#define MLP_STAMP_AT(k)                                                                                          \
  if constexpr (PARTIAL == MLP_STAMP_PART) mlp_stamps[(bid & (MLP_STAMP_WGS - 1)) * MLP_STAMP_N + (k)] = __builtin_amdgcn_s_memtime();
#else
#define MLP_STAMP_AT(k)
#endif

template <int I, int N, typename F> __device__ __forceinline__ void sfor(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    sfor<I + 1, N>(f);
  }
}


// ---- the hand-over work of one hidden chunk (128 features x 32 tokens per wave = 16 quads of 4 values per lane) as a list of single
// instructions ("ops"), hosted a few at a time in the issue slots behind the MFMAs of the two phases that follow the chunk's phase A:
//   park  (quad q): bias read | 4 x (accumulator + bias) | 2 x pack            -> pre-activations, rounded to the operand type
//   gelu  (quad q): 4 x unpack | clamp | square | DEG x Horner | 0.5 + u p | x . | 2 x pack   (GeluFit<E>: gelu_fold_n value for value)
// Measured (tools/ubench/mfma_fill.hip, one wave per SIMD): up to 6 scalar VALU instructions (or a ds_read_b128 + 4) behind a
// v_mfma_f32_32x32x16 are free (32.8 -> 34 cycles per MFMA), every further one costs its 4 cycles; ONE v_pk_fma_f32 in the same
// place costs 17 cycles and each further one 4.5 — packed fp32 never overlaps the matrix pipe.  So the list is scalar.
template <int DEG> struct ChunkOps {
  // park list: op 0 = bias read of quad 0, then per quad { bias read of the NEXT quad (its use is two gaps away) | 4 adds | 2 packs }
  static constexpr int PARK_Q = 7, GELU_Q = 4 * (DEG + 5) + 2, NPARK = 1 + 16 * PARK_Q, N = NPARK + 16 * GELU_Q;
  static constexpr int cost_before(int k) {               // issued instructions of ops [0, k): an accumulator add is a v_accvgpr_read + v_add
    if (k >= NPARK) return 1 + 16 * 11 + (k - NPARK);
    if (k == 0) return 0;
    const int q = (k - 1) / PARK_Q, o = (k - 1) % PARK_Q;
    return 1 + q * 11 + (o == 0 ? 0 : o <= 4 ? 1 + 2 * (o - 1) : 9 + (o - 5));
  }
  // first op of gap j when ops [K0, K1) are spread over NG gaps by cost (j = NG: K1)
  static constexpr int first_op(int K0, int K1, int NG, int j) {
    const int w = cost_before(K1) - cost_before(K0);
    int k = K0;
    while (k < K1 && (cost_before(k) - cost_before(K0)) * NG < j * w) ++k;
    return j >= NG ? K1 : k;
  }
  // split point of a chunk's list between the NB gaps of the phase B and the NA gaps of the phase A that host it
  static constexpr int split(int NB, int NA) {
    const int total = cost_before(N);
    int k = 0;
    while (k < N && cost_before(k) * (NA + NB) < total * NB) ++k;
    return k < NPARK ? NPARK : k;                          // the park ops all sit in phase B: phase A overwrites the accumulators
  }
};

template <int BIT, typename T> __device__ __forceinline__ T ld_act(const T* p) {
  if constexpr ((MLP_NT & BIT) != 0) return __builtin_nontemporal_load(p);
  else return *p;
}
template <int BIT, typename T> __device__ __forceinline__ void st_act(T* p, const T& v) {
  if constexpr ((MLP_NT & BIT) != 0) __builtin_nontemporal_store(v, p);
  else *p = v;
}

constexpr int MLP_PT = 128;                              // tokens per workgroup
constexpr int MLP_STAGE = 16384;                         // bytes per ring stage: 4 row blocks x 8 k-chunks x 512 B
constexpr int MLP_RING = 8;

// NCW = hidden chunks run by one workgroup: H/128 (whole MLP of its panel), or — PARTIAL, second launch — a slice
// of them for the panels of the last, partially filled round of CUs: the split workgroups write fp32 partial
// outputs to a scratch buffer and mlp_reduce_kernel adds them, bias2 and the residual in a fixed order.
// PROJ: the attention output projection + residual runs first, in the same workgroup:  x <- x + a . Wp^T + bp.
// Its weight copy has the rows of every 32-row block permuted (api.hip put_op_blocked rowperm) so that the swapped
// C-layout hands each lane 8 CONSECUTIVE output features per (tile, register octet) — exactly the fp32 chunks
// 4t+2half, 4t+2half+1 the LayerNorm below expects in xv[]: accumulators -> (+bias, +residual) -> xv, no exchange.
template <int D, int H> constexpr int mlp_smem_bytes() { return MLP_RING * MLP_STAGE + (H + 6 * D) * 4; }
// the workgroup's LDS: ONE object for both bodies of a kernel (whole panels / split parts) — a pointer parameter would make
// the compiler lose the address space (and, measured, spill 370 registers in the LayerNorm)
constexpr int MLP_PAIR_XCH = 16384 + 1024;              // PAIR: 4 fragments x 4 waves of the LayerNorm's hand-over per round + the row statistics
__shared__ __attribute__((aligned(16))) char mlp_smem[mlp_smem_bytes<384, 1536>() + MLP_PAIR_XCH];   // 163 840 B = the CU's 160 KB (one workgroup per CU anyway)

// body of one workgroup: bid = its index among the workgroups of its kind (whole panels / split parts)
// PAIR (round 6, calls of 33-83 crops: 64-token panels): the workgroup owns TWO 32-token tiles; the waves 2j, 2j+1 share tile j.  Both run
// the projection and the LayerNorm of the tile (redundantly: no exchange), then wave p = w & 1 takes hidden tiles 2p, 2p+1 of every
// chunk in phase A (row blocks 2p, 2p+1 of the ring stage), applies bias + GELU to those 64 hidden features, and multiplies them — k half
// p of the chunk — into all D outputs in phase B, whose ring stages are re-dealt as [output tile 2g + (w & 1)][k half w >> 1] so that every
// wave finds its two tiles in every stage: 8 MFMAs per wave and stage in both phases, the SAME weight stream as a 128-token panel (a
// stage still feeds all four waves).  The two partial sums of a tile meet through LDS (the drained ring) at the end; wave 2j writes.
// KEEP (PROJ forms that share a token tile between several accumulations: the split parts of a panel, the two waves of a PAIR): true =
// this one keeps the new row x + bp + a.Wp^T in its fc2 accumulators (part 0 / wave 2j), false = its partial sums start at zero — the
// first phase-B MFMA of every output tile takes the instruction's zero operand, so the row's 192 registers are dead behind the LayerNorm
// (before: 576 read / multiply-by-0-or-1 / write instructions per part next to 21-46 spilled registers reloaded inside the chunk loop).
// PPW (PAIR only) = which wave of the pair this body is (w & 1): a compile-time constant, the kernel branches per wave.  PAIR && PARTIAL
// ("pair parts", calls of <= 27 crops): the 64-token panel's hidden chunks are ALSO dealt over (H / 128) / NCW workgroups — half the
// projection / LayerNorm work per wave of a 128-token part, half as many partial sums for the reduction launch; KEEP then says whether this
// part keeps the row in the tiles its wave owns (part 0) or starts every tile from zero.
template <typename E, int D, int H, int NCW, bool PARTIAL, bool PROJ, bool PAIR = false, bool KEEP = true, int PPW = 0>
__device__ __forceinline__ void mlp_fused_body(const MlpArgs& a, const int bid) {
  static_assert(KEEP || (PROJ && (PARTIAL || PAIR)), "mlp: KEEP = false is a split part / the second wave of a pair behind the projection");
  static_assert(mlp_smem_bytes<D, H>() <= (int)sizeof(mlp_smem), "mlp: LDS object too small");
  static_assert(!PAIR || (PROJ && D % 128 == 0 && (PARTIAL || (NCW == H / 128 && KEEP))), "mlp: the pair form has the projection in front; whole MLP, or split parts");
  static_assert(PAIR || PPW == 0, "mlp: PPW is the pair form's wave index");
  char* smem = mlp_smem;
  typedef typename Op16<E>::V8 V8;
  constexpr int KC = D / 8;                              // 16-B k chunks per xn row
  constexpr int NC = NCW;                                // hidden chunks of this workgroup (H / 128 in total)
  constexpr int SA = D / 64;                             // ring stages per phase A
  constexpr int OT = D / 32;                             // output tiles (32 features each) per token block
  constexpr int OG = OT / 4;                             // output tile groups of 4 (one ring stage holds 4 row blocks)
  constexpr int SB = 2 * OG;                             // ring stages per phase B: (group, k half)
  constexpr int SP = PROJ ? OG * SA : 0;                 // ring stages of the projection: (output group, k stage)
  constexpr int NS = SP + NC * (SA + SB);                // ring stages per panel
  static_assert(D % 128 == 0 && H % 128 == 0 && (H / 128) % NCW == 0, "mlp: D and H must be multiples of 128");
  constexpr int NXF = D / 16;                            // xn B-operand fragments per lane (one per k16 step)
  constexpr int R = MLP_RING;
  char* sW = smem;
  float* sB1 = reinterpret_cast<float*>(smem + R * MLP_STAGE);
  float* sB2 = sB1 + H;
  float* sG = sB2 + D;                                   // norm2 weight / bias: LDS reads do not queue behind the ring's DMAs
  float* sBt = sG + D;
  float* sBp = sBt + D;                                  // proj bias (row-permuted like its weight)
  float* sGn = sBp + D;                                  // next block's norm1 weight / bias (second output)
  float* sBn = sGn + D;

  const int tid = threadIdx.x, lane = tid & 63, r31 = lane & 31, half = lane >> 5;
  const int w = wave_id();
  // PROJ: the projection bias and bias2 are added by ONE extra MFMA per output tile instead of 16 x (read accumulator, add, write
  // back) per lane: A = [32 features x 16 k] with k0 = hi(bias), k1 = lo(bias) (two operand-type values: 16+ mantissa bits),
  // B = [16 k x 32 tokens] with rows k0 = k1 = 1.  ~1 000 of the 3 400 VALU instructions between the projection and phase A(0).
  constexpr bool BIAS_MM = PROJ;                         // (the split parts too, round 6: 576 accumulator read / add / write instructions per part)
  auto hilo = [](float b) __attribute__((always_inline)) -> uint32_t {
    const E hi = (E)b;
    const E lo = (E)(b - (float)hi);
    return pack2<E>((float)hi, (float)lo);                 // (exact: both are operand-type values)
  };
  MLP_STAMP_AT(0)
#ifdef MLP_STAMP
  if (PARTIAL == MLP_STAMP_PART) mlp_stamps[(bid & (MLP_STAMP_WGS - 1)) * MLP_STAMP_N + 14] = __builtin_amdgcn_s_memrealtime();
  if (PARTIAL == MLP_STAMP_PART) mlp_stamps[(bid & (MLP_STAMP_WGS - 1)) * MLP_STAMP_N + 15] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) | ((unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) << 32);
#endif
  // First round of workgroups only: spread the start over `stagger` x 32 ticks.  All panels cost the same, so the
  // workgroups of a launch otherwise stay in lock step from the first to the last round: every CU requests its rows at the
  // same moment (61 MB per round: an HBM-bound 12 us during which nothing computes) and stores them at the same moment.
  if (!PARTIAL && a.stagger > 0 && bid < a.stagger_wgs) {
    const unsigned long long until = __builtin_amdgcn_s_memtime() + (unsigned long long)(((bid >> 3) & 31) * a.stagger);
    while (__builtin_amdgcn_s_memtime() < until) __builtin_amdgcn_s_sleep(8);
  }
  MLP_STAMP_AT(1)
  constexpr int SPLIT = (H / 128) / NCW;                 // workgroups per panel
  const int panel = (PARTIAL ? a.panel0 : 0) + bid / SPLIT;
  const int c0 = (bid % SPLIT) * NCW;        // first hidden chunk of this workgroup
  const int pp = PAIR ? PPW : 0;                         // PAIR: which half of a chunk's hidden features (and which output tiles) this wave carries
  const int64_t rb = PAIR ? (int64_t)panel * 2 + (w >> 1) : (int64_t)panel * 4 + w;   // this wave's 32-row block of x
  const char* W1 = static_cast<const char*>(a.W1b);
  const char* W2 = static_cast<const char*>(a.W2p);

  // ---- every request of the prologue is in flight before anything waits: parameters (oldest in the in-order VM queue), the wave's
  // rows, then the first R-1 ring stages.  The parameters go STRAIGHT to LDS by LDS-DMA (round 6, second half): through registers the
  // compiler put s_waitcnt vmcnt(0) in front of their ds_write (an LDS write behind a global_load_lds it knows about) — i.e. the
  // workgroup sat until its rows, attention fragments and all seven ring stages (112 KB) had landed before the first barrier:
  // 14-18 k ticks of every panel and every split part (tools/mlp_part_timeline.py).  Now 16-byte units u = 256 j + tid of the LDS
  // parameter image [b1 | b2 | gamma | beta | bp | gamma_n | beta_n] come from their arrays (16-byte aligned: launch_mlp checks), a lane
  // without a unit (past the end, absent array) is masked off; nothing of it is visible to the compiler, the wait is counted by hand.
  constexpr int U1 = H / 4, UD = D / 4, NU = U1 + 6 * UD, NPI = (NU + 255) / 256;
  const bool second = !PARTIAL && a.xn_out != nullptr;
  {
    const unsigned sP_lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)(smem + R * MLP_STAGE);
#pragma unroll
    for (int j = 0; j < NPI; ++j) {
      const int u = j * 256 + tid;
      const int v = u - U1, arr = v / UD, o = v - arr * UD;
      const float* base = u < U1 ? a.b1 : arr == 0 ? a.b2 : arr == 1 ? a.gamma : arr == 2 ? a.beta : arr == 3 ? (PROJ ? a.bp : nullptr)
                                        : arr == 4 ? (second ? a.gamma_n : nullptr) : (second ? a.beta_n : nullptr);
      const char* src = reinterpret_cast<const char*>(base) + (size_t)(u < U1 ? u : o) * 16;
      const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(sP_lds + (unsigned)(j * 256 + w * 64) * 16u));
      if (u < NU && base != nullptr)
        asm volatile("s_mov_b32 m0, %1\n\t"
                     "s_nop 0\n\t"
                     "global_load_lds_dwordx4 %0, off"
                     :: "v"(src), "s"(dst) : "memory", "m0");
    }
  }
  asm volatile("" ::: "memory");
  // lane = (row r31, half): 16-bit k chunk 2t+half of its row = fp32 chunks 4t+2half, 4t+2half+1
  f32x16 acc1[4];                                        // hT tiles of the current chunk
  f32x16 acc2[OT];                                       // outT tiles
  f32x4 xv[2 * NXF];
  V8 xf[NXF];                                            // B-operand fragments: attention output (PROJ), then LayerNorm(x)
  auto bias_mm = [&](const float* sb, auto T_) __attribute__((always_inline)) {   // acc2[t] += bias (as stored by the prologue)
    constexpr int t = decltype(T_)::value;
    const uint32_t wd = hilo(sb[t * 32 + r31]);            // (hi, lo) operand-type pair of the fp32 bias: 16+ mantissa bits
    const uint32_t one2 = GeluFit<E>::lo ? 0x3f803f80u : 0x3c003c00u;
    const u32x4 af = {half ? 0u : wd, 0u, 0u, 0u}, bf = {half ? 0u : one2, 0u, 0u, 0u};
    acc2[t] = Op16<E>::mfma(__builtin_bit_cast(V8, af), __builtin_bit_cast(V8, bf), acc2[t]);
  };
  const int64_t rbc = rb < (a.rows_alloc >> 5) ? rb : (a.rows_alloc >> 5) - 1;
  const char* xb = reinterpret_cast<const char*>(a.x) + rbc * (D / 4) * 512 + r31 * 16;
  // (PROJ) x is the projection's initial accumulator value (x + bias + a . Wp^T lands where LayerNorm reads it).  Output tile t,
  // registers 4q..4q+3 = fp32 chunk 8t + 4(q>>1) + 2half + (q&1) of the row = xv[4t + q].  The rows of the first output group
  // are requested here, the rest at the start of the projection (they land under its first group of MFMAs): no load latency
  // in the middle of the kernel and never more than 128 VGPRs of rows next to the 96 of the attention fragments.
  // PAIR: wave p of a pair owns output tiles 4g + 2p, 4g + 2p + 1 — their rows, their share of the projection, of both LayerNorms and of the stores
  auto own_tile = [&](int t) __attribute__((always_inline)) { return !PAIR || ((t & 3) >> 1) == pp; };
  auto load_rows = [&](int t0, int t1) __attribute__((always_inline)) {
#pragma unroll
    for (int t = t0; t < t1; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (own_tile(t)) xv[4 * t + q] = ld_act<1>(reinterpret_cast<const f32x4*>(xb + (size_t)(8 * t + 4 * (q >> 1) + 2 * half + (q & 1)) * 512));
  };
  auto rows_to_acc = [&](auto T0_, auto T1_) __attribute__((always_inline)) {
    constexpr int t0 = decltype(T0_)::value, t1 = decltype(T1_)::value;
    load_rows(t0, t1);
#pragma unroll
    for (int t = t0; t < t1; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) if (own_tile(t)) acc2[t][4 * q + e] = xv[4 * t + q][e];
  };
  if constexpr (PROJ) {
    rows_to_acc(std::integral_constant<int, 0>{}, std::integral_constant<int, (MLP_ROWS_LATE && OG > 1) ? 4 : OT>{});
    const char* ab = static_cast<const char*>(a.A) + rbc * KC * 512 + half * 512 + r31 * 16;
#pragma unroll
    for (int t = 0; t < NXF; ++t) xf[t] = ld_act<2>(reinterpret_cast<const V8*>(ab + (size_t)t * 1024));
  } else {
#pragma unroll
    for (int t = 0; t < NXF; ++t) {
      {
      xv[2 * t] = ld_act<1>(reinterpret_cast<const f32x4*>(xb + (size_t)(4 * t + 2 * half) * 512));
      xv[2 * t + 1] = ld_act<1>(reinterpret_cast<const f32x4*>(xb + (size_t)(4 * t + 2 * half + 1) * 512));
      }
    }
  }
  asm volatile("" ::: "memory");

  // ---- ring: stage s of the panel's stream.  Order: A(0) | A(1) | B(0) | A(2) | B(1) | ... | A(NC-1) | B(NC-2) | B(NC-1).
  // Wave w copies row block w of the stage: 4 pieces of 1 KB (two adjacent k-chunk cells each).
  auto stage_src = [&](int s_in) __attribute__((always_inline)) -> const char* {
    int s = s_in < NS ? s_in : NS - 1;                   // a stage index past the end (the rolled loop's "plenty follows"
    int c, r;                                            // flag is optimistic in its last iteration for small D) re-reads the last stage
    if constexpr (PROJ) {
      if (s < SP) {                                      // projection: stage (group g = s / SA, k stage s % SA)
        const int g = s / SA, ks = s - g * SA;
        return static_cast<const char*>(a.Wpp) + ((size_t)(4 * g + w) * KC + 8 * ks) * 512;
      }
      s -= SP;
    }

    bool isA;
    if (s < SA) { c = 0; r = s; isA = true; }
    else {
      const int t = s - SA, p = t / (SA + SB);
      r = t - p * (SA + SB);
      if (p < NC - 1) { isA = r < SA; c = isA ? p + 1 : p; r = isA ? r : r - SA; }
      else { isA = false; c = NC - 1; }                  // trailing B(NC-1): r counts its stages
    }
    if (isA) return W1 + ((size_t)(4 * (c0 + c) + w) * KC + 8 * r) * 512;
    if constexpr (PAIR) return W2 + ((size_t)(2 * r + (w & 1)) * (H / 8) + 16 * (c0 + c) + 8 * (w >> 1)) * 512;   // [tile 2r + (w & 1)][k half w >> 1]
    const int g = r >> 1, kh = r & 1;
    return W2 + ((size_t)(4 * g + w) * (H / 8) + 16 * (c0 + c) + 8 * kh) * 512;
  };
  // The four 1 KB pieces of a wave's share of a stage use ONE source address and ONE LDS base (M0): the piece index is the
  // instruction's immediate offset, which the hardware adds on both sides (7 address / M0 instructions per piece before: a
  // quarter of the loop's non-MFMA issue slots).
  const unsigned lane16 = (unsigned)lane * 16u;
  // Steady state: the wave's four pieces of a stage in ONE inline-asm statement.  hipcc models __builtin_amdgcn_global_load_lds as an
  // access to BOTH address spaces ("pending flat"): after every such instruction its next LDS wait is s_waitcnt lgkmcnt(0) instead
  // of a counted one, i.e. the W-fragment read issued a moment earlier is drained on the spot — 58 full LDS drains per 192 MFMAs
  // in this loop (70 of its 73 lgkmcnt waits were (0)), each exposing ~100 cycles of LDS latency at one wave per SIMD.  Hidden in
  // asm the DMA costs the compiler nothing; its completion is counted by hand anyway (stage_mid: vmcnt).  M0 = LDS base of the
  // pieces (saved / restored: the register is the compiler's), the s_nop is the M0-write -> LDS-DMA wait state.
  const unsigned sW_lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)sW;
  // one piece (the four pieces of a stage go out in four different MFMA gaps behind the barrier, so that the
  // sixteen 1 KB requests of the workgroup do not hit the address unit in one burst)
  auto issue_piece_asm = [&](int s, auto I_) __attribute__((always_inline)) {
    constexpr int i = decltype(I_)::value;
    const char* src = stage_src(s) + lane16;
    const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(sW_lds + (unsigned)((s & (R - 1)) * MLP_STAGE) + (unsigned)w * 4096u));
#if MLP_M0_ONCE
    // The four pieces of a stage share ONE LDS base: M0 is written with piece 0 and stays for pieces 1-3 (issued in later MFMA gaps of
    // the same stage).  Nothing else in the main loop touches M0 (gfx9 LDS instructions do not use it; checked in the emitted code),
    // and the compiler is told it is clobbered.  Before: save / set / wait state / restore around EVERY piece = 4 scalar instructions
    // per piece, 384 of the loop body's 3 649 instructions per 384 MFMAs — in a loop that pays ~4.5 cycles per issued instruction.
    if constexpr (i == 0)
      asm volatile("s_mov_b32 m0, %1\n\t"
                   "s_nop 0\n\t"
                   "global_load_lds_dwordx4 %0, off"
                   :: "v"(src), "s"(dst) : "memory", "m0");
    else
      asm volatile("global_load_lds_dwordx4 %0, off offset:%1" :: "v"(src), "n"(i * 1024) : "memory", "m0");
#else
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\t"
                 "s_mov_b32 m0, %2\n\t"
                 "s_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, off offset:%3\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(src), "s"(dst), "n"(i * 1024) : "memory");
#endif
  };
  // The ring's first PRE stages go out in front of the wait, the other R-1-PRE behind it.  Why not all of them in front: the compiler
  // counts only the loads it knows (rows, attention fragments) — with asm DMAs in flight BEHIND those, its vmcnt(n) in front of a
  // fragment's first use would drain the ring down to n pieces.  So its loads are all consumed ("touched") here, where nothing younger
  // than stage PRE-1 is in flight, and one explicit wait covers parameters, rows, fragments and the first PRE stages: 16 + 160 + 16 PRE KB
  // per workgroup instead of 288 KB (the compiler's own vmcnt(0) in front of the parameters' ds_write, rounds 3-6).
  constexpr int PRE = (MLP_PRE_STAGES < NS ? MLP_PRE_STAGES : NS) < R - 1 ? (MLP_PRE_STAGES < NS ? MLP_PRE_STAGES : NS) : R - 1;
  sfor<0, PRE>([&](auto S0_) {
    sfor<0, 4>([&](auto I_) { issue_piece_asm(decltype(S0_)::value, I_); });
  });
  asm volatile("" ::: "memory");
  if constexpr (PROJ) {
#pragma unroll
    for (int t = 0; t < ((MLP_ROWS_LATE && OG > 1) ? 4 : OT); ++t) if (own_tile(t)) asm volatile("" : "+a"(acc2[t]));
#pragma unroll
    for (int t = 0; t < NXF; ++t) asm volatile("" : "+v"(xf[t]));
  } else {
#pragma unroll
    for (int i = 0; i < 2 * NXF; ++i) asm volatile("" : "+v"(xv[i]));
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                          // everybody's parameters and stages 0 .. PRE-1
  asm volatile("" ::: "memory");
  sfor<PRE, (R - 1 < NS ? R - 1 : NS)>([&](auto S0_) {
    sfor<0, 4>([&](auto I_) { issue_piece_asm(decltype(S0_)::value, I_); });
  });
  asm volatile("" ::: "memory");
  MLP_STAMP_AT(2)

  if constexpr (PARTIAL && !PROJ) {
#pragma unroll
    for (int t = 0; t < OT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc2[t][r] = 0.f;
  }

  // ---- LayerNorm in registers: xv (this lane's half of the row, fp32) -> xf[t] = B-operand fragment of k16 step t.
  // Whole panels: the fp32 row moves on into the fc2 accumulators (acc2 = x + bias2; fc2 accumulates on top), so the
  // residual is read ONCE per block.  W2's rows are permuted per 32 (api.hip rowperm32) such that registers 4q..4q+3 of
  // output tile t ARE the fp32 chunk 8t + 4(q>>1) + 2half + (q&1) = xv[2(2t + (q>>1)) + (q&1)]: no exchange.
  // (PROJ: the row sits in acc2 — chunk i of the lane's order = registers 4(i&3).. of tile i>>2 — else in xv)
  auto row = [&](auto I_) __attribute__((always_inline)) -> f32x4 {
    constexpr int i = decltype(I_)::value;
    if constexpr (PROJ) return f32x4{acc2[i >> 2][4 * (i & 3)], acc2[i >> 2][4 * (i & 3) + 1], acc2[i >> 2][4 * (i & 3) + 2], acc2[i >> 2][4 * (i & 3) + 3]};
    else return xv[i];
  };
  // (PROJ) the row stays in the accumulator registers between the passes: without the pin the compiler keeps pass 1's 192
  // VGPR copies of it alive for the later passes and spills them
  auto pin_row = [&]() __attribute__((always_inline)) {
    if constexpr (PROJ) {
#pragma unroll
      for (int t = 0; t < OT; ++t) if (own_tile(t)) asm volatile("" : "+a"(acc2[t]));
    } else {                                             // (same for the VGPR copy: else (v - mean) of the variance pass is kept, i.e. spilled, for the last pass)
#pragma unroll
      for (int i = 0; i < 2 * NXF; ++i) asm volatile("" : "+v"(xv[i]));
    }
  };
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  auto layernorm_to_xf = [&]() __attribute__((always_inline)) {
    float sm = 0.f;
    pin_row();
    if constexpr (MLP_LN_PK) {
      f32x2 s2 = {0.f, 0.f};
      sfor<0, 2 * NXF>([&](auto I_) { const f32x4 v = row(I_); s2 += f32x2{v[0], v[1]}; s2 += f32x2{v[2], v[3]}; });
      sm = s2[0] + s2[1];
    } else
    sfor<0, 2 * NXF>([&](auto I_) { const f32x4 v = row(I_); sm += (v[0] + v[1]) + (v[2] + v[3]); });
    sm += __shfl_xor(sm, 32, 64);
    const float mean = sm * (1.0f / D);
    float ss = 0.f;
    pin_row();
    if constexpr (MLP_LN_PK) {                             // still two passes: exact statistics of the fp32 row
      f32x2 q2 = {0.f, 0.f};
      const f32x2 nm = {-mean, -mean};
      sfor<0, 2 * NXF>([&](auto I_) {
        const f32x4 v = row(I_);
        const f32x2 d0 = f32x2{v[0], v[1]} + nm, d1 = f32x2{v[2], v[3]} + nm;
        q2 = __builtin_elementwise_fma(d0, d0, q2);
        q2 = __builtin_elementwise_fma(d1, d1, q2);
      });
      ss = q2[0] + q2[1];
    } else
    sfor<0, 2 * NXF>([&](auto I_) {
      const f32x4 v = row(I_);
#pragma unroll
      for (int e = 0; e < 4; ++e) { const float d = v[e] - mean; ss += d * d; }
    });
    ss += __shfl_xor(ss, 32, 64);
    const float rstd = 1.0f / sqrtf(ss * (1.0f / D) + a.eps);
    const f32x2 r2 = {rstd, rstd}, nm2 = {-mean, -mean};
    pin_row();
    sfor<0, NXF>([&](auto T_) {
      constexpr int t = decltype(T_)::value;
      u32x2 pk[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int c = 4 * t + 2 * half + j;
        const f32x4 gm = *reinterpret_cast<const f32x4*>(sG + c * 4);
        const f32x4 bt = *reinterpret_cast<const f32x4*>(sBt + c * 4);
        const f32x4 v = j == 0 ? row(std::integral_constant<int, 2 * t>{}) : row(std::integral_constant<int, 2 * t + 1>{});
        if constexpr (MLP_LN_PK) {                         // ((v - mean) rstd) gamma + beta, two values per instruction (same operations per value)
          const f32x2 o0 = __builtin_elementwise_fma((f32x2{v[0], v[1]} + nm2) * r2, f32x2{gm[0], gm[1]}, f32x2{bt[0], bt[1]});
          const f32x2 o1 = __builtin_elementwise_fma((f32x2{v[2], v[3]} + nm2) * r2, f32x2{gm[2], gm[3]}, f32x2{bt[2], bt[3]});
          pk[j] = pack4<E>(o0[0], o0[1], o1[0], o1[1]);
        } else
        pk[j] = pack4<E>((v[0] - mean) * rstd * gm[0] + bt[0], (v[1] - mean) * rstd * gm[1] + bt[1],
                         (v[2] - mean) * rstd * gm[2] + bt[2], (v[3] - mean) * rstd * gm[3] + bt[3]);
        if constexpr (!PARTIAL && !BIAS_MM) {
          constexpr int tt = t >> 1;
          const int q = 2 * (t & 1) + j;
          const f32x4 bv = *reinterpret_cast<const f32x4*>(sB2 + tt * 32 + 8 * q + 4 * half);
#pragma unroll
          for (int e = 0; e < 4; ++e) acc2[tt][4 * q + e] = v[e] + bv[e];
        }
      }
      const u32x4 q = {pk[0][0], pk[0][1], pk[1][0], pk[1][1]};
      xf[t] = __builtin_bit_cast(V8, q);
      if constexpr (!PARTIAL) __builtin_amdgcn_sched_barrier(0);   // in order: the row's registers become the accumulators' one chunk pair at a time
    });
  };
  // PAIR: norm2 of a token tile by its two waves.  Each holds the new row's features of its own six tiles (fp32 chunks i with tile i >> 2
  // own): partial sums of both statistics passes meet through 1 KB of LDS (own + partner's: the same sum in both waves), each normalises its
  // own 12 operand fragments (fragment t' = tile t' >> 1) and the halves are exchanged through a 16 KB buffer in three rounds of four fragments
  // per wave (round g: fragments 8g + 4p .. + 3; the ring keeps streaming underneath — that is why the buffer is this small).
  auto layernorm_to_xf_pair = [&]() __attribute__((always_inline)) {
    constexpr int PP = PPW;
    float* sS = reinterpret_cast<float*>(smem + mlp_smem_bytes<D, H>() + 16384);   // [sum | sum of squares][wave][token]
    char* xch = smem + mlp_smem_bytes<D, H>();                                      // [wave][fragment j of the round][lane] x 16 B
    pin_row();
    f32x2 s2 = {0.f, 0.f};
    sfor<0, 2 * NXF>([&](auto I_) {
      if constexpr ((((decltype(I_)::value >> 2) & 3) >> 1) == PP) { const f32x4 v = row(I_); s2 += f32x2{v[0], v[1]}; s2 += f32x2{v[2], v[3]}; }
    });
    float sm = s2[0] + s2[1];
    sm += __shfl_xor(sm, 32, 64);
    if (half == 0) sS[w * 32 + r31] = sm;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    sm += sS[(w ^ 1) * 32 + r31];
    const float mean = sm * (1.0f / D);
    pin_row();
    f32x2 q2 = {0.f, 0.f};
    const f32x2 nm = {-mean, -mean};
    sfor<0, 2 * NXF>([&](auto I_) {
      if constexpr ((((decltype(I_)::value >> 2) & 3) >> 1) == PP) {
        const f32x4 v = row(I_);
        const f32x2 d0 = f32x2{v[0], v[1]} + nm, d1 = f32x2{v[2], v[3]} + nm;
        q2 = __builtin_elementwise_fma(d0, d0, q2);
        q2 = __builtin_elementwise_fma(d1, d1, q2);
      }
    });
    float ss = q2[0] + q2[1];
    ss += __shfl_xor(ss, 32, 64);
    if (half == 0) sS[128 + w * 32 + r31] = ss;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    ss += sS[128 + (w ^ 1) * 32 + r31];
    const float rstd = 1.0f / sqrtf(ss * (1.0f / D) + a.eps);
    const f32x2 r2 = {rstd, rstd};
    pin_row();
    sfor<0, NXF>([&](auto T_) {
      constexpr int t = decltype(T_)::value;
      if constexpr ((((t >> 1) & 3) >> 1) == PP) {
        u32x2 pk[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int c = 4 * t + 2 * half + j;
          const f32x4 gm = *reinterpret_cast<const f32x4*>(sG + c * 4);
          const f32x4 bt = *reinterpret_cast<const f32x4*>(sBt + c * 4);
          const f32x4 v = j == 0 ? row(std::integral_constant<int, 2 * t>{}) : row(std::integral_constant<int, 2 * t + 1>{});
          const f32x2 o0 = __builtin_elementwise_fma((f32x2{v[0], v[1]} + nm) * r2, f32x2{gm[0], gm[1]}, f32x2{bt[0], bt[1]});
          const f32x2 o1 = __builtin_elementwise_fma((f32x2{v[2], v[3]} + nm) * r2, f32x2{gm[2], gm[3]}, f32x2{bt[2], bt[3]});
          pk[j] = pack4<E>(o0[0], o0[1], o1[0], o1[1]);
        }
        const u32x4 q = {pk[0][0], pk[0][1], pk[1][0], pk[1][1]};
        xf[t] = __builtin_bit_cast(V8, q);
      }
    });
    sfor<0, OG>([&](auto G_) {                             // hand-over, one output group per round
      constexpr int g = decltype(G_)::value;
#pragma unroll
      for (int j = 0; j < 4; ++j) *reinterpret_cast<V8*>(xch + (w * 4 + j) * 1024 + lane * 16) = xf[8 * g + 4 * PP + j];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
#pragma unroll
      for (int j = 0; j < 4; ++j) xf[8 * g + 4 * (1 - PP) + j] = *reinterpret_cast<const V8*>(xch + ((w ^ 1) * 4 + j) * 1024 + lane * 16);
      if constexpr (g + 1 < OG) {                          // the partner has read before the next round overwrites
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      }
    });
  };
  if constexpr (!PROJ) {
    layernorm_to_xf();
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc1[i][r] = 0.f;
  }

  const int wo = half * 512 + r31 * 16;                  // + (row block i * 8 + 2 * c4) * 512 inside a stage
  const int wo2 = wo + pp * 8192;                        // PAIR: row blocks 2p, 2p+1 of a stage
  int s = 0;                                             // ring stage counter
  struct WF { V8 w[4]; };
  auto load_w = [&](WF& f, const char* st, int c4) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 4; ++i) f.w[i] = *reinterpret_cast<const V8*>(st + wo + (i * 8 + 2 * c4) * 512);
  };
  // middle of stage s: stage s+1 has landed (own pieces; the R-3 younger stages may stay in flight) and, past
  // the barrier, everybody's; every wave holds the rest of stage s in registers and is done with stage s-1, whose
  // slot takes stage s+R-1
#ifdef MLP_STAMP
  unsigned long long vm_wait = 0, bar_wait = 0;          // ticks this wave spent in the mid-stage vmcnt wait / barrier (sum over the panel)
#endif
  auto stage_mid = [&](auto STEADY) __attribute__((always_inline)) {
#ifdef MLP_STAMP
    const unsigned long long t0_ = __builtin_amdgcn_s_memtime();
#endif
    if constexpr (decltype(STEADY)::value) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((R - 3) * 4) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef MLP_STAMP
    const unsigned long long t1_ = __builtin_amdgcn_s_memtime();
#endif
#if MLP_BARRIER_DRAIN
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (A/B) not needed: the slot refilled behind this barrier is stage s-1's, whose fragment
#endif                                                  // reads were all consumed by MFMAs before stage s began; the reads in flight here are stage s's
    if constexpr (!(MLP_DIAG & 8)) __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#ifdef MLP_STAMP
    const unsigned long long t2_ = __builtin_amdgcn_s_memtime();
    vm_wait += t1_ - t0_; bar_wait += t2_ - t1_;
#endif
  };
  WF wf;                                                 // the step's four W fragments (rolling refill, see ring_stage)
  if constexpr (PAIR) {                                  // (stage 0 landed in front of the prologue's barrier)
#pragma unroll
    for (int n = 0; n < 4; ++n) wf.w[n] = *reinterpret_cast<const V8*>(sW + wo2 + ((n & 1) * 8 + 2 * (n >> 1)) * 512);
  } else load_w(wf, sW, 0);
  MLP_STAMP_AT(3)

  // ---- chunk hand-over registers: the set holds the 8 B-operand fragments (8 values each) of one chunk, first as
  // pre-activations (bias added, rounded to the operand type: "parked"), then GELU'd IN PLACE, 4 values at a time.
  // (Two alternating sets, with the GELU spread over both phases, do not fit next to the 96 VGPRs of xn: spills.)
  struct HSet { u32x4 u[8]; };
  typedef GeluFit<E> GF;
  typedef ChunkOps<GF::DEG> CO;
  float gx[4], gu[4], gt[4], gp[4];                      // the quad in flight (one at a time: the list is sequential)
  f32x4 gb[2];                                           // bias of the quad being parked / of the next one (read ahead)
  // op K of chunk (bias base cb) on hand-over set hs
  auto gop = [&](HSet& hs, const int cb, auto K_) __attribute__((always_inline)) {
    constexpr int K = decltype(K_)::value;
    if constexpr (K == 0) gb[0] = *reinterpret_cast<const f32x4*>(sB1 + cb + 4 * half);
    else if constexpr (K < CO::NPARK) {
      constexpr int q = (K - 1) / CO::PARK_Q, o = (K - 1) % CO::PARK_Q, i = q >> 2, r0 = 4 * (q & 3);
      if constexpr (o == 0) {
        constexpr int qn = q + 1;
        if constexpr (qn < 16) gb[qn & 1] = *reinterpret_cast<const f32x4*>(sB1 + cb + (qn >> 2) * 32 + 8 * (qn & 3) + 4 * half);
      } else if constexpr (o <= 4) gx[o - 1] = acc1[i][r0 + o - 1] + gb[q & 1][o - 1];
      else hs.u[q >> 1][2 * (q & 1) + (o - 5)] = pack2<E>(gx[2 * (o - 5)], gx[2 * (o - 5) + 1]);
    } else {
      constexpr int kk = K - CO::NPARK, q = kk / CO::GELU_Q, o = kk % CO::GELU_Q, st = o >> 2, e = o & 3, DEG = GF::DEG;
      if constexpr (o >= 4 * (DEG + 5)) {
        constexpr int j = o - 4 * (DEG + 5);
        hs.u[q >> 1][2 * (q & 1) + j] = pack2<E>(gx[2 * j], gx[2 * j + 1]);
      } else if constexpr (st == 0) {
        const uint32_t wd = hs.u[q >> 1][2 * (q & 1) + (e >> 1)];
        gx[e] = unpack1<E>(wd, e & 1);
      } else if constexpr (st == 1) gu[e] = __builtin_amdgcn_fmed3f(gx[e], -GF::L, GF::L);
      else if constexpr (st == 2) gt[e] = gu[e] * gu[e];
      else if constexpr (st == 3) gp[e] = __builtin_fmaf(GF::c(DEG), gt[e], GF::c(DEG - 1));
      else if constexpr (st < 3 + DEG) gp[e] = __builtin_fmaf(gp[e], gt[e], GF::c(DEG - 1 - (st - 3)));
      else if constexpr (st == 3 + DEG) gp[e] = __builtin_fmaf(gu[e], gp[e], 0.5f);
      else gx[e] = gx[e] * gp[e];
    }
  };
  // ops [K0, K1) of a chunk hosted by a phase of NG MFMAs: the share of gap n
  auto host = [&](HSet* hs, const int cb, auto K0_, auto K1_, auto NG_, auto N_) __attribute__((always_inline)) {
    constexpr int K0 = decltype(K0_)::value, K1 = decltype(K1_)::value, NG = decltype(NG_)::value, n = decltype(N_)::value;
    if constexpr (K1 > K0 && !(MLP_DIAG & 1)) {
      constexpr int lo = CO::first_op(K0, K1, NG, n), hi = CO::first_op(K0, K1, NG, n + 1);
      sfor<lo, hi>([&](auto K_) { gop(*hs, cb, K_); });
    }
  };

  // ---- GELU as packed bursts (round 6).  What the measurements say (tools/ubench/pk_burst.hip, mfma_fill.hip; MLP_DIAG ablations,
  // docs/EXPERIMENTS.md 6-2..6-4): behind an MFMA that also carries its W-fragment read + counted wait only ~2 further VALU instructions
  // are free, every other one costs its full ~4.5 issue cycles — the ~700 scalar GELU instructions of a chunk hosted in the gaps cost
  // as much as if no MFMA ran beside them (the hand-over list was 32 % of the kernel's time), while v_pk_*_f32 (two values per
  // instruction, 5.0 cycles, bit-identical arithmetic) cannot be hosted at all (+17 cycles whenever one follows an MFMA).  So: the park
  // ops (accumulator read + bias + round: scalar, ~1.8 per gap) stay hosted behind B(c-1); the GELU proper runs as EIGHT contiguous
  // bursts per chunk — 8 values = 4 independent packed chains each, matrix pipe idle meanwhile — placed between the MFMAs of A(c+1).
  // Same operations on the same values in the same order as the scalar list (gop): results are bit-identical.
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  auto gelu_burst = [&](HSet& hs, auto O_) __attribute__((always_inline)) {
    constexpr int o = decltype(O_)::value, DEG = GF::DEG;
    __builtin_amdgcn_sched_barrier(0);
    f32x2 bx[4], bu[4], bt[4], bp[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint32_t wd = hs.u[o][j];
      bx[j] = f32x2{unpack1<E>(wd, 0), unpack1<E>(wd, 1)};
      bu[j] = f32x2{__builtin_amdgcn_fmed3f(bx[j][0], -GF::L, GF::L), __builtin_amdgcn_fmed3f(bx[j][1], -GF::L, GF::L)};
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) bt[j] = bu[j] * bu[j];
    {
      const float ch = GF::c(DEG), cl = GF::c(DEG - 1);
#pragma unroll
      for (int j = 0; j < 4; ++j) bp[j] = __builtin_elementwise_fma(f32x2{ch, ch}, bt[j], f32x2{cl, cl});
    }
#pragma unroll
    for (int k = DEG - 2; k >= 0; --k) {
      const float ck = GF::c(k);
#pragma unroll
      for (int j = 0; j < 4; ++j) bp[j] = __builtin_elementwise_fma(bp[j], bt[j], f32x2{ck, ck});
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) bp[j] = __builtin_elementwise_fma(bu[j], bp[j], f32x2{0.5f, 0.5f});
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const f32x2 r = bx[j] * bp[j];
      hs.u[o][j] = pack2<E>(r[0], r[1]);
    }
    __builtin_amdgcn_sched_barrier(0);
  };

  // Mode 2 — ONE burst per octet does the whole hand-over: accumulators + bias (fp32) -> GELU -> round to the operand type.  No parked
  // pre-activation (the GELU input is the fp32 sum: one rounding point fewer than rounds 3-5, i.e. what the unfused reference
  // does), no pack / unpack round trip, nothing hosted in the gaps: per octet 8 v_accvgpr_read + 4 v_pk_add + 8 v_med3 + 36 packed
  // + 4 v_cvt_pk = 60 instructions against 57 + 22 hosted.  All eight sit between the MFMAs of B(c-1) — A(c+1) overwrites acc1.
  // The bias of the NEXT octet is read at the end of a burst (gb[]: its LDS latency hides behind the MFMAs that follow).
  auto bias_octet = [&](const int cb, auto O_) __attribute__((always_inline)) {
    constexpr int o = decltype(O_)::value;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      constexpr int dummy = 0; (void)dummy;
      const int q = 2 * o + h;
      gb[h] = *reinterpret_cast<const f32x4*>(sB1 + cb + (q >> 2) * 32 + 8 * (q & 3) + 4 * half);
    }
  };
  auto fused_burst = [&](HSet& hs, const int cb, auto O_) __attribute__((always_inline)) {
    constexpr int o = decltype(O_)::value, DEG = GF::DEG;
    __builtin_amdgcn_sched_barrier(0);
    f32x2 bx[4], bu[4], bt[4], bp[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {                          // pair j of the octet = values 2(j&1), 2(j&1)+1 of quad 2o + (j>>1)
      constexpr int dummy = 0; (void)dummy;
      const int q = 2 * o + (j >> 1), i = q >> 2, r0 = 4 * (q & 3) + 2 * (j & 1);
      const f32x2 av = {acc1[i][r0], acc1[i][r0 + 1]};
      const f32x2 bv = {gb[j >> 1][2 * (j & 1)], gb[j >> 1][2 * (j & 1) + 1]};
      bx[j] = av + bv;
      bu[j] = f32x2{__builtin_amdgcn_fmed3f(bx[j][0], -GF::L, GF::L), __builtin_amdgcn_fmed3f(bx[j][1], -GF::L, GF::L)};
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) bt[j] = bu[j] * bu[j];
    {
      const float ch = GF::c(DEG), cl = GF::c(DEG - 1);
#pragma unroll
      for (int j = 0; j < 4; ++j) bp[j] = __builtin_elementwise_fma(f32x2{ch, ch}, bt[j], f32x2{cl, cl});
    }
#pragma unroll
    for (int k = DEG - 2; k >= 0; --k) {
      const float ck = GF::c(k);
#pragma unroll
      for (int j = 0; j < 4; ++j) bp[j] = __builtin_elementwise_fma(bp[j], bt[j], f32x2{ck, ck});
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) bp[j] = __builtin_elementwise_fma(bu[j], bp[j], f32x2{0.5f, 0.5f});
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const f32x2 r = bx[j] * bp[j];
      hs.u[o][j] = pack2<E>(r[0], r[1]);
    }
    if constexpr (o < (PAIR ? 3 : 7)) bias_octet(cb, std::integral_constant<int, o + 1>{});
    else bias_octet(cb + 128, std::integral_constant<int, 0>{});        // next chunk's first octet (past the last chunk: a harmless read inside sB1 / sB2)
    __builtin_amdgcn_sched_barrier(0);
  };

  // REM = ring stages that follow this one in the panel's stream (compile time, clamped): the stage DMAs stage
  // s+R-1 iff REM >= R-1 and prefetches stage s+1's fragments iff REM >= 1.  Compile-time so that the steady state
  // is one basic block (a scalar branch between two MFMAs is a bubble with one wave per SIMD; see gemm3.hip).
  auto ring_stage = [&](auto REM, auto&& mfma1, auto NOPF) __attribute__((always_inline)) {   // NOPF: do not prefetch stage s+1's first fragments
    constexpr bool more = decltype(REM)::value >= R - 1, next = decltype(REM)::value >= 1 && !decltype(NOPF)::value;
    // (slot offsets opaque: where the stage counter is static — the two-chunk loop body is 24 stages, a multiple of the ring — the
    // compiler would fold slot + fragment offset into constants beyond the 16-bit ds_read offset field: a v_or per read)
    int so = (s & (R - 1)) * MLP_STAGE, son = ((s + 1) & (R - 1)) * MLP_STAGE;
    asm volatile("" : "+s"(so), "+s"(son));
    const char* st = sW + so;
    const char* stn = sW + son;
    // ONE fragment set, refilled in a rolling fashion: right after MFMA (c4, i) has consumed wf.w[i], the same
    // registers receive fragment i of step c4+1 (of stage s+1's step 0 after step 3) — 16 VGPRs instead of 32.  (Two alternating
    // sets with the four reads of a step issued as a group and ONE counted wait per step — 75 instead of 216 s_waitcnt per chunk —
    // measured 2 % slower in the burst form of the kernel and 4 % slower in this one, same box: the satisfied waits are cheap, the
    // read-per-MFMA interleave is what hides the LDS latency.)
    sfor<0, 4>([&](auto C4) {
      constexpr int c4 = decltype(C4)::value;
      if constexpr (c4 == 2) {
        stage_mid(std::integral_constant<bool, (decltype(REM)::value >= R - 2)>{});
      }
      sfor<0, 4>([&](auto I) {
        constexpr int i = decltype(I)::value;
        mfma1(C4, I, wf.w[i]);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (MLP_DIAG & 2) asm volatile("" : "+v"(wf.w[i]));
        else if constexpr (c4 < 3) wf.w[i] = *reinterpret_cast<const V8*>(st + wo + (i * 8 + 2 * (c4 + 1)) * 512);
        else if constexpr (next) wf.w[i] = *reinterpret_cast<const V8*>(stn + wo + (i * 8) * 512);
        if constexpr (more && c4 >= 2 && (i & 1) == 0 && !(MLP_DIAG & 4)) issue_piece_asm(s + R - 1, std::integral_constant<int, (c4 - 2) * 2 + (i >> 1)>{});   // behind the barrier: slot of stage s-1 is free
      });
    });
    ++s;
  };
  constexpr int FAR = 1 << 20;                           // "plenty of stages follow"
  // PAIR: a stage is 8 MFMAs of this wave, n = (k16 step c4 = n >> 1, row block 2p + (n & 1)); the four fragment registers roll with a
  // distance of four MFMAs as above (fragment n + 4: of this stage up to n = 3, of the next one behind the mid-stage barrier at n = 4).
  auto load_w_pair = [&](WF& f, const char* st) __attribute__((always_inline)) {
#pragma unroll
    for (int n = 0; n < 4; ++n) f.w[n] = *reinterpret_cast<const V8*>(st + wo2 + ((n & 1) * 8 + 2 * (n >> 1)) * 512);
  };
  auto ring_stage_pair = [&](auto REM, auto&& mfma1, auto NOPF) __attribute__((always_inline)) {
    constexpr bool more = decltype(REM)::value >= R - 1, next = decltype(REM)::value >= 1 && !decltype(NOPF)::value;
    int so = (s & (R - 1)) * MLP_STAGE, son = ((s + 1) & (R - 1)) * MLP_STAGE;
    asm volatile("" : "+s"(so), "+s"(son));
    const char* st = sW + so;
    const char* stn = sW + son;
    sfor<0, 8>([&](auto N_) {
      constexpr int n = decltype(N_)::value, c4 = n >> 1, i = n & 1;
      if constexpr (n == 4) stage_mid(std::integral_constant<bool, (decltype(REM)::value >= R - 2)>{});
      mfma1(std::integral_constant<int, c4>{}, std::integral_constant<int, i>{}, wf.w[n & 3]);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (MLP_DIAG & 2) asm volatile("" : "+v"(wf.w[n & 3]));
      else if constexpr (n < 4) wf.w[n & 3] = *reinterpret_cast<const V8*>(st + wo2 + (i * 8 + 2 * (c4 + 2)) * 512);
      else if constexpr (next) wf.w[n & 3] = *reinterpret_cast<const V8*>(stn + wo2 + (i * 8 + 2 * (c4 - 2)) * 512);
      if constexpr (more && n >= 4 && !(MLP_DIAG & 4)) issue_piece_asm(s + R - 1, std::integral_constant<int, n - 4>{});   // behind the barrier: slot of stage s-1 is free
    });
    ++s;
  };

  // ---- phase A of a chunk (SA stages x 4 k16 steps x 4 tiles) / phase B ((group g, k half kh) stages; B-operand = fragment 4 kh + c4 of `hs`),
  // each hosting ops [K0, K1) of the chunk with bias base cb on set hd in the issue slots behind its MFMAs
  auto phase_a_h = [&](auto AFTER, HSet* hd, const int cb, auto K0_, auto K1_) __attribute__((always_inline)) {
    sfor<0, SA>([&](auto KS) {
      constexpr int ks = decltype(KS)::value;
      constexpr int rem = decltype(AFTER)::value >= FAR ? FAR : decltype(AFTER)::value + (SA - 1 - ks);
      if constexpr (PAIR) {
        ring_stage_pair(std::integral_constant<int, rem>{}, [&](auto C4, auto I, const V8& wfrag) __attribute__((always_inline)) {
          constexpr int c4 = decltype(C4)::value, i = decltype(I)::value;
          if constexpr (ks == 0 && c4 == 0) {
            const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            acc1[i] = Op16<E>::mfma(wfrag, xf[0], z);
          } else acc1[i] = Op16<E>::mfma(wfrag, xf[ks * 4 + c4], acc1[i]);
        }, std::false_type{});
      } else
      ring_stage(std::integral_constant<int, rem>{}, [&](auto C4, auto I, const V8& wfrag) __attribute__((always_inline)) {
        constexpr int c4 = decltype(C4)::value, i = decltype(I)::value;
        if constexpr (ks == 0 && c4 == 0) {
          const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
          acc1[i] = Op16<E>::mfma(wfrag, xf[0], z);        // the chunk's accumulators start from the instruction's zero operand
        } else acc1[i] = Op16<E>::mfma(wfrag, xf[ks * 4 + c4], acc1[i]);
        if constexpr (MLP_GELU_BURST == 2) {
        } else if constexpr (MLP_GELU_BURST == 1 && !(MLP_DIAG & 1)) {       // K1 > K0: this phase A carries the GELU of the chunk parked in `hd`
          constexpr int n = ks * 16 + c4 * 4 + i, per = SA * 16 / 8;
          if constexpr (decltype(K1_)::value > decltype(K0_)::value && n % per == per / 2) gelu_burst(*hd, std::integral_constant<int, n / per>{});
        } else
        host(hd, cb, K0_, K1_, std::integral_constant<int, SA * 16>{}, std::integral_constant<int, ks * 16 + c4 * 4 + i>{});
      }, std::false_type{});
    });
  };
  auto phase_b_h = [&](auto AFTER, const HSet& hs, HSet* hd, const int cb, auto K0_, auto K1_, auto FIRSTB) __attribute__((always_inline)) {
    constexpr bool zc = decltype(FIRSTB)::value && !KEEP;     // B(0) of a body that does not keep the row: every tile's first MFMA starts from zero
    sfor<0, SB>([&](auto SBI) {
      constexpr int sb = decltype(SBI)::value;
      constexpr int g = sb >> 1, kh = sb & 1;
      constexpr int rem = decltype(AFTER)::value >= FAR ? FAR : decltype(AFTER)::value + (SB - 1 - sb);
      if constexpr (PAIR) {                                // stage sb = output tiles 2 sb, 2 sb + 1 x this wave's k half (its own four fragments)
        static_assert(!PAIR || MLP_GELU_BURST == 2, "mlp: the pair form hands over in fused bursts");
        ring_stage_pair(std::integral_constant<int, rem>{}, [&](auto C4, auto I, const V8& wfrag) __attribute__((always_inline)) {
          constexpr int c4 = decltype(C4)::value, i = decltype(I)::value;
          const V8 hb = __builtin_bit_cast(V8, hs.u[c4]);
          if constexpr (decltype(FIRSTB)::value && ((sb & 1) != PPW || !KEEP) && c4 == 0) {   // B(0): a tile of the partner (tiles 2 sb, 2 sb + 1 are wave (sb & 1)'s), or a part that keeps no row: start from zero
            const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            acc2[2 * sb + i] = Op16<E>::mfma(wfrag, hb, z);
          } else
          acc2[2 * sb + i] = Op16<E>::mfma(wfrag, hb, acc2[2 * sb + i]);
          constexpr int n = sb * 8 + c4 * 2 + i, per = SB * 8 / 4;
          if constexpr (decltype(K1_)::value > decltype(K0_)::value && n % per == per / 2 && !(MLP_DIAG & 1)) fused_burst(*hd, cb, std::integral_constant<int, n / per>{});
        }, std::false_type{});
      } else
      ring_stage(std::integral_constant<int, rem>{}, [&](auto C4, auto I, const V8& wfrag) __attribute__((always_inline)) {
        constexpr int c4 = decltype(C4)::value, i = decltype(I)::value;
        const V8 hb = __builtin_bit_cast(V8, hs.u[4 * kh + c4]);
        if constexpr (zc && kh == 0 && c4 == 0) {
          const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
          acc2[4 * g + i] = Op16<E>::mfma(wfrag, hb, z);
        } else
        acc2[4 * g + i] = Op16<E>::mfma(wfrag, hb, acc2[4 * g + i]);
        if constexpr (MLP_GELU_BURST == 2) {
          constexpr int n = sb * 16 + c4 * 4 + i, per = SB * 16 / 8;
          if constexpr (decltype(K1_)::value > decltype(K0_)::value && n % per == per / 2 && !(MLP_DIAG & 1)) fused_burst(*hd, cb, std::integral_constant<int, n / per>{});
        } else
        host(hd, cb, K0_, K1_, std::integral_constant<int, SB * 16>{}, std::integral_constant<int, sb * 16 + c4 * 4 + i>{});
      }, std::false_type{});
    });
  };

  if constexpr (PROJ) {
    // ---- projection: outT[D x 32 tok] = x + bias + Wpp . a^T, (group g, k stage) ring stages; B-operand = attention fragments.
    // The accumulators of a group start at its rows + bias (first group: requested in the prologue; the others: requested
    // below, landed under the first group's MFMAs).  The new row y never leaves the accumulators: LayerNorm reads it there,
    // fc2 accumulates on top of it (+ bias2), the epilogue stores it.  Of a split tail panel only part 0 keeps the row (the
    // reduction then adds no residual).
    sfor<0, OG>([&](auto G_) {
      constexpr int g = decltype(G_)::value;
      if constexpr (g > 0 && g < 3) { MLP_STAMP_AT(16 + g) }   // (slots 17, 18: start of output groups 1, 2)
      if constexpr (MLP_ROWS_LATE && g + 1 < OG) {         // the next group's rows: requested now, landed under this group's MFMAs
        rows_to_acc(std::integral_constant<int, 4 * (g + 1)>{}, std::integral_constant<int, 4 * (g + 2)>{});
        __builtin_amdgcn_sched_barrier(0);
      }
      sfor<0, SA>([&](auto KS) {
        constexpr int ks = decltype(KS)::value;
        if constexpr (PAIR) {                              // the wave's two tiles of the group: row blocks 2p, 2p + 1 of the stage, 8 MFMAs
          constexpr int PP = PPW;
          ring_stage_pair(std::integral_constant<int, FAR>{}, [&](auto C4, auto I, const V8& wfrag) __attribute__((always_inline)) {
            constexpr int c4 = decltype(C4)::value, i = decltype(I)::value;
            acc2[4 * g + 2 * PP + i] = Op16<E>::mfma(wfrag, xf[ks * 4 + c4], acc2[4 * g + 2 * PP + i]);
          }, std::integral_constant<bool, (g == OG - 1 && ks == SA - 1)>{});
        } else
        ring_stage(std::integral_constant<int, FAR>{}, [&](auto C4, auto I, const V8& wfrag) __attribute__((always_inline)) {
          constexpr int c4 = decltype(C4)::value, i = decltype(I)::value;
          acc2[4 * g + i] = Op16<E>::mfma(wfrag, xf[ks * 4 + c4], acc2[4 * g + i]);
        }, std::integral_constant<bool, (g == OG - 1 && ks == SA - 1)>{});   // nothing of the ring held in registers across the LayerNorm
      });
    });
    MLP_STAMP_AT(4)
    if constexpr (PAIR) {
      constexpr int PP = PPW;
      sfor<0, OT>([&](auto T_) { if constexpr (((decltype(T_)::value & 3) >> 1) == PP) bias_mm(sBp, T_); });
      layernorm_to_xf_pair();
      if constexpr (!PARTIAL) sfor<0, OT>([&](auto T_) { if constexpr (((decltype(T_)::value & 3) >> 1) == PP) bias_mm(sB2, T_); });   // each tile's bias2 once: in its owner (pair parts: the reduction adds it)
    } else {
    sfor<0, OT>([&](auto T_) { bias_mm(sBp, T_); });
    layernorm_to_xf();
    if constexpr (!PARTIAL && KEEP) sfor<0, OT>([&](auto T_) { bias_mm(sB2, T_); });   // (after the statistics: the LayerNorm is of the row without bias2; split parts: the reduction adds it)
    }
    if constexpr (PAIR) load_w_pair(wf, sW + (s & (R - 1)) * MLP_STAGE);
    else load_w(wf, sW + (s & (R - 1)) * MLP_STAGE, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i)                          // (not before: the 64 registers are free for the compiler up to here)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc1[i][r] = 0.f;
  }

  MLP_STAMP_AT(5)
  // Schedule (ring stream order A(0) | A(1) | B(0) | A(2) | B(1) | ...).  ONE hand-over set: parked after A(c) — once B(c-1)
  // has consumed the previous contents — activated in place under A(c+1), consumed by B(c):
  //   A(0) park | {A(c)+gelu(c-1) | B(c-1) | park(c)} c=1..NC-1 | gelu(NC-1) | B(NC-1)
  static_assert(NC >= 3 || (NC == 2 && MLP_GELU_BURST == 2), "mlp: at least three hidden chunks (two: fused-burst hand-over only)");
  typedef std::integral_constant<int, FAR> Far;
  // Schedule, same ring stream order.  TWO hand-over sets (chunk c uses set c & 1): while B(c-1) consumes one, the ops of chunk c
  // fill the other — its park ops and the first part of its GELU behind B(c-1)'s MFMAs, the rest behind A(c+1)'s:
  //   A(0) | park(0) | A(1)+gelu(0) | { B(c-1)+ops(c)[0,KB) | A(c+1)+ops(c)[KB,N) } c=1..NC-2 | B(NC-2)+ops(NC-1)[0,KB) | ops(NC-1)[KB,N) | B(NC-1)
  // (only the first chunk's park and the last chunk's second part run without MFMAs beside them)
  constexpr int KB = MLP_GELU_BURST ? CO::NPARK : CO::split(SB * 16, SA * 16);   // (bursts: B(c-1) hosts exactly the park ops)
  typedef std::integral_constant<int, 0> K0_; typedef std::integral_constant<int, CO::NPARK> KP_;
  typedef std::integral_constant<int, KB> KB_; typedef std::integral_constant<int, CO::N> KN_;
  HSet S2[2];
  const int cb0 = c0 * 128 + (PAIR ? 64 * pp : 0);        // (PAIR: the wave's two hidden tiles of a chunk)
  constexpr int NOCT = PAIR ? 4 : 8;                     // octets (B-operand fragments) per chunk and wave
  if constexpr (NC == 2) {
    // Two chunks per workgroup (round 6: the 6-way split of calls of <= 27 crops — 12 chunks over six workgroups per panel):
    //   A(0) | hand-over(0) | A(1) | B(0) + hand-over(1) | B(1)        (ring stream A(0) | A(1) | B(0) | B(1), as stage_src deals it)
    phase_a_h(Far{}, nullptr, 0, K0_{}, K0_{});                             // A(0): 3 phases = 18 stages follow
    if constexpr (!(MLP_DIAG & 1)) {
      bias_octet(cb0, std::integral_constant<int, 0>{});
      sfor<0, NOCT>([&](auto O_) { fused_burst(S2[0], cb0, O_); });
    }
    phase_a_h(std::integral_constant<int, 2 * SB>{}, nullptr, 0, K0_{}, K0_{});                                   // A(1)
    phase_b_h(std::integral_constant<int, SB>{}, S2[0], &S2[1], cb0 + 128, K0_{}, KB_{}, std::true_type{});      // B(0) + hand-over(1)
    phase_b_h(std::integral_constant<int, 0>{}, S2[1], nullptr, 0, K0_{}, K0_{}, std::false_type{});              // B(1)
  } else {
  phase_a_h(Far{}, nullptr, 0, K0_{}, K0_{});                               // A(0)
  if constexpr (MLP_GELU_BURST == 2) {
    if constexpr (!(MLP_DIAG & 1)) {
      bias_octet(cb0, std::integral_constant<int, 0>{});
      sfor<0, NOCT>([&](auto O_) { fused_burst(S2[0], cb0, O_); });         // hand-over(0): the only one without MFMAs around it
    }
  } else if constexpr (!(MLP_DIAG & 1))
  sfor<0, CO::NPARK>([&](auto K_) { gop(S2[0], cb0, K_); });              // park(0)
  MLP_STAMP_AT(6)
  phase_a_h(Far{}, &S2[0], cb0, KP_{}, KN_{});                            // A(1) + gelu(0)
  auto pair_step = [&](auto PAR, auto AFTER_A, const int c, auto FIRSTB) __attribute__((always_inline)) {   // chunk c, c & 1 == PAR; FIRSTB: c == 1 (its phase B is B(0))
    constexpr int par = decltype(PAR)::value;
    constexpr int after_b = decltype(AFTER_A)::value >= FAR ? FAR : decltype(AFTER_A)::value + SA;
    phase_b_h(std::integral_constant<int, after_b>{}, S2[par ^ 1], &S2[par], cb0 + c * 128, K0_{}, KB_{}, FIRSTB);   // B(c-1) + ops(c)[0, KB)
    phase_a_h(AFTER_A, &S2[par], cb0 + c * 128, KB_{}, KN_{});                                                // A(c+1) + ops(c)[KB, N)
  };
  {
    int c = 1;                                           // chunks 1 .. NC-3 rolled in pairs (static set roles), chunk NC-2 peeled (static stage counts)
    if constexpr (KEEP && !PAIR) {
#pragma unroll 1
      for (; c + 1 <= NC - 3; c += 2) {
        pair_step(std::integral_constant<int, 1>{}, Far{}, c, std::false_type{});
        pair_step(std::integral_constant<int, 0>{}, Far{}, c + 1, std::false_type{});
      }
      if constexpr ((NC - 3) % 2 == 1) pair_step(std::integral_constant<int, 1>{}, Far{}, NC - 3, std::false_type{});
    } else {                                             // chunk 1 peeled too: its phase B is B(0), whose MFMAs start the accumulators from zero
      if constexpr (NC - 3 >= 1) {
        pair_step(std::integral_constant<int, 1>{}, Far{}, 1, std::true_type{});
        c = 2;
#pragma unroll 1
        for (; c + 1 <= NC - 3; c += 2) {
          pair_step(std::integral_constant<int, 0>{}, Far{}, c, std::false_type{});
          pair_step(std::integral_constant<int, 1>{}, Far{}, c + 1, std::false_type{});
        }
        if constexpr ((NC - 3) >= 2 && (NC - 4) % 2 == 1) pair_step(std::integral_constant<int, 0>{}, Far{}, NC - 3, std::false_type{});
      }
    }
  }
  MLP_STAMP_AT(7)
  pair_step(std::integral_constant<int, (NC - 2) & 1>{}, std::integral_constant<int, 2 * SB>{}, NC - 2, std::integral_constant<bool, NC == 3>{});      // B(NC-3) | A(NC-1)
  phase_b_h(std::integral_constant<int, SB>{}, S2[(NC - 2) & 1], &S2[(NC - 1) & 1], cb0 + (NC - 1) * 128, K0_{}, KB_{}, std::false_type{});   // B(NC-2)
  if constexpr (MLP_GELU_BURST == 2) {
  } else if constexpr (MLP_GELU_BURST == 1 && !(MLP_DIAG & 1)) sfor<0, 8>([&](auto O_) { gelu_burst(S2[(NC - 1) & 1], O_); });
  else if constexpr (!(MLP_DIAG & 1))
  sfor<KB, CO::N>([&](auto K_) { gop(S2[(NC - 1) & 1], cb0 + (NC - 1) * 128, K_); });
  phase_b_h(std::integral_constant<int, 0>{}, S2[(NC - 1) & 1], nullptr, 0, K0_{}, K0_{}, std::false_type{});  // B(NC-1)
  }
  MLP_STAMP_AT(8)
  // ---- epilogue.  lane = token r31 of row block rb; registers 4q..4q+3 of tile t = fp32 chunk cq(t, q) of the row (W2's
  // rows are permuted per 32: api.hip rowperm32).  Whole panels: acc2 already holds x + bias2 + fc2 — nothing is re-read.
  int hf = half;
  asm volatile("" : "+v"(hf));                           // opaque: else the LayerNorm's 48 chunk offsets are kept (spilled) across the main loop for this
  auto cq = [&](int t, int q) __attribute__((always_inline)) { return 8 * t + 4 * (q >> 1) + 2 * hf + (q & 1); };
  if constexpr (PAIR) {
    // The two partial sums of a token tile meet in LDS, and each wave of the pair finishes HALF of the tile's output features: tiles
    // 4g + 2p, 4g + 2p + 1 are wave p's ("own").  The ring is drained (every stage consumed): a wave parks the 16 x 6 values per lane of
    // the tiles it does not own as [tile][quad][lane] x 16 B (the pair's 48 KB slab holds each tile once), adds its partner's for its
    // own tiles, stores their part of x and — the row statistics exchanged through 1 KB behind the slabs — of the second output.
    constexpr int PP = PPW;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                        // nobody still reads the last stages
    asm volatile("" ::: "memory");
    char* px = smem + (w >> 1) * (OT * 4 * 1024) + lane * 16;
    sfor<0, OT>([&](auto T_) {
      constexpr int t = decltype(T_)::value;
      if constexpr (((t & 3) >> 1) != PP) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 o = {acc2[t][4 * q], acc2[t][4 * q + 1], acc2[t][4 * q + 2], acc2[t][4 * q + 3]};
          *reinterpret_cast<f32x4*>(px + (t * 4 + q) * 1024) = o;
        }
      }
    });
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const bool live = rb * 32 + r31 < a.M;
    char* xr = reinterpret_cast<char*>(a.x) + rb * (D / 4) * 512 + r31 * 16;
    // (pair parts: the own tiles' sums go to the scratch [part][row block][D/4 chunks][32][16 B]; together the two waves write the whole row)
    char* pr = reinterpret_cast<char*>(a.partial) + (((int64_t)(bid % SPLIT) * a.tail_rb + (rb - (int64_t)a.panel0 * 2)) * (D / 4)) * 512 + r31 * 16;
    float sm = 0.f;
    sfor<0, OT>([&](auto T_) {
      constexpr int t = decltype(T_)::value;
      if constexpr (((t & 3) >> 1) == PP) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(px + (t * 4 + q) * 1024);
#pragma unroll
          for (int e = 0; e < 4; ++e) acc2[t][4 * q + e] += v[e];
          const f32x4 o = {acc2[t][4 * q], acc2[t][4 * q + 1], acc2[t][4 * q + 2], acc2[t][4 * q + 3]};
          sm += (o[0] + o[1]) + (o[2] + o[3]);
          if constexpr (PARTIAL) *reinterpret_cast<f32x4*>(pr + (size_t)cq(t, q) * 512) = o;
          else if (live) st_act<4>(reinterpret_cast<f32x4*>(xr + (size_t)cq(t, q) * 512), o);
        }
      }
    });
    MLP_STAMP_AT(9)
    if (!PARTIAL && a.xn_out) {                                      // (uniform: every wave passes the two barriers below)
      float* sS = reinterpret_cast<float*>(smem + 2 * (OT * 4 * 1024));   // [sum | sum of squares][wave][token]
      sm += __shfl_xor(sm, 32, 64);
      if (half == 0) sS[w * 32 + r31] = sm;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      sm += sS[(w ^ 1) * 32 + r31];                      // (own + partner's: the same sum in both waves)
      const float mean = sm * (1.0f / D);
      f32x2 q2 = {0.f, 0.f};
      const f32x2 nm = {-mean, -mean};
      sfor<0, OT>([&](auto T_) {
        constexpr int t = decltype(T_)::value;
        if constexpr (((t & 3) >> 1) == PP) {
#pragma unroll
          for (int r = 0; r < 16; r += 2) {
            const f32x2 d = f32x2{acc2[t][r], acc2[t][r + 1]} + nm;
            q2 = __builtin_elementwise_fma(d, d, q2);
          }
        }
      });
      float ss = q2[0] + q2[1];
      ss += __shfl_xor(ss, 32, 64);
      if (half == 0) sS[128 + w * 32 + r31] = ss;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      ss += sS[128 + (w ^ 1) * 32 + r31];
      const float rstd = 1.0f / sqrtf(ss * (1.0f / D) + a.eps);
      const f32x2 r2 = {rstd, rstd}, nm2 = {-mean, -mean};
      char* nr = static_cast<char*>(a.xn_out) + rb * (D / 8) * 512 + r31 * 16;
      sfor<0, OT>([&](auto T_) {
        constexpr int t = decltype(T_)::value;
        if constexpr (((t & 3) >> 1) == PP) {
#pragma unroll
          for (int p = 0; p < 2; ++p) {
            u32x2 pk[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              const int q = 2 * p + j, c = cq(t, q);
              const f32x4 gm = *reinterpret_cast<const f32x4*>(sGn + c * 4);
              const f32x4 bt = *reinterpret_cast<const f32x4*>(sBn + c * 4);
              const f32x2 o0 = __builtin_elementwise_fma((f32x2{acc2[t][4 * q], acc2[t][4 * q + 1]} + nm2) * r2, f32x2{gm[0], gm[1]}, f32x2{bt[0], bt[1]});
              const f32x2 o1 = __builtin_elementwise_fma((f32x2{acc2[t][4 * q + 2], acc2[t][4 * q + 3]} + nm2) * r2, f32x2{gm[2], gm[3]}, f32x2{bt[2], bt[3]});
              pk[j] = pack4<E>(o0[0], o0[1], o1[0], o1[1]);
            }
            const u32x4 o = {pk[0][0], pk[0][1], pk[1][0], pk[1][1]};
            if (live) st_act<8>(reinterpret_cast<u32x4*>(nr + (size_t)(4 * t + 2 * p + hf) * 512), o);
          }
        }
      });
    }
  }
  if constexpr (PARTIAL && !PAIR) {
    // fp32 partial sums -> scratch [part][tail row block][D/4 chunks][32][16 B]
    const int64_t rbl = rb - (int64_t)a.panel0 * 4;
    char* pr = reinterpret_cast<char*>(a.partial) + (((int64_t)(bid % SPLIT) * a.tail_rb + rbl) * (D / 4)) * 512 + r31 * 16;
    sfor<0, OT>([&](auto T_) {
      constexpr int t = decltype(T_)::value;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 o = {acc2[t][4 * q], acc2[t][4 * q + 1], acc2[t][4 * q + 2], acc2[t][4 * q + 3]};
        *reinterpret_cast<f32x4*>(pr + (size_t)cq(t, q) * 512) = o;
      }
    });
  } else if (!PAIR && !PARTIAL && rb * 32 + r31 < a.M) {
    char* xr = reinterpret_cast<char*>(a.x) + rb * (D / 4) * 512 + r31 * 16;
    float sm = 0.f;
    sfor<0, OT>([&](auto T_) {
      constexpr int t = decltype(T_)::value;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 o = {acc2[t][4 * q], acc2[t][4 * q + 1], acc2[t][4 * q + 2], acc2[t][4 * q + 3]};
        sm += (o[0] + o[1]) + (o[2] + o[3]);
        st_act<4>(reinterpret_cast<f32x4*>(xr + (size_t)cq(t, q) * 512), o);
      }
    });
    MLP_STAMP_AT(9)
    if (a.xn_out) {
      // ---- second output.  The lane pair (r31, half 0 / 1) holds the whole new row: two-pass statistics (one cross-half
      // exchange each) and the next block's norm1 applied on the way out, rounded to the operand type (same arithmetic as
      // layernorm_blocked_kernel).  Registers 8p..8p+7 of tile t = 16-bit chunk 4t + 2p + half: one 16-byte store.
      sm += __shfl_xor(sm, 32, 64);
      const float mean = sm * (1.0f / D);
      float ss = 0.f;
      if constexpr (MLP_LN_PK) {
        f32x2 q2 = {0.f, 0.f};
        const f32x2 nm = {-mean, -mean};
        sfor<0, OT>([&](auto T_) {
          constexpr int t = decltype(T_)::value;
#pragma unroll
          for (int r = 0; r < 16; r += 2) {
            const f32x2 d = f32x2{acc2[t][r], acc2[t][r + 1]} + nm;
            q2 = __builtin_elementwise_fma(d, d, q2);
          }
        });
        ss = q2[0] + q2[1];
      } else
      sfor<0, OT>([&](auto T_) {
        constexpr int t = decltype(T_)::value;
#pragma unroll
        for (int r = 0; r < 16; ++r) { const float d = acc2[t][r] - mean; ss += d * d; }
      });
      ss += __shfl_xor(ss, 32, 64);
      const float rstd = 1.0f / sqrtf(ss * (1.0f / D) + a.eps);
      const f32x2 r2 = {rstd, rstd}, nm2 = {-mean, -mean};
      char* nr = static_cast<char*>(a.xn_out) + rb * (D / 8) * 512 + r31 * 16;
      sfor<0, OT>([&](auto T_) {
        constexpr int t = decltype(T_)::value;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          u32x2 pk[2];
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const int q = 2 * p + j, c = cq(t, q);       // fp32 chunk c = features 4c..4c+3 = half j of 16-bit chunk c >> 1
            const f32x4 gm = *reinterpret_cast<const f32x4*>(sGn + c * 4);
            const f32x4 bt = *reinterpret_cast<const f32x4*>(sBn + c * 4);
            if constexpr (MLP_LN_PK) {
              const f32x2 o0 = __builtin_elementwise_fma((f32x2{acc2[t][4 * q], acc2[t][4 * q + 1]} + nm2) * r2, f32x2{gm[0], gm[1]}, f32x2{bt[0], bt[1]});
              const f32x2 o1 = __builtin_elementwise_fma((f32x2{acc2[t][4 * q + 2], acc2[t][4 * q + 3]} + nm2) * r2, f32x2{gm[2], gm[3]}, f32x2{bt[2], bt[3]});
              pk[j] = pack4<E>(o0[0], o0[1], o1[0], o1[1]);
            } else
            pk[j] = pack4<E>((acc2[t][4 * q] - mean) * rstd * gm[0] + bt[0], (acc2[t][4 * q + 1] - mean) * rstd * gm[1] + bt[1],
                             (acc2[t][4 * q + 2] - mean) * rstd * gm[2] + bt[2], (acc2[t][4 * q + 3] - mean) * rstd * gm[3] + bt[3]);
          }
          const u32x4 o = {pk[0][0], pk[0][1], pk[1][0], pk[1][1]};
          st_act<8>(reinterpret_cast<u32x4*>(nr + (size_t)(4 * t + 2 * p + hf) * 512), o);
        }
      });
    }
  }
  MLP_STAMP_AT(10)
#ifdef MLP_STAMP
  if constexpr (PARTIAL == MLP_STAMP_PART) { mlp_stamps[(bid & (MLP_STAMP_WGS - 1)) * MLP_STAMP_N + 12] = vm_wait; mlp_stamps[(bid & (MLP_STAMP_WGS - 1)) * MLP_STAMP_N + 13] = bar_wait; }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  MLP_STAMP_AT(11)
#ifdef MLP_STAMP
  if (PARTIAL == MLP_STAMP_PART) mlp_stamps[(bid & (MLP_STAMP_WGS - 1)) * MLP_STAMP_N + 16] = __builtin_amdgcn_s_memrealtime();
#endif
}

// One launch for the whole panels (workgroups [0, main_wgs)) AND the split parts of the tail panels (TNCW hidden chunks each; TNCW = 0:
// no tail): the parts are dispatched last, i.e. into the ragged end that the start spread of the first round leaves on the CUs.
template <typename E, int D, int H, int TNCW, bool PROJ>
__global__ __launch_bounds__(256, 1) void mlp_fused_kernel(MlpArgs a) {
  if constexpr (TNCW == 0) {
    mlp_fused_body<E, D, H, H / 128, false, PROJ>(a, (int)blockIdx.x);
  } else {
    if ((int)blockIdx.x < a.main_wgs) mlp_fused_body<E, D, H, H / 128, false, PROJ>(a, (int)blockIdx.x);
    else {
      const int b = (int)blockIdx.x - a.main_wgs;
      if constexpr (PROJ) {                              // part 0 of a panel keeps the new row, the others start at zero (KEEP)
        if (b % ((H / 128) / TNCW) == 0) mlp_fused_body<E, D, H, TNCW, true, true, false, true>(a, b);
        else mlp_fused_body<E, D, H, TNCW, true, true, false, false>(a, b);
      } else mlp_fused_body<E, D, H, TNCW, true, PROJ>(a, b);
    }
  }
}

// 64-token panels, wave pairs (PAIR above): one workgroup per two row blocks
// TNCW = 0: the whole MLP of a 64-token panel per workgroup; TNCW > 0 ("pair parts"): (H / 128) / TNCW workgroups per panel with TNCW hidden
// chunks each, fp32 partial sums to the scratch, reduced by the reduction + LayerNorm launch like the 128-token parts'
template <typename E, int D, int H, int TNCW>
__global__ __launch_bounds__(256, 1) void mlp_pair_kernel(MlpArgs a) {
  const bool w1 = (wave_id() & 1) != 0;                    // (every body passes the same barriers)
  if constexpr (TNCW == 0) {
    if (!w1) mlp_fused_body<E, D, H, H / 128, false, true, true, true, 0>(a, (int)blockIdx.x);
    else mlp_fused_body<E, D, H, H / 128, false, true, true, true, 1>(a, (int)blockIdx.x);
  } else {
    const bool keep = (int)blockIdx.x % ((H / 128) / TNCW) == 0;   // part 0 of a panel keeps the new row (in the tiles each of its waves owns)
    if (keep) {
      if (!w1) mlp_fused_body<E, D, H, TNCW, true, true, true, true, 0>(a, (int)blockIdx.x);
      else mlp_fused_body<E, D, H, TNCW, true, true, true, true, 1>(a, (int)blockIdx.x);
    } else {
      if (!w1) mlp_fused_body<E, D, H, TNCW, true, true, true, false, 0>(a, (int)blockIdx.x);
      else mlp_fused_body<E, D, H, TNCW, true, true, true, false, 1>(a, (int)blockIdx.x);
    }
  }
}

// x[row block rb0 + i] = (add_x ? x : 0) + bias2 + sum over parts (fixed order) of the partial outputs; one thread per 16-byte chunk slot
__global__ __launch_bounds__(256) void mlp_reduce_kernel(float* x, const float* partial, const float* b2, int64_t rb0, int tail_rb,
                                                         int D, int nparts, int64_t M, int add_x) {
  const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;          // (row block, chunk, row) in blocked order
  const int64_t per_rb = (int64_t)(D / 4) * 32;
  if (id >= (int64_t)tail_rb * per_rb) return;
  const int64_t rbl = id / per_rb;
  const int rem = (int)(id - rbl * per_rb), chunk = rem >> 5, row = rem & 31;
  if ((rb0 + rbl) * 32 + row >= M) return;
  f32x4* xp = reinterpret_cast<f32x4*>(x) + (rb0 + rbl) * per_rb + rem;
  f32x4 v = *xp;
  const f32x4 bv = *reinterpret_cast<const f32x4*>(b2 + chunk * 4);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int p = 0; p < nparts; ++p) {
    const f32x4 pv = reinterpret_cast<const f32x4*>(partial)[((int64_t)p * tail_rb + rbl) * per_rb + rem];
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[e] += pv[e];
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] = acc[e] + bv[e] + (add_x ? v[e] : 0.f);   // with the projection in front part 0 carries the whole new row
  *xp = v;
}

template <typename E>
int launch_mlp_pair(const MlpArgs& a, int tncw, unsigned grid, hipStream_t s) {
  switch (tncw) {
    case 0: hipLaunchKernelGGL((mlp_pair_kernel<E, 384, 1536, 0>), dim3(grid), dim3(256), 0, s, a); return check_launch("mlp_pair");
    case 2: hipLaunchKernelGGL((mlp_pair_kernel<E, 384, 1536, 2>), dim3(grid), dim3(256), 0, s, a); break;
    case 4: hipLaunchKernelGGL((mlp_pair_kernel<E, 384, 1536, 4>), dim3(grid), dim3(256), 0, s, a); break;
    case 6: hipLaunchKernelGGL((mlp_pair_kernel<E, 384, 1536, 6>), dim3(grid), dim3(256), 0, s, a); break;
    default: return fail(EFFOCR_EINVAL, "mlp_pair: hidden chunks per part must be 0 (whole), 2, 4 or 6");
  }
  return check_launch("mlp_pair_parts");
}

template <typename E, bool PROJ>
int launch_mlp(const MlpArgs& a_in, hipStream_t s) {
  MlpArgs a = a_in;
  for (const void* q : {(const void*)a.b1, (const void*)a.b2, (const void*)a.gamma, (const void*)a.beta, (const void*)a.bp, (const void*)a.gamma_n, (const void*)a.beta_n})
    if ((reinterpret_cast<uintptr_t>(q) & 15) != 0) return fail(EFFOCR_EINVAL, "mlp_fused: bias / LayerNorm parameter arrays must be 16-byte aligned (they reach LDS by 16-byte DMA)");
  const int npanels = (a.M + MLP_PT - 1) / MLP_PT;
  if (a.D == 128 && a.H == 512) {
    a.panel0 = 0; a.main_wgs = npanels; a.stagger_wgs = 0;
    hipLaunchKernelGGL((mlp_fused_kernel<E, 128, 512, 0, PROJ>), dim3((unsigned)npanels), dim3(256), 0, s, a);
    return check_launch("mlp_fused");
  }
  if (!(a.D == 384 && a.H == 1536)) return fail(EFFOCR_EUNSUPPORTED, "mlp_fused: (D, H) must be (384, 1536) or (128, 512)");
  // One workgroup per CU: the panels of the last, partially filled round are cut along the hidden dimension into
  // 4 (or 2) workgroups each, which write partial outputs to the caller's scratch; a small kernel reduces them.
  const int slots = device_cus();
  if constexpr (PROJ) {
    // Calls of 30-83 crops (47+ 128-token panels, 64-token panels within one round): 64-token panels on wave pairs — no partial sums in
    // HBM, no reduction launch.  tools/pair_sweep.py, same box, final form (projection, both LayerNorms and the stores split between the
    // two waves of a pair): 40 / 48 / 64 / 80 crops 0.88 / 0.87 / 0.86 / 0.84 of the split parts' call time, 32 crops 0.97, 28 crops 1.00,
    // 24 crops 1.00, 16 crops 1.07 (there the 4- / 6-way parts win: a sixth of the weight stream per CU).
    const int np64 = (a.M + 63) / 64;
    const bool fits = np64 <= slots && a.rows_alloc >= np64 * 64;
    // Calls of <= 41 crops: "pair parts" — the 64-token panels' hidden chunks dealt over as many workgroups as fit one round: 6 (<= 13 crops),
    // 3 (<= 27) or 2 (<= 36 crops = 112 pair panels: beyond, the whole-pair form below is faster — same box, encoder only, 2-way parts vs whole pair
    // panels: 28 / 32 / 36 / 40 crops 0.751 / 0.774 / 0.809 / 0.853 vs 0.819 / 0.808 / 0.814 / 0.824 ms) instead of 128-token panels dealt 6- / 4-way:
    // half the projection / LayerNorm per wave, half the partial sums for the reduction launch.
    // (<= 13 crops: 6-way, two chunks per workgroup — the parts' fixed share is small enough now for the shorter chunk phase to pay)
    const int pparts = (np64 * 6 <= slots && !a.no_split6) ? 6 : np64 * 3 <= slots ? 3 : 2;
    if (fits && (pparts > 2 || np64 * 16 <= slots * 7) && a.pair == 0 && !a.no_pair_parts && !a.no_tail_split && a.partial &&
        a.partial_bytes >= (size_t)pparts * np64 * 64 * a.D * sizeof(float)) {
      a.panel0 = 0; a.main_wgs = 0; a.stagger_wgs = 0; a.tail_rb = np64 * 2;
      const int prec = std::is_same<E, __bf16>::value ? PREC_BF16 : PREC_FP16;
      int rc = prec == PREC_BF16 ? mlp_pair_launch_bf16(a, 12 / pparts, (unsigned)(np64 * pparts), s) : mlp_pair_launch_f16(a, 12 / pparts, (unsigned)(np64 * pparts), s);
      if (rc) return rc;
      if (a.xn_out && MLP_FUSED_REDUCE_LN)
        return reduce_layernorm_rows_blocked(prec, a.x, (int64_t)a.M, a.D, a.partial, a.b2_logical, pparts, a.tail_rb, 0, a.gamma_n, a.beta_n, a.eps, a.xn_out, s);
      const int64_t slots4 = (int64_t)a.tail_rb * (a.D / 4) * 32;
      hipLaunchKernelGGL(mlp_reduce_kernel, dim3((unsigned)((slots4 + 255) / 256)), dim3(256), 0, s, a.x, a.partial, a.b2_logical, (int64_t)0, a.tail_rb, a.D, pparts, (int64_t)a.M, 0);
      rc = check_launch("mlp_reduce");
      if (rc || !a.xn_out) return rc;
      return layernorm_rows_blocked(prec, a.x, (int64_t)a.M, a.D, a.gamma_n, a.beta_n, a.eps, a.xn_out, s);
    }
    if (fits && (a.pair > 0 || (a.pair == 0 && !a.no_tail_split && npanels * 11 > slots * 2))) {
      a.panel0 = 0; a.main_wgs = np64; a.stagger_wgs = 0; a.tail_rb = 0;
      return std::is_same<E, __bf16>::value ? mlp_pair_launch_bf16(a, 0, (unsigned)np64, s) : mlp_pair_launch_f16(a, 0, (unsigned)np64, s);
    }
  }
  const int tail = a.no_tail_split ? 0 : npanels % slots;
  int split = 1;
  for (const int cand : {6, 4, 2}) {                     // 6: calls of <= 27 crops — six two-chunk parts per panel (16 crops: 0.89 -> 0.8 ms)
    if (cand == 6 && a.no_split6) continue;
    if (tail > 0 && tail * cand <= slots && a.partial && a.partial_bytes >= (size_t)cand * tail * MLP_PT * a.D * sizeof(float)) { split = cand; break; }
  }
  const int main_panels = split > 1 ? npanels - tail : npanels;
  a.panel0 = main_panels;                                // first split panel
  a.main_wgs = main_panels;
  a.tail_rb = tail * 4;
  a.stagger_wgs = main_panels >= (a.stagger_min_rounds > 0 ? a.stagger_min_rounds : 4) * slots ? slots : 0;  // (the spread costs ~0.4 panel times at the end of the launch; measured worth it from 2 rounds on: api.hip)
  const dim3 grid((unsigned)(main_panels + (split > 1 ? tail * split : 0)));
  if (split == 6) hipLaunchKernelGGL((mlp_fused_kernel<E, 384, 1536, 2, PROJ>), grid, dim3(256), 0, s, a);
  else if (split == 4) hipLaunchKernelGGL((mlp_fused_kernel<E, 384, 1536, 3, PROJ>), grid, dim3(256), 0, s, a);
  else if (split == 2) hipLaunchKernelGGL((mlp_fused_kernel<E, 384, 1536, 6, PROJ>), grid, dim3(256), 0, s, a);
  else hipLaunchKernelGGL((mlp_fused_kernel<E, 384, 1536, 0, PROJ>), grid, dim3(256), 0, s, a);
  int rc = check_launch("mlp_fused");
  if (rc || split == 1) return rc;
  const int64_t rb0 = (int64_t)main_panels * 4;          // first row block of the split panels
  const int prec = std::is_same<E, __bf16>::value ? PREC_BF16 : PREC_FP16;
  if (a.xn_out && MLP_FUSED_REDUCE_LN) {
    // the split panels' reduction (parts in a fixed order + bias2 [+ x]) and the second output in ONE launch
    return reduce_layernorm_rows_blocked(prec, reinterpret_cast<float*>(reinterpret_cast<char*>(a.x) + rb0 * (a.D / 4) * 512), (int64_t)a.M - rb0 * 32, a.D,
                                         a.partial, a.b2_logical, split, a.tail_rb, PROJ ? 0 : 1, a.gamma_n, a.beta_n, a.eps,
                                         static_cast<char*>(a.xn_out) + rb0 * (a.D / 8) * 512, s);
  }
  const int64_t slots4 = (int64_t)a.tail_rb * (a.D / 4) * 32;
  hipLaunchKernelGGL(mlp_reduce_kernel, dim3((unsigned)((slots4 + 255) / 256)), dim3(256), 0, s, a.x, a.partial, a.b2_logical,
                     rb0, a.tail_rb, a.D, split, (int64_t)a.M, PROJ ? 0 : 1);
  rc = check_launch("mlp_reduce");
  if (rc || !a.xn_out) return rc;
  // second output for the split panels: the blocked LayerNorm kernel over their rows (a few thousand)
  return layernorm_rows_blocked(prec, reinterpret_cast<const float*>(reinterpret_cast<const char*>(a.x) + rb0 * (a.D / 4) * 512), (int64_t)a.M - rb0 * 32, a.D,
                                a.gamma_n, a.beta_n, a.eps, static_cast<char*>(a.xn_out) + rb0 * (a.D / 8) * 512, s);
}

}  // namespace
}  // namespace effocr
