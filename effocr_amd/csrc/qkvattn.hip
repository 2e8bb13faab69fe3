// Fused attn.qkv + multi-head self-attention of one timm ViT block (bf16 / f16 operands, gfx950):
//     att = softmax(q k^T / 8) v   with   [q | k | v] = xn . Wqkv^T + b,   xn = norm1(x) (16-bit)     (head_dim 64)
// (timm Block: attn(norm1(x)) up to, not including, attn.proj; call site models/encoders.py:58,63 via
// infer_effocr.py:314).  The qkv tensor [tokens, 3*D] never exists in HBM: per ViT-S block and 1024 crops that
// removes 465 MB written + 482 MB read, and the separate LN1+qkv and attention launches (round 1: 3.48 + 1.91 ms
// of a 15.4 ms forward) become one kernel that reads the normalised rows once (16-bit, 155 MB; written by the
// previous block's fused MLP kernel in its epilogue, mlp_kernel.hpp) and writes the attention output (155 MB).
//
// One workgroup = ONE IMAGE (T <= 224 tokens) at a time = 4 waves, one per SIMD (512-register regime of
// mlp_kernel.hpp / gemm3.hip); wave w owns token tiles 2w and 2w+1 (32 tokens each; tile 7 of a 197-token image is
// a dummy).  Everything a wave needs of its own tokens stays in its registers for the whole image:
//   prologue   lane (row, half) loads exactly the 16-byte chunks 2t+half of its row = ITS MFMA operand fragments:
//              the rows exist only as D/16 fragments per tile (96 registers per tile at D = 384), no LDS panel.
//   per head h (loop, rolled):
//     projection  q^T, k^T (SWAPPED: rows = features, cols = tokens) and v (NOT swapped: rows = tokens, cols =
//                 features) of the wave's tiles for head h: 6 feature tiles x D/16 k-steps x tiles MFMAs.  The
//                 weight slice of the head (192 rows of Wqkv) streams through a 6-slot ring of 16 KB stages
//                 (64 features x 128 k) by global_load_lds, five stages ahead; the copy is fragment-blocked in
//                 HBM, so a stage is a verbatim copy of 512-byte cells and fragment reads are conflict-free.
//                 One W fragment feeds both token tiles (0.5 LDS reads per MFMA).
//     hand-over   the MFMA C-layout IS the operand layout the attention needs, with no permutes at all:
//                 * q^T: lane (token, half) holds dims {0-3, 8-11 | 4-7, 12-15} + 16m of its token = a B-operand
//                   fragment of S^T = K Q^T under a fixed permutation of the head dims;
//                 * k^T: the same registers for the key tokens = the A-operand fragment of S^T under the SAME
//                   permutation (a dot product does not care): each lane drops its 16 bytes into LDS at
//                   [key tile][k-step][lane] and every wave later reads the slot of its own lane id;
//                 * v (unswapped; its weight rows permuted per 32 on the host so that the output tile hands a lane 8
//                   consecutive head dims): lane (dim, half) holds keys {0-3, 8-11 | 4-7, 12-15} + 16m of its dim = the
//                   A-operand fragment of O^T = V^T P^T under exactly the key permutation in which the S^T
//                   C-layout hands over P.  Same lane-linear LDS image, [key tile][m][dim tile][lane].
//                 K and V of the image (<= 56 KB) are the only activations that cross waves.
//     attention   per query tile: S^T tiles (keys x queries) so that a lane holds one query's whole score row
//                 (<= 112 registers): row max / sum lane-local + one cross-half exchange, no online rescale;
//                 P tile by tile into O^T = V^T P^T; 16-bit output rows stored fragment-blocked.
// Workgroups are persistent (grid = min(images, CUs)); the rows of the NEXT image are requested right after the
// last head's projection (the fragments are dead from there on), so they land under that head's attention.
#include "common.hpp"
#include "kernels.hpp"
#include <math.h>
#include <type_traits>

#ifndef QA_NT
#define QA_NT 0                                          // non-temporal: 1 row loads, 2 output stores (both measured slower: the fused MLP reads the output next)
#endif
#ifndef QA_ODD_TILE_WAVE
#define QA_ODD_TILE_WAVE 1
#endif
#ifndef QA_BARRIER_DRAIN
#define QA_BARRIER_DRAIN 0
#endif

namespace effocr {
namespace {

// -DQA_STAMP (tools/ab_build.sh variant, never shipped): every wave of every workgroup records s_memtime at the milestones of the head
// it is in (the last head's values stay); tools/qa_timeline.py reads the table through effocr_debug_qa_stamps.  Branch-free: all lanes store.
#ifdef QA_STAMP
constexpr int QA_STAMP_WGS = 256, QA_STAMP_N = 16;
__device__ unsigned long long qa_stamps[QA_STAMP_WGS * 4 * QA_STAMP_N];
#define QA_STAMP_AT(k) qa_stamps[((blockIdx.x & (QA_STAMP_WGS - 1)) * 4 + w) * QA_STAMP_N + (k)] = __builtin_amdgcn_s_memtime();
#else
#define QA_STAMP_AT(k)
#endif

template <int I, int N, typename F> __device__ __forceinline__ void qa_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    qa_for<I + 1, N>(f);
  }
}


constexpr int QA_STAGE = 16384;                          // bytes per ring stage: 2 row blocks x 16 k-chunks x 512 B
constexpr int QA_RING = 6;

// CLS: the LAST transformer block — only the class token's attention row reaches the output (the rest of the block then runs
// on B gathered rows, api.hip): k and v of every token as usual, q and the attention only for token tile 0 (wave 0).
// NPV: P.V steps of 16 keys per query tile — 2 * NTT, or 2 * NTT - 1 when the last 16 keys of the last key tile are all padding
// (T <= 32 * NTT - 16: ViT-S/16 at 224 has 197 tokens, keys 208..223 do not exist): 3 MFMAs + 8 exponentials per query tile less.
template <typename E, int D, int NTT, bool CLS = false, int NPV = 2 * NTT>
__global__ __launch_bounds__(256, 1) void qkvattn_kernel(QkvAttnArgs a) {
  typedef typename Op16<E>::V8 V8;
  constexpr int HEADS = D / 64;
  constexpr int KC = D / 8;                              // 16-B k chunks per weight row
  constexpr int KT = D / 128;                            // 128-k slices of the contraction
  constexpr int NSH = 3 * KT;                            // ring stages per head: (q|k|v, k slice)
  constexpr int NS = HEADS * NSH;                        // ring stages per image
  constexpr int NXF = D / 16;                            // LayerNorm(x) fragments per token tile
  constexpr int R = QA_RING;
  constexpr bool SMALL = NSH < R - 1;                    // miniature test width: a head is shorter than the prefetch distance
  static_assert(D % 128 == 0 && NS >= R - 1, "qkvattn: embed dim must be a multiple of 128");   // (hsplit > 1 needs NSH >= R - 1: launcher)
  __shared__ __attribute__((aligned(16))) char smem[R * QA_STAGE + NTT * 8192 + 3 * D * 4];
  // K / V and the parameters sit in the first 64 KB so that every access is one base register + a 16-bit immediate
  char* sK = smem;                                       // [key tile][k-step 0..3][lane] 16 B
  char* sV = sK + NTT * 4096;                            // [key tile][m 0..1][dim tile 0..1][lane] 16 B
  float* sBias = reinterpret_cast<float*>(sV + NTT * 4096);   // qkv bias [3*D]
  char* sW = smem + NTT * 8192 + 3 * D * 4;              // weight ring

  const int tid = threadIdx.x;
  // lane-derived values are re-derived behind an opaque asm at the top of every image and every head: otherwise LICM
  // hoists ~100 registers of loop-invariant per-lane addresses (x chunks, LDS slots, output cells) out of the image /
  // head loops and the allocator spills the operand fragments instead (1100 dwords of scratch measured)
  int lane = tid & 63, r31 = lane & 31, half = lane >> 5;
  auto refresh_lane = [&]() __attribute__((always_inline)) {
    asm volatile("" : "+v"(lane));
    r31 = lane & 31;
    half = lane >> 5;
  };
  const int w = wave_id();
  QA_STAMP_AT(14)                                        // kernel entry (tools/qa_timeline.py: entry -> first head = the prologue)
  const int T = a.T;
  int64_t tok0 = 0;                                      // first token of the current image
  // Workgroups are persistent (grid = min(images, CUs)): image blockIdx.x, + gridDim.x, ...  The weight stream of an
  // image is cyclic over the heads, and workgroup b starts its cycle at head h0(b): co-resident workgroups then read
  // DIFFERENT parts of Wqkv at any moment (all of them starting at head 0 in lock step serialised on the same L2
  // channels: the first head's projection took 8x as long as the others), and the ring never drains between images.
  // Head split (small batches): HS workgroups share an image, workgroup (slot, grp) runs heads [grp*NH, grp*NH + NH) of the
  // images slot, slot + nslots, ... — every workgroup still holds the image's whole rows (k and v need every token), so a
  // batch of 64 images occupies 192 CUs with 2 heads each instead of 64 CUs with 6.
  const int HS = a.hsplit > 1 ? a.hsplit : 1, NH = HEADS / HS;
  const int grp = (int)blockIdx.x % HS, slot0 = a.img0 + (int)blockIdx.x / HS, nslots = (int)gridDim.x / HS;
  const int hb = grp * NH;                               // first head of this workgroup
  const int h0 = ((int)blockIdx.x >> 3) % NH;            // blocks b, b+8, ... share an XCD (and its L2)
  const char* Wb = static_cast<const char*>(a.Wb);

  // The qkv bias goes straight to LDS by LDS-DMA (16-byte units, inline asm: the compiler sees no LDS-DMA and no load).  As a loop of
  // load -> ds_write it was FIVE serialised memory round trips (s_waitcnt vmcnt(0) in every iteration) in front of the first request for
  // the image's rows: 9.9-11.3 k ticks between kernel entry and the first head of a small call (tools/qa_timeline.py), a quarter of a
  // one-head workgroup's life.  The pieces are the oldest entries of the in-order VM queue: the first stage's counted wait + barrier cover them.
  {
    const unsigned sB_lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)sBias;
#pragma unroll
    for (int j = 0; j < (3 * D / 4 + 255) / 256; ++j) {
      const int u = j * 256 + tid;                         // 16-byte unit of the bias image
      const char* src = reinterpret_cast<const char*>(a.bias) + (size_t)u * 16;
      const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(sB_lds + (unsigned)(j * 256 + (tid >> 6) * 64) * 16u));
      if (u < 3 * D / 4)
        asm volatile("s_mov_b32 m0, %1\n\t"
                     "s_nop 0\n\t"
                     "global_load_lds_dwordx4 %0, off"
                     :: "v"(src), "s"(dst) : "memory", "m0");
    }
  }

  // ---- ring: global stage g = (head, q|k|v, k slice).  Wave w copies row block w>>1, k chunks (w&1)*8..+8 of the
  // stage: 4 pieces of 1 KB (two adjacent 512-byte cells each).  (ih, isec, ikt) = the next stage to be issued.
  int ih = hb + h0, isec = 0, ikt = 0, islot = 0;
  // the four 1 KB pieces of a wave's share of a stage: ONE source address and ONE LDS base, the piece index is the instruction's
  // immediate offset (added on both sides by the hardware) — a quarter of the address / M0 arithmetic of separate pointers
  const unsigned lane16 = (unsigned)(tid & 63) * 16u;
  auto issue_piece = [&](int p) __attribute__((always_inline)) {              // p = 0..3, a constant after inlining
    const int rb = (isec * D + ih * 64) / 32 + (w >> 1);
    const __attribute__((address_space(1))) void* src =
        (const __attribute__((address_space(1))) void*)(Wb + ((size_t)rb * KC + ikt * 16 + (w & 1) * 8) * 512 + lane16);
    __attribute__((address_space(3))) void* dst = (__attribute__((address_space(3))) void*)(sW + islot * QA_STAGE + ((w >> 1) * 16 + (w & 1) * 8) * 512);
    switch (p) {
      case 0: __builtin_amdgcn_global_load_lds(src, dst, 16, 0, 0); break;
      case 1: __builtin_amdgcn_global_load_lds(src, dst, 16, 1024, 0); break;
      case 2: __builtin_amdgcn_global_load_lds(src, dst, 16, 2048, 0); break;
      default: __builtin_amdgcn_global_load_lds(src, dst, 16, 3072, 0); break;
    }
  };
  // Steady state: the pieces as inline asm.  hipcc treats __builtin_amdgcn_global_load_lds as an access to both
  // address spaces ("pending flat") and turns its next LDS wait into s_waitcnt lgkmcnt(0): every piece issued between two MFMAs
  // drained the weight-fragment read issued a moment earlier (mlp_kernel.hpp has the numbers).  Completion is counted by hand
  // (vmcnt at the stage barrier) either way.  M0 = LDS base (saved / restored), s_nop = the M0-write -> LDS-DMA wait state.
  const unsigned sW_lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)sW;
  // one piece (the four pieces of a stage go out behind four different k16 steps after the barrier: the workgroup's sixteen 1 KB requests
  // do not hit the address unit in one burst; tools/ubench/dma_cost.hip: 38.4 -> 36.6 cycles per MFMA for the ring alone)
  auto issue_piece_asm = [&](auto P_) __attribute__((always_inline)) {
    constexpr int pc = decltype(P_)::value;
    const int rb = (isec * D + ih * 64) / 32 + (w >> 1);
    const char* src = Wb + ((size_t)rb * KC + ikt * 16 + (w & 1) * 8) * 512 + lane16;
    const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(sW_lds + (unsigned)(islot * QA_STAGE) + (unsigned)(((w >> 1) * 16 + (w & 1) * 8) * 512)));
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\t"
                 "s_mov_b32 m0, %2\n\t"
                 "s_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, off offset:%3\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(src), "s"(dst), "n"(pc * 1024) : "memory");
  };
  auto issue_advance = [&]() __attribute__((always_inline)) {
    islot = islot + 1 == R ? 0 : islot + 1;
    if (++ikt == KT) { ikt = 0; if (++isec == 3) { isec = 0; ih = ih + 1 == hb + NH ? hb : ih + 1; } }
  };

  auto run = [&](auto NT_) __attribute__((always_inline)) {
    constexpr int NT = decltype(NT_)::value;             // token tiles of this wave (wave-uniform, 0..2)
    constexpr int NA = NT > 0 ? NT : 1;
    V8 xf[NA][NXF];                                      // LayerNorm(x) operand fragments, resident for the whole kernel

    // lane = (row r31, half): operand fragment t of its row = 16-byte chunk 2t+half.  Rows past the image's last token (their keys
    // are masked, their query rows never stored) re-read the last token.  (Zero fragments instead — 59 of the 256 row slots of a
    // 197-token image feeding the matrix pipe operands that do not toggle the multipliers, the chip runs this kernel power-limited —
    // measured 1.1 % SLOWER end to end, twice, same box: DESIGN.md, round 4.)
    auto load_frags = [&](int img) __attribute__((always_inline)) {
#pragma unroll
      for (int tt = 0; tt < NT; ++tt) {
        const int t0 = (2 * w + tt) * 32 + r31;
        const int t = t0 < T ? t0 : T - 1;
        const int64_t tok = (int64_t)img * T + t;
        const char* xb = static_cast<const char*>(a.xn) + (tok >> 5) * (int64_t)KC * 512 + (tok & 31) * 16 + half * 512;   // + compile-time offsets only
#pragma unroll
        for (int i = 0; i < NXF; ++i) {
          xf[tt][i] = (QA_NT & 1) ? __builtin_nontemporal_load(reinterpret_cast<const V8*>(xb + i * 1024)) : *reinterpret_cast<const V8*>(xb + i * 1024);
        }
      }
    };
    const int nimg = (a.B - slot0 + nslots - 1) / nslots;   // images of this workgroup (>= 1)
    const int gtotal = nimg * NH * NSH;                  // ring stages of this workgroup
    int slot = 0;                                        // ring slot of the stage being consumed
    int g = 0;                                           // its index in the workgroup's stream (used by the SMALL path only)

    const float cexp = 0.125f * 1.44269504088896340736f; // head_dim^-0.5 * log2(e)
    // Padded keys (the last key tile of a 193..224-token image) are masked by STARTING their score accumulators at -inf: one
    // per-lane initial tile, computed once, instead of a compare + select per score per query tile per head.
    constexpr bool INITMASK = NTT > 2;
    f32x16 sinit;
#pragma unroll
    for (int r = 0; r < 16; ++r) sinit[r] = ((NTT - 1) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5) >= T) ? -INFINITY : 0.f;

    auto head = [&](auto LAST_, auto PRE_, int h, int hi, int img_next) __attribute__((always_inline)) {
      constexpr bool LAST = decltype(LAST_)::value;        // last head of the workgroup's last image: the ring runs dry
      constexpr bool PRE = decltype(PRE_)::value;          // last head of an image that is not the last: request the next image's rows
      refresh_lane();
      QA_STAMP_AT(0)
      V8 qf[NA][4];                                      // Q^T operand fragments of the wave's tiles (k-step = 16 head dims)
      V8 f0, f1;                                         // the k-step's two W fragments (row blocks 0 / 1 of the stage)
      qa_for<0, 3>([&](auto SEC_) {
        constexpr int sec = decltype(SEC_)::value;       // 0 q, 1 k, 2 v
        f32x16 acc[NA][2];
#pragma unroll
        for (int tt = 0; tt < NA; ++tt)
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tt][i][r] = 0.f;
        qa_for<0, KT>([&](auto KT_) {
          constexpr int kt = decltype(KT_)::value;
          constexpr int sl = sec * KT + kt;              // stage within the head
          constexpr int ft = NSH - 1 - sl;               // LAST: stages that follow in the whole stream
          const char* st = sW + slot * QA_STAGE + half * 512 + r31 * 16;
          const int nslot = slot + 1 == R ? 0 : slot + 1;
          const char* stn = sW + nslot * QA_STAGE + half * 512 + r31 * 16;
          if constexpr (sl == 0) {                       // later stages: requested under the previous stage's last MFMAs
            f0 = *reinterpret_cast<const V8*>(st);
            f1 = *reinterpret_cast<const V8*>(st + 16 * 512);
          }
          qa_for<0, 8>([&](auto KS_) {
            constexpr int ks = decltype(KS_)::value;
            if constexpr (ks == 4) {
              // middle of stage g: stage g+1 has landed (own pieces; the younger stages may stay in flight) and,
              // past the barrier, everybody's; every wave is done with stage g-1, whose slot takes stage g+R-1
              if constexpr (SMALL) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
              else if constexpr (!LAST) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((R - 3) * 4) : "memory");
              else if constexpr (ft >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(((ft < R - 2 ? ft : R - 2) - 1) * 4) : "memory");
#if QA_BARRIER_DRAIN
              asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (A/B) not needed: the slot refilled behind the barrier is stage g-1's, fully consumed
#endif
              __builtin_amdgcn_s_barrier();
              asm volatile("" ::: "memory");
            }
            V8 n0, n1;
            if constexpr (ks < 7) {
              n0 = *reinterpret_cast<const V8*>(st + (2 * (ks + 1)) * 512);
              n1 = *reinterpret_cast<const V8*>(st + (16 + 2 * (ks + 1)) * 512);
            } else if constexpr (sl + 1 < NSH) {         // stage g+1 landed for everybody at this stage's barrier
              n0 = *reinterpret_cast<const V8*>(stn);
              n1 = *reinterpret_cast<const V8*>(stn + 16 * 512);
            }
#pragma unroll
            for (int tt = 0; tt < NT; ++tt) {
              if constexpr (CLS && sec == 0) {           // q only where the class token lives
                if (w == 0 && tt == 0) {
                  acc[tt][0] = Op16<E>::mfma(f0, xf[tt][kt * 8 + ks], acc[tt][0]);
                  acc[tt][1] = Op16<E>::mfma(f1, xf[tt][kt * 8 + ks], acc[tt][1]);
                }
              } else if constexpr (sec < 2) {            // q^T, k^T: rows = features, cols = tokens
                acc[tt][0] = Op16<E>::mfma(f0, xf[tt][kt * 8 + ks], acc[tt][0]);
                acc[tt][1] = Op16<E>::mfma(f1, xf[tt][kt * 8 + ks], acc[tt][1]);
              } else {                                   // v: rows = tokens, cols = features
                acc[tt][0] = Op16<E>::mfma(xf[tt][kt * 8 + ks], f0, acc[tt][0]);
                acc[tt][1] = Op16<E>::mfma(xf[tt][kt * 8 + ks], f1, acc[tt][1]);
              }
            }
            if constexpr (ks >= 4) {                     // behind the stage barrier: the slot of stage g-1 is free
              if constexpr (SMALL) { if (g + R - 1 < gtotal) issue_piece_asm(std::integral_constant<int, ks - 4>{}); }
              else if constexpr (!LAST || ft >= R - 1) issue_piece_asm(std::integral_constant<int, ks - 4>{});
            }
            if constexpr (ks < 7 || sl + 1 < NSH) { f0 = n0; f1 = n1; }
          });
          if constexpr (SMALL) { if (g + R - 1 < gtotal) issue_advance(); }
          else if constexpr (!LAST || ft >= R - 1) issue_advance();
          slot = slot + 1 == R ? 0 : slot + 1;
          ++g;
        });
        // ---- accumulators (+ bias) -> 16-bit operand fragments.  Registers 8m..8m+7 of tile i = fragment (i, m).
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) {
          const int tile = 2 * w + tt;
#pragma unroll
          for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int m = 0; m < 2; ++m) {
              u32x4 p;
              if constexpr (sec < 2) {
                const float* bp = sBias + sec * D + h * 64 + i * 32 + 4 * half;
                const f32x4 b0 = *reinterpret_cast<const f32x4*>(bp + 8 * (2 * m));
                const f32x4 b1 = *reinterpret_cast<const f32x4*>(bp + 8 * (2 * m + 1));
                const u32x2 lo = pack4<E>(acc[tt][i][8 * m] + b0[0], acc[tt][i][8 * m + 1] + b0[1], acc[tt][i][8 * m + 2] + b0[2], acc[tt][i][8 * m + 3] + b0[3]);
                const u32x2 hi = pack4<E>(acc[tt][i][8 * m + 4] + b1[0], acc[tt][i][8 * m + 5] + b1[1], acc[tt][i][8 * m + 6] + b1[2], acc[tt][i][8 * m + 7] + b1[3]);
                p = u32x4{lo[0], lo[1], hi[0], hi[1]};
              } else {
                const float bv = sBias[2 * D + h * 64 + i * 32 + r31];
                const u32x2 lo = pack4<E>(acc[tt][i][8 * m] + bv, acc[tt][i][8 * m + 1] + bv, acc[tt][i][8 * m + 2] + bv, acc[tt][i][8 * m + 3] + bv);
                const u32x2 hi = pack4<E>(acc[tt][i][8 * m + 4] + bv, acc[tt][i][8 * m + 5] + bv, acc[tt][i][8 * m + 6] + bv, acc[tt][i][8 * m + 7] + bv);
                p = u32x4{lo[0], lo[1], hi[0], hi[1]};
              }
              if constexpr (sec == 0) qf[tt][2 * i + m] = __builtin_bit_cast(V8, p);
              else if (tile < NTT) {
                if constexpr (sec == 1) *reinterpret_cast<u32x4*>(sK + ((tile * 4 + 2 * i + m) * 64 + lane) * 16) = p;
                else *reinterpret_cast<u32x4*>(sV + (((tile * 2 + m) * 2 + i) * 64 + lane) * 16) = p;
              }
            }
          }
        }
        QA_STAMP_AT(1 + sec)
      });
      if constexpr (PRE) load_frags(img_next);           // the fragments are dead from here on: the loads land under the attention
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                      // K and V of every tile are in LDS
      asm volatile("" ::: "memory");
      QA_STAMP_AT(4)

      // ---- attention of the wave's query tiles against all keys of the image.  K / V fragments are requested one
      // step ahead by hand (with one wave per SIMD nothing else hides the LDS latency; left to itself the compiler
      // either serialises read -> wait -> MFMA or hoists every read and spills)
      const char* kb = sK + lane * 16;
      const char* vb = sV + lane * 16;
      qa_for<0, (CLS ? 1 : NT)>([&](auto TT_) {
        constexpr int tt = decltype(TT_)::value;
        if constexpr (CLS) { if (w != 0) return; }       // (wave-uniform; the other waves go on to the next head's weight stream)
        if (2 * w + tt >= NTT) return;                   // the dummy tile (wave 3's second tile of a 197-token image) has no queries
        __builtin_amdgcn_sched_barrier(0);               // one query tile at a time (two score rows do not fit)
        const int tq = (2 * w + tt) * 32 + r31;
        f32x16 s[NTT];
        // K fragments one key tile ahead of their MFMAs (two and three tiles ahead, and key tiles in pairs with alternating accumulators,
        // measured +-0 in round 4: the 1.9 k ticks the s_memtime timeline showed for a query tile's 28 Q K^T MFMAs were mostly the stamps)
        constexpr int QA_KDEPTH = 1, KD = QA_KDEPTH + 1;   // lookahead in key tiles / fragment sets in the ring (2 and 3 tiles ahead: +-0)
        V8 kf[KD][4];
#pragma unroll
        for (int d = 0; d < QA_KDEPTH; ++d)
          if (d < NTT) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) kf[d][ks] = *reinterpret_cast<const V8*>(kb + (d * 4 + ks) * 1024);
          }
        qa_for<0, NTT>([&](auto KT_) {
          constexpr int kt = decltype(KT_)::value, cur = kt % KD, nxt = (kt + QA_KDEPTH) % KD;
          if constexpr (kt + QA_KDEPTH < NTT) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) kf[nxt][ks] = *reinterpret_cast<const V8*>(kb + ((kt + QA_KDEPTH) * 4 + ks) * 1024);
          }
          if constexpr (INITMASK && kt == NTT - 1) s[kt] = sinit;
          else {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[kt][r] = 0.f;
          }
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) s[kt] = Op16<E>::mfma(kf[cur][ks], qf[tt][ks], s[kt]);
          __builtin_amdgcn_sched_barrier(0);
        });
        QA_STAMP_AT(5 + tt * 5)
        V8 vf[2][2];
#pragma unroll
        for (int db = 0; db < 2; ++db) vf[0][db] = *reinterpret_cast<const V8*>(vb + db * 1024);
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < NTT; ++kt) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            if (!INITMASK) {                             // (<= 64 tokens: any tile may hold padded keys)
              const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
              if (key >= T) s[kt][r] = -INFINITY;
            }
            mx = fmaxf(mx, s[kt][r]);
          }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        f32x16 o[2];
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[db][r] = 0.f;
        f32x16 lsum;
#pragma unroll
        for (int r = 0; r < 16; ++r) lsum[r] = 0.f;
        const V8 ones = {(E)1.f, (E)1.f, (E)1.f, (E)1.f, (E)1.f, (E)1.f, (E)1.f, (E)1.f};
        const float mxc = mx * cexp;
        __builtin_amdgcn_sched_barrier(0);
        // P = exp2((S - max) * c), 8 keys (one operand fragment) at a time, into O^T = V^T P^T — software-pipelined by one
        // step: a region holds the exponentials of step st and the MFMAs of step st-1, which are independent, so the
        // compiler can put the matrix instructions under the VALU work (with the MFMAs of a step behind ITS OWN
        // exponentials the in-order wave ran them strictly one after the other)
        V8 pf[2];
        float av[8], ev[8];                                // the fragment in the making (one at a time)
        // part 0..2 of fragment st: v_exp_f32 issues at half rate — THREE per MFMA gap are free, each further one costs 8 cycles
        // (tools/ubench/mfma_fill.hip) — so the eight exponentials of a fragment are spread over the step's three MFMA gaps
        // instead of queueing behind its last MFMA: { 2 V reads, args 0-3, exp 0-1 } | { args 4-7, exp 2-4 } | { exp 5-7, 4 packs }
        auto p_part = [&](auto ST_, auto PART_) __attribute__((always_inline)) {
          constexpr int st = decltype(ST_)::value, kt = st >> 1, m = st & 1, part = decltype(PART_)::value;
          if constexpr (part == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) av[j] = __builtin_fmaf(s[kt][8 * m + j], cexp, -mxc);
            ev[0] = __builtin_amdgcn_exp2f(av[0]); ev[1] = __builtin_amdgcn_exp2f(av[1]);
          } else if constexpr (part == 1) {
#pragma unroll
            for (int j = 4; j < 8; ++j) av[j] = __builtin_fmaf(s[kt][8 * m + j], cexp, -mxc);
            ev[2] = __builtin_amdgcn_exp2f(av[2]); ev[3] = __builtin_amdgcn_exp2f(av[3]); ev[4] = __builtin_amdgcn_exp2f(av[4]);
          } else {
            ev[5] = __builtin_amdgcn_exp2f(av[5]); ev[6] = __builtin_amdgcn_exp2f(av[6]); ev[7] = __builtin_amdgcn_exp2f(av[7]);
            const u32x4 pw = {pack2<E>(ev[0], ev[1]), pack2<E>(ev[2], ev[3]), pack2<E>(ev[4], ev[5]), pack2<E>(ev[6], ev[7])};
            pf[st & 1] = __builtin_bit_cast(V8, pw);
          }
        };
        QA_STAMP_AT(6 + tt * 5)
        qa_for<0, 3>([&](auto P_) { p_part(std::integral_constant<int, 0>{}, P_); });
        __builtin_amdgcn_sched_barrier(0);
        qa_for<0, NPV>([&](auto ST_) {
          constexpr int st = decltype(ST_)::value, cur = st & 1, nxt = cur ^ 1;
          constexpr bool more = st + 1 < NPV;
          if constexpr (more) {
#pragma unroll
            for (int db = 0; db < 2; ++db) vf[nxt][db] = *reinterpret_cast<const V8*>(vb + ((st + 1) * 2 + db) * 1024);
          }
          o[0] = Op16<E>::mfma(vf[cur][0], pf[cur], o[0]);
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (more) p_part(std::integral_constant<int, st + 1>{}, std::integral_constant<int, 0>{});
          __builtin_amdgcn_sched_barrier(0);
          o[1] = Op16<E>::mfma(vf[cur][1], pf[cur], o[1]);
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (more) p_part(std::integral_constant<int, st + 1>{}, std::integral_constant<int, 1>{});
          __builtin_amdgcn_sched_barrier(0);
          // row sums on the (otherwise idle) matrix pipe: an all-ones A operand makes every row of the tile sum_k P[k][query],
          // i.e. the softmax denominator of exactly the rounded P that enters the numerator; no per-value adds, no exchange
          lsum = Op16<E>::mfma(ones, pf[cur], lsum);
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (more) p_part(std::integral_constant<int, st + 1>{}, std::integral_constant<int, 2>{});
          __builtin_amdgcn_sched_barrier(0);
        });
        QA_STAMP_AT(7 + tt * 5)
        const float l = lsum[0];
        if (tq < T) {
          const float inv = 1.0f / l;
          // The v rows of the weight copy are permuted per 32 (api.hip rowperm32): registers 8p..8p+7 of dim tile db are the 8
          // CONSECUTIVE head dims 32db + 16p + 8half.. = one whole 16-byte chunk of the output row (8-byte half chunks before).
          char* ob = static_cast<char*>(a.out);
#pragma unroll
          for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int p = 0; p < 2; ++p) {
              const u32x2 lo = pack4<E>(o[db][8 * p] * inv, o[db][8 * p + 1] * inv, o[db][8 * p + 2] * inv, o[db][8 * p + 3] * inv);
              const u32x2 hi = pack4<E>(o[db][8 * p + 4] * inv, o[db][8 * p + 5] * inv, o[db][8 * p + 6] * inv, o[db][8 * p + 7] * inv);
              const u32x4 v = {lo[0], lo[1], hi[0], hi[1]};
              u32x4* op = reinterpret_cast<u32x4*>(ob + blk_off(tok0 + tq, h * 8 + db * 4 + 2 * p + half, D / 8));
              if (QA_NT & 2) __builtin_nontemporal_store(v, op); else *op = v;
            }
        }
        QA_STAMP_AT(8 + tt * 5)
      });
      QA_STAMP_AT(15)
    };

    // (no barrier here: the bias pieces above are older than everything requested below, stage 0's counted wait + barrier make them visible)
    load_frags(slot0);                                   // oldest in the in-order VM queue
#pragma unroll
    for (int s0 = 0; s0 < R - 1; ++s0) {
#pragma unroll
      for (int p = 0; p < 4; ++p) issue_piece(p);
      issue_advance();
    }
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((R - 2) * 4) : "memory");   // stage 0 (own pieces) ...
    __builtin_amdgcn_s_barrier();                        // ... and everybody's
    asm volatile("" ::: "memory");
#pragma unroll 1
    for (int ii = 0; ii < nimg; ++ii) {
      const int img = slot0 + ii * nslots;
      tok0 = (int64_t)img * T;
      refresh_lane();
#pragma unroll 1
      for (int i = 0; i < NH - 1; ++i) {
        const int h = hb + (h0 + i < NH ? h0 + i : h0 + i - NH);
        head(std::false_type{}, std::false_type{}, h, i, 0);
      }
      const int hl = hb + (h0 == 0 ? NH - 1 : h0 - 1);
      if (ii + 1 < nimg) head(std::false_type{}, std::true_type{}, hl, HEADS - 1, img + nslots);
      else head(std::true_type{}, std::false_type{}, hl, HEADS - 1, 0);
    }
  };

  // a tile index >= NTT (wave 3's second tile of a 197-token image) is a dummy: its rows re-read the last token, it writes neither
  // K / V nor output.  (Separate one- AND zero-tile code paths tripled the kernel and made the register allocator spill the
  // fragments: rounds 2-5 ran the two-tile code on every wave.  Round 6: the step is bound by the socket power limit, the dummy
  // tile's MFMAs — 1/8 of the projection's — do no work but draw power; ONE extra body, for the wave whose second tile does not
  // exist, compiles without spills (66 -> 110 KB): kernel 3.62 / 3.66 -> 3.55 / 3.56 ms per 12 launches at a 2 % higher clock.)
#if QA_ODD_TILE_WAVE
  // an odd tile count (197 tokens = 7 tiles): a one-tile body for wave 3; same stages, barriers and DMA pieces as the two-tile body
  if ((NTT & 1) && w == 3) run(std::integral_constant<int, 1>{});
  else
#endif
  run(std::integral_constant<int, 2>{});
}

template <typename E>
int launch_qkvattn(const QkvAttnArgs& a, hipStream_t s) {
  const int ntt = (a.T + 31) / 32;
  const int cus = device_cus();
  // one persistent workgroup per CU (160 KB of LDS each); batches of less than half a round of CUs split every image's heads
  // over hs workgroups (hs = the largest divisor of the head count with B * hs <= CUs; D = 384 only: a head must be at least
  // as long as the ring's prefetch distance)
  const int heads = a.D / 64;
  auto split_for = [&](int64_t images) {                   // largest divisor of the head count with images * hs <= CUs
    if (a.hsplit == 1 || a.D != 384) return 1;
    for (int c = heads; c >= 2; --c)
      if (heads % c == 0 && images * c <= cus && (a.hsplit <= 0 || c <= a.hsplit)) return c;
    return 1;
  };
  const int64_t nimg_launch = (int64_t)a.B - a.img0;
  int hs = split_for(nimg_launch);
  // A batch of several rounds whose last round fills at most half of the CUs (1 139 crops of a configs[4] call: 4 x 256 + 115) runs that tail
  // as a SECOND launch with the heads of every tail image split over hs workgroups — 4.55 instead of 5 image times on the critical path.
  if (a.img0 == 0 && hs == 1 && a.B > cus && a.hsplit != 1) {
    const int tail = a.B % cus, ths = tail ? split_for(tail) : 1;
    if (ths > 1) {
      QkvAttnArgs m = a, t = a;
      m.B = a.B - tail; m.hsplit = 1;                        // images 0 .. B - tail - 1, whole rounds
      t.img0 = a.B - tail;                                   // the rest, split
      const int rc = launch_qkvattn<E>(m, s);
      return rc ? rc : launch_qkvattn<E>(t, s);
    }
  }
  QkvAttnArgs ah = a;
  ah.hsplit = hs;
  const QkvAttnArgs& a2 = ah;
  const dim3 grid((unsigned)(hs > 1 ? nimg_launch * hs : (nimg_launch < cus ? nimg_launch : cus))), blk(256);
  const bool short_tail = a.T <= 32 * ntt - 16;          // the last 16 keys are all padding
  if (a.D == 384 && ntt == 7 && a.cls_only && short_tail) hipLaunchKernelGGL((qkvattn_kernel<E, 384, 7, true, 13>), grid, blk, 0, s, a2);
  else if (a.D == 384 && ntt == 7 && a.cls_only) hipLaunchKernelGGL((qkvattn_kernel<E, 384, 7, true>), grid, blk, 0, s, a2);
  else if (a.D == 384 && ntt == 7 && short_tail) hipLaunchKernelGGL((qkvattn_kernel<E, 384, 7, false, 13>), grid, blk, 0, s, a2);
  else if (a.D == 384 && ntt == 7) hipLaunchKernelGGL((qkvattn_kernel<E, 384, 7>), grid, blk, 0, s, a2);
  else if (a.D == 384 && ntt <= 2) hipLaunchKernelGGL((qkvattn_kernel<E, 384, 2>), grid, blk, 0, s, a2);
  else if (a.D == 128 && ntt == 7) hipLaunchKernelGGL((qkvattn_kernel<E, 128, 7>), grid, blk, 0, s, a2);
  else if (a.D == 128 && ntt <= 2) hipLaunchKernelGGL((qkvattn_kernel<E, 128, 2>), grid, blk, 0, s, a2);
  else return fail(EFFOCR_EUNSUPPORTED, "qkv_attn_fused: (embed dim, tokens) must be (128|384, <=64 or 193..224)");
  return check_launch("qkv_attn_fused");
}

}  // namespace

bool qkv_attn_supported(int prec, int D, int T) {
  const int ntt = (T + 31) / 32;
  return prec != PREC_FP32 && (D == 384 || D == 128) && T >= 1 && (ntt <= 2 || ntt == 7);
}

int qkv_attn_fused(int prec, const QkvAttnArgs& a, hipStream_t s) {
  if (a.B <= 0) return EFFOCR_OK;
  if (!qkv_attn_supported(prec, a.D, a.T)) return fail(EFFOCR_EUNSUPPORTED, "qkv_attn_fused: unsupported (precision, embed dim, tokens)");
  if (a.rows_alloc % 32 || a.rows_alloc < (int64_t)a.B * a.T) return fail(EFFOCR_EINVAL, "qkv_attn_fused: rows_alloc must be a multiple of 32 >= batch * tokens");
  if ((reinterpret_cast<uintptr_t>(a.bias) & 15) != 0) return fail(EFFOCR_EINVAL, "qkv_attn_fused: the bias array must be 16-byte aligned (it reaches LDS by 16-byte DMA)");
  return prec == PREC_BF16 ? launch_qkvattn<__bf16>(a, s) : launch_qkvattn<_Float16>(a, s);
}


}  // namespace effocr

#ifdef QA_STAMP
extern "C" int effocr_debug_qa_stamps(unsigned long long* out, int n) {
  const int m = effocr::QA_STAMP_WGS * 4 * effocr::QA_STAMP_N;
  if (n < m) return -1;
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(effocr::qa_stamps), (size_t)m * 8) == hipSuccess ? 0 : -2;
}
#endif
