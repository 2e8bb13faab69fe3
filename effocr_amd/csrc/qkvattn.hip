// Fused norm1 + attn.qkv + multi-head self-attention of one timm ViT block (bf16 / f16 operands, gfx950):
//     att = softmax(q k^T / 8) v   with   [q | k | v] = LayerNorm(x) . Wqkv^T + b          (head_dim 64)
// (timm Block: attn(norm1(x)) up to, not including, attn.proj; call site models/encoders.py:58,63 via
// infer_effocr.py:314).  The qkv tensor [tokens, 3*D] never exists in HBM: per ViT-S block and 1024 crops that
// removes 465 MB written + 482 MB read, and the separate LN1+qkv and attention launches (round 1: 3.48 + 1.91 ms
// of a 15.4 ms forward) become one kernel that reads x once (fp32, 310 MB) and writes the attention output
// (16-bit, 155 MB).
//
// One workgroup = ONE IMAGE (T <= 224 tokens) = 4 waves, one per SIMD (512-register regime of mlp_kernel.hpp /
// gemm3.hip); wave w owns token tiles 2w and 2w+1 (32 tokens each; tile 7 of a 197-token image does not exist,
// wave 3 runs one tile).  Everything a wave needs of its own tokens stays in its registers for the whole kernel:
//   prologue   LayerNorm of the wave's rows, two lanes per row; lane (row, half) loads exactly the fp32 chunks that
//              make up ITS MFMA operand fragments (k chunks 2t+half): the normalised rows exist only as D/16
//              fragments per tile (96 VGPRs per tile at D = 384), no LDS panel.
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
//                 * v (unswapped): lane (dim, half) holds keys {0-3, 8-11 | 4-7, 12-15} + 16m of its dim = the
//                   A-operand fragment of O^T = V^T P^T under exactly the key permutation in which the S^T
//                   C-layout hands over P.  Same lane-linear LDS image, [key tile][m][dim tile][lane].
//                 K and V of the image (<= 56 KB) are the only activations that cross waves.
//     attention   per query tile: S^T tiles (keys x queries) so that a lane holds one query's whole score row
//                 (<= 112 registers): row max / sum lane-local + one cross-half exchange, no online rescale;
//                 P tile by tile into O^T = V^T P^T; 16-bit output rows stored fragment-blocked.
#include "common.hpp"
#include "kernels.hpp"
#include <math.h>
#include <type_traits>

namespace effocr {
namespace {

template <int I, int N, typename F> __device__ __forceinline__ void qa_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    qa_for<I + 1, N>(f);
  }
}

constexpr int QA_STAGE = 16384;                          // bytes per ring stage: 2 row blocks x 16 k-chunks x 512 B
constexpr int QA_RING = 6;

template <typename E, int D, int NTT>
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
  static_assert(D % 128 == 0 && NS >= R - 1, "qkvattn: embed dim must be a multiple of 128");
  __shared__ __attribute__((aligned(16))) char smem[R * QA_STAGE + NTT * 8192 + 5 * D * 4];
  // K / V and the parameters sit in the first 64 KB so that every access is one base register + a 16-bit immediate
  char* sK = smem;                                       // [key tile][k-step 0..3][lane] 16 B
  char* sV = sK + NTT * 4096;                            // [key tile][m 0..1][dim tile 0..1][lane] 16 B
  float* sG = reinterpret_cast<float*>(sV + NTT * 4096);
  float* sBt = sG + D;
  float* sBias = sBt + D;                                // qkv bias [3*D]
  char* sW = smem + NTT * 8192 + 5 * D * 4;              // weight ring

  const int tid = threadIdx.x, lane = tid & 63, r31 = lane & 31, half = lane >> 5;
  const int w = wave_id();
  const int T = a.T;
  const int64_t tok0 = (int64_t)blockIdx.x * T;          // first token of this image
  const char* Wb = static_cast<const char*>(a.Wb);

  for (int n = tid; n < D; n += 256) { sG[n] = a.gamma[n]; sBt[n] = a.beta[n]; }
  for (int n = tid; n < 3 * D; n += 256) sBias[n] = a.bias[n];

  // ---- ring: global stage g = (head, q|k|v, k slice).  Wave w copies row block w>>1, k chunks (w&1)*8..+8 of the
  // stage: 4 pieces of 1 KB (two adjacent 512-byte cells each).  (ih, isec, ikt) = the next stage to be issued.
  int ih = 0, isec = 0, ikt = 0, islot = 0;
  auto issue_piece = [&](int p) __attribute__((always_inline)) {
    const int rb = (isec * D + ih * 64) / 32 + (w >> 1);
    const char* src = Wb + ((size_t)rb * KC + ikt * 16 + (w & 1) * 8 + 2 * p) * 512 + lane * 16;
    char* dst = sW + islot * QA_STAGE + ((w >> 1) * 16 + (w & 1) * 8 + 2 * p) * 512;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
  };
  auto issue_advance = [&]() __attribute__((always_inline)) {
    islot = islot + 1 == R ? 0 : islot + 1;
    if (++ikt == KT) { ikt = 0; if (++isec == 3) { isec = 0; ++ih; } }
  };

  auto run = [&](auto NT_) __attribute__((always_inline)) {
    constexpr int NT = decltype(NT_)::value;             // token tiles of this wave (wave-uniform, 0..2)
    constexpr int NA = NT > 0 ? NT : 1;
    V8 xf[NA][NXF];                                      // LayerNorm(x) operand fragments, resident for the whole kernel

    // ---- input rows first (oldest in the in-order VM queue).  lane = (row r31, half): 16-bit k chunk 2t+half of
    // its row = fp32 chunks 4t+2half, 4t+2half+1.  Rows past the image's last token re-read the last token
    // (their keys are masked, their query rows never stored).
    auto load_rows = [&](int tt, f32x4 (&xv)[2 * NXF]) __attribute__((always_inline)) {
      int t = (2 * w + tt) * 32 + r31;
      t = t < T ? t : T - 1;
      const int64_t tok = tok0 + t;
      const char* xb = reinterpret_cast<const char*>(a.x) + (tok >> 5) * (int64_t)(D / 4) * 512 + (tok & 31) * 16;
#pragma unroll
      for (int i = 0; i < NXF; ++i) {
        xv[2 * i] = *reinterpret_cast<const f32x4*>(xb + (size_t)(4 * i + 2 * half) * 512);
        xv[2 * i + 1] = *reinterpret_cast<const f32x4*>(xb + (size_t)(4 * i + 2 * half + 1) * 512);
      }
    };
    // LayerNorm in registers: xv (this lane's half of the row, fp32) -> xf[tt][t] = operand fragment of k16 step t
    auto layernorm_to_xf = [&](int tt, const f32x4 (&xv)[2 * NXF]) __attribute__((always_inline)) {
      float sm = 0.f;
#pragma unroll
      for (int i = 0; i < 2 * NXF; ++i) sm += (xv[i][0] + xv[i][1]) + (xv[i][2] + xv[i][3]);
      sm += __shfl_xor(sm, 32, 64);
      const float mean = sm * (1.0f / D);
      float ss = 0.f;
#pragma unroll
      for (int i = 0; i < 2 * NXF; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d = xv[i][e] - mean; ss += d * d; }
      ss += __shfl_xor(ss, 32, 64);
      const float rstd = 1.0f / sqrtf(ss * (1.0f / D) + a.eps);
#pragma unroll
      for (int t = 0; t < NXF; ++t) {
        u32x2 pk[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int c = 4 * t + 2 * half + j;
          const f32x4 gm = *reinterpret_cast<const f32x4*>(sG + c * 4);
          const f32x4 bt = *reinterpret_cast<const f32x4*>(sBt + c * 4);
          const f32x4 v = xv[2 * t + j];
          pk[j] = pack4<E>((v[0] - mean) * rstd * gm[0] + bt[0], (v[1] - mean) * rstd * gm[1] + bt[1],
                           (v[2] - mean) * rstd * gm[2] + bt[2], (v[3] - mean) * rstd * gm[3] + bt[3]);
        }
        const u32x4 q = {pk[0][0], pk[0][1], pk[1][0], pk[1][1]};
        xf[tt][t] = __builtin_bit_cast(V8, q);
      }
    };
    f32x4 xv0[2 * NXF];
    if constexpr (NT > 0) load_rows(0, xv0);
    __syncthreads();                                     // parameters visible before the ring starts filling
#pragma unroll
    for (int s0 = 0; s0 < R - 1; ++s0) {
#pragma unroll
      for (int p = 0; p < 4; ++p) issue_piece(p);
      issue_advance();
    }
    if constexpr (NT > 0) layernorm_to_xf(0, xv0);
    if constexpr (NT > 1) {                              // (both tiles' raw rows at once would need 384 VGPRs)
      __builtin_amdgcn_sched_barrier(0);
      f32x4 xv1[2 * NXF];
      load_rows(1, xv1);
      layernorm_to_xf(1, xv1);
    }

    int slot = 0;                                        // ring slot of the stage being consumed
    int g = 0;                                           // its global index (used by the SMALL path only)
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((R - 2) * 4) : "memory");   // stage 0 (own pieces) ...
    __builtin_amdgcn_s_barrier();                        // ... and everybody's
    asm volatile("" ::: "memory");

    const float cexp = 0.125f * 1.44269504088896340736f; // head_dim^-0.5 * log2(e)

    auto head = [&](auto LAST_, int h) __attribute__((always_inline)) {
      constexpr bool LAST = decltype(LAST_)::value;
      V8 qf[NA][4];                                      // Q^T operand fragments of the wave's tiles (k-step = 16 head dims)
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
          V8 f0 = *reinterpret_cast<const V8*>(st);
          V8 f1 = *reinterpret_cast<const V8*>(st + 16 * 512);
          qa_for<0, 8>([&](auto KS_) {
            constexpr int ks = decltype(KS_)::value;
            if constexpr (ks == 4) {
              // middle of stage g: stage g+1 has landed (own pieces; the younger stages may stay in flight) and,
              // past the barrier, everybody's; every wave is done with stage g-1, whose slot takes stage g+R-1
              if constexpr (SMALL) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
              else if constexpr (!LAST) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((R - 3) * 4) : "memory");
              else if constexpr (ft >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(((ft < R - 2 ? ft : R - 2) - 1) * 4) : "memory");
              asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
              __builtin_amdgcn_s_barrier();
              asm volatile("" ::: "memory");
            }
            V8 n0, n1;
            if constexpr (ks < 7) {
              n0 = *reinterpret_cast<const V8*>(st + (2 * (ks + 1)) * 512);
              n1 = *reinterpret_cast<const V8*>(st + (16 + 2 * (ks + 1)) * 512);
            }
#pragma unroll
            for (int tt = 0; tt < NT; ++tt) {
              if constexpr (sec < 2) {                   // q^T, k^T: rows = features, cols = tokens
                acc[tt][0] = Op16<E>::mfma(f0, xf[tt][kt * 8 + ks], acc[tt][0]);
                acc[tt][1] = Op16<E>::mfma(f1, xf[tt][kt * 8 + ks], acc[tt][1]);
              } else {                                   // v: rows = tokens, cols = features
                acc[tt][0] = Op16<E>::mfma(xf[tt][kt * 8 + ks], f0, acc[tt][0]);
                acc[tt][1] = Op16<E>::mfma(xf[tt][kt * 8 + ks], f1, acc[tt][1]);
              }
            }
            if constexpr (ks >= 4) {
              if constexpr (SMALL) { if (g + R - 1 < NS) issue_piece(ks - 4); }
              else if constexpr (!LAST || ft >= R - 1) issue_piece(ks - 4);
            }
            if constexpr (ks < 7) { f0 = n0; f1 = n1; }
          });
          if constexpr (SMALL) { if (g + R - 1 < NS) issue_advance(); }
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
              else if constexpr (sec == 1) *reinterpret_cast<u32x4*>(sK + ((tile * 4 + 2 * i + m) * 64 + lane) * 16) = p;
              else *reinterpret_cast<u32x4*>(sV + (((tile * 2 + m) * 2 + i) * 64 + lane) * 16) = p;
            }
          }
        }
      });
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                      // K and V of every tile are in LDS
      asm volatile("" ::: "memory");

      // ---- attention of the wave's query tiles against all keys of the image
#pragma unroll
      for (int tt = 0; tt < NT; ++tt) {
        __builtin_amdgcn_sched_barrier(0);               // one query tile at a time (two score rows do not fit)
        const int tq = (2 * w + tt) * 32 + r31;
        f32x16 s[NTT];
#pragma unroll
        for (int kt = 0; kt < NTT; ++kt) {
#pragma unroll
          for (int r = 0; r < 16; ++r) s[kt][r] = 0.f;
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) {
            const V8 kf = *reinterpret_cast<const V8*>(sK + ((kt * 4 + ks) * 64 + lane) * 16);
            s[kt] = Op16<E>::mfma(kf, qf[tt][ks], s[kt]);
          }
          __builtin_amdgcn_sched_barrier(0);             // (all 28 K fragment reads hoisted to the top would cost 112 registers)
        }
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < NTT; ++kt) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            if (kt == NTT - 1 || NTT <= 2) {             // tiles that may hold padded keys
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
        float l = 0.f;
        const float mxc = mx * cexp;
#pragma unroll
        for (int kt = 0; kt < NTT; ++kt) {
#pragma unroll
          for (int m = 0; m < 2; ++m) {
            V8 pf;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              const float p = __builtin_amdgcn_exp2f(fmaf(s[kt][8 * m + j], cexp, -mxc));
              l += p;
              pf[j] = (E)p;
            }
#pragma unroll
            for (int db = 0; db < 2; ++db) {
              const V8 vf = *reinterpret_cast<const V8*>(sV + (((kt * 2 + m) * 2 + db) * 64 + lane) * 16);
              o[db] = Op16<E>::mfma(vf, pf, o[db]);
            }
          }
          __builtin_amdgcn_sched_barrier(0);             // keep every tile's exp / V reads next to its MFMAs
        }
        l += __shfl_xor(l, 32, 64);
        if (tq < T) {
          const float inv = 1.0f / l;
          char* ob = static_cast<char*>(a.out) + half * 8;
#pragma unroll
          for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4)
              *reinterpret_cast<u32x2*>(ob + blk_off(tok0 + tq, (h * 64 + db * 32 + 8 * q4) / 8, D / 8)) =
                  pack4<E>(o[db][4 * q4] * inv, o[db][4 * q4 + 1] * inv, o[db][4 * q4 + 2] * inv, o[db][4 * q4 + 3] * inv);
        }
      }
    };

#pragma unroll 1
    for (int h = 0; h < HEADS - 1; ++h) head(std::false_type{}, h);
    head(std::true_type{}, HEADS - 1);
  };

  const int nt = 2 * w + 1 < NTT ? 2 : (2 * w < NTT ? 1 : 0);
  if (nt == 2) run(std::integral_constant<int, 2>{});
  else if (nt == 1) run(std::integral_constant<int, 1>{});
  else run(std::integral_constant<int, 0>{});
}

template <typename E>
int launch_qkvattn(const QkvAttnArgs& a, hipStream_t s) {
  const int ntt = (a.T + 31) / 32;
  const dim3 grid((unsigned)a.B), blk(256);
  if (a.D == 384 && ntt == 7) hipLaunchKernelGGL((qkvattn_kernel<E, 384, 7>), grid, blk, 0, s, a);
  else if (a.D == 384 && ntt <= 2) hipLaunchKernelGGL((qkvattn_kernel<E, 384, 2>), grid, blk, 0, s, a);
  else if (a.D == 128 && ntt == 7) hipLaunchKernelGGL((qkvattn_kernel<E, 128, 7>), grid, blk, 0, s, a);
  else if (a.D == 128 && ntt <= 2) hipLaunchKernelGGL((qkvattn_kernel<E, 128, 2>), grid, blk, 0, s, a);
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
  return prec == PREC_BF16 ? launch_qkvattn<__bf16>(a, s) : launch_qkvattn<_Float16>(a, s);
}

}  // namespace effocr
