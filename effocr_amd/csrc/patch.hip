// Fused im2col + patch-embedding GEMM of timm's ViT (PatchEmbed: Conv2d(3, D, kernel 16, stride 16) + flatten, then
// x = tokens + pos_embed; models/encoders.py:58,63 via infer_effocr.py:314) for bf16 / f16 operands on gfx950:
//     x[img * T + 1 + p, :] = patch(img, p) . Wpe^T + b + pos_embed[1 + p, :]          (fp32 residual stream, fragment-blocked)
// The unfused pair (im2col16_kernel -> 16-bit patch rows in HBM -> gemm2) moved 1.85 GB per 1024 crops for 0.93 GB of algorithmic
// traffic (617 MB of fp32 pixels in, 310 MB of fp32 tokens out) and took 0.37 ms; here the patch rows never exist.
//
// Geometry = the projection phase of the fused MLP kernel (mlp_kernel.hpp): one workgroup = 4 waves (one per SIMD) = a panel of
// 128 patches for the WHOLE output width; wave w owns patches 32w..32w+31, lane = (patch r31, k half):
//   * im2col is an ADDRESS pattern, not a copy: k = (c, py, px) and a k16 step is one (c, py) row segment of 16 pixels, so the
//     lane's MFMA B-operand fragment (8 consecutive k) is 32 contiguous bytes of the fp32 image: two 16-byte loads straight into
//     registers, 4 v_cvt_pk — no LDS, no staging; the 64-k slab of the NEXT ring stages is requested one slab ahead;
//   * the weight [D, 768] (fragment-blocked copy) streams through an 8-slot ring of 16 KB LDS stages (4 row blocks x 64 k) by
//     LDS-DMA, stage order (k slab, output group): a slab's four fragments feed all D / 32 output tiles (12 MFMAs per fragment);
//   * swapped MFMAs (A = weight rows, B = patches): a lane ends with 4 consecutive features of its patch per register quad ->
//     bias + pos_embed + 16-byte stores into the blocked residual stream.  DW / 2 accumulator registers per lane (192 at DW = 384).
// Round 5: (1) embed dims wider than 384 (ViT-B: 768) are cut into DW = 384-wide output slices, one workgroup per (panel, slice), the
// slices of a panel adjacent in dispatch order so that the second reader of a pixel finds it in the memory-side cache; (2) the crops may
// arrive ALREADY in the operand type (X = E: effocr_crop_transform_batch_ex's 16-bit hand-off, SURVEY f-2): a lane's 8-k fragment is then
// ONE 16-byte load and nothing is converted — the values are the ones pack4 would have produced from the fp32 crop, so both input
// types give bit-identical tokens.
// Steady-state DMA is issued from inline asm (see mlp_kernel.hpp: the builtin makes hipcc drain the LDS queue after every piece).
#include "common.hpp"
#include "kernels.hpp"
#include <type_traits>

constexpr int PE_SLABS = 2;                              // 64-k pixel slabs in flight per lane (32 registers each for fp32 pixels; 3 and 4 measured +-0: not what binds)
constexpr int PE_SLABS16 = 4;                            // ... for 16-bit pixels (16 registers each)
#ifndef PATCH_NT
#define PATCH_NT 1                                        // non-temporal: 1 = pixel loads (read once), 2 = row stores
#endif
namespace effocr {
namespace {

template <int I, int N, typename F> __device__ __forceinline__ void pfor(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    pfor<I + 1, N>(f);
  }
}

constexpr int PE_STAGE = 16384;                          // bytes per ring stage: 4 row blocks x 8 k-chunks x 512 B
constexpr int PE_RING = 8;
constexpr int PE_K = 768;                                // 3 x 16 x 16

template <typename E, typename X, int D>                 // D = output width of ONE workgroup (a.D = the embed dim = D x slices)
__global__ __launch_bounds__(256, 1) void patch_embed_kernel(PatchArgs a) {
  typedef typename Op16<E>::V8 V8;
  constexpr bool X16 = sizeof(X) == 2;                   // crops already in the operand type
  constexpr int NSL = X16 ? PE_SLABS16 : PE_SLABS;
  constexpr int KC = PE_K / 8;                           // 16-byte k chunks per weight row
  constexpr int OT = D / 32, OG = OT / 4;                // output tiles / groups of 4 tiles (one ring stage = one group x 64 k)
  constexpr int KS = PE_K / 64;                          // 64-k slabs
  constexpr int NS = KS * OG;                            // ring stages per panel, order (slab, group)
  constexpr int R = PE_RING;
  static_assert(D % 128 == 0 && D <= 384, "patch_embed: a workgroup's output slice must be 128, 256 or 384 wide");
  static_assert(NS >= R - 1, "patch_embed: the panel's stream must be at least as long as the prefetch distance");
  __shared__ __attribute__((aligned(16))) char smem[R * PE_STAGE];
  const int tid = threadIdx.x, lane = tid & 63, r31 = lane & 31, half = lane >> 5;
  const int w = wave_id();
  const int64_t total = (int64_t)a.B * a.P;
  const int nsl = a.D / D;                               // output slices per panel (1, or 2 at D = 768)
  const int panel = (int)blockIdx.x / nsl, n0 = ((int)blockIdx.x - panel * nsl) * D;   // this workgroup's features n0 .. n0 + D - 1
  const int64_t m = (int64_t)panel * 128 + w * 32 + r31;              // this lane's patch (both half-waves: same patch, other k half)
  const int64_t mc = m < total ? m : total - 1;                       // patches past the end re-read the last one, never stored
  const int img = (int)(mc / a.P), p = (int)(mc - (int64_t)img * a.P);
  const int PW = a.W / 16, ty = p / PW, tx = p - ty * PW;
  // k16 step kk = (c = kk / 16, py = kk % 16): the lane's 8 pixels at  xp + (c * H + py) * W
  const X* xp = static_cast<const X*>(a.x) + ((int64_t)img * 3 * a.H + ty * 16) * a.W + tx * 16 + 8 * half;
  const int64_t plane = (int64_t)a.H * a.W;

  // ---- pixels of a slab: 4 k16 steps x 2 x 16 bytes per lane, converted to 4 operand fragments when the slab is consumed.
  // Two slabs in flight (static even / odd buffers); requested BEFORE the ring's prologue: they are needed first.
  struct Slab { u32x4 v[X16 ? 4 : 8]; };
  auto load_slab = [&](Slab& sl, int ks) __attribute__((always_inline)) {
#pragma unroll
    for (int c4 = 0; c4 < 4; ++c4) {
      const int kk = ks * 4 + c4;
      const X* q = xp + (int64_t)(kk >> 4) * plane + (int64_t)(kk & 15) * a.W;
      if constexpr (X16) {
        sl.v[c4] = (PATCH_NT & 1) ? __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(q)) : *reinterpret_cast<const u32x4*>(q);
      } else {
        sl.v[2 * c4] = (PATCH_NT & 1) ? __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(q)) : *reinterpret_cast<const u32x4*>(q);
        sl.v[2 * c4 + 1] = (PATCH_NT & 1) ? __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(q + 4)) : *reinterpret_cast<const u32x4*>(q + 4);
      }
    }
  };
  // NSL slabs in flight (static buffers, slab ks lives in buffer ks % NSL), requested BEFORE the ring's prologue: they are needed first
  Slab sl[NSL];
#pragma unroll
  for (int i = 0; i < NSL; ++i) load_slab(sl[i], i);
  asm volatile("" ::: "memory");

  // ---- weight ring: stage s = (slab s / OG, group s % OG); wave w copies row block w of the group: 4 pieces of 1 KB
  const char* Wb = static_cast<const char*>(a.Wb);
  const unsigned lane16 = (unsigned)lane * 16u;
  auto stage_src = [&](int s) __attribute__((always_inline)) -> const char* {
    const int ks = s / OG, g = s - ks * OG;
    return Wb + ((size_t)(4 * g + w + (n0 >> 5)) * KC + 8 * ks) * 512 + lane16;
  };
  const unsigned sW_lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
#pragma unroll
  for (int s0 = 0; s0 < R - 1; ++s0) {
    const __attribute__((address_space(1))) void* src = (const __attribute__((address_space(1))) void*)stage_src(s0);
    __attribute__((address_space(3))) void* dst = (__attribute__((address_space(3))) void*)(smem + s0 * PE_STAGE + w * 4096);
    __builtin_amdgcn_global_load_lds(src, dst, 16, 0, 0);
    __builtin_amdgcn_global_load_lds(src, dst, 16, 1024, 0);
    __builtin_amdgcn_global_load_lds(src, dst, 16, 2048, 0);
    __builtin_amdgcn_global_load_lds(src, dst, 16, 3072, 0);
  }
  auto issue_stage_asm = [&](int s) __attribute__((always_inline)) {
    const char* src = stage_src(s);
    const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(sW_lds + (unsigned)((s & (R - 1)) * PE_STAGE) + (unsigned)w * 4096u));
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\t"
                 "s_mov_b32 m0, %2\n\t"
                 "s_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, off\n\t"
                 "global_load_lds_dwordx4 %1, off offset:1024\n\t"
                 "global_load_lds_dwordx4 %1, off offset:2048\n\t"
                 "global_load_lds_dwordx4 %1, off offset:3072\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
  };

  auto to_frags = [&](const Slab& sl, V8 (&xf)[4]) __attribute__((always_inline)) {
#pragma unroll
    for (int c4 = 0; c4 < 4; ++c4) {
      if constexpr (X16) xf[c4] = __builtin_bit_cast(V8, sl.v[c4]);
      else {
        const f32x4 lo = __builtin_bit_cast(f32x4, sl.v[2 * c4]), hi = __builtin_bit_cast(f32x4, sl.v[2 * c4 + 1]);
        const u32x2 p0 = pack4<E>(lo[0], lo[1], lo[2], lo[3]);
        const u32x2 p1 = pack4<E>(hi[0], hi[1], hi[2], hi[3]);
        const u32x4 q = {p0[0], p0[1], p1[0], p1[1]};
        xf[c4] = __builtin_bit_cast(V8, q);
      }
    }
  };

  f32x16 acc[OT];
#pragma unroll
  for (int t = 0; t < OT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  V8 xf[4];

  const int wo = half * 512 + r31 * 16;                  // + (row block i * 8 + 2 * c4) * 512 inside a stage
  struct WF { V8 w[4]; };
  WF wf;
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"((R - 2) * 4) : "memory");   // stage 0 (own pieces; the first pixel loads are older) ...
  __builtin_amdgcn_s_barrier();                          // ... and everybody's
  asm volatile("" ::: "memory");
#pragma unroll
  for (int i = 0; i < 4; ++i) wf.w[i] = *reinterpret_cast<const V8*>(smem + wo + (i * 8) * 512);

  // stage S (compile time): 16 MFMAs = 4 k16 steps x 4 tiles of group g; fragments refilled in a rolling fashion (mlp_kernel.hpp);
  // middle of the stage: stage S+1 has landed for everybody, slot of stage S-1 takes stage S+R-1
  pfor<0, NS>([&](auto S_) {
    constexpr int S = decltype(S_)::value;
    constexpr int ks = S / OG, g = S % OG;
    constexpr bool more = S + R - 1 < NS, next = S + 1 < NS;
    const char* st = smem + (S & (R - 1)) * PE_STAGE;
    const char* stn = smem + ((S + 1) & (R - 1)) * PE_STAGE;
    if constexpr (g == 0) {                              // a new slab: its pixels (requested two slabs ago) become fragments
      to_frags(sl[ks % NSL], xf);
      if constexpr (ks + NSL < KS) load_slab(sl[ks % NSL], ks + NSL);
    }
    pfor<0, 4>([&](auto C4) {
      constexpr int c4 = decltype(C4)::value;
      if constexpr (c4 == 2) {
        // stage S+1 (own pieces) has landed: at least the R-3 younger stages' 4 pieces each may stay in flight (pixel loads issued
        // since only add younger operations: the count stays sufficient)
        if constexpr (S + R - 2 < NS) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((R - 3) * 4) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      }
      pfor<0, 4>([&](auto I) {
        constexpr int i = decltype(I)::value;
        acc[4 * g + i] = Op16<E>::mfma(wf.w[i], xf[c4], acc[4 * g + i]);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (c4 < 3) wf.w[i] = *reinterpret_cast<const V8*>(st + wo + (i * 8 + 2 * (c4 + 1)) * 512);
        else if constexpr (next) wf.w[i] = *reinterpret_cast<const V8*>(stn + wo + (i * 8) * 512);
        if constexpr (more && c4 == 2 && i == 0) issue_stage_asm(S + R - 1);
      });
    });
  });

  // ---- epilogue: registers 4q..4q+3 of tile t = features 32t + 8q + 4half .. +3 of the lane's patch
  if (m < total) {
    const int64_t orow = (int64_t)img * (a.P + 1) + 1 + p;
    const float* posr = a.pos + (int64_t)(1 + p) * a.D + n0;
    const float* biasr = a.bias + n0;
    char* ob = reinterpret_cast<char*>(a.out) + ((orow >> 5) * (a.D >> 2) + (n0 >> 2)) * 512 + (orow & 31) * 16;   // blk_off(orow, n0 / 4 + chunk, a.D / 4) = ob + chunk * 512
    pfor<0, OT>([&](auto T_) {
      constexpr int t = decltype(T_)::value;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int f0 = 32 * t + 8 * q + 4 * half;
        const f32x4 bv = *reinterpret_cast<const f32x4*>(biasr + f0);
        const f32x4 pv = *reinterpret_cast<const f32x4*>(posr + f0);
        const f32x4 o = {acc[t][4 * q] + bv[0] + pv[0], acc[t][4 * q + 1] + bv[1] + pv[1], acc[t][4 * q + 2] + bv[2] + pv[2],
                         acc[t][4 * q + 3] + bv[3] + pv[3]};
        if (PATCH_NT & 2) __builtin_nontemporal_store(o, reinterpret_cast<f32x4*>(ob + (f0 >> 2) * 512)); else *reinterpret_cast<f32x4*>(ob + (f0 >> 2) * 512) = o;
      }
    });
    if (a.cls != nullptr && p == 0) {                      // the class-token row of this image (token 0 = row orow - 1), this slice's features
      const int64_t crow = orow - 1;
      char* cb = reinterpret_cast<char*>(a.out) + ((crow >> 5) * (a.D >> 2) + (n0 >> 2)) * 512 + (crow & 31) * 16;
      for (int c = half; c < D / 4; c += 2)
        *reinterpret_cast<f32x4*>(cb + c * 512) = *reinterpret_cast<const f32x4*>(a.cls + n0 + 4 * c);
    }
  }
}

template <typename E, typename X>
int launch_patch(const PatchArgs& a, hipStream_t s) {
  const int64_t total = (int64_t)a.B * a.P;
  const int64_t panels = (total + 127) / 128;
  // Small calls (round 6): a 384-wide panel streams the whole 590 KB weight into ONE CU (42 us for 1 .. 16 crops, 2 .. 25 workgroups);
  // as three 128-wide slices — same arithmetic per output, the pixels of a panel are re-read out of the L2 — a third of the weight
  // stream and of the MFMAs per CU, on three times the CUs, while the slices still fit one round
  if (a.D == 384 && panels * 3 <= device_cus()) {
    hipLaunchKernelGGL((patch_embed_kernel<E, X, 128>), dim3((unsigned)(panels * 3)), dim3(256), 0, s, a);
    return check_launch("patch_embed_fused");
  }
  const int nsl = a.D > 384 ? a.D / 384 : 1;              // 384-wide output slices of a panel (ViT-B: 2)
  const dim3 grid((unsigned)(panels * nsl)), blk(256);
  switch (a.D) {
    case 128: hipLaunchKernelGGL((patch_embed_kernel<E, X, 128>), grid, blk, 0, s, a); break;
    case 256: hipLaunchKernelGGL((patch_embed_kernel<E, X, 256>), grid, blk, 0, s, a); break;
    case 384: case 768: hipLaunchKernelGGL((patch_embed_kernel<E, X, 384>), grid, blk, 0, s, a); break;
    default: return fail(EFFOCR_EUNSUPPORTED, "patch_embed_fused: embed dim must be 128, 256, 384 or 768");
  }
  return check_launch("patch_embed_fused");
}

}  // namespace

bool patch_embed_fused_supported(int prec, int D) { return (prec == PREC_BF16 || prec == PREC_FP16) && (D == 128 || D == 256 || D == 384 || D == 768); }

int patch_embed_fused(int prec, const PatchArgs& a, hipStream_t s) {
  if (a.B <= 0) return EFFOCR_OK;
  if (!patch_embed_fused_supported(prec, a.D)) return fail(EFFOCR_EUNSUPPORTED, "patch_embed_fused: needs bf16/fp16 and an embed dim of 128, 256, 384 or 768");
  if (a.H % 16 || a.W % 16 || a.P != (a.H / 16) * (a.W / 16)) return fail(EFFOCR_EINVAL, "patch_embed_fused: image size must be a multiple of 16");
  if (a.x16) return prec == PREC_BF16 ? launch_patch<__bf16, __bf16>(a, s) : launch_patch<_Float16, _Float16>(a, s);
  return prec == PREC_BF16 ? launch_patch<__bf16, float>(a, s) : launch_patch<_Float16, float>(a, s);
}

}  // namespace effocr
