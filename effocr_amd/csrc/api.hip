// C ABI of libeffocr_hip.so (include/effocr_hip.h): error plumbing, the encoder handle (parameter
// table, host-side packing / BatchNorm folding, forward orchestration) and thin wrappers over the
// kernel launchers.  All device memory is caller-owned; this file allocates host memory only.
#include "../../include/effocr_hip.h"
#include "common.hpp"
#include "kernels.hpp"


#include <math.h>
#include <string.h>
#include <map>
#include <memory>
#include <string>
#include <vector>

namespace effocr {

static thread_local std::string g_err;

void set_error(const std::string& msg) { g_err = msg; }
int fail(int code, const std::string& msg) { g_err = msg; return code; }
int check_launch(const char* what) {
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(EFFOCR_EHIP, std::string(what) + ": " + hipGetErrorString(e));
  return EFFOCR_OK;
}

int device_cus() {
  static int cache[64] = {0};                            // benign race: every thread computes the same value
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  if (cache[dev] == 0) {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
    cache[dev] = v;
  }
  return cache[dev];
}

namespace {

struct Param { std::string name; std::vector<int64_t> shape; int64_t numel; std::vector<float> data; bool set; };

struct VitCfg { int D, depth, heads, mlp; };
struct VitLayerOff { size_t ln1w, ln1b, qkvb, projb, ln2w, ln2b, fc1b, fc2b, qkvw, projw, fc1w, fc2w;
                     size_t qkvw_b, projw_b, fc1w_b, fc2w_b, projw_pp, fc2w_pp, projb_p, fc2b_p, qkvw_bv, qkvb_v;
                     size_t qkvw_bf, qkv_s, qkv_c, fc1w_bf, fc1_s, fc1_c; };   // _bf: blocked copy of W . diag(gamma) of the LayerNorm in front; _s: its row sums; _c: b + W beta (gemm3's folded LayerNorm)   // qkvw_bv / qkvb_v: the v rows permuted per 32 (qkvattn.hip)   // _pp / b_p: + rows permuted per 32 (proj fused into the MLP kernel)   // fc2w_pp: k also permuted per 16 (fused MLP)   // *_b: fragment-blocked copies (gemm3), 16-bit modes only
struct ConvSpec { std::string w, bn; int cin, cout, k, stride, pad; size_t w_off, b_off; };

}  // namespace
}  // namespace effocr

using namespace effocr;

struct effocr_encoder {
  std::string arch;
  bool is_vit = false;
  int img = 224, prec = PREC_BF16, D = 0;
  std::vector<Param> params;
  std::map<std::string, int> index;
  // ViT
  VitCfg vit{};
  int T = 0, P = 0;
  size_t off_clspos0 = 0, off_pos = 0, off_patchb = 0, off_normw = 0, off_normb = 0, off_patchw = 0, off_patchw_b = 0;   // _b: fragment-blocked copy (patch.hip)
  std::vector<VitLayerOff> layers;
  // resnet18
  std::vector<ConvSpec> convs;      // conv1, then per block conv1, conv2, (downsample)
  size_t wbytes = 0;
  const char* wdev = nullptr;       // device blob after upload
  // optional HIP-event profiler (effocr_encoder_profile_*): one event pair per launch of the
  // selected kernel classes, recorded on the forward's own stream
  int debug = 0;
  int cls_only_last = 1;            // last block: attn.proj + MLP only on the class-token rows (the only rows that reach the output); 0: all tokens (A/B switch)
  int mlp_stagger_min_rounds = 2;   // ... from this many rounds of CUs on (512-crop calls: +7.7 %, 384: +3.5 %, same box; no effect below two rounds)
  int mlp_stagger = 3500;           // fused MLP: start spread of the first round of workgroups, clock ticks per step of 32 (0 = off)
  int use_projf = 1;                // 1: attn.proj + residual fused in front of the fused MLP kernel (the new row stays in the accumulators: -0.9 ms and -0.6 GB of HBM traffic per forward vs the separate row-panel launch); 0: A/B switch
  int use_qkvattn = 1;              // fused norm1 + attn.qkv + attention kernel (qkvattn.hip): no qkv tensor in HBM (0: A/B switch)
  int qa_min_batch = 1;             // fused qkv+attention from this many crops per call on (below: the token-panel LN+qkv kernel + attention kernel)
  int qa_hsplit = 0;                // qkvattn head split: 0 = launcher's choice (small batches: several workgroups per image), 1 = never, n = at most n
  int use_patchf = 1;               // fused im2col + patch-embed GEMM (patch.hip) on the blocked path (0: im2col kernel + gemm2, A/B switch)
  int use_mlp = 1;                  // fused LN2+fc1+GELU+fc2+residual kernel (mlp.hip) on the blocked panel path (0: A/B switch)
  int use_gemm3 = 1;                // 128-row wave-tile GEMM (gemm3.hip) where the blocked layout allows (0: A/B switch)
  int use_lnfold = 1;               // gemm3 path (ViT-B): LayerNorm folded into the residual producers' / qkv, fc1 consumers' epilogues (0: LayerNorm launches, A/B switch)
  int tail_split = 1;               // cut the panels of the last, partially filled round along N (0: A/B switch)
  int split6 = 1;                   // fused MLP: 6-way hidden split for calls of <= 27 crops (0: A/B switch)
  int pair_parts = 1;               // fused MLP: calls of <= 27 crops as 3-way split 64-token pair panels (0: the 6- / 4-way 128-token parts; A/B switch)
  int mlp_pair = 0;                 // fused MLP: 64-token panels on wave pairs: 0 = auto (30-83 crops), 1 = whenever they fit one round, -1 = never (A/B switch)
  int use_blocked = 1;              // fragment-blocked activation layout on the panel path (0: row-major, A/B switch)
  int use_gemm2 = 1;                // 1: glds-ring K-streaming GEMM for fc2 / patch embed, 0: register-staged gemm.hip
  int panel_rows = 128;             // row-panel height: 128 (1 workgroup/CU) or 64 (2 workgroups/CU)
  int use_panel = 1;                // 0: force the K-streaming GEMM + standalone LayerNorm path (A/B switch)
  int chunk = 0;                    // crops per internal sub-batch of the ViT forward (0 = whole batch)
  int prof_mode = 0;                // 0 off, 1 every class, 2 only prof_only
  std::string prof_only;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_pool;
  std::vector<std::pair<int, int>> prof_rec;      // (class id, pool slot)
  std::vector<std::string> prof_names;
  size_t prof_used = 0;
  std::vector<float> prof_ms;                     // filled by profile_collect
  std::vector<int> prof_cnt;
  std::vector<double> prof_work;                  // algorithmic flops (or bytes) per class, summed
  // mode 1 also brackets every launch with two clock samples (s_memtime = shader clocks, s_memrealtime = 100 MHz) outside its event pair
  unsigned long long* prof_clk = nullptr;         // device: [slot][2][2048 CU keys][2] = (memtime, realtime) per CU, before and after (16 MB, allocated by the first mode-1 run)
  std::vector<double> prof_ghz;                   // filled by profile_collect (0 = not sampled)
  static constexpr size_t PROF_CLK_SLOTS = 256;
};

namespace effocr {
namespace {

int64_t prod(const std::vector<int64_t>& s) { int64_t p = 1; for (auto v : s) p *= v; return p; }

void add_param(effocr_encoder* e, const std::string& name, std::vector<int64_t> shape) {
  Param p; p.name = name; p.shape = shape; p.numel = prod(shape); p.set = false;
  e->index[name] = (int)e->params.size();
  e->params.push_back(std::move(p));
}

struct Alloc {
  size_t off = 0;
  size_t take(size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; }
};

void build_vit(effocr_encoder* e) {
  const int D = e->vit.D, depth = e->vit.depth, mlp = e->vit.mlp;
  e->P = (e->img / 16) * (e->img / 16);
  e->T = e->P + 1;
  add_param(e, "cls_token", {1, 1, D});
  add_param(e, "pos_embed", {1, e->T, D});
  add_param(e, "patch_embed.proj.weight", {D, 3, 16, 16});
  add_param(e, "patch_embed.proj.bias", {D});
  for (int i = 0; i < depth; ++i) {
    const std::string p = "blocks." + std::to_string(i) + ".";
    add_param(e, p + "norm1.weight", {D});
    add_param(e, p + "norm1.bias", {D});
    add_param(e, p + "attn.qkv.weight", {3 * D, D});
    add_param(e, p + "attn.qkv.bias", {3 * D});
    add_param(e, p + "attn.proj.weight", {D, D});
    add_param(e, p + "attn.proj.bias", {D});
    add_param(e, p + "norm2.weight", {D});
    add_param(e, p + "norm2.bias", {D});
    add_param(e, p + "mlp.fc1.weight", {mlp, D});
    add_param(e, p + "mlp.fc1.bias", {mlp});
    add_param(e, p + "mlp.fc2.weight", {D, mlp});
    add_param(e, p + "mlp.fc2.bias", {D});
  }
  add_param(e, "norm.weight", {D});
  add_param(e, "norm.bias", {D});

  const size_t es = prec_esize(e->prec);
  Alloc a;
  e->off_clspos0 = a.take((size_t)D * 4);
  e->off_pos = a.take((size_t)e->T * D * 4);
  e->off_patchb = a.take((size_t)D * 4);
  e->layers.resize(depth);
  for (int i = 0; i < depth; ++i) {
    VitLayerOff& L = e->layers[i];
    L.ln1w = a.take((size_t)D * 4); L.ln1b = a.take((size_t)D * 4);
    L.qkvb = a.take((size_t)3 * D * 4); L.projb = a.take((size_t)D * 4);
    L.ln2w = a.take((size_t)D * 4); L.ln2b = a.take((size_t)D * 4);
    L.fc1b = a.take((size_t)mlp * 4); L.fc2b = a.take((size_t)D * 4);
  }
  e->off_normw = a.take((size_t)D * 4);
  e->off_normb = a.take((size_t)D * 4);
  e->off_patchw = a.take((size_t)D * 768 * es);
  for (int i = 0; i < depth; ++i) {
    VitLayerOff& L = e->layers[i];
    L.qkvw = a.take((size_t)3 * D * D * es);
    L.projw = a.take((size_t)D * D * es);
    L.fc1w = a.take((size_t)mlp * D * es);
    L.fc2w = a.take((size_t)D * mlp * es);
  }
  if (e->prec != PREC_FP32) {
    e->off_patchw_b = a.take((size_t)D * 768 * es);
    for (int i = 0; i < depth; ++i) {
      VitLayerOff& L = e->layers[i];
      L.fc2w_b = a.take((size_t)D * mlp * es);
      L.fc2w_pp = a.take((size_t)D * mlp * es);
      L.projw_pp = a.take((size_t)D * D * es);
      L.projb_p = a.take((size_t)D * 4);
      L.fc2b_p = a.take((size_t)D * 4);
      L.qkvw_bv = a.take((size_t)3 * D * D * es);
      L.qkvb_v = a.take((size_t)3 * D * 4);
      {                                                  // gemm3 can run every linear (default where no row-panel kernel exists)
        L.qkvw_b = a.take((size_t)3 * D * D * es);
        L.projw_b = a.take((size_t)D * D * es);
        L.fc1w_b = a.take((size_t)mlp * D * es);
        L.qkvw_bf = a.take((size_t)3 * D * D * es); L.qkv_s = a.take((size_t)3 * D * 4); L.qkv_c = a.take((size_t)3 * D * 4);
        L.fc1w_bf = a.take((size_t)mlp * D * es); L.fc1_s = a.take((size_t)mlp * 4); L.fc1_c = a.take((size_t)mlp * 4);
      }
    }
  }
  e->wbytes = a.off;
}

void add_bn(effocr_encoder* e, const std::string& p, int c) {
  add_param(e, p + ".weight", {c}); add_param(e, p + ".bias", {c});
  add_param(e, p + ".running_mean", {c}); add_param(e, p + ".running_var", {c});
}

constexpr int CONV1_KPAD = 160;   // 7*7*3 = 147 im2col columns padded to 5 K-stages of 32

void build_resnet18(effocr_encoder* e) {
  const int widths[4] = {64, 128, 256, 512};
  add_param(e, "conv1.weight", {64, 3, 7, 7});
  add_bn(e, "bn1", 64);
  e->convs.push_back({"conv1.weight", "bn1", 3, 64, 7, 2, 3, 0, 0});
  int cin = 64;
  for (int li = 1; li <= 4; ++li) {
    const int w = widths[li - 1];
    for (int bi = 0; bi < 2; ++bi) {
      const std::string p = "layer" + std::to_string(li) + "." + std::to_string(bi) + ".";
      const int stride = (bi == 0 && li > 1) ? 2 : 1;
      add_param(e, p + "conv1.weight", {w, cin, 3, 3}); add_bn(e, p + "bn1", w);
      add_param(e, p + "conv2.weight", {w, w, 3, 3}); add_bn(e, p + "bn2", w);
      e->convs.push_back({p + "conv1.weight", p + "bn1", cin, w, 3, stride, 1, 0, 0});
      e->convs.push_back({p + "conv2.weight", p + "bn2", w, w, 3, 1, 1, 0, 0});
      if (bi == 0 && li > 1) {
        add_param(e, p + "downsample.0.weight", {w, cin, 1, 1}); add_bn(e, p + "downsample.1", w);
        e->convs.push_back({p + "downsample.0.weight", p + "downsample.1", cin, w, 1, stride, 0, 0, 0});
      }
      cin = w;
    }
  }
  Alloc a;
  for (size_t i = 0; i < e->convs.size(); ++i) {
    ConvSpec& c = e->convs[i];
    const size_t K = (i == 0) ? (size_t)CONV1_KPAD : (size_t)c.k * c.k * c.cin;
    c.w_off = a.take((size_t)c.cout * K * 4);
    c.b_off = a.take((size_t)c.cout * 4);
  }
  e->wbytes = a.off;
}

uint16_t f32_to_bf16(float f) {
  uint32_t u; memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);   // NaN
  u += 0x7fffu + ((u >> 16) & 1u);                                            // round to nearest even
  return (uint16_t)(u >> 16);
}
uint16_t f32_to_f16(float f) { _Float16 h = (_Float16)f; uint16_t u; memcpy(&u, &h, 2); return u; }

void put_f32(std::vector<char>& blob, size_t off, const float* src, size_t n) { memcpy(blob.data() + off, src, n * 4); }
void put_op(std::vector<char>& blob, size_t off, const float* src, size_t n, int prec) {
  if (prec == PREC_FP32) { memcpy(blob.data() + off, src, n * 4); return; }
  uint16_t* d = reinterpret_cast<uint16_t*>(blob.data() + off);
  if (prec == PREC_BF16) for (size_t i = 0; i < n; ++i) d[i] = f32_to_bf16(src[i]);
  else for (size_t i = 0; i < n; ++i) d[i] = f32_to_f16(src[i]);
}

// [N,K] fp32 -> 16-bit fragment-blocked [N/32][K/8][32 rows][8 elements] (common.hpp blk_off); N % 32 == 0, K % 8 == 0
// perm16: element e of chunk c holds k = 16*(c/2) + (e&3) + 8*(e>>2) + 4*(c&1) instead of 8c + e — the order in
// which the swapped-MFMA C-layout of the previous GEMM hands its values over as a B-operand (mlp.hip); K % 16 == 0.
// rowperm: packed row 32b + p holds source row 32b + rowperm32(p) — then the swapped C-layout gives every lane 8
// CONSECUTIVE output features per register octet (= the B-operand / LayerNorm layout of the next stage).
int rowperm32(int p) {
  const int h = (p >> 2) & 1, r = (p & 3) + 4 * (p >> 3);
  return 8 * (2 * (r >> 3) + h) + (r & 7);
}
void put_op_blocked(std::vector<char>& blob, size_t off, const float* src, int N, int K, int prec, bool perm16 = false, bool rowperm = false,
                    int rowperm_from = 0) {                // rowperm applies to rows >= rowperm_from (a multiple of 32)
  uint16_t* d = reinterpret_cast<uint16_t*>(blob.data() + off);
  const int kch = K / 8;
  for (int n = 0; n < N; ++n) {
    const int ns = (rowperm && n >= rowperm_from) ? (n & ~31) + rowperm32(n & 31) : n;
    for (int c = 0; c < kch; ++c) {
      uint16_t* cell = d + ((size_t)(n >> 5) * kch + c) * 256 + (n & 31) * 8;
      for (int e = 0; e < 8; ++e) {
        const int k = perm16 ? 16 * (c >> 1) + (e & 3) + 8 * (e >> 2) + 4 * (c & 1) : c * 8 + e;
        const float v = src[(size_t)ns * K + k];
        cell[e] = prec == PREC_BF16 ? f32_to_bf16(v) : f32_to_f16(v);
      }
    }
  }
}
void put_f32_rowperm(std::vector<char>& blob, size_t off, const float* src, int n) {
  float* d = reinterpret_cast<float*>(blob.data() + off);
  for (int i = 0; i < n; ++i) d[i] = src[(i & ~31) + rowperm32(i & 31)];
}

const std::vector<float>& P(const effocr_encoder* e, const std::string& n) { return e->params[e->index.at(n)].data; }

// LayerNorm folded into the linear behind it (gemm3.hip, GemmArgs::lnf): y = W (gamma (x - mean) rstd + beta) + b
//   = rstd (W' x - mean s) + c   with   W' = W diag(gamma) (rounded to the operand type, blocked),  s[n] = sum_k W'[n][k] (of the ROUNDED
// values: exactly what the MFMAs multiply),  c = b + W beta (fp32).
void put_lnfold(std::vector<char>& blob, size_t w_off, size_t s_off, size_t c_off, const float* W, const float* b, const float* gamma,
                const float* beta, int N, int K, int prec) {
  std::vector<float> wf((size_t)N * K);
  float* sd = reinterpret_cast<float*>(blob.data() + s_off);
  float* cd = reinterpret_cast<float*>(blob.data() + c_off);
  for (int n = 0; n < N; ++n) {
    double ss = 0.0, cc = b[n];
    for (int k = 0; k < K; ++k) {
      const float v = W[(size_t)n * K + k] * gamma[k];
      wf[(size_t)n * K + k] = v;
      float r;
      if (prec == PREC_BF16) { const uint32_t u = (uint32_t)f32_to_bf16(v) << 16; memcpy(&r, &u, 4); }
      else r = (float)(_Float16)v;
      ss += (double)r;
      cc += (double)W[(size_t)n * K + k] * (double)beta[k];
    }
    sd[n] = (float)ss; cd[n] = (float)cc;
  }
  put_op_blocked(blob, w_off, wf.data(), N, K, prec);
}

void pack_vit(const effocr_encoder* e, std::vector<char>& blob) {
  const int D = e->vit.D;
  std::vector<float> cp(D);
  const auto& cls = P(e, "cls_token"); const auto& pos = P(e, "pos_embed");
  for (int d = 0; d < D; ++d) cp[d] = cls[d] + pos[d];
  put_f32(blob, e->off_clspos0, cp.data(), D);
  put_f32(blob, e->off_pos, pos.data(), pos.size());
  put_f32(blob, e->off_patchb, P(e, "patch_embed.proj.bias").data(), D);
  put_op(blob, e->off_patchw, P(e, "patch_embed.proj.weight").data(), (size_t)D * 768, e->prec);
  if (e->prec != PREC_FP32 && D % 32 == 0) put_op_blocked(blob, e->off_patchw_b, P(e, "patch_embed.proj.weight").data(), D, 768, e->prec);
  for (int i = 0; i < e->vit.depth; ++i) {
    const std::string p = "blocks." + std::to_string(i) + ".";
    const VitLayerOff& L = e->layers[i];
    put_f32(blob, L.ln1w, P(e, p + "norm1.weight").data(), D);
    put_f32(blob, L.ln1b, P(e, p + "norm1.bias").data(), D);
    put_f32(blob, L.qkvb, P(e, p + "attn.qkv.bias").data(), 3 * (size_t)D);
    put_f32(blob, L.projb, P(e, p + "attn.proj.bias").data(), D);
    put_f32(blob, L.ln2w, P(e, p + "norm2.weight").data(), D);
    put_f32(blob, L.ln2b, P(e, p + "norm2.bias").data(), D);
    put_f32(blob, L.fc1b, P(e, p + "mlp.fc1.bias").data(), e->vit.mlp);
    put_f32(blob, L.fc2b, P(e, p + "mlp.fc2.bias").data(), D);
    put_op(blob, L.qkvw, P(e, p + "attn.qkv.weight").data(), 3 * (size_t)D * D, e->prec);
    put_op(blob, L.projw, P(e, p + "attn.proj.weight").data(), (size_t)D * D, e->prec);
    put_op(blob, L.fc1w, P(e, p + "mlp.fc1.weight").data(), (size_t)e->vit.mlp * D, e->prec);
    put_op(blob, L.fc2w, P(e, p + "mlp.fc2.weight").data(), (size_t)D * e->vit.mlp, e->prec);
    if (e->prec != PREC_FP32) {
      put_op_blocked(blob, L.fc2w_b, P(e, p + "mlp.fc2.weight").data(), D, e->vit.mlp, e->prec);
      put_op_blocked(blob, L.fc2w_pp, P(e, p + "mlp.fc2.weight").data(), D, e->vit.mlp, e->prec, true, true);
      put_op_blocked(blob, L.projw_pp, P(e, p + "attn.proj.weight").data(), D, D, e->prec, false, true);
      put_f32_rowperm(blob, L.projb_p, P(e, p + "attn.proj.bias").data(), D);
      put_f32_rowperm(blob, L.fc2b_p, P(e, p + "mlp.fc2.bias").data(), D);
      put_op_blocked(blob, L.qkvw_bv, P(e, p + "attn.qkv.weight").data(), 3 * D, D, e->prec, false, true, 2 * D);
      put_f32(blob, L.qkvb_v, P(e, p + "attn.qkv.bias").data(), 2 * (size_t)D);
      put_f32_rowperm(blob, L.qkvb_v + (size_t)2 * D * 4, P(e, p + "attn.qkv.bias").data() + 2 * D, D);
      {
        put_op_blocked(blob, L.qkvw_b, P(e, p + "attn.qkv.weight").data(), 3 * D, D, e->prec);
        put_op_blocked(blob, L.projw_b, P(e, p + "attn.proj.weight").data(), D, D, e->prec);
        put_op_blocked(blob, L.fc1w_b, P(e, p + "mlp.fc1.weight").data(), e->vit.mlp, D, e->prec);
        put_lnfold(blob, L.qkvw_bf, L.qkv_s, L.qkv_c, P(e, p + "attn.qkv.weight").data(), P(e, p + "attn.qkv.bias").data(),
                   P(e, p + "norm1.weight").data(), P(e, p + "norm1.bias").data(), 3 * D, D, e->prec);
        put_lnfold(blob, L.fc1w_bf, L.fc1_s, L.fc1_c, P(e, p + "mlp.fc1.weight").data(), P(e, p + "mlp.fc1.bias").data(),
                   P(e, p + "norm2.weight").data(), P(e, p + "norm2.bias").data(), e->vit.mlp, D, e->prec);
      }
    }
  }
  put_f32(blob, e->off_normw, P(e, "norm.weight").data(), D);
  put_f32(blob, e->off_normb, P(e, "norm.bias").data(), D);
}

// BatchNorm (eval, eps 1e-5) folded into the conv: w' = w * g/sqrt(v+eps), b' = beta - mean*g/sqrt(v+eps);
// weight re-laid-out from torch [Cout,Cin,KH,KW] to [Cout][KH][KW][Cin] (K-contiguous, Cin fastest).
void pack_resnet(const effocr_encoder* e, std::vector<char>& blob) {
  for (size_t ci = 0; ci < e->convs.size(); ++ci) {
    const ConvSpec& c = e->convs[ci];
    const auto& w = P(e, c.w);
    const auto& g = P(e, c.bn + ".weight"); const auto& bt = P(e, c.bn + ".bias");
    const auto& mu = P(e, c.bn + ".running_mean"); const auto& var = P(e, c.bn + ".running_var");
    const int K = c.k * c.k * c.cin;
    const int Kp = (ci == 0) ? CONV1_KPAD : K;
    float* wd = reinterpret_cast<float*>(blob.data() + c.w_off);
    float* bd = reinterpret_cast<float*>(blob.data() + c.b_off);
    for (int co = 0; co < c.cout; ++co) {
      const double sc = (double)g[co] / sqrt((double)var[co] + 1e-5);
      bd[co] = (float)((double)bt[co] - (double)mu[co] * sc);
      for (int kk = 0; kk < Kp; ++kk) wd[(size_t)co * Kp + kk] = 0.f;
      for (int ky = 0; ky < c.k; ++ky)
        for (int kx = 0; kx < c.k; ++kx)
          for (int cc = 0; cc < c.cin; ++cc) {
            const float v = w[(((size_t)co * c.cin + cc) * c.k + ky) * c.k + kx];
            wd[(size_t)co * Kp + (ky * c.k + kx) * c.cin + cc] = (float)((double)v * sc);
          }
    }
  }
}

int prof_class(effocr_encoder* e, const char* name) {
  for (size_t i = 0; i < e->prof_names.size(); ++i) if (e->prof_names[i] == name) return (int)i;
  e->prof_names.push_back(name);
  e->prof_work.push_back(0.0);
  return (int)e->prof_names.size() - 1;
}

// Runs `launch` (which enqueues exactly one kernel class on stream s), bracketed by an event pair
// when the profiler is armed for that class.  `work` = algorithmic flops of the launch.
template <typename F>
int timed(effocr_encoder* e, const char* name, double work, hipStream_t s, F launch) {
  const bool on = e->prof_mode == 1 || (e->prof_mode == 2 && e->prof_only == name);
  if (!on) return launch();
  if (e->prof_used == e->prof_pool.size()) {
    hipEvent_t a, b;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return fail(EFFOCR_EHIP, "profile: hipEventCreate failed");
    e->prof_pool.push_back({a, b});
  }
  const int cls = prof_class(e, name);
  const size_t slot = e->prof_used++;
  const bool clk = e->prof_mode == 1 && slot < effocr_encoder::PROF_CLK_SLOTS &&
                   (e->prof_clk || hipMalloc(&e->prof_clk, effocr_encoder::PROF_CLK_SLOTS * 8192 * sizeof(unsigned long long)) == hipSuccess && hipMemset(e->prof_clk, 0, effocr_encoder::PROF_CLK_SLOTS * 8192 * sizeof(unsigned long long)) == hipSuccess);
  if (clk) (void)clock_sample(e->prof_clk + slot * 8192, s);
  (void)hipEventRecord(e->prof_pool[slot].first, s);
  const int rc = launch();
  (void)hipEventRecord(e->prof_pool[slot].second, s);
  if (clk) (void)clock_sample(e->prof_clk + slot * 8192 + 4096, s);
  e->prof_rec.push_back({cls, (int)slot});
  e->prof_work[cls] += work;
  return rc;
}

struct VitWs { size_t status, x, xn, qkv, att, h, stats, total, rows, hbytes; };
VitWs vit_ws(const effocr_encoder* e, int B) {
  // rows padded to the panel height (128) so that the row-panel kernels store without bounds checks
  const size_t M = align_up((size_t)B * e->T, 128), D = e->vit.D, es = prec_esize(e->prec);
  Alloc a; VitWs w;
  w.rows = M;
  w.status = a.take(256);                    // int32 status word at workspace offset 0 (effocr_encoder_check_status)
  w.x = a.take(M * D * 4);
  w.xn = a.take(M * D * es);
  w.qkv = a.take(M * 3 * D * es);
  w.att = a.take(M * D * es);
  size_t hb = M * e->vit.mlp * es, pb = (size_t)B * e->P * 768 * es;
  w.hbytes = hb > pb ? hb : pb;
  {
    // the fused MLP cuts the panels of a partially filled round of CUs 4- or 2-way over the hidden dimension and needs
    // split x tail x 128 x D fp32 partial rows in this buffer (launch_mlp).  For calls of <= 32 crops the hidden-sized buffer is
    // SMALLER than the 4-way partials (M x 1536 x 2 bytes against 4 x M x 384 x 4): the launcher then silently ran whole panels on a
    // tenth of the chip (16 crops: 73 us per block on 25 CUs; round 6).  Size for the split the launcher will choose.
    const size_t cus = (size_t)device_cus(), np = M / 128, tail = np % cus;
    const size_t split = (tail && tail * 6 <= cus) ? 6 : (tail && tail * 4 <= cus) ? 4 : (tail && tail * 2 <= cus) ? 2 : 0;
    const size_t need = split * tail * 128 * D * 4;
    if (need > w.hbytes) w.hbytes = need;
  }
  w.h = a.take(w.hbytes);                   // patches (im2col rows) alias the MLP hidden buffer
  w.stats = a.take(M * 64);                 // per-row (sum, sum of squares) slice partials of the folded LayerNorm (gemm3 path)
  w.total = a.off;
  return w;
}

int vit_forward(effocr_encoder* e, const void* x, int x16, int B, float* emb, int l2, char* ws, hipStream_t s) {
  const VitWs w = vit_ws(e, B);
  const int D = e->vit.D, T = e->T, Pn = e->P, M = B * T, prec = e->prec;
  const char* wb = e->wdev;
  float* xs = reinterpret_cast<float*>(ws + w.x);
  void* xn = ws + w.xn; void* qkv = ws + w.qkv; void* att = ws + w.att; void* hb = ws + w.h;
  auto F = [&](size_t off) { return reinterpret_cast<const float*>(wb + off); };
  int rc;
  const double Md = (double)M, Dd = (double)D, Hd = (double)e->vit.mlp;
  const bool panel = e->use_panel && panel_gemm_supported(prec, 3 * D, D) && panel_gemm_supported(prec, e->vit.mlp, D);
  const bool g2 = e->use_gemm2 && gemm2_supported(prec, D, e->vit.mlp);
  const bool g2p = e->use_gemm2 && gemm2_supported(prec, D, 768);
  // fragment-blocked activations (x fp32, qkv, attention output, MLP hidden) need every producer/consumer on the fast path
  // widths without row-panel kernels (ViT-B): LayerNorm kernel + gemm3 for all four linears, everything blocked
  const bool g3all = !panel && e->use_gemm3 && gemm3_supported(prec, 3 * D, D) && gemm3_supported(prec, D, D) &&
                     gemm3_supported(prec, e->vit.mlp, D) && gemm3_supported(prec, D, e->vit.mlp);
  const int blk = (e->use_blocked && g2p && ((panel && g2) || g3all)) ? 1 : 0;
  const bool mlpf = blk && panel && e->use_mlp && mlp_fused_supported(prec, D, e->vit.mlp);
  const bool projf = mlpf && e->use_projf;
  // one image per workgroup at a time: worth it from ~3/4 of a round of CUs on; small batches (the reference's 64-crop calls)
  // keep the token-panel kernels, which spread 64 x 197 tokens over every CU.  use_qkvattn = 2 forces it (tests).
  const bool qaf = blk && panel && mlpf && qkv_attn_supported(prec, D, T) && (e->use_qkvattn == 2 || (e->use_qkvattn == 1 && B >= e->qa_min_batch));
  const bool g3 = blk && e->use_gemm3 && gemm3_supported(prec, D, e->vit.mlp);
  const bool patchf = blk && e->use_patchf && patch_embed_fused_supported(prec, D);
  if (!patchf && (rc = timed(e, "im2col_patch16", 0.0, s, [&] { return im2col_patch16(prec, x, x16, B, e->img, e->img, hb, s); }))) return rc;
  // the status word is STICKY: forwards only ever OR into it (final_cls_norm), effocr_encoder_check_status reads and clears it — so one
  // check covers every forward issued with this workspace since the previous check (sub-batches, slices of a large call, async callers)
  int* status = reinterpret_cast<int*>(ws + w.status);
  if (!patchf && (rc = set_cls_rows(F(e->off_clspos0), xs, B, T, D, blk, nullptr, s))) return rc;   // (the fused patch embedding writes the class-token rows itself)
  GemmArgs g{};
  if (patchf) {                                          // pixels -> tokens in one kernel: the patch rows never exist in HBM
    PatchArgs pa{};
    pa.x = x; pa.x16 = x16; pa.B = B; pa.H = e->img; pa.W = e->img; pa.Wb = wb + e->off_patchw_b; pa.bias = F(e->off_patchb); pa.pos = F(e->off_pos);
    pa.out = xs; pa.D = D; pa.P = Pn; pa.cls = F(e->off_clspos0);
    if ((rc = timed(e, "patch_embed_fused", 2.0 * B * Pn * Dd * 768.0, s, [&] { return patch_embed_fused(prec, pa, s); }))) return rc;
  } else {
  g.X = hb; g.ldx = 768; g.W = wb + e->off_patchw; g.ldw = 768; g.bias = F(e->off_patchb);
  g.out = xs; g.ldo = D; g.pos = F(e->off_pos); g.M = B * Pn; g.N = D; g.K = 768; g.P = Pn; g.blk_out = blk;
  if ((rc = timed(e, "gemm_patch_embed", 2.0 * B * Pn * Dd * 768.0, s, [&] { return g2p ? gemm2_nt(prec, EPI_PATCH, g, s) : gemm_nt(prec, EPI_PATCH, g, s); }))) return rc;
  }
  const float* cls_x = nullptr;                        // compact class-token rows after the last block (cls_only_last)
  bool xn_ready = false;                               // xn holds norm1(x) of the coming block (written by the previous block's MLP epilogue)
  for (int i = 0; i < e->vit.depth; ++i) {
    const VitLayerOff& L = e->layers[i];
    if (panel) {
      // row-panel kernels: LayerNorm fused into the A-panel load, no xn buffer, no LayerNorm launches
      {
      PanelArgs p{};
      if (qaf) {                                         // qkv + attention in one kernel, one image per workgroup at a time
        if (!xn_ready && (rc = timed(e, "layernorm", 0.0, s, [&] { return layernorm_rows_blocked(prec, xs, M, D, F(L.ln1w), F(L.ln1b), 1e-6f, xn, s); }))) return rc;
        xn_ready = false;
        QkvAttnArgs q{};
        q.xn = xn; q.Wb = wb + L.qkvw_bv; q.bias = F(L.qkvb_v); q.out = att;
        q.B = B; q.T = T; q.D = D; q.rows_alloc = (int64_t)w.rows; q.hsplit = e->qa_hsplit;
        q.cls_only = (i + 1 == e->vit.depth && e->cls_only_last && projf) ? 1 : 0;   // the rest of the last block runs on the class-token rows only
        // work executed: the class-token variant projects q and runs the attention for ONE 32-token tile per image (where it exists: D = 384, 193..224 tokens)
        const bool cls_k = q.cls_only && D == 384 && T > 192;
        const double qa_work = cls_k ? 2.0 * Md * 2.0 * Dd * Dd + 2.0 * (32.0 * B) * Dd * Dd + 4.0 * B * e->vit.heads * 32.0 * T * 64.0
                                     : 2.0 * Md * 3.0 * Dd * Dd + 4.0 * B * e->vit.heads * (double)T * T * 64.0;
        if ((rc = timed(e, "qkv_attn_fused", qa_work, s, [&] { return qkv_attn_fused(prec, q, s); }))) return rc;
      } else {
      p.A = xs; p.lda = D; p.gamma = F(L.ln1w); p.beta = F(L.ln1b); p.eps = 1e-6f; p.W = wb + L.qkvw; p.bias = F(L.qkvb);
      p.out = qkv; p.ldo = 3 * D; p.M = M; p.N = 3 * D; p.K = D; p.rows_padded = 1; p.debug = e->debug; p.panel_rows = e->panel_rows; p.no_tail_split = !e->tail_split;
      p.blk_a = blk; p.blk_out = blk;
      if ((rc = timed(e, "panel_ln_qkv", 2.0 * Md * 3.0 * Dd * Dd, s, [&] { return panel_gemm(prec, PRO_LN, EPI_BIAS, p, s); }))) return rc;
      if ((rc = timed(e, "attention", 4.0 * B * e->vit.heads * (double)T * T * 64.0, s, [&] { return attention(prec, qkv, att, B, T, e->vit.heads, blk, s); }))) return rc;
      }
      if (!projf) {
      p = PanelArgs{};
      p.A = att; p.lda = D; p.W = wb + L.projw; p.bias = F(L.projb); p.out = xs; p.ldo = D; p.resid = xs; p.ldr = D;
      p.M = M; p.N = D; p.K = D; p.rows_padded = 1; p.debug = e->debug; p.panel_rows = e->panel_rows; p.no_tail_split = !e->tail_split; p.blk_a = blk; p.blk_out = blk;
      if ((rc = timed(e, "panel_proj_resid", 2.0 * Md * Dd * Dd, s, [&] { return panel_gemm(prec, PRO_COPY, EPI_BIAS_RESID, p, s); }))) return rc;
      }
      }
      if (mlpf) {
        MlpArgs m{};
        m.x = xs; m.gamma = F(L.ln2w); m.beta = F(L.ln2b); m.eps = 1e-6f; m.W1b = wb + L.fc1w_b; m.b1 = F(L.fc1b);
        m.W2p = wb + L.fc2w_pp; m.b2 = F(L.fc2b_p); m.b2_logical = F(L.fc2b); m.M = M; m.D = D; m.H = e->vit.mlp; m.rows_alloc = (int)w.rows;
        m.partial = reinterpret_cast<float*>(hb); m.partial_bytes = w.hbytes; m.no_tail_split = !e->tail_split; m.no_split6 = !e->split6; m.pair = e->mlp_pair; m.no_pair_parts = !e->pair_parts; m.stagger = e->mlp_stagger; m.stagger_min_rounds = e->mlp_stagger_min_rounds;   // the hidden buffer is free on this path
        if (projf) {                                     // attn.proj + residual runs inside the same kernel, in front
          m.A = att; m.Wpp = wb + L.projw_pp; m.bp = F(L.projb_p);
          if (i + 1 == e->vit.depth && e->cls_only_last) {
            // last block: only row 0 of every image reaches the output and proj / LayerNorm / MLP act per row: run them on the B
            // gathered class-token rows (compact buffers in the qkv region, free from here on) instead of B*T rows
            const size_t Bp = align_up((size_t)B, 128);
            float* xc = reinterpret_cast<float*>(qkv);
            void* ac = static_cast<char*>(qkv) + Bp * D * 4;    // Bp*D*6 bytes <= the region's M*D*6 for every (B, T)
            if ((rc = timed(e, "gather_cls", 0.0, s, [&] { return gather_cls_rows_blocked(xs, att, B, T, D, xc, ac, s); }))) return rc;
            m.x = xc; m.A = ac; m.M = B; m.rows_alloc = (int)Bp;
            const double Bd = B;
            if ((rc = timed(e, "proj_mlp_cls", 4.0 * Bd * Hd * Dd + 2.0 * Bd * Dd * Dd, s, [&] { return mlp_fused(prec, m, s); }))) return rc;
            cls_x = xc;
            continue;
          }
          if (qaf && i + 1 < e->vit.depth) {
            m.xn_out = xn; m.gamma_n = F(e->layers[i + 1].ln1w); m.beta_n = F(e->layers[i + 1].ln1b);
            xn_ready = true;
          }
          if ((rc = timed(e, "proj_mlp_fused", 4.0 * Md * Hd * Dd + 2.0 * Md * Dd * Dd, s, [&] { return mlp_fused(prec, m, s); }))) return rc;
          continue;
        }
        if (qaf && i + 1 < e->vit.depth) {               // second output: the next block's norm1(x), consumed by its qkv+attention kernel
          m.xn_out = xn; m.gamma_n = F(e->layers[i + 1].ln1w); m.beta_n = F(e->layers[i + 1].ln1b);
          xn_ready = true;
        }
        if ((rc = timed(e, "mlp_fused", 4.0 * Md * Hd * Dd, s, [&] { return mlp_fused(prec, m, s); }))) return rc;
        continue;
      }
      PanelArgs p{};
      p.A = xs; p.lda = D; p.gamma = F(L.ln2w); p.beta = F(L.ln2b); p.eps = 1e-6f; p.W = wb + L.fc1w; p.bias = F(L.fc1b);
      p.out = hb; p.ldo = e->vit.mlp; p.M = M; p.N = e->vit.mlp; p.K = D; p.rows_padded = 1; p.debug = e->debug; p.panel_rows = e->panel_rows; p.no_tail_split = !e->tail_split;
      p.blk_a = blk; p.blk_out = blk;
      if ((rc = timed(e, "panel_ln_fc1_gelu", 2.0 * Md * Hd * Dd, s, [&] { return panel_gemm(prec, PRO_LN, EPI_BIAS_GELU, p, s); }))) return rc;
    } else if (blk) {
      // LayerNorm folded into the linears either side of it (gemm3.hip): the residual producers (proj, fc2) also write the new row as 16-bit
      // operands into xn + per-row slice sums into `stats`; qkv / fc1 read those and finish the normalisation in their epilogues.  Only block
      // 0's norm1 (rows written by the patch embedding) is a LayerNorm launch: 1 instead of 24 per forward.
      const bool fold = e->use_lnfold && g3 && gemm3_lnfold_supported(D);
      float* stats = reinterpret_cast<float*>(ws + w.stats);
      // fl: 0 plain, 1 consumer of a folded LayerNorm (X = xn as un-normalised operands), 2 producer (residual epilogue + xn + stats)
      auto lin = [&](const void* X, int K, size_t wblk, const float* bias, void* out, int N, int epi, int fl = 0, const float* cs = nullptr) {
        GemmArgs q{};
        q.X = X; q.ldx = K; q.Wblk = wb + wblk; q.bias = bias; q.out = out; q.ldo = N; q.M = M; q.N = N; q.K = K;
        q.blk_x = 1; q.blk_out = 1; q.rows_alloc = (int)w.rows; q.no_tail_split = !e->tail_split;
        if (epi == EPI_BIAS_RESID) { q.resid = xs; q.ldr = N; }
        if (fl == 1) { q.lnf = 1; q.lnf_stats = stats; q.lnf_s = cs; q.lnf_eps = 1e-6f; }
        if (fl == 2) { q.stats = stats; q.x16 = xn; }
        return gemm3_nt(prec, epi, q, s);
      };
      const bool fold1 = fold && i > 0;                    // norm1 of this block was folded by the previous block's fc2
      if (!fold1 && (rc = timed(e, "layernorm", 0.0, s, [&] { return layernorm_rows_blocked(prec, xs, M, D, F(L.ln1w), F(L.ln1b), 1e-6f, xn, s); }))) return rc;
      if ((rc = timed(e, "gemm_qkv", 2.0 * Md * 3.0 * Dd * Dd, s, [&] {
            return fold1 ? lin(xn, D, L.qkvw_bf, F(L.qkv_c), qkv, 3 * D, EPI_BIAS, 1, F(L.qkv_s)) : lin(xn, D, L.qkvw_b, F(L.qkvb), qkv, 3 * D, EPI_BIAS); }))) return rc;
      if ((rc = timed(e, "attention", 4.0 * B * e->vit.heads * (double)T * T * 64.0, s, [&] { return attention(prec, qkv, att, B, T, e->vit.heads, 1, s); }))) return rc;
      const size_t Bp = align_up((size_t)B, 256);
      if (i + 1 == e->vit.depth && e->cls_only_last && g3 && Bp <= w.rows) {
        // last block: only the class-token rows reach the output (see the fused path above): proj, LayerNorm, fc1, fc2 on B gathered rows.
        // Compact buffers: x and the attention rows in the qkv region (free after the attention), norm2 in xn, the hidden rows in hb.
        float* xc = reinterpret_cast<float*>(qkv);
        void* ac = static_cast<char*>(qkv) + Bp * D * 4;
        if ((rc = timed(e, "gather_cls", 0.0, s, [&] { return gather_cls_rows_blocked(xs, att, B, T, D, xc, ac, s); }))) return rc;
        auto linc = [&](const void* X, int K, size_t wblk, const float* bias, void* out, int N, int epi, int fl = 0, const float* cs = nullptr) {
          GemmArgs q{};
          q.X = X; q.ldx = K; q.Wblk = wb + wblk; q.bias = bias; q.out = out; q.ldo = N; q.M = B; q.N = N; q.K = K;
          q.blk_x = 1; q.blk_out = 1; q.rows_alloc = (int)Bp; q.no_tail_split = !e->tail_split;
          if (epi == EPI_BIAS_RESID) { q.resid = xc; q.ldr = N; }
          if (fl == 1) { q.lnf = 1; q.lnf_stats = stats; q.lnf_s = cs; q.lnf_eps = 1e-6f; }
          if (fl == 2) { q.stats = stats; q.x16 = xn; }
          return gemm3_nt(prec, epi, q, s);
        };
        const double Bd = B;
        if ((rc = timed(e, "cls_proj_resid", 2.0 * Bd * Dd * Dd, s, [&] { return linc(ac, D, L.projw_b, F(L.projb), xc, D, EPI_BIAS_RESID, fold ? 2 : 0); }))) return rc;
        if (!fold && (rc = timed(e, "layernorm", 0.0, s, [&] { return layernorm_rows_blocked(prec, xc, B, D, F(L.ln2w), F(L.ln2b), 1e-6f, xn, s); }))) return rc;
        if ((rc = timed(e, "cls_fc1_gelu", 2.0 * Bd * Hd * Dd, s, [&] {
              return fold ? linc(xn, D, L.fc1w_bf, F(L.fc1_c), hb, e->vit.mlp, EPI_BIAS_GELU, 1, F(L.fc1_s)) : linc(xn, D, L.fc1w_b, F(L.fc1b), hb, e->vit.mlp, EPI_BIAS_GELU); }))) return rc;
        if ((rc = timed(e, "cls_fc2_resid", 2.0 * Bd * Dd * Hd, s, [&] { return linc(hb, e->vit.mlp, L.fc2w_b, F(L.fc2b), xc, D, EPI_BIAS_RESID); }))) return rc;
        cls_x = xc;
        continue;
      }
      if ((rc = timed(e, "gemm_proj_resid", 2.0 * Md * Dd * Dd, s, [&] { return lin(att, D, L.projw_b, F(L.projb), xs, D, EPI_BIAS_RESID, fold ? 2 : 0); }))) return rc;
      if (!fold && (rc = timed(e, "layernorm", 0.0, s, [&] { return layernorm_rows_blocked(prec, xs, M, D, F(L.ln2w), F(L.ln2b), 1e-6f, xn, s); }))) return rc;
      if ((rc = timed(e, "gemm_fc1_gelu", 2.0 * Md * Hd * Dd, s, [&] {
            return fold ? lin(xn, D, L.fc1w_bf, F(L.fc1_c), hb, e->vit.mlp, EPI_BIAS_GELU, 1, F(L.fc1_s)) : lin(xn, D, L.fc1w_b, F(L.fc1b), hb, e->vit.mlp, EPI_BIAS_GELU); }))) return rc;
      if (fold && g3 && i + 1 < e->vit.depth) {            // fc2 + residual, and the next block's norm1 folded: its rows + statistics
        GemmArgs q{};
        q.X = hb; q.ldx = e->vit.mlp; q.Wblk = wb + L.fc2w_b; q.bias = F(L.fc2b); q.out = xs; q.ldo = D; q.resid = xs; q.ldr = D;
        q.M = M; q.N = D; q.K = e->vit.mlp; q.blk_x = 1; q.blk_out = 1; q.rows_alloc = (int)w.rows; q.no_tail_split = !e->tail_split;
        q.stats = stats; q.x16 = xn;
        if ((rc = timed(e, "gemm_fc2_resid", 2.0 * Md * Dd * Hd, s, [&] { return gemm3_nt(prec, EPI_BIAS_RESID, q, s); }))) return rc;
        continue;
      }
    } else {
      if ((rc = timed(e, "layernorm", 0.0, s, [&] { return layernorm_rows(prec, xs, M, D, F(L.ln1w), F(L.ln1b), 1e-6f, xn, s); }))) return rc;
      g = GemmArgs{};
      g.X = xn; g.ldx = D; g.W = wb + L.qkvw; g.ldw = D; g.bias = F(L.qkvb); g.out = qkv; g.ldo = 3 * D;
      g.M = M; g.N = 3 * D; g.K = D;
      if ((rc = timed(e, "gemm_qkv", 2.0 * Md * 3.0 * Dd * Dd, s, [&] { return gemm_nt(prec, EPI_BIAS, g, s); }))) return rc;
      if ((rc = timed(e, "attention", 4.0 * B * e->vit.heads * (double)T * T * 64.0, s, [&] { return attention(prec, qkv, att, B, T, e->vit.heads, 0, s); }))) return rc;
      g = GemmArgs{};
      g.X = att; g.ldx = D; g.W = wb + L.projw; g.ldw = D; g.bias = F(L.projb); g.out = xs; g.ldo = D;
      g.resid = xs; g.ldr = D; g.M = M; g.N = D; g.K = D;
      if ((rc = timed(e, "gemm_proj_resid", 2.0 * Md * Dd * Dd, s, [&] { return gemm_nt(prec, EPI_BIAS_RESID, g, s); }))) return rc;
      if ((rc = timed(e, "layernorm", 0.0, s, [&] { return layernorm_rows(prec, xs, M, D, F(L.ln2w), F(L.ln2b), 1e-6f, xn, s); }))) return rc;
      g = GemmArgs{};
      g.X = xn; g.ldx = D; g.W = wb + L.fc1w; g.ldw = D; g.bias = F(L.fc1b); g.out = hb; g.ldo = e->vit.mlp;
      g.M = M; g.N = e->vit.mlp; g.K = D;
      if ((rc = timed(e, "gemm_fc1_gelu", 2.0 * Md * Hd * Dd, s, [&] { return gemm_nt(prec, EPI_BIAS_GELU, g, s); }))) return rc;
    }
    g = GemmArgs{};
    g.X = hb; g.ldx = e->vit.mlp; g.W = wb + L.fc2w; g.ldw = e->vit.mlp; g.bias = F(L.fc2b); g.out = xs; g.ldo = D;
    g.resid = xs; g.ldr = D; g.M = M; g.N = D; g.K = e->vit.mlp; g.blk_x = blk; g.blk_out = blk;
    g.Wblk = wb + L.fc2w_b; g.rows_alloc = (int)w.rows; g.no_tail_split = !e->tail_split;
    if ((rc = timed(e, "gemm_fc2_resid", 2.0 * Md * Dd * Hd, s, [&] {
          return g3 ? gemm3_nt(prec, EPI_BIAS_RESID, g, s) : g2 ? gemm2_nt(prec, EPI_BIAS_RESID, g, s) : gemm_nt(prec, EPI_BIAS_RESID, g, s); }))) return rc;
  }
  if (cls_x) return timed(e, "final_cls_norm", 0.0, s, [&] { return final_cls_norm(cls_x, B, 1, D, F(e->off_normw), F(e->off_normb), 1e-6f, l2, 1, emb, status, s); });
  return timed(e, "final_cls_norm", 0.0, s, [&] { return final_cls_norm(xs, B, T, D, F(e->off_normw), F(e->off_normb), 1e-6f, l2, blk, emb, status, s); });
}

constexpr size_t CONV_SPLIT_BYTES = (size_t)16 << 20;  // split-K scratch of conv2d_nhwc: <= 256 partial tiles of 128 x 128 fp32
struct ResWs { size_t col, a, b, c, split, total; };
ResWs resnet_ws(const effocr_encoder* e, int B) {
  const size_t oh = (size_t)e->img / 2;
  Alloc al; ResWs w;
  w.col = al.take((size_t)B * oh * oh * CONV1_KPAD * 4);
  const size_t act = (size_t)B * oh * oh * 64 * 4;     // largest activation: conv1 output
  w.a = al.take(act); w.b = al.take(act); w.c = al.take(act);
  w.split = al.take(CONV_SPLIT_BYTES);
  w.total = al.off;
  return w;
}

int resnet_forward(effocr_encoder* e, const float* x, int B, float* emb, int l2, char* ws, hipStream_t s) {
  const ResWs w = resnet_ws(e, B);
  const char* wb = e->wdev;
  float* col = reinterpret_cast<float*>(ws + w.col);
  float* bufs[3] = {reinterpret_cast<float*>(ws + w.a), reinterpret_cast<float*>(ws + w.b),
                    reinterpret_cast<float*>(ws + w.c)};
  auto WT = [&](const ConvSpec& c) { return reinterpret_cast<const float*>(wb + c.w_off); };
  auto BS = [&](const ConvSpec& c) { return reinterpret_cast<const float*>(wb + c.b_off); };
  int rc;
  int H = e->img, OH = H / 2;
  // conv1 7x7/2 (+bn1+relu): im2col rows [B*OH*OW, 160] then a 1x1 implicit GEMM
  if ((rc = im2col_conv1(x, col, B, H, H, OH, OH, s))) return rc;
  ConvArgs a{};
  a.in = col; a.w = WT(e->convs[0]); a.bias = BS(e->convs[0]); a.resid = nullptr; a.out = bufs[0];
  a.B = B * OH * OH; a.H = 1; a.W = 1; a.Cin = CONV1_KPAD; a.Cout = 64; a.KH = 1; a.KW = 1; a.stride = 1; a.pad = 0;
  a.OH = 1; a.OW = 1; a.relu = 1;
  float* split = reinterpret_cast<float*>(ws + w.split);
  a.partial = split; a.partial_bytes = CONV_SPLIT_BYTES;
  if ((rc = conv2d_nhwc(a, s))) return rc;
  H = OH; OH = (H + 2 - 3) / 2 + 1;
  if ((rc = maxpool3x3s2_nhwc(bufs[0], bufs[1], B, H, H, 64, OH, OH, s))) return rc;
  H = OH;
  int cur = 1;                       // bufs[cur] holds the block input
  size_t ci = 1;
  for (int li = 1; li <= 4; ++li) {
    for (int bi = 0; bi < 2; ++bi) {
      const ConvSpec& c1 = e->convs[ci];
      const ConvSpec& c2 = e->convs[ci + 1];
      const bool down = (bi == 0 && li > 1);
      const int t1 = (cur + 1) % 3, t2 = (cur + 2) % 3;
      const int OHb = (H + 2 - 3) / c1.stride + 1;
      ConvArgs k1{};
      k1.in = bufs[cur]; k1.w = WT(c1); k1.bias = BS(c1); k1.out = bufs[t1];
      k1.B = B; k1.H = H; k1.W = H; k1.Cin = c1.cin; k1.Cout = c1.cout; k1.KH = 3; k1.KW = 3; k1.stride = c1.stride; k1.pad = 1;
      k1.OH = OHb; k1.OW = OHb; k1.relu = 1; k1.partial = split; k1.partial_bytes = CONV_SPLIT_BYTES;
      if ((rc = conv2d_nhwc(k1, s))) return rc;
      const float* idt = bufs[cur];
      if (down) {
        const ConvSpec& cd = e->convs[ci + 2];
        ConvArgs kd{};
        kd.in = bufs[cur]; kd.w = WT(cd); kd.bias = BS(cd); kd.out = bufs[t2];
        kd.B = B; kd.H = H; kd.W = H; kd.Cin = cd.cin; kd.Cout = cd.cout; kd.KH = 1; kd.KW = 1; kd.stride = cd.stride; kd.pad = 0;
        kd.OH = OHb; kd.OW = OHb; kd.relu = 0; kd.partial = split; kd.partial_bytes = CONV_SPLIT_BYTES;
        if ((rc = conv2d_nhwc(kd, s))) return rc;
        idt = bufs[t2];
      }
      // conv2 + bn2 + identity + relu; output overwrites the block input buffer when it is not the identity
      float* outb = down ? bufs[cur] : bufs[t2];
      ConvArgs k2{};
      k2.in = bufs[t1]; k2.w = WT(c2); k2.bias = BS(c2); k2.resid = idt; k2.out = outb;
      k2.B = B; k2.H = OHb; k2.W = OHb; k2.Cin = c2.cin; k2.Cout = c2.cout; k2.KH = 3; k2.KW = 3; k2.stride = 1; k2.pad = 1;
      k2.OH = OHb; k2.OW = OHb; k2.relu = 1; k2.partial = split; k2.partial_bytes = CONV_SPLIT_BYTES;
      if ((rc = conv2d_nhwc(k2, s))) return rc;
      cur = down ? cur : t2;
      H = OHb;
      ci += down ? 3 : 2;
    }
  }
  return global_avgpool_nhwc(bufs[cur], emb, B, H * H, 512, l2, s);
}

hipStream_t S(void* s) { return static_cast<hipStream_t>(s); }

}  // namespace
}  // namespace effocr

extern "C" {

int effocr_abi_version(void) { return EFFOCR_ABI_VERSION; }
const char* effocr_last_error(void) { return effocr::g_err.c_str(); }

int effocr_encoder_create(const char* arch, int img_size, int precision, effocr_encoder_t** out) {
  if (!arch || !out) return fail(EFFOCR_EINVAL, "encoder_create: NULL argument");
  if (precision < 0 || precision > 2) return fail(EFFOCR_EINVAL, "encoder_create: unknown precision");
  std::unique_ptr<effocr_encoder> e(new effocr_encoder());
  e->arch = arch; e->img = img_size; e->prec = precision;
  const std::string a = arch;
  if (a == "vit_small_patch16_224") e->vit = {384, 12, 6, 1536};
  else if (a == "vit_base_patch16_224") e->vit = {768, 12, 12, 3072};
  else if (a == "vit_tiny_test") e->vit = {128, 2, 2, 512};
  else if (a != "resnet18") return fail(EFFOCR_EUNSUPPORTED, "encoder_create: unsupported architecture '" + a + "'");
  if (a == "resnet18") {
    if (img_size < 32 || img_size % 32) return fail(EFFOCR_EINVAL, "resnet18: img_size must be a positive multiple of 32");
    e->is_vit = false; e->D = 512;
    e->prec = PREC_FP32;           // the conv path runs exact-fp32 MFMA in every mode
    build_resnet18(e.get());
  } else {
    if (img_size < 16 || img_size % 16) return fail(EFFOCR_EINVAL, "vit: img_size must be a positive multiple of 16");
    const int T = (img_size / 16) * (img_size / 16) + 1;
    if (!(T <= 64 || (T > 192 && T <= 224)))
      return fail(EFFOCR_EUNSUPPORTED, "vit: token count must be <= 64 or in (192, 224] (img_size 224)");
    e->is_vit = true; e->D = e->vit.D;
    build_vit(e.get());
  }
  *out = e.release();
  return EFFOCR_OK;
}

void effocr_encoder_destroy(effocr_encoder_t* enc) {
  if (!enc) return;
  for (auto& ev : enc->prof_pool) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
  if (enc->prof_clk) (void)hipFree(enc->prof_clk);
  delete enc;
}
int effocr_encoder_embed_dim(const effocr_encoder_t* enc) { return enc ? enc->D : 0; }
int effocr_encoder_num_params(const effocr_encoder_t* enc) { return enc ? (int)enc->params.size() : 0; }
const char* effocr_encoder_param_name(const effocr_encoder_t* enc, int i) {
  if (!enc || i < 0 || i >= (int)enc->params.size()) return nullptr;
  return enc->params[i].name.c_str();
}
int64_t effocr_encoder_param_numel(const effocr_encoder_t* enc, int i) {
  if (!enc || i < 0 || i >= (int)enc->params.size()) return -1;
  return enc->params[i].numel;
}

int effocr_encoder_set_param(effocr_encoder_t* enc, const char* name, const float* host, int64_t numel) {
  if (!enc || !name || !host) return fail(EFFOCR_EINVAL, "set_param: NULL argument");
  auto it = enc->index.find(name);
  if (it == enc->index.end()) return fail(EFFOCR_EINVAL, std::string("set_param: unknown parameter '") + name + "'");
  Param& p = enc->params[it->second];
  if (p.numel != numel)
    return fail(EFFOCR_EINVAL, std::string("set_param: '") + name + "' expects " + std::to_string(p.numel) +
                                   " elements, got " + std::to_string(numel));
  p.data.assign(host, host + numel);
  p.set = true;
  return EFFOCR_OK;
}

size_t effocr_encoder_weights_bytes(const effocr_encoder_t* enc) { return enc ? enc->wbytes : 0; }

int effocr_encoder_upload(effocr_encoder_t* enc, void* weights_dev, size_t bytes) {
  if (!enc || !weights_dev) return fail(EFFOCR_EINVAL, "upload: NULL argument");
  if (bytes < enc->wbytes) return fail(EFFOCR_EWORKSPACE, "upload: weight buffer too small");
  for (const Param& p : enc->params)
    if (!p.set) return fail(EFFOCR_ESTATE, "upload: parameter '" + p.name + "' was never set");
  std::vector<char> blob(enc->wbytes, 0);
  if (enc->is_vit) pack_vit(enc, blob); else pack_resnet(enc, blob);
  const hipError_t er = hipMemcpy(weights_dev, blob.data(), enc->wbytes, hipMemcpyHostToDevice);
  if (er != hipSuccess) return fail(EFFOCR_EHIP, std::string("upload: hipMemcpy: ") + hipGetErrorString(er));
  enc->wdev = static_cast<const char*>(weights_dev);
  return EFFOCR_OK;
}

size_t effocr_encoder_workspace_bytes(const effocr_encoder_t* enc, int batch) {
  if (!enc || batch <= 0) return 0;
  if (!enc->is_vit) return resnet_ws(enc, batch).total;
  return vit_ws(enc, (enc->chunk > 0 && enc->chunk < batch) ? enc->chunk : batch).total;
}

int effocr_encoder_set_option(effocr_encoder_t* enc, const char* name, int value) {
  if (!enc || !name) return fail(EFFOCR_EINVAL, "set_option: NULL argument");
  const std::string n = name;
  if (n == "use_panel") { enc->use_panel = value; return EFFOCR_OK; }
  if (n == "debug") { enc->debug = value; return EFFOCR_OK; }
  if (n == "use_gemm2") { enc->use_gemm2 = value; return EFFOCR_OK; }
  if (n == "use_blocked") { enc->use_blocked = value; return EFFOCR_OK; }
  if (n == "tail_split") { enc->tail_split = value; return EFFOCR_OK; }
  if (n == "split6") { enc->split6 = value; return EFFOCR_OK; }
  if (n == "mlp_pair") { enc->mlp_pair = value; return EFFOCR_OK; }
  if (n == "pair_parts") { enc->pair_parts = value; return EFFOCR_OK; }
  if (n == "use_gemm3") { enc->use_gemm3 = value; return EFFOCR_OK; }
  if (n == "use_lnfold") { enc->use_lnfold = value; return EFFOCR_OK; }
  if (n == "use_mlp") { enc->use_mlp = value; return EFFOCR_OK; }
  if (n == "use_qkvattn") { enc->use_qkvattn = value; return EFFOCR_OK; }
  if (n == "qa_min_batch") { enc->qa_min_batch = value; return EFFOCR_OK; }
  if (n == "qa_hsplit") { enc->qa_hsplit = value; return EFFOCR_OK; }
  if (n == "use_patchf") { enc->use_patchf = value; return EFFOCR_OK; }
  if (n == "use_projf") { enc->use_projf = value; return EFFOCR_OK; }
  if (n == "cls_only_last") { enc->cls_only_last = value; return EFFOCR_OK; }
  if (n == "mlp_stagger_min_rounds") { enc->mlp_stagger_min_rounds = value < 1 ? 1 : value; return EFFOCR_OK; }
  if (n == "mlp_stagger") { enc->mlp_stagger = value < 0 ? 0 : value; return EFFOCR_OK; }
  if (n == "panel_rows") { if (value != 64 && value != 128) return fail(EFFOCR_EINVAL, "set_option: panel_rows must be 64 or 128"); enc->panel_rows = value; return EFFOCR_OK; }
  if (n == "chunk") { if (value < 0) return fail(EFFOCR_EINVAL, "set_option: chunk < 0"); enc->chunk = value; return EFFOCR_OK; }
  return fail(EFFOCR_EINVAL, "set_option: unknown option '" + n + "'");
}

int effocr_encoder_set_chunk(effocr_encoder_t* enc, int crops_per_chunk) {
  if (!enc || crops_per_chunk < 0) return fail(EFFOCR_EINVAL, "set_chunk: bad argument");
  enc->chunk = crops_per_chunk;
  return EFFOCR_OK;
}

int effocr_encoder_forward(effocr_encoder_t* enc, const float* x_dev, int batch, float* emb_dev, int l2_normalize,
                           void* workspace_dev, size_t workspace_bytes, void* stream) {
  return effocr_encoder_forward_ex(enc, x_dev, EFFOCR_PREC_FP32, batch, emb_dev, l2_normalize, workspace_dev, workspace_bytes, stream);
}

int effocr_encoder_forward_ex(effocr_encoder_t* enc, const void* x_dev, int x_dtype, int batch, float* emb_dev, int l2_normalize,
                              void* workspace_dev, size_t workspace_bytes, void* stream) {
  if (!enc) return fail(EFFOCR_EINVAL, "forward: NULL encoder");
  if (x_dtype != PREC_FP32 && !(enc->is_vit && enc->prec != PREC_FP32 && x_dtype == enc->prec))
    return fail(x_dtype < 0 || x_dtype > 2 ? EFFOCR_EINVAL : EFFOCR_EUNSUPPORTED,
                "forward: crops must be fp32, or (ViT, 16-bit precision modes) already in the encoder's own operand type");
  const int x16 = x_dtype != PREC_FP32;
  if (batch < 0) return fail(EFFOCR_EINVAL, "forward: negative batch");
  if (batch == 0) return EFFOCR_OK;
  if (!x_dev || !emb_dev || !workspace_dev) return fail(EFFOCR_EINVAL, "forward: NULL device pointer");
  if (!enc->wdev) return fail(EFFOCR_ESTATE, "forward: weights were not uploaded");
  if (workspace_bytes < effocr_encoder_workspace_bytes(enc, batch)) return fail(EFFOCR_EWORKSPACE, "forward: workspace too small");
  if ((int64_t)batch * (enc->is_vit ? enc->T : enc->img * enc->img) >= (int64_t)1 << 30)
    return fail(EFFOCR_EUNSUPPORTED, "forward: batch too large for 32-bit row indices");
  char* ws = static_cast<char*>(workspace_dev);
  if (!enc->is_vit) return resnet_forward(enc, static_cast<const float*>(x_dev), batch, emb_dev, l2_normalize, ws, S(stream));
  // sub-batches: all activations of `chunk` crops (~1.6 MB per ViT-S crop) stay resident in the
  // 256 MiB Infinity Cache between consecutive kernels instead of round-tripping through HBM
  const int chunk = enc->chunk > 0 ? enc->chunk : batch;
  const size_t img_bytes = (size_t)3 * enc->img * enc->img * (x16 ? 2 : 4);
  for (int b0 = 0; b0 < batch; b0 += chunk) {
    const int cb = (batch - b0 < chunk) ? batch - b0 : chunk;
    const int rc = vit_forward(enc, static_cast<const char*>(x_dev) + (size_t)b0 * img_bytes, x16, cb, emb_dev + (size_t)b0 * enc->D, l2_normalize, ws, S(stream));
    if (rc) return rc;
  }
  return EFFOCR_OK;
}

int effocr_clock_sample(void* out_dev, void* stream) {
  if (!out_dev) return fail(EFFOCR_EINVAL, "clock_sample: NULL pointer");
  return clock_sample(static_cast<unsigned long long*>(out_dev), S(stream));
}

int effocr_encoder_check_status(const effocr_encoder_t* enc, const void* workspace_dev, void* stream) {
  if (!enc || !workspace_dev) return fail(EFFOCR_EINVAL, "check_status: NULL argument");
  if (!enc->is_vit) return EFFOCR_OK;                     // the CNN path computes in fp32 throughout
  int st = 0;
  // on the caller's stream (not the null stream, which would synchronise with every blocking stream of the process)
  hipError_t er = hipMemcpyAsync(&st, workspace_dev, sizeof(int), hipMemcpyDeviceToHost, S(stream));   // VitWs::status = offset 0
  if (er == hipSuccess) er = hipStreamSynchronize(S(stream));
  if (er == hipSuccess && st != 0) er = hipMemsetAsync(const_cast<void*>(workspace_dev), 0, sizeof(int), S(stream));   // read-and-clear
  if (er != hipSuccess) return fail(EFFOCR_EHIP, std::string("check_status: ") + hipGetErrorString(er));
  if (st != 0)
    return fail(EFFOCR_EOVERFLOW, enc->prec == PREC_FP16
                    ? "forward: non-finite embedding — an f16 operand overflowed (|q|, |k|, |v| or an fc1 pre-activation beyond 65504) or the input was not finite; use precision bf16 or fp32 for this checkpoint"
                    : "forward: non-finite embedding — the input crops or the weights hold inf / nan");
  return EFFOCR_OK;
}

int effocr_encoder_reset_status(const effocr_encoder_t* enc, void* workspace_dev, void* stream) {
  if (!enc || !workspace_dev) return fail(EFFOCR_EINVAL, "reset_status: NULL argument");
  const hipError_t er = hipMemsetAsync(workspace_dev, 0, 256, S(stream));
  if (er != hipSuccess) return fail(EFFOCR_EHIP, std::string("reset_status: ") + hipGetErrorString(er));
  return EFFOCR_OK;
}

int effocr_encoder_profile_begin(effocr_encoder_t* enc, int mode, const char* only_class) {
  if (!enc || mode < 0 || mode > 2) return fail(EFFOCR_EINVAL, "profile_begin: bad argument");
  if (mode == 2 && !only_class) return fail(EFFOCR_EINVAL, "profile_begin: mode 2 needs a class name");
  enc->prof_mode = mode;
  enc->prof_only = only_class ? only_class : "";
  enc->prof_rec.clear(); enc->prof_used = 0;
  enc->prof_names.clear(); enc->prof_work.clear(); enc->prof_ms.clear(); enc->prof_cnt.clear();
  return EFFOCR_OK;
}

int effocr_encoder_profile_collect(effocr_encoder_t* enc) {
  if (!enc) return fail(EFFOCR_EINVAL, "profile_collect: NULL encoder");
  const bool clk = enc->prof_mode == 1 && enc->prof_clk;
  enc->prof_mode = 0;
  enc->prof_ms.assign(enc->prof_names.size(), 0.f);
  enc->prof_cnt.assign(enc->prof_names.size(), 0);
  enc->prof_ghz.assign(enc->prof_names.size(), 0.0);
  if (clk && !enc->prof_rec.empty()) {            // shader clock of a class = sum of shader ticks / sum of 100 MHz ticks over its launches
    if (hipEventSynchronize(enc->prof_pool[enc->prof_rec.back().second].second) != hipSuccess) return fail(EFFOCR_EHIP, "profile_collect: hipEventSynchronize failed");
    if (hipDeviceSynchronize() != hipSuccess) return fail(EFFOCR_EHIP, "profile_collect: hipDeviceSynchronize failed");
    const size_t n = enc->prof_used < effocr_encoder::PROF_CLK_SLOTS ? enc->prof_used : effocr_encoder::PROF_CLK_SLOTS;
    std::vector<unsigned long long> h(n * 8192);
    if (hipMemcpy(h.data(), enc->prof_clk, n * 8192 * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) return fail(EFFOCR_EHIP, "profile_collect: clock samples");
    std::vector<double> st(enc->prof_names.size(), 0.0), rt(enc->prof_names.size(), 0.0);
    for (const auto& r : enc->prof_rec) {
      if ((size_t)r.second >= n) continue;
      const unsigned long long* q = h.data() + (size_t)r.second * 8192;
      unsigned long long dmin = ~0ull;              // a CU one of the two samples did not reach keeps an older pair: longer interval, skipped
      for (int x = 0; x < 2048; ++x) {
        const unsigned long long t0 = q[2 * x + 1], t1 = q[4096 + 2 * x + 1];
        if (t0 && t1 > t0 && t1 - t0 < dmin) dmin = t1 - t0;
      }
      for (int x = 0; x < 2048; ++x) {
        const unsigned long long t0 = q[2 * x + 1], t1 = q[4096 + 2 * x + 1];
        if (!t0 || t1 <= t0 || t1 - t0 > dmin + dmin / 4 + 1000) continue;
        st[r.first] += (double)(q[4096 + 2 * x] - q[2 * x]); rt[r.first] += (double)(t1 - t0);
      }
    }
    for (size_t i = 0; i < st.size(); ++i) enc->prof_ghz[i] = rt[i] > 0 ? st[i] / rt[i] * 0.1 : 0.0;
  }
  for (const auto& r : enc->prof_rec) {
    const auto& ev = enc->prof_pool[r.second];
    if (hipEventSynchronize(ev.second) != hipSuccess) return fail(EFFOCR_EHIP, "profile_collect: hipEventSynchronize failed");
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, ev.first, ev.second) != hipSuccess) return fail(EFFOCR_EHIP, "profile_collect: hipEventElapsedTime failed");
    enc->prof_ms[r.first] += ms;
    enc->prof_cnt[r.first] += 1;
  }
  enc->prof_rec.clear(); enc->prof_used = 0;
  return (int)enc->prof_names.size();
}

int effocr_encoder_profile_get(const effocr_encoder_t* enc, int i, const char** name, double* total_ms, int* launches,
                               double* total_work) {
  if (!enc || i < 0 || i >= (int)enc->prof_ms.size()) return fail(EFFOCR_EINVAL, "profile_get: bad index");
  if (name) *name = enc->prof_names[i].c_str();
  if (total_ms) *total_ms = enc->prof_ms[i];
  if (launches) *launches = enc->prof_cnt[i];
  if (total_work) *total_work = enc->prof_work[i];
  return EFFOCR_OK;
}

int effocr_encoder_profile_clock(const effocr_encoder_t* enc, int i, double* shader_ghz) {
  if (!enc || i < 0 || i >= (int)enc->prof_ghz.size() || !shader_ghz) return fail(EFFOCR_EINVAL, "profile_clock: bad argument");
  *shader_ghz = enc->prof_ghz[i];
  return EFFOCR_OK;
}

size_t effocr_knn_workspace_bytes(int64_t nq, int64_t ntotal, int d, int k) { return knn_workspace_bytes(nq, ntotal, d, k); }

int effocr_knn_ip_topk(const float* q_dev, int64_t nq, const float* xb_dev, int64_t ntotal, int d, int k,
                       float* dist_dev, int64_t* idx_dev, void* workspace_dev, size_t workspace_bytes, void* stream) {
  if (nq > 0 && (!q_dev || !dist_dev || !idx_dev)) return fail(EFFOCR_EINVAL, "knn: NULL device pointer");
  if (nq > 0 && ntotal > 0 && !xb_dev) return fail(EFFOCR_EINVAL, "knn: NULL index pointer");
  return knn_ip_topk(q_dev, nq, xb_dev, ntotal, d, k, dist_dev, idx_dev, workspace_dev, workspace_bytes, S(stream));
}

int effocr_l2_normalize(const float* x_dev, int64_t n, int d, float* y_dev, void* stream) {
  if (n > 0 && (!x_dev || !y_dev)) return fail(EFFOCR_EINVAL, "l2_normalize: NULL device pointer");
  return l2_normalize_rows(x_dev, n, d, y_dev, S(stream));
}

size_t effocr_knn_screen_workspace_bytes(int64_t nq, int64_t ntotal, int d, int k) { return knn_screen_workspace_bytes(nq, ntotal, d, k); }

int effocr_knn_set_option(const char* name, int value) {
  if (!name) return fail(EFFOCR_EINVAL, "knn_set_option: NULL name");
  if (std::string(name) == "force_tile") { knn_force_tile_kernel(value); return EFFOCR_OK; }
  if (std::string(name) == "wg_target") { knn_set_wg_target(value); return EFFOCR_OK; }
  if (std::string(name) == "two_pass_screen") { knn_two_pass_screen(value); return EFFOCR_OK; }
  if (std::string(name) == "q16_tile") { knn_q16_tile(value); return EFFOCR_OK; }
  if (std::string(name) == "qs") { knn_qs_option(0, value); return EFFOCR_OK; }
  if (std::string(name) == "qs_wgs") { knn_qs_option(1, value); return EFFOCR_OK; }
  if (std::string(name) == "stream_min_rows") { knn_qs_option(3, value); return EFFOCR_OK; }
  if (std::string(name) == "qs_qt") { knn_qs_option(4, value); return EFFOCR_OK; }
  if (std::string(name) == "qs_fine") { knn_qs_option(5, value); return EFFOCR_OK; }
  return fail(EFFOCR_EINVAL, std::string("knn_set_option: unknown option '") + name + "'");
}

size_t effocr_knn_screen_flag_offset(int64_t nq, int64_t ntotal, int d, int k) { return knn_screen_flag_offset(nq, ntotal, d, k); }

int effocr_knn_ip_topk_screened(const float* q_dev, int64_t nq, const float* xb_dev, const void* xb_bf16_dev, int64_t ntotal, int d, int k,
                                float xnorm_max, float* dist_dev, int64_t* idx_dev, void* workspace_dev, size_t workspace_bytes, void* stream) {
  if (nq > 0 && (!q_dev || !xb_dev || !xb_bf16_dev || !dist_dev || !idx_dev || !workspace_dev)) return fail(EFFOCR_EINVAL, "knn(screened): NULL device pointer");
  return knn_ip_topk_screened(q_dev, nq, xb_dev, xb_bf16_dev, nullptr, ntotal, d, k, xnorm_max, dist_dev, idx_dev, workspace_dev, workspace_bytes, S(stream));
}

int effocr_knn_ip_topk_screened2(const float* q_dev, int64_t nq, const float* xb_dev, const void* xb_bf16_dev, const void* xb_bf16_blk_dev,
                                 int64_t ntotal, int d, int k, float xnorm_max, float* dist_dev, int64_t* idx_dev, void* workspace_dev,
                                 size_t workspace_bytes, void* stream) {
  if (nq > 0 && (!q_dev || !xb_dev || (!xb_bf16_dev && !xb_bf16_blk_dev) || !dist_dev || !idx_dev || !workspace_dev))
    return fail(EFFOCR_EINVAL, "knn(screened): NULL device pointer");
  return knn_ip_topk_screened(q_dev, nq, xb_dev, xb_bf16_dev, xb_bf16_blk_dev, ntotal, d, k, xnorm_max, dist_dev, idx_dev, workspace_dev, workspace_bytes, S(stream));
}

size_t effocr_bf16_blocked_bytes(int64_t n_rows, int d) { return n_rows <= 0 || d <= 0 ? 0 : (size_t)((n_rows + 63) / 64 * 64) * (size_t)d * 2; }

int effocr_convert_bf16_blocked(const float* src_dev, int64_t n_rows, int d, void* dst_dev, void* stream) {
  if (n_rows > 0 && (!src_dev || !dst_dev)) return fail(EFFOCR_EINVAL, "convert_bf16_blocked: NULL device pointer");
  if (n_rows < 0 || d <= 0) return fail(EFFOCR_EINVAL, "convert_bf16_blocked: bad sizes");
  return convert_bf16_blocked(src_dev, n_rows, d, dst_dev, S(stream));
}

int effocr_convert_bf16(const float* src_dev, int64_t n, void* dst_dev, void* stream) {
  if (n > 0 && (!src_dev || !dst_dev)) return fail(EFFOCR_EINVAL, "convert_bf16: NULL device pointer");
  if (n < 0) return fail(EFFOCR_EINVAL, "convert_bf16: n < 0");
  return convert_bf16(src_dev, n, dst_dev, S(stream));
}

int effocr_crop_transform(const uint8_t* image_dev, int height, int width, int64_t row_stride, const int32_t* boxes_dev,
                          int n, int size, int antialias, const float* mean, const float* stdv, const float* fill,
                          float* out_dev, void* stream) {
  if (n < 0 || height <= 0 || width <= 0 || row_stride < (int64_t)3 * width) return fail(EFFOCR_EINVAL, "crop_transform: bad image geometry");
  if (n > 0 && (!image_dev || !boxes_dev || !out_dev || !mean || !stdv || !fill)) return fail(EFFOCR_EINVAL, "crop_transform: NULL pointer");
  for (int c = 0; c < 3 && n > 0; ++c)
    if (!(stdv[c] != 0.f)) return fail(EFFOCR_EINVAL, "crop_transform: std must be non-zero");
  return crop_transform(image_dev, 1, 0, height, width, row_stride, boxes_dev, 4, n, size, antialias, mean, stdv, fill, out_dev, PREC_FP32, S(stream));
}

int effocr_crop_transform_batch(const uint8_t* images_dev, int n_images, int64_t image_stride, int height, int width, int64_t row_stride,
                                const int32_t* boxes_dev, int64_t n, int size, int antialias, const float* mean, const float* stdv,
                                const float* fill, float* out_dev, void* stream) {
  return effocr_crop_transform_batch_ex(images_dev, n_images, image_stride, height, width, row_stride, boxes_dev, n, size, antialias, mean, stdv,
                                        fill, EFFOCR_PREC_FP32, out_dev, stream);
}

int effocr_crop_transform_batch_ex(const uint8_t* images_dev, int n_images, int64_t image_stride, int height, int width, int64_t row_stride,
                                   const int32_t* boxes_dev, int64_t n, int size, int antialias, const float* mean, const float* stdv,
                                   const float* fill, int out_dtype, void* out_dev, void* stream) {
  if (out_dtype < 0 || out_dtype > 2) return fail(EFFOCR_EINVAL, "crop_transform_batch: unknown output type");
  if (n < 0 || n_images <= 0 || height <= 0 || width <= 0 || row_stride < (int64_t)3 * width || image_stride < row_stride * height)
    return fail(EFFOCR_EINVAL, "crop_transform_batch: bad image geometry");
  if (n > 0 && (!images_dev || !boxes_dev || !out_dev || !mean || !stdv || !fill)) return fail(EFFOCR_EINVAL, "crop_transform_batch: NULL pointer");
  for (int c = 0; c < 3 && n > 0; ++c)
    if (!(stdv[c] != 0.f)) return fail(EFFOCR_EINVAL, "crop_transform_batch: std must be non-zero");
  return crop_transform(images_dev, n_images, image_stride, height, width, row_stride, boxes_dev, 5, n, size, antialias, mean, stdv, fill, out_dev, out_dtype, S(stream));
}

int effocr_gather_rows(const float* src_dev, const int64_t* keep_rows_dev, int64_t n_keep, int d, float* dst_dev, void* stream) {
  if (n_keep > 0 && (!src_dev || !keep_rows_dev || !dst_dev)) return fail(EFFOCR_EINVAL, "gather_rows: NULL device pointer");
  return gather_rows(src_dev, keep_rows_dev, n_keep, d, dst_dev, S(stream));
}

int effocr_op_linear(int precision, int epilogue, const void* x_dev, const void* w_dev, const float* bias_dev,
                     const float* resid_dev, void* out_dev, int m, int n, int k, void* stream) {
  if (epilogue < 0 || epilogue > 2) return fail(EFFOCR_EINVAL, "op_linear: unknown epilogue");
  if (panel_gemm_supported(precision, n, k)) {          // same dispatch rule as the encoder forward
    PanelArgs p{};
    p.A = x_dev; p.lda = k; p.W = w_dev; p.bias = bias_dev; p.out = out_dev; p.ldo = n; p.resid = resid_dev; p.ldr = n;
    p.M = m; p.N = n; p.K = k;
    return panel_gemm(precision, PRO_COPY, epilogue, p, S(stream));
  }
  GemmArgs g{};
  g.X = x_dev; g.ldx = k; g.W = w_dev; g.ldw = k; g.bias = bias_dev; g.out = out_dev; g.ldo = n;
  g.resid = resid_dev; g.ldr = n; g.M = m; g.N = n; g.K = k;
  if (gemm2_supported(precision, n, k)) return gemm2_nt(precision, epilogue, g, S(stream));   // same rule as the forward
  return gemm_nt(precision, epilogue, g, S(stream));
}

int effocr_op_ln_linear(int precision, int epilogue, const float* x_dev, const float* gamma_dev, const float* beta_dev,
                        float eps, const void* w_dev, const float* bias_dev, const float* resid_dev, void* out_dev,
                        int m, int n, int k, void* stream) {
  if (epilogue < 0 || epilogue > 2) return fail(EFFOCR_EINVAL, "op_ln_linear: unknown epilogue");
  PanelArgs p{};
  p.A = x_dev; p.lda = k; p.gamma = gamma_dev; p.beta = beta_dev; p.eps = eps; p.W = w_dev; p.bias = bias_dev;
  p.out = out_dev; p.ldo = n; p.resid = resid_dev; p.ldr = n; p.M = m; p.N = n; p.K = k;
  return panel_gemm(precision, PRO_LN, epilogue, p, S(stream));
}


int effocr_op_linear_blocked(int precision, int epilogue, const void* x_blk_dev, const void* w_blk_dev, const float* bias_dev,
                             const float* resid_blk_dev, void* out_blk_dev, int m, int n, int k, int rows_alloc, void* stream) {
  if (epilogue < 0 || epilogue > 2) return fail(EFFOCR_EINVAL, "op_linear_blocked: unknown epilogue");
  if (m > 0 && (!x_blk_dev || !w_blk_dev || !bias_dev || !out_blk_dev || (epilogue == EPI_BIAS_RESID && !resid_blk_dev)))
    return fail(EFFOCR_EINVAL, "op_linear_blocked: NULL device pointer");
  GemmArgs g{};
  g.X = x_blk_dev; g.ldx = k; g.Wblk = w_blk_dev; g.bias = bias_dev; g.out = out_blk_dev; g.ldo = n;
  g.resid = resid_blk_dev; g.ldr = n; g.M = m; g.N = n; g.K = k; g.blk_x = 1; g.blk_out = 1; g.rows_alloc = rows_alloc;
  return gemm3_nt(precision, epilogue, g, S(stream));
}

int effocr_op_mlp_blocked(int precision, float* x_blk_dev, const float* gamma_dev, const float* beta_dev, float eps,
                          const void* w1_blk_dev, const float* b1_dev, const void* w2_perm_dev, const float* b2_perm_dev, const float* b2_dev,
                          int m, int d, int h, int rows_alloc, void* scratch_dev, size_t scratch_bytes, void* stream) {
  if (m > 0 && (!x_blk_dev || !gamma_dev || !beta_dev || !w1_blk_dev || !b1_dev || !w2_perm_dev || !b2_perm_dev || !b2_dev))
    return fail(EFFOCR_EINVAL, "op_mlp_blocked: NULL device pointer");
  MlpArgs a{};
  a.x = x_blk_dev; a.gamma = gamma_dev; a.beta = beta_dev; a.eps = eps; a.W1b = w1_blk_dev; a.b1 = b1_dev; a.W2p = w2_perm_dev; a.b2 = b2_perm_dev;
  a.b2_logical = b2_dev;
  a.M = m; a.D = d; a.H = h; a.rows_alloc = rows_alloc; a.partial = static_cast<float*>(scratch_dev); a.partial_bytes = scratch_bytes;
  return mlp_fused(precision, a, S(stream));
}

int effocr_op_mlp_ln_blocked(int precision, float* x_blk_dev, const float* gamma_dev, const float* beta_dev, float eps,
                             const void* w1_blk_dev, const float* b1_dev, const void* w2_perm_dev, const float* b2_perm_dev, const float* b2_dev,
                             const float* gamma_next_dev, const float* beta_next_dev, void* xn_blk_dev,
                             int m, int d, int h, int rows_alloc, void* scratch_dev, size_t scratch_bytes, void* stream) {
  if (m > 0 && (!x_blk_dev || !gamma_dev || !beta_dev || !w1_blk_dev || !b1_dev || !w2_perm_dev || !b2_perm_dev || !b2_dev || !gamma_next_dev || !beta_next_dev || !xn_blk_dev))
    return fail(EFFOCR_EINVAL, "op_mlp_ln_blocked: NULL device pointer");
  MlpArgs a{};
  a.x = x_blk_dev; a.gamma = gamma_dev; a.beta = beta_dev; a.eps = eps; a.W1b = w1_blk_dev; a.b1 = b1_dev; a.W2p = w2_perm_dev; a.b2 = b2_perm_dev;
  a.b2_logical = b2_dev;
  a.M = m; a.D = d; a.H = h; a.rows_alloc = rows_alloc; a.partial = static_cast<float*>(scratch_dev); a.partial_bytes = scratch_bytes;
  a.xn_out = xn_blk_dev; a.gamma_n = gamma_next_dev; a.beta_n = beta_next_dev;
  return mlp_fused(precision, a, S(stream));
}

int effocr_op_proj_mlp_blocked(int precision, float* x_blk_dev, const void* a_blk_dev, const void* wp_perm_dev, const float* bp_perm_dev,
                               const float* gamma_dev, const float* beta_dev, float eps, const void* w1_blk_dev, const float* b1_dev,
                               const void* w2_perm_dev, const float* b2_perm_dev, const float* b2_dev, int m, int d, int h, int rows_alloc,
                               void* scratch_dev, size_t scratch_bytes, void* stream) {
  if (m > 0 && (!x_blk_dev || !a_blk_dev || !wp_perm_dev || !bp_perm_dev || !gamma_dev || !beta_dev || !w1_blk_dev || !b1_dev || !w2_perm_dev || !b2_perm_dev || !b2_dev))
    return fail(EFFOCR_EINVAL, "op_proj_mlp_blocked: NULL device pointer");
  MlpArgs a{};
  a.x = x_blk_dev; a.gamma = gamma_dev; a.beta = beta_dev; a.eps = eps; a.W1b = w1_blk_dev; a.b1 = b1_dev; a.W2p = w2_perm_dev; a.b2 = b2_perm_dev;
  a.b2_logical = b2_dev; a.A = a_blk_dev; a.Wpp = wp_perm_dev; a.bp = bp_perm_dev;
  a.M = m; a.D = d; a.H = h; a.rows_alloc = rows_alloc; a.partial = static_cast<float*>(scratch_dev); a.partial_bytes = scratch_bytes;
  return mlp_fused(precision, a, S(stream));
}

int effocr_op_qkv_attn_blocked(int precision, const void* xn_blk_dev, const void* wqkv_blk_dev, const float* bias_dev,
                               void* out_blk_dev, int batch, int tokens, int d, int rows_alloc, void* stream) {
  if (batch > 0 && (!xn_blk_dev || !wqkv_blk_dev || !bias_dev || !out_blk_dev))
    return fail(EFFOCR_EINVAL, "op_qkv_attn_blocked: NULL device pointer");
  if (batch < 0 || tokens < 1) return fail(EFFOCR_EINVAL, "op_qkv_attn_blocked: bad batch / tokens");
  QkvAttnArgs q{};
  q.xn = xn_blk_dev; q.Wb = wqkv_blk_dev; q.bias = bias_dev; q.out = out_blk_dev;
  q.B = batch; q.T = tokens; q.D = d; q.rows_alloc = rows_alloc;
  return qkv_attn_fused(precision, q, S(stream));
}

int effocr_op_layernorm_blocked(int out_precision, const float* x_blk_dev, int64_t rows, int d, const float* gamma_dev,
                                const float* beta_dev, float eps, void* out_blk_dev, void* stream) {
  if (rows > 0 && (!x_blk_dev || !gamma_dev || !beta_dev || !out_blk_dev)) return fail(EFFOCR_EINVAL, "op_layernorm_blocked: NULL device pointer");
  return layernorm_rows_blocked(out_precision, x_blk_dev, rows, d, gamma_dev, beta_dev, eps, out_blk_dev, S(stream));
}

int effocr_op_layernorm(int out_precision, const float* x_dev, int64_t rows, int d, const float* gamma_dev,
                        const float* beta_dev, float eps, void* out_dev, void* stream) {
  return layernorm_rows(out_precision, x_dev, rows, d, gamma_dev, beta_dev, eps, out_dev, S(stream));
}

int effocr_op_attention(int precision, const void* qkv_dev, void* out_dev, int batch, int tokens, int heads, void* stream) {
  return attention(precision, qkv_dev, out_dev, batch, tokens, heads, 0, S(stream));
}

}  // extern "C"
