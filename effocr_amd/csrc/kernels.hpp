// Internal launcher interface between the C-ABI layer (api.hip) and the kernel files.
#pragma once
#include "common.hpp"

namespace effocr {

enum { PREC_BF16 = 0, PREC_FP16 = 1, PREC_FP32 = 2 };
enum { EPI_BIAS = 0, EPI_BIAS_GELU = 1, EPI_BIAS_RESID = 2, EPI_PATCH = 3 };

static inline int prec_esize(int prec) { return prec == PREC_FP32 ? 4 : 2; }

struct GemmArgs {
  const void* X; int64_t ldx;       // activations [M,K] (elements of the operand type)
  const void* W; int64_t ldw;       // weights     [N,K]
  const float* bias;                // [N]
  void* out; int64_t ldo;           // [M,N] (operand type, or fp32 for RESID / PATCH)
  const float* resid; int64_t ldr;  // EPI_BIAS_RESID: fp32 residual (may alias out)
  const float* pos;                 // EPI_PATCH: pos_embed rows [1+P, N] fp32
  int M, N, K;
  int P;                            // EPI_PATCH: patches per image
  int blk_x;                        // gemm2: X operand is fragment-blocked (16-B chunks of 8 elements)
  int blk_out;                      // gemm2: out / resid (fp32, chunks of 4) are fragment-blocked
  const void* Wblk;                 // gemm3: W in the fragment-blocked layout [N/32][K/8][32][16 B]
  int rows_alloc;                   // gemm3: rows addressable in X / out / resid (multiple of 32; 0 = round M up)
  int no_tail_split;                // gemm3: 1 = one launch of 256-token tiles only (A/B switch)
  // gemm3, LayerNorm folded into the linears either side of it (ViT-B path, api.hip): the producer of a residual row (EPI_BIAS_RESID)
  // also writes the row as 16-bit operands (x16) and per-row partial (sum, sum of squares) of its 128-feature slices (stats: [row][8][2]
  // fp32, slice = 2 * column tile + wave half); the consumer (lnf = 1: X = that x16, Wblk = W . diag(gamma), bias = b + W beta,
  // lnf_s[n] = sum_k Wblk[n][k]) finishes y = rstd (acc - mean s) + bias in its epilogue — no LayerNorm launch, no fp32 re-read of x.
  float* stats; void* x16;          // producer outputs (NULL: none)
  const float* lnf_stats; const float* lnf_s; int lnf; float lnf_eps;   // consumer inputs
};

// gemm.hip
int gemm_nt(int prec, int epi, const GemmArgs& g, hipStream_t s);

// gemm2.hip — K-streaming GEMM with a glds ring for both operands (long K: mlp.fc2, patch embedding)
bool gemm2_supported(int prec, int N, int K);
int gemm2_nt(int prec, int epi, const GemmArgs& g, hipStream_t s);

// gemm3.hip — 256 x {192,256} tiles, 128-row wave tiles, all operands fragment-blocked (fc2; every ViT-B linear)
bool gemm3_supported(int prec, int N, int K);
bool gemm3_lnfold_supported(int D);                    // D-wide rows: LayerNorm can be folded between a gemm3 producer and consumer
int gemm3_nt(int prec, int epi, const GemmArgs& g, hipStream_t s);

// mlp.hip — fused LayerNorm + fc1 + GELU + fc2 + residual over the blocked residual stream (hidden stays on chip)
struct MlpArgs {
  float* x;                         // fp32 residual stream, fragment-blocked, updated in place
  const float* gamma; const float* beta; float eps;   // norm2
  const void* W1b; const float* b1; // fc1 weight [H,D] fragment-blocked, bias [H]
  const void* W2p; const float* b2; // fc2 weight [D,H] fragment-blocked, k index permuted per 16 AND rows permuted per 32 (put_op_blocked perm16 +
                                    // rowperm), bias [D] permuted like the rows (put_f32_rowperm)
  int M, D, H;
  int rows_alloc;                   // rows addressable in x (multiple of 32, >= M)
  float* partial; size_t partial_bytes;   // optional scratch for the tail split (>= 4 * 64 * 128 * D * 4 bytes covers every case)
  int no_tail_split;                // 1: single launch (A/B switch)
  int no_split6;                    // 1: never the 6-way split of calls of <= 27 crops (A/B switch)
  int no_pair_parts;                // 1: never the 3-way "pair parts" of calls of <= 27 crops (A/B switch: the 6- / 4-way 128-token parts instead)
  int pair;                         // 64-token panels on wave pairs (mlp_kernel.hpp PAIR; projection form only): 0 = where the launcher finds them faster, 1 = whenever they fit one round of CUs, -1 = never
  int panel0, tail_rb, stagger_wgs, main_wgs; // set by the launcher
  int stagger;                      // > 0: the first round of workgroups starts spread over 32 x stagger clock ticks (see mlp_kernel.hpp)
  int stagger_min_rounds;           // ... when the launch has at least this many rounds of CUs (0 = 4)
  // optional leading projection + residual (attn.proj): x <- x + A . Wp^T + bp, fused in front of the MLP.  Then Wpp is
  // Wp fragment-blocked with the rows of every 32-row block permuted, bp permuted alike.  b2_logical is the unpermuted
  // fc2 bias (tail reduction; always required).
  const void* A; const void* Wpp; const float* bp; const float* b2_logical;
  // optional second output: xn_out (16-bit, fragment-blocked) = LayerNorm(x_new; gamma_n, beta_n) — the NEXT block's
  // norm1, computed in the epilogue where a lane pair holds the whole new row (feeds qkvattn.hip)
  void* xn_out; const float* gamma_n; const float* beta_n;
};
bool mlp_fused_supported(int prec, int D, int H);
int mlp_fused(int prec, const MlpArgs& a, hipStream_t s);

// qkvattn.hip — fused norm1 + attn.qkv + softmax(q k^T / 8) v per image (the qkv tensor never exists in HBM)
struct QkvAttnArgs {
  const void* xn;                   // norm1(x), 16-bit fragment-blocked [rows_alloc, D]
  const void* Wb; const float* bias;   // attn.qkv weight [3D, D] fragment-blocked, bias [3D]
  void* out;                        // attention output, 16-bit fragment-blocked [rows_alloc, D] (feature = head * 64 + dim)
  int B, T, D;                      // images, tokens per image, embed dim (heads = D / 64)
  int64_t rows_alloc;               // rows addressable in xn / out (multiple of 32, >= B * T)
  int cls_only;                     // 1: only the attention rows of token tile 0 of every image are needed (last block); a hint — shapes
                                    // without the specialised kernel compute every row
  int img0;                         // first image of this launch (images img0 .. B - 1; set by the launcher: 0, or the tail images of a batch)
  int hsplit;                       // workgroups per image (each runs heads / hsplit heads): 0 = chosen by the launcher (small batches
                                    // spread over the CUs), 1 = never split, n = at most n
};
bool qkv_attn_supported(int prec, int D, int T);
int qkv_attn_fused(int prec, const QkvAttnArgs& a, hipStream_t s);

// patch.hip — fused im2col + patch-embedding GEMM (+ bias + pos_embed) straight from the NCHW fp32 crops into the blocked residual stream
struct PatchArgs {
  const void* x; int B, H, W;       // crops [B,3,H,W] fp32 — or, with x16, already in the operand type (H, W multiples of 16)
  int x16;                          // 1: x holds 16-bit values of the kernel's operand type (the crop transform's 16-bit hand-off)
  const void* Wb;                   // patch_embed.proj.weight [D, 768] fragment-blocked (16-bit), k = (c, py, px)
  const float* bias; const float* pos;   // [D]; pos_embed rows [1 + P, D] fp32
  const float* cls;                 // optional [D]: cls_token + pos_embed[0] — the workgroup that holds an image's patch 0 also writes its class-token
                                    // row img * (P + 1) (one launch less per forward: set_cls_rows)
  float* out;                       // fp32 residual stream, fragment-blocked: row img * (P + 1) + 1 + p
  int D, P;                         // embed dim, patches per image
};
bool patch_embed_fused_supported(int prec, int D);
int patch_embed_fused(int prec, const PatchArgs& a, hipStream_t s);

// panel.hip — row-panel GEMM with optional fused LayerNorm prologue (K = embed dim)
enum { PRO_COPY = 0, PRO_LN = 1 };
struct PanelArgs {
  const void* A; int64_t lda;       // PRO_COPY: operand-type [M,K]; PRO_LN: fp32 residual stream [M,K] (dense rows)
  const float* gamma; const float* beta; float eps;   // PRO_LN
  const void* W;                    // [N,K] operand type, dense rows
  const float* bias;                // [N]
  void* out; int64_t ldo;
  const float* resid; int64_t ldr;  // EPI_BIAS_RESID (may alias out)
  int M, N, K;
  unsigned long long* dbg;          // optional timeline buffer (experiments)
  int debug;                        // experiment switches (bit0: skip epilogue stores)
  int blk_a;                        // A (PRO_COPY: 16-bit, PRO_LN: fp32 x) is fragment-blocked
  int blk_out;                      // out (and resid) are fragment-blocked
  int panel_rows;                   // 128 (one workgroup per CU) or 64 (two); 0 = default
  int rows_padded;                  // out / resid buffers are addressable up to the next multiple of 128 rows
  int no_tail_split;                // 1: one workgroup per panel even in the last, partially filled round (A/B switch)
  int tail_first, tail_split;       // set by the launcher: panels >= tail_first are cut along N into tail_split workgroups
};
bool panel_gemm_supported(int prec, int N, int K);
int panel_gemm(int prec, int pro, int epi, const PanelArgs& a, hipStream_t s);

// vit_ops.hip
int reduce_layernorm_rows_blocked(int prec_out, float* x, int64_t rows, int D, const float* partial, const float* b2, int nparts, int tail_rb,
                                  int add_x, const float* gamma, const float* beta, float eps, void* out, hipStream_t s);
int layernorm_rows_blocked(int prec_out, const float* x, int64_t rows, int D, const float* gamma, const float* beta,
                           float eps, void* out, hipStream_t s);
int layernorm_rows(int prec_out, const float* x, int64_t rows, int D, const float* gamma, const float* beta,
                   float eps, void* out, hipStream_t s);
int im2col_patch16(int prec_out, const void* x, int x16, int B, int H, int W, void* out, hipStream_t s);   // x16: x is already 16-bit (prec_out's type)
int clock_sample(unsigned long long* out, hipStream_t s);
int set_cls_rows(const float* cls_pos0, float* x, int B, int T, int D, int blocked, int* status_zero, hipStream_t s);
// rows img*T of x (fp32 blocked) and att (16-bit blocked) -> compact blocked buffers of B rows (last block: class tokens only)
int gather_cls_rows_blocked(const float* x, const void* att, int B, int T, int D, float* xc, void* ac, hipStream_t s);
int attention(int prec, const void* qkv, void* out, int B, int T, int heads, int blocked, hipStream_t s);
int final_cls_norm(const float* x, int B, int T, int D, const float* gamma, const float* beta, float eps,
                   int l2norm, int blocked, float* emb, int* status, hipStream_t s);

// knn.hip
size_t knn_workspace_bytes(int64_t B, int64_t N, int D, int k);
int knn_ip_topk(const float* q, int64_t B, const float* xb, int64_t N, int D, int k, float* dist, int64_t* idx,
                void* ws, size_t ws_bytes, hipStream_t s);
// screened search for large indexes: bf16-MFMA screening with a rigorous error bound + exact re-rank; results are
// bit-identical to knn_ip_topk.  xb16 = bf16 copy of xb (convert_bf16), xnorm_max >= every row's L2 norm.
size_t knn_screen_workspace_bytes(int64_t B, int64_t N, int D, int k);
void knn_set_wg_target(int n);
void knn_q16_tile(int on);            // A/B: 0 = <= 16 queries on the 32-wide streaming tile too (default 1: 16-wide tile, v_mfma_f32_16x16x4_f32)
void knn_two_pass_screen(int on);     // A/B: 1 = the screened search collects candidates with a second bf16 scan (default 0: from pass 1's chunk lists)
void knn_force_tile_kernel(int on);   // A/B: 1 = the 128-query tile kernel also for <= 32 queries (default 0: streaming kernel)
size_t knn_screen_flag_offset(int64_t B, int64_t N, int D, int k);   // byte offset of the int32 overflow flag in that workspace
// xb16: row-major bf16 copy (convert_bf16; may be NULL when xblk serves the call), xblk: fragment-blocked bf16 copy
// (convert_bf16_blocked: [ceil(N / 64) * 2][D / 8][32][8], zero rows past N; may be NULL) — with it, D in {128, 384, 768} and k <= 16 the
// screening pass is the Q-stationary kernel (knn_qs_kernel)
int knn_ip_topk_screened(const float* q, int64_t B, const float* xb, const void* xb16, const void* xblk, int64_t N, int D, int k, float xnorm_max,
                         float* dist, int64_t* idx, void* ws, size_t ws_bytes, hipStream_t s);
int convert_bf16_blocked(const float* src, int64_t N, int D, void* dst, hipStream_t s);
void knn_qs_option(int which, int value);   // 0: on / off (A/B), 1: workgroups per launch (0 = one per CU)
int convert_bf16(const float* src, int64_t n, void* dst, hipStream_t s);
int l2_normalize_rows(const float* x, int64_t B, int D, float* y, hipStream_t s);
int gather_rows(const float* src, const int64_t* rows, int64_t n, int D, float* dst, hipStream_t s);

// transform.hip — create_paired_transform over a box list (crop, pad to square, /255, bilinear resize, normalise)
// out_prec: PREC_FP32 (float out) or PREC_BF16 / PREC_FP16 — the fp32 result rounded once (round-to-nearest-even) to that type
int crop_transform(const uint8_t* img, int n_img, int64_t img_stride, int H, int W, int64_t stride, const int* boxes, int box_ld, int64_t n, int S,
                   int antialias, const float* mean, const float* stdv, const float* fill, void* out, int out_prec, hipStream_t s);

// resnet.hip
struct ConvArgs {
  const float* in;      // NHWC fp32 [B,H,W,in_ld] (channels in_off .. in_off+Cin of every pixel)
  const float* w;       // [Cout][KH*KW*Cin] fp32, BN folded
  const float* bias;    // [Cout] folded BN shift
  const float* resid;   // optional NHWC [B,OH,OW,res_ld] (+ res_off)
  float* out;           // NHWC [B,OH,OW,out_ld] (channels out_off .. out_off+Cout)
  int B, H, W, Cin, Cout, KH, KW, stride, pad, OH, OW, relu;
  // channel-slice addressing (0 = dense: ld = channel count, offset 0): lets producers write straight into the slices of a
  // concatenation buffer and consumers read one (YOLOv5 C3 / SPPF / neck concats never materialise a copy)
  int in_ld, in_off, out_ld, out_off, res_ld, res_off;
  int silu;             // epilogue x * sigmoid(x) (ultralytics Conv = Conv2d + BN + SiLU); relu and silu are exclusive
  float* partial; size_t partial_bytes;   // optional scratch: a launch of few tiles and a long K is split over K (fp32 partial tiles
                        // + a fixed-order reduction kernel that applies the epilogue); set by the caller, used by conv2d_nhwc when it pays
  int ksplit;           // set by conv2d_nhwc
  const void* w16;      // optional: the weights rounded to bf16, [Cout][ceil(KH*KW*Cin / 64) * 64] zero-padded -> bf16-operand MFMAs
                        // (activations rounded in the stage loader, fp32 accumulation / epilogue); NULL = exact fp32 operands
};
int conv2d_nhwc(const ConvArgs& a, hipStream_t s);
// conv1 7x7/2 pad 3 im2col straight from the NCHW input: rows [B*OH*OW][160] (147 taps (ky,kx,c) + zero pad)
int im2col_conv1(const float* x, float* col, int B, int H, int W, int OH, int OW, hipStream_t s);
int maxpool3x3s2_nhwc(const float* in, float* out, int B, int H, int W, int C, int OH, int OW, hipStream_t s);
int global_avgpool_nhwc(const float* in, float* out, int B, int HW, int C, int l2norm, hipStream_t s);

// yolo.hip — the non-conv pieces of the YOLOv5 localizer (onnx_engines/localizer_engine.py) on NHWC fp32 activations
// generic im2col for a stem conv with few input channels: x NCHW [B,Cin,H,W] -> col [B*OH*OW][kpad], k = (ky*KW + kx)*Cin + c, zero padded
int im2col_nchw(const float* x, float* col, int B, int Cin, int H, int W, int KH, int KW, int stride, int pad, int OH, int OW, int kpad, hipStream_t s);
// YOLOv5 stem Conv(3, 32, 6, 2, 2) + bias + SiLU straight from NCHW (w: [32][w_ld] rows of (ky*6 + kx)*3 + c taps) -> NHWC channel slice
// wt (optional): the same weights transposed to [108 taps][32 channels] — with it, an even W and an 8-byte aligned x the scalar-weight kernel runs
int stem6x6s2_nchw(const float* x, const float* w, int w_ld, const float* wt, const float* bias, float* out, int B, int H, int W, int OH, int OW, int out_ld,
                   int out_off, int silu, hipStream_t s);
int upsample2x_nhwc(const float* in, int in_ld, int in_off, float* out, int out_ld, int out_off, int B, int H, int W, int C, hipStream_t s);
int maxpool5_nhwc(const float* in, int in_ld, int in_off, float* out, int out_ld, int out_off, int B, int H, int W, int C, hipStream_t s);
// Detect head decode of one level: raw [B,ny,nx,raw_ld] (channel a*no + o) -> pred [B,total,no] rows row0 + (a*ny + y)*nx + x
int yolo_decode(const float* raw, int raw_ld, float* pred, int B, int ny, int nx, int na, int no, float stride, const float* anchors_px,
                int64_t total, int64_t row0, hipStream_t s);
int letterbox_u8(const uint8_t* img, int H, int W, int64_t row_stride, int bgr, int out_h, int out_w, int new_h, int new_w, int top, int left,
                 float fill, float* out, hipStream_t s);
size_t nms_workspace_bytes(int n, int max_nms);
bool nms_greedy_applies(int n, int max_det, int max_nms);
int nms_yolo_batch(const float* pred, int B, int n, int nc, float conf_thres, float iou_thres, int max_det, int max_nms, float max_wh, int agnostic,
                   float* out, int* count, void* ws, size_t ws_bytes, hipStream_t s);
int nms_yolo(const float* pred, int n, int nc, float conf_thres, float iou_thres, int max_det, int max_nms, float max_wh, int agnostic,
             float* out, int* count, void* ws, size_t ws_bytes, hipStream_t s);

}  // namespace effocr
