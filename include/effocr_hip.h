/* effocr_hip.h — C ABI of libeffocr_hip.so: the MI355X (gfx950) implementation of EffOCR's
 * recognizer hot path (encoder forward -> L2 normalise -> exact inner-product top-k).
 *
 * The reference (dell-research-harvard/effocr) is pure Python and has no FFI of its own; its
 * "operator API" for this path is three Python call conventions.  Each entry point below names the
 * reference interface it sits underneath (file:line into the reference tree); the Python classes
 * that keep those conventions bit-for-bit live in effocr_amd/ and call ONLY this header's
 * functions through ctypes.  INTEGRATION.md shows the reference-side binding.
 *
 * Conventions
 *   - plain C types only; every *_dev pointer is caller-owned DEVICE memory (hipMalloc / torch
 *     allocator).  The library allocates no device memory, ever; host memory only inside the
 *     encoder handle (parameter staging).
 *   - every call that takes a `stream` is asynchronous on that hipStream_t (passed as void*; NULL =
 *     the default stream) and performs no device synchronisation.
 *   - return value: 0 on success, a negative EFFOCR_E* code on failure; the message is available
 *     from effocr_last_error() (thread-local).
 */
#ifndef EFFOCR_HIP_H
#define EFFOCR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* bumped whenever an exported signature or the meaning of an argument changes (round 1: 1; round 2 added
 * arguments to effocr_op_mlp_blocked without a bump — callers must treat 1 as "unknown layout"); a caller built
 * against another value must refuse to call into the library (effocr_amd/_lib.py does) */
#define EFFOCR_ABI_VERSION 9

enum effocr_status {
  EFFOCR_OK = 0,
  EFFOCR_EINVAL = -1,        /* bad argument (shape, NULL pointer, unknown name)            */
  EFFOCR_EUNSUPPORTED = -2,  /* valid request outside what the kernels implement            */
  EFFOCR_EWORKSPACE = -3,    /* caller-provided workspace / blob too small                  */
  EFFOCR_EHIP = -4,          /* HIP runtime error (launch failure, memcpy failure)          */
  EFFOCR_ESTATE = -5,        /* call order violated (e.g. forward before upload)            */
  EFFOCR_EOVERFLOW = -6      /* non-finite result: a 16-bit operand overflowed (f16 mode) or the input was not finite */
};

/* arithmetic type of the encoder's MFMA operands (accumulation, LayerNorm, softmax, the residual
 * stream and the embedding are always fp32) */
enum effocr_precision {
  EFFOCR_PREC_BF16 = 0,      /* v_mfma_f32_32x32x16_bf16 — the BASELINE.json configuration   */
  EFFOCR_PREC_FP16 = 1,      /* v_mfma_f32_32x32x16_f16                                      */
  EFFOCR_PREC_FP32 = 2       /* v_mfma_f32_32x32x2_f32, exact fp32 — the parity mode         */
};

int effocr_abi_version(void);
const char* effocr_last_error(void);

/* ------------------------------------------------------------------------------------------
 * Encoder engine.  Replaces: onnx_engines/recognizer_engine.py:6-27 (EffRecognizer: ORT session
 * over enc_best.onnx, run(imgs[B,3,H,W] f32) -> [embs[B,D] f32]) and the torch twin
 * models/encoders.py:50-70 (AutoEncoderFactory("timm", name) -> timm.create_model(name,
 * num_classes=0); forward(x) = pooled features; load(ckpt) = state dict with "net." keys),
 * called at infer_effocr.py:177-179,314 and infer_effocr_onnx_multi.py:161-163,491-494.
 * ------------------------------------------------------------------------------------------ */
typedef struct effocr_encoder effocr_encoder_t;

/* arch: "resnet18" | "vit_small_patch16_224" | "vit_base_patch16_224" ("vit_tiny_test" for tests).
 * img_size: input H = W (224 for the ViTs' pos_embed; any multiple of 32 for resnet18). */
int effocr_encoder_create(const char* arch, int img_size, int precision, effocr_encoder_t** out);
void effocr_encoder_destroy(effocr_encoder_t* enc);
int effocr_encoder_embed_dim(const effocr_encoder_t* enc);

/* Parameter table = the timm state-dict keys WITHOUT the "net." prefix (models/encoders.py:60). */
int effocr_encoder_num_params(const effocr_encoder_t* enc);
const char* effocr_encoder_param_name(const effocr_encoder_t* enc, int i);
int64_t effocr_encoder_param_numel(const effocr_encoder_t* enc, int i);
/* copy one fp32 HOST tensor (C-contiguous, torch layout) into the handle's staging area */
int effocr_encoder_set_param(effocr_encoder_t* enc, const char* name, const float* host, int64_t numel);

/* Packed device image of the weights: LayerNorm/bias/pos_embed in fp32, GEMM operands in the
 * handle's precision, BatchNorm folded into the conv weights (resnet18). */
size_t effocr_encoder_weights_bytes(const effocr_encoder_t* enc);
/* pack all staged parameters and copy them into weights_dev (synchronous; not on the hot path) */
int effocr_encoder_upload(effocr_encoder_t* enc, void* weights_dev, size_t bytes);

size_t effocr_encoder_workspace_bytes(const effocr_encoder_t* enc, int batch);
/* x_dev  : [B,3,img,img] fp32 NCHW, C-contiguous (the tensor infer_effocr.py:313 stacks)
 * emb_dev: [B,D] fp32.  l2_normalize != 0 fuses F.normalize(p=2,dim=1) (infer_effocr.py:316). */
int effocr_encoder_forward(effocr_encoder_t* enc, const float* x_dev, int batch, float* emb_dev,
                           int l2_normalize, void* workspace_dev, size_t workspace_bytes, void* stream);
/* The same with the crops' element type stated (ABI 6; SURVEY §8 f-2's hand-off "uint8 line image + box list -> [B,3,224,224] bf16"):
 * x_dtype = EFFOCR_PREC_FP32 (x_dev as above) or — ViT encoders in a 16-bit precision mode only — the encoder's OWN operand type
 * (EFFOCR_PREC_BF16 / EFFOCR_PREC_FP16: x_dev holds [B,3,img,img] 16-bit values, what effocr_crop_transform_batch_ex writes).  The patch
 * embedding rounds an fp32 crop to that type before its MFMAs anyway, so a crop that was rounded once by the producer yields
 * BIT-IDENTICAL embeddings at half the input bytes.  Any other combination: EFFOCR_EUNSUPPORTED. */
int effocr_encoder_forward_ex(effocr_encoder_t* enc, const void* x_dev, int x_dtype, int batch, float* emb_dev,
                              int l2_normalize, void* workspace_dev, size_t workspace_bytes, void* stream);

/* Measurement aid (bench.py): one small launch on `stream` (1024 one-wave workgroups) that stores { s_memtime (shader-clock ticks),
 * s_memrealtime (100 MHz) } of every CU it reaches into out_dev[2 key], out_dev[2 key + 1], key = XCC_ID * 256 + SE * 32 + SH * 16 + CU
 * < 2048 (out_dev: 4096 uint64, zeroed by the caller; s_memtime is a per-CU counter, so two samples are compared CU by CU).
 * d(memtime) / d(memrealtime) x 100 MHz = the average shader clock over the bracketed interval — the chip is power-managed:
 * tools/ubench/mfma_f16_vs_bf16.hip measures 2.37 GHz idle-data, 1.73 GHz (bf16) / 1.58 GHz (f16) under saturated MFMA load. */
int effocr_clock_sample(void* out_dev, void* stream);

/* Status of EVERY forward issued with this workspace since the previous check (ViT; the CNN path is fp32 throughout and always
 * reports OK): the forward keeps an int32 status word at workspace offset 0 and only ever ORs into it (its last kernel); this call
 * copies it on `stream`, synchronises the stream, CLEARS it if it was set and returns EFFOCR_EOVERFLOW if an embedding came out
 * non-finite — one check covers the slices of a large call, internal sub-batches and any number of asynchronous forwards (ABI 6;
 * up to ABI 5 the first kernel of each forward zeroed the word, so only the last forward was visible).  The caller zeroes the first
 * 256 bytes of a freshly allocated workspace once (effocr_encoder_reset_status, or a memset).  In f16 mode that is how operand overflow surfaces:
 * q / k / v and the fc1 pre-activations are rounded to f16 (max 65504), an overflow becomes inf, and an inf anywhere in a block turns
 * the LayerNorm / softmax of every row it feeds into nan — it cannot stay hidden in a finite embedding.  (The reference computes in
 * fp32, infer_effocr.py:314-316; precision "bf16" / "fp32" have fp32's exponent range.)  Not on the hot path: call it where the
 * caller synchronises anyway. */
int effocr_encoder_check_status(const effocr_encoder_t* enc, const void* workspace_dev, void* stream);
/* zero the status word of a (freshly allocated) workspace, asynchronously on `stream` */
int effocr_encoder_reset_status(const effocr_encoder_t* enc, void* workspace_dev, void* stream);

/* ViT only: run the forward in internal sub-batches of `crops_per_chunk` crops (0 = whole batch) so
 * that the activations between consecutive kernels stay in the 256 MiB Infinity Cache.  Results
 * are identical for every setting; it only changes the workspace size and the launch count. */
int effocr_encoder_set_chunk(effocr_encoder_t* enc, int crops_per_chunk);
/* tuning / A-B switches (ViT, 16-bit precisions; every combination is parity-tested against the oracle,
 * results differ only by operand-rounding noise of the precision mode).  [default]
 *   "use_panel"   [1] row-panel GEMMs with the LayerNorm fused into the panel load; 0: K-streaming GEMMs + LayerNorm kernel
 *   "use_blocked" [1] fragment-blocked activation layout; 0: row-major activations
 *   "use_qkvattn" [1] attn.qkv + attention as one kernel per image (ViT-S / 128-wide; no qkv tensor in memory) for batches of
 *                     192 crops or more (one image per workgroup needs ~a round of CUs); 2: for every batch size;
 *                     0: row-panel LN1+qkv kernel + attention kernel
 *   "use_mlp"     [1] LN2+fc1+GELU+fc2+residual as one kernel; 0: row-panel fc1 + K-streaming fc2
 *   "use_gemm3"   [1] 128-row wave-tile GEMM over blocked operands (fc2; every ViT-B linear); 0: gemm2 / gemm
 *   "use_gemm2"   [1] DMA-ring GEMM for the patch embedding (and fc2 when use_gemm3 = 0)
 *   "tail_split"  [1] split the panels / tiles of the last, partially filled round of CUs
 *   "cls_only_last" [1] last transformer block: attn.proj + MLP only on the class-token row of every image — the only row that
 *                 reaches the embedding (global_pool = 'token'); proj / LayerNorm / MLP act per row, so the result is the same
 *                 (0: all tokens, A/B switch)
 *   "split6"      [1] fused MLP: calls of <= 27 crops cut their panels 6-way over the hidden dimension (0: 4-way)
 *   "pair_parts"  [1] fused proj+MLP: calls of <= 36 crops (and the class-token rows of the last block) deal the hidden chunks of their
 *                 64-token pair panels over 6 / 3 / 2 workgroups + the reduction launch (0: 128-token panels over six / four, whole pair panels from 30 crops)
 *   "mlp_pair"    [0] fused proj+MLP: 64-token panels whose wave pairs split a chunk's hidden features (no partial sums in HBM, no
 *                 reduction launch): 0 = for calls of 37..83 crops (below: "pair_parts"), 1 = whenever the 64-token panels fit one round of CUs, -1 = never
 *   "mlp_stagger" [3500] fused MLP kernel: the first round of workgroups starts spread over 32 x this many clock ticks, so
 *                 that the CUs do not request / store their rows all at the same moment (0 = off)
 *   "mlp_stagger_min_rounds" [2] ... for launches of at least this many rounds of CUs (512-crop calls: +7.7 %; no effect below two)
 *   "use_projf"   [1] attn.proj + residual fused in front of the fused MLP kernel (0: its own row-panel launch)
 *   "use_patchf"  [1] fused im2col + patch-embedding GEMM (embed dims 128 / 256 / 384 / 768); 0: im2col kernel + DMA-ring GEMM
 *   "qa_min_batch" [1] fused qkv + attention kernel from this many crops per call on (below: row-panel qkv + attention kernels)
 *   "qa_hsplit"   [0] fused qkv + attention: workgroups per image (heads split over them); 0 = launcher's choice (calls of less
 *                 than a round of CUs split), 1 = never, n = at most n
 *   "use_lnfold"  [1] gemm3 path (embed dims that are multiples of 256: ViT-B): LayerNorm folded into the linears either side of it —
 *                 attn.proj / mlp.fc2 also write the new residual row as 16-bit operands + per-row partial sums, attn.qkv / mlp.fc1
 *                 multiply by W . diag(gamma) and finish rstd (acc - mean s) + (b + W beta) in their epilogues: 1 LayerNorm launch
 *                 per forward instead of 24 (0: LayerNorm launches, A/B switch)
 *   "panel_rows"  [128] row-panel height, 64 or 128;  "chunk" (= set_chunk);  "debug" (experiment hooks) */
int effocr_encoder_set_option(effocr_encoder_t* enc, const char* name, int value);

/* HIP-event profiler for bench.py's roofline object: while armed, every launch of the selected
 * kernel classes inside effocr_encoder_forward is bracketed by an event pair on the forward's own
 * stream.  mode 0 = off, 1 = every class, 2 = only `only_class` (e.g. "gemm_fc1_gelu").
 * profile_collect synchronises the recorded events, disarms, and returns the number of classes;
 * profile_get reads class i: summed duration (ms), launch count and summed algorithmic FLOPs. */
int effocr_encoder_profile_begin(effocr_encoder_t* enc, int mode, const char* only_class);
int effocr_encoder_profile_collect(effocr_encoder_t* enc);
int effocr_encoder_profile_get(const effocr_encoder_t* enc, int i, const char** name, double* total_ms,
                               int* launches, double* total_work);
/* mode 1 only: the shader clock (GHz) the launches of class i ran at — each launch is also bracketed, outside its event pair, by two
 * samples of s_memtime (shader clocks) and s_memrealtime (100 MHz); 0 if not sampled.  The chip is power-limited: MFMA-dense kernels
 * run at 1.2-1.4 GHz, not at the 2.4 GHz the 2.5 PFLOP/s peak is quoted at ("Clocks"). */
int effocr_encoder_profile_clock(const effocr_encoder_t* enc, int i, double* shader_ghz);

/* ------------------------------------------------------------------------------------------
 * k-NN engine.  Replaces faiss.IndexFlatIP as driven by pytorch_metric_learning's FaissKNN:
 * infer_effocr.py:184-187,207,211,317; infer_effocr_onnx_multi.py:496-500,509,372;
 * train_effocr_recognizer.py:47-52.
 * ------------------------------------------------------------------------------------------ */
size_t effocr_knn_workspace_bytes(int64_t nq, int64_t ntotal, int d, int k);
/* IndexFlatIP.search: q_dev [nq,d] fp32, xb_dev [ntotal,d] fp32 row-major ->
 * dist_dev [nq,k] fp32 descending, idx_dev [nq,k] int64; k > ntotal pads (-FLT_MAX, -1).
 * Scores are the ascending-k fp32 fmaf chain; equal scores rank by ascending id.  k <= 32. */
int effocr_knn_ip_topk(const float* q_dev, int64_t nq, const float* xb_dev, int64_t ntotal, int d, int k,
                       float* dist_dev, int64_t* idx_dev, void* workspace_dev, size_t workspace_bytes,
                       void* stream);
/* The same search for LARGE indexes (e.g. BASELINE configs[3], 1M x 768): a bf16-MFMA screening pass finds every row
 * whose approximate score s^ lies within 2*eps of the k-th largest s^ (eps = c*|q|*xnorm_max bounds |s^ - s| rigorously:
 * operand rounding 2^-8 + 2^-16, fp32 accumulation 4*d*2^-24), the candidates are re-ranked with the exact ascending-k
 * fmaf chain, and — gated on the device — the exact search runs over everything if a query had more than 512
 * candidates.  dist / idx are BIT-IDENTICAL to effocr_knn_ip_topk for every input.  For nq <= 128 against >= 65 536 rows with
 * d % 192 == 0 (the ONNX driver's 64-crop batches, infer_effocr_onnx_multi.py:372) the screening pass STREAMS the bf16 copy once per
 * 64 queries at the HBM rate (1M x 384, 64 queries: 0.33 ms); larger batches use the 128-query tile kernel.
 *   xb_bf16_dev  bf16 copy of xb_dev made with effocr_convert_bf16;  xnorm_max >= the L2 norm of every index row
 *   d % 64 == 0, k <= 32, ntotal >= k. */
size_t effocr_knn_screen_workspace_bytes(int64_t nq, int64_t ntotal, int d, int k);
/* Process-wide A/B switch of the exact search: "force_tile" [0] — 1 makes effocr_knn_ip_topk use the 128-query tile kernel
 * also for <= 32 queries (by default those go to the streaming kernel that reads the index at the HBM rate; results are
 * bit-identical either way).  "wg_target" [1024] — workgroups a tile-kernel launch aims for when it cuts the index into chunks
 * (set it before effocr_knn_workspace_bytes: the partial-list area follows the chunk count; results are bit-identical).
 * "q16_tile" [1] — calls of <= 16 queries use the 16-wide query tile (v_mfma_f32_16x16x4_f32); 0: the 32-wide one.
 * "two_pass_screen" [0] — 1: the screened search collects its candidates with a second bf16 scan of the index instead of reading
 * them out of the first scan's per-chunk lists (A/B switches; results are bit-identical either way). */
int effocr_knn_set_option(const char* name, int value);
/* Byte offset, inside the screened search's workspace, of its int32 OVERFLOW FLAG: non-zero after a call in which some
 * query had more than 512 candidates within the error band, i.e. the call also ran the exact pass (results are
 * identical either way; a caller that sees it repeatedly should use effocr_knn_ip_topk directly). */
size_t effocr_knn_screen_flag_offset(int64_t nq, int64_t ntotal, int d, int k);
int effocr_knn_ip_topk_screened(const float* q_dev, int64_t nq, const float* xb_dev, const void* xb_bf16_dev, int64_t ntotal, int d,
                                int k, float xnorm_max, float* dist_dev, int64_t* idx_dev, void* workspace_dev,
                                size_t workspace_bytes, void* stream);
int effocr_convert_bf16(const float* src_dev, int64_t n, void* dst_dev, void* stream);
/* ABI 6: the screened search with a FRAGMENT-BLOCKED bf16 copy of the index (cells [row / 32][k chunk of 8][row % 32][16 B] — the layout
 * of the encoder's weight copies; effocr_convert_bf16_blocked writes it, effocr_bf16_blocked_bytes sizes it: rows are padded to a
 * multiple of 64 with zeros).  With it, d in {128, 384, 768} and k <= 16 the screening pass is the Q-STATIONARY kernel: the queries of a
 * workgroup (256 at d <= 384, 128 at d = 768) live in registers as MFMA operand fragments, the index streams through a 6-slot LDS-DMA
 * ring as verbatim 512-byte cells (conflict-free fragment reads, no transposes), NO per-lane lists (block maxima only, below) — for EVERY index size:
 * BASELINE configs[1]'s own 10 000-row search and configs[3]'s 1M x 768 alike.  xb_bf16_dev (row-major copy) may be NULL then, except
 * for calls of 17..128 queries against >= 65 536 rows, which keep the streaming screen when it is given.  Results are bit-identical to
 * effocr_knn_ip_topk either way.  The pass writes only the MAXIMUM approximate score per 16-row block (and per sub-chunk); a rank-count
 * kernel turns the sub-chunk maxima into a per-query threshold, a collect kernel gathers the blocks above it, and the re-rank kernel
 * re-scores their rows from the blocked copy before the exact fmaf chains.  Workspace: effocr_knn_screen_workspace_bytes
 * (it includes the block maxima: ntotal / 16 * nq * 4 bytes).  effocr_knn_set_option: "qs" [1] (0: ignore the blocked copy, A/B),
 * "qs_wgs" [0 = one per CU], "stream_min_rows" [65536] (33..128 queries: index rows from which effocr_knn_ip_topk streams the index
 * instead of running the tile kernel; measured slower below, tools/knn_small_time.py). */
size_t effocr_bf16_blocked_bytes(int64_t n_rows, int d);
int effocr_convert_bf16_blocked(const float* src_dev, int64_t n_rows, int d, void* dst_dev, void* stream);
int effocr_knn_ip_topk_screened2(const float* q_dev, int64_t nq, const float* xb_dev, const void* xb_bf16_dev, const void* xb_bf16_blk_dev,
                                 int64_t ntotal, int d, int k, float xnorm_max, float* dist_dev, int64_t* idx_dev, void* workspace_dev,
                                 size_t workspace_bytes, void* stream);
/* torch.nn.functional.normalize(x, p=2, dim=1) (infer_effocr.py:316; PML InferenceModel default) */
int effocr_l2_normalize(const float* x_dev, int64_t n, int d, float* y_dev, void* stream);
/* IndexFlat.remove_ids compaction (infer_effocr.py:211): dst[i] = src[keep_rows[i]] */
int effocr_gather_rows(const float* src_dev, const int64_t* keep_rows_dev, int64_t n_keep, int d,
                       float* dst_dev, void* stream);

/* ------------------------------------------------------------------------------------------
 * Crop pre-processing: `create_paired_transform(size)` applied to every box of one
 * page / line image — utils/datasets_utils.py:69-90 (MedianPad(override) = pad right/bottom to a
 * square), :166-172 (ToTensor, Resize((size,size)), Normalize); per-box call site
 * infer_effocr.py:286-293, infer_effocr_onnx_multi.py:326-345.
 *   image_dev  uint8 HWC RGB, `row_stride` bytes between rows (>= 3*width)
 *   boxes_dev  int32 [n,4] = x0,y0,x1,y1 exactly as numpy slicing im[y0:y1, x0:x1] resolves them
 *              (0 <= x0 < x1 <= width, 0 <= y0 < y1 <= height; the host wrapper rejects empty boxes
 *              with ValueError like PIL does); an invalid box yields a zero crop, never a bad read
 *   antialias  torchvision's tensor Resize default: 1 from 0.17 on, 0 before
 *   mean/std/fill  3 host floats each (ImageNet mean/std, pad colour on the 0..255 scale)
 *   out_dev    fp32 [n,3,size,size] (the encoder's input layout); size % 4 == 0; any n (launched in slices of 65535 boxes)
 * ------------------------------------------------------------------------------------------ */
int effocr_crop_transform(const uint8_t* image_dev, int height, int width, int64_t row_stride,
                          const int32_t* boxes_dev, int n, int size, int antialias, const float* mean,
                          const float* std, const float* fill, float* out_dev, void* stream);

/* The same over the boxes of SEVERAL images of one geometry in one launch — the crops of a whole run_effocr call
 * (infer_effocr_onnx_multi.py:313-345 cuts every line image's boxes on the host, one PIL read per line):
 *   images_dev  n_images uint8 HWC images, `image_stride` bytes apart
 *   boxes_dev   int32 [n,5] = x0,y0,x1,y1,image index; a box naming no image, or an empty one, yields a ZERO crop — what
 *               create_batches substitutes for a crop whose transform failed (:145-147, TransformationThread :196-200)
 *   n           any count (launched in slices of 65535 boxes) */
int effocr_crop_transform_batch(const uint8_t* images_dev, int n_images, int64_t image_stride, int height, int width,
                                int64_t row_stride, const int32_t* boxes_dev, int64_t n, int size, int antialias,
                                const float* mean, const float* std, const float* fill, float* out_dev, void* stream);
/* ... with the output element type stated (ABI 6): out_dtype = EFFOCR_PREC_FP32 (as above) or EFFOCR_PREC_BF16 / EFFOCR_PREC_FP16 —
 * out_dev [n,3,size,size] 16-bit, each value the fp32 result rounded ONCE (round to nearest even) to that type: the encoder's input
 * for effocr_encoder_forward_ex (half the bytes written here and read there; same embeddings bit for bit). */
int effocr_crop_transform_batch_ex(const uint8_t* images_dev, int n_images, int64_t image_stride, int height, int width,
                                   int64_t row_stride, const int32_t* boxes_dev, int64_t n, int size, int antialias,
                                   const float* mean, const float* std, const float* fill, int out_dtype, void* out_dev, void* stream);

/* ------------------------------------------------------------------------------------------
 * Localizer engine: the YOLOv5 character / word detector the reference runs through ONNXRuntime
 * (onnx_engines/localizer_engine.py:14-66 EffLocalizer with model_backend == 'yolo'; driver
 * infer_effocr_onnx_multi.py:236-262).  Same handle protocol as the encoder: create -> set_param x N ->
 * upload -> forward.  arch "yolov5s" (ultralytics v6 yaml); parameter names are the ultralytics state-dict
 * keys ("model.0.conv.weight", "model.0.bn.running_var", ..., "model.24.m.2.bias", "model.24.anchors").
 *   forward:  x_dev [B,3,in_h,in_w] fp32 (letterboxed RGB, 0..1 — what load_localizer_img builds, :75-85)
 *             -> pred_dev [B, num_predictions, 5 + num_classes] fp32 = the exported model's output 0
 *             (xywh in input pixels, objectness, class probabilities; 25200 rows at 640 x 640).
 * ------------------------------------------------------------------------------------------ */
typedef struct effocr_localizer effocr_localizer_t;
int effocr_localizer_create(const char* arch, int num_classes, int in_h, int in_w, effocr_localizer_t** out);
void effocr_localizer_destroy(effocr_localizer_t* loc);
int effocr_localizer_num_params(const effocr_localizer_t* loc);
const char* effocr_localizer_param_name(const effocr_localizer_t* loc, int i);
int64_t effocr_localizer_param_numel(const effocr_localizer_t* loc, int i);
int effocr_localizer_set_param(effocr_localizer_t* loc, const char* name, const float* host, int64_t numel);
size_t effocr_localizer_weights_bytes(const effocr_localizer_t* loc);
int effocr_localizer_upload(effocr_localizer_t* loc, void* weights_dev, size_t bytes);
/* "bf16_operands" [0]: 1 = every convolution that carries an activation runs with bf16-rounded operands (weights rounded once at
 * upload, activations in the stage loader) on v_mfma_f32_32x32x16_bf16, fp32 accumulation / bias / SiLU / residual; Detect's 1x1
 * heads keep fp32 operands.  0 = fp32 operands everywhere (v_mfma_f32_32x32x2_f32: the oracle's arithmetic up to summation order).
 * "direct_stem" [1]: the stem Conv(3, 32, 6, 2, 2) as a direct kernel from the NCHW input; 0 = im2col rows + the implicit GEMM (A/B
 * switch; set it before effocr_localizer_workspace_bytes — the im2col rows, 840 MB at 16 images, exist only on that path). */
int effocr_localizer_set_option(effocr_localizer_t* loc, const char* name, int value);
int64_t effocr_localizer_num_predictions(const effocr_localizer_t* loc);
int effocr_localizer_num_outputs(const effocr_localizer_t* loc);             /* 5 + num_classes */
size_t effocr_localizer_workspace_bytes(const effocr_localizer_t* loc, int batch);
int effocr_localizer_forward(effocr_localizer_t* loc, const float* x_dev, int batch, float* pred_dev, void* workspace_dev,
                             size_t workspace_bytes, void* stream);
/* EffLocalizer.letterbox + load_localizer_img (localizer_engine.py:75-85,107-138) for ONE image on the device:
 * image_dev uint8 HWC (bgr != 0: cv2.imread order, channels are reversed like the reference's [::-1]) is resized to
 * new_w x new_h with cv2.resize(INTER_LINEAR)'s fixed-point arithmetic, placed at (left, top) in an out_w x out_h canvas of
 * grey 114, scaled by 1/255 and written CHW (RGB) to out_dev [3,out_h,out_w] fp32.  The caller computes new_w/new_h/top/left
 * with the reference's formula (effocr_amd/localizer_engine.py letterbox_geometry). */
int effocr_letterbox(const uint8_t* image_dev, int height, int width, int64_t row_stride, int bgr, int out_h, int out_w, int new_h,
                     int new_w, int top, int left, float* out_dev, void* stream);
/* EffLocalizer.non_max_suppression (localizer_engine.py:171-277; single-label branch, no masks) for ONE image:
 * pred_dev [n, 5 + num_classes] -> out_dev [<= max_det, 6] = (x1, y1, x2, y2, conf, cls) in confidence order, *count_dev rows.
 * objectness > conf_thres, conf = obj * best class > conf_thres, at most max_nms (<= 32768) boxes by confidence, boxes offset by
 * cls * max_wh unless agnostic, greedy IoU > iou_thres suppression (torchvision.ops.nms), first max_det.  Equal confidences rank
 * by ascending prediction row (the reference's argsort leaves that order open). */
size_t effocr_nms_workspace_bytes(int n, int max_nms);
int effocr_nms(const float* pred_dev, int n, int num_classes, float conf_thres, float iou_thres, int max_det, int max_nms, float max_wh,
               int agnostic, float* out_dev, int* count_dev, void* workspace_dev, size_t workspace_bytes, void* stream);
/* The same for the `batch` images of one network call (the reference loops `for xi, x in enumerate(prediction)`,
 * localizer_engine.py:212): pred_dev [batch, n, 5 + num_classes] -> out_dev [batch, max_det, 6], count_dev [batch].
 * n <= 25600 and n <= max_nms (any max_det since ABI 5: the reference's default is 1000): ONE launch, one workgroup per image, greedy
 * keep-best / kill-overlaps rounds (cost = kept boxes x one pass over the candidates; a text line keeps tens), no workspace
 * (workspace_dev may be NULL); otherwise the per-image path above, image after image on the stream, with one workspace.
 * effocr_nms_batch_workspace_bytes says which: 0 = the one-launch path.  Identical rows either way. */
size_t effocr_nms_batch_workspace_bytes(int n, int max_det, int max_nms);
int effocr_nms_batch(const float* pred_dev, int batch, int n, int num_classes, float conf_thres, float iou_thres, int max_det, int max_nms,
                     float max_wh, int agnostic, float* out_dev, int* count_dev, void* workspace_dev, size_t workspace_bytes, void* stream);

/* ABI 8 — box stage of run_effocr (infer_effocr_onnx_multi.py:252-256,275-288,313-320; the stable sorted(..., key=x[axis]) of
 * en_preprocess / jp_preprocess :70-73,133-135) for all `lines` of one localizer call, two launches:
 * rows_dev [lines, max_det, 6] = (x0, y0, x1, y1, conf, label) as effocr_nms_batch wrote them, counts_dev [lines]; characters = label 0.
 *   sorted_dev  [lines, max_det, 4]  the boxes with the characters first, STABLY sorted by x0 (axis 0) or y0 (axis 1), the rest behind in row order
 *   n_chars_dev [lines]              characters per line
 *   boxes5_dev  [lines * max_det, 5] int32 (x0, y0, x1, y1, line): the crop slice of every character, compact, in line order — torch.round (half
 *                                    to even), x size / 640 in float64, Python round(), numpy slice resolution, full height (width if `vertical`)
 *   total_dev   [1]                  number of rows written to boxes5_dev
 * max_det <= 4096.  Bit-identical to the torch expression it replaces (tests/test_gpu_pipeline.py). */
int effocr_parse_char_boxes(const float* rows_dev, const int* counts_dev, int lines, int max_det, int height, int width, int axis, int vertical,
                            float* sorted_dev, int* n_chars_dev, int* boxes5_dev, int* total_dev, void* stream);

/* ------------------------------------------------------------------------------------------
 * Individual encoder operators (exported so that each kernel is parity-tested on its own).
 * Operand buffers are in `precision`'s element type unless stated fp32.
 * ------------------------------------------------------------------------------------------ */
enum effocr_epilogue { EFFOCR_EPI_BIAS = 0, EFFOCR_EPI_BIAS_GELU = 1, EFFOCR_EPI_BIAS_RESID = 2 };
/* out[m][n] = epi(sum_k x[m][k] w[n][k] + bias[n]); RESID: out (fp32) = resid (fp32) + ... */
int effocr_op_linear(int precision, int epilogue, const void* x_dev, const void* w_dev, const float* bias_dev,
                     const float* resid_dev, void* out_dev, int m, int n, int k, void* stream);
/* out = epi(LayerNorm(x)[m,:] . w[n,:] + bias[n]) with the LayerNorm fused into the operand load
 * (row-panel kernel; bf16/fp16, k in {128,384}, n % 128 == 0, n <= 4k); x_dev is fp32 [m,k]. */
int effocr_op_ln_linear(int precision, int epilogue, const float* x_dev, const float* gamma_dev, const float* beta_dev,
                        float eps, const void* w_dev, const float* bias_dev, const float* resid_dev, void* out_dev,
                        int m, int n, int k, void* stream);
int effocr_op_layernorm(int out_precision, const float* x_dev, int64_t rows, int d, const float* gamma_dev,
                        const float* beta_dev, float eps, void* out_dev, void* stream);
/* qkv_dev [B*T, 3*heads*64] -> out_dev [B*T, heads*64], head_dim fixed at 64 */
int effocr_op_attention(int precision, const void* qkv_dev, void* out_dev, int batch, int tokens, int heads,
                        void* stream);

/* Fast-path operators over the FRAGMENT-BLOCKED activation layout: a [rows, cols] matrix of
 * 16-byte chunks is stored as cells [row/32][chunk][row%32][16 B]; chunk = 8 elements for 16-bit operands, 4
 * for fp32.  Buffers must be addressable up to `rows_alloc` rows (a multiple of 32, >= m).
 *   linear_blocked:    out = epilogue(x . w^T + bias); x_blk [m,k] and w_blk [n,k] in `precision`'s 16-bit
 *                      type, out_blk 16-bit (epilogue 0/1) or fp32 with fp32 resid_blk (epilogue 2, may alias
 *                      out); n % 192 == 0 or n % 256 == 0, k % 128 == 0, k >= 256 (the timm Linear layers of
 *                      ViT-S fc2 and of every ViT-B block; models/encoders.py:58,63)
 *   layernorm_blocked: fp32 x_blk [rows,d] -> 16-bit out_blk, d in {128, 384, 768} */
int effocr_op_linear_blocked(int precision, int epilogue, const void* x_blk_dev, const void* w_blk_dev,
                             const float* bias_dev, const float* resid_blk_dev, void* out_blk_dev, int m, int n, int k,
                             int rows_alloc, void* stream);
/* Fused MLP of one transformer block on the blocked residual stream (timm Block: x + mlp(norm2(x)) with
 * Linear(d,h) -> GELU(erf) -> Linear(h,d); models/encoders.py:58,63):  x_blk (fp32, in/out) <- x + fc2(gelu(fc1(LN(x)))).
 * w1_blk: fc1.weight [h,d] 16-bit fragment-blocked.  w2_perm: fc2.weight [d,h] 16-bit fragment-blocked with the k
 * (hidden) index permuted inside every group of 16: element e of 16-byte chunk c holds
 * k = 16*(c/2) + (e&3) + 8*(e>>2) + 4*(c&1), AND its rows (output features) permuted inside every block of 32:
 * position p holds source row 8*(2*(r>>3) + hh) + (r&7) with hh = (p>>2)&1, r = (p&3) + 4*(p>>3)  ("P32": with it the
 * MFMA accumulators of a lane are the very fp32 chunks its LayerNorm read, so the kernel starts fc2's accumulators at
 * x + bias and never re-reads the residual).  b2_perm: fc2 bias P32-permuted; b2: the same bias unpermuted (tail reduction).
 * (d, h) in {(384,1536), (128,512)}; rows_alloc % 32 == 0, >= m.  The fp32 parameter arrays (gamma, beta, b1, b2_perm and, in the
 * variants below, gamma_next, beta_next, bp_perm) must be 16-byte aligned — they reach the workgroup's LDS by 16-byte DMA
 * (EFFOCR_EINVAL otherwise).
 * scratch_dev (optional, may be NULL): device scratch of scratch_bytes; with >= 64 MiB the 128-row panels of the last,
 * partially filled round of CUs are split over the hidden dimension and reduced in a fixed order (same result
 * bit for bit run to run; differs from the unsplit path only by fp32 summation order). */
int effocr_op_mlp_blocked(int precision, float* x_blk_dev, const float* gamma_dev, const float* beta_dev, float eps,
                          const void* w1_blk_dev, const float* b1_dev, const void* w2_perm_dev, const float* b2_perm_dev, const float* b2_dev,
                          int m, int d, int h, int rows_alloc, void* scratch_dev, size_t scratch_bytes, void* stream);
/* The same kernel with its SECOND output: xn_blk (16-bit blocked [m,d]) = LayerNorm(x_new; gamma_next, beta_next, eps), i.e.
 * the NEXT block's norm1 applied to the updated residual stream, computed in the epilogue where a lane pair holds the
 * whole new row (input of effocr_op_qkv_attn_blocked).  Other arguments as effocr_op_mlp_blocked. */
int effocr_op_mlp_ln_blocked(int precision, float* x_blk_dev, const float* gamma_dev, const float* beta_dev, float eps,
                             const void* w1_blk_dev, const float* b1_dev, const void* w2_perm_dev, const float* b2_perm_dev, const float* b2_dev,
                             const float* gamma_next_dev, const float* beta_next_dev, void* xn_blk_dev,
                             int m, int d, int h, int rows_alloc, void* scratch_dev, size_t scratch_bytes, void* stream);
/* attn.proj + residual fused in front of the MLP (everything a timm Block does after attention):
 *   x_blk <- y + fc2(gelu(fc1(LN(y)))),  y = x + a . wp^T + bp.
 * a_blk [m,d] 16-bit blocked (attention output).
 *   wp_perm  attn.proj.weight [d,d] blocked, rows P32-permuted (see op_mlp_blocked);  bp_perm  its bias, P32-permuted
 *   w2_perm, b2_perm, b2  as in op_mlp_blocked
 * Other arguments as effocr_op_mlp_blocked. */
int effocr_op_proj_mlp_blocked(int precision, float* x_blk_dev, const void* a_blk_dev, const void* wp_perm_dev, const float* bp_perm_dev,
                               const float* gamma_dev, const float* beta_dev, float eps, const void* w1_blk_dev, const float* b1_dev,
                               const void* w2_perm_dev, const float* b2_perm_dev, const float* b2_dev, int m, int d, int h, int rows_alloc,
                               void* scratch_dev, size_t scratch_bytes, void* stream);

/* attn.qkv + multi-head self-attention of one timm Block in ONE kernel, one image per workgroup at a time (timm
 * Attention.forward up to, not including, attn.proj; models/encoders.py:58,63):
 *   out_blk (16-bit [batch*tokens, d] blocked, feature = head*64 + dim) = softmax(q k^T / 8) v,
 *   [q | k | v] = xn_blk . wqkv^T + bias   (heads = d / 64)
 * xn_blk: norm1(x), 16-bit blocked [batch*tokens, d] (effocr_op_layernorm_blocked, or the fused MLP kernel's second output);
 * wqkv_blk: attn.qkv.weight [3d, d] 16-bit fragment-blocked with the v rows (rows 2d..3d) P32-permuted (see
 * effocr_op_mlp_blocked: then a lane of the output tile holds 8 consecutive head dims = one 16-byte store); bias: attn.qkv.bias
 * with its v part permuted alike.  d in {128, 384}; tokens <= 64 or in 193..224.
 * The qkv tensor never exists in device memory.  The bias array must be 16-byte aligned (it reaches LDS by 16-byte DMA; EFFOCR_EINVAL otherwise). */
int effocr_op_qkv_attn_blocked(int precision, const void* xn_blk_dev, const void* wqkv_blk_dev, const float* bias_dev,
                               void* out_blk_dev, int batch, int tokens, int d, int rows_alloc, void* stream);
int effocr_op_layernorm_blocked(int out_precision, const float* x_blk_dev, int64_t rows, int d, const float* gamma_dev,
                                const float* beta_dev, float eps, void* out_blk_dev, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* EFFOCR_HIP_H */
