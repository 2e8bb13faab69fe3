// C ABI of the YOLOv5 localizer engine (include/effocr_hip.h, "localizer" section): the network the reference runs
// through ONNXRuntime in onnx_engines/localizer_engine.py:14-66 (EffLocalizer, model_backend == 'yolo'), restated as a
// fixed sequence of gfx950 kernels.  The architecture is ultralytics YOLOv5 v6 "s" (models/yolov5s.yaml: depth 0.33,
// width 0.50): the reference ships no model definition (it loads an exported .onnx), so the layer table below follows
// the published yaml / common.py; parameter names are the ultralytics state-dict keys (model.<i>....).
//   Conv        = Conv2d(bias=False) + BatchNorm2d(eps 1e-3) + SiLU        -> BN folded on the host, SiLU in the epilogue
//   Bottleneck  = x + cv2(cv1(x)) (shortcut) | cv2(cv1(x))                  cv1 1x1, cv2 3x3, e = 1.0 inside C3
//   C3          = cv3(cat(m(cv1(x)), cv2(x)))                              -> both branches write slices of ONE buffer
//   SPPF        = cv2(cat(x', m(x'), m(m(x')), m(m(m(x')))))  x' = cv1(x)  -> the pool chain walks the slices of one buffer
//   Detect      = per level 1x1 conv (bias) -> sigmoid -> grid / anchor decode -> (bs, sum na*ny*nx, 5 + nc)
// Activations NHWC fp32; convolutions = resnet.hip's implicit GEMM on exact fp32 MFMA.
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
namespace {

struct LParam { std::string name; std::vector<int64_t> shape; int64_t numel; std::vector<float> data; bool set; };
struct Buf { int H, W, C; };                             // NHWC activation buffer (per image)
struct View { int buf, off, C; };                        // channel slice of a buffer
enum { OP_STEM = 0, OP_CONV = 1, OP_UP = 2, OP_POOL = 3, OP_DETECT = 4 };
struct Op {
  int type;
  View in, out, res;                                     // res.buf < 0: no residual
  int conv;                                              // index into convs
  int level;                                             // OP_DETECT
};
struct LConv { std::string w, bn, bias; int cin, cout, cout_pad, k, stride, pad, act; size_t w_off, b_off; int kpad; size_t w16_off; size_t wt_off;
               std::string w2, bn2; int cout1 = 0; };   // w2 / bn2: a second Conv block stacked behind the first cout1 output channels (C3's cv1 | cv2 in one launch)   // w16: the bf16 copy, rows padded to 64 k

}  // namespace
}  // namespace effocr

using namespace effocr;

struct effocr_localizer {
  int nc = 2, no = 7, in_h = 640, in_w = 640;
  std::vector<LParam> params;
  std::map<std::string, int> index;
  std::vector<Buf> bufs;
  std::vector<Op> ops;
  std::vector<LConv> convs;
  int stem_col = -1;                                     // buffer of the stem's im2col rows
  int det_raw[3] = {-1, -1, -1};
  float anchors[3][6];
  size_t wbytes = 0;
  const char* wdev = nullptr;
  int64_t npred = 0;
  int direct_stem = 1;                                   // 0: the stem through im2col + the 1x1 implicit GEMM (A/B)
  int bf16 = 0;                                          // 1: bf16-operand MFMAs for every convolution with an activation (Detect's 1x1 heads stay fp32)
};

namespace effocr {
namespace {

void ladd(effocr_localizer* e, const std::string& name, std::vector<int64_t> shape) {
  LParam p; p.name = name; p.shape = shape; p.numel = 1; for (auto v : shape) p.numel *= v; p.set = false;
  e->index[name] = (int)e->params.size();
  e->params.push_back(std::move(p));
}

struct Builder {
  effocr_localizer* e;
  int new_buf(int H, int W, int C) { e->bufs.push_back({H, W, C}); return (int)e->bufs.size() - 1; }
  View whole(int b) { return {b, 0, e->bufs[b].C}; }
  // ultralytics Conv block `name` (= "<name>.conv" + "<name>.bn"): in -> out slice
  void conv(const std::string& name, View in, View out, int k, int s, int act = 1, View res = {-1, 0, 0}) {
    LConv c; c.w = name + ".conv.weight"; c.bn = name + ".bn"; c.bias = "";
    c.cin = in.C; c.cout = out.C; c.cout_pad = out.C; c.k = k; c.stride = s; c.pad = k / 2; c.act = act; c.kpad = 0;
    ladd(e, c.w, {out.C, in.C, k, k});
    ladd(e, c.bn + ".weight", {out.C}); ladd(e, c.bn + ".bias", {out.C});
    ladd(e, c.bn + ".running_mean", {out.C}); ladd(e, c.bn + ".running_var", {out.C});
    e->convs.push_back(c);
    e->ops.push_back({OP_CONV, in, out, res, (int)e->convs.size() - 1, 0});
  }
  // two Conv blocks of the same geometry on the same input in ONE launch: output channels [0, c1) = block `n1`, [c1, out.C) = block `n2`
  void conv_pair(const std::string& n1, const std::string& n2, View in, View out, int c1, int k, int s) {
    LConv c; c.w = n1 + ".conv.weight"; c.bn = n1 + ".bn"; c.bias = ""; c.w2 = n2 + ".conv.weight"; c.bn2 = n2 + ".bn"; c.cout1 = c1;
    c.cin = in.C; c.cout = out.C; c.cout_pad = out.C; c.k = k; c.stride = s; c.pad = k / 2; c.act = 1; c.kpad = 0;
    for (int h = 0; h < 2; ++h) {
      const std::string& n = h ? n2 : n1;
      const int co = h ? out.C - c1 : c1;
      ladd(e, n + ".conv.weight", {co, in.C, k, k});
      ladd(e, n + ".bn.weight", {co}); ladd(e, n + ".bn.bias", {co}); ladd(e, n + ".bn.running_mean", {co}); ladd(e, n + ".bn.running_var", {co});
    }
    e->convs.push_back(c);
    e->ops.push_back({OP_CONV, in, out, {-1, 0, 0}, (int)e->convs.size() - 1, 0});
  }
  // C3(c1 -> c2, n bottlenecks, shortcut): returns nothing, writes `out`.  cv1 and cv2 (two 1x1 Conv blocks on the same input) run as ONE
  // convolution of 2 c_ output channels straight into the concat buffer [m(cv1(x)) | cv2(x)] (round 4: the input is read once, half the
  // launches / prologues, a wider channel tile); the bottleneck chain starts from slice 0 and its last block writes slice 0 back — in
  // place when n = 1: its 3x3 convolution reads the temporary t, and the residual element is read by the lane that overwrites it.
  void c3(const std::string& name, View in, View out, int n, bool shortcut) {
    const int c_ = out.C / 2, H = e->bufs[in.buf].H, W = e->bufs[in.buf].W;
    const int cat = new_buf(H, W, 2 * c_);
    conv_pair(name + ".cv1", name + ".cv2", in, whole(cat), c_, 1, 1);
    View cur = {cat, 0, c_};
    for (int i = 0; i < n; ++i) {
      const std::string m = name + ".m." + std::to_string(i);
      View t = {new_buf(H, W, c_), 0, c_};
      conv(m + ".cv1", cur, t, 1, 1);
      View nxt = (i == n - 1) ? View{cat, 0, c_} : View{new_buf(H, W, c_), 0, c_};
      conv(m + ".cv2", t, nxt, 3, 1, 1, shortcut ? cur : View{-1, 0, 0});
      cur = nxt;
    }
    conv(name + ".cv3", whole(cat), out, 1, 1);
  }
};

int down(int v) { return (v - 1) / 2 + 1; }              // conv k, stride 2, pad k/2: ceil(v / 2)

void build_yolov5s(effocr_localizer* e) {
  Builder b{e};
  const int H = e->in_h, W = e->in_w;
  const int H1 = down(H), W1 = down(W), H2 = down(H1), W2 = down(W1), H3 = down(H2), W3 = down(W2), H4 = down(H3), W4 = down(W3),
            H5 = down(H4), W5 = down(W4);
  // 0: Conv(3, 32, 6, 2, 2) — stem: im2col rows [.., 128] (108 taps + zeros) then a 1x1 implicit GEMM
  e->stem_col = b.new_buf(H1, W1, 128);
  const int l0 = b.new_buf(H1, W1, 32);
  {
    LConv c; c.w = "model.0.conv.weight"; c.bn = "model.0.bn"; c.bias = ""; c.cin = 3; c.cout = 32; c.cout_pad = 32; c.k = 6; c.stride = 2; c.pad = 2;
    c.act = 1; c.kpad = 128;
    ladd(e, c.w, {32, 3, 6, 6});
    ladd(e, c.bn + ".weight", {32}); ladd(e, c.bn + ".bias", {32}); ladd(e, c.bn + ".running_mean", {32}); ladd(e, c.bn + ".running_var", {32});
    e->convs.push_back(c);
    e->ops.push_back({OP_STEM, {-1, 0, 3}, b.whole(l0), {-1, 0, 0}, 0, 0});
  }
  const int l1 = b.new_buf(H2, W2, 64);   b.conv("model.1", b.whole(l0), b.whole(l1), 3, 2);
  const int l2 = b.new_buf(H2, W2, 64);   b.c3("model.2", b.whole(l1), b.whole(l2), 1, true);
  const int l3 = b.new_buf(H3, W3, 128);  b.conv("model.3", b.whole(l2), b.whole(l3), 3, 2);
  const int cat16 = b.new_buf(H3, W3, 256);               // [up(l14) | l4]
  b.c3("model.4", b.whole(l3), {cat16, 128, 128}, 2, true);
  const int l5 = b.new_buf(H4, W4, 256);  b.conv("model.5", {cat16, 128, 128}, b.whole(l5), 3, 2);
  const int cat12 = b.new_buf(H4, W4, 512);               // [up(l10) | l6]
  b.c3("model.6", b.whole(l5), {cat12, 256, 256}, 3, true);
  const int l7 = b.new_buf(H5, W5, 512);  b.conv("model.7", {cat12, 256, 256}, b.whole(l7), 3, 2);
  const int l8 = b.new_buf(H5, W5, 512);  b.c3("model.8", b.whole(l7), b.whole(l8), 1, true);
  // 9: SPPF(512, 512, 5)
  const int spp = b.new_buf(H5, W5, 1024);
  b.conv("model.9.cv1", b.whole(l8), {spp, 0, 256}, 1, 1);
  for (int i = 0; i < 3; ++i) e->ops.push_back({OP_POOL, {spp, 256 * i, 256}, {spp, 256 * (i + 1), 256}, {-1, 0, 0}, -1, 0});
  const int l9 = b.new_buf(H5, W5, 512);  b.conv("model.9.cv2", b.whole(spp), b.whole(l9), 1, 1);
  // head
  const int cat22 = b.new_buf(H5, W5, 512);               // [l21 | l10]
  b.conv("model.10", b.whole(l9), {cat22, 256, 256}, 1, 1);
  e->ops.push_back({OP_UP, {cat22, 256, 256}, {cat12, 0, 256}, {-1, 0, 0}, -1, 0});                  // 11, 12
  const int l13 = b.new_buf(H4, W4, 256); b.c3("model.13", b.whole(cat12), b.whole(l13), 1, false);
  const int cat19 = b.new_buf(H4, W4, 256);               // [l18 | l14]
  b.conv("model.14", b.whole(l13), {cat19, 128, 128}, 1, 1);
  e->ops.push_back({OP_UP, {cat19, 128, 128}, {cat16, 0, 128}, {-1, 0, 0}, -1, 0});                  // 15, 16
  const int l17 = b.new_buf(H3, W3, 128); b.c3("model.17", b.whole(cat16), b.whole(l17), 1, false);
  b.conv("model.18", b.whole(l17), {cat19, 0, 128}, 3, 2);                                           // 18, 19
  const int l20 = b.new_buf(H4, W4, 256); b.c3("model.20", b.whole(cat19), b.whole(l20), 1, false);
  b.conv("model.21", b.whole(l20), {cat22, 0, 256}, 3, 2);                                           // 21, 22
  const int l23 = b.new_buf(H5, W5, 512); b.c3("model.23", b.whole(cat22), b.whole(l23), 1, false);
  // 24: Detect — 1x1 conv with bias (no BN, no activation); output channels padded to a multiple of 4
  const int feats[3] = {l17, l20, l23};
  const int nout = 3 * e->no, npad = (nout + 3) / 4 * 4;
  e->npred = 0;
  for (int l = 0; l < 3; ++l) {
    const Buf f = e->bufs[feats[l]];
    e->det_raw[l] = b.new_buf(f.H, f.W, npad);
    LConv c; c.w = "model.24.m." + std::to_string(l) + ".weight"; c.bn = ""; c.bias = "model.24.m." + std::to_string(l) + ".bias";
    c.cin = f.C; c.cout = nout; c.cout_pad = npad; c.k = 1; c.stride = 1; c.pad = 0; c.act = 0; c.kpad = 0;
    ladd(e, c.w, {nout, f.C, 1, 1}); ladd(e, c.bias, {nout});
    e->convs.push_back(c);
    e->ops.push_back({OP_CONV, b.whole(feats[l]), {e->det_raw[l], 0, npad}, {-1, 0, 0}, (int)e->convs.size() - 1, 0});
    e->ops.push_back({OP_DETECT, {e->det_raw[l], 0, npad}, {-1, 0, 0}, {-1, 0, 0}, -1, l});
    e->npred += (int64_t)3 * f.H * f.W;
  }
  ladd(e, "model.24.anchors", {3, 3, 2});                 // in units of the level's stride (ultralytics buffer)
  size_t off = 0;
  for (auto& c : e->convs) {
    const size_t K = c.kpad ? (size_t)c.kpad : (size_t)c.k * c.k * c.cin;
    c.w_off = off; off = align_up(off + (size_t)c.cout_pad * K * 4, 256);
    c.b_off = off; off = align_up(off + (size_t)c.cout_pad * 4, 256);
    c.w16_off = off; off = align_up(off + (size_t)c.cout_pad * ((K + 63) / 64 * 64) * 2, 256);
    c.wt_off = 0;
    if (c.kpad) { c.wt_off = off; off = align_up(off + (size_t)c.k * c.k * c.cin * c.cout_pad * 4, 256); }   // stem: [tap][channel] transpose (scalar-weight kernel)
  }
  e->wbytes = off;
}

const std::vector<float>& LP(const effocr_localizer* e, const std::string& n) { return e->params[e->index.at(n)].data; }

void pack_localizer(effocr_localizer* e, std::vector<char>& blob) {
  for (const LConv& c : e->convs) {
    const int K = c.k * c.k * c.cin, Kp = c.kpad ? c.kpad : K;
    float* wd = reinterpret_cast<float*>(blob.data() + c.w_off);
    float* bd = reinterpret_cast<float*>(blob.data() + c.b_off);
    for (int co = 0; co < c.cout_pad; ++co) {
      for (int kk = 0; kk < Kp; ++kk) wd[(size_t)co * Kp + kk] = 0.f;
      bd[co] = 0.f;
      if (co >= c.cout) continue;
      const bool second = !c.w2.empty() && co >= c.cout1;   // stacked pair: rows [cout1, cout) come from the second block
      const auto& w = LP(e, second ? c.w2 : c.w);
      const std::string& bn = second ? c.bn2 : c.bn;
      const int cs = second ? co - c.cout1 : co;            // the row inside its own block
      double sc = 1.0;
      if (!bn.empty()) {                                 // BatchNorm2d(eps = 1e-3: ultralytics initialize_weights) folded in
        const double g = LP(e, bn + ".weight")[cs], bt = LP(e, bn + ".bias")[cs], mu = LP(e, bn + ".running_mean")[cs],
                     var = LP(e, bn + ".running_var")[cs];
        sc = g / sqrt(var + 1e-3);
        bd[co] = (float)(bt - mu * sc);
      } else {
        bd[co] = LP(e, c.bias)[cs];
      }
      for (int ky = 0; ky < c.k; ++ky)
        for (int kx = 0; kx < c.k; ++kx)
          for (int ci = 0; ci < c.cin; ++ci)
            wd[(size_t)co * Kp + (ky * c.k + kx) * c.cin + ci] = (float)((double)w[(((size_t)cs * c.cin + ci) * c.k + ky) * c.k + kx] * sc);
    }
    if (c.kpad) {
      float* wt = reinterpret_cast<float*>(blob.data() + c.wt_off);
      for (int kk = 0; kk < K; ++kk)
        for (int co = 0; co < c.cout_pad; ++co) wt[(size_t)kk * c.cout_pad + co] = wd[(size_t)co * Kp + kk];
    }
    // bf16 copy of the folded weights (round to nearest even), rows zero-padded to a multiple of 64 k
    const int Kp64 = (Kp + 63) / 64 * 64;
    uint16_t* w16 = reinterpret_cast<uint16_t*>(blob.data() + c.w16_off);
    for (int co = 0; co < c.cout_pad; ++co)
      for (int kk = 0; kk < Kp64; ++kk) {
        uint32_t u = 0;
        if (kk < Kp) {
          const float f = wd[(size_t)co * Kp + kk];
          memcpy(&u, &f, 4);
          u = (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;    // finite values only (folded weights)
        }
        w16[(size_t)co * Kp64 + kk] = (uint16_t)u;
      }
  }
  const auto& an = LP(e, "model.24.anchors");
  const float strides[3] = {8.f, 16.f, 32.f};
  for (int l = 0; l < 3; ++l)
    for (int j = 0; j < 6; ++j) e->anchors[l][j] = an[l * 6 + j] * strides[l];      // anchor_grid = anchors * stride (pixels)
}

constexpr size_t LOC_SPLIT_BYTES = (size_t)16 << 20;   // split-K scratch of conv2d_nhwc (the deep layers at small batches: few tiles, long K)
struct LWs { std::vector<size_t> off; size_t split, total; };
LWs localizer_ws(const effocr_localizer* e, int B) {
  LWs w; size_t off = 0;
  for (size_t i = 0; i < e->bufs.size(); ++i) {
    const Buf& b = e->bufs[i];
    w.off.push_back(off);
    if ((int)i == e->stem_col && e->direct_stem) continue;   // the im2col rows of the stem (840 MB at 16 x 640 x 640) exist only on the A/B path
    off = align_up(off + (size_t)B * b.H * b.W * b.C * 4, 256);
  }
  w.split = off; off += LOC_SPLIT_BYTES;
  w.total = off;
  return w;
}

hipStream_t LS(void* s) { return static_cast<hipStream_t>(s); }

}  // namespace
}  // namespace effocr

extern "C" {

int effocr_localizer_create(const char* arch, int num_classes, int in_h, int in_w, effocr_localizer_t** out) {
  if (!arch || !out) return fail(EFFOCR_EINVAL, "localizer_create: NULL argument");
  if (std::string(arch) != "yolov5s") return fail(EFFOCR_EUNSUPPORTED, std::string("localizer_create: unsupported architecture '") + arch + "' (yolov5s)");
  if (num_classes < 1 || num_classes > 80) return fail(EFFOCR_EINVAL, "localizer_create: num_classes must be in 1..80");
  if (in_h < 32 || in_w < 32 || in_h % 32 || in_w % 32) return fail(EFFOCR_EINVAL, "localizer_create: input size must be a positive multiple of 32 (stride)");
  std::unique_ptr<effocr_localizer> e(new effocr_localizer());
  e->nc = num_classes; e->no = num_classes + 5; e->in_h = in_h; e->in_w = in_w;
  build_yolov5s(e.get());
  *out = e.release();
  return EFFOCR_OK;
}
void effocr_localizer_destroy(effocr_localizer_t* loc) { delete loc; }
int effocr_localizer_num_params(const effocr_localizer_t* loc) { return loc ? (int)loc->params.size() : 0; }
const char* effocr_localizer_param_name(const effocr_localizer_t* loc, int i) {
  if (!loc || i < 0 || i >= (int)loc->params.size()) return nullptr;
  return loc->params[i].name.c_str();
}
int64_t effocr_localizer_param_numel(const effocr_localizer_t* loc, int i) {
  if (!loc || i < 0 || i >= (int)loc->params.size()) return -1;
  return loc->params[i].numel;
}
int effocr_localizer_set_param(effocr_localizer_t* loc, const char* name, const float* host, int64_t numel) {
  if (!loc || !name || !host) return fail(EFFOCR_EINVAL, "localizer_set_param: NULL argument");
  auto it = loc->index.find(name);
  if (it == loc->index.end()) return fail(EFFOCR_EINVAL, std::string("localizer_set_param: unknown parameter '") + name + "'");
  LParam& p = loc->params[it->second];
  if (p.numel != numel) return fail(EFFOCR_EINVAL, std::string("localizer_set_param: '") + name + "' expects " + std::to_string(p.numel) + " elements, got " + std::to_string(numel));
  p.data.assign(host, host + numel);
  p.set = true;
  return EFFOCR_OK;
}
size_t effocr_localizer_weights_bytes(const effocr_localizer_t* loc) { return loc ? loc->wbytes : 0; }
int effocr_localizer_upload(effocr_localizer_t* loc, void* weights_dev, size_t bytes) {
  if (!loc || !weights_dev) return fail(EFFOCR_EINVAL, "localizer_upload: NULL argument");
  if (bytes < loc->wbytes) return fail(EFFOCR_EWORKSPACE, "localizer_upload: weight buffer too small");
  for (const LParam& p : loc->params)
    if (!p.set) return fail(EFFOCR_ESTATE, "localizer_upload: parameter '" + p.name + "' was never set");
  std::vector<char> blob(loc->wbytes, 0);
  pack_localizer(loc, blob);
  const hipError_t er = hipMemcpy(weights_dev, blob.data(), loc->wbytes, hipMemcpyHostToDevice);
  if (er != hipSuccess) return fail(EFFOCR_EHIP, std::string("localizer_upload: hipMemcpy: ") + hipGetErrorString(er));
  loc->wdev = static_cast<const char*>(weights_dev);
  return EFFOCR_OK;
}
int effocr_localizer_set_option(effocr_localizer_t* loc, const char* name, int value) {
  if (!loc || !name) return fail(EFFOCR_EINVAL, "localizer_set_option: NULL argument");
  if (std::string(name) == "bf16_operands") { loc->bf16 = value != 0; return EFFOCR_OK; }
  if (std::string(name) == "direct_stem") { loc->direct_stem = value != 0; return EFFOCR_OK; }
  return fail(EFFOCR_EINVAL, std::string("localizer_set_option: unknown option '") + name + "'");
}
int64_t effocr_localizer_num_predictions(const effocr_localizer_t* loc) { return loc ? loc->npred : 0; }
int effocr_localizer_num_outputs(const effocr_localizer_t* loc) { return loc ? loc->no : 0; }
size_t effocr_localizer_workspace_bytes(const effocr_localizer_t* loc, int batch) {
  if (!loc || batch <= 0) return 0;
  return localizer_ws(loc, batch).total;
}

int effocr_localizer_forward(effocr_localizer_t* loc, const float* x_dev, int batch, float* pred_dev, void* workspace_dev, size_t workspace_bytes,
                             void* stream) {
  if (!loc) return fail(EFFOCR_EINVAL, "localizer_forward: NULL localizer");
  if (batch < 0) return fail(EFFOCR_EINVAL, "localizer_forward: negative batch");
  if (batch == 0) return EFFOCR_OK;
  if (!x_dev || !pred_dev || !workspace_dev) return fail(EFFOCR_EINVAL, "localizer_forward: NULL device pointer");
  if (!loc->wdev) return fail(EFFOCR_ESTATE, "localizer_forward: weights were not uploaded");
  const LWs w = localizer_ws(loc, batch);
  if (workspace_bytes < w.total) return fail(EFFOCR_EWORKSPACE, "localizer_forward: workspace too small");
  char* ws = static_cast<char*>(workspace_dev);
  hipStream_t s = LS(stream);
  auto P = [&](int b) { return reinterpret_cast<float*>(ws + w.off[b]); };
  int64_t row0[3]; int64_t acc = 0;
  for (int l = 0; l < 3; ++l) { row0[l] = acc; acc += (int64_t)3 * loc->bufs[loc->det_raw[l]].H * loc->bufs[loc->det_raw[l]].W; }
  int rc;
  for (const Op& op : loc->ops) {
    switch (op.type) {
      case OP_STEM: {
        const LConv& c = loc->convs[op.conv];
        const Buf o = loc->bufs[op.out.buf];
        if (loc->direct_stem && c.k == 6 && c.stride == 2 && c.pad == 2 && c.cin == 3 && c.cout_pad == 32 && c.kpad >= 108) {
          // (fp32 in both precision modes: 108 taps per pixel are VALU work, the operand rounding of the bf16 mode starts at layer 1)
          if ((rc = stem6x6s2_nchw(x_dev, reinterpret_cast<const float*>(loc->wdev + c.w_off), c.kpad, reinterpret_cast<const float*>(loc->wdev + c.wt_off),
                                   reinterpret_cast<const float*>(loc->wdev + c.b_off),
                                   P(op.out.buf), batch, loc->in_h, loc->in_w, o.H, o.W, o.C, op.out.off, c.act, s))) return rc;
          break;
        }
        if ((rc = im2col_nchw(x_dev, P(loc->stem_col), batch, 3, loc->in_h, loc->in_w, c.k, c.k, c.stride, c.pad, o.H, o.W, c.kpad, s))) return rc;
        ConvArgs a{};
        a.in = P(loc->stem_col); a.w = reinterpret_cast<const float*>(loc->wdev + c.w_off); a.bias = reinterpret_cast<const float*>(loc->wdev + c.b_off);
        a.out = P(op.out.buf); a.B = batch * o.H * o.W; a.H = 1; a.W = 1; a.Cin = c.kpad; a.Cout = c.cout_pad; a.KH = 1; a.KW = 1; a.stride = 1; a.pad = 0;
        a.OH = 1; a.OW = 1; a.silu = c.act; a.out_ld = o.C; a.out_off = op.out.off;
        if (loc->bf16 && c.act) a.w16 = loc->wdev + c.w16_off;
        a.partial = reinterpret_cast<float*>(ws + w.split); a.partial_bytes = LOC_SPLIT_BYTES;
        if ((rc = conv2d_nhwc(a, s))) return rc;
        break;
      }
      case OP_CONV: {
        const LConv& c = loc->convs[op.conv];
        const Buf i = loc->bufs[op.in.buf], o = loc->bufs[op.out.buf];
        ConvArgs a{};
        a.in = P(op.in.buf); a.in_ld = i.C; a.in_off = op.in.off;
        a.w = reinterpret_cast<const float*>(loc->wdev + c.w_off); a.bias = reinterpret_cast<const float*>(loc->wdev + c.b_off);
        a.out = P(op.out.buf); a.out_ld = o.C; a.out_off = op.out.off;
        if (op.res.buf >= 0) { a.resid = P(op.res.buf); a.res_ld = loc->bufs[op.res.buf].C; a.res_off = op.res.off; }
        a.B = batch; a.H = i.H; a.W = i.W; a.Cin = c.cin; a.Cout = c.cout_pad; a.KH = c.k; a.KW = c.k; a.stride = c.stride; a.pad = c.pad;
        a.OH = o.H; a.OW = o.W; a.silu = c.act;
        if (loc->bf16 && c.act) a.w16 = loc->wdev + c.w16_off;
        a.partial = reinterpret_cast<float*>(ws + w.split); a.partial_bytes = LOC_SPLIT_BYTES;
        if ((rc = conv2d_nhwc(a, s))) return rc;
        break;
      }
      case OP_UP: {
        const Buf i = loc->bufs[op.in.buf], o = loc->bufs[op.out.buf];
        if ((rc = upsample2x_nhwc(P(op.in.buf), i.C, op.in.off, P(op.out.buf), o.C, op.out.off, batch, i.H, i.W, op.in.C, s))) return rc;
        break;
      }
      case OP_POOL: {
        const Buf i = loc->bufs[op.in.buf];
        if ((rc = maxpool5_nhwc(P(op.in.buf), i.C, op.in.off, P(op.out.buf), i.C, op.out.off, batch, i.H, i.W, op.in.C, s))) return rc;
        break;
      }
      case OP_DETECT: {
        const Buf r = loc->bufs[op.in.buf];
        const float strides[3] = {8.f, 16.f, 32.f};
        if ((rc = yolo_decode(P(op.in.buf), r.C, pred_dev, batch, r.H, r.W, 3, loc->no, strides[op.level], loc->anchors[op.level], loc->npred,
                              row0[op.level], s))) return rc;
        break;
      }
    }
  }
  return EFFOCR_OK;
}

int effocr_letterbox(const uint8_t* image_dev, int height, int width, int64_t row_stride, int bgr, int out_h, int out_w, int new_h, int new_w,
                     int top, int left, float* out_dev, void* stream) {
  if (!image_dev || !out_dev) return fail(EFFOCR_EINVAL, "letterbox: NULL device pointer");
  if (row_stride < (int64_t)3 * width) return fail(EFFOCR_EINVAL, "letterbox: row stride smaller than 3 * width");
  return letterbox_u8(image_dev, height, width, row_stride, bgr, out_h, out_w, new_h, new_w, top, left, 114.0f, out_dev, LS(stream));
}

size_t effocr_nms_workspace_bytes(int n, int max_nms) { return nms_workspace_bytes(n, max_nms); }
size_t effocr_nms_batch_workspace_bytes(int n, int max_det, int max_nms) { return nms_greedy_applies(n, max_det, max_nms) ? 0 : nms_workspace_bytes(n, max_nms); }

int effocr_nms(const float* pred_dev, int n, int num_classes, float conf_thres, float iou_thres, int max_det, int max_nms, float max_wh, int agnostic,
               float* out_dev, int* count_dev, void* workspace_dev, size_t workspace_bytes, void* stream) {
  if (n > 0 && (!pred_dev || !workspace_dev)) return fail(EFFOCR_EINVAL, "nms: NULL device pointer");
  if (!out_dev || !count_dev) return fail(EFFOCR_EINVAL, "nms: NULL output pointer");
  return nms_yolo(pred_dev, n, num_classes, conf_thres, iou_thres, max_det, max_nms, max_wh, agnostic, out_dev, count_dev, workspace_dev, workspace_bytes,
                  LS(stream));
}

int effocr_nms_batch(const float* pred_dev, int batch, int n, int num_classes, float conf_thres, float iou_thres, int max_det, int max_nms, float max_wh,
                     int agnostic, float* out_dev, int* count_dev, void* workspace_dev, size_t workspace_bytes, void* stream) {
  if (batch > 0 && n > 0 && !pred_dev) return fail(EFFOCR_EINVAL, "nms: NULL device pointer");
  if (batch > 0 && (!out_dev || !count_dev)) return fail(EFFOCR_EINVAL, "nms: NULL output pointer");
  if (batch > 0 && n > 0 && !nms_greedy_applies(n, max_det, max_nms) && !workspace_dev) return fail(EFFOCR_EINVAL, "nms: NULL workspace");
  return nms_yolo_batch(pred_dev, batch, n, num_classes, conf_thres, iou_thres, max_det, max_nms, max_wh, agnostic, out_dev, count_dev, workspace_dev,
                        workspace_bytes, LS(stream));
}

}  // extern "C"
