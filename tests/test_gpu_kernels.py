"""-m gpu: every HIP operator of the encoder against a CPU fp64/fp32 restatement, called through the
C ABI (effocr_op_* entry points of include/effocr_hip.h)."""
import math

import numpy as np
import pytest
import torch

from effocr_amd import _lib

pytestmark = pytest.mark.gpu

TDT = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}


def _stream(dev):
    return _lib.current_stream(dev)


def op_linear(L, dev, prec, epi, x, w, bias, resid=None):
    M, K = x.shape
    N = w.shape[0]
    xd, wd, bd = x.to(dev).contiguous(), w.to(dev).contiguous(), bias.to(dev).contiguous()
    if epi == "bias_resid":
        out = resid.to(dev).clone().contiguous()
        rd = out
    else:
        out = torch.empty((M, N), dtype=TDT[prec], device=dev)
        rd = None
    _lib.check(L.effocr_op_linear(_lib.PREC[prec], _lib.EPI[epi], _lib.ptr(xd), _lib.ptr(wd), _lib.ptr(bd),
                                  _lib.ptr(rd), _lib.ptr(out), M, N, K, _stream(dev)), "op_linear")
    torch.cuda.synchronize()
    return out.float().cpu()


@pytest.mark.parametrize("prec", ["bf16", "fp16", "fp32"])
@pytest.mark.parametrize("epi", ["bias", "bias_gelu", "bias_resid"])
@pytest.mark.parametrize("shape", [(197 * 3, 384, 384), (100, 128, 1536), (256, 1152, 384), (1, 128, 128)])
def test_linear(ab_lib, dev, prec, epi, shape):
    hip_lib = ab_lib                                    # A/B build: this entry point reaches kernels outside the product library
    M, N, K = shape
    g = torch.Generator().manual_seed(M * 7 + N + K)
    x = torch.randn(M, K, generator=g).to(TDT[prec])
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(TDT[prec])
    bias = torch.randn(N, generator=g)
    resid = torch.randn(M, N, generator=g)
    got = op_linear(hip_lib, dev, prec, epi, x, w, bias, resid)
    ref = x.double() @ w.double().T + bias.double()
    if epi == "bias_gelu":
        ref = torch.nn.functional.gelu(ref)
    if epi == "bias_resid":
        ref = ref + resid.double()
    # operands are exactly representable, so only the fp32 accumulation order and (for 16-bit
    # outputs) the final rounding differ
    tol = {"bf16": 8e-3, "fp16": 1e-3, "fp32": 2e-5}[prec] if epi != "bias_resid" else 2e-5
    err = (got.double() - ref).abs().max().item()
    scale = ref.abs().max().item()
    assert err <= tol * scale, f"{prec} {epi} {shape}: err {err:.3e} scale {scale:.3e}"


@pytest.mark.parametrize("prec", ["bf16", "fp16", "fp32"])
@pytest.mark.parametrize("D,rows", [(384, 1000), (768, 33), (128, 17), (384, 1)])
def test_layernorm(hip_lib, dev, prec, D, rows):
    g = torch.Generator().manual_seed(D + rows)
    x = torch.randn(rows, D, generator=g) * 3 + 0.5
    gamma, beta = torch.rand(D, generator=g) + 0.5, torch.randn(D, generator=g)
    out = torch.empty((rows, D), dtype=TDT[prec], device=dev)
    xd, gd, bd = x.to(dev), gamma.to(dev), beta.to(dev)
    _lib.check(hip_lib.effocr_op_layernorm(_lib.PREC[prec], _lib.ptr(xd), rows, D, _lib.ptr(gd), _lib.ptr(bd),
                                           1e-6, _lib.ptr(out), _stream(dev)), "op_layernorm")
    torch.cuda.synchronize()
    ref = torch.nn.functional.layer_norm(x.double(), (D,), gamma.double(), beta.double(), 1e-6)
    tol = {"bf16": 8e-3, "fp16": 1e-3, "fp32": 1e-5}[prec]
    err = (out.float().cpu().double() - ref).abs().max().item()
    assert err <= tol * ref.abs().max().item()


@pytest.mark.parametrize("prec", ["bf16", "fp16"])
@pytest.mark.parametrize("epi", ["bias", "bias_gelu", "bias_resid"])
@pytest.mark.parametrize("shape", [(197 * 5, 1152, 384), (300, 384, 384), (129, 1536, 384), (77, 384, 128), (128, 512, 128)])
def test_ln_linear_fused(ab_lib, dev, prec, epi, shape):
    hip_lib = ab_lib                                    # A/B build: this entry point reaches kernels outside the product library
    """Row-panel kernel with the LayerNorm fused into the operand load vs LN -> round -> linear."""
    M, N, K = shape
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g) * 2 + 0.3
    gamma, beta = torch.rand(K, generator=g) + 0.5, torch.randn(K, generator=g) * 0.2
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(TDT[prec])
    bias = torch.randn(N, generator=g)
    resid = torch.randn(M, N, generator=g)
    xd, gd, bd, wd, bsd = x.to(dev), gamma.to(dev), beta.to(dev), w.to(dev), bias.to(dev)
    if epi == "bias_resid":
        out = resid.to(dev).clone()
        rd = out
    else:
        out = torch.full((M, N), float("nan"), dtype=TDT[prec], device=dev)
        rd = None
    _lib.check(hip_lib.effocr_op_ln_linear(_lib.PREC[prec], _lib.EPI[epi], _lib.ptr(xd), _lib.ptr(gd), _lib.ptr(bd), 1e-6,
                                           _lib.ptr(wd), _lib.ptr(bsd), _lib.ptr(rd), _lib.ptr(out), M, N, K, _stream(dev)),
               "op_ln_linear")
    torch.cuda.synchronize()
    xn = torch.nn.functional.layer_norm(x.double(), (K,), gamma.double(), beta.double(), 1e-6).to(TDT[prec]).double()
    ref = xn @ w.double().T + bias.double()
    if epi == "bias_gelu":
        ref = torch.nn.functional.gelu(ref)
    if epi == "bias_resid":
        ref = ref + resid.double()
    got = out.float().cpu().double()
    assert torch.isfinite(got).all()
    # the fused LN rounds its fp32 result to the operand type exactly like the reference expression,
    # but fp32-vs-fp64 LN arithmetic can flip a rounding: allow a few operand ulps
    tol = {"bf16": 1.2e-2, "fp16": 1.5e-3}[prec]
    err = (got - ref).abs().max().item()
    assert err <= tol * ref.abs().max().item(), f"{prec} {epi} {shape}: err {err:.3e} scale {ref.abs().max().item():.3e}"


def attention_ref(qkv, B, T, heads):
    D = heads * 64
    q, k, v = qkv.double().reshape(B, T, 3, heads, 64).permute(2, 0, 3, 1, 4)
    att = ((q * 0.125) @ k.transpose(-2, -1)).softmax(-1)
    return (att @ v).transpose(1, 2).reshape(B * T, D)


@pytest.mark.parametrize("prec", ["bf16", "fp16", "fp32"])
@pytest.mark.parametrize("B,T,heads", [(3, 197, 6), (2, 17, 2), (1, 197, 12), (5, 50, 2), (2, 224, 2), (1, 1, 2)])
def test_attention(hip_lib, dev, prec, B, T, heads):
    D = heads * 64
    g = torch.Generator().manual_seed(B * 1000 + T)
    qkv = (torch.randn(B * T, 3 * D, generator=g) * 1.5).to(TDT[prec])
    qd = qkv.to(dev)
    out = torch.full((B * T, D), float("nan"), dtype=TDT[prec], device=dev)
    _lib.check(hip_lib.effocr_op_attention(_lib.PREC[prec], _lib.ptr(qd), _lib.ptr(out), B, T, heads, _stream(dev)),
               "op_attention")
    torch.cuda.synchronize()
    ref = attention_ref(qkv, B, T, heads)
    got = out.float().cpu().double()
    assert torch.isfinite(got).all()
    tol = {"bf16": 1.5e-2, "fp16": 2e-3, "fp32": 1e-5}[prec]      # P is rounded to the operand type
    err = (got - ref).abs().max().item()
    assert err <= tol * ref.abs().max().item(), f"err {err:.3e} vs scale {ref.abs().max().item():.3e}"


def test_unsupported_shapes_fail_loudly(ab_lib, dev):
    hip_lib = ab_lib                                    # A/B build: this entry point reaches kernels outside the product library
    x = torch.zeros(4, 100, device=dev, dtype=torch.bfloat16)
    w = torch.zeros(128, 100, device=dev, dtype=torch.bfloat16)
    b = torch.zeros(128, device=dev)
    o = torch.zeros(4, 128, device=dev, dtype=torch.bfloat16)
    rc = hip_lib.effocr_op_linear(0, 0, _lib.ptr(x), _lib.ptr(w), _lib.ptr(b), None, _lib.ptr(o), 4, 128, 100, None)
    assert rc == -2 and b"multiple" in hip_lib.effocr_last_error()
    rc = hip_lib.effocr_op_attention(0, _lib.ptr(x), _lib.ptr(o), 1, 100, 2, None)
    assert rc == -2


# ---------------------------------------------------------------- fragment-blocked fast-path operators (gemm3, blocked LN)
def to_blocked(t, rows_alloc):
    """[rows, cols] tensor -> flat buffer in the cell layout [rows_alloc/32][cols/ch][32][ch] (ch = 16 bytes)."""
    rows, cols = t.shape
    ch = 16 // t.element_size()
    buf = torch.zeros((rows_alloc, cols), dtype=t.dtype)
    buf[:rows] = t
    return buf.view(rows_alloc // 32, 32, cols // ch, ch).permute(0, 2, 1, 3).contiguous().view(-1)


def from_blocked(flat, rows, cols, rows_alloc):
    ch = 16 // flat.element_size()
    return flat.view(rows_alloc // 32, cols // ch, 32, ch).permute(0, 2, 1, 3).contiguous().view(rows_alloc, cols)[:rows]


@pytest.mark.parametrize("prec", ["bf16", "fp16"])
@pytest.mark.parametrize("epi", ["bias", "bias_gelu", "bias_resid"])
@pytest.mark.parametrize("shape", [(197 * 40, 384, 1536), (300, 192, 256), (1000, 768, 768), (70000, 256, 384), (1, 2304, 768), (33, 3072, 768), (70000, 768, 256), (140000, 192, 256)])
def test_linear_blocked_gemm3(hip_lib, dev, prec, epi, shape):
    """gemm3 through effocr_op_linear_blocked: both tile widths (192 / 256), the small-tile tail launch
    ((70000, 256): 274 token tiles -> one full round + tail), ragged last row blocks, several tiles per persistent workgroup ((70000, 768): 3; (140000, 192): 2), K = 256 (8 stages: the shortest tile the continuous ring runs: 3 + 1 + 4)."""
    M, N, K = shape
    g = torch.Generator().manual_seed(M + 3 * N + K)
    x = torch.randn(M, K, generator=g).to(TDT[prec])
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(TDT[prec])
    bias = torch.randn(N, generator=g)
    resid = torch.randn(M, N, generator=g)
    ra = (M + 127) // 128 * 128
    xd, wd, bd = to_blocked(x, ra).to(dev), to_blocked(w, N).to(dev), bias.to(dev)
    if epi == "bias_resid":
        out = to_blocked(resid, ra).to(dev)
        rd = out
    else:
        out = torch.zeros(ra * N, dtype=TDT[prec], device=dev)
        rd = None
    _lib.check(hip_lib.effocr_op_linear_blocked(_lib.PREC[prec], _lib.EPI[epi], _lib.ptr(xd), _lib.ptr(wd), _lib.ptr(bd),
                                                _lib.ptr(rd), _lib.ptr(out), M, N, K, ra, _stream(dev)), "op_linear_blocked")
    torch.cuda.synchronize()
    got = from_blocked(out.cpu(), M, N, ra).double()
    ref = x.double() @ w.double().T + bias.double()
    if epi == "bias_gelu":
        ref = torch.nn.functional.gelu(ref)
    if epi == "bias_resid":
        ref = ref + resid.double()
    tol = {"bf16": 8e-3, "fp16": 1e-3}[prec] if epi != "bias_resid" else 2e-5
    err, scale = (got - ref).abs().max().item(), ref.abs().max().item()
    assert err <= tol * scale, f"{prec} {epi} {shape}: err {err:.3e} scale {scale:.3e}"
    if epi != "bias_resid":                                 # rows past M (padding of the last blocks) stay untouched
        pad = from_blocked(out.cpu(), ra, N, ra)[M:]
        assert not pad.any()


def test_linear_blocked_argument_checks(hip_lib, dev):
    z = torch.zeros(1 << 16, dtype=torch.bfloat16, device=dev)
    f = torch.zeros(1 << 12, dtype=torch.float32, device=dev)
    call = lambda m, n, k, ra: hip_lib.effocr_op_linear_blocked(0, 0, _lib.ptr(z), _lib.ptr(z), _lib.ptr(f), None, _lib.ptr(z), m, n, k, ra, _stream(dev))
    assert call(32, 128, 128, 32) == -2          # N neither % 192 nor % 256
    assert call(32, 192, 320, 32) == -2          # K % 128
    assert call(32, 192, 128, 32) == -2          # K < 256 (8 ring stages: the shortest tile the continuous ring runs)
    assert call(32, 192, 256, 16) == -1          # rows_alloc < m
    assert call(0, 192, 256, 0) == 0
    assert hip_lib.effocr_op_linear_blocked(2, 0, _lib.ptr(z), _lib.ptr(z), _lib.ptr(f), None, _lib.ptr(z), 32, 192, 256, 32, _stream(dev)) == -2   # fp32


@pytest.mark.parametrize("prec", ["bf16", "fp16"])
@pytest.mark.parametrize("rows,D", [(197 * 7, 768), (50, 384), (33, 128), (1, 768)])
def test_layernorm_blocked(hip_lib, dev, prec, rows, D):
    g = torch.Generator().manual_seed(rows + D)
    x = torch.randn(rows, D, generator=g) * 3 + torch.randn(rows, 1, generator=g)
    gamma, beta = torch.randn(D, generator=g), torch.randn(D, generator=g)
    ra = (rows + 31) // 32 * 32
    xd = to_blocked(x, ra).to(dev)
    out = torch.zeros(ra * D, dtype=TDT[prec], device=dev)
    gd, bd = gamma.to(dev), beta.to(dev)                     # keep alive: ptr() of a temporary would dangle
    _lib.check(hip_lib.effocr_op_layernorm_blocked(_lib.PREC[prec], _lib.ptr(xd), rows, D, _lib.ptr(gd), _lib.ptr(bd),
                                                   1e-6, _lib.ptr(out), _stream(dev)), "op_layernorm_blocked")
    torch.cuda.synchronize()
    got = from_blocked(out.cpu(), rows, D, ra).double()
    ref = torch.nn.functional.layer_norm(x.double(), (D,), gamma.double(), beta.double(), 1e-6)
    tol = {"bf16": 4e-3, "fp16": 5e-4}[prec]
    assert ((got - ref).abs().max() / ref.abs().max()).item() <= tol


def perm16_columns(w):
    """fc2 weight copy for the fused MLP: inside every group of 16 hidden indices the order is
    [0-3, 8-11 | 4-7, 12-15] (what the swapped-MFMA C-layout hands over as a B-operand; include/effocr_hip.h)."""
    base = torch.tensor([0, 1, 2, 3, 8, 9, 10, 11, 4, 5, 6, 7, 12, 13, 14, 15])
    idx = (torch.arange(w.shape[1] // 16)[:, None] * 16 + base[None, :]).reshape(-1)
    return w[:, idx].contiguous()


@pytest.mark.parametrize("prec", ["bf16", "fp16"])
@pytest.mark.parametrize("scratch", [False, True])
@pytest.mark.parametrize("shape", [(197 * 5, 384, 1536), (100, 128, 512), (1, 384, 1536), (128 * 3, 128, 512), (40000, 384, 1536), (128 * 140, 384, 1536)])
def test_mlp_fused_blocked(ab_lib, dev, prec, shape, scratch):
    hip_lib = ab_lib                                    # A/B build: this entry point reaches kernels outside the product library
    """mlp.hip: x + fc2(gelu(fc1(LN(x)))) in one kernel vs an fp64 restatement with the same operand rounding points
    (LN output and GELU output rounded to the operand type, as the unfused path does)."""
    M, D, H = shape
    g = torch.Generator().manual_seed(M + D)
    x = torch.randn(M, D, generator=g) * 2 + 0.3 * torch.randn(M, 1, generator=g)
    gamma, beta = 1 + 0.2 * torch.randn(D, generator=g), 0.1 * torch.randn(D, generator=g)
    w1 = (torch.randn(H, D, generator=g) / math.sqrt(D)).to(TDT[prec])
    w2 = (torch.randn(D, H, generator=g) / math.sqrt(H)).to(TDT[prec])
    b1, b2 = 0.5 * torch.randn(H, generator=g), 0.5 * torch.randn(D, generator=g)
    ra = (M + 127) // 128 * 128
    xd = to_blocked(x, ra).to(dev)
    w1d, w2d = to_blocked(w1, H).to(dev), to_blocked(perm32_rows(perm16_columns(w2)), D).to(dev)
    gd, bd, b1d, b2d, b2pd = gamma.to(dev), beta.to(dev), b1.to(dev), b2.to(dev), perm32_rows(b2).to(dev)
    # with scratch the panels of the last partially filled round of CUs are split over the hidden dimension:
    # M = 985 -> 8 panels x 4 parts; 40000 -> 256 + 57 x 4; 17920 -> 140 panels x 2 parts (140 * 4 > 256 CUs)
    sc = torch.empty(64 << 20, dtype=torch.uint8, device=dev) if scratch else None
    _lib.check(hip_lib.effocr_op_mlp_blocked(_lib.PREC[prec], _lib.ptr(xd), _lib.ptr(gd), _lib.ptr(bd), 1e-6, _lib.ptr(w1d), _lib.ptr(b1d),
                                             _lib.ptr(w2d), _lib.ptr(b2pd), _lib.ptr(b2d), M, D, H, ra, _lib.ptr(sc), sc.numel() if scratch else 0,
                                             _stream(dev)), "op_mlp_blocked")
    torch.cuda.synchronize()
    got = from_blocked(xd.cpu(), M, D, ra).double()
    xn = torch.nn.functional.layer_norm(x.double(), (D,), gamma.double(), beta.double(), 1e-6).to(TDT[prec]).double()
    hid = torch.nn.functional.gelu(xn @ w1.double().T + b1.double()).to(TDT[prec]).double()
    delta = hid @ w2.double().T + b2.double()
    ref = x.double() + delta
    tol = {"bf16": 1e-2, "fp16": 1.5e-3}[prec]
    err, scale = (got - ref).abs().max().item(), delta.abs().max().item()
    assert err <= tol * scale, f"{prec} {shape}: err {err:.3e} vs delta scale {scale:.3e}"
    if ra > M:                                             # padding rows untouched
        assert torch.equal(from_blocked(xd.cpu(), ra, D, ra)[M:], torch.zeros(ra - M, D))


def perm32_rows(t):
    """P32 of include/effocr_hip.h along dim 0: position p of every block of 32 holds source index
    8*(2*(r>>3) + hh) + (r&7), hh = (p>>2)&1, r = (p&3) + 4*(p>>3)."""
    p = torch.arange(32)
    hh, r = (p >> 2) & 1, (p & 3) + 4 * (p >> 3)
    src = 8 * (2 * (r >> 3) + hh) + (r & 7)
    idx = (torch.arange(t.shape[0] // 32)[:, None] * 32 + src[None, :]).reshape(-1)
    return t[idx].contiguous()


@pytest.mark.parametrize("prec", ["bf16", "fp16"])
@pytest.mark.parametrize("scratch", [False, True])
@pytest.mark.parametrize("shape", [(197 * 5, 384, 1536), (100, 128, 512), (40000, 384, 1536)])
def test_proj_mlp_fused_blocked(hip_lib, dev, prec, shape, scratch):
    """mlp.hip with the attention projection in front: y = x + a.Wp^T + bp; x <- y + fc2(gelu(fc1(LN(y))))."""
    M, D, H = shape
    g = torch.Generator().manual_seed(M + D + 1)
    x = torch.randn(M, D, generator=g) * 2 + 0.3 * torch.randn(M, 1, generator=g)
    av = torch.randn(M, D, generator=g).to(TDT[prec])
    wp = (torch.randn(D, D, generator=g) / math.sqrt(D)).to(TDT[prec])
    bp = 0.5 * torch.randn(D, generator=g)
    gamma, beta = 1 + 0.2 * torch.randn(D, generator=g), 0.1 * torch.randn(D, generator=g)
    w1 = (torch.randn(H, D, generator=g) / math.sqrt(D)).to(TDT[prec])
    w2 = (torch.randn(D, H, generator=g) / math.sqrt(H)).to(TDT[prec])
    b1, b2 = 0.5 * torch.randn(H, generator=g), 0.5 * torch.randn(D, generator=g)
    ra = (M + 127) // 128 * 128
    xd, ad = to_blocked(x, ra).to(dev), to_blocked(av, ra).to(dev)
    wpd, bpd = to_blocked(perm32_rows(wp), D).to(dev), perm32_rows(bp).to(dev)
    w1d, w2d = to_blocked(w1, H).to(dev), to_blocked(perm32_rows(perm16_columns(w2)), D).to(dev)
    gd, btd, b1d, b2pd, b2d = gamma.to(dev), beta.to(dev), b1.to(dev), perm32_rows(b2).to(dev), b2.to(dev)
    sc = torch.empty(64 << 20, dtype=torch.uint8, device=dev) if scratch else None
    _lib.check(hip_lib.effocr_op_proj_mlp_blocked(_lib.PREC[prec], _lib.ptr(xd), _lib.ptr(ad), _lib.ptr(wpd), _lib.ptr(bpd), _lib.ptr(gd), _lib.ptr(btd),
                                                  1e-6, _lib.ptr(w1d), _lib.ptr(b1d), _lib.ptr(w2d), _lib.ptr(b2pd), _lib.ptr(b2d), M, D, H, ra,
                                                  _lib.ptr(sc), sc.numel() if scratch else 0, _stream(dev)), "op_proj_mlp_blocked")
    torch.cuda.synchronize()
    got = from_blocked(xd.cpu(), M, D, ra).double()
    y = x.double() + av.double() @ wp.double().T + bp.double()
    xn = torch.nn.functional.layer_norm(y, (D,), gamma.double(), beta.double(), 1e-6).to(TDT[prec]).double()
    hid = torch.nn.functional.gelu(xn @ w1.double().T + b1.double()).to(TDT[prec]).double()
    delta = hid @ w2.double().T + b2.double()
    ref = y + delta
    tol = {"bf16": 1e-2, "fp16": 1.5e-3}[prec]
    err, scale = (got - ref).abs().max().item(), delta.abs().max().item()
    assert err <= tol * scale, f"{prec} {shape}: err {err:.3e} vs delta scale {scale:.3e}"


@pytest.mark.parametrize("prec", ["bf16", "fp16"])
@pytest.mark.parametrize("B,T,D", [(3, 197, 384), (1, 197, 384), (5, 50, 128), (2, 17, 128), (2, 224, 128), (4, 64, 384), (2, 193, 384), (1, 1, 128), (300, 197, 384), (1024, 197, 384)])
def test_qkv_attn_fused_blocked(hip_lib, dev, prec, B, T, D):
    """qkvattn.hip: attn.qkv + softmax(q k^T / 8) v in one kernel (persistent workgroups, one image at a time: 300 / 1024
    images = 2 / 4 images per workgroup on 256 CUs) vs an fp64 restatement with the same operand rounding points
    (q / k / v and P rounded to the operand type)."""
    heads = D // 64
    M = B * T
    g = torch.Generator().manual_seed(B * 1000 + T + D)
    xn = torch.randn(M, D, generator=g).to(TDT[prec])
    w = (torch.randn(3 * D, D, generator=g) / math.sqrt(D)).to(TDT[prec])
    bias = 0.5 * torch.randn(3 * D, generator=g)
    ra = (M + 127) // 128 * 128
    wv, bvp = w.clone(), bias.clone()                      # the kernel's weight copy: v rows (and bias) P32-permuted
    wv[2 * D:], bvp[2 * D:] = perm32_rows(w[2 * D:]), perm32_rows(bias[2 * D:])
    xd, wd, bd = to_blocked(xn, ra).to(dev), to_blocked(wv, 3 * D).to(dev), bvp.to(dev)
    out = torch.zeros(ra * D, dtype=TDT[prec], device=dev)
    _lib.check(hip_lib.effocr_op_qkv_attn_blocked(_lib.PREC[prec], _lib.ptr(xd), _lib.ptr(wd), _lib.ptr(bd), _lib.ptr(out), B, T, D, ra,
                                                  _stream(dev)), "op_qkv_attn_blocked")
    torch.cuda.synchronize()
    got = from_blocked(out.cpu(), M, D, ra).double()
    if B > 300:                                            # full batch: check a sample of images (the fp64 reference of 1024 is slow)
        sel = torch.tensor([0, 1, 255, 256, 511, 700, 1022, 1023])
        rows = (sel[:, None] * T + torch.arange(T)[None, :]).reshape(-1)
        xn, got, Bc = xn[rows], got[rows], len(sel)
    else:
        Bc = B
    qkv = (xn.double() @ w.double().T + bias.double()).to(TDT[prec])
    ref = attention_ref(qkv, Bc, T, heads)
    assert torch.isfinite(got).all()
    tol = {"bf16": 1.5e-2, "fp16": 2e-3}[prec]    # q, k, v and P are rounded to the operand type
    err, scale = (got - ref).abs().max().item(), ref.abs().max().item()
    assert err <= tol * scale, f"{prec} {(B, T, D)}: err {err:.3e} vs scale {scale:.3e}"
    if ra > M:                                             # padding rows untouched
        assert not from_blocked(out.cpu(), ra, D, ra)[M:].any()


def test_qkv_attn_fused_argument_checks(hip_lib, dev):
    z = torch.zeros(1 << 16, dtype=torch.bfloat16, device=dev)
    f = torch.zeros(1 << 16, dtype=torch.float32, device=dev)
    call = lambda prec, b, t, d, ra: hip_lib.effocr_op_qkv_attn_blocked(prec, _lib.ptr(z), _lib.ptr(z), _lib.ptr(f), _lib.ptr(z), b, t, d, ra, _stream(dev))
    assert call(0, 1, 100, 128, 128) == -2        # 65..192 tokens: no kernel
    assert call(0, 1, 50, 256, 128) == -2         # embed dim
    assert call(2, 1, 50, 128, 128) == -2         # fp32
    assert call(0, 2, 50, 128, 64) == -1          # rows_alloc < batch * tokens
    assert call(0, 0, 50, 128, 0) == 0
    assert hip_lib.effocr_op_qkv_attn_blocked(0, None, _lib.ptr(z), _lib.ptr(f), _lib.ptr(z), 1, 50, 128, 128, None) == -1


@pytest.mark.parametrize("prec", ["bf16", "fp16"])
@pytest.mark.parametrize("scratch", [False, True])
@pytest.mark.parametrize("shape", [(197 * 5, 384, 1536), (100, 128, 512), (40000, 384, 1536)])
def test_mlp_fused_second_output(ab_lib, dev, prec, shape, scratch):
    hip_lib = ab_lib                                    # A/B build: this entry point reaches kernels outside the product library
    """mlp.hip with the next block's norm1 as a second output (xn = LN(x_new), 16-bit blocked): both the in-kernel epilogue
    (whole panels) and the blocked-LayerNorm launch over the split tail panels (scratch given: 40000 rows = 256 + 57 panels)."""
    M, D, H = shape
    g = torch.Generator().manual_seed(M + D + 7)
    x = torch.randn(M, D, generator=g) * 2 + 0.3 * torch.randn(M, 1, generator=g)
    gamma, beta = 1 + 0.2 * torch.randn(D, generator=g), 0.1 * torch.randn(D, generator=g)
    gn, bn = 1 + 0.3 * torch.randn(D, generator=g), 0.2 * torch.randn(D, generator=g)
    w1 = (torch.randn(H, D, generator=g) / math.sqrt(D)).to(TDT[prec])
    w2 = (torch.randn(D, H, generator=g) / math.sqrt(H)).to(TDT[prec])
    b1, b2 = 0.5 * torch.randn(H, generator=g), 0.5 * torch.randn(D, generator=g)
    ra = (M + 127) // 128 * 128
    xd = to_blocked(x, ra).to(dev)
    w1d, w2d = to_blocked(w1, H).to(dev), to_blocked(perm32_rows(perm16_columns(w2)), D).to(dev)
    gd, bd, b1d, b2d, gnd, bnd, b2pd = gamma.to(dev), beta.to(dev), b1.to(dev), b2.to(dev), gn.to(dev), bn.to(dev), perm32_rows(b2).to(dev)
    xn = torch.zeros(ra * D, dtype=TDT[prec], device=dev)
    sc = torch.empty(64 << 20, dtype=torch.uint8, device=dev) if scratch else None
    _lib.check(hip_lib.effocr_op_mlp_ln_blocked(_lib.PREC[prec], _lib.ptr(xd), _lib.ptr(gd), _lib.ptr(bd), 1e-6, _lib.ptr(w1d), _lib.ptr(b1d),
                                                _lib.ptr(w2d), _lib.ptr(b2pd), _lib.ptr(b2d), _lib.ptr(gnd), _lib.ptr(bnd), _lib.ptr(xn), M, D, H, ra,
                                                _lib.ptr(sc), sc.numel() if scratch else 0, _stream(dev)), "op_mlp_ln_blocked")
    torch.cuda.synchronize()
    got_x = from_blocked(xd.cpu(), M, D, ra)
    got_n = from_blocked(xn.cpu(), M, D, ra).double()
    # the second output is the LayerNorm of the kernel's OWN first output (fp32), rounded to the operand type
    ref_n = torch.nn.functional.layer_norm(got_x.double(), (D,), gn.double(), bn.double(), 1e-6)
    tol = {"bf16": 4e-3, "fp16": 5e-4}[prec]
    assert ((got_n - ref_n).abs().max() / ref_n.abs().max()).item() <= tol
    xnr = torch.nn.functional.layer_norm(x.double(), (D,), gamma.double(), beta.double(), 1e-6).to(TDT[prec]).double()
    hid = torch.nn.functional.gelu(xnr @ w1.double().T + b1.double()).to(TDT[prec]).double()
    delta = hid @ w2.double().T + b2.double()
    tolx = {"bf16": 1e-2, "fp16": 1.5e-3}[prec]
    assert (got_x.double() - (x.double() + delta)).abs().max().item() <= tolx * delta.abs().max().item()
    if ra > M:
        assert not from_blocked(xn.cpu(), ra, D, ra)[M:].any()
