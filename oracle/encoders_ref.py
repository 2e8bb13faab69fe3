"""Oracle A: plain-torch fp32 CPU restatement of the recognizer encoders.

TEST INFRASTRUCTURE (see oracle/__init__.py).  "parity unpinned" by the reference itself.

The reference builds its encoder with ``timm.create_model(name, num_classes=0)`` wrapped in
``AutoEncoder`` whose forward is ``self.net(x)`` (models/encoders.py:54-64) and calls it at
infer_effocr.py:314.  ``num_classes=0`` replaces the classifier with identity, so the output is
the pooled feature: global-average-pool for resnet18 ([B,512]) and the final-LayerNorm CLS token
for vit_{small,base}_patch16_224 ([B,384] / [B,768]).  timm is not installable here, so the
forward passes below restate the published architectures as pure functions of a state dict that
uses timm's parameter names (a real ``enc_best.pth`` has those names behind a ``net.`` prefix,
models/encoders.py:60,69).

Everything is fp32 on CPU, written with the most literal torch ops available (no fused
scaled_dot_product_attention, no folded BN) so that it is an independent check of the HIP path.
"""
import math
import torch
import torch.nn.functional as F

VIT_CFG = {
    # name: (embed_dim, depth, heads)
    "vit_small_patch16_224": (384, 12, 6),
    "vit_base_patch16_224": (768, 12, 12),
    # a 2-block / 128-dim miniature of the same architecture used by fast tests only
    "vit_tiny_test": (128, 2, 2),
}
RESNET_CFG = {"resnet18": ((2, 2, 2, 2), (64, 128, 256, 512))}
LN_EPS = 1e-6      # timm VisionTransformer uses partial(nn.LayerNorm, eps=1e-6)
BN_EPS = 1e-5      # nn.BatchNorm2d default


def strip_prefix(sd, prefix="net."):
    """models/encoders.py:60 stores the timm module as ``self.net`` -> keys ``net.<timm key>``."""
    if all(k.startswith(prefix) for k in sd):
        return {k[len(prefix):]: v for k, v in sd.items()}
    return dict(sd)


def embed_dim(arch):
    if arch in VIT_CFG:
        return VIT_CFG[arch][0]
    return RESNET_CFG[arch][1][-1]


# --------------------------------------------------------------------------- ViT
def vit_forward(arch, sd, x):
    """timm VisionTransformer.forward with num_classes=0, global_pool='token' (SURVEY.md 3.4).

    x: [B,3,H,W] fp32 with H,W multiples of 16 and (H/16)*(W/16)+1 == pos_embed tokens.
    """
    D, depth, heads = VIT_CFG[arch]
    hd = D // heads
    B = x.shape[0]
    # patch embed: conv 16x16 stride 16, flatten(2).transpose(1,2)
    t = F.conv2d(x, sd["patch_embed.proj.weight"], sd["patch_embed.proj.bias"], stride=16)
    t = t.flatten(2).transpose(1, 2)                       # [B, P, D]
    cls = sd["cls_token"].expand(B, -1, -1)
    t = torch.cat([cls, t], dim=1) + sd["pos_embed"]       # [B, 1+P, D]
    N = t.shape[1]
    for i in range(depth):
        p = f"blocks.{i}."
        h = F.layer_norm(t, (D,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], LN_EPS)
        qkv = F.linear(h, sd[p + "attn.qkv.weight"], sd[p + "attn.qkv.bias"])
        qkv = qkv.reshape(B, N, 3, heads, hd).permute(2, 0, 3, 1, 4)   # [3,B,H,N,hd]
        q, k, v = qkv[0], qkv[1], qkv[2]
        att = (q * (hd ** -0.5)) @ k.transpose(-2, -1)     # [B,H,N,N]
        att = att.softmax(dim=-1)
        o = (att @ v).transpose(1, 2).reshape(B, N, D)
        t = t + F.linear(o, sd[p + "attn.proj.weight"], sd[p + "attn.proj.bias"])
        h = F.layer_norm(t, (D,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], LN_EPS)
        h = F.linear(h, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"])
        h = F.gelu(h)                                      # exact erf GELU
        t = t + F.linear(h, sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])
    t = F.layer_norm(t, (D,), sd["norm.weight"], sd["norm.bias"], LN_EPS)
    return t[:, 0]


# --------------------------------------------------------------------------- resnet18
def _bn(x, sd, p):
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"],
                        sd[p + ".weight"], sd[p + ".bias"], training=False, eps=BN_EPS)


def resnet_forward(arch, sd, x):
    """timm ResNet(BasicBlock,[2,2,2,2]).forward with num_classes=0."""
    depths, widths = RESNET_CFG[arch]
    y = F.conv2d(x, sd["conv1.weight"], None, stride=2, padding=3)
    y = F.relu(_bn(y, sd, "bn1"))
    y = F.max_pool2d(y, kernel_size=3, stride=2, padding=1)
    for li, (nb, w) in enumerate(zip(depths, widths), start=1):
        for bi in range(nb):
            p = f"layer{li}.{bi}."
            stride = 2 if (bi == 0 and li > 1) else 1
            idt = y
            z = F.conv2d(y, sd[p + "conv1.weight"], None, stride=stride, padding=1)
            z = F.relu(_bn(z, sd, p + "bn1"))
            z = F.conv2d(z, sd[p + "conv2.weight"], None, stride=1, padding=1)
            z = _bn(z, sd, p + "bn2")
            if (p + "downsample.0.weight") in sd:
                idt = F.conv2d(y, sd[p + "downsample.0.weight"], None, stride=stride)
                idt = _bn(idt, sd, p + "downsample.1")
            y = F.relu(z + idt)
    return y.mean(dim=(2, 3))                              # global average pool + flatten


def encoder_forward(arch, sd, x):
    """x [B,3,H,W] fp32 -> pooled embedding [B,D] fp32 (NOT yet L2-normalised)."""
    sd = strip_prefix(sd)
    x = x.to(torch.float32)
    with torch.no_grad():
        if arch in VIT_CFG:
            return vit_forward(arch, sd, x)
        if arch in RESNET_CFG:
            return resnet_forward(arch, sd, x)
    raise NotImplementedError(arch)


def l2_normalize(emb):
    """torch.nn.functional.normalize(emb, p=2, dim=1) — infer_effocr.py:316.
    y = x / max(||x||_2, 1e-12)."""
    return emb / emb.norm(p=2, dim=1, keepdim=True).clamp_min(1e-12)
