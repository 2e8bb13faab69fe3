"""Encoder parameter tables, seeded random initialisation and checkpoint I/O.

The reference obtains its encoder from ``timm.create_model(name, num_classes=0)`` and saves /
loads it as a torch state dict whose keys carry a ``net.`` prefix (models/encoders.py:56-70,
train_effocr_recognizer.py:65-72).  This module knows the parameter names and shapes of the three
architectures BASELINE.json names (resnet18, vit_small_patch16_224, vit_base_patch16_224) so that
a real ``enc_best.pth`` drops in, and it produces the seeded random-init weights the benchmark
and the tests use (there is no network for checkpoints).
"""
from collections import OrderedDict
import math
import torch

VIT_CFG = {
    # name: (embed_dim, depth, heads, mlp_ratio)
    "vit_small_patch16_224": (384, 12, 6, 4),
    "vit_base_patch16_224": (768, 12, 12, 4),
    "vit_tiny_test": (128, 2, 2, 4),          # miniature used only by fast tests
}
RESNET_CFG = {"resnet18": ((2, 2, 2, 2), (64, 128, 256, 512))}
PATCH = 16


def is_vit(arch):
    return arch in VIT_CFG


def embed_dim(arch):
    if arch in VIT_CFG:
        return VIT_CFG[arch][0]
    if arch in RESNET_CFG:
        return RESNET_CFG[arch][1][-1]
    raise NotImplementedError(f"unsupported encoder architecture {arch!r}")


def param_shapes(arch, img_size=224):
    """Ordered {timm key: shape} for ``arch`` (buffers such as BN running stats included)."""
    s = OrderedDict()
    if arch in VIT_CFG:
        D, depth, heads, r = VIT_CFG[arch]
        ntok = (img_size // PATCH) ** 2 + 1
        s["cls_token"] = (1, 1, D)
        s["pos_embed"] = (1, ntok, D)
        s["patch_embed.proj.weight"] = (D, 3, PATCH, PATCH)
        s["patch_embed.proj.bias"] = (D,)
        for i in range(depth):
            p = f"blocks.{i}."
            s[p + "norm1.weight"] = (D,)
            s[p + "norm1.bias"] = (D,)
            s[p + "attn.qkv.weight"] = (3 * D, D)
            s[p + "attn.qkv.bias"] = (3 * D,)
            s[p + "attn.proj.weight"] = (D, D)
            s[p + "attn.proj.bias"] = (D,)
            s[p + "norm2.weight"] = (D,)
            s[p + "norm2.bias"] = (D,)
            s[p + "mlp.fc1.weight"] = (r * D, D)
            s[p + "mlp.fc1.bias"] = (r * D,)
            s[p + "mlp.fc2.weight"] = (D, r * D)
            s[p + "mlp.fc2.bias"] = (D,)
        s["norm.weight"] = (D,)
        s["norm.bias"] = (D,)
        return s
    if arch in RESNET_CFG:
        depths, widths = RESNET_CFG[arch]

        def bn(p, c):
            s[p + ".weight"] = (c,)
            s[p + ".bias"] = (c,)
            s[p + ".running_mean"] = (c,)
            s[p + ".running_var"] = (c,)

        s["conv1.weight"] = (64, 3, 7, 7)
        bn("bn1", 64)
        cin = 64
        for li, (nb, w) in enumerate(zip(depths, widths), start=1):
            for bi in range(nb):
                p = f"layer{li}.{bi}."
                s[p + "conv1.weight"] = (w, cin, 3, 3)
                bn(p + "bn1", w)
                s[p + "conv2.weight"] = (w, w, 3, 3)
                bn(p + "bn2", w)
                if bi == 0 and li > 1:
                    s[p + "downsample.0.weight"] = (w, cin, 1, 1)
                    bn(p + "downsample.1", w)
                cin = w
        return s
    raise NotImplementedError(f"unsupported encoder architecture {arch!r}")


def init_state_dict(arch, seed=0, img_size=224, scale="unit"):
    """Seeded random-init fp32 CPU state dict with timm key names.

    scale="timm": trunc_normal(0.02) linears / kaiming convs, LN and BN at identity — what
    timm's own initialisers give.  scale="unit" (default): fan-in-scaled weights, non-trivial
    LN/BN affine terms and running statistics, so that attention is far from uniform and every
    term of every kernel (biases, gamma/beta, BN folding) is exercised by the parity tests.
    The generator is the CPU Philox stream, identical on every machine with this torch build.
    """
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    sd = OrderedDict()

    def randn(shape, std):
        return torch.randn(shape, generator=g, dtype=torch.float32) * std

    def uniform(shape, lo, hi):
        return torch.rand(shape, generator=g, dtype=torch.float32) * (hi - lo) + lo

    for k, shp in param_shapes(arch, img_size).items():
        leaf = k.rsplit(".", 1)[-1]
        if scale == "timm":
            if k in ("cls_token",):
                v = randn(shp, 1e-6)
            elif k == "pos_embed":
                v = randn(shp, 0.02).clamp_(-0.04, 0.04)
            elif leaf == "running_mean":
                v = torch.zeros(shp)
            elif leaf == "running_var":
                v = torch.ones(shp)
            elif len(shp) == 1:
                v = torch.ones(shp) if leaf == "weight" else torch.zeros(shp)
            elif len(shp) == 4 and arch in RESNET_CFG:
                v = randn(shp, math.sqrt(2.0 / (shp[0] * shp[2] * shp[3])))
            else:
                v = randn(shp, 0.02).clamp_(-0.04, 0.04)
        else:
            if k in ("cls_token", "pos_embed"):
                v = randn(shp, 0.5)
            elif leaf == "running_mean":
                v = randn(shp, 0.1)
            elif leaf == "running_var":
                v = uniform(shp, 0.5, 1.5)
            elif len(shp) == 1:
                is_norm = ("norm" in k) or (".bn" in k) or k.startswith("bn") or ("downsample.1" in k)
                if leaf == "weight" and is_norm:
                    v = uniform(shp, 0.5, 1.5)
                else:
                    v = randn(shp, 0.1)
            else:
                fan_in = 1
                for d in shp[1:]:
                    fan_in *= d
                gain = math.sqrt(2.0) if arch in RESNET_CFG else 1.0
                v = randn(shp, gain / math.sqrt(fan_in))
        sd[k] = v.contiguous()
    return sd


def strip_prefix(sd, prefix="net."):
    """models/encoders.py:60 keeps the timm module as ``self.net`` -> keys ``net.<timm key>``."""
    keys = list(sd.keys())
    if keys and all(k.startswith(prefix) for k in keys):
        return OrderedDict((k[len(prefix):], v) for k, v in sd.items())
    return OrderedDict(sd)


def load_checkpoint(path):
    """Read encoder weights the way ``AutoEncoder.load`` does (models/encoders.py:66-70), but with
    ``map_location='cpu'`` so that it works on any box; also accepts ``.safetensors``."""
    path = str(path)
    if path.endswith(".safetensors"):
        from safetensors.torch import load_file
        sd = load_file(path)
    else:
        sd = torch.load(path, map_location="cpu", weights_only=True)
        if isinstance(sd, dict) and "state_dict" in sd and isinstance(sd["state_dict"], dict):
            sd = sd["state_dict"]
    return strip_prefix(sd)


def save_checkpoint(sd, path, prefix="net."):
    """Write a state dict with the reference's ``net.`` key prefix (train_effocr_recognizer.py:65-72)."""
    out = OrderedDict((prefix + k, v.detach().cpu().contiguous()) for k, v in strip_prefix(sd).items())
    path = str(path)
    if path.endswith(".safetensors"):
        from safetensors.torch import save_file
        save_file(out, path)
    else:
        torch.save(out, path)


def infer_arch(sd):
    """Guess the architecture of a checkpoint from its parameter shapes."""
    sd = strip_prefix(sd)
    if "conv1.weight" in sd and "layer4.1.conv2.weight" in sd:
        return "resnet18"
    if "pos_embed" in sd:
        D = sd["pos_embed"].shape[-1]
        depth = 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("blocks."))
        for name, (d, dep, _, _) in VIT_CFG.items():
            if d == D and dep == depth:
                return name
    raise ValueError("cannot infer encoder architecture from checkpoint keys")


def check_state_dict(arch, sd, img_size=224):
    """Raise ValueError listing missing / mis-shaped parameters (num_batches_tracked etc. ignored)."""
    want = param_shapes(arch, img_size)
    bad = []
    for k, shp in want.items():
        if k not in sd:
            bad.append(f"missing {k}")
        elif tuple(sd[k].shape) != tuple(shp):
            bad.append(f"{k}: shape {tuple(sd[k].shape)} != {tuple(shp)}")
    if bad:
        raise ValueError(f"state dict does not match {arch}: " + "; ".join(bad[:8]) +
                         (f" (+{len(bad) - 8} more)" if len(bad) > 8 else ""))
