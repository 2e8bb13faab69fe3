"""Oracle B: an independent implementation of the same architectures, from ``transformers``.

TEST INFRASTRUCTURE (see oracle/__init__.py).  timm cannot be installed here, but
``transformers`` ships ViT and ResNet implementations written by other people; loading the SAME
seeded weights into them (key-remapped) and getting the same embeddings as oracle A
(oracle/encoders_ref.py) is the strongest pin of the architecture restatement available offline
(SURVEY.md 8c).  Only used by tests/test_oracle_encoders.py and tests/golden/make_golden.py.
"""
import torch

from .encoders_ref import VIT_CFG, RESNET_CFG, strip_prefix


def hf_vit_forward(arch, sd, x):
    from transformers import ViTConfig, ViTModel
    sd = strip_prefix(sd)
    D, depth, heads = VIT_CFG[arch]
    cfg = ViTConfig(hidden_size=D, num_hidden_layers=depth, num_attention_heads=heads,
                    intermediate_size=4 * D, hidden_act="gelu", layer_norm_eps=1e-6,
                    image_size=x.shape[-1], patch_size=16, num_channels=3, qkv_bias=True,
                    hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    try:
        cfg._attn_implementation = "eager"
    except Exception:
        pass
    m = ViTModel(cfg, add_pooling_layer=False).eval()
    new = {
        "embeddings.cls_token": sd["cls_token"],
        "embeddings.position_embeddings": sd["pos_embed"],
        "embeddings.patch_embeddings.projection.weight": sd["patch_embed.proj.weight"],
        "embeddings.patch_embeddings.projection.bias": sd["patch_embed.proj.bias"],
        "layernorm.weight": sd["norm.weight"],
        "layernorm.bias": sd["norm.bias"],
    }
    for i in range(depth):
        p, q = f"blocks.{i}.", f"layers.{i}."  # transformers>=5 key names
        wq, wk, wv = sd[p + "attn.qkv.weight"].chunk(3, dim=0)
        bq, bk, bv = sd[p + "attn.qkv.bias"].chunk(3, dim=0)
        new[q + "attention.q_proj.weight"] = wq
        new[q + "attention.q_proj.bias"] = bq
        new[q + "attention.k_proj.weight"] = wk
        new[q + "attention.k_proj.bias"] = bk
        new[q + "attention.v_proj.weight"] = wv
        new[q + "attention.v_proj.bias"] = bv
        new[q + "attention.o_proj.weight"] = sd[p + "attn.proj.weight"]
        new[q + "attention.o_proj.bias"] = sd[p + "attn.proj.bias"]
        new[q + "layernorm_before.weight"] = sd[p + "norm1.weight"]
        new[q + "layernorm_before.bias"] = sd[p + "norm1.bias"]
        new[q + "layernorm_after.weight"] = sd[p + "norm2.weight"]
        new[q + "layernorm_after.bias"] = sd[p + "norm2.bias"]
        new[q + "mlp.fc1.weight"] = sd[p + "mlp.fc1.weight"]
        new[q + "mlp.fc1.bias"] = sd[p + "mlp.fc1.bias"]
        new[q + "mlp.fc2.weight"] = sd[p + "mlp.fc2.weight"]
        new[q + "mlp.fc2.bias"] = sd[p + "mlp.fc2.bias"]
    missing, unexpected = m.load_state_dict({k: v.clone() for k, v in new.items()}, strict=False)
    assert not unexpected, unexpected
    assert not [k for k in missing if "pooler" not in k], missing
    with torch.no_grad():
        return m(pixel_values=x.float()).last_hidden_state[:, 0]


def hf_resnet_forward(arch, sd, x):
    from transformers import ResNetConfig, ResNetModel
    sd = strip_prefix(sd)
    depths, widths = RESNET_CFG[arch]
    cfg = ResNetConfig(num_channels=3, embedding_size=64, hidden_sizes=list(widths),
                       depths=list(depths), layer_type="basic", hidden_act="relu",
                       downsample_in_first_stage=False)
    m = ResNetModel(cfg).eval()

    def bn(dst, src, out):
        for leaf in ("weight", "bias", "running_mean", "running_var"):
            out[f"{dst}.{leaf}"] = sd[f"{src}.{leaf}"]

    new = {"embedder.embedder.convolution.weight": sd["conv1.weight"]}
    bn("embedder.embedder.normalization", "bn1", new)
    for li, nb in enumerate(depths, start=1):
        for bi in range(nb):
            p = f"layer{li}.{bi}."
            q = f"encoder.stages.{li - 1}.layers.{bi}."
            new[q + "layer.0.convolution.weight"] = sd[p + "conv1.weight"]
            bn(q + "layer.0.normalization", p + "bn1", new)
            new[q + "layer.1.convolution.weight"] = sd[p + "conv2.weight"]
            bn(q + "layer.1.normalization", p + "bn2", new)
            if (p + "downsample.0.weight") in sd:
                new[q + "shortcut.convolution.weight"] = sd[p + "downsample.0.weight"]
                bn(q + "shortcut.normalization", p + "downsample.1", new)
    missing, unexpected = m.load_state_dict({k: v.clone() for k, v in new.items()}, strict=False)
    assert not unexpected, unexpected
    assert not [k for k in missing if "num_batches_tracked" not in k], missing
    with torch.no_grad():
        return m(pixel_values=x.float()).pooler_output.flatten(1)


def hf_encoder_forward(arch, sd, x):
    if arch in VIT_CFG:
        return hf_vit_forward(arch, sd, x)
    return hf_resnet_forward(arch, sd, x)
