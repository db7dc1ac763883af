"""CPU oracle of the CLIP image embedding (TEST INFRASTRUCTURE ONLY: imported by tests/, never by the product path).

Functional fp32 PyTorch restatement of FrozenCLIPImageEmbedder (ldm/modules/encoders/modules.py:343-382) as
`SyncMultiviewDiffusion.prepare` calls it (morphable_diffusion.py:487-488):

  preprocess (:363-371)   kornia.geometry.resize(x, (224, 224), 'bicubic', align_corners=True, antialias=False)
                          [kornia's resize is torch.nn.functional.interpolate with the same arguments], (x+1)/2,
                          normalise with CLIP's mean / std
  encode_image            `clip.load(...)`'s VisionTransformer.forward (openai/CLIP clip/model.py, the release
                          requirements.txt installs from git, un-pinned): conv1 (patch embedding, no bias), class token,
                          positional embedding, ln_pre, `layers` ResidualAttentionBlocks (x += MHA(ln_1 x);
                          x += c_proj(QuickGELU(c_fc(ln_2 x))), QuickGELU(v) = v * sigmoid(1.702 v)), ln_post on the
                          class token, @ proj
  encode (:381-382)       unsqueeze(1) -> [B, 1, 768]

Neither `clip` nor `kornia` is importable here and neither is vendored in the reference tree, so this oracle is pinned
against an INDEPENDENT implementation of the same published model: transformers' CLIPVisionModelWithProjection on
seeded weights (tools/make_goldens.py --only-clip maps the openai key names onto the transformers ones). That pins the
transformer arithmetic; the kornia -> F.interpolate equivalence is taken from kornia's source as published, not
verified here (stated in DESIGN.md).
"""
import torch
import torch.nn.functional as F

P = "clip_image_encoder.model.visual."
MEAN = (0.48145466, 0.4578275, 0.40821073)
STD = (0.26862954, 0.26130258, 0.27577711)


def preprocess(x, size=224):
    """modules.py:363-371; x: [B, 3, H, W] in [-1, 1]."""
    x = F.interpolate(x, size=(size, size), mode="bicubic", align_corners=True, antialias=False)
    x = (x + 1.0) / 2.0
    mean = torch.tensor(MEAN, dtype=x.dtype).view(1, 3, 1, 1)
    std = torch.tensor(STD, dtype=x.dtype).view(1, 3, 1, 1)
    return (x - mean) / std


def _ln(W, p, x):
    return F.layer_norm(x, (x.shape[-1],), W[p + ".weight"], W[p + ".bias"], 1e-5)


def _attention(W, p, x, heads):
    """nn.MultiheadAttention(width, heads) self-attention, batch-first restatement."""
    B, T, C = x.shape
    d = C // heads
    qkv = F.linear(x, W[p + ".in_proj_weight"], W[p + ".in_proj_bias"])
    q, k, v = qkv.view(B, T, 3, heads, d).permute(2, 0, 3, 1, 4)  # each [B, heads, T, d]
    a = torch.softmax((q * d ** -0.5) @ k.transpose(-1, -2), dim=-1)
    o = (a @ v).permute(0, 2, 1, 3).reshape(B, T, C)
    return F.linear(o, W[p + ".out_proj.weight"], W[p + ".out_proj.bias"])


def vision_transformer(W, cfg, x):
    """x: preprocessed [B, 3, image, image] -> [B, embed]."""
    B = x.shape[0]
    h = F.conv2d(x, W[P + "conv1.weight"], None, stride=cfg.patch)            # [B, width, g, g]
    h = h.reshape(B, cfg.width, -1).permute(0, 2, 1)                           # [B, g*g, width]
    cls = W[P + "class_embedding"].view(1, 1, -1).expand(B, 1, -1)
    h = torch.cat([cls, h], 1) + W[P + "positional_embedding"]
    h = _ln(W, P + "ln_pre", h)
    for i in range(cfg.layers):
        p = f"{P}transformer.resblocks.{i}"
        h = h + _attention(W, p + ".attn", _ln(W, p + ".ln_1", h), cfg.heads)
        m = F.linear(_ln(W, p + ".ln_2", h), W[p + ".mlp.c_fc.weight"], W[p + ".mlp.c_fc.bias"])
        m = m * torch.sigmoid(1.702 * m)
        h = h + F.linear(m, W[p + ".mlp.c_proj.weight"], W[p + ".mlp.c_proj.bias"])
    return _ln(W, P + "ln_post", h[:, 0, :]) @ W[P + "proj"]


def encode(W, cfg, x):
    """FrozenCLIPImageEmbedder.encode: x [B, 3, H, W] in [-1, 1] -> [B, 1, embed]."""
    return vision_transformer(W, cfg, preprocess(x, cfg.image)).unsqueeze(1)
