"""CPU oracle of the first-stage decoder (TEST INFRASTRUCTURE ONLY: imported by tests/, never by the product path).

Functional fp32 PyTorch restatement of AutoencoderKL.decode (ldm/models/autoencoder.py:330-333) =
post_quant_conv + Decoder.forward (ldm/modules/diffusionmodules/model.py:535-568), with ResnetBlock (:121-141),
AttnBlock (:178-202), Upsample (:53-57), Normalize = GroupNorm(32, eps 1e-6) (:38-39), nonlinearity = swish (:33-35).
Pinned: tests/golden/vae_*.npz are outputs of the reference module itself (tools/make_goldens.py --only-vae).
"""
import torch
import torch.nn.functional as F

P = "first_stage_model."


def _gn(W, p, x):
    return F.group_norm(x, 32, W[p + ".weight"], W[p + ".bias"], 1e-6)


def _swish(x):
    return x * torch.sigmoid(x)


def _conv(W, p, x, pad):
    return F.conv2d(x, W[p + ".weight"], W[p + ".bias"], padding=pad)


def resnet_block(W, p, x):
    """model.py:121-141 with temb = None."""
    h = _conv(W, p + ".conv1", _swish(_gn(W, p + ".norm1", x)), 1)
    h = _conv(W, p + ".conv2", _swish(_gn(W, p + ".norm2", h)), 1)
    if (p + ".nin_shortcut.weight") in W:
        x = _conv(W, p + ".nin_shortcut", x, 0)
    return x + h


def attn_block(W, p, x):
    """model.py:178-202: single-head attention over the h*w positions, scale c^-0.5."""
    h_ = _gn(W, p + ".norm", x)
    q, k, v = (_conv(W, p + "." + n, h_, 0) for n in ("q", "k", "v"))
    b, c, h, w = q.shape
    q = q.reshape(b, c, h * w).permute(0, 2, 1)
    k = k.reshape(b, c, h * w)
    w_ = torch.softmax(torch.bmm(q, k) * (int(c) ** (-0.5)), dim=2)
    v = v.reshape(b, c, h * w)
    h_ = torch.bmm(v, w_.permute(0, 2, 1)).reshape(b, c, h, w)
    return x + _conv(W, p + ".proj_out", h_, 0)


def decode(W, cfg, z):
    """AutoencoderKL.decode(z): z [B, embed_dim, h, w] -> [B, out_ch, 8h, 8w]."""
    d = P + "decoder."
    h = _conv(W, P + "post_quant_conv", z, 0)
    h = _conv(W, d + "conv_in", h, 1)
    h = resnet_block(W, d + "mid.block_1", h)
    h = attn_block(W, d + "mid.attn_1", h)
    h = resnet_block(W, d + "mid.block_2", h)
    for lvl in reversed(range(len(cfg.ch_mult))):
        for i in range(cfg.num_res_blocks + 1):
            h = resnet_block(W, d + f"up.{lvl}.block.{i}", h)
        if lvl != 0:
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")
            h = _conv(W, d + f"up.{lvl}.upsample.conv", h, 1)
    h = _swish(_gn(W, d + "norm_out", h))
    return _conv(W, d + "conv_out", h, 1)


def encode_moments(W, cfg, x):
    """AutoencoderKL.encode(x).parameters (autoencoder.py:324-328): Encoder.forward (model.py:434-459) then quant_conv.
    x [B,3,H,W] -> moments [B, 2*embed_dim, H/8, W/8] = (mean | logvar); the posterior's sample()/mode() are host code."""
    e = P + "encoder."
    h = _conv(W, e + "conv_in", x, 1)
    nlev = len(cfg.ch_mult)
    for lvl in range(nlev):
        for i in range(cfg.num_res_blocks):
            h = resnet_block(W, e + f"down.{lvl}.block.{i}", h)
        if lvl != nlev - 1:  # Downsample (model.py:72-76): zero pad right/bottom by one, conv k3 s2 p0
            h = F.conv2d(F.pad(h, (0, 1, 0, 1)), W[e + f"down.{lvl}.downsample.conv.weight"],
                         W[e + f"down.{lvl}.downsample.conv.bias"], stride=2)
    h = resnet_block(W, e + "mid.block_1", h)
    h = attn_block(W, e + "mid.attn_1", h)
    h = resnet_block(W, e + "mid.block_2", h)
    h = _conv(W, e + "conv_out", _swish(_gn(W, e + "norm_out", h)), 1)
    return _conv(W, P + "quant_conv", h, 0)
