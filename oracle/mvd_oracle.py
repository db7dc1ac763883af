"""ORACLE -- TEST INFRASTRUCTURE ONLY.  CPU restatement (plain PyTorch fp32, functional, own code) of the
reference's multi-view denoising step.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
may import this module, and only as the checker / the timed CPU baseline -- never as the product path.

Pinning: every function below is checked against golden vectors produced by running the reference's own
Python (imported from /root/reference in the build container by tools/make_goldens.py) on identical seeded
inputs; see tests/test_oracle_golden.py.  The one exception is the sparse voxel CNN (``sparse_conv_net``):
its arithmetic lives in spconv (requirements.txt:18 ``spconv-cu113``, version un-pinned, not vendored, not
importable here).  It is restated from spconv's documented semantics and checked against a dense-masked
emulation of the same semantics -> PARITY UNPINNED for that one layer.

Every function cites the reference file:line it follows.  Tensors are NCHW / NCDHW fp32 like the reference.
``W`` is a flat dict keyed by the reference's state_dict names.
"""
import math

import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------------------- primitives
def silu(x):
    return x * torch.sigmoid(x)


def group_norm(x, groups, w, b, eps):
    """nn.GroupNorm over (C/groups, *spatial); biased variance.  util.py:199-216 (eps 1e-5),
    modules/attention.py:85-86 (eps 1e-6)."""
    B, C = x.shape[:2]
    xg = x.reshape(B, groups, -1).double()
    mean = xg.mean(-1, keepdim=True)
    var = ((xg - mean) ** 2).mean(-1, keepdim=True)
    y = ((xg - mean) / torch.sqrt(var + eps)).float().reshape(x.shape)
    shape = [1, C] + [1] * (x.dim() - 2)
    return y * w.reshape(shape) + b.reshape(shape)


def layer_norm(x, w, b, eps=1e-5):
    """nn.LayerNorm over the last dim (modules/attention.py:257-259)."""
    xd = x.double()
    mean = xd.mean(-1, keepdim=True)
    var = ((xd - mean) ** 2).mean(-1, keepdim=True)
    return ((xd - mean) / torch.sqrt(var + eps)).float() * w + b


def gelu_erf(x):
    """F.gelu default = exact erf form (modules/attention.py:44)."""
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


def timestep_embedding(t, dim, max_period=10000):
    """util.py:151-171: cat(cos(t f), sin(t f)), f_i = exp(-ln(max_period) i / half)."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(half, dtype=torch.float32) / half)
    args = t[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


def sample_zeros_align(vol, coords):
    """F.grid_sample(mode='bilinear', padding_mode='zeros', align_corners=True) restated with explicit
    index arithmetic, for 2-D ([B,C,H,W] with coords [...,2]=(x,y)) and 3-D ([B,C,D,H,W], (x,y,z)).
    coords: [B, P, nd] normalised to [-1,1]; returns [B, C, P].
    Used at morphable_diffusion.py:218,229,255,315."""
    B, C = vol.shape[:2]
    dims = list(vol.shape[2:])  # (D,)H,W
    nd = len(dims)
    P = coords.shape[1]
    flat = vol.reshape(B, C, -1)
    # pixel-space position along each axis; coords[...,0] indexes the LAST spatial dim
    pos = [(coords[..., a] + 1.0) * 0.5 * (dims[nd - 1 - a] - 1) for a in range(nd)]
    lo = [torch.floor(p) for p in pos]
    fr = [p - l for p, l in zip(pos, lo)]
    out = torch.zeros(B, C, P, dtype=vol.dtype)
    for corner in range(1 << nd):
        wgt = torch.ones(B, P, dtype=vol.dtype)
        idx = torch.zeros(B, P, dtype=torch.long)
        ok = torch.ones(B, P, dtype=torch.bool)
        stride = 1
        for a in range(nd):  # a=0 is x (fastest)
            bit = (corner >> a) & 1
            ia = lo[a] + bit
            wgt = wgt * (fr[a] if bit else (1.0 - fr[a]))
            size = dims[nd - 1 - a]
            ok = ok & (ia >= 0) & (ia <= size - 1)
            idx = idx + ia.clamp(0, size - 1).long() * stride
            stride *= size
        g = torch.gather(flat, 2, idx[:, None, :].expand(B, C, P))
        out = out + g * (wgt * ok.to(vol.dtype))[:, None, :]
    return out


# ------------------------------------------------------------------------------------------ schedules
def ddim_timesteps(num_ddim=50, num_ddpm=1000):
    """util.py:46-60 'uniform': range(0, T, T//S) + 1."""
    c = num_ddpm // num_ddim
    return torch.arange(0, num_ddpm, c, dtype=torch.long) + 1


def ddim_tables(num_ddim=50, eta=1.0, num_ddpm=1000, linear_start=0.00085, linear_end=0.0120):
    """morphable_diffusion.py:428-450 (schedule buffers) and :658-672 (DDIM tables)."""
    betas = torch.linspace(linear_start ** 0.5, linear_end ** 0.5, num_ddpm, dtype=torch.float32) ** 2
    alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
    ts = ddim_timesteps(num_ddim, num_ddpm)
    a = alphas_cumprod[ts].double()
    a_prev = torch.cat([alphas_cumprod[0:1], alphas_cumprod[ts[:-1]]], 0)
    sig = eta * torch.sqrt((1 - a_prev) / (1 - a) * (1 - a / a_prev))
    a32 = a.float()
    return {"timesteps": ts, "alphas": a32, "alphas_prev": a_prev.float(), "sigmas": sig.float(),
            "sqrt_one_minus_alphas": torch.sqrt(1.0 - a32).float()}


def ddim_update(x, eps, tab, index, noise=None):
    """denoise_apply_impl, morphable_diffusion.py:675-698.  ``noise`` is the N(0,1) draw (None on the
    last step, index 0)."""
    a_t, a_prev = tab["alphas"][index], tab["alphas_prev"][index]
    s1m, sig = tab["sqrt_one_minus_alphas"][index], tab["sigmas"][index]
    pred_x0 = (x - s1m * eps) / a_t.sqrt()
    dir_xt = torch.clamp(1.0 - a_prev - sig ** 2, min=1e-7).sqrt() * eps
    x_prev = a_prev.sqrt() * pred_x0 + dir_xt
    if noise is not None:
        x_prev = x_prev + sig * noise
    return x_prev


# ----------------------------------------------------------------------------------------------- UNet
def res_block(W, p, x, emb):
    """ResBlock._forward, openaimodel.py:256-276 (no up/down, no scale-shift, dropout 0)."""
    h = silu(group_norm(x, 32, W[p + ".in_layers.0.weight"], W[p + ".in_layers.0.bias"], 1e-5))
    h = F.conv2d(h, W[p + ".in_layers.2.weight"], W[p + ".in_layers.2.bias"], padding=1)
    e = F.linear(silu(emb), W[p + ".emb_layers.1.weight"], W[p + ".emb_layers.1.bias"])
    h = h + e[:, :, None, None]
    h = silu(group_norm(h, 32, W[p + ".out_layers.0.weight"], W[p + ".out_layers.0.bias"], 1e-5))
    h = F.conv2d(h, W[p + ".out_layers.3.weight"], W[p + ".out_layers.3.bias"], padding=1)
    if p + ".skip_connection.weight" in W:
        x = F.conv2d(x, W[p + ".skip_connection.weight"], W[p + ".skip_connection.bias"])
    return x + h


def multihead_attention(W, p, x, ctx, heads):
    """CrossAttention.forward, modules/attention.py:179-203 (no mask, dropout 0)."""
    q = F.linear(x, W[p + ".to_q.weight"])
    k = F.linear(ctx, W[p + ".to_k.weight"])
    v = F.linear(ctx, W[p + ".to_v.weight"])
    B, T, Cq = q.shape
    S = k.shape[1]
    d = Cq // heads
    q = q.reshape(B, T, heads, d).permute(0, 2, 1, 3)
    k = k.reshape(B, S, heads, d).permute(0, 2, 1, 3)
    v = v.reshape(B, S, heads, d).permute(0, 2, 1, 3)
    sim = torch.matmul(q, k.transpose(-1, -2)) * (d ** -0.5)
    o = torch.matmul(torch.softmax(sim, dim=-1), v)
    o = o.permute(0, 2, 1, 3).reshape(B, T, Cq)
    return F.linear(o, W[p + ".to_out.0.weight"], W[p + ".to_out.0.bias"])


def spatial_transformer(W, p, x, context, heads):
    """SpatialTransformer.forward modules/attention.py:325-336 + BasicTransformerBlock._forward :265-269
    + FeedForward/GEGLU :37-73."""
    B, C, H, Wd = x.shape
    h = group_norm(x, 32, W[p + ".norm.weight"], W[p + ".norm.bias"], 1e-6)
    h = F.conv2d(h, W[p + ".proj_in.weight"], W[p + ".proj_in.bias"])
    t = h.reshape(B, C, H * Wd).permute(0, 2, 1)
    tb = p + ".transformer_blocks.0"
    n1 = layer_norm(t, W[tb + ".norm1.weight"], W[tb + ".norm1.bias"])
    t = multihead_attention(W, tb + ".attn1", n1, n1, heads) + t
    n2 = layer_norm(t, W[tb + ".norm2.weight"], W[tb + ".norm2.bias"])
    t = multihead_attention(W, tb + ".attn2", n2, context, heads) + t
    n3 = layer_norm(t, W[tb + ".norm3.weight"], W[tb + ".norm3.bias"])
    a, g = F.linear(n3, W[tb + ".ff.net.0.proj.weight"], W[tb + ".ff.net.0.proj.bias"]).chunk(2, dim=-1)
    t = F.linear(a * gelu_erf(g), W[tb + ".ff.net.2.weight"], W[tb + ".ff.net.2.bias"]) + t
    h = t.permute(0, 2, 1).reshape(B, C, H, Wd)
    return F.conv2d(h, W[p + ".proj_out.weight"], W[p + ".proj_out.bias"]) + x


def depth_attention(W, p, x, context, heads=4):
    """DepthAttention.forward, attention.py:26-47: softmax over the depth axis of the view frustum."""
    b, inner, h, w = x.shape
    D = context.shape[2]
    hd = inner // heads
    q = F.conv2d(x, W[p + ".to_q.weight"]).reshape(b, heads, hd, 1, h, w)
    k = F.conv3d(context, W[p + ".to_k.weight"]).reshape(b, heads, hd, D, h, w)
    v = F.conv3d(context, W[p + ".to_v.weight"]).reshape(b, heads, hd, D, h, w)
    sim = (q * k).sum(2) * (hd ** -0.5)  # b,heads,D,h,w
    attn = torch.softmax(sim, dim=2)
    out = (v * attn[:, :, None]).sum(3).reshape(b, inner, h, w)
    return F.conv2d(out, W[p + ".to_out.weight"])


def depth_transformer(W, p, x, context):
    """DepthTransformer._forward, attention.py:78-84 (proj_in :53-57, proj_context :58-62, proj_out :64-71)."""
    h = F.conv2d(x, W[p + ".proj_in.0.weight"], W[p + ".proj_in.0.bias"])
    h = silu(group_norm(h, 8, W[p + ".proj_in.1.weight"], W[p + ".proj_in.1.bias"], 1e-5))
    c = F.conv3d(context, W[p + ".proj_context.0.weight"])
    c = torch.relu(group_norm(c, 8, W[p + ".proj_context.1.weight"], W[p + ".proj_context.1.bias"], 1e-5))
    h = depth_attention(W, p + ".depth_attn", h, c)
    h = torch.relu(group_norm(h, 8, W[p + ".proj_out.0.weight"], W[p + ".proj_out.0.bias"], 1e-5))
    h = F.conv2d(h, W[p + ".proj_out.2.weight"], padding=1)
    h = torch.relu(group_norm(h, 8, W[p + ".proj_out.3.weight"], W[p + ".proj_out.3.bias"], 1e-5))
    h = F.conv2d(h, W[p + ".proj_out.5.weight"], padding=1)
    return h + x


def _run_ops(W, pre, ops, h, emb, context, heads):
    for op in ops:
        p = pre + op.name
        if op.kind == "conv_in":
            h = F.conv2d(h, W[p + ".weight"], W[p + ".bias"], padding=1)
        elif op.kind == "res":
            h = res_block(W, p, h, emb)
        elif op.kind == "st":
            h = spatial_transformer(W, p, h, context, heads)
        elif op.kind == "down":  # Downsample, openaimodel.py:135-161
            h = F.conv2d(h, W[p + ".op.weight"], W[p + ".op.bias"], stride=2, padding=1)
        elif op.kind == "up":  # Upsample, openaimodel.py:92-120: nearest x2 then conv
            h = h.repeat_interleave(2, dim=2).repeat_interleave(2, dim=3)
            h = F.conv2d(h, W[p + ".conv.weight"], W[p + ".conv.bias"], padding=1)
        else:
            raise ValueError(op.kind)
    return h


def unet_forward(W, plan, x, timesteps, context, source_dict, prefix="model.diffusion_model."):
    """DepthWiseAttention.forward, attention.py:117-138 (base UNetModel.forward openaimodel.py:745)."""
    cfg = plan.cfg
    emb = timestep_embedding(timesteps, cfg.model_channels)
    emb = F.linear(emb, W[prefix + "time_embed.0.weight"], W[prefix + "time_embed.0.bias"])
    emb = F.linear(silu(emb), W[prefix + "time_embed.2.weight"], W[prefix + "time_embed.2.bias"])
    hs = []
    h = x
    for ops in plan.input_blocks:
        h = _run_ops(W, prefix, ops, h, emb, context, cfg.num_heads)
        hs.append(h)
    h = _run_ops(W, prefix, plan.middle, h, emb, context, cfg.num_heads)
    h = depth_transformer(W, prefix + "middle_conditions", h, source_dict[h.shape[-1]])
    for bi, ops in enumerate(plan.output_blocks):
        h = torch.cat([h, hs.pop()], dim=1)
        h = _run_ops(W, prefix, ops, h, emb, context, cfg.num_heads)
        if bi in plan.out_cond_of_block:
            k = plan.out_cond_of_block[bi]
            h = depth_transformer(W, prefix + f"output_conditions.{k}", h, source_dict[h.shape[-1]])
    h = silu(group_norm(h, 32, W[prefix + "out.0.weight"], W[prefix + "out.0.bias"], 1e-5))
    return F.conv2d(h, W[prefix + "out.2.weight"], W[prefix + "out.2.bias"], padding=1)


def predict_with_unconditional_scale(W, plan, x, t, clip_embed, volume_feats, x_concat, scale,
                                     prefix="model.diffusion_model."):
    """UNetWrapper.predict_with_unconditional_scale, morphable_diffusion.py:132-149:
    batch = [cond || uncond]; uncond = zero clip / zero volumes / zero concat; concat latents / 0.18215."""
    x_ = torch.cat([x, x], 0)
    t_ = torch.cat([t, t], 0)
    c_ = torch.cat([clip_embed, torch.zeros_like(clip_embed)], 0)
    v_ = {k: torch.cat([v, torch.zeros_like(v)], 0) for k, v in volume_feats.items()}
    xc = torch.cat([x_concat, torch.zeros_like(x_concat)], 0).clone()
    xc[:, :4] = xc[:, :4] / 0.18215
    s, s_uc = unet_forward(W, plan, torch.cat([x_, xc], 1), t_, c_, v_, prefix).chunk(2)
    return s_uc + scale * (s - s_uc)


# ------------------------------------------------------------------------------------ mesh conditioner
def embed_time(W, t, dim=256):
    """SyncMultiviewDiffusion.embed_time, morphable_diffusion.py:491-494 (+ :452-458)."""
    e = timestep_embedding(t, dim)
    e = F.linear(e, W["time_embed.0.weight"], W["time_embed.0.bias"])
    return F.linear(silu(e), W["time_embed.2.weight"], W["time_embed.2.bias"])


def viewpoint_embedding(batch):
    """get_viewpoint_embedding, morphable_diffusion.py:383-397: [d_elev, sin d_az, cos d_az, 0] (rad)."""
    d_e = torch.deg2rad(batch["target_elevation"]) - torch.deg2rad(batch["input_elevation"])
    d_a = torch.deg2rad(batch["target_azimuth"]) - torch.deg2rad(batch["input_azimuth"])
    return torch.stack([d_e, torch.sin(d_a), torch.cos(d_a), torch.zeros_like(d_a)], -1)


def target_encoder(W, x, t, v, p="spatial_volume.target_encoder."):
    """NoisyTargetViewEncoder.forward network.py:196-207 + Image2DResBlockWithTV :163-179."""
    h = F.conv2d(x, W[p + "init_conv.weight"], W[p + "init_conv.bias"], padding=1)
    for i in range(3):
        q = f"{p}out_conv{i}."
        te = F.linear(t, W[q + "time_embed.weight"].flatten(1), W[q + "time_embed.bias"])
        ve = F.linear(v, W[q + "view_embed.weight"].flatten(1), W[q + "view_embed.bias"])
        r = h + (te + ve)[:, :, None, None]
        r = silu(group_norm(r, 8, W[q + "conv.0.weight"], W[q + "conv.0.bias"], 1e-5))
        r = F.conv2d(r, W[q + "conv.2.weight"], W[q + "conv.2.bias"], padding=1)
        r = silu(group_norm(r, 8, W[q + "conv.3.weight"], W[q + "conv.3.bias"], 1e-5))
        r = F.conv2d(r, W[q + "conv.5.weight"], W[q + "conv.5.bias"], padding=1)
        h = h + r
    h = silu(group_norm(h, 8, W[p + "final_out.0.weight"], W[p + "final_out.0.bias"], 1e-5))
    return F.conv2d(h, W[p + "final_out.2.weight"], W[p + "final_out.2.bias"], padding=1)


def projection_matrix(ratio, K, RT, projection):
    """construct_project_matrix, utils.py:46-69.  K [B,4,4], RT [B,3,4] -> [B,4,4]."""
    B = K.shape[0]
    bottom = torch.tensor([0.0, 0.0, 0.0, 1.0]).reshape(1, 1, 4).expand(B, 1, 4)
    if projection == "perspective":
        S = torch.diag(torch.tensor([ratio, ratio, 1.0]))
        return torch.cat([S[None] @ K[:, :3, :3] @ RT, bottom], 1)
    if projection == "orthographic":
        return K @ torch.cat([RT, bottom], 1)
    raise NotImplementedError(projection)


def lattice(V, length):
    """morphable_diffusion.py:197-200: [V^3,3] world xyz, x fastest (index = (z*V + y)*V + x)."""
    lin = torch.linspace(-length, length, V, dtype=torch.float32)
    z, y, x = torch.meshgrid(lin, lin, lin, indexing="ij")
    return torch.stack([x, y, z], -1).reshape(-1, 3)


def warp_coordinates(pts, size, input_size, K, RT, projection):
    """get_warp_coordinates utils.py:71-76 + project_and_normalize :20-43.  pts [P,3] -> [B,P,2] in the
    normalised grid_sample frame of a size x size feature map."""
    P4 = projection_matrix(size / input_size, K, RT, projection)
    cam = torch.einsum("bij,pj->bpi", P4[:, :3, :3], pts) + P4[:, None, :3, 3]
    if projection == "perspective":
        w = cam[..., 2:3].clamp(min=1e-4)
        return cam[..., :2] / w / ((size - 1) / 2) - 1.0
    return cam[..., :2]


def vertex_features(W, vcfg, x_noisy, t_embed, v_embed, batch):
    """morphable_diffusion.py:203-229: per view encode -> unproject to the V^3 lattice (bilinear) ->
    trilinear gather at the mesh vertices.  Returns [B,N,16,Nv]."""
    B, N = x_noisy.shape[:2]
    V = vcfg.spatial_volume_size
    pts = lattice(V, vcfg.spatial_volume_length)
    verts = batch["vertices"] / vcfg.spatial_volume_length
    outs = []
    for ni in range(N):
        f = target_encoder(W, x_noisy[:, ni], t_embed, v_embed[:, ni])
        uv = warp_coordinates(pts, f.shape[-1], vcfg.input_image_size, batch["target_K"][:, ni],
                              batch["target_RT"][:, ni], vcfg.projection)
        vol = sample_zeros_align(f, uv).reshape(B, -1, V, V, V)
        outs.append(sample_zeros_align(vol, verts))
    return torch.stack(outs, 1)


def fuse_views(W, feats, p="spatial_volume.smpl_feature_extractor."):
    """SMPLFeatureExtractor.forward network.py:41-72 with filter_channels [16,16], no_residual False:
    one k=1 Conv1d per view then the mean over views.  [B,N,16,Nv] -> [B,Nv,16]."""
    B, N, C, Nv = feats.shape
    y = F.conv1d(feats.reshape(B * N, C, Nv), W[p + "conv0.weight"], W[p + "conv0.bias"])
    return y.reshape(B, N, -1, Nv).mean(1).permute(0, 2, 1)


def _bn_relu(W, p, x, train=False, bn_update=None, rows=None):
    """BatchNorm1d(eps 1e-3, momentum 0.01: network.py:105) over the active rows + ReLU.  eval: running statistics; train (the
    module is in train mode during training_step): statistics of this sample's active rows, biased variance.  bn_update (a
    dict, optional): receives the running buffers as nn.BatchNorm1d leaves them after this call -- running = 0.99 running +
    0.01 batch statistic, the variance as the unbiased estimate -- chained over calls (a later call starts from the dict).
    rows (optional, train mode): index of x's row for every row of the layer's feature matrix -- several vertices in one voxel
    keep a feature row each (copies of the voxel's representative), and BatchNorm1d's batch statistics run over all of them."""
    if train:
        xs = x if rows is None else x[rows]
        mean, var = xs.mean(0), xs.var(0, unbiased=False)
        if bn_update is not None:
            n = xs.shape[0]
            rm = bn_update.get(p + ".running_mean", W[p + ".running_mean"])
            rv = bn_update.get(p + ".running_var", W[p + ".running_var"])
            bn_update[p + ".running_mean"] = 0.99 * rm + 0.01 * mean
            bn_update[p + ".running_var"] = 0.99 * rv + 0.01 * var * (n / max(n - 1, 1))
    else:
        mean, var = W[p + ".running_mean"], W[p + ".running_var"]
    y = (x - mean) / torch.sqrt(var + 1e-3) * W[p + ".weight"] + W[p + ".bias"]
    return torch.relu(y)


def _index_grid(coords, shape):
    grid = torch.full(tuple(shape), -1, dtype=torch.long)
    grid[coords[:, 0], coords[:, 1], coords[:, 2]] = torch.arange(coords.shape[0])
    return grid


def _subm_conv(feats, coords, shape, weight):
    """SubMConv3d(k=3, bias=False): outputs only at the active input sites."""
    grid = F.pad(_index_grid(coords, shape), (1, 1, 1, 1, 1, 1), value=-1)
    out = torch.zeros(feats.shape[0], weight.shape[0])
    for kd in range(3):
        for kh in range(3):
            for kw in range(3):
                nb = grid[coords[:, 0] + kd, coords[:, 1] + kh, coords[:, 2] + kw]
                ok = nb >= 0
                out[ok] += feats[nb[ok]] @ weight[:, :, kd, kh, kw].t()
    return out


def _strided_conv(feats, coords, shape, weight):
    """SparseConv3d(k=3, s=2, p=1, bias=False): an output site is active iff an active input lies in its
    receptive field; out dim = (D - 1) // 2 + 1."""
    oshape = [(s - 1) // 2 + 1 for s in shape]
    cand = []
    for kd in range(3):
        for kh in range(3):
            for kw in range(3):
                num = coords + 1 - torch.tensor([kd, kh, kw])
                ok = ((num % 2) == 0).all(1)
                o = num[ok] // 2
                ok2 = ((o >= 0) & (o < torch.tensor(oshape))).all(1)
                cand.append(o[ok2])
    lin = torch.cat(cand)
    key = (lin[:, 0] * oshape[1] + lin[:, 1]) * oshape[2] + lin[:, 2]
    key = torch.unique(key)  # sorted -> deterministic row order (z, y, x)
    ocoords = torch.stack([key // (oshape[1] * oshape[2]), (key // oshape[2]) % oshape[1], key % oshape[2]], 1)
    grid = F.pad(_index_grid(coords, shape), (1, 2, 1, 2, 1, 2), value=-1)
    out = torch.zeros(ocoords.shape[0], weight.shape[0])
    for kd in range(3):
        for kh in range(3):
            for kw in range(3):
                nb = grid[2 * ocoords[:, 0] + kd, 2 * ocoords[:, 1] + kh, 2 * ocoords[:, 2] + kw]
                ok = nb >= 0
                out[ok] += feats[nb[ok]] @ weight[:, :, kd, kh, kw].t()
    return out, ocoords, oshape


def sparse_conv_net(W, feats, coords, out_sh, p="spatial_volume.xyzc_net.", train=False, bn_update=None):
    """SparseConvNet.forward network.py:85-96 (double_conv :109, stride_conv :152, triple_conv :127),
    eval-mode BatchNorm1d(eps 1e-3).  feats [Nv,16], coords [Nv,3] (z,y,x) int; returns the
    dense [1,64,D/4,H/4,W/4] volume.  PARITY UNPINNED (spconv absent) -- see module header.
    Several vertices in one voxel (real FLAME meshes at 5 mm have them): spconv's hash table keeps ONE row per voxel
    and every neighbour lookup -- the centre tap included -- goes through it, so all rows of a voxel compute the same
    output from that representative's feature.  Which row wins is a race in spconv; here (and in the HIP engine) the
    FIRST occurrence is the representative, i.e. later duplicates are dropped."""
    coords = coords.long()
    shape = [int(s) for s in out_sh]
    key = (coords[:, 0] * shape[1] + coords[:, 1]) * shape[2] + coords[:, 2]
    seen, keep = set(), []
    for i, k in enumerate(key.tolist()):
        if k not in seen:
            seen.add(k)
            keep.append(i)
    rows0 = None
    if len(keep) != coords.shape[0]:
        first = {}
        for j, i in enumerate(keep):
            first[key[i].item()] = j
        rows0 = torch.tensor([first[k] for k in key.tolist()])  # feature row of every vertex = its voxel's representative
        coords, feats = coords[keep], feats[keep]
    x = feats
    for blk, n in (("conv0", 2), ("down0", 1), ("conv1", 2), ("down1", 1), ("conv2", 3)):
        for i in range(n):
            w = W[f"{p}{blk}.{3 * i}.weight"]
            if blk.startswith("down"):
                x, coords, shape = _strided_conv(x, coords, shape, w)
            else:
                x = _subm_conv(x, coords, shape, w)
            x = _bn_relu(W, f"{p}{blk}.{3 * i + 1}", x, train, bn_update, rows=rows0 if blk == "conv0" else None)
    dense = torch.zeros([x.shape[1]] + shape)
    dense[:, coords[:, 0], coords[:, 1], coords[:, 2]] = x.t()
    return dense[None]


def latent_volume(vcfg, feature_volume, bounds_min_xyz, out_sh):
    """morphable_diffusion.py:232-257: sample the sparse-CNN output at the V^3 lattice.  The normalisation
    divides voxel coordinates by the FULL-resolution out_sh although the volume is out_sh/4 (reference
    behaviour, kept).  Returns [1,64,V,V,V]."""
    V = vcfg.spatial_volume_size
    pts = lattice(V, vcfg.spatial_volume_length)  # xyz
    g = (pts - bounds_min_xyz[None]) / vcfg.voxel_size  # voxel units, xyz order
    sh_xyz = torch.tensor([float(out_sh[2]), float(out_sh[1]), float(out_sh[0])])
    g = g / sh_xyz * 2 - 1
    return sample_zeros_align(feature_volume, g[None]).reshape(1, -1, V, V, V)


def construct_spatial_volume(W, vcfg, x_noisy, t_embed, v_embed, batch, train=False, bn_update=None):
    """SpatialVolumeNet.construct_spatial_volume, morphable_diffusion.py:182-263 (use_spatial_volume False).
    train: the sparse CNN's BatchNorm layers use batch statistics (module in train mode)."""
    B = x_noisy.shape[0]
    fused = fuse_views(W, vertex_features(W, vcfg, x_noisy, t_embed, v_embed, batch))  # B,Nv,16
    vols = []
    for bi in range(B):
        fv = sparse_conv_net(W, fused[bi], batch["coord"][bi], batch["out_sh"][bi], train=train, bn_update=bn_update)
        vols.append(latent_volume(vcfg, fv, batch["bounds"][bi, 0], batch["out_sh"][bi])[0])
    return torch.stack(vols)


def frustum_points(vcfg, RT, K):
    """create_target_volume utils.py:79-153 with near/far from the camera distance
    (morphable_diffusion.py:281-299).  RT [M,3,4], K [M,4,4] -> world xyz [M,3,D,H,W]."""
    M = RT.shape[0]
    S, D = vcfg.frustum_volume_size, vcfg.frustum_volume_depth
    cam_pos = -(RT[:, :, :3].transpose(1, 2) @ RT[:, :, 3:])[:, :, 0]
    dist = torch.linalg.norm(cam_pos, dim=-1)
    near, far = dist - vcfg.frustum_volume_length, dist + vcfg.frustum_volume_length
    depth = torch.linspace(0, 1, D)[None] * (far - near)[:, None] + near[:, None]  # M,D
    ys, xs = torch.meshgrid(torch.arange(S, dtype=torch.float32), torch.arange(S, dtype=torch.float32), indexing="ij")
    xs, ys = xs.reshape(-1), ys.reshape(-1)
    if vcfg.projection == "perspective":
        Pinv = torch.linalg.inv(projection_matrix(S / vcfg.input_image_size, K, RT, "perspective"))
        pix = torch.stack([xs, ys, torch.ones_like(xs)], 0)  # 3,HW
        g = pix[None, :, None, :] * depth[:, None, :, None]  # M,3,D,HW
        world = torch.einsum("mij,mjdp->midp", Pinv[:, :3, :3], g) + Pinv[:, :3, 3][:, :, None, None]
    elif vcfg.projection == "orthographic":
        pix = torch.stack([2 * xs / (S - 1) - 1, 2 * ys / (S - 1) - 1, torch.ones_like(xs)], 0)
        Kinv = torch.linalg.inv(K)
        cam = torch.einsum("mij,jp->mip", Kinv[:, :3, :3], pix)[:, :, None, :].repeat(1, 1, D, 1)
        cam[:, 2] = depth[:, :, None]
        RTinv = torch.linalg.inv(projection_matrix(1.0, torch.eye(4)[None].repeat(M, 1, 1), RT, "orthographic"))
        world = torch.einsum("mij,mjdp->midp", RTinv[:, :3, :3], cam) + RTinv[:, :3, 3][:, :, None, None]
    else:
        raise NotImplementedError(vcfg.projection)
    return world.reshape(M, 3, D, S, S)


def frustum_net(W, x, t, v, p="spatial_volume.frustum_volume_feats."):
    """FrustumTV3DNet.forward network.py:332-347; FrustumTVBlock :285-297, FrustumTVUpBlock :299-311."""
    def film(q, h):
        te = F.linear(t, W[q + "t_conv.weight"].flatten(1), W[q + "t_conv.bias"])
        ve = F.linear(v, W[q + "v_conv.weight"].flatten(1), W[q + "v_conv.bias"])
        return h + (te + ve)[:, :, None, None, None]

    def block(i, h, stride):
        q = f"{p}conv{i}."
        h = silu(group_norm(film(q, h), 8, W[q + "bn.weight"], W[q + "bn.bias"], 1e-5))
        return F.conv3d(h, W[q + "conv.weight"], W[q + "conv.bias"], stride=stride, padding=1)

    def up(i, h):
        q = f"{p}up{i}."
        h = silu(group_norm(film(q, h), 8, W[q + "norm.weight"], W[q + "norm.bias"], 1e-5))
        return F.conv_transpose3d(h, W[q + "conv.weight"], W[q + "conv.bias"], stride=2, padding=1, output_padding=1)

    x0 = F.conv3d(x, W[p + "conv0.weight"], W[p + "conv0.bias"], padding=1)
    x1 = block(2, block(1, x0, 2), 1)
    x2 = block(4, block(3, x1, 2), 1)
    x3 = block(6, block(5, x2, 2), 1)
    x2 = up(0, x3) + x2
    x1 = up(1, x2) + x1
    x0 = up(2, x1) + x0
    w = x.shape[-1]
    return {w: x0, w // 2: x1, w // 4: x2, w // 8: x3}


def construct_view_frustum_volume(W, vcfg, spatial_volume, t_embed, v_embed, target_indices, batch):
    """SpatialVolumeNet.construct_view_frustum_volume, morphable_diffusion.py:265-320."""
    B, TN = target_indices.shape
    bi = torch.arange(B)[:, None]
    RT = batch["target_RT"][bi, target_indices].reshape(B * TN, 3, 4)
    K = batch["target_K"][bi, target_indices].reshape(B * TN, 4, 4)
    xyz = frustum_points(vcfg, RT, K) / vcfg.spatial_volume_length
    M = B * TN
    vol = spatial_volume[:, None].expand(-1, TN, -1, -1, -1, -1).reshape(M, *spatial_volume.shape[1:])
    D, S = vcfg.frustum_volume_depth, vcfg.frustum_volume_size
    feats = sample_zeros_align(vol, xyz.reshape(M, 3, -1).transpose(1, 2)).reshape(M, -1, D, S, S)
    v_ = v_embed[bi, target_indices].reshape(M, -1)
    t_ = t_embed[:, None].expand(-1, TN, -1).reshape(M, -1)
    return frustum_net(W, feats, t_, v_)


# ---------------------------------------------------------------------------------------- one DDIM step
def denoise_apply(W, plan, vcfg, tab, x_target_noisy, x_input, clip_embed, time_steps, index, scale,
                  batch, batch_view_num=1, noise=None, return_eps=False):
    """SyncDDIMSampler.denoise_apply, morphable_diffusion.py:701-739 -- the BASELINE unit of work.
    ``noise``: explicit N(0,1) tensor for the eta=1 stochastic term (None -> is_step0 behaviour).
    ``return_eps``: also return the guided noise prediction e_t (:736) that denoise_apply_impl consumes."""
    B, N, C, H, Wd = x_target_noisy.shape
    v_embed = viewpoint_embedding(batch)
    t_embed = embed_time(W, time_steps, vcfg.time_dim)
    sv = construct_spatial_volume(W, vcfg, x_target_noisy, t_embed, v_embed, batch)
    e_t = []
    for ni in range(0, N, batch_view_num):
        xs = x_target_noisy[:, ni:ni + batch_view_num]
        VN = xs.shape[1]
        idx = torch.arange(ni, ni + VN)[None].repeat(B, 1)
        vf = construct_view_frustum_volume(W, vcfg, sv, t_embed, v_embed, idx, batch)
        rep = lambda z: z[:, None].expand(-1, VN, *z.shape[1:]).reshape(B * VN, *z.shape[1:])
        xs_ = xs.reshape(B * VN, C, H, Wd)
        if scale != 1.0:
            e = predict_with_unconditional_scale(W, plan, xs_, rep(time_steps), rep(clip_embed), vf, rep(x_input), scale)
        else:
            xc = rep(x_input).clone()
            xc[:, :4] = xc[:, :4] / 0.18215
            e = unet_forward(W, plan, torch.cat([xs_, xc], 1), rep(time_steps), rep(clip_embed), vf)
        e_t.append(e.reshape(B, VN, 4, H, Wd))
    e_t = torch.cat(e_t, 1)
    x_prev = ddim_update(x_target_noisy, e_t, tab, index, noise)
    return (x_prev, e_t) if return_eps else x_prev


def sample(W, plan, vcfg, x_input, clip_embed, scale, batch, num_ddim=50, eta=1.0, batch_view_num=1, log_every_t=50,
           generator=None, latent_size=32):
    """SyncDDIMSampler.sample, morphable_diffusion.py:742-776: x_T ~ N(0,1) drawn first, then the steps in
    flip(ddim_timesteps) order with index = total-1-i, one randn_like per step except index 0 (:695-697), intermediates
    at index % log_every_t == 0 or the first step (:772-773).  Returns (x, x_inter, eps_per_step)."""
    tab = ddim_tables(num_ddim, eta)
    B, N = clip_embed.shape[0], vcfg.num_views
    x = torch.randn([B, N, 4, latent_size, latent_size], generator=generator)
    ts = tab["timesteps"]
    total = len(ts)
    inter, eps_all = [], []
    for i, step in enumerate(torch.flip(ts, [0]).tolist()):
        index = total - i - 1
        time_steps = torch.full((B,), int(step), dtype=torch.long)
        noise = torch.randn(x.shape, generator=generator) if index != 0 else None
        x, e = denoise_apply(W, plan, vcfg, tab, x, x_input, clip_embed, time_steps, index, scale, batch,
                             batch_view_num=batch_view_num, noise=noise, return_eps=True)
        eps_all.append(e)
        if index % log_every_t == 0 or index == total - 1:
            inter.append(x)
    return x, inter, eps_all


# ------------------------------------------------------------------------------------------ training step
def drop_masks(drop_random):
    """UNetWrapper.get_drop_scheme + the masks of UNetWrapper.forward (morphable_diffusion.py:84-115), 'default' scheme:
    u <= 0.05 drops everything, (0.05, 0.1] the concatenated latent, (0.1, 0.15] the volumes, (0.15, 0.2] the CLIP token.
    Returns the three keep-masks (clip, volume, concat) as float tensors [B]."""
    u = drop_random
    drop_clip = (u > 0.15) & (u <= 0.2)
    drop_volume = (u > 0.1) & (u <= 0.15)
    drop_concat = (u > 0.05) & (u <= 0.1)
    drop_all = u <= 0.05
    return 1.0 - (drop_clip | drop_all).float(), 1.0 - (drop_volume | drop_all).float(), 1.0 - (drop_concat | drop_all).float()


def add_noise(x_start, t, noise, num_ddpm=1000, linear_start=0.00085, linear_end=0.0120):
    """SyncMultiviewDiffusion.add_noise, morphable_diffusion.py:551-565 (schedule buffers :428-450)."""
    betas = torch.linspace(linear_start ** 0.5, linear_end ** 0.5, num_ddpm, dtype=torch.float32) ** 2
    ac = torch.cumprod(1.0 - betas, dim=0)
    shape = (x_start.shape[0],) + (1,) * (x_start.dim() - 1)
    return ac.sqrt()[t].view(shape) * x_start + (1.0 - ac).sqrt()[t].view(shape) * noise


def training_step(W, plan, vcfg, x0, x_input, clip_embed, batch, time_steps, noise, target_index, drop_random=None):
    """SyncMultiviewDiffusion.training_step, morphable_diffusion.py:520-549, with the random draws passed in (time_steps [B],
    noise like x0 [B,N,4,h,w], target_index [B,1], drop_random [B] or None = no condition dropout) and ``prepare`` replaced
    by its outputs (x0 = target latents, x_input, clip_embed).  Differentiable: tensors of W with requires_grad receive
    gradients from ``loss.backward()``.  Returns (loss, noise_predict)."""
    B = x0.shape[0]
    x_noisy = add_noise(x0, time_steps, noise)
    v_embed = viewpoint_embedding(batch)
    t_embed = embed_time(W, time_steps, vcfg.time_dim)
    sv = construct_spatial_volume(W, vcfg, x_noisy, t_embed, v_embed, batch, train=True)
    vf = construct_view_frustum_volume(W, vcfg, sv, t_embed, v_embed, target_index, batch)
    ar = torch.arange(B)[:, None]
    xs = x_noisy[ar, target_index][:, 0]
    clip_, xc = clip_embed, x_input
    if drop_random is not None:
        mc, mv, mx = drop_masks(drop_random)
        clip_ = clip_ * mc.view(B, 1, 1)
        vf = {k: v * mv.view(B, 1, 1, 1, 1) for k, v in vf.items()}
        xc = xc * mx.view(B, 1, 1, 1)
    xc = xc.clone()
    xc[:, :4] = xc[:, :4] / 0.18215
    pred = unet_forward(W, plan, torch.cat([xs, xc], 1), time_steps, clip_, vf)
    target = noise[ar, target_index][:, 0]
    return ((target - pred) ** 2).mean(), pred
