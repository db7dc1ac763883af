"""Structure of the multi-view denoiser: block plan + checkpoint key/shape manifest.

Everything here is derived from the reference's constructor arguments (configs/facescape.yaml:26-42) and
mirrors the state_dict layout documented in SURVEY.md Appendix B so that a reference checkpoint's
``state_dict`` can be uploaded key-for-key:

  model.diffusion_model.*   UNet  (ldm/modules/diffusionmodules/openaimodel.py:444-727 ctor,
                                   ldm/models/diffusion/attention.py:87-115 conditioning blocks)
  spatial_volume.*          mesh conditioner + frustum net (ldm/models/diffusion/morphable_diffusion.py:151-180,
                                   ldm/models/diffusion/network.py)
  time_embed.*              256-d step embedding MLP (morphable_diffusion.py:452-458)

The plan is a flat list of ops consumed by the oracle, by the HIP engine (through the C-ABI config struct)
and by the weight uploader, so the three can never disagree about structure.
"""
from dataclasses import dataclass, field
from typing import Dict, List, Tuple

UNET_PREFIX = "model.diffusion_model."
SV_PREFIX = "spatial_volume."


@dataclass
class UNetConfig:
    """kwargs of ldm.models.diffusion.attention.DepthWiseAttention (configs/facescape.yaml:28-42)."""
    volume_dims: Tuple[int, int, int, int] = (64, 128, 256, 512)
    image_size: int = 32
    in_channels: int = 8
    out_channels: int = 4
    model_channels: int = 320
    attention_resolutions: Tuple[int, ...] = (4, 2, 1)
    num_res_blocks: int = 2
    channel_mult: Tuple[int, ...] = (1, 2, 4, 4)
    num_heads: int = 8
    use_spatial_transformer: bool = True
    transformer_depth: int = 1
    context_dim: int = 768
    use_checkpoint: bool = True
    legacy: bool = False

    def validate(self):
        # the reference's own ctor asserts (openaimodel.py:474-509) for the supported subset
        if not self.use_spatial_transformer:
            raise NotImplementedError("only use_spatial_transformer=True is on the hot path")
        if self.transformer_depth != 1 or self.legacy:
            raise NotImplementedError("transformer_depth must be 1 and legacy False (facescape.yaml:40,42)")
        if len(self.channel_mult) != 4 or self.channel_mult[2] != self.channel_mult[3]:
            raise NotImplementedError("channel_mult must have 4 levels with mult[2]==mult[3] (attention.py:97-115)")
        if self.model_channels % 32:
            raise ValueError("model_channels must be divisible by 32 (GroupNorm32)")
        for d in self.volume_dims:
            if d % 8:
                raise ValueError("volume_dims must be divisible by 8 (GroupNorm(8))")


@dataclass
class Op:
    kind: str  # conv_in | res | st | down | up
    name: str  # key prefix relative to the UNet prefix, e.g. "input_blocks.1.0"
    cin: int = 0
    cout: int = 0


@dataclass
class UNetPlan:
    cfg: UNetConfig
    input_blocks: List[List[Op]] = field(default_factory=list)
    middle: List[Op] = field(default_factory=list)
    output_blocks: List[List[Op]] = field(default_factory=list)
    # (name, dim, context_dim) for middle_conditions + output_conditions[k]
    conditions: List[Tuple[str, int, int]] = field(default_factory=list)
    # output block index -> output_conditions index (attention.py:100)
    out_cond_of_block: Dict[int, int] = field(default_factory=dict)


def build_unet_plan(cfg: UNetConfig) -> UNetPlan:
    cfg.validate()
    mc = cfg.model_channels
    plan = UNetPlan(cfg)
    plan.input_blocks.append([Op("conv_in", "input_blocks.0.0", cfg.in_channels, mc)])
    chans = [mc]
    ch, ds, bi = mc, 1, 1
    nlev = len(cfg.channel_mult)
    for level, mult in enumerate(cfg.channel_mult):
        for _ in range(cfg.num_res_blocks):
            ops = [Op("res", f"input_blocks.{bi}.0", ch, mult * mc)]
            ch = mult * mc
            if ds in cfg.attention_resolutions:
                ops.append(Op("st", f"input_blocks.{bi}.1", ch, ch))
            plan.input_blocks.append(ops)
            chans.append(ch)
            bi += 1
        if level != nlev - 1:
            plan.input_blocks.append([Op("down", f"input_blocks.{bi}.0", ch, ch)])
            chans.append(ch)
            bi += 1
            ds *= 2
    plan.middle = [Op("res", "middle_block.0", ch, ch), Op("st", "middle_block.1", ch, ch),
                   Op("res", "middle_block.2", ch, ch)]
    bi = 0
    for level in reversed(range(nlev)):
        mult = cfg.channel_mult[level]
        for i in range(cfg.num_res_blocks + 1):
            ich = chans.pop()
            ops = [Op("res", f"output_blocks.{bi}.0", ch + ich, mc * mult)]
            ch = mc * mult
            j = 1
            if ds in cfg.attention_resolutions:
                ops.append(Op("st", f"output_blocks.{bi}.{j}", ch, ch))
                j += 1
            if level and i == cfg.num_res_blocks:
                ops.append(Op("up", f"output_blocks.{bi}.{j}", ch, ch))
                ds //= 2
            plan.output_blocks.append(ops)
            bi += 1
    d0, d1, d2, d3 = cfg.volume_dims
    c2, c1, c0 = mc * cfg.channel_mult[2], mc * cfg.channel_mult[1], mc * cfg.channel_mult[0]
    plan.conditions = [("middle_conditions", c2, d3),
                       ("output_conditions.0", c2, d2), ("output_conditions.1", c2, d2),
                       ("output_conditions.2", c2, d1), ("output_conditions.3", c1, d1),
                       ("output_conditions.4", c1, d1), ("output_conditions.5", c1, d0),
                       ("output_conditions.6", c0, d0), ("output_conditions.7", c0, d0),
                       ("output_conditions.8", c0, d0)]
    plan.out_cond_of_block = {3 + k: k for k in range(9)}
    return plan


def _res_keys(p, cin, cout, temb):
    ks = {p + ".in_layers.0.weight": (cin,), p + ".in_layers.0.bias": (cin,),
          p + ".in_layers.2.weight": (cout, cin, 3, 3), p + ".in_layers.2.bias": (cout,),
          p + ".emb_layers.1.weight": (cout, temb), p + ".emb_layers.1.bias": (cout,),
          p + ".out_layers.0.weight": (cout,), p + ".out_layers.0.bias": (cout,),
          p + ".out_layers.3.weight": (cout, cout, 3, 3), p + ".out_layers.3.bias": (cout,)}
    if cin != cout:
        ks[p + ".skip_connection.weight"] = (cout, cin, 1, 1)
        ks[p + ".skip_connection.bias"] = (cout,)
    return ks


def _st_keys(p, c, ctx):
    t = p + ".transformer_blocks.0"
    ks = {p + ".norm.weight": (c,), p + ".norm.bias": (c,),
          p + ".proj_in.weight": (c, c, 1, 1), p + ".proj_in.bias": (c,),
          p + ".proj_out.weight": (c, c, 1, 1), p + ".proj_out.bias": (c,)}
    for a, kd in (("attn1", c), ("attn2", ctx)):
        ks[f"{t}.{a}.to_q.weight"] = (c, c)
        ks[f"{t}.{a}.to_k.weight"] = (c, kd)
        ks[f"{t}.{a}.to_v.weight"] = (c, kd)
        ks[f"{t}.{a}.to_out.0.weight"] = (c, c)
        ks[f"{t}.{a}.to_out.0.bias"] = (c,)
    ks[f"{t}.ff.net.0.proj.weight"] = (8 * c, c)
    ks[f"{t}.ff.net.0.proj.bias"] = (8 * c,)
    ks[f"{t}.ff.net.2.weight"] = (c, 4 * c)
    ks[f"{t}.ff.net.2.bias"] = (c,)
    for n in ("norm1", "norm2", "norm3"):
        ks[f"{t}.{n}.weight"] = (c,)
        ks[f"{t}.{n}.bias"] = (c,)
    return ks


def _cond_keys(p, dim, cc):
    inner = 2 * cc  # 4 heads x (cc // 2)  (attention.py:97-115)
    return {p + ".proj_in.0.weight": (inner, dim, 1, 1), p + ".proj_in.0.bias": (inner,),
            p + ".proj_in.1.weight": (inner,), p + ".proj_in.1.bias": (inner,),
            p + ".proj_context.0.weight": (cc, cc, 1, 1, 1),
            p + ".proj_context.1.weight": (cc,), p + ".proj_context.1.bias": (cc,),
            p + ".depth_attn.to_q.weight": (inner, inner, 1, 1),
            p + ".depth_attn.to_k.weight": (inner, cc, 1, 1, 1),
            p + ".depth_attn.to_v.weight": (inner, cc, 1, 1, 1),
            p + ".depth_attn.to_out.weight": (inner, inner, 1, 1),
            p + ".proj_out.0.weight": (inner,), p + ".proj_out.0.bias": (inner,),
            p + ".proj_out.2.weight": (inner, inner, 3, 3),
            p + ".proj_out.3.weight": (inner,), p + ".proj_out.3.bias": (inner,),
            p + ".proj_out.5.weight": (dim, inner, 3, 3)}


def unet_manifest(cfg: UNetConfig, prefix: str = UNET_PREFIX) -> Dict[str, Tuple[int, ...]]:
    """state_dict key -> shape for the UNet (SURVEY.md Appendix B key patterns)."""
    plan = build_unet_plan(cfg)
    mc = cfg.model_channels
    temb = 4 * mc
    ks: Dict[str, Tuple[int, ...]] = {
        "time_embed.0.weight": (temb, mc), "time_embed.0.bias": (temb,),
        "time_embed.2.weight": (temb, temb), "time_embed.2.bias": (temb,)}
    for ops in plan.input_blocks + [plan.middle] + plan.output_blocks:
        for op in ops:
            if op.kind == "conv_in":
                ks[op.name + ".weight"] = (op.cout, op.cin, 3, 3)
                ks[op.name + ".bias"] = (op.cout,)
            elif op.kind == "res":
                ks.update(_res_keys(op.name, op.cin, op.cout, temb))
            elif op.kind == "st":
                ks.update(_st_keys(op.name, op.cout, cfg.context_dim))
            elif op.kind == "down":
                ks[op.name + ".op.weight"] = (op.cout, op.cin, 3, 3)
                ks[op.name + ".op.bias"] = (op.cout,)
            elif op.kind == "up":
                ks[op.name + ".conv.weight"] = (op.cout, op.cin, 3, 3)
                ks[op.name + ".conv.bias"] = (op.cout,)
    ks["out.0.weight"] = (mc,)
    ks["out.0.bias"] = (mc,)
    ks["out.2.weight"] = (cfg.out_channels, mc, 3, 3)
    ks["out.2.bias"] = (cfg.out_channels,)
    for name, dim, cc in plan.conditions:
        ks.update(_cond_keys(name, dim, cc))
    return {prefix + k: v for k, v in ks.items()}


@dataclass
class VolumeConfig:
    """SpatialVolumeNet ctor (morphable_diffusion.py:152-180) + the knobs the reference hard-wires.

    ``num_views`` is SMPLFeatureExtractor.num_views, hard-coded to 16 in the reference
    (morphable_diffusion.py:165-167, SURVEY gotcha G3); ``input_image_size`` is not forwarded from the
    model config in the reference either (gotcha G4).  Both are explicit here.
    """
    time_dim: int = 256
    view_dim: int = 4
    num_views: int = 16
    input_image_size: int = 256
    frustum_volume_depth: int = 48
    spatial_volume_size: int = 32
    spatial_volume_length: float = 0.5
    frustum_volume_length: float = 0.86603
    projection: str = "perspective"
    frustum_dims: Tuple[int, int, int, int] = (64, 128, 256, 512)
    voxel_size: float = 0.005  # hard-coded at morphable_diffusion.py:239 and generate_face.py:221

    @property
    def frustum_volume_size(self):
        return self.input_image_size // 8


def volume_manifest(cfg: VolumeConfig, prefix: str = SV_PREFIX) -> Dict[str, Tuple[int, ...]]:
    td, vd = cfg.time_dim, cfg.view_dim
    ks: Dict[str, Tuple[int, ...]] = {}
    te = "target_encoder."
    ks[te + "init_conv.weight"] = (16, 4, 3, 3)
    ks[te + "init_conv.bias"] = (16,)
    for i in range(3):
        p = f"{te}out_conv{i}."
        ks[p + "time_embed.weight"] = (16, td, 1, 1)
        ks[p + "time_embed.bias"] = (16,)
        ks[p + "view_embed.weight"] = (16, vd, 1, 1)
        ks[p + "view_embed.bias"] = (16,)
        for j, kind in ((0, "n"), (2, "c"), (3, "n"), (5, "c")):
            if kind == "n":
                ks[f"{p}conv.{j}.weight"] = (16,)
                ks[f"{p}conv.{j}.bias"] = (16,)
            else:
                ks[f"{p}conv.{j}.weight"] = (16, 16, 3, 3)
                ks[f"{p}conv.{j}.bias"] = (16,)
    ks[te + "final_out.0.weight"] = (16,)
    ks[te + "final_out.0.bias"] = (16,)
    ks[te + "final_out.2.weight"] = (16, 16, 3, 3)
    ks[te + "final_out.2.bias"] = (16,)
    ks["smpl_feature_extractor.conv0.weight"] = (16, 16, 1)
    ks["smpl_feature_extractor.conv0.bias"] = (16,)
    # sparse voxel CNN (network.py:74-161): (block, n_layers, cin, cout); layer i occupies slots 3i..3i+2
    for blk, n, cin, cout in (("conv0", 2, 16, 16), ("down0", 1, 16, 32), ("conv1", 2, 32, 32),
                              ("down1", 1, 32, 64), ("conv2", 3, 64, 64)):
        for i in range(n):
            ci = cin if i == 0 else cout
            # dense-emulation layout [cout, cin, kd, kh, kw]; real spconv checkpoints use spconv's own layout
            # ([cout,3,3,3,cin] or [3,3,3,cin,cout]), recognised by shape in csrc/engine_weights.hip
            ks[f"xyzc_net.{blk}.{3 * i}.weight"] = (cout, ci, 3, 3, 3)
            for s in ("weight", "bias", "running_mean", "running_var"):
                ks[f"xyzc_net.{blk}.{3 * i + 1}.{s}"] = (cout,)
    d = cfg.frustum_dims
    fv = "frustum_volume_feats."
    ks[fv + "conv0.weight"] = (d[0], 64, 3, 3, 3)
    ks[fv + "conv0.bias"] = (d[0],)
    io = [(d[0], d[1]), (d[1], d[1]), (d[1], d[2]), (d[2], d[2]), (d[2], d[3]), (d[3], d[3])]
    for i, (ci, co) in enumerate(io, start=1):
        p = f"{fv}conv{i}."
        ks[p + "t_conv.weight"] = (ci, td, 1, 1, 1)
        ks[p + "t_conv.bias"] = (ci,)
        ks[p + "v_conv.weight"] = (ci, vd, 1, 1, 1)
        ks[p + "v_conv.bias"] = (ci,)
        ks[p + "bn.weight"] = (ci,)
        ks[p + "bn.bias"] = (ci,)
        ks[p + "conv.weight"] = (co, ci, 3, 3, 3)
        ks[p + "conv.bias"] = (co,)
    for i, (ci, co) in enumerate([(d[3], d[2]), (d[2], d[1]), (d[1], d[0])]):
        p = f"{fv}up{i}."
        ks[p + "t_conv.weight"] = (ci, td, 1, 1, 1)
        ks[p + "t_conv.bias"] = (ci,)
        ks[p + "v_conv.weight"] = (ci, vd, 1, 1, 1)
        ks[p + "v_conv.bias"] = (ci,)
        ks[p + "norm.weight"] = (ci,)
        ks[p + "norm.bias"] = (ci,)
        ks[p + "conv.weight"] = (ci, co, 3, 3, 3)  # ConvTranspose3d: (in, out, k, k, k)
        ks[p + "conv.bias"] = (co,)
    return {prefix + k: v for k, v in ks.items()}


def time_embed_manifest(dim: int = 256) -> Dict[str, Tuple[int, ...]]:
    return {"time_embed.0.weight": (dim, dim), "time_embed.0.bias": (dim,),
            "time_embed.2.weight": (dim, dim), "time_embed.2.bias": (dim,)}


def full_manifest(ucfg: UNetConfig, vcfg: VolumeConfig) -> Dict[str, Tuple[int, ...]]:
    m = {}
    m.update(unet_manifest(ucfg))
    m.update(volume_manifest(vcfg))
    m.update(time_embed_manifest(vcfg.time_dim))
    return m


# ---------------------------------------------------------------------------------------------------------
# First-stage decoder: AutoencoderKL.decode = post_quant_conv + Decoder
# (ldm/models/autoencoder.py:330-333, ldm/modules/diffusionmodules/model.py:462-568)
# ---------------------------------------------------------------------------------------------------------
VAE_PREFIX = "first_stage_model."


@dataclass(frozen=True)
class VaeConfig:
    """ddconfig of morphable_diffusion.py:405-416 (decoder side)."""
    ch: int = 128
    ch_mult: Tuple[int, ...] = (1, 2, 4, 4)
    num_res_blocks: int = 2
    z_channels: int = 4
    embed_dim: int = 4
    out_ch: int = 3


def vae_decoder_manifest(cfg: VaeConfig = VaeConfig(), prefix: str = VAE_PREFIX) -> Dict[str, Tuple[int, ...]]:
    ks: Dict[str, Tuple[int, ...]] = {}

    def conv(p, cin, cout, k):
        ks[p + ".weight"] = (cout, cin, k, k)
        ks[p + ".bias"] = (cout,)

    def norm(p, c):
        ks[p + ".weight"] = (c,)
        ks[p + ".bias"] = (c,)

    def res(p, cin, cout):
        norm(p + ".norm1", cin)
        conv(p + ".conv1", cin, cout, 3)
        norm(p + ".norm2", cout)
        conv(p + ".conv2", cout, cout, 3)
        if cin != cout:
            conv(p + ".nin_shortcut", cin, cout, 1)

    d = prefix + "decoder."
    conv(prefix + "post_quant_conv", cfg.embed_dim, cfg.z_channels, 1)
    nres = len(cfg.ch_mult)
    block_in = cfg.ch * cfg.ch_mult[-1]
    conv(d + "conv_in", cfg.z_channels, block_in, 3)
    res(d + "mid.block_1", block_in, block_in)
    norm(d + "mid.attn_1.norm", block_in)
    for n in ("q", "k", "v", "proj_out"):
        conv(d + "mid.attn_1." + n, block_in, block_in, 1)
    res(d + "mid.block_2", block_in, block_in)
    for lvl in reversed(range(nres)):
        block_out = cfg.ch * cfg.ch_mult[lvl]
        for i in range(cfg.num_res_blocks + 1):
            res(d + f"up.{lvl}.block.{i}", block_in, block_out)
            block_in = block_out
        if lvl != 0:
            conv(d + f"up.{lvl}.upsample.conv", block_in, block_in, 3)
    norm(d + "norm_out", block_in)
    conv(d + "conv_out", block_in, cfg.out_ch, 3)
    return ks


def vae_encoder_manifest(cfg: VaeConfig = VaeConfig(), prefix: str = VAE_PREFIX, in_channels: int = 3) -> Dict[str, Tuple[int, ...]]:
    """AutoencoderKL.encode = Encoder (model.py:368-459, double_z) + quant_conv (autoencoder.py:302,324-328)."""
    ks: Dict[str, Tuple[int, ...]] = {}

    def conv(p, cin, cout, k):
        ks[p + ".weight"] = (cout, cin, k, k)
        ks[p + ".bias"] = (cout,)

    def norm(p, c):
        ks[p + ".weight"] = (c,)
        ks[p + ".bias"] = (c,)

    def res(p, cin, cout):
        norm(p + ".norm1", cin)
        conv(p + ".conv1", cin, cout, 3)
        norm(p + ".norm2", cout)
        conv(p + ".conv2", cout, cout, 3)
        if cin != cout:
            conv(p + ".nin_shortcut", cin, cout, 1)

    e = prefix + "encoder."
    conv(e + "conv_in", in_channels, cfg.ch, 3)
    in_mult = (1,) + tuple(cfg.ch_mult)
    block_in = cfg.ch
    for lvl in range(len(cfg.ch_mult)):
        block_in = cfg.ch * in_mult[lvl]
        block_out = cfg.ch * cfg.ch_mult[lvl]
        for i in range(cfg.num_res_blocks):
            res(e + f"down.{lvl}.block.{i}", block_in, block_out)
            block_in = block_out
        if lvl != len(cfg.ch_mult) - 1:
            conv(e + f"down.{lvl}.downsample.conv", block_in, block_in, 3)
    res(e + "mid.block_1", block_in, block_in)
    norm(e + "mid.attn_1.norm", block_in)
    for n in ("q", "k", "v", "proj_out"):
        conv(e + "mid.attn_1." + n, block_in, block_in, 1)
    res(e + "mid.block_2", block_in, block_in)
    norm(e + "norm_out", block_in)
    conv(e + "conv_out", block_in, 2 * cfg.z_channels, 3)
    conv(prefix + "quant_conv", 2 * cfg.z_channels, 2 * cfg.embed_dim, 1)
    return ks


CLIP_PREFIX = "clip_image_encoder.model.visual."


@dataclass(frozen=True)
class ClipConfig:
    """Vision tower of the CLIP model FrozenCLIPImageEmbedder loads (ldm/modules/encoders/modules.py:343-382;
    `clip.load('ViT-L/14')`, openai/CLIP clip/model.py VisionTransformer): ViT-L/14 at 224^2."""
    width: int = 1024
    layers: int = 24
    heads: int = 16
    patch: int = 14
    image: int = 224
    embed: int = 768

    @property
    def tokens(self) -> int:
        return (self.image // self.patch) ** 2 + 1


def clip_manifest(cfg: ClipConfig = ClipConfig(), prefix: str = CLIP_PREFIX) -> Dict[str, Tuple[int, ...]]:
    """state_dict keys / shapes of `clip_image_encoder.model.visual` (openai/CLIP naming)."""
    w = cfg.width
    ks: Dict[str, Tuple[int, ...]] = {
        prefix + "conv1.weight": (w, 3, cfg.patch, cfg.patch),
        prefix + "class_embedding": (w,),
        prefix + "positional_embedding": (cfg.tokens, w),
        prefix + "proj": (w, cfg.embed),
    }
    for n in ("ln_pre", "ln_post"):
        ks[prefix + n + ".weight"] = (w,)
        ks[prefix + n + ".bias"] = (w,)
    for i in range(cfg.layers):
        p = f"{prefix}transformer.resblocks.{i}."
        ks[p + "attn.in_proj_weight"] = (3 * w, w)
        ks[p + "attn.in_proj_bias"] = (3 * w,)
        ks[p + "attn.out_proj.weight"] = (w, w)
        ks[p + "attn.out_proj.bias"] = (w,)
        for n in ("ln_1", "ln_2"):
            ks[p + n + ".weight"] = (w,)
            ks[p + n + ".bias"] = (w,)
        ks[p + "mlp.c_fc.weight"] = (4 * w, w)
        ks[p + "mlp.c_fc.bias"] = (4 * w,)
        ks[p + "mlp.c_proj.weight"] = (w, 4 * w)
        ks[p + "mlp.c_proj.bias"] = (w,)
    return ks
