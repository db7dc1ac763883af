"""Synthetic inputs with the reference's batch-dict schema (generate_face.py:227-241).

No licensed data (FaceScape / THuman / FLAME template) is available, so cameras, meshes and latents are
generated from fixed seeds.  The arithmetic that turns a mesh into ``coord/out_sh/bounds`` follows
generate_face.py:211-225 (= ldm/data/facescape.py:165-180); the virtual camera arc follows
generate_face.py:25-45,166-173; the orthographic intrinsics follow assets/thuman_meta.pkl as probed in
SURVEY.md section 8(d).
"""
import math
from typing import Dict

import torch

VOXEL = 0.005


def camera_arc(num_views: int = 16, radius: float = 4.5, focal: float = 1545.23757707405, center: float = 128.0):
    """Virtual trajectory: yaw in linspace(-90, 90, N) deg, R = euler_xyz(-180, yaw, 0), t = -R @ position.
    Returns K [N,4,4] (3x3 intrinsics in the top-left of eye(4)) and RT [N,3,4] world->camera."""
    Ks, RTs = [], []
    cx, sx = math.cos(math.radians(-180.0)), math.sin(math.radians(-180.0))
    Rx = torch.tensor([[1.0, 0.0, 0.0], [0.0, cx, -sx], [0.0, sx, cx]], dtype=torch.float64)
    for i in range(num_views):
        yaw = -90.0 + 180.0 * i / (num_views - 1) if num_views > 1 else 0.0
        a = math.radians(yaw)
        Ry = torch.tensor([[math.cos(a), 0.0, math.sin(a)], [0.0, 1.0, 0.0], [-math.sin(a), 0.0, math.cos(a)]],
                          dtype=torch.float64)
        R = Ry @ Rx  # extrinsic x then y (z = 0)
        pos = torch.tensor([radius * math.sin(a), 0.0, radius * math.cos(a)], dtype=torch.float64)
        RT = torch.cat([R, (-R @ pos)[:, None]], 1)
        K = torch.eye(4, dtype=torch.float64)
        K[0, 0] = K[1, 1] = focal
        K[0, 2] = K[1, 2] = center
        Ks.append(K)
        RTs.append(RT)
    return torch.stack(Ks).float(), torch.stack(RTs).float()


def ortho_cameras(num_views: int = 16, radius: float = 1.5, scale: float = 1.6667):
    """THuman-style rig: orthographic 4x4 K = diag(s, s, s, 1) and a full azimuth ring of poses."""
    Ks, RTs = [], []
    for i in range(num_views):
        a = 2.0 * math.pi * i / num_views
        Ry = torch.tensor([[math.cos(a), 0.0, math.sin(a)], [0.0, 1.0, 0.0], [-math.sin(a), 0.0, math.cos(a)]],
                          dtype=torch.float64)
        R = Ry.t()
        pos = torch.tensor([radius * math.sin(a), 0.0, radius * math.cos(a)], dtype=torch.float64)
        RTs.append(torch.cat([R, (-R @ pos)[:, None]], 1))
        K = torch.eye(4, dtype=torch.float64)
        K[0, 0] = K[1, 1] = K[2, 2] = scale
        Ks.append(K)
    return torch.stack(Ks).float(), torch.stack(RTs).float()


from .batch import voxelize  # noqa: E402  (generate_face.py:211-225; one implementation, in batch.py)


def ellipsoid_mesh(num_vertices: int = 5023, seed: int = 1, radii=(0.22, 0.28, 0.25), dedup: bool = True):
    """FLAME-sized point set on an ellipsoid inside the +-0.5 cube.  ``dedup`` drops vertices whose voxel
    index collides with an earlier vertex (spconv leaves duplicate coordinates undefined)."""
    g = torch.Generator().manual_seed(seed)
    n = torch.randn(num_vertices, 3, generator=g)
    n = n / n.norm(dim=1, keepdim=True)
    v = n * torch.tensor(radii)
    if dedup:
        coord, out_sh, _ = voxelize(v)
        key = (coord[:, 0].long() * 4096 + coord[:, 1].long()) * 4096 + coord[:, 2].long()
        seen, keep = set(), []
        for i, k in enumerate(key.tolist()):
            if k not in seen:
                seen.add(k)
                keep.append(i)
        v = v[keep]
        # bounds are unchanged only if the extreme vertices survive; recompute from the kept set
    return v.contiguous()


def make_batch(num_views: int = 16, projection: str = "perspective", num_vertices: int = 5023,
               mesh_seed: int = 1, batch_size: int = 1, radii=(0.22, 0.28, 0.25),
               image_size: int = 256) -> Dict[str, torch.Tensor]:
    """Batch dict with the keys the hot path reads (morphable_diffusion.py:205-252,281-296,389-392).
    ``image_size`` scales the pinhole intrinsics (focal and principal point are given for 256-pixel renders)."""
    if projection == "perspective":
        K, RT = camera_arc(num_views, focal=1545.23757707405 * image_size / 256.0, center=image_size / 2.0)
    else:
        K, RT = ortho_cameras(num_views)
    verts = ellipsoid_mesh(num_vertices, mesh_seed, radii)
    # iterate: de-duplication can move the bounding box, which shifts every voxel index
    for _ in range(4):
        coord, out_sh, bounds = voxelize(verts)
        key = (coord[:, 0].long() * 4096 + coord[:, 1].long()) * 4096 + coord[:, 2].long()
        if torch.unique(key).numel() == key.numel():
            break
        seen, keep = set(), []
        for i, k in enumerate(key.tolist()):
            if k not in seen:
                seen.add(k)
                keep.append(i)
        verts = verts[keep].contiguous()
    B = batch_size
    rep = lambda t: t[None].repeat(B, *([1] * t.dim()))
    zeros = torch.zeros(B, num_views)
    return {"input_elevation": torch.zeros(B, 1), "input_azimuth": torch.zeros(B, 1),
            "target_elevation": zeros.clone(), "target_azimuth": zeros.clone(),
            "target_K": rep(K), "target_RT": rep(RT), "vertices": rep(verts),
            "coord": rep(coord), "out_sh": rep(out_sh), "bounds": rep(bounds)}


def make_latents(num_views: int = 16, latent: int = 32, seed: int = 6033, batch_size: int = 1):
    """x_T (seed 6033 = generate_face.sh:34), input latent ~ randn * 0.18215, CLIP token ~ randn."""
    g = torch.Generator().manual_seed(seed)
    x_T = torch.randn(batch_size, num_views, 4, latent, latent, generator=g)
    x_in = torch.randn(batch_size, 4, latent, latent, generator=g) * 0.18215
    clip = torch.randn(batch_size, 1, 768, generator=g)
    return x_T, x_in, clip
