"""Host-side batch construction: the step just before the hot path, mirroring
generate_face.py of the reference so that a caller can go from (image, FLAME mesh) to the batch dict that
``SyncMultiviewDiffusion.sample`` consumes.  Pure tensor plumbing on the CPU (a few thousand vertices, 16 cameras):

  generate_camera_trajectory   generate_face.py:25-45   virtual 16-camera arc (positions + xyz Euler angles, degrees)
  virtual_cameras              generate_face.py:161-173 K (3x3 intrinsics in eye(4)) and RT = [R | -R p] per camera
  align_flame_vertices         generate_face.py:203-213 hard-coded similarity that maps MICA-optimised FLAME meshes
                                                        into the +-0.5 cube frame used in training
  voxelize                     generate_face.py:215-225 5 mm voxel indices (zyx), grid size rounded up to 4k
  cameras_from_dict            generate_face.py:137-139,161-164 the 'real' trajectory: intrinsics / extrinsics lists of a
                                                        camera dict (assets/facescape_test_traj.pkl, cameras.json)
  build_batch                  generate_face.py:227-243 the batch dict (leading batch dimension of 1)
  stack_batches                eval/generate_all_facescape.py:176-186 B > 1: per-sample dicts concatenated on dim 0
  load_model                   generate_face.py:71-78   YAML -> instantiate_from_config -> torch.load(ckpt)['state_dict']
"""
import math
from typing import Dict, Tuple

import numpy as np
import torch

FOCAL_256 = 1545.23757707405
FLAME_SCALE = 1.087
FLAME_POSE = (1.6811e+00, -2.6845e-02, -2.8883e-02, 8.5418e-04, -3.4041e-03, 1.0564e-02)
VOXEL = 0.005


def generate_camera_trajectory(num_cameras: int = 16):
    """Camera positions on a radius-4.5 half circle and their (x, y, z) Euler angles in degrees."""
    radius, x_angle, z_angle = 4.5, -180, 0
    positions, rotations = [], []
    for y_angle in np.linspace(-90, 90, num_cameras):
        a = np.radians(y_angle)
        positions.append((radius * np.sin(a), 0, radius * np.cos(a)))
        rotations.append((x_angle, y_angle, z_angle))
    return positions, rotations


def euler_xyz_matrix(angles_deg) -> np.ndarray:
    """scipy's Rotation.from_euler('xyz', angles, degrees=True).as_matrix(): extrinsic rotations about x, then y,
    then z, i.e. R = Rz @ Ry @ Rx."""
    ax, ay, az = (math.radians(float(a)) for a in angles_deg)
    cx, sx, cy, sy, cz, sz = math.cos(ax), math.sin(ax), math.cos(ay), math.sin(ay), math.cos(az), math.sin(az)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]], dtype=np.float64)
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]], dtype=np.float64)
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]], dtype=np.float64)
    return Rz @ Ry @ Rx


def virtual_cameras(num_cameras: int = 16, image_size: int = 256) -> Tuple[torch.Tensor, torch.Tensor]:
    """K [N,4,4] and RT [N,3,4] (world -> camera) of the virtual trajectory."""
    positions, rotations = generate_camera_trajectory(num_cameras)
    Ks, RTs = [], []
    for p, r in zip(positions, rotations):
        K = np.eye(4)
        f = FOCAL_256 * image_size / 256.0
        K[:3, :3] = np.array([[f, 0.0, image_size / 2.0], [0.0, f, image_size / 2.0], [0.0, 0.0, 1.0]])
        R = euler_xyz_matrix(r)
        RT = np.zeros((3, 4))
        RT[:3, :3] = R
        RT[:3, 3] = (-R @ np.asarray(p, dtype=np.float64).reshape(3, 1)).reshape(3)
        Ks.append(K)
        RTs.append(RT)
    return torch.tensor(np.array(Ks)).float(), torch.tensor(np.array(RTs)).float()


def cameras_from_dict(camera_dict, num_views: int = 16, views=None) -> Tuple[torch.Tensor, torch.Tensor]:
    """The 'real' camera trajectory (generate_face.py:137-139,161-164): ``camera_dict['intrinsics'][i]`` is a 3x3 K placed
    in the top-left of eye(4), ``camera_dict['extrinsics'][i]`` a 3x4 world->camera [R|t].  ``views``: the indices (or keys)
    to take, default 0..num_views-1.  Raises KeyError / IndexError like the reference if an entry is missing."""
    Ks, RTs = [], []
    for idx in (range(num_views) if views is None else views):
        K = np.eye(4)
        K[:3, :3] = np.array(camera_dict["intrinsics"][idx], dtype=np.float64)
        RT = np.array(camera_dict["extrinsics"][idx], dtype=np.float64)
        if K[:3, :3].shape != (3, 3) or RT.shape != (3, 4):
            raise ValueError(f"camera {idx}: expected 3x3 intrinsics and 3x4 extrinsics")
        Ks.append(K)
        RTs.append(RT)
    return torch.tensor(np.array(Ks)).float(), torch.tensor(np.array(RTs)).float()


def so3_exponential_map(log_rot: torch.Tensor, eps: float = 1e-4) -> torch.Tensor:
    """Rodrigues' formula, the arithmetic of pytorch3d.transforms.so3_exponential_map (generate_face.py:18,208):
    R = I + sin(t)/t K + (1 - cos t)/t^2 K^2 with K the cross-product matrix of the axis-angle vector, t = |v|
    clamped at sqrt(eps)."""
    v = log_rot.reshape(-1, 3).double()
    t = torch.clamp((v * v).sum(1), min=eps).sqrt()
    K = torch.zeros(v.shape[0], 3, 3, dtype=torch.float64)
    K[:, 0, 1], K[:, 0, 2] = -v[:, 2], v[:, 1]
    K[:, 1, 0], K[:, 1, 2] = v[:, 2], -v[:, 0]
    K[:, 2, 0], K[:, 2, 1] = -v[:, 1], v[:, 0]
    f1 = (torch.sin(t) / t)[:, None, None]
    f2 = ((1.0 - torch.cos(t)) / (t * t))[:, None, None]
    R = torch.eye(3, dtype=torch.float64)[None] + f1 * K + f2 * (K @ K)
    return R.to(log_rot.dtype)


def align_flame_vertices(verts: torch.Tensor) -> torch.Tensor:
    """MICA-optimised FLAME vertices -> the +-0.5 cube frame of the FaceScape fits."""
    v = verts.float() * FLAME_SCALE
    pose = torch.tensor(FLAME_POSE).reshape(1, -1)
    R = so3_exponential_map(pose[:, :3])[0]
    T = pose[0, 3:]
    v = (R @ v.T).T + T.reshape(-1, 3)
    v = v * 2.5
    return (torch.tensor([[1., 0., 0.], [0., 0., 1.], [0., -1., 0.]]) @ v.T).T


def voxelize(vertices: torch.Tensor):
    """coord [Nv,3] int32 (z,y,x voxel indices at 5 mm), out_sh [3] int32 (a multiple of 4), bounds [2,3]."""
    min_xyz, max_xyz = vertices.min(0).values, vertices.max(0).values
    dhw = vertices[:, [2, 1, 0]]
    min_dhw, max_dhw = min_xyz[[2, 1, 0]], max_xyz[[2, 1, 0]]
    coord = torch.round((dhw - min_dhw) / VOXEL).int()
    out_sh = torch.ceil((max_dhw - min_dhw) / VOXEL).int()
    out_sh = (out_sh | 3) + 1
    return coord, out_sh, torch.stack([min_xyz, max_xyz], 0)


def build_batch(input_image: torch.Tensor, vertices: torch.Tensor, num_views: int = 16, image_size: int = 256,
                device="cpu", cameras=None) -> Dict[str, torch.Tensor]:
    """input_image [H,W,3] in [-1,1]; vertices [Nv,3] already in the cube frame (see align_flame_vertices).
    ``cameras``: (K [N,4,4], RT [N,3,4]) of a 'real' trajectory (cameras_from_dict); default: the virtual arc."""
    K, RT = virtual_cameras(num_views, image_size) if cameras is None else cameras
    if K.shape[0] != num_views or RT.shape[0] != num_views:
        raise ValueError("cameras must hold num_views entries")
    coord, out_sh, bounds = voxelize(vertices)
    d = {"target_image": input_image[None].repeat(num_views, 1, 1, 1), "input_image": input_image,
         "input_elevation": torch.zeros(1), "input_azimuth": torch.zeros(1),
         "target_elevation": torch.zeros(num_views), "target_azimuth": torch.zeros(num_views),
         "target_K": K, "target_RT": RT, "vertices": vertices.float(), "out_sh": out_sh, "coord": coord, "bounds": bounds}
    return {k: v.unsqueeze(0).to(device) for k, v in d.items()}


def stack_batches(samples) -> Dict[str, torch.Tensor]:
    """eval/generate_all_facescape.py:176-186: per-sample batch dicts (leading dimension 1) concatenated into one batch of
    B samples.  All samples must share the vertex count (a fixed-topology mesh), as torch.concat requires there too."""
    return {k: torch.cat([s[k] for s in samples], 0) for k in samples[0]}


def load_model(cfg, ckpt, device="cuda:0", **overrides):
    """generate_face.py:71-78 ``load_model``: read the YAML (configs/facescape.yaml layout), instantiate ``config.model``
    through the ``target:`` reflection, torch.load the checkpoint and load ``ckpt['state_dict']`` with strict=False.
    ``cfg`` is a path or an already parsed dict; ``overrides`` are extra constructor kwargs (e.g. workspace_gb)."""
    from .model import instantiate_from_config
    if isinstance(cfg, (str, bytes)) or hasattr(cfg, "__fspath__"):
        import yaml
        with open(cfg) as f:
            cfg = yaml.safe_load(f)
    mc = dict(cfg["model"])
    mc["params"] = dict(mc.get("params", {}), device=device, **overrides)
    model = instantiate_from_config(mc)
    print(f"loading model from {ckpt} ...")
    state = torch.load(ckpt, map_location="cpu")
    model.load_state_dict(state["state_dict"], strict=False)
    return model.eval()


def views_to_uint8(x_sample: torch.Tensor, input_image: torch.Tensor) -> np.ndarray:
    """The image strip generate_face.py:244-252 saves: the input view followed by the N generated views side by side,
    samples stacked vertically.  x_sample [B,N,3,H,W] in [-1,1] (``model.sample``), input_image [B,H,W,3] -> uint8
    [B*H, (N+1)*W, 3]."""
    x = torch.cat([input_image.unsqueeze(1).permute(0, 1, 4, 2, 3).to(x_sample), x_sample], 1)
    x = (torch.clamp(x, -1.0, 1.0) + 1.0) * 0.5
    x = (x.permute(0, 1, 3, 4, 2).cpu().numpy() * 255).astype(np.uint8)
    rows = np.concatenate([x[:, i] for i in range(x.shape[1])], 2)
    return np.concatenate(list(rows), 0)


def save_image_grid(x_sample: torch.Tensor, batch, path: str):
    """SyncMultiviewDiffusion.log_image (morphable_diffusion.py:589-599): per sample the input view followed by the N
    generated views, samples stacked vertically, written as one image file."""
    from PIL import Image
    Image.fromarray(views_to_uint8(x_sample, batch["input_image"])).save(path)


def neus2_transform(Ks: torch.Tensor, RTs: torch.Tensor, image_size: int = 256) -> Dict:
    """The ``transform.json`` dictionary of generate_face.py:145-154,173-186 (NeuS2 input): per view the camera-to-world
    matrix with the y and z axes flipped (OpenCV -> OpenGL camera) and the 3x3 intrinsics."""
    d = {"w": image_size, "h": image_size, "aabb_scale": 1.0, "scale": 1.0, "offset": [0.5, 0.5, 0.5], "frames": []}
    for i in range(RTs.shape[0]):
        E = np.eye(4)
        E[:3, :4] = RTs[i].double().cpu().numpy()
        c2w = np.linalg.inv(E)
        c2w[:, 1] *= -1
        c2w[:, 2] *= -1
        d["frames"].append({"file_path": f"images/{str(i).zfill(2)}.png", "transform_matrix": c2w.tolist(),
                            "intrinsic_matrix": Ks[i, :3, :3].double().cpu().numpy().tolist()})
    return d


def neus2_view_bgra(strip: np.ndarray, idx: int, image_size: int = 256) -> np.ndarray:
    """generate_face.py:256-261: view ``idx`` of the first sample's strip as BGRA, alpha = 0 where the pixel is
    near-white background (all channels > 240)."""
    img = strip[:image_size, idx * image_size:(idx + 1) * image_size, :]
    out = np.zeros((image_size, image_size, 4))
    out[:, :, :3] = img[:, :, ::-1]
    out[:, :, 3] = (~np.all(img > 240, axis=-1)).astype(np.float64) * 255  # 255 = foreground
    return out
