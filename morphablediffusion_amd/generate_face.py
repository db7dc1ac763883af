"""``python -m morphablediffusion_amd.generate_face`` -- the reference's inference entry script (generate_face.py:90-262) on the
MI355X engine: same flags, same batch construction, same outputs.

  --input_img IMG --exp_img IMG --mesh MESH.{ply,obj} --cfg configs/facescape.yaml --ckpt CKPT --output_dir DIR
  [--cfg_scale 2.0] [--batch_view_num 8] [--seed 6033] [--sampler ddim] [--sample_steps 50]
  [--camera_trajectory virtual|real] [--prepare_neus2_data]

writes ``<output_dir>/<input>_<exp>.png`` (the input view followed by the 16 generated views, generate_face.py:244-253) and, with
--prepare_neus2_data, ``<output_dir>/neus2_data/<input>_<exp>/{transform.json, images/00..15.png}`` (:145-192,255-262).

Differences, all outside the denoising path and stated here:
  * the reference runs carvekit's BackgroundRemoval (a third-party segmentation network, generate_face.py:46-69) on the input
    image; it is not part of this repository.  An input with an alpha channel is used as the matte (exactly what the reference
    does with carvekit's RGBA output, process_im :79-88); an input without one is taken to be already on a white background
    (alpha = 1), which is what the reference's demo inputs are after matting;
  * meshes are read by a small PLY / OBJ vertex reader instead of trimesh (``process=False``: vertices as stored);
  * --camera_file names the pickle / JSON of the 'real' trajectory (default: the reference's ./assets/facescape_test_traj.pkl);
  * --exp_img only names the output, as in the reference (the expression enters through the mesh).
"""
import argparse
import json
import os
import pickle
import struct
from pathlib import Path

import numpy as np
import torch

from . import batch as B

NUM_VIEWS = 16  # hard-wired in the reference (generate_face.py:141,156,256)


def read_mesh_vertices(path) -> np.ndarray:
    """Vertex positions [Nv,3] (float64) of a Wavefront OBJ or a PLY file (ascii, binary little / big endian), in file order --
    what ``trimesh.load(path, process=False).vertices`` returns (generate_face.py:200)."""
    path = str(path)
    if path.lower().endswith(".obj"):
        vs = []
        with open(path) as f:
            for line in f:
                if line.startswith("v "):
                    p = line.split()
                    vs.append([float(p[1]), float(p[2]), float(p[3])])
        if not vs:
            raise ValueError(f"{path}: no vertices")
        return np.asarray(vs, dtype=np.float64)
    if not path.lower().endswith(".ply"):
        raise ValueError(f"{path}: expected a .ply or .obj mesh")
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError(f"{path}: not a PLY file")
        fmt, nverts, props, in_vertex = None, 0, [], False
        while True:
            line = f.readline()
            if not line:
                raise ValueError(f"{path}: truncated PLY header")
            t = line.decode("ascii", "replace").split()
            if not t:
                continue
            if t[0] == "format":
                fmt = t[1]
            elif t[0] == "element":
                in_vertex = t[1] == "vertex"
                if in_vertex:
                    nverts = int(t[2])
            elif t[0] == "property" and in_vertex:
                if t[1] == "list":
                    raise ValueError(f"{path}: list property on vertices is not supported")
                props.append((t[2], t[1]))
            elif t[0] == "end_header":
                break
        names = [p[0] for p in props]
        if not all(a in names for a in "xyz"):
            raise ValueError(f"{path}: vertex element has no x/y/z")
        if fmt == "ascii":
            out = np.empty((nverts, 3))
            ix = [names.index(a) for a in "xyz"]
            for i in range(nverts):
                p = f.readline().split()
                out[i] = [float(p[j]) for j in ix]
            return out
        codes = {"char": "b", "int8": "b", "uchar": "B", "uint8": "B", "short": "h", "int16": "h", "ushort": "H", "uint16": "H",
                 "int": "i", "int32": "i", "uint": "I", "uint32": "I", "float": "f", "float32": "f", "double": "d", "float64": "d"}
        end = "<" if fmt == "binary_little_endian" else ">"
        rec = struct.Struct(end + "".join(codes[p[1]] for p in props))
        raw = f.read(rec.size * nverts)
        if len(raw) != rec.size * nverts:
            raise ValueError(f"{path}: truncated vertex data")
        ix = [names.index(a) for a in "xyz"]
        return np.asarray([[r[j] for j in ix] for r in rec.iter_unpack(raw)], dtype=np.float64)


def load_input_image(path, size=256) -> torch.Tensor:
    """generate_face.py:117-123 + process_im (:79-88): RGBA -> white background, bicubic resize to size x size, [-1,1], HWC."""
    from PIL import Image
    im = Image.open(path)
    rgba = np.asarray(im.convert("RGBA")).astype(np.float32) / 255.0
    mask = rgba[:, :, 3:]
    rgb = rgba[:, :, :3] * mask + 1 - mask
    im = Image.fromarray(np.uint8(rgb * 255.0)).convert("RGB").resize((size, size), resample=Image.BICUBIC)
    return torch.from_numpy(np.asarray(im).astype(np.float32) / 255.0) * 2.0 - 1.0


def load_camera_dict(path):
    if str(path).lower().endswith(".json"):
        with open(path) as f:
            return json.load(f)
    with open(path, "rb") as f:
        return pickle.load(f)


def build_parser():
    p = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    p.add_argument("--input_img", type=str, required=True)
    p.add_argument("--exp_img", type=str, required=True)
    p.add_argument("--mesh", type=str, required=True)
    p.add_argument("--cfg", type=str, default="configs/facescape.yaml")
    p.add_argument("--ckpt", type=str, default="ckpt/facescape_flame.ckpt")
    p.add_argument("--output_dir", type=str, required=True)
    p.add_argument("--cfg_scale", type=float, default=2.0)
    p.add_argument("--batch_view_num", type=int, default=8)
    p.add_argument("--seed", type=int, default=6033)
    p.add_argument("--sampler", type=str, default="ddim")
    p.add_argument("--sample_steps", type=int, default=50)
    p.add_argument("--camera_trajectory", type=str, default="virtual", choices=["real", "virtual"])
    p.add_argument("--prepare_neus2_data", action="store_true")
    p.add_argument("--camera_file", type=str, default="./assets/facescape_test_traj.pkl",
                   help="not a reference flag: the camera dict of --camera_trajectory real (the reference hard-codes this path)")
    p.add_argument("--device", type=str, default="cuda:0", help="not a reference flag")
    return p


def run(flags, model=None):
    """The body of generate_face.py:main (:107-262).  ``model``: an already loaded SyncMultiviewDiffusion (tests); default:
    batch.load_model(flags.cfg, flags.ckpt).  Returns (strip uint8 [256, 17*256, 3], output path)."""
    from PIL import Image
    from .model import SyncDDIMSampler, SyncMultiviewDiffusion
    img_name = flags.input_img.split("/")[-1].split(".")[0]
    exp_name = flags.exp_img.split("/")[-1].split(".")[0]
    torch.random.manual_seed(flags.seed)
    input_img = load_input_image(flags.input_img)
    if model is None:
        model = B.load_model(flags.cfg, flags.ckpt, device=flags.device)
    assert isinstance(model, SyncMultiviewDiffusion)
    Path(flags.output_dir).mkdir(exist_ok=True, parents=True)
    if flags.sampler != "ddim":
        raise NotImplementedError
    sampler = SyncDDIMSampler(model, flags.sample_steps, latent_size=model.image_size // 8)
    if flags.camera_trajectory == "real":
        cams = B.cameras_from_dict(load_camera_dict(flags.camera_file), NUM_VIEWS)
    else:
        cams = B.virtual_cameras(NUM_VIEWS, 256)
    verts = B.align_flame_vertices(torch.from_numpy(read_mesh_vertices(flags.mesh)).float())
    data = B.build_batch(input_img, verts, num_views=NUM_VIEWS, image_size=256, device=model.device, cameras=cams)
    neus2_root = None
    if flags.prepare_neus2_data:
        neus2_root = os.path.join(flags.output_dir, "neus2_data", f"{img_name}_{exp_name}")
        os.makedirs(os.path.join(neus2_root, "images"), exist_ok=True)
        with open(os.path.join(neus2_root, "transform.json"), "w") as f:
            json.dump(B.neus2_transform(cams[0], cams[1], 256), f, indent=4)
    x_sample = model.sample(sampler, data, flags.cfg_scale, flags.batch_view_num)
    strip = B.views_to_uint8(x_sample, data["input_image"])
    output_fn = Path(flags.output_dir) / f"{img_name}_{exp_name}.png"
    Image.fromarray(strip).save(output_fn)
    if neus2_root:
        # generate_face.py:255-262 slices idx*256 of the strip for idx in range(16): view 0 of the export is the INPUT view and
        # the last generated view is dropped -- reproduced as is
        for idx in range(NUM_VIEWS):
            bgra = B.neus2_view_bgra(strip, idx, 256)
            rgba = np.concatenate([bgra[:, :, 2::-1], bgra[:, :, 3:]], -1).astype(np.uint8)  # cv2.imwrite(BGRA) == PNG(RGBA)
            Image.fromarray(rgba, "RGBA").save(os.path.join(neus2_root, f"images/{str(idx).zfill(2)}.png"))
    return strip, str(output_fn)


def main(argv=None):
    run(build_parser().parse_args(argv))


if __name__ == "__main__":
    main()
