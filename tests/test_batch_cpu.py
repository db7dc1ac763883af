"""Host-side batch construction against independent implementations: scipy's Rotation (which
the reference calls, generate_face.py:170,208) for the camera rotations and the axis-angle exponential, and the
virtual-camera rig the goldens were generated with."""
import numpy as np
import pytest
import torch
from scipy.spatial.transform import Rotation as Rot

from morphablediffusion_amd import batch as B
from morphablediffusion_amd import synthetic


def test_camera_trajectory_matches_scipy_and_synthetic_rig():
    pos, rot = B.generate_camera_trajectory(16)
    assert len(pos) == 16 and rot[0] == (-180, -90.0, 0) and abs(pos[0][0] + 4.5) < 1e-12
    K, RT = B.virtual_cameras(16)
    for i in range(16):
        R = Rot.from_euler("xyz", np.array(rot[i]), True).as_matrix()
        assert np.allclose(RT[i, :, :3].numpy(), R, atol=1e-6)
        assert np.allclose(RT[i, :, 3].numpy(), -R @ np.array(pos[i]), atol=1e-5)
    Ks, RTs = synthetic.camera_arc(16)  # the rig behind tests/golden/*
    assert torch.allclose(K, Ks, atol=1e-4) and torch.allclose(RT, RTs, atol=1e-5)
    K5, _ = B.virtual_cameras(4, image_size=512)
    assert abs(K5[0, 0, 0].item() - 2 * B.FOCAL_256) < 1e-3 and K5[0, 0, 2].item() == 256.0


def test_so3_exponential_map_matches_scipy():
    g = torch.Generator().manual_seed(0)
    v = torch.randn(8, 3, generator=g)
    v[0] = torch.tensor(B.FLAME_POSE[:3])
    R = B.so3_exponential_map(v)
    want = Rot.from_rotvec(v.numpy().astype(np.float64)).as_matrix()
    assert np.allclose(R.numpy(), want, atol=1e-5)


def test_align_voxelize_and_batch_schema():
    g = torch.Generator().manual_seed(1)
    raw = torch.randn(500, 3, generator=g) * 0.03
    v = B.align_flame_vertices(raw)
    # similarity: pairwise distances scale by 1.087 * 2.5
    d0 = (raw[0] - raw[1]).norm() * B.FLAME_SCALE * 2.5
    assert abs((v[0] - v[1]).norm() - d0) < 1e-5
    coord, out_sh, bounds = B.voxelize(v)
    c2, s2, b2 = synthetic.voxelize(v)
    assert torch.equal(coord, c2) and torch.equal(out_sh, s2) and torch.equal(bounds, b2)
    assert coord.dtype == torch.int32 and (out_sh % 4 == 0).all() and (coord.max(0).values < out_sh).all()
    img = torch.zeros(256, 256, 3)
    d = B.build_batch(img, v)
    ref = synthetic.make_batch(16, "perspective", 100)
    assert set(ref) <= set(d)
    for k in ref:
        assert d[k].dim() == ref[k].dim() and d[k].dtype == ref[k].dtype, k
    assert d["target_K"].shape == (1, 16, 4, 4) and d["target_RT"].shape == (1, 16, 3, 4)
    assert d["input_image"].shape == (1, 256, 256, 3) and d["target_image"].shape == (1, 16, 256, 256, 3)


def test_output_strip_and_neus2_export():
    """generate_face.py:145-186,244-261: image strip, transform.json contents, BGRA views."""
    g = torch.Generator().manual_seed(2)
    xs = torch.rand(2, 3, 3, 256, 256, generator=g) * 3.0 - 1.5        # out-of-range values are clamped
    inp = torch.rand(2, 256, 256, 3, generator=g) * 2.0 - 1.0
    strip = B.views_to_uint8(xs, inp)
    assert strip.shape == (512, 4 * 256, 3) and strip.dtype == np.uint8
    want = ((inp[1].clamp(-1, 1) + 1) * 0.5 * 255).numpy().astype(np.uint8)
    assert np.array_equal(strip[256:, :256], want)
    v = ((xs[0, 2].clamp(-1, 1) + 1) * 0.5).permute(1, 2, 0).numpy() * 255
    assert np.array_equal(strip[:256, 3 * 256:], v.astype(np.uint8))
    Ks, RTs = B.virtual_cameras(16)
    d = B.neus2_transform(Ks, RTs)
    assert len(d["frames"]) == 16 and d["frames"][3]["file_path"] == "images/03.png" and d["offset"] == [0.5, 0.5, 0.5]
    c2w = np.array(d["frames"][5]["transform_matrix"])
    # undo the axis flip: the matrix inverts the extrinsics
    c2w[:, 1] *= -1
    c2w[:, 2] *= -1
    E = np.eye(4)
    E[:3, :4] = RTs[5].numpy()
    assert np.allclose(c2w @ E, np.eye(4), atol=1e-5)
    assert np.allclose(np.array(d["frames"][5]["intrinsic_matrix"]), Ks[5, :3, :3].numpy())
    strip[:256, 256:512] = 255
    strip[10, 300] = (0, 10, 250)
    bgra = B.neus2_view_bgra(strip, 1)
    assert bgra.shape == (256, 256, 4) and bgra[0, 0, 3] == 0 and bgra[10, 300 - 256, 3] == 255
    assert tuple(bgra[10, 300 - 256, :3]) == (250, 10, 0)


def test_real_camera_dict_and_stacked_batches():
    """generate_face.py:137-139,161-164 ('real' trajectory from a camera dict) and the eval driver's B > 1 batch
    (eval/generate_all_facescape.py:176-186)."""
    from morphablediffusion_amd import batch as BT
    K, RT = BT.virtual_cameras(4)
    d = {"intrinsics": [K[i, :3, :3].tolist() for i in range(4)], "extrinsics": [RT[i].tolist() for i in range(4)]}
    K2, RT2 = BT.cameras_from_dict(d, 4)
    assert torch.equal(K2, K) and torch.equal(RT2, RT)
    K3, _ = BT.cameras_from_dict(d, 2, views=[3, 1])
    assert torch.equal(K3, K[[3, 1]])
    v = torch.rand(50, 3) * 0.4 - 0.2
    img = torch.zeros(256, 256, 3)
    s0 = BT.build_batch(img, v, num_views=4)
    s1 = BT.build_batch(img, v * 0.9, num_views=4, cameras=(K2 * 1.0, RT2 + 0.01))
    both = BT.stack_batches([s0, s1])
    assert both["target_RT"].shape == (2, 4, 3, 4) and both["coord"].shape == (2, 50, 3)
    assert torch.equal(both["target_RT"][1], RT + 0.01) and torch.equal(both["vertices"][0], v)
    import pytest
    with pytest.raises(ValueError):
        BT.build_batch(img, v, num_views=4, cameras=(K[:3], RT[:3]))


def test_camera_trajectory_vs_reference_golden():
    """tests/golden/cameras.npz: generate_camera_trajectory extracted from the reference's generate_face.py (:25-45) and the
    RT matrices its main loop builds with scipy (:166-173), for 16 and 8 cameras."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "cameras.npz"))
    for n in (16, 8):
        pos, rot = B.generate_camera_trajectory(n)
        assert np.allclose(np.array(pos), g[f"positions{n}"], atol=1e-12) and np.allclose(np.array(rot), g[f"rotations{n}"], atol=1e-12)
        K, RT = B.virtual_cameras(n)
        assert np.allclose(RT.numpy(), g[f"RT{n}"], atol=1e-6)


def test_cli_flags_mesh_readers_and_image_loading(tmp_path):
    """python -m morphablediffusion_amd.generate_face: the reference's flags with its defaults (generate_face.py:91-106), mesh
    readers (what trimesh.load(process=False).vertices returns) and the input-image preparation (process_im :79-88)."""
    import struct
    from PIL import Image
    from morphablediffusion_amd import generate_face as GF
    fl = GF.build_parser().parse_args(["--input_img", "a/in.png", "--exp_img", "b/kiss.jpg", "--mesh", "m.ply", "--output_dir", "o"])
    assert (fl.cfg, fl.ckpt, fl.cfg_scale, fl.batch_view_num, fl.seed, fl.sampler, fl.sample_steps, fl.camera_trajectory,
            fl.prepare_neus2_data) == ("configs/facescape.yaml", "ckpt/facescape_flame.ckpt", 2.0, 8, 6033, "ddim", 50, "virtual", False)
    with pytest.raises(SystemExit):
        GF.build_parser().parse_args(["--input_img", "a.png"])  # --exp_img / --mesh / --output_dir are required
    v = np.random.RandomState(0).randn(37, 3)
    (tmp_path / "m.obj").write_text("# test\n" + "".join(f"v {a:.9f} {b:.9f} {c:.9f}\n" for a, b, c in v) + "f 1 2 3\n")
    assert np.allclose(GF.read_mesh_vertices(tmp_path / "m.obj"), v, atol=1e-8)
    hdr = "ply\nformat {}\nelement vertex 37\nproperty float x\nproperty float y\nproperty float z\nproperty uchar red\n" \
          "element face 1\nproperty list uchar int vertex_indices\nend_header\n"
    (tmp_path / "a.ply").write_text(hdr.format("ascii 1.0") + "".join(f"{a:.9f} {b:.9f} {c:.9f} 7\n" for a, b, c in v) + "3 0 1 2\n")
    assert np.allclose(GF.read_mesh_vertices(tmp_path / "a.ply"), v, atol=1e-8)
    for fmt, e in (("binary_little_endian 1.0", "<"), ("binary_big_endian 1.0", ">")):
        with open(tmp_path / "b.ply", "wb") as f:
            f.write(hdr.format(fmt).encode())
            for a, b, c in v:
                f.write(struct.pack(e + "fffB", a, b, c, 7))
            f.write(struct.pack(e + "Biii", 3, 0, 1, 2))
        assert np.allclose(GF.read_mesh_vertices(tmp_path / "b.ply"), v.astype(np.float32), atol=0)
    with pytest.raises(ValueError):
        GF.read_mesh_vertices(tmp_path / "m.stl")
    # RGBA input: composited on white with its alpha; RGB input: used as is; both resized to 256 (bicubic), [-1,1], HWC
    rgba = np.zeros((64, 64, 4), np.uint8)
    rgba[16:48, 16:48] = (255, 0, 0, 255)
    Image.fromarray(rgba, "RGBA").save(tmp_path / "in.png")
    im = GF.load_input_image(tmp_path / "in.png")
    assert im.shape == (256, 256, 3) and im.dtype == torch.float32
    assert torch.allclose(im[2, 2], torch.ones(3)) and torch.allclose(im[128, 128], torch.tensor([1.0, -1.0, -1.0]))
    Image.fromarray(rgba[:, :, :3]).save(tmp_path / "rgb.jpg")
    assert GF.load_input_image(tmp_path / "rgb.jpg").shape == (256, 256, 3)
