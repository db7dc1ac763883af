"""Import harness for the read-only reference at /root/reference (build container ONLY).

This file is test infrastructure: it is used by tools/make_goldens.py to run the reference's own
Python on CPU and dump golden vectors into tests/golden/.  Nothing under tests/ -m gpu, bench.py or
__graft_entry__.smoke() imports it, and /root/reference does not exist on the GPU box.

The reference's non-arithmetic dependencies that are absent from this image are replaced by inert stubs
(torchvision, cv2, omegaconf, pytorch_lightning, skimage, trimesh, clip, taming).  Two dependencies carry
arithmetic and are restated here from their documented behaviour:

* kornia.create_meshgrid(H, W, normalized_coordinates=False) -> [1,H,W,2] with [...,0]=x in 0..W-1,
  [...,1]=y in 0..H-1  (used at ldm/models/diffusion/utils.py:103,117).
* spconv (requirements.txt:18, version un-pinned, not vendored): SubMConv3d / SparseConv3d /
  SparseConvTensor / SparseSequential, emulated densely:
    - SparseConvTensor(features[Nv,C], indices[Nv,4]=(b,z,y,x), spatial_shape, batch) scatters rows into a
      dense [B,C,D,H,W] tensor plus an occupancy mask (last write wins on duplicate coordinates, so
      callers de-duplicate coordinates first: spconv leaves duplicates undefined);
    - SubMConv3d(k=3, bias=False): conv3d(pad=1) evaluated only at active sites (output * mask);
    - SparseConv3d(k=3, s=2, p=1, bias=False): conv3d(stride 2, pad 1); a site is active iff its
      receptive field holds an active input;
    - BatchNorm1d / ReLU inside SparseSequential act on active rows only (re-masked afterwards);
    - .dense() returns the [B,C,D,H,W] tensor with inactive sites zero.
  PARITY UNPINNED for this layer: the reference ships no test for it and spconv itself cannot be
  imported here; the emulation follows spconv's documented semantics.
"""
import sys
import types

import torch
import torch.nn as nn
import torch.nn.functional as F

REFERENCE_ROOT = "/root/reference"


def _mod(name):
    m = types.ModuleType(name)
    sys.modules[name] = m
    return m


class _DenseSparseTensor:
    def __init__(self, dense, mask):
        self.dense_feats = dense  # [B,C,D,H,W]
        self.mask = mask  # [B,1,D,H,W] float {0,1}

    def dense(self):
        return self.dense_feats


def _make_sparse_conv_tensor(features, indices, spatial_shape, batch_size):
    D, H, W = [int(s) for s in spatial_shape]
    C = features.shape[1]
    dense = torch.zeros(batch_size, C, D, H, W, dtype=features.dtype, device=features.device)
    mask = torch.zeros(batch_size, 1, D, H, W, dtype=features.dtype, device=features.device)
    idx = indices.long()
    dense[idx[:, 0], :, idx[:, 1], idx[:, 2], idx[:, 3]] = features
    mask[idx[:, 0], 0, idx[:, 1], idx[:, 2], idx[:, 3]] = 1.0
    return _DenseSparseTensor(dense, mask)


class _SubMConv3d(nn.Module):
    def __init__(self, cin, cout, k, bias=False, indice_key=None):
        super().__init__()
        assert not bias
        self.k = k
        self.weight = nn.Parameter(torch.randn(cout, cin, k, k, k) * (1.0 / (cin * k ** 3) ** 0.5))

    def forward(self, x):
        y = F.conv3d(x.dense_feats, self.weight, padding=self.k // 2) * x.mask
        return _DenseSparseTensor(y, x.mask)


class _SparseConv3d(nn.Module):
    def __init__(self, cin, cout, k, stride, padding=0, bias=False, indice_key=None):
        super().__init__()
        assert not bias
        self.k, self.s, self.p = k, stride, padding
        self.weight = nn.Parameter(torch.randn(cout, cin, k, k, k) * (1.0 / (cin * k ** 3) ** 0.5))

    def forward(self, x):
        y = F.conv3d(x.dense_feats, self.weight, stride=self.s, padding=self.p)
        ones = torch.ones(1, 1, self.k, self.k, self.k, dtype=y.dtype)
        m = (F.conv3d(x.mask, ones, stride=self.s, padding=self.p) > 0).to(y.dtype)
        return _DenseSparseTensor(y * m, m)


class _SparseSequential(nn.Sequential):
    def forward(self, x):
        for layer in self:
            if isinstance(layer, (_SubMConv3d, _SparseConv3d)):
                x = layer(x)
            elif isinstance(layer, nn.BatchNorm1d):
                d = x.dense_feats
                B, C = d.shape[:2]
                act = x.mask.expand_as(d) > 0
                rows = d.permute(0, 2, 3, 4, 1)[x.mask[:, 0] > 0]  # [Nactive, C]
                rows = layer(rows)
                out = torch.zeros_like(d).permute(0, 2, 3, 4, 1).contiguous()
                out[x.mask[:, 0] > 0] = rows
                x = _DenseSparseTensor(out.permute(0, 4, 1, 2, 3).contiguous(), x.mask)
            else:
                x = _DenseSparseTensor(layer(x.dense_feats) * x.mask, x.mask)
        return x


def install_stubs():
    if "ldm" in sys.modules:
        return
    sys.dont_write_bytecode = True  # never write __pycache__ into the read-only reference tree
    for name in ["torchvision", "torchvision.transforms", "cv2", "skimage", "skimage.io", "trimesh", "clip",
                 "taming", "taming.modules", "taming.modules.vqvae", "taming.modules.vqvae.quantize"]:
        _mod(name)
    sys.modules["skimage.io"].imsave = lambda *a, **k: None
    sys.modules["skimage.io"].imread = lambda *a, **k: None
    for name in ["PIL", "PIL.Image", "PIL.ImageDraw", "PIL.ImageFont", "matplotlib", "matplotlib.pyplot"]:
        try:
            __import__(name)
        except Exception:
            _mod(name)
    sys.modules["taming.modules.vqvae.quantize"].VectorQuantizer2 = object
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]

    oc = _mod("omegaconf")
    ocl = _mod("omegaconf.listconfig")

    class ListConfig(list):
        pass

    ocl.ListConfig = ListConfig
    oc.listconfig = ocl

    pl = _mod("pytorch_lightning")

    class LightningModule(nn.Module):
        @property
        def device(self):
            return next(self.parameters()).device

        def log(self, *a, **k):
            pass

    pl.LightningModule = LightningModule

    kornia = _mod("kornia")

    def create_meshgrid(H, W, normalized_coordinates=False):
        assert not normalized_coordinates
        ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
        return torch.stack([xs, ys], -1)[None]

    kornia.create_meshgrid = create_meshgrid

    sp = _mod("spconv")
    spp = _mod("spconv.pytorch")
    spc = _mod("spconv.pytorch.core")
    spconv_ = _mod("spconv.pytorch.conv")
    spm = _mod("spconv.pytorch.modules")
    sp.pytorch = spp
    spc.SparseConvTensor = _make_sparse_conv_tensor
    spconv_.SparseConv3d = _SparseConv3d
    spconv_.SubMConv3d = _SubMConv3d
    spm.SparseSequential = _SparseSequential

    sys.path.insert(0, REFERENCE_ROOT)
    # the real encoders module pulls transformers/kornia/clip at import; replace it wholesale
    enc = _mod("ldm.modules.encoders.modules")

    class FrozenCLIPImageEmbedder(nn.Module):
        def __init__(self, *a, **k):
            super().__init__()

        def encode(self, x):
            raise RuntimeError("CLIP is out of scope for the oracle harness")

    enc.FrozenCLIPImageEmbedder = FrozenCLIPImageEmbedder


def import_reference():
    """Returns a namespace with the reference modules on the hot path."""
    install_stubs()
    import importlib

    ns = types.SimpleNamespace()
    ns.attention = importlib.import_module("ldm.models.diffusion.attention")
    ns.network = importlib.import_module("ldm.models.diffusion.network")
    ns.utils = importlib.import_module("ldm.models.diffusion.utils")
    ns.openaimodel = importlib.import_module("ldm.modules.diffusionmodules.openaimodel")
    ns.mattention = importlib.import_module("ldm.modules.attention")
    ns.dutil = importlib.import_module("ldm.modules.diffusionmodules.util")
    return ns


def import_reference_full():
    ns = import_reference()
    import importlib

    ns.md = importlib.import_module("ldm.models.diffusion.morphable_diffusion")
    return ns
