"""ctypes binding of libmvd_hip.so (C ABI in include/mvd.h).

The HIP library is the product path: there is NO CPU / PyTorch fallback.  If the shared object is missing
(or fails to load) every entry point raises, loudly.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# MVD_DTYPE (read once, at the first load): "f16" (default) -> libmvd_hip.so, "bf16" -> libmvd_hip_bf16.so, the same sources
# built with bfloat16 MFMA operands and storage (csrc/common.h MVD_BF16; `make -C csrc bf16`): the training dtype of BASELINE
# configs[3].  One process uses one of them.  MVD_LIB_PATH: A/B timing of two builds of the same ABI (development aid).
DTYPE = os.environ.get("MVD_DTYPE", "f16")
if DTYPE not in ("f16", "bf16"):
    raise ValueError(f"MVD_DTYPE must be f16 or bf16, not {DTYPE!r}")
LIB_PATH = os.environ.get("MVD_LIB_PATH", os.path.join(_HERE, "libmvd_hip.so" if DTYPE == "f16" else "libmvd_hip_bf16.so"))

SYMBOLS = [
    "mvd_create", "mvd_destroy", "mvd_last_error", "mvd_compute_dtype", "mvd_upload_weight", "mvd_set_precision_level", "mvd_set_vae_precision", "mvd_finalize_weights", "mvd_unet_forward", "mvd_unet_block",
    "mvd_embed_time", "mvd_select_sample", "mvd_set_mesh", "mvd_set_cameras", "mvd_set_mesh_async", "mvd_set_cameras_async", "mvd_set_samples_async", "mvd_rulebook_build", "mvd_rulebook_table", "mvd_vertex_features", "mvd_vertex_view_features", "mvd_vertex_features_stream_safe", "mvd_fuse_vertex_features",
    "mvd_comm_unique_id", "mvd_comm_init", "mvd_comm_destroy", "mvd_exchange_view_features", "mvd_comm_all_reduce", "mvd_train_sync_gradients",
    "mvd_stage_target_encoder", "mvd_stage_sparse_dense", "mvd_set_volume_ready_event", "mvd_volume_from_fused", "mvd_volume_from_fused_train", "mvd_mse_loss", "mvd_set_volume", "mvd_train_enable", "mvd_train_param_count", "mvd_train_param_info", "mvd_train_arena_size", "mvd_train_adopt_arena", "mvd_train_zero_grad", "mvd_train_unet_step", "mvd_train_get_grad", "mvd_train_get_tensor", "mvd_train_bn_calls", "mvd_train_cond_backward", "mvd_train_conditioner_backward", "mvd_train_conditioner_backward_batch", "mvd_train_adamw_step", "mvd_train_repack", "mvd_train_repack_async", "mvd_train_grad_bucket_count", "mvd_train_grad_bucket", "mvd_train_grad_bucket_wait", "mvd_train_set_bucket_snapshot",
    "mvd_frustum_volumes", "mvd_frustum_volumes_batch", "mvd_denoise_views", "mvd_denoise_views_batch", "mvd_op_conv", "mvd_op_linear", "mvd_op_group_norm",
    "mvd_op_layer_norm", "mvd_op_attention", "mvd_op_attention_bwd", "mvd_op_group_norm_bwd", "mvd_op_layer_norm_bwd", "mvd_op_conv3d", "mvd_op_st_tail", "mvd_op_st_head", "mvd_bench_conv", "mvd_bench_linear", "mvd_bench_group_norm", "mvd_probe_config", "mvd_probe_report", "mvd_vae_decode", "mvd_vae_encode",
    "mvd_clip_encode", "mvd_clip_embed_dim",
]


def csrc_sha16():
    """First 16 hex digits of the SHA-256 over the library's sources (csrc/*.hip, *.h, Makefile, include/mvd.h, sorted by name):
    the build a measurement belongs to.  tools/pmc_traffic.py stamps it into profiles/pmc_traffic.json and bench.py reports
    roofline.traffic only when the stamp matches the tree it runs from."""
    import glob
    import hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(_HERE, "csrc", "*.hip")) + glob.glob(os.path.join(_HERE, "csrc", "*.h")) +
                   [os.path.join(_HERE, "csrc", "Makefile"), os.path.join(_HERE, "..", "include", "mvd.h")])
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


class UNetConfigC(C.Structure):
    _fields_ = [("image_size", C.c_int), ("in_channels", C.c_int), ("out_channels", C.c_int),
                ("model_channels", C.c_int), ("num_res_blocks", C.c_int), ("channel_mult", C.c_int * 4),
                ("num_heads", C.c_int), ("context_dim", C.c_int), ("volume_dims", C.c_int * 4),
                ("attention_levels", C.c_int)]


class VolumeConfigC(C.Structure):
    _fields_ = [("time_dim", C.c_int), ("view_dim", C.c_int), ("num_views", C.c_int), ("input_image_size", C.c_int),
                ("frustum_volume_depth", C.c_int), ("spatial_volume_size", C.c_int),
                ("spatial_volume_length", C.c_float), ("frustum_volume_length", C.c_float), ("projection", C.c_int),
                ("frustum_dims", C.c_int * 4), ("voxel_size", C.c_float)]


class MvdError(RuntimeError):
    pass


_lib = None


def load():
    """Loads libmvd_hip.so; raises MvdError if the HIP extension has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MvdError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                       f"(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    lib.mvd_last_error.restype = C.c_char_p
    lib.mvd_compute_dtype.restype = C.c_char_p
    lib.mvd_destroy.restype = None
    for name in SYMBOLS:
        if not hasattr(lib, name):
            raise MvdError(f"libmvd_hip.so does not export {name}")
    for name in SYMBOLS:
        if name not in ("mvd_last_error", "mvd_destroy", "mvd_compute_dtype"):
            getattr(lib, name).restype = C.c_int
    lib.mvd_train_arena_size.restype = C.c_int64
    lib.mvd_train_bn_calls.restype = C.c_int64
    if "MVD_LIB_PATH" not in os.environ and lib.mvd_compute_dtype().decode() != DTYPE:
        raise MvdError(f"{LIB_PATH} computes in {lib.mvd_compute_dtype().decode()}, MVD_DTYPE asks for {DTYPE}")
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        raise MvdError(load().mvd_last_error().decode())


def ptr(t):
    """Device/host pointer of a torch tensor (None -> NULL)."""
    if t is None:
        return C.c_void_p(0)
    assert t.is_contiguous(), "tensor must be contiguous at the C boundary"
    return C.c_void_p(t.data_ptr())
