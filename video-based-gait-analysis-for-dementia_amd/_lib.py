"""ctypes binding of libgrnet_hip.so (the C ABI declared in include/grnet_hip.h).

The product path has NO fallback: if the HIP library is missing or fails to load, importing the
model raises.  Nothing here touches ``oracle/``.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# GRNET_LIB_PATH: a diagnostic build of the same library (make ABLATION=1 BUILD=build_abl LIB=../libgrnet_hip_abl.so), tools/ only
LIB_PATH = os.environ.get("GRNET_LIB_PATH") or os.path.join(_HERE, "libgrnet_hip.so")

OK, EINVAL, ENOENT, ENOMEM, EHIP, ESTATE = 0, -22, -2, -12, -5, -1
DTYPE_F32, DTYPE_I64 = 0, 1
OPT_USE_GRAPH, OPT_CONV_TILE, OPT_MULTI_LANE, OPT_WINOGRAD, OPT_BF16_CHAIN, OPT_GRU_MODE, OPT_BF16_MIN_FRAMES = 1, 2, 3, 7, 8, 9, 10


class Outputs(C.Structure):
    """grnet_outputs_t"""
    _fields_ = [(n, C.c_void_p) for n in (
        "theta", "verts", "kp_2d", "kp_3d", "rotmat", "point_local_feat", "cam_shape_feats", "pred_rot6d",
        "features", "part_attn", "smpl_feats")]


class GaitOutputs(C.Structure):
    """grnet_gait_outputs_t"""
    _fields_ = [(n, C.c_void_p) for n in ("pred_avg", "pred_phase", "pred_cparam", "point_local_feat")]


EXPORTS = {
    "grnet_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int]),
    "grnet_load_tensor": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.POINTER(C.c_int64), C.c_int, C.c_int]),
    "grnet_load_smpl": (C.c_int, [C.c_void_p] + [C.c_void_p] * 7),
    "grnet_finalize_weights": (C.c_int, [C.c_void_p]),
    "grnet_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(Outputs), C.c_void_p]),
    "grnet_gru_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_void_p]),
    "grnet_tsattn_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "grnet_set_option": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "grnet_tune": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int]),
    "grnet_get_tuning": (C.c_int, [C.c_void_p, C.c_int, C.c_char_p, C.c_int]),
    "grnet_set_tuning": (C.c_int, [C.c_void_p, C.c_int, C.c_char_p]),
    "grnet_num_kernel_launches": (C.c_int, [C.c_void_p]),
    "grnet_num_conv_launches": (C.c_int, [C.c_void_p]),
    "grnet_conv_flops_per_frame": (C.c_double, [C.c_void_p]),
    "grnet_conv_executed_flops_per_frame": (C.c_double, [C.c_void_p]),
    "grnet_describe_conv": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int32), C.c_char_p, C.c_int]),
    "grnet_describe_conv_macs": (C.c_double, [C.c_void_p, C.c_int]),
    "grnet_conv_executed_flops_per_frame_n": (C.c_double, [C.c_void_p, C.c_int]),
    "grnet_conv_kernel_info": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_double)]),
    "grnet_time_conv": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_float)]),
    "grnet_op_timeline": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_char_p, C.c_int]),
    "grnet_time_convs": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_float)]),
    "grnet_op_conv2d": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                  C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "grnet_op_conv_chain": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                      C.POINTER(C.c_float), C.c_void_p]),
    "grnet_op_bilinear2x": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "grnet_smpl_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_void_p]),
    "grnet_crop_normalise": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_float, C.c_int,
                                       C.c_void_p, C.c_void_p]),
    "grnet_crop_normalise_cv": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                                          C.c_void_p]),
    "grnet_crop_normalise_cv_maps": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                                          C.c_void_p]),
    "grnet_head_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(Outputs), C.c_void_p]),
    "grnet_gait_correct": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                     C.POINTER(Outputs), C.POINTER(GaitOutputs), C.c_void_p]),
    "grnet_op_rot6d_to_rotmat": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "grnet_op_rotmat_to_aa": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "grnet_debug_tensor": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int, C.c_void_p, C.POINTER(C.c_int64), C.c_void_p]),
    "grnet_comm_probe": (C.c_int, []),
    "grnet_comm_unique_id": (C.c_int, [C.c_void_p, C.c_int]),
    "grnet_comm_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "grnet_comm_adopt": (C.c_int, [C.POINTER(C.c_void_p), C.c_void_p, C.c_int, C.c_int]),
    "grnet_allgather": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "grnet_comm_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "grnet_comm_destroy": (None, [C.c_void_p]),
    "grnet_comm_last_error": (C.c_char_p, []),
    "grnet_last_error": (C.c_char_p, [C.c_void_p]),
    "grnet_version": (C.c_char_p, []),
    "grnet_destroy": (None, [C.c_void_p]),
}

_lib = None


def load():
    """Load the shared library once; raise loudly if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    # torch bundles its own HIP runtime under the same SONAMEs as /opt/rocm's: import it FIRST so that
    # this library binds to the runtime torch allocates tensors with (one runtime per process).
    import torch  # noqa: F401
    if not os.path.isfile(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C video-based-gait-analysis-for-dementia_amd/csrc`). There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in EXPORTS.items():
        fn = getattr(lib, name)          # AttributeError here = header/library mismatch
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


class GrnetError(RuntimeError):
    pass


COMM_ID_BYTES = 128


def check_comm(lib, rc, what):
    if rc != 0:
        raise GrnetError(f"{what} failed with code {rc}: {lib.grnet_comm_last_error().decode()}")


def check(lib, handle, rc, what):
    if rc != 0:
        msg = lib.grnet_last_error(handle).decode() if handle else ""
        raise GrnetError(f"{what} failed with code {rc}: {msg}")
