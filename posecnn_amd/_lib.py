"""ctypes binding of libposecnn_hip.so (the C-ABI declared in include/posecnn_hip.h).

This is the Python-side analogue of the reference's ``tf.load_op_library('<op>.so')`` stubs
(lib/hough_voting_gpu_layer/hough_voting_gpu_op.py:4-7 and siblings). The library is built
in-tree by ``__graft_entry__.build()`` / ``make -C posecnn_amd/csrc``. There is NO fallback:
if the shared object is missing or an entry point fails, the call raises.
"""
import ctypes
import os
from ctypes import (POINTER, c_char_p, c_float, c_int, c_int32, c_int64, c_size_t, c_void_p)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libposecnn_hip.so")

PCNN_OK = 0
PCNN_EINVAL = -1
PCNN_EWORKSPACE = -2
PCNN_EHIP = -3
PCNN_ENULL = -4

MAX_ROI = 128
HOUGH_ROWS_CAPACITY = MAX_ROI * 9
VERTEX_CHANNELS = 3
POSE_CHANNELS = 4

_P = c_void_p

# name -> (restype, argtypes); must list every symbol of include/posecnn_hip.h
SIGNATURES = {
    "pcnn_abi_version": (c_int, []),
    "pcnn_last_error_string": (c_char_p, []),
    "pcnn_status_string": (c_char_p, [c_int]),
    "pcnn_hough_voting_workspace_bytes": (c_int, [c_int, c_int, c_int, c_int, c_float, c_int, c_int, POINTER(c_size_t)]),
    "pcnn_hough_voting_debug_layout": (c_int, [c_int, c_int, c_int, c_int, c_float, c_int, c_int, POINTER(c_size_t)]),
    "pcnn_hough_voting_fwd": (c_int, [_P, _P, _P, _P, _P,
                                      c_int, c_int, c_int, c_int, c_int, c_int,
                                      c_int, c_float, c_float, c_int, c_float, c_int, c_int, c_int,
                                      _P, _P, _P, _P, _P, _P,
                                      _P, c_size_t, _P]),
    "pcnn_hough_voting_lowres_fwd": (c_int, [_P, _P, _P, c_int, c_int, _P, _P, _P,
                                             c_int, c_int, c_int, c_int, c_int, c_int,
                                             c_int, c_float, c_float, c_int, c_float, c_int, c_int, c_int,
                                             _P, _P, _P, _P, _P, _P,
                                             _P, c_size_t, _P]),
    "pcnn_deconv_bilinear_bwd": (c_int, [_P, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P]),
    "pcnn_smooth_l1_vertex_workspace_bytes": (c_int, [_P]),
    "pcnn_smooth_l1_vertex_fwd": (c_int, [_P, _P, _P, c_int64, c_float, _P, _P, c_size_t, _P]),
    "pcnn_smooth_l1_vertex_bwd": (c_int, [_P, _P, _P, _P, _P, c_int64, c_float, _P, _P]),
    "pcnn_winograd_input_fwd": (c_int, [_P, c_int, c_int, c_int, c_int, _P, _P]),
    "pcnn_winograd_output_fwd": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P]),
    "pcnn_winograd43_input_fwd": (c_int, [_P, c_int, c_int, c_int, c_int, _P, _P]),
    "pcnn_winograd43_output_fwd": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P]),
    "pcnn_winograd43_output_both_fwd": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, _P, _P, _P]),
    "pcnn_fc_rows_fwd": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, _P, _P, _P, _P, c_size_t, _P]),
    "pcnn_fc_rows_cols_fwd": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, _P]),
    "pcnn_fc_rows_split_fwd": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, _P]),
    "pcnn_fc_rows_workspace_bytes": (c_int, [c_int, c_int, c_int, POINTER(c_size_t)]),
    "pcnn_fc_skinny_workspace_bytes": (c_int, [c_int, c_int, c_int, POINTER(c_size_t), POINTER(c_int)]),
    "pcnn_fc_skinny_fwd": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, _P, _P, _P, _P, c_size_t, _P, c_int, _P]),
    "pcnn_head_lowres_fwd": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P, _P]),
    "pcnn_head_lowres_mfma_fwd": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P, _P]),
    "pcnn_det_assemble_packed_fwd": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_float, _P, _P, _P]),
    "pcnn_pose_l2_normalize_fwd": (c_int, [_P, _P, _P, c_int, c_int, _P, _P]),
    "pcnn_det_assemble_fwd": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, _P, _P, _P]),
    "pcnn_winograd43_conv_fwd": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, c_size_t, _P]),
    "pcnn_winograd43_conv_workspace_bytes": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, POINTER(c_size_t)]),
    "pcnn_conv3x3_c3_winograd43_fwd": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P]),
    "pcnn_conv3x3_c3_winograd43_raw_fwd": (c_int, [_P, c_int, _P, c_int, _P, _P, _P, c_int, c_int, c_int, c_int, _P, _P]),
    "pcnn_conv1_1_conv1_2_fused_fwd": (c_int, [_P, _P, _P, _P, c_int, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P]),
    "pcnn_conv1_1_conv1_2_fused_raw_fwd": (c_int, [_P, c_int, _P, c_int, _P, _P, _P, _P, c_int, _P, c_int, c_int, c_int, c_int, _P, _P]),
    "pcnn_conv3x3_c3_fwd": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P, _P]),
    "pcnn_bias_relu_pool2_fwd": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, _P, _P]),
    "pcnn_hough_voting_bwd": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P]),
    "pcnn_roi_pool_fwd": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, c_int,
                                  c_int, c_int, c_float, c_int, _P, _P, _P]),
    "pcnn_roi_pool_bwd": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int,
                                  c_int, c_int, c_float, c_int, _P, _P]),
    "pcnn_roi_pool_add2_fwd": (c_int, [_P, c_int, c_int, c_float, _P, c_int, c_int, c_float,
                                       _P, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P, _P]),
    "pcnn_roi_pool_add2_live_fwd": (c_int, [_P, c_int, c_int, c_float, _P, c_int, c_int, c_float,
                                            _P, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P, _P]),
    "pcnn_hard_label_fwd": (c_int, [_P, _P, c_int64, c_int, c_float, _P, _P]),
    "pcnn_hard_label_bwd": (c_int, [_P, _P, c_int64, c_int, _P]),
    "pcnn_average_distance_workspace_bytes": (c_int, [c_int, c_int, c_int, POINTER(c_size_t)]),
    "pcnn_average_distance_fwd": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_float, _P,
                                          _P, _P, _P, c_size_t, _P]),
    "pcnn_average_distance_bwd": (c_int, [_P, _P, c_int, c_int, _P, _P]),
    "pcnn_backproject_fwd": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int,
                                     c_int, c_int, c_int, c_float, _P, _P, _P, _P]),
    "pcnn_backproject_workspace_bytes": (c_int, [c_int, c_int, c_int, c_int, POINTER(c_size_t)]),
    "pcnn_backproject_ws_fwd": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int,
                                        c_int, c_int, c_int, c_float, _P, _P, _P, _P, c_size_t, _P]),
    "pcnn_backproject_bwd": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P]),
    "pcnn_softmax_argmax_fwd": (c_int, [_P, c_int64, c_int, _P, _P, _P]),
    "pcnn_deconv_bilinear_fwd": (c_int, [_P, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, c_int, _P, _P]),
    "pcnn_bias_act_fwd": (c_int, [_P, _P, c_int64, c_int, c_int, _P, _P]),
    "pcnn_upscore_softmax_argmax_fwd": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, _P]),
    "pcnn_upscore_softmax_argmax_hard_fwd": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, _P, c_float, _P, _P]),
    "pcnn_icp_backproject_fwd": (c_int, [_P, _P, c_int, c_int, c_int, c_float, c_float, c_float, c_float, c_float, _P, _P]),
    "pcnn_icp_refine_workspace_bytes": (c_int, [c_int, c_int, c_int, POINTER(c_size_t)]),
    "pcnn_icp_refine_fwd": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_float, c_float, c_float, c_float, c_float, c_float, c_float,
                                    c_int, _P, _P, _P, c_size_t, _P]),
    "pcnn_render_mesh_workspace_bytes": (c_int, [c_int, c_int, c_int, POINTER(c_size_t)]),
    "pcnn_render_mesh_fwd": (c_int, [_P, _P, _P, c_int, c_int, _P, c_int, c_int, c_int, c_float, c_float, c_float, c_float, c_float, c_float,
                                     c_float, _P, _P, _P, _P, c_size_t, _P]),
    "pcnn_icp_center_workspace_bytes": (c_int, [c_int, c_int, POINTER(c_size_t)]),
    "pcnn_icp_center_fwd": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_float, _P, _P, _P, c_size_t, _P]),
    "pcnn_icp_score_workspace_bytes": (c_int, [c_int, c_int, c_int, POINTER(c_size_t)]),
    "pcnn_icp_score_fwd": (c_int, [_P, _P, _P, c_int, c_int, _P, c_int, c_float, c_float, c_float, c_float, c_float, _P, _P, c_size_t, _P]),
    "pcnn_icp_polish_fwd": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_float, c_float, c_int, _P, _P, _P]),
    "pcnn_crc32c": (ctypes.c_uint32, [_P, c_size_t, ctypes.c_uint32]),
    "pcnn_profile_enable": (c_int, [c_int]),
    "pcnn_profile_reset": (c_int, []),
    "pcnn_profile_report": (ctypes.c_long, [ctypes.c_char_p, ctypes.c_long]),
}


def profile_enable(on=True):
    """Bracket every library kernel launch with HIP events on its launch stream."""
    check("pcnn_profile_enable", lib().pcnn_profile_enable(1 if on else 0))


def profile_report(reset=True):
    """-> {kernel_name: {"calls", "total_ms", "avg_us"}} for launches since the last reset."""
    import json
    L = lib()
    n = L.pcnn_profile_report(None, 0)
    buf = ctypes.create_string_buffer(int(n) + 16)
    L.pcnn_profile_report(buf, len(buf))
    if reset:
        L.pcnn_profile_reset()
    rep = json.loads(buf.value.decode())
    # template instances are launched as `(kernel<a, b>)`: drop the macro's parentheses
    return {k.strip("()"): v for k, v in rep.items()}

_lib = None


class PoseCNNHipError(RuntimeError):
    """A libposecnn_hip.so entry point returned a negative pcnn_status."""

    def __init__(self, fn, status, message):
        super().__init__("%s failed: status %d (%s)" % (fn, status, message))
        self.status = status


def lib():
    """Load (once) and return the ctypes handle; raises if the HIP library is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "libposecnn_hip.so not found at %s — build it with "
                "`python -c 'import __graft_entry__ as g; g.build()'` or "
                "`make -C posecnn_amd/csrc`. There is no CPU/PyTorch fallback." % LIB_PATH)
        # The HIP runtime the process uses must be ONE: torch ships its own libamdhip64 under torch/lib, this library names the
        # system's by soname. Loaded after torch, the soname resolves to torch's already-mapped copy; loaded BEFORE it, the
        # process ends up with two runtimes and every launch on a torch stream fails with "no ROCm-capable device is detected"
        # (round 6: `build()` followed by `smoke()` in one process did exactly that). So torch's goes in first, always.
        import torch  # noqa: F401
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if the symbol is missing: fail loudly
            fn.restype = res
            fn.argtypes = args
        if handle.pcnn_abi_version() != 2:
            raise RuntimeError("libposecnn_hip.so ABI version mismatch")
        _lib = handle
    return _lib


def check(fn_name, status):
    if status != PCNN_OK:
        msg = lib().pcnn_last_error_string()
        msg = msg.decode("utf-8", "replace") if msg else ""
        if status in (PCNN_EINVAL, PCNN_ENULL):
            # the reference raises InvalidArgument for the same conditions (OP_REQUIRES)
            raise ValueError("%s: %s" % (fn_name, msg))
        raise PoseCNNHipError(fn_name, status, msg)
