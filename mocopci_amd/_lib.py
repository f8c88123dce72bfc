"""ctypes binding of libmocopci_hip.so (C ABI declared in include/mocopci_hip.h).

There is no CPU fallback: if the shared library is missing, or a tensor is not a
contiguous CUDA(HIP) tensor of the right dtype, the call raises.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# MCP_HIP_LIB overrides the library path (A/B runs of two builds on one device); default: the in-tree build
SO_PATH = os.environ.get("MCP_HIP_LIB") or os.path.join(_HERE, "libmocopci_hip.so")

_i, _f, _p = ctypes.c_int, ctypes.c_float, ctypes.c_void_p

# name -> argtypes (restype is int unless listed in _RESTYPES); mirrors include/mocopci_hip.h
SIGNATURES = {
    "mcp_abi_version": [],
    "mcp_error_string": [_i],
    "mcp_furthest_point_sampling": [_i, _i, _i, _p, _p, _p, _p],
    "mcp_fps_workspace_bytes": [_i, _i, _i],
    "mcp_furthest_point_sampling_ws": [_i, _i, _i, _p, _p, _p, _p, ctypes.c_size_t, _p],
    "mcp_furthest_point_sampling_fresh": [_i, _i, _i, _p, _p, _p, _p, ctypes.c_size_t, _p],
    "mcp_gather_points": [_i, _i, _i, _i, _p, _p, _p, _p],
    "mcp_gather_points_grad": [_i, _i, _i, _i, _p, _p, _p, _p],
    "mcp_group_points": [_i, _i, _i, _i, _i, _p, _p, _p, _p],
    "mcp_group_points_grad": [_i, _i, _i, _i, _i, _p, _p, _p, _p],
    "mcp_group_points_grad_sorted": [_i, _i, _i, _i, _p, _p, _p, _p, _p],
    "mcp_three_interpolate_grad_sorted": [_i, _i, _i, _i, _p, _p, _p, _p, _p, _p],
    "mcp_ball_query": [_i, _i, _i, _f, _i, _p, _p, _p, _p],
    "mcp_query_and_group": [_i, _i, _i, _i, _f, _i, _i, _p, _p, _p, _p, _p],
    "mcp_three_nn": [_i, _i, _i, _p, _p, _p, _p, _p],
    "mcp_three_interpolate": [_i, _i, _i, _i, _p, _p, _p, _p, _p],
    "mcp_three_interpolate_grad": [_i, _i, _i, _i, _p, _p, _p, _p, _p],
    "mcp_knn": [_i, _i, _i, _i, _i, _p, _p, _p, _p, _p],
    "mcp_knn_tile_size": [],
    "mcp_build_cloud": [_i, _i, _p, _p, _p, _p, _p],
    "mcp_morton_codes": [_i, _i, _p, _p, _p, _p],
    "mcp_tile_boxes": [_i, _i, _p, _p, _p],
    "mcp_knn_pruned": [_i] * 5 + [_p] * 8,
    "mcp_knn_cosine": [_i] * 5 + [_p] * 6,
    "mcp_group_rows": [_i, _i, _i, _i, _p, _p, _p, _p],
    "mcp_interp3": [_i, _i, _i, _i, _p, _p, _p, _p, _p, _p, _p],
    "mcp_interp3_weights": [_i, _i, _i, _p, _p, _p, _p, _p],
    "mcp_interp3_apply": [_i, _i, _i, _i, _p, _p, _p, _p, _p],
    "mcp_interp3_apply_grad": [_i, _i, _i, _i, _p, _p, _p, _p, _p],
    "mcp_interp3_apply_grad_sorted": [_i, _i, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p],
    "mcp_group_rows_grad": [_i, _i, _i, _i, _p, _p, _p, _p],
    "mcp_group_rows_grad_sorted": [_i, _i, _i, _i, _p, _p, _p, _p, _p],
    "mcp_group_rows_add_leaky": [_i, _i, _i, _i, _i, _f, _p, _p, _p, _p, _p],
    "mcp_fusion": [_i, _i, _i] + [_p] * 12,
    "mcp_fusion_grad_floats": [],
    "mcp_fusion_grad_workspace_bytes": [_i, _i],
    "mcp_fusion_grad": [_i, _i, _i] + [_p] * 15 + [ctypes.c_size_t, _p],
    "mcp_fusion_bn_floats": [],
    "mcp_fusion_bn_workspace_bytes": [_i, _i],
    "mcp_fusion_bn_forward": [_i, _i, _i] + [_p] * 10 + [_f] + [_p] * 4 + [ctypes.c_size_t, _p],
    "mcp_fusion_bn_grad_workspace_bytes": [_i, _i],
    "mcp_fusion_bn_backward": [_i, _i, _i] + [_p] * 22 + [ctypes.c_size_t, _p],
    "mcp_fusion_bn_forward_save": [_i, _i, _i] + [_p] * 10 + [_f] + [_p] * 7 + [ctypes.c_size_t, _p],
    "mcp_fusion_bn_backward_saved": [_i, _i, _i] + [_p] * 25 + [ctypes.c_size_t, _p],
    "mcp_cross_packed_floats": [_i],
    "mcp_cross_pack": [_i, _p, _p, _p, _p, _p, _p],
    "mcp_cross_volume": [_i] * 5 + [_p] * 7 + [_i, _p, _p, _p],
    "mcp_cross_grad_floats": [_i],
    "mcp_cross_grad_workspace_bytes": [_i, _i, _i],
    "mcp_cross_grad": [_i] * 5 + [_p] * 17 + [ctypes.c_size_t, _p],
    "mcp_pointconv_agg": [_i] * 5 + [_p] * 12,
    "mcp_pointconv_agg_grad_floats": [],
    "mcp_pointconv_agg_grad_workspace_bytes": [_i, _i],
    "mcp_pointconv_agg_grad": [_i] * 5 + [_p] * 16 + [ctypes.c_size_t, _p],
    "mcp_pointconv_linear": [_i] * 5 + [_p] * 11 + [_i, _f, _p, _p],
    "mcp_attention_small": [_i] * 5 + [_p, _i, _p, _i, _p, _i, _f, _p, _i, _p],
    "mcp_attention_wide": [_i] * 5 + [_p, _i, _p, _i, _p, _i, _f, _p, _i, _p],
    "mcp_attention_small_grad_workspace_bytes": [_i, _i, _i],
    "mcp_attention_small_dropout": [_i] * 5 + [_p, _i, _p, _i, _p, _i, _f, _f, ctypes.c_uint, _p, _p],
    "mcp_attention_small_grad": [_i] * 5 + [_p, _i, _p, _i, _p, _i, _f, _f, ctypes.c_uint, _p, _p, _p, _p, _p, ctypes.c_size_t, _p],
    "mcp_attention_small_lse": [_i] * 5 + [_p, _i, _p, _i, _p, _i, _f, _f, ctypes.c_uint, _p, _p, _p],
    "mcp_attention_small_grad_lse": [_i] * 5 + [_p, _i, _p, _i, _p, _i, _f, _f, ctypes.c_uint, _p, _p, _p, _p, _p, _p, ctypes.c_size_t, _p],
    "mcp_chamfer_nn": [_i, _i, _i, _p, _p, _p, _p, _p],
    "mcp_ptblock_packed_floats": [],
    "mcp_ptblock_pack": [_p] * 10,
    "mcp_ptblock_attention": [_i] * 5 + [_p] * 8,
    "mcp_ptblock_grad_floats": [],
    "mcp_ptblock_grad_workspace_bytes": [_i, _i],
    "mcp_ptblock_grad": [_i] * 5 + [_p] * 21 + [ctypes.c_size_t, _p],
    "mcp_linear_packed_floats": [_i, _i, _p],
    "mcp_linear_pack": [_i, _i, _p, _p, _p, _p, _p],
    "mcp_attention": [_i, _i, _i, _i, _i, _p, _i, _p, _i, _p, _i, _i, _f, _p, _i, _p],
    "mcp_add_layernorm": [ctypes.c_longlong, _i, _p, ctypes.c_longlong, _p, ctypes.c_longlong, _p, _p, _p, _f, _p, ctypes.c_longlong, _p],
    "mcp_mfa_prepare": [_i, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p],
    "mcp_linear": [ctypes.c_longlong, _i, _i, _p, _p, _p, _f, _p, _p, _i, _p, _i, _p],
    "mcp_linear_as": [ctypes.c_longlong, ctypes.c_longlong, _i, _i, _p, _p, _p, _f, _p, _p, _i, _p, _i, _p],
    "mcp_scatter_segments_workspace_bytes": [_i, _i, _i],
    "mcp_scatter_segments": [_i, _i, _i, _p, _p, _p, _p, ctypes.c_size_t, _p],
    "mcp_prelu_dropout": [ctypes.c_longlong, _p, _p, _f, ctypes.c_uint, _p, _p],
    "mcp_prelu_dropout_grad_workspace_bytes": [ctypes.c_longlong],
    "mcp_prelu_dropout_grad": [ctypes.c_longlong, _p, _p, _p, _f, ctypes.c_uint, _p, _p, _p, ctypes.c_size_t, _p],
    "mcp_linear_wgrad_workspace_bytes": [ctypes.c_longlong, _i, _i],
    "mcp_linear_wgrad": [ctypes.c_longlong, _i, _i, _p, _i, _p, _i, _p, _p, _p, ctypes.c_size_t, _p],
    "mcp_linear_narrow": [ctypes.c_longlong, _i, _i, _p, _i, _p, _p, _f, _p, _i, _p],
    "mcp_mlp2_packed_floats": [_i, _i, _i],
    "mcp_mlp2_pack": [_i, _i, _i, _p, _p, _p, _p, _p, _p],
    "mcp_mlp2": [ctypes.c_longlong, _i, _i, _i, _f, _p, _i, _p, _i, _p, _p, _i, _p],
    "mcp_emd": [_i, _i, _i, _p, _p, _p, _p, _p, _p],
    "mcp_prof_enable": [_i],
    "mcp_prof_collect": [_i, _p, _p],
}
_RESTYPES = {"mcp_error_string": ctypes.c_char_p, "mcp_fps_workspace_bytes": ctypes.c_size_t, "mcp_fusion_grad_workspace_bytes": ctypes.c_size_t,
             "mcp_cross_grad_workspace_bytes": ctypes.c_size_t, "mcp_pointconv_agg_grad_workspace_bytes": ctypes.c_size_t,
             "mcp_fusion_bn_workspace_bytes": ctypes.c_size_t, "mcp_fusion_bn_grad_workspace_bytes": ctypes.c_size_t,
             "mcp_ptblock_grad_workspace_bytes": ctypes.c_size_t, "mcp_attention_small_grad_workspace_bytes": ctypes.c_size_t,
             "mcp_linear_wgrad_workspace_bytes": ctypes.c_size_t, "mcp_scatter_segments_workspace_bytes": ctypes.c_size_t,
             "mcp_prelu_dropout_grad_workspace_bytes": ctypes.c_size_t}

_lib = None


def load():
    """Load the HIP library; raises with a build hint if it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise RuntimeError(
                f"{SO_PATH} not found: build it with `make -C mocopci_amd/csrc` "
                "(or python -c 'import __graft_entry__ as g; g.build()'). There is no CPU fallback.")
        lib = ctypes.CDLL(SO_PATH)
        for name, argtypes in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the symbol is not exported
            fn.argtypes = argtypes
            fn.restype = _RESTYPES.get(name, _i)
        _lib = lib
    return _lib


class Unsupported(RuntimeError):
    """MCP_ERR_UNSUPPORTED: the compiled kernels do not cover this size (a RuntimeError, so existing handlers still see it)."""


def check(rc):
    if rc != 0:
        msg = load().mcp_error_string(rc)
        raise (Unsupported if rc == 10002 else RuntimeError)(f"libmocopci_hip: error {rc}: {msg.decode() if msg else '?'}")


def fptr(t):
    if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
        raise RuntimeError(f"expected a contiguous float32 CUDA tensor, got {t.dtype} {t.device} contiguous={t.is_contiguous()}")
    return t.data_ptr()


def iptr(t):
    if not (t.is_cuda and t.dtype == torch.int32 and t.is_contiguous()):
        raise RuntimeError(f"expected a contiguous int32 CUDA tensor, got {t.dtype} {t.device} contiguous={t.is_contiguous()}")
    return t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream
