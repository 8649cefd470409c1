"""ctypes binding of libvtaco_hip.so (the C ABI declared in include/vtaco_hip.h).

There is no CPU fallback: if the library is missing, or a tensor is not on a HIP
device, every op raises.  ``import torch`` must come first so that the library
resolves libamdhip64.so.7 to the copy torch has already loaded (one HIP runtime
per process: device pointers are only meaningful inside the runtime that made
them).
"""
from __future__ import annotations

import ctypes
import os

import torch  # noqa: F401  (must precede the CDLL below)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VTACO_HIP_LIB") or os.path.join(_HERE, "libvtaco_hip.so")   # env: perf-variant builds

VT_MAX_BLOCKS = 8
c_float_p = ctypes.POINTER(ctypes.c_float)


class VtError(RuntimeError):
    pass


class DecoderParams(ctypes.Structure):
    """Mirror of ``vt_decoder_params`` (include/vtaco_hip.h)."""
    _fields_ = [
        ("hidden", ctypes.c_int32), ("c_dim", ctypes.c_int32),
        ("n_blocks", ctypes.c_int32), ("p_in", ctypes.c_int32),
        ("fc_p_w", ctypes.c_void_p), ("fc_p_b", ctypes.c_void_p),
        ("fc_c_w", ctypes.c_void_p * VT_MAX_BLOCKS), ("fc_c_b", ctypes.c_void_p * VT_MAX_BLOCKS),
        ("fc0_w", ctypes.c_void_p * VT_MAX_BLOCKS), ("fc0_b", ctypes.c_void_p * VT_MAX_BLOCKS),
        ("fc1_w", ctypes.c_void_p * VT_MAX_BLOCKS), ("fc1_b", ctypes.c_void_p * VT_MAX_BLOCKS),
        ("fc_out_w", ctypes.c_void_p), ("fc_out_b", ctypes.c_void_p),
        ("fc_out2_w", ctypes.c_void_p), ("fc_out2_b", ctypes.c_void_p),
    ]


class FusionUnit(ctypes.Structure):
    """Mirror of ``vt_fusion_unit``."""
    _fields_ = [(n, ctypes.c_void_p) for n in ("WK", "WQ", "WV", "trans_conv", "linear1_w", "linear1_b",
                                                "linear2_w", "linear2_b", "norm2_w", "norm2_b")]


class FusionParams(ctypes.Structure):
    """Mirror of ``vt_fusion_params``."""
    _fields_ = [("d_model", ctypes.c_int32), ("key_dim", ctypes.c_int32),
                ("self_attn", FusionUnit), ("cross_attn", FusionUnit)]


class FusionGrads(ctypes.Structure):
    """Mirror of ``vt_fusion_grads`` (two ``vt_fusion_unit_grads``: the same ten names, gradient buffers)."""
    _fields_ = [("self_attn", FusionUnit), ("cross_attn", FusionUnit)]


VT_UNET_MAX_LEVELS = 6


class UnetConv(ctypes.Structure):
    """Mirror of ``vt_unet3d_conv``."""
    _fields_ = [("gn_w", ctypes.c_void_p), ("gn_b", ctypes.c_void_p), ("packed", ctypes.c_void_p),
                ("cin", ctypes.c_int32), ("cout", ctypes.c_int32), ("packed_bf16x3", ctypes.c_void_p),
                ("packed_f16x3", ctypes.c_void_p), ("packed_f16x3_thin", ctypes.c_void_p),
                ("packed_f16x3_up", ctypes.c_void_p)]


class UnetParams(ctypes.Structure):
    """Mirror of ``vt_unet3d_params``."""
    _fields_ = [("n_levels", ctypes.c_int32), ("groups", ctypes.c_int32), ("eps", ctypes.c_double),
                ("enc", (UnetConv * 2) * VT_UNET_MAX_LEVELS), ("dec", (UnetConv * 2) * VT_UNET_MAX_LEVELS),
                ("final_w", ctypes.c_void_p), ("final_b", ctypes.c_void_p), ("out_channels", ctypes.c_int32),
                ("final_packed_f16x3", ctypes.c_void_p)]


VT_PLANE_UNET_MAX_DEPTH = 5


class PlaneUnetParams(ctypes.Structure):
    """Mirror of ``vt_plane_unet_params``."""
    _fields_ = [("depth", ctypes.c_int32), ("in_channels", ctypes.c_int32), ("start_filts", ctypes.c_int32), ("num_classes", ctypes.c_int32),
                ("down_w", (ctypes.c_void_p * 2) * VT_PLANE_UNET_MAX_DEPTH), ("down_b", (ctypes.c_void_p * 2) * VT_PLANE_UNET_MAX_DEPTH),
                ("up_tw", ctypes.c_void_p * VT_PLANE_UNET_MAX_DEPTH), ("up_tb", ctypes.c_void_p * VT_PLANE_UNET_MAX_DEPTH),
                ("up_w", (ctypes.c_void_p * 2) * VT_PLANE_UNET_MAX_DEPTH), ("up_b", (ctypes.c_void_p * 2) * VT_PLANE_UNET_MAX_DEPTH),
                ("final_w", ctypes.c_void_p), ("final_b", ctypes.c_void_p)]


class PlaneUnetGrads(ctypes.Structure):
    """Mirror of ``vt_plane_unet_grads``."""
    _fields_ = [("down_w", (ctypes.c_void_p * 2) * VT_PLANE_UNET_MAX_DEPTH), ("down_b", (ctypes.c_void_p * 2) * VT_PLANE_UNET_MAX_DEPTH),
                ("up_tw", ctypes.c_void_p * VT_PLANE_UNET_MAX_DEPTH), ("up_tb", ctypes.c_void_p * VT_PLANE_UNET_MAX_DEPTH),
                ("up_w", (ctypes.c_void_p * 2) * VT_PLANE_UNET_MAX_DEPTH), ("up_b", (ctypes.c_void_p * 2) * VT_PLANE_UNET_MAX_DEPTH),
                ("final_w", ctypes.c_void_p), ("final_b", ctypes.c_void_p)]


# name -> (restype, argtypes); kept in step with include/vtaco_hip.h (tests/test_abi.py
# parses the header and checks that every declared symbol is exported and listed here)
_VP, _I, _I64, _F, _D, _SZ = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_double, ctypes.c_size_t
SIGNATURES = {
    "vt_abi_version": (_I, []),
    "vt_last_error": (ctypes.c_char_p, []),
    "vt_decoder_blob_bytes": (_SZ, [_I, _I, _I]),
    "vt_decoder_pack": (_I, [ctypes.POINTER(DecoderParams), _VP, _SZ, _VP]),
    "vt_grid_to_channels_last": (_I, [_VP, _VP, _I, _I, _I, _I, _I, _VP]),
    "vt_grid_from_channels_last": (_I, [_VP, _VP, _I, _I, _I, _I, _I, _VP]),
    "vt_decode_fwd": (_I, [_VP, _I, _I, _I, _VP, _I64, _I, _F, _I64, _VP, _VP, _D, _VP, _VP, _VP, _VP]),
    "vt_tactile_assign": (_I, [_VP, _I, _I64, _I, _F, _I64, _VP, _VP, _VP, _I, _I, _I, _D, _VP, _VP]),
    "vt_decode_fwd_ids": (_I, [_VP, _I, _I, _I, _VP, _I64, _I, _F, _I64, _VP, _VP, _I, _VP, _D, _VP, _VP]),
    "vt_sample_grid": (_I, [_VP, _I, _I, _I, _VP, _I64, _I, _F, _I64, _D, _VP, _VP]),
    "vt_decode_mlp_fwd": (_I, [_VP, _I, _I, _VP, _I64, _I, _F, _I64, _VP, _VP, _VP]),
"vt_decode_mlp_fwd_f16x3": (_I, [_VP, _I, _I, _VP, _I64, _I, _F, _I64, _VP, _VP, _VP]),
    "vt_fusion_workspace_bytes": (_SZ, [_I, _I]),
    "vt_fusion_workspace_bytes_wide": (_SZ, [_I, _I, _I]),
    "vt_fusion_fwd": (_I, [_VP, _VP, _I, _I, ctypes.POINTER(FusionParams), _VP, _SZ, _VP, _VP]),
    "vt_fusion_fwd_ids": (_I, [_VP, _VP, _I, _VP, _VP, _I, _I, ctypes.POINTER(FusionParams), _VP, _SZ, _VP, _VP]),
    "vt_fusion_saved_bytes": (_SZ, [_I, _I]),
    "vt_fusion_bwd_workspace_bytes": (_SZ, [_I, _I]),
    "vt_fusion_fwd_train": (_I, [_VP, _VP, _I, _I, ctypes.POINTER(FusionParams), _F, ctypes.c_ulonglong, _VP, _SZ, _VP, _SZ, _VP, _VP]),
    "vt_fusion_bwd": (_I, [_VP, _VP, _VP, _I, _I, ctypes.POINTER(FusionParams), _F, ctypes.c_ulonglong, _VP, _SZ, _VP, _SZ, _VP, _VP,
                           ctypes.POINTER(FusionGrads), _VP]),
    "vt_fusion_dropout_mask": (_I, [_F, ctypes.c_ulonglong, _I, _I, _I, _VP, _VP]),
    "vt_fusion_saved_bytes_wide": (_SZ, [_I, _I, _I]),
    "vt_fusion_bwd_workspace_bytes_wide": (_SZ, [_I, _I, _I]),
    "vt_fusion_dropout_mask_wide": (_I, [_F, ctypes.c_ulonglong, _I, _I, _I, _I, _VP, _VP]),
    "vt_decoder_blob_t_bytes": (_SZ, [_I, _I, _I]),
    "vt_decoder_pack_bf16x3": (_I, [ctypes.POINTER(DecoderParams), _VP, _SZ, _VP]),
    "vt_decode_fwd_bf16x3": (_I, [_VP, _I, _I, _I, _VP, _I64, _I, _F, _I64, _VP, _VP, _VP, _I, _VP, _D, _VP, _VP, _VP]),
    "vt_decoder_pack_f16x3": (_I, [ctypes.POINTER(DecoderParams), _VP, _SZ, _VP]),
    "vt_decode_fwd_f16x3": (_I, [_VP, _I, _I, _I, _VP, _I64, _I, _F, _I64, _VP, _VP, _VP, _I, _VP, _D, _VP, _VP, _VP]),
    "vt_decode_range_status": (_I, [ctypes.POINTER(ctypes.c_uint32), _I, _VP]),
    "vt_decode_last_clock": (_I, [ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_int),
                                  ctypes.POINTER(ctypes.c_uint64), _I, ctypes.POINTER(ctypes.c_int), _VP]),
    "vt_decoder_pack_f16f8": (_I, [ctypes.POINTER(DecoderParams), _VP, _SZ, _VP]),
    "vt_decoder_wide_blob_bytes": (_SZ, [_I, _I, _I, _I]),
    "vt_decoder_wide_blob_f16x3_bytes": (_SZ, [_I, _I, _I, _I]),
    "vt_decoder_pack_wide": (_I, [ctypes.POINTER(DecoderParams), _VP, _SZ, _VP]),
    "vt_decoder_pack_wide_f16x3": (_I, [ctypes.POINTER(DecoderParams), _VP, _SZ, _VP]),
    "vt_decode_fwd_wide": (_I, [_VP, _I, _I, _I, _VP, _I64, _I, _F, _I64, _VP, _VP, _I, _I, _I, _D, _VP, _VP, _VP]),
    "vt_decode_fwd_wide_f16x3": (_I, [_VP, _I, _I, _I, _VP, _I64, _I, _F, _I64, _VP, _VP, _I, _I, _I, _D, _VP, _VP, _VP]),
    "vt_decode_fwd_wide_ids": (_I, [_VP, _I, _I, _I, _VP, _I64, _I, _F, _I64, _VP, _VP, _I, _VP, _I, _I, _I, _D, _VP, _VP, _VP]),
    "vt_decode_wide_f16x3_workspace_bytes": (_SZ, [_I64, _I, _I, _I, _I]),
    "vt_decode_fwd_wide_f16x3_ws": (_I, [_VP, _I, _I, _I, _VP, _I64, _I, _F, _I64, _VP, _VP, _I, _I, _I, _D, _VP, _VP, _VP, _SZ, _VP]),
    "vt_decode_fwd_wide_f16x3_ids": (_I, [_VP, _I, _I, _I, _VP, _I64, _I, _F, _I64, _VP, _VP, _I, _VP, _I, _I, _I, _D, _VP, _VP, _VP]),
    "vt_decode_mlp_fwd_wide": (_I, [_VP, _I, _I, _VP, _I64, _I, _F, _I64, _VP, _I, _I, _I, _VP, _VP, _VP]),
    "vt_decode_mlp_fwd_wide_f16x3": (_I, [_VP, _I, _I, _VP, _I64, _I, _F, _I64, _VP, _I, _I, _I, _VP, _VP, _VP]),
    "vt_decode_wide_save_floats": (_SZ, [_I64, _I, _I, _I]),
    "vt_decode_wide_gws_floats": (_SZ, [_I64, _I, _I, _I]),
    "vt_decode_fwd_wide_train": (_I, [_VP, _I, _I, _I, _VP, _I64, _VP, _VP, _I, _I, _I, _D, _VP, _VP, _VP, _VP]),
    "vt_decoder_wide_blob_t_bytes": (_SZ, [_I, _I, _I]),
    "vt_decoder_pack_wide_t": (_I, [ctypes.POINTER(DecoderParams), _VP, _SZ, _VP]),
    "vt_decode_bwd_wide": (_I, [_I, _I, _I, _VP, _I64, _VP, _I, _I, _I, _D, _VP, _VP, _VP, _VP, _VP, _VP, _VP]),
    "vt_decode_mlp_fwd_wide_train": (_I, [_VP, _I, _I, _VP, _I64, _VP, _I, _I, _I, _VP, _VP, _VP, _VP]),
    "vt_decode_mlp_bwd_wide": (_I, [_I, _I, _VP, _I64, _VP, _I, _I, _I, _VP, _VP, _VP, _VP, _VP, _VP]),
    "vt_decode_f16f8_covers": (_I, [_I, _I, _I, _F, _I64, _I64, _D]),
    "vt_decode_fwd_f16f8": (_I, [_VP, _I, _I, _I, _I64, _I, _F, _I64, _VP, _VP, _VP, _I, _VP, _D, _VP, _VP]),
    "vt_decoder_pack_t": (_I, [ctypes.POINTER(DecoderParams), _VP, _SZ, _VP]),
    "vt_decode_save_bytes": (_SZ, [_I64]),
    "vt_decode_gws_bytes": (_SZ, [_I64]),
    "vt_decode_bwd": (_I, [_I, _I, _I, _VP, _I64, _I, _F, _I64, _D, _VP, _VP, _VP, _VP, _VP, _VP, _VP]),
    "vt_decode_wgrad_workspace_bytes": (_SZ, [_I64]),
    "vt_decode_wgrad_floats": (_SZ, [_I]),
    "vt_decode_wgrad": (_I, [_I, _VP, _I64, _I, _F, _I64, _VP, _VP, _VP, _VP, _VP, _SZ, _VP, _VP]),
    "vt_decode_bwd_contact": (_I, [_I, _I, _I, _VP, _I64, _I, _F, _I64, _D, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP]),
    "vt_decode_wgrad_floats_contact": (_SZ, [_I]),
    "vt_decode_wgrad_contact": (_I, [_I, _VP, _I64, _I, _F, _I64, _VP, _VP, _VP, _VP, _VP, _VP, _SZ, _VP, _VP]),
    "vt_mc_workspace_bytes": (_SZ, [_I, _I, _I]),
    "vt_mc_count": (_I, [_VP, _I, _I, _I, _D, _I, _VP, _SZ, _VP]),
    "vt_mc_read_counts": (_I, [_VP, ctypes.POINTER(_I), ctypes.POINTER(_I), ctypes.POINTER(_D), _VP]),
    "vt_mc_count_notify": (_I, [_VP, _I, _I, _I, _D, _I, _VP, _SZ, _VP, ctypes.POINTER(ctypes.c_int)]),
    "vt_mc_echo_slot": (_I, [ctypes.POINTER(ctypes.c_int)]),
    "vt_mc_echo_release": (_I, [_I]),
    "vt_mc_count_echo": (_I, [_VP, _I, _I, _I, _D, _I, _VP, _SZ, _VP, _I]),
    "vt_mc_echo_arm": (_I, [_I]),
    "vt_mc_echo_wait": (_I, [_I, _VP, ctypes.POINTER(_I), ctypes.POINTER(_I), ctypes.POINTER(_D)]),
    "vt_mc_read_counts_begin": (_I, [_VP, _VP, ctypes.POINTER(_I)]),
    "vt_mc_read_counts_end": (_I, [_I, ctypes.POINTER(_I), ctypes.POINTER(_I), ctypes.POINTER(_D)]),
    "vt_mc_emit": (_I, [_VP, _I, _I, _I, _VP, _VP, _I, _VP, _I, _I, _F, _F, _VP]),
    "vt_voxel_build": (_I, [_VP, _I, _I, _I, _D, _VP, _VP, _VP, _VP, _VP]),
    "vt_voxel_build_clear": (_I, [_VP, _I, _I, _I, _D, _VP, _VP, _VP, _VP, _VP, _SZ, _VP]),
    "vt_voxel_build_clear_flags": (_I, [_VP, _I, _I, _I, _D, _VP, _VP, _VP, _VP, _VP, _SZ, _VP, _VP]),
    "vt_voxel_pool_max_fwd": (_I, [_VP, _VP, _VP, _VP, _I, _I, _I, _VP, _VP, _VP]),
    "vt_voxel_pool_mean": (_I, [_VP, _VP, _VP, _VP, _I, _I, _I, _VP, _VP]),
    "vt_voxel_pool_max_bwd": (_I, [_VP, _VP, _VP, _VP, _VP, _I, _I, _I, _VP, _VP]),
    "vt_voxel_pool_max_sum_fwd": (_I, [_VP, _I, _VP, _VP, _VP, _I, _I, _I, _VP, _VP, _VP]),
    "vt_voxel_pool_max_sum_bwd": (_I, [_VP, _I, _VP, _VP, _VP, _VP, _I, _I, _I, _VP, _VP]),
    "vt_voxel_scatter_mean_fwd": (_I, [_VP, _VP, _VP, _VP, _VP, _I, _I, _I, _I, _VP, _VP]),
    "vt_voxel_scatter_mean_bwd": (_I, [_VP, _VP, _VP, _VP, _I, _I, _I, _I, _VP, _VP]),
    "vt_winding_number": (_I, [_VP, _I, _VP, _I, _VP, _I64, _VP, _VP]),
    "vt_winding_number_scenes": (_I, [_VP, _I, _VP, _I64, _VP, _VP]),
    "vt_contact_scan": (_I, [_VP, _VP, _VP, _I, _I, _D, _VP, _VP, _VP]),
    "vt_contact_points": (_I, [_VP, _VP, _VP, _VP, _VP, _VP, _I, _I, _I, _I, _D, _I, _I, _VP, _VP, _VP]),
    "vt_linear_rows": (_I, [_VP, _VP, _VP, _I64, _I, _I, _VP, _VP]),
    "vt_resblock_fc": (_I, [_VP, _I, _VP, _I, _I64, _VP, _VP, _VP, _VP, _VP, _I, _I, _VP, _VP]),
    "vt_pointnet_mlp_stat_blocks": (_I, [_I, _I]),
    "vt_pointnet_mlp_fused": (_I, [_VP, _I, _I, _VP, _VP, _VP, _VP, _VP, _VP, _I, _VP, _VP, _I, _VP, _VP, _VP, _I, _VP, _VP, _VP]),
    "vt_resblock_fc_bwd": (_I, [_VP, _I, _VP, _I, _I64, _VP, _VP, _VP, _VP, _I, _I, _VP, _VP, _VP, _VP, _VP, _VP]),
    "vt_rows_wgrad_workspace_bytes": (_SZ, [_I64, _I, _I]),
    "vt_rows_wgrad": (_I, [_VP, _I, _VP, _I, _VP, _I, _I, _I64, _VP, _SZ, _VP, _VP, _VP]),
    "vt_resblock_wgrad_workspace_bytes": (_SZ, [_I64, _I, _I, _I, _I]),
    "vt_resblock_wgrad": (_I, [_VP, _I, _VP, _I, _I64, _VP, _VP, _VP, _I, _I, _VP, _SZ, _VP, _VP, _VP, _VP, _VP, _VP]),
    "vt_plane_build": (_I, [_VP, _I, _I, _I, _D, _I, _VP, _VP, _VP, _VP, _VP]),
    "vt_plane_build_multi": (_I, [_VP, _I, _I, _I, _D, _I, _VP, _VP, _VP, _VP, _VP, _VP]),
    "vt_plane_scatter_mean_multi_fwd": (_I, [_VP, _I, _VP, _VP, _VP, _VP, _I, _I, _I, _I, _VP, _VP]),
    "vt_plane_scatter_mean_multi_bwd": (_I, [_VP, _I, _VP, _VP, _VP, _I, _I, _I, _I, _VP, _VP]),
    "vt_plane_scatter_mean_fwd": (_I, [_VP, _VP, _VP, _VP, _VP, _I, _I, _I, _I, _VP, _VP]),
    "vt_plane_scatter_mean_bwd": (_I, [_VP, _VP, _VP, _VP, _I, _I, _I, _I, _VP, _VP]),
    "vt_plane_unet_supported": (_I, [_I, _I, _I, _I, _I, _I]),
    "vt_plane_unet_blob_bytes": (_SZ, [_I, _I, _I, _I]),
    "vt_plane_unet_workspace_bytes": (_SZ, [_I, _I, _I, _I, _I, _I, _I]),
    "vt_plane_unet_pack": (_I, [ctypes.POINTER(PlaneUnetParams), _VP, _SZ, _VP]),
    "vt_plane_unet_fwd": (_I, [_VP, _I, _I, _I, ctypes.POINTER(PlaneUnetParams), _VP, _VP, _SZ, _VP, _VP]),
    "vt_plane_unet_bwd_workspace_bytes": (_SZ, [_I, _I, _I, _I, _I, _I, _I]),
    "vt_plane_unet_bwd": (_I, [_VP, _I, _I, _I, ctypes.POINTER(PlaneUnetParams), _VP, _VP, _VP, _VP, _SZ, ctypes.POINTER(PlaneUnetGrads), _VP, _VP]),
    "vt_mano_pack": (_I, [_VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP]),
    "vt_mano_pack_side": (_I, [_VP, _VP, _VP, _VP, _VP, _VP, _VP, _I, _VP, _VP]),
    "vt_mano_fwd": (_I, [_VP, _I, _VP, _I, _VP, _VP, _VP]),
    "vt_mano_bwd": (_I, [_VP, _I, _VP, _I, _VP, _VP, _VP, _VP]),
    "vt_conv3d_packed_floats": (_SZ, [_I, _I]),
    "vt_conv3d_pack": (_I, [_VP, _I, _I, _VP, _VP]),
    "vt_stats_floats": (_SZ, [_I, _I, _I, _I, _I]),
    "vt_conv3d_stat_blocks": (_I, [_I, _I, _I, _I, _I, _I]),
    "vt_decode_mlp_fwd_train": (_I, [_VP, _I, _I, _VP, _I64, _I, _F, _I64, _VP, _VP, _VP, _VP]),
    "vt_decode_mlp_bwd": (_I, [_I, _I, _VP, _I64, _I, _F, _I64, _VP, _VP, _VP, _VP, _VP, _VP]),
    "vt_sample_grid_bwd": (_I, [_I, _I, _I, _VP, _I64, _I, _F, _I64, _D, _VP, _VP, _VP]),
    "vt_sample_grid_bwd_sorted": (_I, [_I, _I, _I, _VP, _I64, _D, _VP, _VP, _VP, _VP, _VP, _VP]),
    "vt_decode_bwd_dc": (_I, [_I, _I, _I, _VP, _I64, _D, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP]),
    "vt_relu_mask": (_I, [_VP, _VP, _VP, _I64, _VP]),
    "vt_relu_mask_absmax": (_I, [_VP, _VP, _VP, _I64, _VP, _VP]),
    "vt_conv3d_wgrad_workspace_bytes": (_SZ, [_I, _I, _I, _I, _I, _I]),
    "vt_conv3d_wgrad": (_I, [_VP, _I, _VP, _I, _I, _I, _I, _I, _VP, _VP, _I, _VP, _SZ, _VP, _VP]),
    "vt_conv3d_wgrad_f16x3_workspace_bytes": (_SZ, [_I, _I, _I, _I, _I, _I]),
    "vt_conv3d_wgrad_f16x3": (_I, [_VP, _I, _VP, _I, _I, _I, _I, _I, _VP, _VP, _I, _VP, _VP, _SZ, _VP, _VP]),
    "vt_conv1x1_bwd_workspace_bytes": (_SZ, []),
    "vt_conv1x1_bwd_masked": (_I, [_VP, _VP, _VP, _I64, _VP, _VP, _VP, _VP, _VP, _SZ, _VP]),
    "vt_conv3d_wgrad_f16x3_up_workspace_bytes": (_SZ, [_I, _I, _I, _I, _I, _I, _I]),
    "vt_conv3d_wgrad_f16x3_up": (_I, [_VP, _I, _VP, _I, _I, _I, _I, _I, _VP, _VP, _I, _VP, _VP, _SZ, _VP, _VP]),
    "vt_conv3d_wgrad_f16x3_sparse_workspace_bytes": (_SZ, [_I, _I, _I, _I, _I, _I]),
    "vt_conv3d_wgrad_f16x3_sparse": (_I, [_VP, _I, _I, _I, _I, _I, _VP, _VP, _VP, _I, _VP, _VP, _SZ, _VP, _VP]),
    "vt_conv3d_xstats_blocks": (_I, [_I, _I, _I, _I, _I, _I]),
    "vt_conv3d_gcr_f16x3_xstats": (_I, [_VP, _I, _I, _I, _I, _I, _VP, _I, _VP, _VP, _VP, _VP, _VP]),
    "vt_gn_bwd_from_part": (_I, [_VP, _I, _VP, _I, _I, _I, _I, _I, _VP, _I, _VP, _I, _VP, _I, _VP, _D, _VP, _I, _VP, _VP, _VP, _VP, _I, _VP, _VP, _VP, _VP]),
    "vt_gn_bwd_masked": (_I, [_VP, _I, _VP, _I, _I, _I, _I, _I, _VP, _I, _VP, _I, _VP, _I, _VP, _D, _VP, _I, _VP, _VP, _VP, _VP, _I, _VP, _VP, _VP, _VP]),
    "vt_gn_bwd": (_I, [_VP, _I, _VP, _I, _I, _I, _I, _I, _VP, _I, _VP, _I, _VP, _I, _VP, _D, _VP, _I, _VP, _VP, _VP, _VP, _VP]),
    "vt_maxpool3d_cl_bwd": (_I, [_VP, _VP, _I, _I, _I, _I, _I, _VP, _VP]),
    "vt_maxpool3d_cl_bwd_fork": (_I, [_VP, _VP, _VP, _I, _I, _I, _I, _I, _VP, _VP, _VP]),
    "vt_conv3d_pack_bf16x3": (_I, [_VP, _I, _I, _VP, _VP]),
    "vt_conv3d_packed_floats_f16x3": (_SZ, [_I, _I]),
    "vt_conv3d_pack_f16x3": (_I, [_VP, _I, _I, _VP, _VP]),
    "vt_conv3d_pack_f16x3_t": (_I, [_VP, _I, _I, _VP, _VP]),
    "vt_conv3d_stat_blocks_f16x3": (_I, [_I, _I, _I, _I, _I, _I]),
    "vt_conv3d_gcr_f16x3": (_I, [_VP, _I, _VP, _I, _I, _I, _I, _I, _VP, _VP, _I, _I, _VP, _VP, _VP]),
    "vt_conv3d_up_packed_floats": (_SZ, [_I, _I]),
    "vt_conv3d_pack_f16x3_up": (_I, [_VP, _I, _I, _I, _VP, _VP]),
    "vt_conv3d_up_covers": (_I, [_I, _I, _I, _I, _I, _I, _I]),
    "vt_conv3d_gcr_f16x3_up": (_I, [_VP, _I, _VP, _I, _I, _I, _I, _I, _VP, _VP, _VP, _I, _I, _VP, _VP, _VP]),
    "vt_conv3d_gcr_f16x3_scaled": (_I, [_VP, _I, _VP, _I, _I, _I, _I, _I, _VP, _VP, _I, _I, _VP, _VP, _VP, _VP]),
    "vt_conv3d_final_fusable": (_I, [_I, _I, _I, _I, _I, _I]),
    "vt_conv1x1_pack_f16x3": (_I, [_VP, _I, _I, _VP, _VP]),
    "vt_conv3d_gcr_f16x3_final": (_I, [_VP, _I, _VP, _I, _I, _I, _I, _I, _VP, _VP, _I, _VP, _VP, _VP, _VP]),
    "vt_conv3d_gcr_f16x3_final_keep": (_I, [_VP, _I, _VP, _I, _I, _I, _I, _I, _VP, _VP, _I, _VP, _VP, _VP, _VP, _VP]),
    "vt_conv3d_stat_blocks_bf16x3": (_I, [_I, _I, _I, _I, _I, _I]),
    "vt_conv3d_gcr_bf16x3": (_I, [_VP, _I, _VP, _I, _I, _I, _I, _I, _VP, _VP, _I, _I, _VP, _VP, _VP]),
    "vt_conv3d_ksplit_workspace_bytes": (_SZ, [_I, _I, _I, _I, _I, _I]),
    "vt_conv3d_stat_blocks_ksplit": (_I, [_I, _I, _I, _I, _I, _I]),
    "vt_conv3d_gcr_bf16x3_ksplit": (_I, [_VP, _I, _VP, _I, _I, _I, _I, _I, _VP, _VP, _I, _I, _VP, _VP, _VP, _SZ, _VP]),
    "vt_conv3d_gcr_f16x3_thin_ksplit": (_I, [_VP, _I, _VP, _I, _I, _I, _I, _I, _VP, _VP, _I, _I, _VP, _VP, _VP, _SZ, _VP]),
    "vt_conv3d_gcr_f16x3_thin": (_I, [_VP, _I, _VP, _I, _I, _I, _I, _I, _VP, _VP, _I, _I, _VP, _VP, _VP]),
    "vt_conv3d_pack_f16x3_thin": (_I, [_VP, _I, _I, _VP, _VP]),
    "vt_channel_stats": (_I, [_VP, _I, _I64, _I, _I, _VP, _VP]),
    "vt_gn_scale_shift": (_I, [_VP, _I, _I, _VP, _I, _I, _I, _I64, _I, _VP, _VP, _D, _VP, _VP]),
    "vt_conv3d_gcr": (_I, [_VP, _I, _VP, _I, _I, _I, _I, _I, _VP, _VP, _I, _I, _VP, _VP, _VP]),
    "vt_unet3d_workspace_bytes": (_SZ, [_I, _I, ctypes.POINTER(UnetParams)]),
    "vt_unet3d_fwd": (_I, [_VP, _I, _I, ctypes.POINTER(UnetParams), _VP, _SZ, _VP, _VP]),
    "vt_unet3d_fwd_stats": (_I, [_VP, _VP, _I, _I, _I, ctypes.POINTER(UnetParams), _VP, _SZ, _VP, _VP]),
    "vt_unet3d_skip_layers": (_I, [_I, _I, ctypes.POINTER(UnetParams)]),
    "vt_unet3d_fwd_skip": (_I, [_VP, _VP, _I, _VP, _I, _I, ctypes.POINTER(UnetParams), _VP, _SZ, _VP, _VP]),
    "vt_conv3d_gcr_f16x3_skip": (_I, [_VP, _I, _I, _I, _I, _I, _VP, _VP, _I, _I, _VP, _VP, _VP, _VP]),
    "vt_voxel_tile_flags": (_I, [_VP, _I, _I, _I, _VP, _VP]),
    "vt_maxpool3d_cl": (_I, [_VP, _I, _I, _I, _I, _I, _VP, _VP]),
    "vt_maxpool3d_cl_stats": (_I, [_VP, _I, _I, _I, _I, _I, _VP, _I, _VP, _VP]),
    "vt_conv1x1_cl": (_I, [_VP, _I64, _I, _VP, _VP, _I, _VP, _VP]),
    "vt_voxel_scatter_mean_cl_fwd": (_I, [_VP, _VP, _VP, _VP, _VP, _I, _I, _I, _I, _VP, _VP]),
    "vt_voxel_scatter_mean_cl_bwd": (_I, [_VP, _VP, _VP, _VP, _I, _I, _I, _I, _VP, _VP]),
}

_lib = None


def load():
    """Load the shared library once; raises VtError with build instructions if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise VtError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C vtaco_amd/csrc` (hipcc, gfx950). There is no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)        # AttributeError if the .so does not export it
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().vt_last_error().decode("utf-8", "replace")
        raise VtError(f"{what} failed (code {rc}): {msg}")


def dev_ptr(t, name="tensor", dtype=torch.float32):
    """Device pointer of a contiguous HIP tensor (raises for CPU tensors)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise VtError(f"{name} must live on a HIP device (got {t.device}); vtaco_amd has no CPU path")
    if t.dtype != dtype:
        raise VtError(f"{name} must be {dtype} (got {t.dtype})")
    if not t.is_contiguous():
        raise VtError(f"{name} must be contiguous")
    return ctypes.c_void_p(t.data_ptr())


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream_ptr():
    """The current HIP stream of the current device as the C ABI's ``void *stream`` (the raw handle straight from the framework where
    it offers it: building a ``torch.cuda.Stream`` object per call cost ~10 us of host time, 470 times per training step)."""
    if _raw_stream is not None:
        return ctypes.c_void_p(_raw_stream(torch.cuda.current_device()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
