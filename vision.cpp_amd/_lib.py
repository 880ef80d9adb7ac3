"""ctypes declarations of the C ABI in lib/libvisioncpp.so.

Mirror of the reference's bindings/python/visioncpp/_lib.py (:36-171): same structures, same
11 `visp_*` prototypes, plus the batched extension (include/visp_c_api.h part 2) and the
kernel-level `vx_*` launchers (include/visp_hip_kernels.h) that the parity tests drive.
There is no fallback: if the library is missing or a symbol cannot be resolved this raises.
"""
from __future__ import annotations

import ctypes
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_int32, c_int64, c_size_t, c_uint8, c_void_p
from pathlib import Path


class Error(Exception):
    pass


class ImageView(ctypes.Structure):  # == visp_image_view
    _fields_ = [("width", c_int32), ("height", c_int32), ("stride", c_int32), ("format", c_int32), ("data", c_void_p)]


class DepthAnyInfo(ctypes.Structure):
    _fields_ = [("patch_size", c_int32), ("embed_dim", c_int32), ("n_layers", c_int32), ("n_heads", c_int32),
                ("image_size", c_int32), ("image_multiple", c_int32), ("feature_layers", c_int32 * 4),
                ("max_depth", c_float)]


class Timing(ctypes.Structure):
    _fields_ = [("name", ctypes.c_char * 32), ("ms", c_float), ("launches", c_int32), ("flops", c_double), ("bytes", c_double)]


class EsrganInfo(ctypes.Structure):  # == visp_esrgan_info
    _fields_ = [(n, c_int32) for n in ("scale", "n_blocks", "n_filters", "growth", "tile_group")]


class DconvArgs(ctypes.Structure):  # == vx_dconv_args
    _fields_ = [
        ("x", c_void_p), ("x_plane", c_int64), ("cin", c_int), ("up2", c_int), ("B", c_int), ("H", c_int), ("W", c_int),
        ("w", c_void_p), ("bias", c_void_p), ("cout", c_int), ("epi", c_int), ("act", c_int),
        ("s1", c_float), ("res1", c_void_p), ("res1_plane", c_int64), ("s2", c_float), ("res2", c_void_p), ("res2_plane", c_int64),
        ("out", c_void_p), ("out_plane", c_int64), ("x_residual", c_int),
        ("x_pix", c_int64), ("out_pix", c_int64), ("res1_pix", c_int64), ("res2_pix", c_int64), ("a_relu", c_int),
        ("head_w", c_void_p), ("head_bias", c_float), ("head_scale", c_float), ("stamps", c_void_p),
        ("bil_hs", c_int), ("bil_ws", c_int), ("res2_hs", c_int), ("res2_ws", c_int),
    ]


class TileLayout(ctypes.Structure):  # == vx_tile_layout
    _fields_ = [(n, c_int) for n in ("image_w", "image_h", "overlap_x", "overlap_y", "n_x", "n_y", "tile_w", "tile_h")]


DC_F16, DC_RGB_F32, DC_HEAD_F32 = 0, 1, 2


class GemmArgs(ctypes.Structure):  # == vx_gemm_args
    _fields_ = [
        ("A", c_void_p), ("lda", c_int64), ("a_group", c_int), ("a_group_stride", c_int), ("a_row_off", c_int),
        ("conv_kh", c_int), ("conv_kw", c_int), ("conv_stride", c_int), ("conv_pad", c_int),
        ("conv_H", c_int), ("conv_W", c_int), ("conv_Cin", c_int), ("conv_OH", c_int), ("conv_OW", c_int),
        ("a_relu", c_int),
        ("W", c_void_p), ("bias", c_void_p), ("M", c_int), ("N", c_int), ("K", c_int),
        ("epi", c_int), ("out", c_void_p), ("ldo", c_int64), ("relu", c_int),
        ("lambda_", c_void_p), ("pos", c_void_p), ("tokens_P", c_int),
        ("q", c_void_p), ("k", c_void_p), ("vt", c_void_p), ("qkv_T", c_int), ("qkv_Tp", c_int), ("qkv_H", c_int),
        ("q_scale", c_float),
        ("ps_s", c_int), ("ps_Cout", c_int), ("ps_H", c_int), ("ps_W", c_int),
        ("res1", c_void_p), ("res2", c_void_p), ("n_valid", c_int), ("stages", c_int), ("head_bias", c_float), ("head_scale", c_float),
        ("post_gelu", c_int), ("win_ws", c_int), ("win_res", c_int), ("win_res_h", c_int), ("win_shift", c_int), ("k_splits", c_int), ("k_partial", c_void_p), ("debug_stamps", c_void_p),
    ]


class GemmFp8Args(ctypes.Structure):  # == vx_gemm_fp8_args
    _fields_ = [("A", c_void_p), ("a_scale", c_void_p), ("W", c_void_p), ("w_scale", c_void_p), ("bias", c_void_p), ("M", c_int), ("N", c_int), ("Kp", c_int),
                ("n_valid", c_int), ("out", c_void_p), ("ldo", c_int64), ("act", c_int), ("res", c_void_p)]


class RcuArgs(ctypes.Structure):  # == vx_rcu_args
    _fields_ = [("x", c_void_p), ("w1", c_void_p), ("b1", c_void_p), ("w2", c_void_p), ("b2", c_void_p), ("res2", c_void_p), ("wp", c_void_p), ("bp", c_void_p),
                ("out", c_void_p), ("B", c_int), ("H", c_int), ("W", c_int)]


class DinoBlockArgs(ctypes.Structure):  # == vx_dino_block_args
    _fields_ = [
        ("att", c_void_p), ("x", c_void_p), ("w_mlp", c_void_p), ("vec_mlp", c_void_p), ("w_qkv", c_void_p), ("vec_qkv", c_void_p),
        ("vec_tap", c_void_p), ("feat", c_void_p), ("q", c_void_p), ("k", c_void_p), ("v", c_void_p),
        ("M", c_int), ("T", c_int), ("H", c_int), ("q_scale", c_float), ("eps", c_float), ("cap_x1", c_void_p), ("stamps", c_void_p),
    ]


EPI_F16, EPI_F16_GELU, EPI_F16_RELU, EPI_RESID_F32, EPI_TOKENS, EPI_QKV, EPI_PIXSHUF, EPI_F16_ADD, EPI_HEAD_OUT = range(9)

import os

# VISP_LIBRARY selects another build of the same ABI (same-box A/B comparisons of kernel variants)
LIB_PATH = Path(os.environ["VISP_LIBRARY"]).resolve() if os.environ.get("VISP_LIBRARY") else Path(__file__).resolve().parent / "lib" / "libvisioncpp.so"

# every symbol include/visp_c_api.h and include/visp_hip_kernels.h declare
C_API_SYMBOLS = [
    "visp_get_last_error", "visp_image_destroy", "visp_backend_load_all", "visp_device_init", "visp_device_destroy",
    "visp_device_type", "visp_device_name", "visp_device_description", "visp_model_detect_family", "visp_model_load",
    "visp_model_destroy", "visp_model_compute",
    "visp_hip_device_init", "visp_model_load_ex", "visp_depthany_weights_arena", "visp_depthany_weights_ready",
    "visp_depthany_get_info", "visp_depthany_image_extent", "visp_depthany_reserve",
    "visp_depthany_compute_batch_device", "visp_depthany_compute_batch_host", "visp_depthany_compute_f32", "visp_depthany_compute_sharded", "visp_depthany_use_graph", "visp_depthany_set_schedule",
    "visp_depthany_pipeline_create", "visp_depthany_pipeline_destroy", "visp_depthany_pipeline_input", "visp_depthany_pipeline_submit", "visp_depthany_pipeline_wait",
    "visp_depthany_set_split", "visp_depthany_enable_captures", "visp_depthany_read_capture", "visp_depthany_enable_timing",
    "visp_depthany_read_timing",
    "visp_esrgan_get_info", "visp_esrgan_set_tile_group", "visp_esrgan_weights_arena", "visp_esrgan_weights_ready",
    "visp_esrgan_tile_layout", "visp_esrgan_compute_batch_device", "visp_esrgan_compute_batch_host",
    "visp_esrgan_generate_host", "visp_esrgan_enable_timing", "visp_esrgan_read_timing",
    "visp_sam_encode", "visp_sam_read_embedding", "visp_sam_encode_batch_device", "visp_sam_encode_batch_host",
    "visp_sam_set_fp8_mlp", "visp_sam_weights_arena", "visp_sam_weights_ready", "visp_sam_enable_timing", "visp_sam_read_timing",
    "visp_sam_enable_captures", "visp_sam_read_capture", "visp_sam_compute", "visp_sam_read_masks", "visp_image_scale", "visp_image_u8_to_f32", "visp_image_normalize", "visp_gguf_validate",
    "visp_birefnet_image_extent", "visp_birefnet_compute_batch_device", "visp_birefnet_compute_batch_host",
    "visp_swin_load", "visp_swin_output_dims", "visp_swin_encode_batch_device", "visp_swin_encode_batch_host", "visp_swin_enable_captures",
    "visp_swin_read_capture", "visp_swin_enable_timing", "visp_swin_read_timing", "visp_swin_set_mask_mode",
    "visp_weights_load", "visp_weights_create", "visp_weights_add", "visp_weights_destroy",
    "visp_file_load", "visp_file_destroy", "visp_file_n_tensors", "visp_file_get_int", "visp_file_get_int_array", "visp_file_get_string", "visp_weights_from_file",
    "visp_graph_create", "visp_graph_destroy", "visp_graph_add_weight", "visp_graph_find_weight", "visp_graph_input",
    "visp_graph_op", "visp_graph_set_name", "visp_graph_get_tensor", "visp_graph_output", "visp_graph_tensor_info", "visp_graph_read_constant",
    "visp_graph_allocate", "visp_graph_set_fused_models", "visp_graph_use_hip_graph", "visp_graph_compute", "visp_graph_tensor_set", "visp_graph_tensor_get", "visp_graph_describe",
]
KERNEL_SYMBOLS = [
    "vx_last_error", "vx_device_count", "vx_set_device", "vx_device_info", "vx_malloc", "vx_free", "vx_memset",
    "vx_memcpy_h2d", "vx_memcpy_d2h", "vx_memcpy_d2d", "vx_malloc_host", "vx_free_host", "vx_memcpy_h2d_async", "vx_memcpy_d2h_async", "vx_event_sync", "vx_stream_create", "vx_stream_destroy", "vx_stream_sync",
    "vx_event_create", "vx_event_destroy", "vx_event_record", "vx_event_elapsed_ms", "vx_stream_wait_event", "vx_graph_begin_capture",
    "vx_graph_end_capture", "vx_graph_launch", "vx_graph_destroy", "vx_gemm_f16", "vx_gemm_pick_k_splits", "vx_gemm_fp8_supported", "vx_gemm_fp8", "vx_quantize_rows_e4m3", "vx_quantize_rows_e4m3_host", "vx_conv3x3_supported", "vx_conv3x3_f16", "vx_rcu_supported", "vx_rcu_fused_f16", "vx_nearest_f16", "vx_image_planes_f32", "vx_esrgan_tiles_in_f32",
    "vx_attention_f16", "vx_attention_set_fast_limit", "vx_attention_set_stamps",
    "vx_layernorm_f32_f16", "vx_preprocess_patches", "vx_preprocess_f32", "vx_write_cls_rows", "vx_bilinear_ac_f16",
    "vx_head_out_f32", "vx_minmax_normalize", "vx_f32_to_u8",
    "vx_dconv3x3_f16", "vx_dconv_bilinear_supported", "vx_dconv_prepare", "vx_tv_preprocess", "vx_dwconv3x3_f16", "vx_layernorm_f16", "vx_window_attention_f16", "vx_window_attention_bias_bytes", "vx_window_attention_pack_bias",
    "vx_window_reverse_add_f16", "vx_add_gelu_f16", "vx_add_rows_f16", "vx_small_attention_f16", "vx_sam_interpolate", "vx_mbconv_dw_pw_supported", "vx_mbconv_pack_w3", "vx_mbconv_dw_pw_f16", "vx_esrgan_tiles_in", "vx_esrgan_tiles_out",
    "vx_bf_preprocess_half", "vx_bf_patches", "vx_bf_resize_f16", "vx_bf_deform_cols_f16", "vx_bf_mean_f16", "vx_bf_broadcast_f16", "vx_bf_mul_sigmoid_f16", "vx_bf_sigmoid_out_f32",
    "vx_swin_attention_pack_bias", "vx_window_attention_masked_f16", "vx_swin_layernorm_f16", "vx_swin_layernorm_strided_f16", "vx_swin_merge_layernorm_f16", "vx_swin_window_reverse_add_f16",
    "vx_headconv_frag_bytes", "vx_headconv_pack", "vx_headconv_supported", "vx_headconv_bil_f16", "vx_headconv_set_stamps",
    "vx_copy_strided_f16", "vx_binary_rows", "vx_unary_f16", "vx_convert", "vx_im2col_patches_f32", "vx_conv1x1_to1_f32",
    "vx_dino_block_supported", "vx_dino_block_mlp_bytes", "vx_dino_block_qkv_bytes", "vx_dino_block16_pack_mlp", "vx_dino_block16_pack_qkv", "vx_dino_block16_f16",
]


def init() -> ctypes.CDLL:
    if not LIB_PATH.exists():
        raise OSError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                      f"(make -C vision.cpp_amd/csrc). There is no CPU fallback.")
    lib = ctypes.CDLL(str(LIB_PATH))
    for sym in C_API_SYMBOLS + KERNEL_SYMBOLS:
        getattr(lib, sym)  # AttributeError if the build dropped a declared symbol

    lib.visp_get_last_error.restype = c_char_p
    lib.visp_backend_load_all.argtypes = [c_char_p]
    lib.visp_backend_load_all.restype = c_int32
    lib.visp_image_destroy.argtypes = [c_void_p]
    lib.visp_image_destroy.restype = None
    lib.visp_device_init.argtypes = [c_int32, POINTER(c_void_p)]
    lib.visp_device_init.restype = c_int32
    lib.visp_hip_device_init.argtypes = [c_int32, POINTER(c_void_p)]
    lib.visp_hip_device_init.restype = c_int32
    lib.visp_device_destroy.argtypes = [c_void_p]
    lib.visp_device_destroy.restype = None
    lib.visp_device_type.argtypes = [c_void_p]
    lib.visp_device_type.restype = c_int32
    lib.visp_device_name.argtypes = [c_void_p]
    lib.visp_device_name.restype = c_char_p
    lib.visp_device_description.argtypes = [c_void_p]
    lib.visp_device_description.restype = c_char_p
    lib.visp_model_detect_family.argtypes = [c_char_p, POINTER(c_int32)]
    lib.visp_model_detect_family.restype = c_int32
    lib.visp_model_load.argtypes = [c_char_p, c_void_p, c_int32, POINTER(c_void_p)]
    lib.visp_model_load.restype = c_int32
    lib.visp_model_load_ex.argtypes = [c_char_p, c_void_p, c_int32, c_int32, POINTER(c_void_p)]
    lib.visp_model_load_ex.restype = c_int32
    lib.visp_model_destroy.argtypes = [c_void_p, c_int32]
    lib.visp_model_destroy.restype = None
    lib.visp_model_compute.argtypes = [c_void_p, c_int32, POINTER(ImageView), c_int32, POINTER(c_int32), c_int32,
                                       POINTER(ImageView), POINTER(c_void_p)]
    lib.visp_model_compute.restype = c_int32

    lib.visp_depthany_weights_arena.argtypes = [c_void_p, POINTER(c_void_p), POINTER(c_size_t)]
    lib.visp_depthany_weights_ready.argtypes = [c_void_p]
    lib.visp_depthany_get_info.argtypes = [c_void_p, POINTER(DepthAnyInfo)]
    lib.visp_depthany_image_extent.argtypes = [c_void_p, c_int32, c_int32, POINTER(c_int32), POINTER(c_int32)]
    lib.visp_depthany_reserve.argtypes = [c_void_p, c_int32, c_int32, c_int32]
    lib.visp_depthany_compute_batch_device.argtypes = [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p]
    lib.visp_depthany_compute_batch_host.argtypes = [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p]
    lib.visp_depthany_compute_f32.argtypes = [c_void_p, POINTER(ImageView), POINTER(ImageView), POINTER(c_void_p)]
    lib.visp_depthany_compute_sharded.argtypes = [POINTER(c_void_p), c_int32, c_void_p, c_int32, c_int32, c_int32, c_void_p]
    lib.visp_depthany_use_graph.argtypes = [c_void_p, c_int32]
    lib.visp_depthany_set_schedule.argtypes = [c_void_p, c_int32]
    lib.visp_depthany_pipeline_create.argtypes = [c_void_p, c_int32, c_int32, c_int32, c_int32, POINTER(c_void_p)]
    lib.visp_depthany_pipeline_destroy.argtypes = [c_void_p]
    lib.visp_depthany_pipeline_destroy.restype = None
    lib.visp_depthany_pipeline_input.argtypes = [c_void_p, POINTER(c_void_p)]
    lib.visp_depthany_pipeline_submit.argtypes = [c_void_p, c_void_p, POINTER(c_int32)]
    lib.visp_depthany_pipeline_wait.argtypes = [c_void_p, c_int32, POINTER(c_void_p)]
    lib.visp_depthany_enable_captures.argtypes = [c_void_p, c_int32]
    lib.visp_depthany_read_capture.argtypes = [c_void_p, c_char_p, c_void_p, c_int64, POINTER(c_int64), POINTER(c_int64)]
    lib.visp_depthany_enable_timing.argtypes = [c_void_p, c_int32]
    lib.visp_depthany_read_timing.argtypes = [c_void_p, POINTER(Timing), c_int32, POINTER(c_int32)]
    lib.visp_esrgan_get_info.argtypes = [c_void_p, POINTER(EsrganInfo)]
    lib.visp_esrgan_set_tile_group.argtypes = [c_void_p, c_int32]
    lib.visp_esrgan_weights_arena.argtypes = [c_void_p, POINTER(c_void_p), POINTER(c_size_t)]
    lib.visp_esrgan_weights_ready.argtypes = [c_void_p]
    lib.visp_esrgan_tile_layout.argtypes = [c_int32, c_int32, c_int32, POINTER(c_int32)]
    lib.visp_esrgan_compute_batch_device.argtypes = [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p]
    lib.visp_esrgan_compute_batch_host.argtypes = [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p]
    lib.visp_esrgan_generate_host.argtypes = [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p]
    lib.visp_esrgan_enable_timing.argtypes = [c_void_p, c_int32]
    lib.visp_esrgan_read_timing.argtypes = [c_void_p, POINTER(Timing), c_int32, POINTER(c_int32)]
    lib.visp_sam_encode.argtypes = [c_void_p, POINTER(ImageView)]
    lib.visp_sam_read_embedding.argtypes = [c_void_p, c_void_p, c_int64, POINTER(c_int64)]
    lib.visp_sam_encode_batch_device.argtypes = [c_void_p, c_void_p, c_int32, c_void_p, c_void_p]
    lib.visp_sam_encode_batch_host.argtypes = [c_void_p, c_void_p, c_int32, c_void_p]
    lib.visp_sam_weights_arena.argtypes = [c_void_p, POINTER(c_void_p), POINTER(c_size_t)]
    lib.visp_sam_set_fp8_mlp.argtypes = [c_void_p, c_int32]
    lib.visp_sam_weights_ready.argtypes = [c_void_p]
    lib.visp_sam_compute.argtypes = [c_void_p, POINTER(c_int32), c_int32, POINTER(ImageView), POINTER(c_void_p)]
    lib.visp_sam_read_masks.argtypes = [c_void_p, c_void_p, c_int64, POINTER(ctypes.c_float)]
    lib.visp_sam_enable_captures.argtypes = [c_void_p, c_int32]
    lib.visp_sam_read_capture.argtypes = [c_void_p, c_char_p, c_void_p, c_int64, POINTER(c_int64), POINTER(c_int64)]
    lib.visp_sam_enable_timing.argtypes = [c_void_p, c_int32]
    lib.visp_sam_read_timing.argtypes = [c_void_p, POINTER(Timing), c_int32, POINTER(c_int32)]
    lib.visp_gguf_validate.argtypes = [c_char_p, POINTER(c_int32)]
    lib.visp_birefnet_image_extent.argtypes = [c_void_p, c_int32, c_int32, POINTER(c_int32), POINTER(c_int32)]
    lib.visp_birefnet_compute_batch_device.argtypes = [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p]
    lib.visp_birefnet_compute_batch_host.argtypes = [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p]
    lib.visp_swin_load.argtypes = [c_char_p, c_void_p, POINTER(c_void_p)]
    lib.visp_swin_output_dims.argtypes = [c_void_p, c_int32, c_int32, POINTER(c_int32)]
    lib.visp_swin_encode_batch_device.argtypes = [c_void_p, c_void_p, c_int32, c_int32, c_int32, POINTER(c_void_p), c_void_p]
    lib.visp_swin_encode_batch_host.argtypes = [c_void_p, c_void_p, c_int32, c_int32, c_int32, POINTER(c_void_p)]
    lib.visp_swin_enable_captures.argtypes = [c_void_p, c_int32]
    lib.visp_swin_set_mask_mode.argtypes = [c_void_p, c_int32]
    lib.visp_depthany_set_split.argtypes = [c_void_p, c_int32]
    lib.visp_swin_read_capture.argtypes = [c_void_p, c_char_p, c_void_p, c_int64, POINTER(c_int64), POINTER(c_int64)]
    lib.visp_swin_enable_timing.argtypes = [c_void_p, c_int32]
    lib.visp_swin_read_timing.argtypes = [c_void_p, POINTER(Timing), c_int32, POINTER(c_int32)]
    lib.visp_image_scale.argtypes = [POINTER(ImageView), c_int32, c_int32, POINTER(ImageView), POINTER(c_void_p)]
    lib.visp_image_u8_to_f32.argtypes = [POINTER(ImageView), c_int32, POINTER(c_float), POINTER(c_float), POINTER(ImageView), POINTER(c_void_p)]
    lib.visp_image_normalize.argtypes = [POINTER(ImageView), c_float, c_float, POINTER(ImageView), POINTER(c_void_p)]
    lib.visp_graph_create.argtypes = [c_void_p, POINTER(c_void_p)]
    lib.visp_graph_destroy.argtypes = [c_void_p]
    lib.visp_weights_load.argtypes = [c_char_p, POINTER(c_void_p)]
    lib.visp_weights_create.argtypes = [POINTER(c_void_p)]
    lib.visp_file_load.argtypes = [c_char_p, POINTER(c_void_p)]
    lib.visp_file_destroy.argtypes = [c_void_p]
    lib.visp_file_n_tensors.argtypes = [c_void_p, POINTER(c_int64)]
    lib.visp_file_get_int.argtypes = [c_void_p, c_char_p, POINTER(c_int32)]
    lib.visp_file_get_int_array.argtypes = [c_void_p, c_char_p, POINTER(c_int32), c_int64]
    lib.visp_file_get_string.argtypes = [c_void_p, c_char_p, c_char_p, c_int64, POINTER(c_int64)]
    lib.visp_weights_from_file.argtypes = [c_void_p, POINTER(c_void_p)]
    lib.visp_weights_add.argtypes = [c_void_p, c_char_p, c_int32, POINTER(c_int64), c_void_p]
    lib.visp_weights_destroy.argtypes = [c_void_p]
    lib.visp_graph_add_weight.argtypes = [c_void_p, c_char_p, c_int32, POINTER(c_int64), c_void_p, POINTER(c_int32)]
    lib.visp_graph_find_weight.argtypes = [c_void_p, c_char_p, POINTER(c_int32)]
    lib.visp_graph_input.argtypes = [c_void_p, c_int32, POINTER(c_int64), c_char_p, POINTER(c_int32)]
    lib.visp_graph_op.argtypes = [c_void_p, c_int32, POINTER(c_int32), c_int32, POINTER(c_int64), c_int32, POINTER(c_float), c_int32, POINTER(c_int32)]
    lib.visp_graph_set_name.argtypes = [c_void_p, c_int32, c_char_p]
    lib.visp_graph_get_tensor.argtypes = [c_void_p, c_char_p, POINTER(c_int32)]
    lib.visp_graph_output.argtypes = [c_void_p, c_int32, c_char_p]
    lib.visp_graph_tensor_info.argtypes = [c_void_p, c_int32, POINTER(c_int32), POINTER(c_int64), POINTER(c_int32)]
    lib.visp_graph_read_constant.argtypes = [c_void_p, c_int32, c_void_p, c_int64]
    lib.visp_graph_allocate.argtypes = [c_void_p, c_void_p]
    lib.visp_graph_use_hip_graph.argtypes = [c_void_p, c_int32]
    lib.visp_graph_compute.argtypes = [c_void_p]
    lib.visp_graph_tensor_set.argtypes = [c_void_p, c_int32, c_void_p, c_size_t]
    lib.visp_graph_tensor_get.argtypes = [c_void_p, c_int32, c_void_p, c_size_t, c_int32]
    lib.visp_graph_describe.argtypes = [c_void_p, c_char_p, c_int64, POINTER(c_int64)]
    lib.visp_graph_set_fused_models.argtypes = [c_void_p, c_int32]
    for name in C_API_SYMBOLS[15:]:
        getattr(lib, name).restype = c_int32
    lib.visp_depthany_pipeline_destroy.restype = None
    lib.visp_graph_destroy.restype = None
    lib.visp_weights_destroy.restype = None
    lib.visp_file_destroy.restype = None

    lib.vx_last_error.restype = c_char_p
    lib.vx_device_info.argtypes = [c_int, c_char_p, c_int, c_char_p, c_int, POINTER(c_size_t), POINTER(c_size_t), POINTER(c_int)]
    lib.vx_malloc.argtypes = [POINTER(c_void_p), c_size_t]
    lib.vx_free.argtypes = [c_void_p]
    lib.vx_memset.argtypes = [c_void_p, c_int, c_size_t, c_void_p]
    lib.vx_memcpy_h2d.argtypes = [c_void_p, c_void_p, c_size_t, c_void_p]
    lib.vx_memcpy_d2h.argtypes = [c_void_p, c_void_p, c_size_t, c_void_p]
    lib.vx_memcpy_d2d.argtypes = [c_void_p, c_void_p, c_size_t, c_void_p]
    lib.vx_stream_create.argtypes = [POINTER(c_void_p)]
    lib.vx_stream_destroy.argtypes = [c_void_p]
    lib.vx_stream_sync.argtypes = [c_void_p]
    lib.vx_event_create.argtypes = [POINTER(c_void_p)]
    lib.vx_event_destroy.argtypes = [c_void_p]
    lib.vx_event_record.argtypes = [c_void_p, c_void_p]
    lib.vx_event_elapsed_ms.argtypes = [c_void_p, c_void_p, POINTER(c_float)]
    lib.vx_stream_wait_event.argtypes = [c_void_p, c_void_p]
    lib.vx_graph_begin_capture.argtypes = [c_void_p]
    lib.vx_graph_end_capture.argtypes = [c_void_p, POINTER(c_void_p)]
    lib.vx_graph_launch.argtypes = [c_void_p, c_void_p]
    lib.vx_graph_destroy.argtypes = [c_void_p]
    lib.vx_gemm_f16.argtypes = [POINTER(GemmArgs), c_void_p]
    lib.vx_gemm_fp8_supported.argtypes = [c_int, c_int]
    lib.vx_gemm_fp8.argtypes = [POINTER(GemmFp8Args), c_void_p]
    lib.vx_rcu_supported.argtypes = [c_int, c_int]
    lib.vx_rcu_fused_f16.argtypes = [POINTER(RcuArgs), c_void_p]
    lib.vx_quantize_rows_e4m3.argtypes = [c_void_p, c_int64, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]
    lib.vx_quantize_rows_e4m3_host.argtypes = [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]
    lib.vx_gemm_pick_k_splits.argtypes = [c_int, c_int, c_int]
    lib.vx_conv3x3_supported.argtypes = [POINTER(GemmArgs)]
    lib.vx_conv3x3_f16.argtypes = [POINTER(GemmArgs), c_void_p]
    lib.vx_attention_f16.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]
    lib.vx_attention_set_fast_limit.argtypes = [c_float]
    lib.vx_attention_set_stamps.argtypes = [c_void_p]
    lib.vx_headconv_frag_bytes.restype = c_size_t
    lib.vx_headconv_set_stamps.argtypes = [c_void_p]
    lib.vx_headconv_pack.argtypes = [c_void_p, c_int, c_void_p]
    lib.vx_headconv_supported.argtypes = [c_int, c_int, c_int, c_int, c_int, c_int]
    lib.vx_headconv_bil_f16.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_float, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]
    lib.vx_layernorm_f32_f16.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_void_p]
    lib.vx_preprocess_patches.argtypes = [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, POINTER(c_float), POINTER(c_float), c_void_p]
    lib.vx_preprocess_f32.argtypes = [c_void_p, c_void_p, c_int, c_int, c_int, POINTER(c_float), POINTER(c_float), c_void_p]
    lib.vx_write_cls_rows.argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]
    lib.vx_bilinear_ac_f16.argtypes = [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]
    lib.vx_head_out_f32.argtypes = [c_void_p, c_void_p, c_float, c_float, c_void_p, c_int64, c_int, c_void_p]
    lib.vx_minmax_normalize.argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_int64, c_void_p]
    lib.vx_f32_to_u8.argtypes = [c_void_p, c_void_p, c_int64, c_void_p]
    lib.vx_tv_preprocess.argtypes = [c_void_p, c_void_p, c_int64, c_void_p]
    lib.vx_dwconv3x3_f16.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]
    lib.vx_layernorm_f16.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_float, c_int, c_int, c_int, c_void_p]
    lib.vx_window_attention_f16.argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]
    lib.vx_bf_preprocess_half.argtypes = [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]
    lib.vx_bf_patches.argtypes = [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]
    lib.vx_bf_resize_f16.argtypes = [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]
    lib.vx_bf_deform_cols_f16.argtypes = [c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]
    lib.vx_bf_mean_f16.argtypes = [c_void_p, c_int, c_void_p, c_void_p, c_int, c_int64, c_int, c_void_p]
    lib.vx_bf_broadcast_f16.argtypes = [c_void_p, c_int, c_void_p, c_int, c_int, c_int64, c_int, c_void_p]
    lib.vx_bf_mul_sigmoid_f16.argtypes = [c_void_p, c_int, c_void_p, c_int, c_int64, c_int, c_void_p]
    lib.vx_bf_sigmoid_out_f32.argtypes = [c_void_p, c_int, c_void_p, c_int64, c_void_p]
    lib.vx_swin_attention_pack_bias.argtypes = [c_void_p, c_int, c_int, c_void_p]
    lib.vx_window_attention_masked_f16.argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]
    lib.vx_swin_layernorm_f16.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_float, c_int, c_int, c_int, c_int, c_int, c_void_p]
    lib.vx_swin_layernorm_strided_f16.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_float, c_int, c_int, c_void_p]
    lib.vx_swin_merge_layernorm_f16.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_void_p]
    lib.vx_swin_window_reverse_add_f16.argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]
    lib.vx_add_rows_f16.argtypes = [c_void_p, c_int, c_void_p, c_int64, c_void_p, c_int64, c_void_p]
    lib.vx_small_attention_f16.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]
    lib.vx_sam_interpolate.argtypes = [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int, c_int, c_int, c_void_p]
    lib.vx_mbconv_pack_w3.argtypes = [c_void_p, c_void_p]
    lib.vx_mbconv_dw_pw_supported.argtypes = [c_int, c_int, c_int]
    lib.vx_mbconv_dw_pw_f16.argtypes = [c_void_p] * 7 + [c_int] * 5 + [c_void_p]
    lib.vx_window_attention_bias_bytes.argtypes = [c_int, c_int]
    lib.vx_window_attention_bias_bytes.restype = c_size_t
    lib.vx_window_attention_pack_bias.argtypes = [c_void_p, c_int, c_int, c_void_p]
    lib.vx_window_reverse_add_f16.argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]
    lib.vx_add_gelu_f16.argtypes = [c_void_p, c_void_p, c_void_p, c_int64, c_void_p]
    lib.vx_dconv3x3_f16.argtypes = [POINTER(DconvArgs), c_void_p]
    lib.vx_dconv_bilinear_supported.argtypes = [c_int, c_int, c_int, c_int, c_int]
    lib.vx_esrgan_tiles_in.argtypes = [c_void_p, c_int, c_int, c_int, c_int, POINTER(TileLayout), c_void_p, c_void_p]
    lib.vx_esrgan_tiles_out.argtypes = [c_void_p, c_int, POINTER(TileLayout), c_void_p, c_void_p, c_void_p]
    for name in KERNEL_SYMBOLS[1:]:
        getattr(lib, name).restype = c_int
    lib.vx_attention_set_fast_limit.restype = None
    lib.vx_dino_block_supported.argtypes = [c_int, c_int, c_int]
    lib.vx_dino_block_mlp_bytes.argtypes = []
    lib.vx_dino_block_mlp_bytes.restype = c_size_t
    lib.vx_dino_block_qkv_bytes.argtypes = []
    lib.vx_dino_block_qkv_bytes.restype = c_size_t
    lib.vx_dino_block16_pack_mlp.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p]
    lib.vx_dino_block16_pack_qkv.argtypes = [c_void_p, c_void_p]
    lib.vx_dino_block16_f16.argtypes = [POINTER(DinoBlockArgs), c_void_p]
    return lib


_lib: ctypes.CDLL | None = None


def get_lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        _lib = init()
    return _lib


def check(return_value: int):
    if return_value == 0:
        raise Error(get_lib().visp_get_last_error().decode())


def vx_check(return_value: int):
    if return_value == 0:
        raise Error(get_lib().vx_last_error().decode())


def path_to_char_p(p) -> bytes:
    return str(p).encode()
