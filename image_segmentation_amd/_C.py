"""ctypes binding of liblsam.so (include/lsam.h).  No torch types cross the boundary:
only raw device pointers, sizes and the HIP stream handle.

The product path fails loudly if the HIP library is missing -- there is no CPU or
PyTorch fallback anywhere in this package."""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liblsam.so")

F32, BF16 = 0, 1
EPI_GELU, EPI_GELU_BWD, EPI_OUT_F32, EPI_SAVE_PRE, EPI_RELU, EPI_SIGMOID, EPI_FEW_ROWS = 1, 2, 4, 8, 16, 32, 64


class LsamError(RuntimeError):
    pass


class HieraConfig(C.Structure):
    _fields_ = [("dtype", C.c_int), ("img_size", C.c_int), ("embed_dim", C.c_int),
                ("num_heads", C.c_int), ("num_stages", C.c_int), ("stages", C.c_int * 4),
                ("window_spec", C.c_int * 4), ("n_global", C.c_int),
                ("global_att_blocks", C.c_int * 8), ("q_pool", C.c_int), ("lora_rank", C.c_int),
                ("n_lora_blocks", C.c_int)]


class BlockParams(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "norm1_w", "norm1_b", "qkv_w", "qkv_b", "attn_proj_w", "attn_proj_b", "norm2_w", "norm2_b",
        "fc1_w", "fc1_b", "fc2_w", "fc2_b", "proj_w", "proj_b")]


class LoraParams(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("a_q", "b_q", "a_v", "b_v")]


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise LsamError(
                f"{LIB_PATH} not found: build it with `make -C image_segmentation_amd/csrc` "
                "(or __graft_entry__.build()).  There is no fallback path.")
        L = C.CDLL(LIB_PATH)
        L.lsam_last_error.restype = C.c_char_p
        for name in ("lsam_hiera_arena_bytes", "lsam_hiera_workspace_bytes", "lsam_outer_reduce_ws",
                     "lsam_window_block_image_bytes", "lsam_window_block_bias_floats",
                     "lsam_mlp_block_image_bytes", "lsam_mlp_block_bias_floats",
                     "lsam_tail_block_image_bytes", "lsam_tail_block_vec_floats", "lsam_patch_embed_image_bytes",
                     "lsam_lateral_conv_image_bytes"):
            getattr(L, name).restype = C.c_size_t
        L.lsam_hiera_arena_bytes.argtypes = [C.c_void_p]
        L.lsam_hiera_workspace_bytes.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.lsam_outer_reduce_ws.argtypes = [C.c_int, C.c_int]
        L.lsam_hiera_destroy.restype = None
        L.lsam_hiera_destroy.argtypes = [C.c_void_p]
        L.lsam_hiera_create.argtypes = [C.POINTER(HieraConfig), C.POINTER(C.c_void_p)]
        L.lsam_hiera_num_blocks.argtypes = [C.c_void_p]
        L.lsam_hiera_block_info.argtypes = [C.c_void_p, C.c_int] + [C.POINTER(C.c_int)] * 6
        L.lsam_hiera_stage_info.argtypes = [C.c_void_p, C.c_int] + [C.POINTER(C.c_int)] * 2
        L.lsam_hiera_pack.argtypes = [C.c_void_p, C.POINTER(BlockParams), C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        L.lsam_hiera_pack_lora.argtypes = [C.c_void_p, C.POINTER(LoraParams), C.c_void_p, C.c_void_p]
        L.lsam_hiera_forward.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_void_p),
                                         C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
        L.lsam_hiera_backward.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_void_p),
                                          C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        L.lsam_hiera_backward_block.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p),
                                                C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        vp, i, f, l = C.c_void_p, C.c_int, C.c_float, C.c_long
        L.lsam_linear.argtypes = [i, vp, i, vp, i, vp, vp, i, i, vp, i, vp, i, vp, i, i, i, i, i, vp]
        L.lsam_linear_maxpool2x2.argtypes = [vp, i, vp, i, vp, vp, i, i, i, i, i, i, vp]
        L.lsam_pack_matrix.argtypes = [i, vp, l, l, i, i, vp, i, vp]
        L.lsam_layernorm_fwd.argtypes = [i, vp, i, vp, vp, vp, i, vp, i, i, f, vp]
        L.lsam_layernorm_bwd.argtypes = [i, vp, i, vp, i, vp, vp, vp, i, vp, i, vp, i, i, i, vp]
        L.lsam_attention_fwd.argtypes = [i, vp, i, vp, vp, vp, i, vp, i, i, i, i, i, i, i, vp]
        L.lsam_attention_fwd_pooled_q.argtypes = [i, vp, i, vp, vp, vp, i, vp, i, vp, i, i, i, i, i, i, vp]
        L.lsam_attention_bwd.argtypes = [i, vp, i, vp, vp, i, vp, vp, i, i, i, i, i, i, i, vp]
        L.lsam_attention_bwd_qpool.argtypes = [i, vp, i, vp, vp, i, vp, vp, i, i, i, i, i, i, i, i, vp]
        L.lsam_maxpool2x2.argtypes = [vp, vp, i, i, i, i, vp]
        L.lsam_spatial_mean.argtypes = [vp, vp, i, i, i, vp]
        L.lsam_cast.argtypes = [i, vp, i, vp, i, i, i, vp]
        L.lsam_im2col_patch.argtypes = [i, vp, vp, i, i, i, vp]
        L.lsam_outer_reduce.argtypes = [i, vp, i, vp, i, i, i, i, vp, l, l, vp, vp]
        L.lsam_outer_reduce_static.argtypes = [i, vp, i, vp, i, i, i, i, vp, l, l, vp]
        L.lsam_outer_reduce_static_max_ws.restype = C.c_size_t
        L.lsam_outer_reduce_static_max_ws.argtypes = []
        L.lsam_outer_reduce_static_init.argtypes = []
        for name in ("lsam_window_block_image_bytes", "lsam_window_block_bias_floats",
                     "lsam_mlp_block_image_bytes", "lsam_mlp_block_bias_floats",
                     "lsam_tail_block_image_bytes", "lsam_tail_block_vec_floats", "lsam_patch_embed_image_bytes"):
            getattr(L, name).argtypes = [i]
        L.lsam_add_rows.argtypes = [i, vp, i, vp, i, i, i, vp, i, i, i, vp]
        L.lsam_mha_fwd.argtypes = [i, vp, i, l, vp, i, l, vp, i, l, vp, i, l, i, i, i, i, i, vp]
        L.lsam_pe_encode.argtypes = [vp, vp, vp, i, i, vp]
        L.lsam_upscale2x2_fwd.argtypes = [i, vp, vp, vp, vp, f, vp, i, i, i, i, vp]
        L.lsam_mask_gate.argtypes = [vp, i, vp, vp, i, i, i, i, f, vp]
        L.lsam_bilinear_upsample.argtypes = [vp, vp, i, i, i, i, vp]
        L.lsam_lateral_conv_image_bytes.argtypes = [i, i]
        L.lsam_neck_conv_image_bytes.restype = C.c_size_t
        L.lsam_neck_conv_image_bytes.argtypes = [i, i, i]
        L.lsam_neck_conv_pack.argtypes = [i, vp, i, i, vp, vp]
        L.lsam_neck_level_fwd.argtypes = [i, vp, vp, vp, vp, vp, i, i, i, i, i, vp]
        L.lsam_neck_topdown_bwd.argtypes = [i, vp, vp, vp, i, i, i, i, vp]
        L.lsam_act_bwd.argtypes = [i, vp, i, vp, vp, C.c_long, vp]
        L.lsam_mha_bwd.argtypes = [i, vp, vp, vp, vp, vp, vp, vp, vp, i, i, i, i, i, vp]
        L.lsam_upscale2x2_bwd.argtypes = [i, vp, vp, vp, vp, vp, C.c_float, vp, i, i, i, i, vp]
        L.lsam_mask_gate_bwd.argtypes = [vp, vp, vp, i, i, i, i, i, vp]
        L.lsam_bilinear_upsample_bwd.argtypes = [vp, vp, i, i, i, i, vp]
        L.lsam_batched_outer_ws_floats.restype = C.c_size_t
        L.lsam_batched_outer_ws_floats.argtypes = [i, i, i]
        L.lsam_batched_outer.argtypes = [i, vp, vp, vp, vp, i, i, i, i, vp]
        L.lsam_mask_loss_ws_floats.restype = C.c_size_t
        L.lsam_mask_loss_ws_floats.argtypes = [i, i]
        L.lsam_mask_loss_sums.argtypes = [vp, vp, vp, vp, i, i, i, C.c_float, C.c_float, vp]
        L.lsam_mask_loss_bwd.argtypes = [vp, vp, vp, vp, i, i, i, C.c_float, C.c_float, vp]
        L.lsam_kv_window_attention_image_bytes.restype = C.c_size_t
        L.lsam_kv_window_attention_image_bytes.argtypes = [i, i, i]
        L.lsam_kv_window_attention_pack.argtypes = [vp, vp, i, i, i, vp, vp]
        L.lsam_kv_window_attention_fwd.argtypes = [vp, i, vp, i, vp, vp, i, i, i, i, i, i, i, vp]
        L.lsam_neck_conv_split_image_bytes.restype = C.c_size_t
        L.lsam_neck_conv_split_image_bytes.argtypes = [i, i]
        L.lsam_neck_conv_pack_split.argtypes = [vp, i, i, vp, vp]
        L.lsam_neck_level_fwd_split.argtypes = [vp, vp, vp, vp, vp, vp, i, i, i, i, i, vp]
        L.lsam_qkv_window4_attention_image_bytes.restype = C.c_size_t
        L.lsam_qkv_window4_attention_image_bytes.argtypes = [i, i, i, i]
        L.lsam_qkv_window4_attention_pack.argtypes = [vp, vp, i, i, i, i, vp, vp]
        L.lsam_qkv_window4_attention_fwd.argtypes = [vp, i, vp, i, vp, vp, i, i, i, i, i, i, i, vp]
        L.lsam_stream_linear_image_bytes.restype = C.c_size_t
        L.lsam_stream_linear_image_bytes.argtypes = [i, i]
        L.lsam_stream_linear_pack.argtypes = [vp, i, i, vp, vp]
        L.lsam_stream_linear_fwd.argtypes = [vp, i, vp, vp, vp, i, i, i, i, vp]
        L.lsam_lateral_conv_pack.argtypes = [vp, i, i, vp, vp]
        L.lsam_lateral_conv_fwd.argtypes = [vp, vp, vp, vp, i, i, i, vp]
        L.lsam_patch_embed_pack.argtypes = [vp, i, vp, vp]
        L.lsam_patch_embed_fwd.argtypes = [vp, vp, vp, vp, vp, i, i, i, vp]
        L.lsam_tail_block_pack.argtypes = [vp] * 8 + [i, vp, vp, vp]
        L.lsam_tail_block_fwd.argtypes = [vp, i, vp, vp, vp, vp, vp, i, vp, vp, i, i, f, vp]
        L.lsam_window_block_pack.argtypes = [vp, vp, vp, vp, i, i, vp, vp, vp]
        L.lsam_window_block_fwd.argtypes = [vp, vp, vp, vp, vp, vp, i, i, i, i, i, f, vp]
        L.lsam_mlp_block_pack.argtypes = [vp, vp, vp, vp, i, vp, vp, vp]
        L.lsam_window_block_attention_fwd.argtypes = [vp, vp, vp, vp, vp, vp, i, i, i, i, i, f, vp]
        L.lsam_mlp_block_proj_image_bytes.restype = C.c_size_t
        L.lsam_mlp_block_proj_image_bytes.argtypes = [i]
        L.lsam_mlp_block_proj_pack.argtypes = [vp, vp, i, vp, vp, vp]
        L.lsam_mlp_block_fwd_proj.argtypes = [vp, vp, i, vp, vp, vp, vp, vp, vp, vp, vp, i, vp, vp, i, i, f, vp]
        L.lsam_mlp_block_fwd.argtypes = [vp, vp, vp, vp, vp, vp, i, i, f, vp]
        L.lsam_mlp_block_fwd_next_norm.argtypes = [vp, vp, vp, vp, vp, vp, vp, i, vp, vp, i, i, f, vp]
        L.lsam_prof_arm.argtypes = [i]
        L.lsam_prof_read.argtypes = [i, vp, vp, vp, vp]
        L.lsam_prof_bracket_cost.argtypes = [vp, i, vp]
        _lib = L
    return _lib


def check(rc: int) -> None:
    if rc != 0:
        raise LsamError(f"liblsam error {rc}: {lib().lsam_last_error().decode()}")


def ptr(t) -> int:
    return 0 if t is None else t.data_ptr()


def stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def dtype_code(dt: torch.dtype) -> int:
    if dt == torch.float32:
        return F32
    if dt == torch.bfloat16:
        return BF16
    raise LsamError(f"unsupported compute dtype {dt}; use torch.float32 or torch.bfloat16")


EXPORTED = [
    "lsam_last_error", "lsam_version", "lsam_prof_arm", "lsam_prof_read", "lsam_prof_bracket_cost", "lsam_linear", "lsam_pack_matrix", "lsam_layernorm_fwd",
    "lsam_layernorm_bwd", "lsam_attention_fwd", "lsam_attention_bwd", "lsam_maxpool2x2", "lsam_spatial_mean", "lsam_cast",
    "lsam_im2col_patch", "lsam_outer_reduce_ws", "lsam_outer_reduce", "lsam_hiera_create",
    "lsam_hiera_destroy", "lsam_hiera_num_blocks", "lsam_hiera_block_info", "lsam_hiera_stage_info",
    "lsam_hiera_arena_bytes", "lsam_hiera_workspace_bytes", "lsam_hiera_pack", "lsam_hiera_pack_lora",
    "lsam_hiera_forward", "lsam_hiera_backward", "lsam_hiera_backward_block",
    "lsam_window_block_image_bytes", "lsam_window_block_bias_floats", "lsam_window_block_pack",
    "lsam_window_block_fwd", "lsam_mlp_block_image_bytes", "lsam_mlp_block_bias_floats", "lsam_mlp_block_pack",
    "lsam_mlp_block_fwd", "lsam_tail_block_image_bytes", "lsam_tail_block_vec_floats", "lsam_tail_block_pack",
    "lsam_tail_block_fwd", "lsam_patch_embed_image_bytes", "lsam_patch_embed_pack", "lsam_patch_embed_fwd", "lsam_linear_maxpool2x2",
    "lsam_lateral_conv_image_bytes", "lsam_lateral_conv_pack", "lsam_lateral_conv_fwd", "lsam_mlp_block_fwd_next_norm", "lsam_attention_fwd_pooled_q",
    "lsam_add_rows", "lsam_mha_fwd", "lsam_pe_encode", "lsam_upscale2x2_fwd", "lsam_mask_gate", "lsam_bilinear_upsample",
    "lsam_attention_bwd_qpool", "lsam_neck_conv_image_bytes", "lsam_neck_conv_pack", "lsam_neck_level_fwd",
    "lsam_neck_topdown_bwd", "lsam_act_bwd", "lsam_mha_bwd", "lsam_upscale2x2_bwd", "lsam_mask_gate_bwd", "lsam_bilinear_upsample_bwd",
    "lsam_batched_outer_ws_floats", "lsam_batched_outer", "lsam_mask_loss_ws_floats", "lsam_mask_loss_sums", "lsam_mask_loss_bwd", "lsam_kv_window_attention_image_bytes", "lsam_kv_window_attention_pack", "lsam_kv_window_attention_fwd",
    "lsam_neck_conv_split_image_bytes", "lsam_neck_conv_pack_split", "lsam_neck_level_fwd_split",
    "lsam_window_block_attention_fwd", "lsam_mlp_block_proj_image_bytes", "lsam_mlp_block_proj_pack", "lsam_mlp_block_fwd_proj",
    "lsam_qkv_window4_attention_image_bytes", "lsam_qkv_window4_attention_pack", "lsam_qkv_window4_attention_fwd",
    "lsam_stream_linear_image_bytes", "lsam_stream_linear_pack", "lsam_stream_linear_fwd",
    "lsam_outer_reduce_static_max_ws", "lsam_outer_reduce_static_init", "lsam_outer_reduce_static",
]
