"""ctypes binding of libcgan_hip.so (the C ABI declared in include/climategan_hip.h).

The product path has NO fallback: if the shared library is missing or a call fails, a RuntimeError is
raised.  Build it with ``python -c "import __graft_entry__ as g; g.build()"`` or
``make -C climategan_amd/csrc``.
"""
import ctypes as C
import os
from pathlib import Path

_HERE = Path(__file__).resolve().parent
# CGAN_LIB: another build of the same library (same-box A/B measurements of a kernel change); default: the in-tree build
LIB_PATH = Path(os.environ["CGAN_LIB"]).resolve() if os.environ.get("CGAN_LIB") else _HERE / "libcgan_hip.so"

CGAN_F16, CGAN_BF16, CGAN_F32 = 0, 1, 2
ACT_NONE, ACT_RELU, ACT_LRELU, ACT_TANH, ACT_SIGMOID = 0, 1, 2, 3, 4
PAD_ZERO, PAD_REFLECT = 0, 1
ABI_VERSION = 1


class ConvDesc(C.Structure):
    _fields_ = [
        ("dtype", C.c_int32), ("n", C.c_int32), ("h_in", C.c_int32), ("w_in", C.c_int32),
        ("c_in", C.c_int32), ("c_out", C.c_int32),
        ("kh", C.c_int32), ("kw", C.c_int32), ("stride", C.c_int32), ("pad", C.c_int32), ("dilation", C.c_int32),
        ("pad_mode", C.c_int32), ("h_out", C.c_int32), ("w_out", C.c_int32), ("in_upsample", C.c_int32),
        ("act", C.c_int32), ("act_slope", C.c_float), ("has_bias", C.c_int32), ("has_residual", C.c_int32),
        ("residual_upsample", C.c_int32),
    ]


class NormStatsDesc(C.Structure):
    _fields_ = [("dtype", C.c_int32), ("n", C.c_int32), ("hw", C.c_int32), ("c", C.c_int32), ("eps", C.c_float)]


class SpadeDesc(C.Structure):
    _fields_ = [
        ("dtype", C.c_int32), ("n", C.c_int32), ("h", C.c_int32), ("w", C.c_int32), ("c", C.c_int32),
        ("x_upsample", C.c_int32), ("cond_h", C.c_int32), ("cond_w", C.c_int32), ("cond_c", C.c_int32),
        ("hidden", C.c_int32), ("ksize", C.c_int32), ("act", C.c_int32), ("act_slope", C.c_float),
    ]


class SnItem(C.Structure):
    _fields_ = [("w_bar", C.c_void_p), ("u", C.c_void_p), ("v", C.c_void_p), ("sigma", C.c_void_p),
                ("workspace", C.c_void_p), ("rows", C.c_int32), ("cols", C.c_int32)]


class PackItem(C.Structure):
    _fields_ = [("w_oihw", C.c_void_p), ("bias", C.c_void_p), ("sigma", C.c_void_p), ("packed", C.c_void_p),
                ("bias_out", C.c_void_p), ("c_out", C.c_int32), ("c_in", C.c_int32), ("kh", C.c_int32),
                ("kw", C.c_int32), ("transposed", C.c_int32)]


class AdamItem(C.Structure):
    _fields_ = [("p", C.c_void_p), ("g", C.c_void_p), ("m", C.c_void_p), ("v", C.c_void_p), ("copy", C.c_void_p),
                ("numel", C.c_int64)]


_P = C.c_void_p
_SIGNATURES = {
    # name: (restype, argtypes)
    "cgan_version": (C.c_int, []),
    "cgan_last_error": (C.c_char_p, []),
    "cgan_conv2d_packed_weight_bytes": (C.c_size_t, [C.POINTER(ConvDesc)]),
    "cgan_conv2d_pack_weight": (C.c_int, [_P, _P, _P, _P, _P, C.POINTER(ConvDesc), _P]),
    "cgan_conv2d_nhwc_fwd": (C.c_int, [_P, _P, _P, _P, _P, C.POINTER(ConvDesc), _P]),
    "cgan_conv2d_pack_weight_batched": (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, _P]),
    "cgan_conv2d_dgrad_packed_weight_bytes": (C.c_size_t, [C.POINTER(ConvDesc)]),
    "cgan_conv2d_pack_weight_dgrad": (C.c_int, [_P, _P, _P, C.POINTER(ConvDesc), _P]),
    "cgan_conv2d_nhwc_bwd_data": (C.c_int, [_P, _P, _P, C.POINTER(ConvDesc), _P]),
    "cgan_conv2d_nhwc_fwd_pair": (C.c_int, [_P, _P, _P, _P, _P, C.POINTER(ConvDesc), _P]),
    "cgan_pair_expand_weight": (C.c_int, [_P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P]),
    "cgan_pair_instnorm_stats": (C.c_int, [_P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_float, _P]),
    "cgan_pair_spade_apply": (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                        C.c_int32, C.c_float, _P]),
    "cgan_pair_make_m_cond_workspace_bytes": (C.c_size_t, [C.c_int32]),
    "cgan_pair_make_m_cond": (C.c_int, [_P, _P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                        _P, C.c_size_t, _P]),
    "cgan_pair_from_nchw": (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P]),
    "cgan_pair_to_nchw": (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P]),
    "cgan_pair_to_nhwc": (C.c_int, [_P, _P, C.c_int32, C.c_int64, C.c_int32, _P]),
    "cgan_pair_maxpool3x3s2": (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P]),
    "cgan_pair_resize_bilinear": (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                            C.c_int32, C.c_int32, _P]),
    "cgan_pair_resize_nearest": (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                           _P]),
    "cgan_pair_resize_bicubic": (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P]),
    "cgan_pair_mul": (C.c_int, [_P, _P, _P, C.c_int32, C.c_int64, C.c_int32, _P]),
    "cgan_pair_copy_channels": (C.c_int, [_P, _P, C.c_int32, C.c_int64, C.c_int32, C.c_int32, C.c_int32, _P]),
    "cgan_conv2d_nhwc_bwd_data_add": (C.c_int, [_P, _P, _P, _P, C.POINTER(ConvDesc), _P]),
    "cgan_conv2d_nhwc_bwd_data_relu": (C.c_int, [_P, _P, _P, _P, C.POINTER(ConvDesc), _P]),
    "cgan_conv2d_nhwc_bwd_data_add_relu": (C.c_int, [_P, _P, _P, _P, _P, C.POINTER(ConvDesc), _P]),
    "cgan_conv2d_kernel_kind": (C.c_int, [C.POINTER(ConvDesc), C.c_int32]),
    "cgan_conv2d_kernel_kind_on": (C.c_int, [C.POINTER(ConvDesc), C.c_int32, _P]),
    "cgan_conv2d_bind_workspace": (C.c_int, [_P, _P, C.c_size_t]),
    "cgan_rccl_load": (C.c_int, [C.c_char_p]),
    "cgan_rccl_loaded": (C.c_int, []),
    "cgan_comm_unique_id": (C.c_int, [_P]),
    "cgan_comm_init_rank": (C.c_int, [C.POINTER(C.c_void_p), C.c_int32, _P, C.c_int32]),
    "cgan_comm_destroy": (C.c_int, [_P]),
    "cgan_allreduce_bucket": (C.c_int, [_P, C.c_int64, C.c_int32, _P, _P]),
    "cgan_seg_counts": (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_int32, _P, _P, _P]),
    "cgan_resize_crop_geometry": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                            C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "cgan_resize_u8": (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, C.c_int32, _P, C.c_int32, _P,
                                 _P, C.c_size_t, _P]),
    "cgan_resize_crop_u8_workspace_bytes": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32]),
    "cgan_resize_crop_u8": (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, C.c_int32, _P, C.c_int32, _P,
                                      _P, C.c_size_t, _P]),
    "cgan_conv2d_bwd_weight_workspace_bytes": (C.c_size_t, [C.POINTER(ConvDesc)]),
    "cgan_conv2d_nhwc_bwd_weight": (C.c_int, [_P, _P, _P, _P, C.POINTER(ConvDesc), _P, C.c_size_t, _P]),
    "cgan_instnorm_stats_workspace_bytes": (C.c_size_t, [C.POINTER(NormStatsDesc)]),
    "cgan_instnorm_stats": (C.c_int, [_P, _P, _P, C.POINTER(NormStatsDesc), _P, C.c_size_t, _P]),
    "cgan_norm_act_apply": (C.c_int, [_P, _P, _P, _P, C.POINTER(NormStatsDesc), C.c_int32, C.c_float, _P]),
    "cgan_make_m_cond_bwd_nhwc": (C.c_int, [_P, _P, _P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                            C.c_int32, _P]),
    "cgan_resize_nearest_bwd_nhwc": (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                               C.c_int32, C.c_int32, C.c_int32, _P]),
    "cgan_norm_add_act_apply": (C.c_int, [_P, _P, _P, _P, _P, C.POINTER(NormStatsDesc), C.c_int32, C.c_float, _P]),
    "cgan_spade_packed_weight_bytes": (C.c_size_t, [C.POINTER(SpadeDesc)]),
    "cgan_spade_pack_weights": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, C.POINTER(SpadeDesc), _P]),
    "cgan_spade_fused_fwd": (C.c_int, [_P, _P, _P, _P, _P, _P, C.POINTER(SpadeDesc), _P]),
    "cgan_spade_fused_fwd_train": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, C.POINTER(SpadeDesc), _P]),
    "cgan_spectral_norm_workspace_bytes": (C.c_size_t, [C.c_int32, C.c_int32]),
    "cgan_spectral_norm_power_iter": (C.c_int, [_P, _P, _P, _P, C.c_int32, C.c_int32, _P, C.c_size_t, _P]),
    "cgan_spectral_norm_power_iter_batched": (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, _P]),
    "cgan_extra_adam_multi_tensor": (C.c_int, [_P, C.c_int32, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_double,
                                               C.c_double, C.c_double, C.c_double, C.c_double, _P]),
    "cgan_nchw_to_nhwc": (C.c_int, [_P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P]),
    "cgan_nhwc_to_nchw": (C.c_int, [_P, _P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P]),
    "cgan_resize_nearest_nhwc": (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                           C.c_int32, C.c_int32, C.c_int32, _P]),
    "cgan_avgpool3x3s2_nhwc": (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P]),
    "cgan_maxpool3x3s2_nhwc": (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P]),
    "cgan_resize_bilinear_nhwc": (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                            C.c_int32, C.c_int32, _P]),
    "cgan_resize_bicubic_nhwc": (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                           C.c_int32, _P]),
    "cgan_sumpool2x2_nhwc": (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P]),
    "cgan_reflect_pad_bwd_nhwc": (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P]),
    "cgan_copy_channels_nhwc": (C.c_int, [_P, _P, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P]),
    "cgan_eltwise_nhwc": (C.c_int, [_P, _P, _P, C.c_int32, C.c_int32, C.c_int64, _P]),
    "cgan_fold_bn": (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_float, _P, _P, C.c_int32, C.c_int64, _P]),
    "cgan_act_bwd": (C.c_int, [_P, _P, _P, C.c_int32, C.c_int32, C.c_float, C.c_int64, _P]),
    "cgan_instnorm_act_bwd_workspace_bytes": (C.c_size_t, [C.POINTER(NormStatsDesc)]),
    "cgan_instnorm_act_bwd": (C.c_int, [_P, _P, _P, _P, C.POINTER(NormStatsDesc), C.c_int32, C.c_float, _P, C.c_size_t,
                                        _P]),
    "cgan_spade_bwd_prepare": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, C.POINTER(SpadeDesc), _P]),
    "cgan_spade_hidden_bwd_workspace_bytes": (C.c_size_t, [C.POINTER(SpadeDesc)]),
    "cgan_spade_hidden_bwd": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, C.c_size_t, C.POINTER(SpadeDesc), _P]),
    "cgan_batchnorm_train_stats": (C.c_int, [_P, _P, _P, C.c_float, _P, _P, _P, _P, _P, _P, _P, C.POINTER(NormStatsDesc), _P,
                                             C.c_size_t, _P]),
    "cgan_batchnorm_train_stats_from_partials": (C.c_int, [_P, C.c_int32, _P, _P, C.c_float, _P, _P, _P, _P, _P, _P, _P,
                                                           C.POINTER(NormStatsDesc), _P]),
    "cgan_conv2d_stats_chunk_pixels": (C.c_int32, [C.POINTER(ConvDesc)]),
    "cgan_conv2d_nhwc_fwd_stats": (C.c_int, [_P, _P, _P, _P, _P, C.c_size_t, C.POINTER(ConvDesc), _P]),
    "cgan_bn_train_prepare": (C.c_int, [_P, _P, _P, _P, C.c_float, C.c_float, C.c_int64, _P, _P, _P, _P, _P, C.c_int32, _P]),
    "cgan_batchnorm_act_bwd_workspace_bytes": (C.c_size_t, [C.c_int32]),
    "cgan_batchnorm_act_bwd": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, C.c_int32, C.c_int64, C.c_int32,
                                         C.c_int32, C.c_float, _P, C.c_size_t, _P]),
    "cgan_batchnorm_act_bwd_grouped": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, C.c_int32, C.c_int64,
                                                 C.c_int32, C.c_int32, C.c_int32, C.c_float, _P, C.c_size_t, _P]),
    "cgan_bce_logits_nhwc": (C.c_int, [_P, C.c_int32, C.c_int64, C.c_int32, C.c_float, C.c_float, _P, _P, _P]),
    "cgan_hinge_nhwc": (C.c_int, [_P, C.c_int32, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_float, _P, _P, _P]),
    "cgan_l1_nhwc": (C.c_int, [_P, _P, C.c_int32, C.c_int64, C.c_float, _P, _P, _P]),
    "cgan_mse_const_nhwc": (C.c_int, [_P, C.c_int32, C.c_int64, C.c_int32, C.c_float, C.c_float, _P, _P, _P]),
    "cgan_painter_aux_losses": (C.c_int, [_P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_float,
                                          C.c_float, C.c_float, _P, _P, _P]),
    "cgan_resize_bicubic_bwd_nhwc": (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                               C.c_int32, _P]),
    "cgan_spectral_norm_bwd": (C.c_int, [_P, _P, _P, _P, _P, C.c_int32, C.c_int32, _P, _P]),
    "cgan_painter_heads_fwd": (C.c_int, [_P, _P, _P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P]),
    "cgan_painter_heads_bwd": (C.c_int, [_P, _P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P]),
    "cgan_avgpool3x3s2_bwd_nhwc": (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P]),
    "cgan_maxpool2x2_nhwc": (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P]),
    "cgan_maxpool2x2_bwd_nhwc": (C.c_int, [_P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P]),
    "cgan_maxpool2x2_relu_bwd_nhwc": (C.c_int, [_P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P]),
    "cgan_softmax_nhwc": (C.c_int, [_P, _P, C.c_int32, C.c_int64, C.c_int32, _P]),
    "cgan_softmax_bwd_nhwc": (C.c_int, [_P, _P, _P, C.c_int32, C.c_int64, C.c_int32, _P]),
    "cgan_sigmoid_pair_nhwc": (C.c_int, [_P, _P, C.c_int32, C.c_int64, _P]),
    "cgan_sigmoid_pair_bwd_nhwc": (C.c_int, [_P, _P, _P, C.c_int32, C.c_int64, _P]),
    "cgan_softmax_ce_nhwc": (C.c_int, [_P, _P, C.c_int32, C.c_int64, C.c_int32, C.c_float, _P, _P, _P]),
    "cgan_tv_nhwc": (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_float, _P, _P, _P]),
    "cgan_advent_entropy_pair_nhwc": (C.c_int, [_P, _P, _P, C.c_int32, C.c_int64, C.c_int32, C.c_int32, _P]),
    "cgan_entropy_pair_from_nchw": (C.c_int, [_P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P]),
    "cgan_entropy_pair_from_nchw_bwd": (C.c_int, [_P, _P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P]),
    "cgan_advent_entropy_pair_bwd_nhwc": (C.c_int, [_P, _P, _P, _P, C.c_int32, C.c_int64, C.c_int32, C.c_int32, _P]),
    "cgan_entropy_map_nhwc": (C.c_int, [_P, _P, _P, C.c_int32, C.c_int64, C.c_int32, _P]),
    "cgan_entropy_map_bwd_nhwc": (C.c_int, [_P, _P, _P, _P, C.c_int32, C.c_int64, C.c_int32, _P]),
    "cgan_minent_nhwc": (C.c_int, [_P, C.c_int32, C.c_int64, C.c_int32, C.c_int32, C.c_float, C.c_float, _P, _P, _P, _P]),
    "cgan_bce_logits_map_nhwc": (C.c_int, [_P, _P, C.c_int32, C.c_int64, C.c_float, _P, _P, _P]),
    "cgan_ground_intersection_nhwc": (C.c_int, [_P, _P, C.c_int32, C.c_int64, C.c_float, _P, _P]),
    "cgan_affine_sum_nhwc": (C.c_int, [_P, C.c_int32, C.c_int64, C.c_int32, C.c_float, C.c_float, _P, _P, _P]),
    "cgan_resize_bilinear_bwd_workspace_bytes": (C.c_size_t, [C.c_int32] * 4),
    "cgan_resize_bilinear_bwd_nhwc": (C.c_int, [_P, _P] + [C.c_int32] * 8 + [_P, C.c_size_t, _P]),
    "cgan_maxpool3x3s2_bwd_nhwc": (C.c_int, [_P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P]),
    "cgan_add_act_nhwc": (C.c_int, [_P, _P, _P, C.c_int32, C.c_int32, C.c_float, C.c_int64, _P]),
    "cgan_slice_channels_nhwc": (C.c_int, [_P, _P, C.c_int64, C.c_int32, C.c_int32, C.c_int32, _P]),
    "cgan_sigm_loss_workspace_bytes": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32]),
    "cgan_sigm_loss_nhwc": (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_int32, C.c_float,
                                      _P, _P, _P, C.c_size_t, _P]),
    "cgan_normalize_u8_workspace_bytes": (C.c_size_t, [C.c_int32]),
    "cgan_normalize_u8_nhwc": (C.c_int, [_P, C.c_int32, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, C.c_size_t,
                                         _P]),
    "cgan_binarize": (C.c_int, [_P, C.c_int32, _P, _P, C.c_float, C.c_int64, _P]),
    "cgan_smog_workspace_bytes": (C.c_size_t, [C.c_int32]),
    "cgan_smog_nchw": (C.c_int, [_P, _P, C.c_int32, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                 C.c_float, C.c_float, C.c_float, C.POINTER(C.c_float), _P, C.c_size_t, _P]),
    "cgan_cloudy_cond_workspace_bytes": (C.c_size_t, [C.c_int32, C.c_int32]),
    "cgan_cloudy_cond_nhwc": (C.c_int, [_P, _P, _P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                        C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float, _P,
                                        C.c_size_t, _P]),
    "cgan_bn_eval_stats": (C.c_int, [_P, _P, _P, _P, C.c_float, _P, _P, C.c_int32, C.c_int32, _P]),
    "cgan_make_m_cond_workspace_bytes": (C.c_size_t, [C.c_int32]),
    "cgan_make_m_cond_nhwc": (C.c_int, [_P, _P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                        C.c_int32, C.c_int32, _P, C.c_size_t, _P]),
    "cgan_wildfire_workspace_bytes": (C.c_size_t, [C.c_int32] * 6),
    "cgan_wildfire_nchw": (C.c_int, [_P, _P, C.c_int32, _P] + [C.c_int32] * 8 + [C.c_float, C.c_float, C.c_int32,
                                                                               C.c_float, _P, C.c_size_t, _P]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)

_lib = None


def _open(path):
    if not path.exists():
        raise RuntimeError(
            "climategan_amd: %s not found -- the HIP extension is not built (run __graft_entry__.build() or "
            "`make -C climategan_amd/csrc both`).  There is no CPU/PyTorch fallback on the product path." % path)
    try:
        # import torch first so that libamdhip64.so.7 resolves to the runtime torch already loaded
        import torch  # noqa: F401
    except Exception:
        pass
    lib = C.CDLL(str(path), mode=getattr(os, "RTLD_NOW", 2))
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError -> missing symbol: let it propagate loudly
        fn.restype = res
        fn.argtypes = args
    v = lib.cgan_version()
    if v != ABI_VERSION:
        raise RuntimeError("climategan_amd: %s ABI version %d != expected %d" % (path.name, v, ABI_VERSION))
    return lib


def load():
    """Load (once) and return the ctypes handle of the library in use (the product library unless ``load_dev`` switched).
    Raises RuntimeError if the library is not built."""
    global _lib, _product
    if _lib is not None:
        return _lib
    _product = _open(LIB_PATH)
    _lib = _product
    if os.environ.get("CGAN_DEV_LIB") == "1":        # a whole run on the development build (same-box A/B with knobs)
        load_dev()
    return _lib


_product = None
_dev = None
# CGAN_LIB_DEV: another build of the development library (same-box A/B of a kernel change with tools/bench_*.py)
DEV_LIB_PATH = Path(os.environ["CGAN_LIB_DEV"]).resolve() if os.environ.get("CGAN_LIB_DEV") else _HERE / "libcgan_hip_dev.so"


def load_dev():
    """Switch this process to the DEVELOPMENT build (libcgan_hip_dev.so = the same sources with -DCGAN_DEV: the
    ``cgan_debug_set_*`` knobs for kernel selection, ablation bits and in-kernel timestamps) and return its handle.  The
    product library exports none of them.  tools/ and the tests that run every kernel variant on the same cases use this;
    ``use_product()`` switches back.  Knob from the environment: CGAN_DEBUG_GEMM_WS=<n>."""
    global _lib, _dev
    if _dev is None:
        _dev = _open(DEV_LIB_PATH)
        ws = os.environ.get("CGAN_DEBUG_GEMM_WS")
        if ws:
            _dev.cgan_debug_set_gemm_ws(C.c_int(int(ws)))
        cg = os.environ.get("CGAN_DEBUG_WGRAD_COOP_G")          # 2: never the 16-wave weight-gradient tile, 4: wherever it applies
        if cg:
            _dev.cgan_debug_set_wgrad_coop_g(C.c_int(int(cg)))
        bj = os.environ.get("CGAN_DEBUG_BN_JITTER")              # "<ppm>,<seed>": cgan_debug_set_bn_jitter (R5 DESIGN 4.13)
        if bj:
            _dev.cgan_debug_set_bn_jitter(C.c_int(int(bj.split(",")[0])), C.c_int(int(bj.split(",")[1]) if "," in bj else 0))
        ck = os.environ.get("CGAN_DEBUG_CONV_KERNEL")            # cgan_debug_set_conv_kernel (4: without round 5's GEMM launches)
        if ck:
            _dev.cgan_debug_set_conv_kernel(C.c_int(int(ck)))
    _lib = _dev
    return _dev


def use_product():
    """Back to the product library (after ``load_dev``)."""
    global _lib
    load()
    _lib = _product
    return _lib


# Development aid (bench.py --call-log, tools/step_hbm_budget.py): when CALL_LOG is a list, every checked C-ABI call appends
# (entry point, bytes of the device operands handed to it through ops._ptr since the previous call) -- the bytes the call has to
# touch at least once; calls that take item tables add their operands with log_bytes().
CALL_LOG = None
_pending_bytes = 0


def log_bytes(n):
    global _pending_bytes
    if CALL_LOG is not None:
        _pending_bytes += int(n)


def check(rc, what):
    global _pending_bytes
    if CALL_LOG is not None:
        CALL_LOG.append((what, _pending_bytes))
        _pending_bytes = 0
    if rc != 0:
        msg = load().cgan_last_error()
        raise RuntimeError("%s failed (status %d): %s" % (what, rc, msg.decode() if msg else "?"))
