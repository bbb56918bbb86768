"""ctypes binding of libsgamd.so (the C ABI declared in include/sgamd.h).

The product path has NO fallback: if the HIP library is missing or a kernel is asked to run on a non-GPU
tensor, we raise. (The CPU oracle under /oracle is test infrastructure and is never imported from here.)
"""
import ctypes as C
import os
import subprocess
import sys

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsgamd.so")
if os.environ.get("SG_LIBSGAMD"):      # same-box A/B of two builds of the library (tools/sessions/*.sh); never set by the package, the tests or bench.py
    LIB_PATH = os.path.abspath(os.environ["SG_LIBSGAMD"])
CSRC = os.path.join(_HERE, "csrc")

F32, BF16, F64 = 0, 1, 2
PIX_RELU, PIX_UPSAMPLE, PIX_QUAD, PIX_TRANSPOSED = 1, 2, 4, 8
Q_POOL, Q_UP = 0, 1
EPI_OUT_F32, EPI_ATOMIC, EPI_POOL, EPI_RELU, EPI_RES_F32 = 1, 2, 4, 8, 16

_vp, _i, _f, _ll = C.c_void_p, C.c_int, C.c_float, C.c_longlong


class ConvFwdDesc(C.Structure):
    _fields_ = [("dtype", _i), ("N", _i), ("Hs", _i), ("Ws", _i), ("C", _i), ("ldx", _i), ("Ho", _i), ("Wo", _i), ("Cout", _i),
                ("R", _i), ("S", _i), ("stride", _i), ("pad_h", _i), ("pad_w", _i), ("pix_flags", _i), ("epi_flags", _i),
                ("alpha", _f), ("beta", _f), ("x", _vp), ("w", _vp), ("bias", _vp), ("res", _vp), ("mask", _vp), ("out", _vp),
                ("alpha_ptr", _vp), ("ldo", _i), ("ldr", _i), ("ldm", _i)]


class ConvSkipDesc(C.Structure):
    _fields_ = [("main", ConvFwdDesc), ("x2", _vp), ("w2", _vp), ("bias2", _vp), ("C2", _i), ("ldx2", _i), ("x2_up", _i), ("stats", _vp)]


class ConvWgradDesc(C.Structure):
    _fields_ = [("dtype", _i), ("N", _i), ("xHs", _i), ("xWs", _i), ("C", _i), ("ldx", _i), ("x_flags", _i), ("gHs", _i), ("gWs", _i),
                ("Cout", _i), ("ldg", _i), ("g_flags", _i), ("Ho", _i), ("Wo", _i), ("R", _i), ("S", _i), ("stride", _i),
                ("pad_h", _i), ("pad_w", _i), ("alpha", _f), ("x", _vp), ("dy", _vp), ("dw", _vp), ("alpha_ptr", _vp),
                ("splits", _i), ("no_tr", _i), ("work", _vp), ("work_floats", _ll), ("dbias", _vp)]


class ConvQDesc(C.Structure):
    _fields_ = [("dtype", _i), ("form", _i), ("N", _i), ("Hl", _i), ("Wl", _i), ("C", _i), ("ldx", _i), ("Cout", _i),
                ("pix_flags", _i), ("epi_flags", _i), ("alpha", _f), ("beta", _f), ("x", _vp), ("wq", _vp), ("bias", _vp), ("res", _vp),
                ("mask", _vp), ("out", _vp), ("alpha_ptr", _vp), ("ldo", _i), ("ldr", _i), ("ldm", _i),
                ("x2", _vp), ("w2q", _vp), ("bias2", _vp), ("C2", _i), ("ldx2", _i), ("stats", _vp), ("x2_norelu", _i)]


class ConvQWgradDesc(C.Structure):
    _fields_ = [("dtype", _i), ("form", _i), ("N", _i), ("Hl", _i), ("Wl", _i), ("C", _i), ("ldx", _i), ("x_flags", _i), ("Cout", _i),
                ("ldg", _i), ("alpha", _f), ("alpha_ptr", _vp), ("x", _vp), ("dy", _vp), ("dw", _vp), ("dbias", _vp), ("work", _vp),
                ("work_floats", _ll), ("splits", _i)]


class QuadItem(C.Structure):
    _fields_ = [("src", _vp), ("dst", _vp), ("M", _i), ("Cs", _i), ("mode", _i), ("pad_", _i)]


class LinearItem(C.Structure):
    _fields_ = [("w", _vp), ("y", _vp), ("bias", _vp), ("out", _vp), ("rows", _i), ("K", _i), ("ldy", _i), ("ldo", _i)]


class GemmDesc(C.Structure):
    _fields_ = [("dtype", _i), ("p_form", _i), ("q_form", _i), ("I", _i), ("J", _i), ("K", _i), ("batch", _i),
                ("p", _vp), ("p_bstride", _ll), ("ldp", _i), ("q", _vp), ("q_bstride", _ll), ("ldq", _i),
                ("out", _vp), ("out_bstride", _ll), ("ldo", _i), ("bias", _vp), ("res", _vp), ("res_bstride", _ll), ("ldr", _i),
                ("beta", _f), ("alpha", _f), ("alpha_ptr", _vp), ("epi_flags", _i), ("splits", _i), ("no_tr", _i)]


class SnLayer(C.Structure):
    _fields_ = [("w", _vp), ("u", _vp), ("v", _vp), ("sigma", _vp), ("u_snap", _vp), ("v_snap", _vp), ("w_fwd", _vp), ("w_dgrad", _vp),
                ("w_f32", _vp), ("rows", _i), ("cols", _i), ("Cin", _i), ("RS", _i), ("do_power_iter", _i), ("apply_sn", _i),
                ("rows_pad", _i), ("work_off", _ll), ("trans", _i), ("dgrad_noflip", _i), ("Cin_pad", _i)]


class SnBwdLayer(C.Structure):
    _fields_ = [("dwt", _vp), ("w", _vp), ("u", _vp), ("v", _vp), ("sigma", _vp), ("dw", _vp), ("rows", _i), ("cols", _i),
                ("Cin", _i), ("RS", _i), ("natural", _i), ("apply_sn", _i), ("trans", _i), ("Cin_pad", _i)]


class AugDesc(C.Structure):
    _fields_ = [("N", _i), ("C", _i), ("H", _i), ("W", _i), ("ops", _i), ("cut_h", _i), ("cut_w", _i), ("max_t", _i), ("color", _vp), ("geom", _vp)]


AUG_BRIGHTNESS, AUG_SATURATION, AUG_CONTRAST, AUG_FLIP, AUG_TRANSLATE, AUG_TRANSLATE_REFLECT, AUG_CUTOUT = 1, 2, 4, 8, 16, 32, 64

_lib = None


def build(force=False, verbose=False):
    """Compile every HIP source for gfx950 into libsgamd.so (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-C", CSRC, "-j", str(max(2, os.cpu_count() or 2))]
    if force:
        subprocess.run(["make", "-C", CSRC, "clean"], check=True, stdout=subprocess.DEVNULL)
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or res.returncode != 0:
        sys.stderr.write(res.stdout)
    if res.returncode != 0:
        raise RuntimeError("building libsgamd.so failed")
    return LIB_PATH


_PROTOS = {
    "sg_prof_enable": [_i],
    "sg_prof_collect": [C.POINTER(C.c_double), _i],
    "sg_conv2d_fwd": [C.POINTER(ConvFwdDesc), _vp],
    "sg_conv2d_fwd_skip": [C.POINTER(ConvSkipDesc), _vp],
    "sg_conv2d_fwd_skip_ok": [C.POINTER(ConvSkipDesc)],
    "sg_conv_rs_launches": [],
    "sg_conv2d_q": [C.POINTER(ConvQDesc), _vp],
    "sg_conv2d_q_ok": [C.POINTER(ConvQDesc)],
    "sg_conv2d_q_stat_rows": [C.POINTER(ConvQDesc)],
    "sg_conv2d_fwd_skip_stat_rows": [C.POINTER(ConvSkipDesc)],
    "sg_bn_stats_from_tiles": [_vp, _i, _i, _vp, _vp],
    "sg_quad_pack": [_i, _i, _vp, _vp, _i, _i, _vp],
    "sg_quad_pack_batch": [_i, _vp, C.POINTER(QuadItem), _i, _vp],
    "sg_linear_group": [_vp, C.POINTER(LinearItem), _i, _i, _vp],
    "sg_conv2d_q_wgrad_plan": [C.POINTER(ConvQWgradDesc), C.POINTER(_i), C.POINTER(_ll)],
    "sg_conv2d_q_wgrad": [C.POINTER(ConvQWgradDesc), _vp],
    "sg_prof_collect_ex": [C.POINTER(C.c_double), _i],
    "sg_prof_collect_tags": [C.POINTER(C.c_double), _i],
    "sg_conv2d_wgrad": [C.POINTER(ConvWgradDesc), _vp],
    "sg_conv2d_wgrad_plan": [C.POINTER(ConvWgradDesc), C.POINTER(_i), C.POINTER(_ll)],
    "sg_conv2d_wgrad_fuses_bias": [C.POINTER(ConvWgradDesc)],
    "sg_gemm": [C.POINTER(GemmDesc), _vp],
    "sg_nchw_to_nhwc": [_i, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "sg_nhwc_to_nchw": [_i, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "sg_nchw_grad_to_nhwc": [_i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "sg_avgpool2_fwd": [_i, _vp, _vp, _i, _i, _i, _i, _vp],
    "sg_avgpool2_bwd": [_i, _vp, _vp, _i, _i, _i, _i, _vp],
    "sg_maxpool2_fwd": [_i, _vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _vp],
    "sg_maxpool2_bwd": [_i, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "sg_softmax_rows": [_i, _vp, _vp, _ll, _i, _vp],
    "sg_softmax_rows_bwd": [_i, _vp, _vp, _vp, _ll, _i, _vp],
    "sg_axpby": [_i, _vp, _vp, _ll, _f, _f, _vp],
    "sg_convert": [_i, _i, _vp, _vp, _ll, _vp],
    "sg_add_relu": [_i, _vp, _vp, _vp, _ll, _vp],
    "sg_relu_mask": [_i, _vp, _vp, _vp, _ll, _vp],
    "sg_dot": [_i, _vp, _vp, _ll, _vp, _f, _vp, _vp],
    "sg_colsum": [_i, _vp, _i, _vp, _i, _ll, _i, _vp, _f, _vp],
    "sg_bn_partial_stats": [_i, _vp, _i, _ll, _i, _vp, _vp],
    "sg_bn_finalize": [_vp, C.c_double, _i, _f, _f, _vp, _vp, _vp, _vp, _vp],
    "sg_comm_unique_id": [_vp],
    "sg_comm_init_rank": [_vp, _i, _i, C.POINTER(_vp)],
    "sg_comm_size": [_vp, C.POINTER(_i)],
    "sg_comm_destroy": [_vp],
    "sg_allreduce_flat": [_vp, _vp, _ll, _i, _vp],
    "sg_bn_stats_sync": [_i, _vp, _i, _ll, _i, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _vp],
    "sg_set_f32_mode": [_i],
    "sg_get_f32_mode": [],
    "sg_reduce_scatter_flat": [_vp, _vp, _ll, _vp],
    "sg_allgather_flat": [_vp, _vp, _ll, _vp],
    "sg_p2p_create": [_i, _i, _ll, C.POINTER(_vp), _vp],
    "sg_p2p_connect": [_vp, _vp],
    "sg_p2p_destroy": [_vp],
    "sg_p2p_timeouts": [_vp, C.POINTER(_i)],
    "sg_p2p_allreduce_f64": [_vp, _vp, _i, _vp],
    "sg_bn_finalize_p2p": [_vp, _vp, C.c_double, _i, _f, _f, _vp, _vp, _vp, _vp, _vp],
    "sg_bn_stats_sync_p2p": [_i, _vp, _i, _ll, _i, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _vp],
    "sg_bn_from_running": [_vp, _vp, _i, _f, _vp, _vp, _vp],
    "sg_bn_apply": [_i, _vp, _vp, _i, _ll, _i, _vp, _vp, _vp, _vp, _i, _i, _vp],
    "sg_bn_bwd_reduce": [_i, _vp, _vp, _i, _ll, _i, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp],
    "sg_bn_bwd_finalize": [_vp, _i, _i, _vp, _i, _vp, _vp, _vp, _vp],
    "sg_bn_bwd_apply": [_i, _vp, _vp, _vp, _i, _ll, _i, _vp, _vp, _vp, _vp, _i, _i, _vp, C.c_double, _i, _vp],
    "sg_bn_bwd_apply_res": [_i, _vp, _vp, _vp, _i, _ll, _i, _vp, _vp, _vp, _vp, _i, _i, _vp, C.c_double, _i, _vp, _vp],
    "sg_bn_bwd2_reduce": [_i, _vp, _vp, _vp, _i, _ll, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp],
    "sg_bn_bwd2_finalize": [_vp, _i, _i, _vp, _vp],
    "sg_bn_bwd2_dgain": [_vp, _vp, C.c_double, _vp, _i, _i, _vp, _vp],
    "sg_bn_bwd2_apply": [_i, _vp, _vp, _vp, _vp, _vp, _i, _ll, _i, _vp, _vp, _vp, _vp, _i, _vp, C.c_double, _i, _vp],
    "sg_attn_fused_ok": [_i, _i, _i, _i, _i],
    "sg_attn_fwd_fused_ok": [_i, _i, _i, _i, _i],
    "sg_attn_fwd_flash_ok": [_i, _i, _i, _i, _i],
    "sg_attn_fwd_fused": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "sg_attn_bwd_fused_ok": [_i, _i, _i, _i, _i],
    "sg_attn_bwd_fused": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "sg_attn_probs_fwd": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "sg_attn_ds_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "sg_slice_up_fwd": [_i, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "sg_slice_up_bwd": [_i, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "sg_copy_channels": [_i, _vp, _i, _vp, _i, _ll, _i, _vp],
    "sg_interp_rows": [_vp, _vp, _vp, _vp, _i, _ll, _vp],
    "sg_gp_fwd": [_i, _vp, _i, _ll, _vp, _vp, _vp],
    "sg_gp_bwd": [_i, _vp, _vp, _vp, _vp, _i, _ll, _vp],
    "sg_masked_sum_hw": [_i, _vp, _vp, _vp, _i, _i, _i, _vp],
    "sg_sn_forward": [_i, _vp, C.POINTER(SnLayer), _i, _f, _vp, _ll, _vp],
    "sg_sn_backward": [_vp, C.POINTER(SnBwdLayer), _i, _vp, _ll, _vp],
    "sg_embedding_fwd": [_vp, _vp, _vp, _i, _i, _i, _vp],
    "sg_embedding_bwd": [_vp, _vp, _vp, _i, _i, _i, _vp],
    "sg_relu_sum_hw_fwd": [_i, _vp, _vp, _i, _i, _i, _vp],
    "sg_relu_sum_hw_bwd": [_i, _vp, _vp, _vp, _i, _i, _i, _vp],
    "sg_pd_head_fwd": [_vp, _vp, _vp, _vp, _vp, _i, _i, _vp],
    "sg_pd_head_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp],
    "sg_loss_d": [_i, _vp, _vp, _i, _vp, _vp, _vp, _vp],
    "sg_loss_g": [_i, _vp, _i, _vp, _vp, _vp],
    "sg_adam_ema": [_vp, _vp, _vp, _vp, _vp, _ll, _f, _f, _f, _f, _f, _i, _f, _f, _vp],
    "sg_ema_lerp": [_vp, _vp, _ll, _f, _vp],
    "sg_quantize_resize_normalize": [_i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp],
    "sg_pil_resize_normalize": [_i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _i, _vp, _vp, _i, _i, _vp],
    "sg_pool2d": [_i, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "sg_global_avgpool": [_i, _vp, _vp, _i, _i, _i, _vp],
    "sg_feat_moments_accumulate": [_vp, _i, _i, _vp, _vp, _vp],
    "sg_topk_hits": [_vp, _i, _i, _vp, _i, _i, _vp, _vp],
    "sg_row_normalize_fwd": [_vp, _vp, _vp, _i, _i, _f, _vp],
    "sg_row_normalize_bwd": [_vp, _vp, _vp, _vp, _i, _i, _vp],
    "sg_row_dot": [_vp, _vp, _vp, _i, _i, _vp],
    "sg_row_scale": [_vp, _vp, _vp, _i, _i, _i, _vp],
    "sg_class_loss": [_i, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp],
    "sg_contrastive_loss": [_i, _vp, _vp, _vp, _i, _f, _f, _vp, _vp, _vp, _vp, _vp],
    "sg_gather_cols": [_vp, _vp, _i, _i, _vp, _vp],
    "sg_scatter_cols": [_vp, _vp, _i, _i, _vp, _vp],
    "sg_chol_lower": [_vp, _i, _vp, _vp],
    "sg_dgemm_tn": [_vp, _vp, _vp, _i, _vp],
    "sg_jacobi_sweep": [_vp, _i, _vp, _vp],
    "sg_row_norm_sum": [_vp, _i, _vp, _vp],
    "sg_row_sqnorm": [_vp, _i, _i, _vp, _vp],
    "sg_kth_smallest_rows": [_vp, _ll, _i, _i, _i, _vp, _vp, _vp],
    "sg_prdc_rows": [_vp, _ll, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "sg_topk_select": [_vp, _i, _i, _vp, _vp, _vp],
    "sg_lecam": [_vp, _vp, _i, _f, _f, _vp, _vp, _vp, _vp],
    "sg_u8_to_nhwc": [_i, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "sg_gather_images_u8": [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp],
    "sg_topk_scatter": [_vp, _vp, _i, _vp, _i, _vp],
    "sg_bias_act": [_i, _vp, _vp, _vp, _vp, _vp, _vp, _ll, _ll, _i, _i, _i, _f, _f, _f, _vp],
    "sg_upfirdn2d": [_i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _f, _vp],
    "sg_filtered_lrelu": [_i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _f, _f, _f, _i, _vp],
    "sg_augment_work_floats": [C.POINTER(AugDesc)],       # returns a count (call through lib(), not call())
    "sg_augment_fwd": [C.POINTER(AugDesc), _vp, _vp, _vp, _vp],
    "sg_augment_bwd": [C.POINTER(AugDesc), _vp, _vp, _vp, _vp],
    "sg_mse_work_floats": [],                              # returns a count
    "sg_mse_fwd": [_vp, _vp, _ll, _vp, _vp, _vp],
    "sg_mse_bwd": [_vp, _vp, _vp, _ll, _vp, _vp, _vp],
    "sg_loss_ls_d": [_vp, _vp, _i, _vp, _vp, _vp, _vp],
    "sg_loss_ls_g": [_vp, _i, _vp, _vp, _vp],
    "sg_fm_work_floats": [_i],                             # returns a count
    "sg_fm_loss": [_vp, _vp, _i, _i, _vp, _vp, _vp, _vp],
    "sg_exp_fwd": [_vp, _vp, _ll, _vp],
    "sg_exp_bwd": [_vp, _vp, _vp, _ll, _vp],
    "sg_normal_nll": [_vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp],
    "sg_maxpool2_gather": [_i, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    "sg_softmax_rows_bwd2": [_vp, _vp, _vp, _vp, _ll, _i, _vp],
    "sg_scale_by_ptr": [_i, _vp, _vp, _vp, _ll, _vp],
    "sg_tanh_bwd": [_vp, _vp, _vp, _ll, _vp],
    "sg_tanh_bwd2": [_vp, _vp, _vp, _vp, _ll, _vp],
    "sg_clamp_flat": [_vp, _ll, _f, _f, _vp],
    "sg_select_rows": [_vp, _vp, _vp, _vp, _i, _ll, _vp],
    "sg_sign_count": [_vp, _i, _vp, _vp],
    "sg_reflect_pad2d_fwd": [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp],
    "sg_reflect_pad2d_bwd": [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp],
    "sg_affine_sample_fwd": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "sg_affine_sample_bwd": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "sg_color_affine": [_vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "sg_fir_reflect": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp],
    "sg_ada_noise_cutout": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
}


def exported_symbols():
    """Names every entry point include/sgamd.h declares (used by the CPU-side ABI test)."""
    return sorted(list(_PROTOS.keys()) + ["sg_last_error", "sg_version"])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` (make -C {CSRC}). "
                "There is no CPU/eager fallback on the product path.")
        l = C.CDLL(LIB_PATH)
        l.sg_last_error.restype = C.c_char_p
        l.sg_version.restype = _i
        for name, args in _PROTOS.items():
            fn = getattr(l, name)
            fn.argtypes = args
            fn.restype = _i
        l.sg_conv_rs_launches.restype = _ll
        # SG_F32_MODE=bf16x3: the fp32 convolutions (forward, data gradient, weight gradient) of the generic engine process-wide on the split-precision
        # path (functional.f32_mode is the scoped form); default = exact fp32 MFMA
        if os.environ.get("SG_F32_MODE", "exact") == "bf16x3":
            l.sg_set_f32_mode(3)
        _lib = l
    return _lib


def check(rc, name=""):
    if rc != 0:
        raise RuntimeError(f"{name}: {lib().sg_last_error().decode()} (rc={rc})")


# Entry points that write parameters / buffers of a network through raw pointers (invisible to torch's version counters): every call moves the
# epoch the weight bank's frozen-network cache is keyed on (bank.WeightBank.begin_forward).
_STATE_WRITERS = frozenset(("sg_adam_ema", "sg_ema_lerp", "sg_allreduce_flat", "sg_clamp_flat", "sg_allgather_flat"))
write_epoch = [0]


def call(name, *args):
    if name in _STATE_WRITERS:
        write_epoch[0] += 1
    rc = getattr(lib(), name)(*args)
    if rc != 0:
        raise RuntimeError(f"{name}: {lib().sg_last_error().decode()} (rc={rc})")


def stream():
    return torch.cuda.current_stream().cuda_stream


def upload_bytes(raw, device):
    """A descriptor table (bytes-like) -> uint8 tensor on `device`, WITHOUT stalling the launching thread. A host-to-device copy from pageable memory blocks the host until
    the stream has reached the copy: one such upload inside a training step collapses the host's lead over the GPU to zero (round 6, tools/host_probe.py: the first steps
    after a synchronisation were issued in ~110 ms instead of ~22 ms because of ONE table upload per step). From pinned memory the copy is queued like a kernel; torch's
    caching host allocator keeps the staging block alive until the copy has run."""
    t = torch.frombuffer(bytearray(raw), dtype=torch.uint8)
    if torch.device(device).type == "cuda":
        return t.pin_memory().to(device, non_blocking=True)
    return t.to(device)


def dt(t):
    """SG dtype code of a tensor / torch dtype."""
    d = t if isinstance(t, torch.dtype) else t.dtype
    if d == torch.float32:
        return F32
    if d == torch.bfloat16:
        return BF16
    raise RuntimeError(f"unsupported dtype {d} (float32 / bfloat16 only)")


def require_gpu(dev):
    if dev.type != "cuda":
        raise RuntimeError("studiogan_amd modules run on the GPU only (no CPU fallback on the product path)")


def ptr(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("studiogan_amd kernels need a GPU tensor (no CPU fallback on the product path)")
    return t.data_ptr()
