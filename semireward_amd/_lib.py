"""ctypes binding of libsrhip.so -- the ONLY compute path of this package.

There is deliberately no fallback: if the HIP library is missing or fails to load, importing an op
raises.  Signatures mirror include/srhip.h one to one.
"""
import ctypes
import os
from ctypes import c_double, c_float, c_int, c_long, c_longlong, c_uint, c_ulonglong, c_void_p

# torch must be imported BEFORE libsrhip.so is dlopen'ed: torch ships its own libamdhip64; if libsrhip pulled the
# system copy in first the process would hold two HIP runtimes and our launches would see "no device" (hipError 100).
import torch  # noqa: F401  (plumbing: device memory + streams)

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libsrhip.so")

EPI_BF16, EPI_GELU_BF16, EPI_RESID_F32, EPI_DGELU_BF16, EPI_F32 = range(5)

P, I, F, L, Dbl, U = c_void_p, c_int, c_float, c_long, c_double, c_uint
SIGNATURES = {
    "srhip_gemm_nt": (I, [I, P, I, P, I, P, I, I, I, I, P, P, I, P, P, I, F, F, P]),
    "srhip_gemm_nt_plan": (I, [I, I, I, I, F]),
    "srhip_gemm_small_max_grid": (I, [I]),
    "srhip_gemm_nt_grouped_f32": (I, [P, I, I, F, F, P]),
    "srhip_gemm_nt_grouped_n64_f32": (I, [P, I, I, F, F, P]),
    "srhip_gemm_tn_grouped_f32": (I, [P, I, I, F, F, P]),
    "srhip_gemm_tn_grouped_pp_f32": (I, [P, I, I, F, F, P]),
    "srhip_slab_reduce_f32": (I, [P, I, I, P]),
    "srhip_attn_fwd": (I, [P, P, P, I, I, I, F, P]),
    "srhip_attn_bwd": (I, [P, P, P, P, P, P, I, I, I, F, P]),
    "srhip_layernorm_fwd": (I, [P, P, P, F, P, P, P, I, I, P]),
    "srhip_layernorm_bwd": (I, [P, P, P, P, P, P, P, P, I, I, P]),
    "srhip_layernorm_bwd_cast": (I, [P, P, P, P, P, P, P, P, P, P, I, I, I, P]),
    "srhip_layernorm_bwd_part": (I, [P, P, P, P, P, P, P, I, P, P, I, I, I, P]),
    "srhip_ln_grad_reduce": (I, [P, P, I, I, I, P]),
    "srhip_mlp_fused": (I, [P, P, P, P, F, P, P, P, P, P, I, I, P, P, P, P, P, I, I, I, P]),
    "srhip_attn_block_supported": (I, [I, I, I]),
    "srhip_attn_block_fused": (I, [P, P, P, P, P, P, I, I, I, I, F, P]),
    "srhip_gelu_eval": (I, [P, P, P, I, P]),
    "srhip_mlp_fused_proj": (I, [P, P, P, P, P, P, I, P, P, F, P, P, P, P, P, I, P, P, P, I, I, I, P]),
    "srhip_adam_bias_corrections": (I, [F, F, I, P]),
    "srhip_adamw_flat_dyn": (I, [P, P, P, P, P, P, P, I, P, P, P, F, F, F, Dbl, F, P, I, P]),
    "srhip_adam_flat_dyn": (I, [P, P, P, P, L, F, F, F, F, P, P]),
    "srhip_droppath_fill_cols_dyn": (I, [P, P, P, I, I, I, P, c_ulonglong, P]),
    "srhip_patch_embed_fwd": (I, [P, P, P, P, P, P, P, I, I, I, I, I, P]),
    "srhip_patch_embed_bwd": (I, [P, P, P, P, P, P, P, I, I, I, I, I, P]),
    "srhip_patch_embed_bwd_ws_floats": (L, [I, I, I, I, I]),
    "srhip_patch_embed_bwd_ws": (I, [P, P, P, P, P, P, P, P, I, I, I, I, I, P]),
    "srhip_patch_im2col": (I, [P, P, P, I, I, I, I, P]),
    "srhip_patch_assemble": (I, [P, P, P, P, P, I, I, I, P]),
    "srhip_patch_grad_operands": (I, [P, P, P, P, I, I, I, P]),
    "srhip_cls_head_fwd": (I, [P, P, P, F, P, P, P, P, P, P, I, I, I, I, P]),
    "srhip_cls_head_fwd_scatter": (I, [P, P, P, F, P, P, P, P, P, P, P, P, P, I, I, I, I, P]),
    "srhip_cls_head_bwd": (I, [P, P, P, P, P, P, P, P, P, P, P, I, I, I, I, P]),
    "srhip_cast_scale_rows": (I, [P, P, I, P, L, I, P]),
    "srhip_transpose_to_bf16": (I, [P, I, I, P, I, I, I, I, I, P, P]),
    "srhip_transpose_batched": (I, [P, I, I, P]),
    "srhip_cast_f32_bf16": (I, [P, P, L, P]),
    "srhip_droppath_fill": (I, [P, P, I, I, c_ulonglong, P]),
    "srhip_droppath_fill_cols": (I, [P, P, P, I, I, I, c_ulonglong, P]),
    "srhip_row_max": (I, [P, I, P, P, P, I, I, P]),
    "srhip_row_max_strided": (I, [P, I, P, P, P, I, I, I, c_longlong, P]),
    "srhip_flexmatch_mask": (I, [P, P, P, F, P, P, P, P, I, I, I, I, P]),
    "srhip_flexmatch_mask_passes": (I, [P, P, P, F, P, P, P, P, I, I, I, I, I, P]),
    "srhip_flexmatch_rebuild_hist": (I, [P, P, I, I, P]),
    "srhip_fixed_mask": (I, [P, F, P, I, P]),
    "srhip_freematch_stats": (I, [P, P, P, P, I, I, P]),
    "srhip_freematch_update": (I, [P, I, P, P, P, P, P, P, P, P, I, I, F, F, I, I, P]),
    "srhip_freematch_entropy": (I, [P, P, P, P, F, P, P, P, I, I, I, P]),
    "srhip_distalign": (I, [P, P, I, P, I, P, P, P, Dbl, P, P, P, I, I, P]),
    "srhip_softmatch_mask": (I, [P, I, P, P, Dbl, I, P, I, P]),
    "srhip_reward_mask2": (I, [P, P, P, P, I, I, P]),
    "srhip_masked_ce": (I, [P, P, P, P, F, P, P, I, I, P]),
    "srhip_rewarder_param_count": (L, [I, I]),
    "srhip_rewarder_ws_floats": (L, [I, I]),
    "srhip_generator_param_count": (L, [I]),
    "srhip_rewarder_t_floats": (L, [I]),
    "srhip_generator_t_floats": (L, [I]),
    "srhip_rewarder_prepare": (I, [P, P, I, I, P]),
    "srhip_generator_prepare": (I, [P, P, I, P]),
    "srhip_rewarder_fwd": (I, [P, P, P, P, P, P, I, I, I, I, I, P]),
    "srhip_rewarder_fwd_strided": (I, [P, P, P, c_longlong, P, P, P, P, I, I, I, I, I, P]),
    "srhip_rewarder_bwd": (I, [P, P, P, P, P, P, P, I, I, I, P]),
    "srhip_generator_fwd": (I, [P, P, P, P, P, I, I, P]),
    "srhip_sr_target": (I, [P, P, P, I, I, P]),
    "srhip_label_error": (I, [P, I, P]),
    "srhip_index_error": (I, [P, I, P]),
    "srhip_prof_enable": (I, [I]),
    "srhip_prof_count": (I, []),
    "srhip_prof_elapsed_ms": (I, [I, I, P]),
    "srhip_adam_flat": (I, [P, P, P, P, L, F, F, F, F, I, P]),
    "srhip_adamw_flat": (I, [P, P, P, P, P, P, P, I, P, P, F, F, F, F, I, Dbl, F, P, I, P]),
    "srhip_clip_grad_ws_floats": (I, []),
    "srhip_clip_grad_coef": (I, [P, L, F, F, P, P, P]),
    "srhip_nchw_to_nhwc_bf16": (I, [P, P, I, I, I, I, P]),
    "srhip_im2col": (I, [P, P, I, I, I, I, I, I, I, P]),
    "srhip_im2col_bn": (I, [P, P, P, P, P, F, I, P, I, I, I, I, I, I, I, P]),
    "srhip_col2im": (I, [P, P, I, I, I, I, I, I, I, I, P]),
    "srhip_conv_weight_prep": (I, [P, P, P, I, I, I, I, P]),
    "srhip_add_unpad": (I, [P, P, I, I, I, I, P]),
    "srhip_conv_weight_prep_grouped": (I, [P, I, c_longlong, P]),
    "srhip_add_unpad_grouped": (I, [P, I, c_longlong, P]),
    "srhip_conv_weight_flip_grouped": (I, [P, I, c_longlong, P]),
    "srhip_bn_ws_doubles": (ctypes.c_longlong, []),
    "srhip_wrn_conv_supported": (I, [I, I, I]),
    "srhip_wrn_conv_last_plan": (I, []),
    "srhip_bn_acc_doubles": (ctypes.c_longlong, [I]),
    "srhip_wrn_conv_bn": (I, [P, I, P, P, P, P, P, F, F, P, P, P, P, F, I, P, P, P, I, I, I, I, I, I, I, I, P, I, P]),
    "srhip_wrn_head": (I, [P, I, P, P, P, P, P, F, F, P, P, P, P, F, I, P, P, P, P, I, I, I, I, I, P]),
    "srhip_wrn_conv_bn_passes": (I, [P, I, P, P, P, P, P, F, F, P, P, P, P, F, I, P, P, P, I, I, I, I, I, I, I, I, P, I, I, P]),
    "srhip_wrn_head_passes": (I, [P, I, P, P, P, P, P, F, F, P, P, P, P, F, I, P, P, P, P, I, I, I, I, I, I, P]),
    "srhip_bn_stats": (I, [P, F, F, I, P, P, P, P, P, I, I, P]),
    "srhip_bn_act": (I, [P, P, P, P, P, F, F, I, P, P, I, I, P]),
    "srhip_bn_accumulate": (I, [P, P, I, I, P]),
    "srhip_bn_fold": (I, [P, Dbl, F, F, I, P, P, P, P, P, I, P]),
    "srhip_bn_bwd_reduce": (I, [P, P, P, P, P, P, F, P, I, I, P]),
    "srhip_bn_bwd_apply": (I, [P, P, P, P, P, P, F, P, P, P, P, P, P, Dbl, P, I, I, P]),
    "srhip_bn_fwd": (I, [P, P, P, F, F, F, I, I, P, P, P, P, P, P, P, I, I, P]),
    "srhip_bn_bwd": (I, [P, P, P, P, P, P, F, P, P, P, P, P, I, I, P]),
    "srhip_avgpool_fwd": (I, [P, P, I, I, I, P]),
    "srhip_avgpool_bwd": (I, [P, P, I, I, I, P]),
    "srhip_fc_fwd": (I, [P, P, P, P, I, I, I, P]),
    "srhip_fc_bwd": (I, [P, P, P, P, P, P, I, I, I, P]),
    "srhip_sgd_flat": (I, [P, P, P, P, P, I, L, F, F, F, P, Dbl, I, I, P]),
    "srhip_gemm_nt_resid_dropout": (I, [P, I, P, I, P, I, I, I, I, P, P, I, U, U, F, P]),
    "srhip_gemm_nt_resid_ln_dropout": (I, [P, I, P, I, P, I, I, I, I, P, P, P, P, P, U, U, F, P]),
    "srhip_attn_masked_fwd": (I, [P, P, P, P, I, I, I, F, U, U, F, P]),
    "srhip_attn_masked_bwd": (I, [P, P, P, P, P, P, P, I, I, I, F, U, U, F, P]),
    "srhip_embed_ln_fwd": (I, [P, I, P, P, P, P, P, P, F, P, P, P, P, I, I, I, U, U, F, P]),
    "srhip_embed_ln_bwd": (I, [P, P, I, P, P, P, P, P, P, P, P, P, P, P, P, I, I, I, I, U, U, F, P]),
    "srhip_postln_fwd": (I, [P, P, P, F, P, P, P, P, I, I, P]),
    "srhip_postln_bwd": (I, [P, P, P, P, P, P, P, P, P, I, I, U, U, F, P]),
    "srhip_postln_bwd_part": (I, [P, P, P, P, P, P, P, P, I, I, I, U, U, F, P]),
    "srhip_meanpool_fwd": (I, [P, P, P, I, I, I, U, U, F, P]),
    "srhip_meanpool_bwd": (I, [P, P, P, I, I, I, U, U, F, P]),
    "srhip_gelu_f32": (I, [P, P, L, P]),
    "srhip_gelu_bwd_f32": (I, [P, P, P, L, P]),
    "srhip_mask_lengths": (I, [P, I, P, I, I, P]),
    "srhip_dropout_cast": (I, [P, P, L, U, U, F, P]),
    "srhip_augment": (I, [P, I, I, I, I, I, I, P, P, P, P, P, P, P, P]),
    "srhip_gemm_nt_dropout": (I, [I, P, I, P, I, P, I, I, I, I, P, P, P, I, U, U, F, P]),
    "srhip_w2v_conv0": (I, [I, P, P, P, P, P, P, P, P, P, P, P, I, I, I, I, I, I, I, F, P]),
    "srhip_w2v_conv_weight_prep": (I, [P, P, P, I, I, I, P]),
    "srhip_w2v_conv_wgrad_add": (I, [P, P, I, I, I, I, P]),
    "srhip_w2v_col2im_dgelu": (I, [P, P, P, I, I, I, I, I, I, P]),
    "srhip_w2v_featln_fwd": (I, [P, P, P, F, P, P, P, I, I, I, I, P]),
    "srhip_w2v_featln_bwd": (I, [P, P, P, P, P, P, P, P, P, I, I, I, I, P]),
    "srhip_w2v_spec_mask_fwd": (I, [P, P, P, L, I, P]),
    "srhip_w2v_spec_mask_bwd": (I, [P, P, P, P, I, I, I, I, I, P]),
    "srhip_w2v_pos_stage": (I, [P, P, I, I, I, I, I, I, I, L, P]),
    "srhip_w2v_weightnorm_prep": (I, [P, P, P, P, P, I, I, I, P]),
    "srhip_w2v_weightnorm_bwd": (I, [P, P, P, P, P, P, I, I, I, P]),
    "srhip_w2v_pos_finish_fwd": (I, [P, P, P, P, P, F, P, P, P, P, P, I, I, I, I, I, U, U, F, P]),
    "srhip_w2v_pos_finish_bwd": (I, [P, P, P, P, P, P, P, P, P, P, I, I, I, I, I, U, U, F, P]),
}

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "semireward_amd: %s is missing -- the HIP extension is the only compute path. "
                "Build it with `python -c 'import __graft_entry__ as g; g.build()'`." % LIB_PATH)
        h = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(h, name)          # AttributeError here == header / library mismatch: fail loudly
            fn.restype, fn.argtypes = res, args
        _lib = h
    return _lib


def check(rc, name):
    if rc != 0:
        raise RuntimeError("libsrhip: %s failed with code %d%s" % (
            name, rc, " (invalid argument)" if rc == -1 else " (hip launch error %d)" % (-rc - 2)))
