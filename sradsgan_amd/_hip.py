"""ctypes binding of libsradsgan_hip.so (the C ABI declared in include/sradsgan_hip.h).

There is deliberately NO fallback: if the shared library is missing or a call fails, the caller gets
an exception.  PyTorch is only used to own device memory and streams; raw pointers cross the ABI.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'lib', 'libsradsgan_hip.so')

_vp, _i, _f, _l, _sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_long, ctypes.c_size_t

# name -> (restype, argtypes); must list every symbol of include/sradsgan_hip.h (tests check this)
SIGNATURES = {
    'srhip_last_error': (ctypes.c_char_p, []),
    'srhip_abi_version': (_i, []),
    'srhip_stream_fork': (_i, [_vp, _vp]),
    'srhip_debug_set': (_i, [_i, _i]),
    'srhip_probe_config': (_i, [_i] * 7),
    'srhip_probe_read': (_i, [_vp, _vp, _i]),
    'srhip_set_conv_math': (_i, [_i]),
    'srhip_get_conv_math': (_i, []),
    'srhip_resample_ksize': (_i, [_i, _i, _i]),
    'srhip_resample_coeffs': (_i, [_i, _i, _i, _vp, _vp]),
    'srhip_resample_pass_u8': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    'srhip_u8_to_float': (_i, [_vp, _vp, _l, _vp]),
    'srhip_packed_elems': (_sz, [_i] * 5),
    'srhip_pack_entry_bytes': (_i, []),
    'srhip_packed_is_fast': (_i, [_i] * 5),
    'srhip_pack_weights_batched': (_i, [_vp, _i, _vp]),
    'srhip_pack_weight': (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    'srhip_conv2d_fwd': (_i, [_vp] * 7 + [_i] * 12 + [_f, _i, _vp]),
    'srhip_conv2d_dgrad': (_i, [_vp] * 5 + [_f] + [_i] * 13 + [_vp]),
    'srhip_conv2d_wgrad_workspace': (_sz, [_i] * 9),
    'srhip_conv2d_wgrad_can_accumulate': (_i, [_i] * 4),
    'srhip_conv2d_wgrad_multi_ok': (_i, [_i] * 9),
    'srhip_conv2d_wgrad_multi': (_i, [_i] + [_vp] * 4 + [_i, _vp, _sz] + [_i] * 11 + [_vp]),
    'srhip_pp_guard': (_i, [_i]),
    'srhip_pp_plane_pixels': (_l, [_i] * 3),
    'srhip_pp_from_f32': (_i, [_vp, _vp] + [_i] * 5 + [_vp]),
    'srhip_pp_to_f32': (_i, [_vp, _vp] + [_i] * 5 + [_vp]),
    'srhip_conv2d_pp_ok': (_i, [_i] * 5),
    'srhip_conv2d_fwd_dual': (_i, [_vp] * 9 + [_i] * 12 + [_f, _i, _vp]),
    'srhip_conv2d_fwd_pp': (_i, [_vp, _i, _vp, _vp, _vp, _i, _vp, _sz, _vp] + [_i] * 5 + [_f, _i, _vp]),
    'srhip_conv2d_dgrad_pp': (_i, [_vp, _i, _vp, _vp, _i, _vp, _vp, _f] + [_i] * 5 + [_vp]),
    'srhip_conv2d_dgrad_res3': (_i, [_vp] * 6 + [_i] * 9 + [_vp]),
    'srhip_conv2d_dgrad_pp_res3': (_i, [_vp, _i] + [_vp] * 5 + [_i] * 5 + [_vp]),
    'srhip_conv2d_pp_sign_bytes': (_sz, [_i] * 4),
    'srhip_conv2d_fwd_pp_signs': (_i, [_vp, _i, _vp, _vp, _vp, _vp, _sz] + [_i] * 5 + [_f, _vp]),
    'srhip_conv2d_dgrad_pp_signs': (_i, [_vp, _i, _vp, _vp, _vp, _sz, _f] + [_i] * 5 + [_vp]),
    'srhip_conv2d_wgrad_pp_ok': (_i, [_i] * 5),
    'srhip_conv2d_wgrad_pp_workspace': (_sz, [_i] * 8),
    'srhip_conv2d_wgrad_pp': (_i, [_i, _vp, _vp, _i, _i, _vp, _vp, _i, _vp, _sz] + [_i] * 6 + [_vp]),
    'srhip_conv2d_wgrad_act_ok': (_i, [_i] * 9),
    'srhip_conv2d_wgrad_act': (_i, [_vp] * 3 + [_f] + [_vp] * 3 + [_sz] + [_i] * 11 + [_vp]),
    'srhip_conv2d_wgrad': (_i, [_vp] * 6 + [_i, _vp, _sz] + [_i] * 11 + [_vp]),
    'srhip_colsum_workspace': (_sz, [_l, _i]),
    'srhip_colsum': (_i, [_vp, _vp, _vp, _sz, _l, _i, _i, _vp]),
    'srhip_lrelu_bwd': (_i, [_vp, _vp, _vp, _l, _f, _vp]),
    'srhip_lrelu_mask_bytes': (_sz, [_l]),
    'srhip_lrelu_bwd_bits': (_i, [_vp, _vp, _vp, _vp, _l, _f, _vp]),
    'srhip_maxpool2x2_fwd': (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    'srhip_maxpool2x2_bwd': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    'srhip_maxpool2x2_fwd_idx': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    'srhip_maxpool2x2_bwd_idx': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    'srhip_pixel_shuffle_fwd': (_i, [_vp, _vp, _i, _i, _i, _i, _i, _f, _i, _vp]),
    'srhip_pixel_shuffle_bwd': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _i, _vp]),
    'srhip_attn_tail_workspace': (_sz, [_i]),
    'srhip_attn_tail_fwd': (_i, [_vp] * 12 + [_sz] + [_i] * 5 + [_vp]),
    'srhip_attn_tail_eval': (_i, [_vp] * 9 + [_sz] + [_i] * 5 + [_vp]),
    'srhip_attn_tail_bwd_workspace': (_sz, [_i] * 3),
    'srhip_attn_tail_bwd_spatial': (_i, [_vp] * 10 + [_i, _vp, _sz] + [_i] * 4 + [_vp]),
    'srhip_attn_tail_mlp_workspace': (_sz, [_i, _i]),
    'srhip_attn_tail_bwd_mlp': (_i, [_vp] * 10 + [_i, _vp, _sz] + [_i] * 3 + [_vp]),
    'srhip_attn_tail_bwd_fused_workspace': (_sz, [_i] * 4),
    'srhip_attn_tail_bwd': (_i, [_vp] * 14 + [_i] + [_vp] * 2 + [_i, _vp, _sz] + [_i] * 5 + [_vp]),
    'srhip_attn_tail_bwd_pp': (_i, [_vp] * 15 + [_i] + [_vp] * 2 + [_i, _vp, _sz] + [_i] * 5 + [_vp]),
    'srhip_attn_tail_bwd_channel': (_i, [_vp] * 4 + [_i] * 4 + [_vp]),
    'srhip_cgam_workspace': (_sz, [_i, _i]),
    'srhip_cgam_fwd': (_i, [_vp] * 5 + [_sz] + [_i] * 3 + [_vp]),
    'srhip_cgam_bwd': (_i, [_vp] * 6 + [_i, _vp, _sz] + [_i] * 3 + [_vp]),
    'srhip_sgam_flash_fwd': (_i, [_vp] * 8 + [_i] * 4 + [_vp]),
    'srhip_sgam_flash_bwd_workspace': (_sz, [_i, _i]),
    'srhip_sgam_flash_bwd': (_i, [_vp] * 11 + [_i, _vp, _sz] + [_i] * 4 + [_vp]),
    'srhip_cbam_pool_hw': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    'srhip_cbam_unpool_hw': (_i, [_vp, _vp, _vp, _i, _i, _i, _vp]),
    'srhip_cbam_pool_c': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    'srhip_cbam_unpool_c': (_i, [_vp, _vp, _vp, _i, _i, _i, _vp]),
    'srhip_cbam_scale': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    'srhip_cbam_dot': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    'srhip_sigmoid_fwd': (_i, [_vp, _vp, _l, _i, _i, _vp]),
    'srhip_sigmoid_bwd': (_i, [_vp, _vp, _vp, _l, _i, _i, _vp]),
    'srhip_sigmoid_bwd_bwd': (_i, [_vp] * 5 + [_l, _i, _i, _vp]),
    'srhip_reduce_workspace': (_sz, []),
    'srhip_l1_mean_fwd': (_i, [_vp] * 4 + [_sz, _l, _vp]),
    'srhip_l1_mean_bwd': (_i, [_vp] * 5 + [_l, _vp]),
    'srhip_mean_fwd': (_i, [_vp] * 3 + [_sz, _l, _vp]),
    'srhip_mean_bwd': (_i, [_vp] * 2 + [_l, _vp]),
    'srhip_gp_norm_penalty_fwd': (_i, [_vp] * 3 + [_sz, _l, _i, _vp]),
    'srhip_gp_norm_penalty_bwd': (_i, [_vp] * 3 + [_l, _i, _vp]),
    'srhip_bn_eval_fwd': (_i, [_vp] * 6 + [_l, _i, _f, _f, _i, _vp]),
    'srhip_dp_id_bytes': (_i, []),
    'srhip_dp_unique_id': (_i, [_vp]),
    'srhip_dp_init': (_i, [_vp, _i, _i]),
    'srhip_dp_world': (_i, []),
    'srhip_dp_rank': (_i, []),
    'srhip_dp_allreduce_bucket': (_i, [_vp, _sz, _vp]),
    'srhip_dp_broadcast': (_i, [_vp, _sz, _i, _vp]),
    'srhip_dp_finalize': (_i, []),
    'srhip_bn_workspace': (_sz, [_l, _i]),
    'srhip_bn_train_fwd': (_i, [_vp] * 9 + [_sz, _l, _i, _f, _f, _f, _i, _vp]),
    'srhip_bn_train_bwd': (_i, [_vp] * 10 + [_sz, _l, _i, _f, _i, _vp]),
    'srhip_bn_bwd2_workspace': (_sz, [_l, _i]),
    'srhip_bn_train_bwd_bwd': (_i, [_vp] * 11 + [_sz, _l, _i, _f, _i, _vp]),
    'srhip_clam_pool_segments': (_i, []),
    'srhip_clam_pool_max_segments': (_i, []),
    'srhip_clam_pool_partial': (_i, [_vp, _vp, _sz, _i, _i, _i, _i, _vp]),
    'srhip_conv2d_fwd_pool': (_i, [_vp, _vp, _vp, _vp, _vp, _sz, _vp] + [_i] * 8 + [_vp]),
    'srhip_attn_tail_fwd_pooled': (_i, [_vp, _vp, _sz, _i] + [_vp] * 10 + [_i] * 5 + [_vp]),
    'srhip_attn_tail_eval_pooled': (_i, [_vp, _vp, _vp, _sz, _i] + [_vp] * 6 + [_i] * 5 + [_vp]),
    'srhip_sum_n': (_i, [_vp, _i, _vp, _l, _vp]),
    'srhip_cat_channels': (_i, [_vp, _vp, _i, _vp, _l, _vp]),
    'srhip_split_channels': (_i, [_vp, _vp, _i, _vp, _l, _vp]),
    'srhip_bn_train_bwd_acc': (_i, [_vp] * 12 + [_sz, _l, _i, _f, _i, _vp]),
    'srhip_bn_train_bwd_acc_x': (_i, [_vp] * 12 + [_sz, _l, _i, _f, _i, _vp]),
    'srhip_bn_train_bwd_acc_xa': (_i, [_vp] * 13 + [_sz, _l, _i, _f, _i, _vp]),
    'srhip_bn_train_bwd_bwd_acc': (_i, [_vp] * 12 + [_sz, _l, _i, _f, _i, _vp]),
    'srhip_bn_train_bwd_bwd_acc_x': (_i, [_vp] * 12 + [_sz, _l, _i, _f, _i, _vp]),
    'srhip_metric_blocks': (_i, []),
    'srhip_quant_sse': (_i, [_vp, _vp, _vp, _i, _l, _vp]),
    'srhip_ssim_u8': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    'srhip_metric_finish': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, ctypes.c_double, _vp]),
    'srhip_adam_step': (_i, [_vp] * 5 + [_l] + [_f] * 6 + [_vp]),
}

_lib = None


class HipLibraryError(RuntimeError):
    pass


# arithmetic of the conv contraction unless SRADSGAN_CONV_MATH overrides it (include/sradsgan_hip.h, srhip_set_conv_math)
DEFAULT_CONV_MATH = 'bf16x3'


def lib():
    """Load (once) and return the shared library; raises HipLibraryError when it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipLibraryError(
                'libsradsgan_hip.so is not built (%s). Build it with `make -C sradsgan_amd/csrc` or '
                '`python -c "import __graft_entry__ as g; g.build()"`. There is no CPU fallback.' % LIB_PATH)
        handle = ctypes.CDLL(os.environ.get('SRHIP_LIB', LIB_PATH))     # SRHIP_LIB: same-box A/B of two builds (debug)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype, fn.argtypes = res, args
        mode = os.environ.get('SRADSGAN_CONV_MATH', DEFAULT_CONV_MATH)
        if mode not in ('fp32', 'bf16x3', 'half'):
            raise HipLibraryError("SRADSGAN_CONV_MATH must be 'fp32', 'bf16x3' or 'half', got %r" % mode)
        handle.srhip_set_conv_math({'fp32': 0, 'bf16x3': 1, 'half': 2}[mode])
        for item in os.environ.get('SRHIP_DEBUG', '').split(','):      # experiment knobs, "key:value,..." (sradsgan_hip.h)
            if item:
                key, value = item.split(':')
                handle.srhip_debug_set(int(key), int(value))
        _lib = handle
    return _lib


def check(rc, what=''):
    if rc != 0:
        msg = lib().srhip_last_error()
        raise HipLibraryError('%s failed (%d): %s' % (what or 'srhip call', rc, msg.decode() if msg else '?'))
