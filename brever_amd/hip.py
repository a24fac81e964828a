"""ctypes binding of ``libbrever_hip.so`` (C ABI: ``include/brever_hip.h``).

The HIP library is the product; there is no CPU or PyTorch fallback. Every
helper here raises ``RuntimeError`` when the shared library is missing or when
it is handed a tensor that is not on a ROCm device.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# BRV_LIB_PATH: load another build of the same library (tools/: diagnostic builds)
LIB_PATH = os.environ.get('BRV_LIB_PATH') or os.path.join(_HERE, 'csrc', 'libbrever_hip.so')

_c_i64 = ctypes.c_int64
_c_f32 = ctypes.c_float
_c_ptr = ctypes.c_void_p


class CtnConfig(ctypes.Structure):
    """``brv_ctn_config`` -- ConvTasNet.__init__ hyper-parameters."""
    _fields_ = [(name, ctypes.c_int32) for name in (
        'filters', 'filter_length', 'bottleneck_channels', 'hidden_channels',
        'skip_channels', 'kernel_size', 'layers', 'repeats', 'output_sources',
        'causal')]


class LaunchOpts(ctypes.Structure):
    """``brv_launch_opts`` -- per-call options of the Conv-TasNet entry points."""
    _fields_ = [('size', ctypes.c_uint32), ('flags', ctypes.c_uint32), ('cu_eighths', ctypes.c_int32),
                ('wg_target', ctypes.c_int32), ('prof', ctypes.c_void_p)]


OPT_NO_FWD_FUSE, OPT_NO_BWD_FUSE, OPT_NO_WS, OPT_DWPW2_WS = 0x001, 0x002, 0x004, 0x008
OPT_NO_DZ_FUSE, OPT_NO_DZ1_FUSE, OPT_NO_WGRAD_FULL, OPT_NO_WGRAD_SPLIT = 0x010, 0x020, 0x040, 0x080
OPT_NO_PW1_RC, OPT_PW1_RC_WGRAD, OPT_PW1_RC_TILES, OPT_DWPW2_V2, OPT_NO_WGRAD_128, OPT_BWD_PERSIST = 0x100, 0x200, 0x400, 0x800, 0x1000, 0x2000
# environment switch -> option flag (read by the HOST at call time; the library itself reads no
# environment). '0' selects the flag for the *_FUSE switches, any value for the BRV_NO_* ones.
_ENV_FLAGS = (('BRV_FWD_FUSE', OPT_NO_FWD_FUSE, '0'), ('BRV_BWD_FUSE', OPT_NO_BWD_FUSE, '0'),
              ('BRV_NO_WS', OPT_NO_WS, None), ('BRV_DWPW2_WS', OPT_DWPW2_WS, '1'),
              ('BRV_NO_DZ_FUSE', OPT_NO_DZ_FUSE, None), ('BRV_NO_DZ1_FUSE', OPT_NO_DZ1_FUSE, None),
              ('BRV_NO_WGRAD_FULL', OPT_NO_WGRAD_FULL, None), ('BRV_NO_WGRAD_SPLIT', OPT_NO_WGRAD_SPLIT, None),
              ('BRV_PW1_RC', OPT_NO_PW1_RC, '0'), ('BRV_PW1_RC_WGRAD', OPT_PW1_RC_WGRAD, '1'),
              ('BRV_PW1_RC_TILES', OPT_PW1_RC_TILES, '1'), ('BRV_DWPW2_V2', OPT_DWPW2_V2, '1'),
              ('BRV_WGRAD_128', OPT_NO_WGRAD_128, '0'), ('BRV_BWD_PERSIST', OPT_BWD_PERSIST, '1'))
_prof = None            # profiler handle of this process's calls (prof_enable)


def launch_opts(cu_eighths=8):
    """Options for one Conv-TasNet call: A/B switches from the environment, the share of
    the chip the persistent kernels take, the active profiler. Returns the struct (keep it alive for
    the duration of the call) -- pass ``ctypes.byref`` of it."""
    flags = 0
    for name, flag, on_value in _ENV_FLAGS:
        v = os.environ.get(name)
        if v is not None and (on_value is None or v == on_value):
            flags |= flag
    target = os.environ.get('BRV_WG_TARGET')
    return LaunchOpts(ctypes.sizeof(LaunchOpts), flags, int(cu_eighths),
                      int(target) if target else 0, _prof)


def opts_ptr(opts):
    return ctypes.byref(opts)


# name -> (restype, argtypes); the export test checks every name resolves.
SIGNATURES = {
    'brv_version': (ctypes.c_int, []),
    'brv_last_error': (ctypes.c_char_p, []),
    'brv_prof_create': (_c_ptr, [ctypes.c_int]),
    'brv_prof_collect': (_c_i64, [_c_ptr, ctypes.c_char_p, _c_i64]),
    'brv_prof_destroy': (None, [_c_ptr]),
    'brv_ctn_param_count': (_c_i64, [_c_ptr]),
    'brv_ctn_param_tensors': (_c_i64, [_c_ptr]),
    'brv_ctn_param_offset': (_c_i64, [_c_ptr, _c_i64]),
    'brv_ctn_frames': (_c_i64, [_c_ptr, _c_i64]),
    'brv_ctn_prepared_bytes': (_c_i64, [_c_ptr]),
    'brv_ctn_workspace_bytes': (_c_i64, [_c_ptr, _c_i64, _c_i64]),
    'brv_ctn_workspace_offset': (_c_i64, [_c_ptr, _c_i64, _c_i64,
                                          ctypes.c_char_p, _c_i64]),
    'brv_ctn_prepare': (ctypes.c_int, [_c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr]),
    'brv_ctn_forward': (ctypes.c_int, [_c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_i64,
                                       _c_ptr, _c_i64, _c_i64, _c_ptr, _c_ptr]),
    'brv_ctn_backward': (ctypes.c_int, [_c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_i64,
                                        _c_ptr, _c_ptr, _c_i64, _c_i64,
                                        _c_ptr, _c_ptr]),
    'brv_ctn_backward_part': (ctypes.c_int, [_c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_i64,
                                             _c_ptr, _c_ptr, _c_i64, _c_i64,
                                             ctypes.c_int32, ctypes.c_int32, _c_ptr, _c_ptr]),
    'brv_ctn_grad_bucket': (ctypes.c_int, [_c_ptr, ctypes.c_int32, ctypes.c_int32,
                                           _c_ptr, _c_ptr]),
    'brv_ctn_f32_backward_part': (ctypes.c_int, [_c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr,
                                                 _c_ptr, _c_i64, _c_i64, ctypes.c_int32,
                                                 ctypes.c_int32, _c_ptr]),
    'brv_ctn_f32_workspace_bytes': (_c_i64, [_c_ptr, _c_i64, _c_i64]),
    'brv_ctn_f32_forward': (ctypes.c_int, [_c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr,
                                           _c_i64, _c_i64, _c_ptr]),
    'brv_ctn_f32_backward': (ctypes.c_int, [_c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr,
                                            _c_ptr, _c_i64, _c_i64, _c_ptr]),
    'brv_resample_poly': (ctypes.c_int, [_c_ptr, _c_ptr, _c_ptr, _c_ptr] + [_c_i64]*7 + [_c_ptr]),
    'brv_stoi_frames': (_c_i64, [_c_i64]),
    'brv_stoi_compact': (ctypes.c_int, [_c_ptr, _c_ptr, _c_ptr, _c_i64, _c_i64, _c_ptr, _c_ptr,
                                        _c_i64, _c_ptr, _c_ptr, _c_ptr, _c_i64, _c_f32, _c_ptr]),
    'brv_stoi_bands': (ctypes.c_int, [_c_ptr, _c_ptr, _c_ptr, _c_i64, _c_i64, _c_i64, _c_i64,
                                      _c_ptr]),
    'brv_stoi_correlate': (ctypes.c_int, [_c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_i64, _c_i64,
                                          ctypes.c_int, _c_f32, _c_ptr]),
    'brv_flac_info': (ctypes.c_int, [_c_ptr, _c_i64, _c_ptr, _c_ptr, _c_ptr, _c_ptr]),
    'brv_flac_decode': (_c_i64, [_c_ptr, _c_i64, _c_ptr, _c_i64]),
    'brv_flac_encode16': (_c_i64, [_c_ptr, _c_i64, ctypes.c_int32, _c_ptr, _c_i64]),
    'brv_loss_scratch_bytes': (_c_i64, [_c_i64, _c_i64]),
    'brv_snr_forward_strided': (ctypes.c_int, [_c_ptr, _c_ptr, _c_i64, _c_i64, _c_ptr, _c_i64, _c_i64,
                                                _c_i64, _c_i64, _c_ptr, _c_ptr, _c_ptr]),
    'brv_snr_backward_strided': (ctypes.c_int, [_c_ptr, _c_ptr, _c_i64, _c_i64, _c_ptr, _c_i64, _c_i64,
                                                 _c_i64, _c_i64, _c_ptr, _c_ptr, _c_ptr, _c_ptr]),
    'brv_snr_forward': (ctypes.c_int, [_c_ptr, _c_ptr, _c_ptr, _c_i64, _c_i64,
                                       _c_i64, _c_i64, _c_ptr, _c_ptr, _c_ptr]),
    'brv_snr_backward': (ctypes.c_int, [_c_ptr, _c_ptr, _c_ptr, _c_i64, _c_i64,
                                        _c_i64, _c_i64, _c_ptr, _c_ptr, _c_ptr,
                                        _c_ptr]),
    'brv_sisnr_forward': (ctypes.c_int, [_c_ptr, _c_ptr, _c_ptr, _c_i64,
                                         _c_i64, _c_i64, _c_i64, _c_ptr,
                                         _c_ptr, _c_ptr]),
    'brv_sisnr_backward': (ctypes.c_int, [_c_ptr, _c_ptr, _c_ptr, _c_i64, _c_i64,
                                          _c_i64, _c_i64, _c_ptr, _c_ptr, _c_ptr,
                                          _c_ptr]),
    'brv_mse_backward': (ctypes.c_int, [_c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_i64,
                                        _c_i64, _c_i64, _c_i64, _c_ptr, _c_ptr,
                                        _c_ptr]),
    'brv_mse_forward': (ctypes.c_int, [_c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_i64,
                                       _c_i64, _c_i64, _c_i64, _c_ptr, _c_ptr,
                                       _c_ptr]),
    'brv_stft_frames': (_c_i64, [_c_i64, _c_i64, _c_i64]),
    'brv_stft_forward': (ctypes.c_int, [_c_ptr, _c_ptr, _c_ptr, _c_i64, _c_i64,
                                        _c_i64, _c_i64, _c_f32, _c_f32, _c_ptr]),
    'brv_istft_backward': (ctypes.c_int, [_c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr,
                                          _c_i64, _c_i64, _c_i64, _c_i64, _c_f32,
                                          _c_f32, _c_ptr]),
    'brv_dft64_forward': (ctypes.c_int, [_c_ptr, _c_ptr, _c_ptr] + [_c_i64]*7 + [_c_f32, _c_f32, _c_ptr]),
    'brv_dft64_synthesis': (ctypes.c_int, [_c_ptr, _c_ptr, _c_ptr] + [_c_i64]*4 + [_c_f32, _c_f32, _c_ptr]),
    'brv_overlap_add': (ctypes.c_int, [_c_ptr, _c_ptr, _c_ptr] + [_c_i64]*6 + [_c_ptr]),
    'brv_pad_signal': (ctypes.c_int, [_c_ptr, _c_ptr, _c_i64, _c_i64, _c_i64, _c_i64, ctypes.c_int, _c_ptr]),
    'brv_polar': (ctypes.c_int, [_c_ptr, _c_ptr, _c_ptr, _c_i64, _c_ptr]),
    'brv_mag_phase': (ctypes.c_int, [_c_ptr, _c_ptr, _c_ptr, _c_i64, _c_ptr]),
    'brv_spec_compress': (ctypes.c_int, [_c_ptr, _c_ptr, _c_i64, _c_f32, _c_f32, _c_ptr]),
    'brv_spec_compress_backward': (ctypes.c_int, [_c_ptr, _c_ptr, _c_ptr, _c_i64, _c_f32, _c_f32, _c_ptr]),
    'brv_matmul_f32': (ctypes.c_int, [_c_ptr, _c_ptr, _c_ptr, _c_i64, _c_i64,
                                      _c_i64, _c_i64, _c_i64, _c_ptr]),
    'brv_stft_adjoint': (ctypes.c_int, [_c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_i64, _c_i64,
                                        _c_i64, _c_i64, _c_f32, _c_ptr]),
    'brv_apply_mask': (ctypes.c_int, [_c_ptr, _c_ptr, _c_ptr, _c_i64, _c_i64, _c_i64, _c_ptr]),
    'brv_l1_forward': (ctypes.c_int, [_c_ptr, _c_ptr, _c_ptr, _c_i64, _c_i64, _c_ptr]),
    'brv_l1_backward': (ctypes.c_int, [_c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_i64, _c_i64,
                                       ctypes.c_int, _c_ptr]),
    'brv_mag_l1_forward': (ctypes.c_int, [_c_ptr, _c_ptr, _c_ptr, _c_i64, _c_i64, _c_ptr]),
    'brv_mag_l1_backward': (ctypes.c_int, [_c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_i64, _c_i64,
                                           _c_ptr]),
    'brv_framed_dft_forward': (ctypes.c_int, [_c_ptr, _c_ptr, _c_ptr] + [_c_i64]*6
                               + [_c_f32, _c_f32, _c_ptr]),
    'brv_framed_dft_transpose': (ctypes.c_int, [_c_ptr, _c_ptr, _c_ptr, _c_ptr] + [_c_i64]*6
                                 + [_c_f32, _c_f32, _c_ptr]),
    'brv_gemm_f32': (ctypes.c_int, [_c_ptr, _c_ptr, _c_ptr] + [_c_i64]*10
                     + [ctypes.c_int, ctypes.c_int, _c_i64, _c_i64, _c_i64, _c_ptr,
                        ctypes.c_int, _c_ptr]),
    'brv_gemm_f32_workspace_bytes': (ctypes.c_int64, [_c_i64]*4 + [ctypes.c_int, ctypes.c_int, _c_i64]),
    'brv_gemm_f32_ws': (ctypes.c_int, [_c_ptr, _c_ptr, _c_ptr] + [_c_i64]*10
                        + [ctypes.c_int, ctypes.c_int, _c_i64, _c_i64, _c_i64, _c_ptr,
                           ctypes.c_int, _c_ptr, _c_i64, _c_ptr]),
    'brv_gemm_bf16': (ctypes.c_int, [_c_ptr, _c_ptr, _c_ptr] + [_c_i64]*10
                      + [ctypes.c_int, ctypes.c_int, _c_i64, _c_i64, _c_i64, _c_ptr,
                         ctypes.c_int, _c_ptr]),
    'brv_gemm_bf16_mixed': (ctypes.c_int, [_c_ptr, _c_ptr, _c_ptr] + [_c_i64]*10
                            + [ctypes.c_int, ctypes.c_int, _c_i64, _c_i64, _c_i64, _c_ptr,
                               ctypes.c_int, ctypes.c_int, _c_ptr]),
    'brv_fbe_power': (ctypes.c_int, [_c_ptr, _c_ptr, _c_i64, _c_i64, _c_i64, _c_ptr]),
    'brv_compress': (ctypes.c_int, [_c_ptr, _c_ptr, _c_i64, ctypes.c_int, _c_f32, _c_ptr]),
    'brv_interaural_coherence': (ctypes.c_int, [_c_ptr, _c_ptr, _c_i64, _c_i64, _c_i64, _c_f32, _c_ptr]),
    'brv_binaural': (ctypes.c_int, [_c_ptr, _c_ptr, _c_i64, _c_i64, ctypes.c_int, _c_f32, _c_ptr]),
    'brv_col_normalize': (ctypes.c_int, [_c_ptr, _c_i64, _c_i64, _c_i64, _c_f32, _c_ptr]),
    'brv_deltas': (ctypes.c_int, [_c_ptr, _c_ptr, _c_i64, _c_i64, _c_i64, _c_ptr]),
    'brv_irm': (ctypes.c_int, [_c_ptr, _c_ptr, _c_ptr, _c_i64, _c_f32, _c_ptr]),
    'brv_stack_frames': (ctypes.c_int, [_c_ptr, _c_ptr, _c_i64, _c_i64, _c_i64, _c_i64, _c_ptr]),
    'brv_static_norm': (ctypes.c_int, [_c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_i64, _c_i64, _c_i64,
                                       _c_ptr]),
    'brv_cumulative_norm': (ctypes.c_int, [_c_ptr, _c_ptr, _c_i64, _c_i64, _c_f32, _c_ptr]),
    'brv_relu_dropout_forward': (ctypes.c_int, [_c_ptr, _c_ptr, _c_ptr, _c_i64, _c_f32, _c_ptr]),
    'brv_relu_dropout_backward': (ctypes.c_int, [_c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_i64, _c_f32,
                                                 _c_ptr]),
    'brv_dropout_apply': (ctypes.c_int, [_c_ptr, _c_ptr, _c_ptr, _c_i64, _c_f32, _c_ptr]),
    'brv_sigmoid_forward': (ctypes.c_int, [_c_ptr, _c_ptr, _c_i64, _c_ptr]),
    'brv_sigmoid_backward': (ctypes.c_int, [_c_ptr, _c_ptr, _c_ptr, _c_i64, _c_ptr]),
    'brv_row_sum': (ctypes.c_int, [_c_ptr, _c_ptr, _c_i64, _c_i64, _c_i64, _c_ptr]),
    'brv_masked_mean_spec': (ctypes.c_int, [_c_ptr, _c_ptr, _c_ptr, _c_i64, _c_i64, _c_i64,
                                            _c_ptr]),
    'brv_conv2d_forward': (ctypes.c_int, [_c_ptr, _c_ptr, _c_ptr, _c_ptr] + [_c_i64]*13
                           + [ctypes.c_int, _c_f32, _c_ptr]),
    'brv_conv_transpose2d_forward': (ctypes.c_int, [_c_ptr, _c_ptr, _c_ptr, _c_ptr] + [_c_i64]*15
                                     + [ctypes.c_int, _c_f32, _c_ptr]),
    'brv_batchnorm2d_forward': (ctypes.c_int, [_c_ptr]*9 + [_c_i64]*3
                                + [_c_f32, _c_f32, ctypes.c_int, _c_ptr]),
    'brv_lstm_recurrent_forward': (ctypes.c_int, [_c_ptr]*6 + [_c_i64]*4 + [_c_ptr]),
    'brv_conv2d_wgrad': (ctypes.c_int, [_c_ptr, _c_ptr, _c_ptr, _c_ptr] + [_c_i64]*15
                         + [ctypes.c_int, _c_f32, _c_ptr]),
    'brv_batchnorm2d_backward': (ctypes.c_int, [_c_ptr]*11 + [_c_i64, _c_i64, _c_i64, _c_ptr]),
    'brv_lstm_recurrent_backward': (ctypes.c_int, [_c_ptr]*5 + [_c_i64]*4 + [_c_ptr]),
    'brv_lstm_recurrent_bf16_supported': (ctypes.c_int, [_c_i64]),
    'brv_lstm_recurrent_forward_bf16': (ctypes.c_int, [_c_ptr]*6 + [_c_i64]*4 + [_c_ptr]),
    'brv_lstm_recurrent_backward_bf16': (ctypes.c_int, [_c_ptr]*5 + [_c_i64]*4 + [_c_ptr]),
    'brv_lstm_tile_supported': (ctypes.c_int, [_c_i64]),
    'brv_lstm_tile_forward': (ctypes.c_int, [_c_ptr]*6 + [_c_i64]*7 + [ctypes.c_int, _c_ptr]),
    'brv_lstm_tile_backward': (ctypes.c_int, [_c_ptr]*5 + [_c_i64]*7 + [ctypes.c_int, _c_ptr]),
    'brv_dccrn_apply_mask_backward': (ctypes.c_int, [_c_ptr]*7 + [_c_i64, _c_ptr]),
    'brv_istft_env_divide': (ctypes.c_int, [_c_ptr, _c_ptr, _c_ptr] + [_c_i64]*5 + [_c_ptr]),
    'brv_combine': (ctypes.c_int, [_c_ptr, _c_ptr, _c_ptr, _c_i64, _c_f32, _c_ptr]),
    'brv_dccrn_apply_mask': (ctypes.c_int, [_c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_i64,
                                            _c_ptr]),
    'brv_causal_groupnorm_scratch_bytes': (_c_i64, [_c_i64, _c_i64, _c_i64]),
    'brv_causal_groupnorm_forward': (ctypes.c_int, [_c_ptr]*6 + [_c_i64]*5 + [_c_f32, _c_ptr]),
    'brv_causal_groupnorm_backward': (ctypes.c_int, [_c_ptr]*9 + [_c_i64]*5 + [_c_ptr]),
    'brv_rownorm_forward': (ctypes.c_int, [_c_ptr]*6 + [_c_i64]*4 + [_c_f32, _c_ptr]),
    'brv_rownorm_scratch_bytes': (_c_i64, [_c_i64, _c_i64]),
    'brv_rownorm_backward': (ctypes.c_int, [_c_ptr]*10 + [_c_i64]*4 + [_c_ptr]),
    'brv_col_sum_scratch_bytes': (_c_i64, [_c_i64, _c_i64]),
    'brv_linear_small_supported': (ctypes.c_int, [_c_i64, _c_i64, _c_i64]),
    'brv_linear_small': (ctypes.c_int, [_c_ptr]*4 + [_c_i64]*6 + [ctypes.c_int, ctypes.c_int, _c_ptr]),
    'brv_linear_small_wgrad_supported': (ctypes.c_int, [_c_i64, _c_i64, _c_i64]),
    'brv_linear_small_wgrad_scratch_bytes': (_c_i64, [_c_i64, _c_i64]),
    'brv_linear_small_wgrad': (ctypes.c_int, [_c_ptr]*4 + [_c_i64]*6 + [_c_ptr]),
    'brv_col_sum': (ctypes.c_int, [_c_ptr, _c_ptr, _c_ptr, _c_i64, _c_i64, _c_i64, _c_ptr]),
    'brv_col_sum_bf16': (ctypes.c_int, [_c_ptr, _c_ptr, _c_ptr, _c_i64, _c_i64, _c_i64, _c_ptr]),
    'brv_row_std': (ctypes.c_int, [_c_ptr, _c_ptr, _c_i64, _c_i64, _c_ptr]),
    'brv_row_scale': (ctypes.c_int, [_c_ptr]*3 + [_c_i64, _c_i64, ctypes.c_int, _c_ptr]),
    'brv_head_permute_supported': (ctypes.c_int, [_c_i64]*3),
    'brv_head_permute': (ctypes.c_int, [_c_ptr]*2 + [_c_i64]*5 + [ctypes.c_int, _c_ptr]),
    'brv_cplx_moments': (ctypes.c_int, [_c_ptr, _c_ptr, _c_i64, _c_i64, _c_i64, _c_ptr]),
    'brv_cplx_affine_forward': (ctypes.c_int, [_c_ptr]*5 + [_c_i64]*3 + [_c_ptr]),
    'brv_cplx_affine_backward': (ctypes.c_int, [_c_ptr]*9 + [_c_i64]*3 + [_c_ptr]),
    'brv_cplx_moments_backward': (ctypes.c_int, [_c_ptr]*3 + [_c_i64]*3 + [_c_ptr]),
    'brv_im2col': (ctypes.c_int, [_c_ptr, _c_ptr] + [_c_i64]*12 + [_c_ptr]),
    'brv_col2im': (ctypes.c_int, [_c_ptr, _c_ptr, _c_ptr] + [_c_i64]*12 + [_c_ptr]),
    'brv_im2col_bf16': (ctypes.c_int, [_c_ptr, _c_ptr] + [_c_i64]*12 + [_c_ptr]),
    'brv_col2im_bf16': (ctypes.c_int, [_c_ptr, _c_ptr, _c_ptr] + [_c_i64]*12 + [_c_ptr]),
    'brv_complex_weight_pack': (ctypes.c_int, [_c_ptr, _c_ptr, _c_ptr, _c_i64, _c_i64, _c_f32, _c_ptr]),
    'brv_complex_weight_unpack': (ctypes.c_int, [_c_ptr, _c_ptr, _c_ptr, _c_i64, _c_i64, _c_f32, _c_ptr]),
    'brv_groupnorm_scratch_bytes': (_c_i64, [_c_i64, _c_i64]),
    'brv_groupnorm_fold': (ctypes.c_int, [_c_ptr]*11 + [_c_i64]*4 + [_c_f32, _c_ptr]),
    'brv_groupnorm_backward': (ctypes.c_int, [_c_ptr]*12 + [_c_i64]*4 + [ctypes.c_int, _c_ptr]),
    'brv_affine_act_backward': (ctypes.c_int, [_c_ptr]*9 + [_c_i64]*3 + [ctypes.c_int, _c_ptr]),
    'brv_silu_backward': (ctypes.c_int, [_c_ptr, _c_ptr, _c_ptr, _c_i64, _c_ptr]),
    'brv_softmax_rows_backward': (ctypes.c_int, [_c_ptr, _c_ptr, _c_ptr, _c_i64, _c_i64, _c_ptr]),
    'brv_affine_act': (ctypes.c_int, [_c_ptr]*4 + [_c_i64]*3 + [ctypes.c_int, _c_ptr]),
    'brv_silu': (ctypes.c_int, [_c_ptr, _c_ptr, _c_i64, _c_ptr]),
    'brv_softmax_rows': (ctypes.c_int, [_c_ptr, _c_ptr, _c_i64, _c_i64, _c_ptr]),
    'brv_fir_resample2d': (ctypes.c_int, [_c_ptr, _c_ptr, _c_ptr] + [_c_i64]*8
                           + [ctypes.c_int, _c_f32, _c_ptr]),
    'brv_axpby': (ctypes.c_int, [_c_ptr, _c_f32, _c_ptr, _c_f32, _c_ptr, _c_i64, _c_ptr]),
    'brv_fourier_features': (ctypes.c_int, [_c_ptr, _c_ptr, _c_ptr, _c_i64, _c_i64, _c_ptr]),
    'brv_conv2d_packed_size': (_c_i64, [_c_i64, _c_i64, _c_i64]),
    'brv_conv2d_pack_f16': (ctypes.c_int, [_c_ptr, _c_ptr, _c_i64, _c_i64, _c_i64, _c_ptr]),
    'brv_conv2d_mfma_forward': (ctypes.c_int, [_c_ptr]*6 + [ctypes.c_int, _c_ptr] + [_c_i64]*8
                                + [_c_f32, _c_ptr]),
    'brv_gemm_bf16_conv': (ctypes.c_int, [_c_ptr]*3 + [_c_i64]*9 + [ctypes.c_int, ctypes.c_int]
                           + [_c_i64]*3 + [_c_ptr, ctypes.c_int, ctypes.c_int] + [_c_i64]*11 + [_c_ptr]),
    'brv_dccrn_apply_mask_batched': (ctypes.c_int, [_c_ptr]*3 + [_c_i64, _c_i64, _c_ptr]),
    'brv_dccrn_apply_mask_backward_batched': (ctypes.c_int, [_c_ptr]*4 + [_c_i64, _c_i64, _c_ptr]),
    'brv_complex_bias_pack': (ctypes.c_int, [_c_ptr]*3 + [_c_i64, _c_ptr]),
    'brv_complex_bias_unpack': (ctypes.c_int, [_c_ptr]*3 + [_c_i64, _c_ptr]),
    'brv_cconv_packed_bytes': (_c_i64, [_c_i64, _c_i64]),
    'brv_cconv_pack': (ctypes.c_int, [_c_ptr, _c_ptr] + [_c_i64]*4 + [_c_ptr]),
    'brv_cconv_pack_complex': (ctypes.c_int, [_c_ptr]*4 + [_c_i64]*3 + [_c_f32] + [_c_ptr]*3 + [_c_i64]*4 + [_c_ptr]
                               + [_c_i64]*4 + [_c_ptr]),
    'brv_cconv_rows': (ctypes.c_int, [_c_ptr, _c_ptr, _c_i64, _c_ptr, _c_ptr, _c_ptr, _c_ptr] + [_c_i64]*6
                       + [ctypes.c_int32, _c_ptr]),
    'brv_cconv_wgrad_workspace_bytes': (_c_i64, [_c_i64]*4),
    'brv_cconv_wgrad': (ctypes.c_int, [_c_ptr]*5 + [_c_i64]*6 + [_c_ptr]),
    'brv_cconv_rows_bf16': (ctypes.c_int, [_c_ptr, _c_ptr, _c_i64, _c_ptr, _c_ptr, _c_ptr, _c_ptr] + [_c_i64]*6
                            + [ctypes.c_int32, _c_ptr]),
    'brv_cconv_wgrad_bf16': (ctypes.c_int, [_c_ptr]*5 + [_c_i64]*6 + [_c_ptr]),
    'brv_cconv_rows_ex': (ctypes.c_int, [_c_ptr, _c_ptr, _c_i64, _c_ptr, _c_ptr, _c_ptr, _c_ptr] + [_c_i64]*6
                          + [ctypes.c_int32]*3 + [_c_ptr]),
    'brv_batchnorm2d_forward_bf16io': (ctypes.c_int, [_c_ptr]*9 + [_c_i64]*3
                                       + [_c_f32, _c_f32, ctypes.c_int, _c_ptr]),
    'brv_batchnorm2d_backward_bf16io': (ctypes.c_int, [_c_ptr]*12 + [_c_i64, _c_i64, _c_i64, _c_ptr]),
    'brv_complex_mix_forward': (ctypes.c_int, [_c_ptr, _c_ptr, _c_ptr, _c_i64, _c_ptr]),
    'brv_complex_mix_backward': (ctypes.c_int, [_c_ptr, _c_ptr, _c_ptr, _c_i64, _c_ptr]),
    'brv_batchnorm2d_backward_ex': (ctypes.c_int, [_c_ptr, ctypes.c_int32, _c_ptr, _c_ptr, ctypes.c_int32] + [_c_ptr]*6
                                    + [ctypes.c_int32] + [_c_ptr]*4 + [_c_i64, _c_i64, _c_i64, _c_ptr]),
    'brv_batchnorm2d_forward_bf16': (ctypes.c_int, [_c_ptr]*9 + [_c_i64]*3
                                     + [_c_f32, _c_f32, ctypes.c_int, _c_ptr]),
    'brv_batchnorm2d_backward_bf16': (ctypes.c_int, [_c_ptr]*12 + [_c_i64, _c_i64, _c_i64, _c_ptr]),
    'brv_conv_nhwc_packed_size': (_c_i64, [_c_i64, _c_i64, _c_i64]),
    'brv_conv_nhwc_pack': (ctypes.c_int, [_c_ptr, _c_ptr, _c_i64, _c_i64, _c_i64, _c_ptr]),
    'brv_conv_nhwc_forward': (ctypes.c_int, [_c_ptr, _c_i64, _c_i64, _c_ptr, _c_i64, _c_i64, _c_ptr,
                                             _c_ptr, _c_ptr, _c_i64, _c_ptr, _c_ptr, ctypes.c_int,
                                             _c_ptr] + [_c_i64]*6 + [_c_f32, _c_ptr, _c_ptr]),
    'brv_conv_nhwc_forward_gn': (ctypes.c_int, [_c_ptr, _c_i64, _c_i64, _c_ptr, _c_i64, _c_i64, _c_ptr,
                                                _c_ptr, _c_ptr, _c_i64] + [_c_ptr]*7 + [_c_i64, _c_f32, _c_ptr,
                                                ctypes.c_int, _c_ptr] + [_c_i64]*6 + [_c_f32, _c_ptr, _c_ptr]),
    'brv_conv_nhwc_split_ws_bytes': (_c_i64, [_c_i64]*6),
    'brv_conv_nhwc_forward_ws': (ctypes.c_int, [_c_ptr, _c_i64, _c_i64, _c_ptr, _c_i64, _c_i64, _c_ptr,
                                                _c_ptr, _c_ptr, _c_i64, _c_ptr, _c_ptr, ctypes.c_int,
                                                _c_ptr] + [_c_i64]*6 + [_c_f32, _c_ptr, _c_ptr, _c_i64, _c_ptr]),
    'brv_conv_nhwc_forward_gn_ws': (ctypes.c_int, [_c_ptr, _c_i64, _c_i64, _c_ptr, _c_i64, _c_i64, _c_ptr,
                                                   _c_ptr, _c_ptr, _c_i64] + [_c_ptr]*7 + [_c_i64, _c_f32, _c_ptr,
                                                   ctypes.c_int, _c_ptr] + [_c_i64]*6
                                    + [_c_f32, _c_ptr, _c_ptr, _c_i64, _c_ptr]),
    'brv_groupnorm_fold_chan2': (ctypes.c_int, [_c_ptr, _c_i64, _c_ptr, _c_i64] + [_c_ptr]*7
                                 + [_c_i64]*3 + [_c_f32, _c_ptr]),
    'brv_nchw_to_nhwc_f16': (ctypes.c_int, [_c_ptr, _c_ptr] + [_c_i64]*4 + [_c_ptr]),
    'brv_nhwc_f16_to_nchw': (ctypes.c_int, [_c_ptr, _c_ptr] + [_c_i64]*4 + [_c_ptr]),
    'brv_nhwc_chan_stats': (ctypes.c_int, [_c_ptr, _c_ptr] + [_c_i64]*6 + [_c_ptr]),
    'brv_groupnorm_fold_chan': (ctypes.c_int, [_c_ptr]*8 + [_c_i64]*4 + [_c_f32, _c_ptr]),
    'brv_nhwc_affine_act': (ctypes.c_int, [_c_ptr]*4 + [_c_i64]*4 + [ctypes.c_int, _c_ptr]),
    'brv_nhwc_fir_resample2d': (ctypes.c_int, [_c_ptr]*3 + [_c_i64]*9 + [ctypes.c_int, _c_f32, _c_ptr]),
    'brv_nhwc_fir_resample2d_dual': (ctypes.c_int, [_c_ptr]*3 + [ctypes.c_int] + [_c_ptr]*3 + [_c_i64]*10
                                     + [ctypes.c_int, _c_f32, _c_ptr]),
    'brv_nhwc_axpby': (ctypes.c_int, [_c_ptr, _c_f32, _c_ptr, _c_f32, _c_ptr, _c_i64, _c_ptr]),
    'brv_nhwc_conv1x1_packed_size': (_c_i64, [_c_i64, _c_i64, _c_i64]),
    'brv_nhwc_conv1x1_pack': (ctypes.c_int, [_c_ptr, _c_ptr, _c_i64, _c_i64, _c_i64, _c_ptr]),
    'brv_nhwc_conv1x1_forward': (ctypes.c_int, [_c_ptr, _c_i64, _c_i64, _c_ptr, _c_i64, _c_i64,
                                                _c_ptr, _c_ptr, _c_ptr] + [_c_i64]*3 + [_c_f32, _c_ptr]),
    'brv_nhwc_conv3x3_small_pack': (ctypes.c_int, [_c_ptr, _c_ptr, _c_i64, _c_i64, _c_ptr]),
    'brv_nhwc_conv3x3_small': (ctypes.c_int, [_c_ptr]*5 + [ctypes.c_int, _c_ptr, _c_ptr] + [_c_i64]*6
                               + [_c_ptr]),
    'brv_nhwc_add_pointwise': (ctypes.c_int, [_c_ptr]*5 + [_c_i64]*5 + [_c_f32, _c_ptr]),
    'brv_si_scale_forward': (ctypes.c_int, [_c_ptr]*5 + [_c_i64]*3 + [_c_f32, _c_ptr]),
    'brv_si_scale_backward': (ctypes.c_int, [_c_ptr]*6 + [_c_i64]*3 + [_c_ptr]),
    'brv_ema_update': (ctypes.c_int, [_c_ptr, _c_ptr, _c_f32, _c_i64, _c_ptr]),
    'brv_clip_adam_step2': (ctypes.c_int, [_c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_ptr, _c_i64] + [_c_f32]*6
                            + [_c_i64, _c_ptr, ctypes.c_int32, _c_ptr, _c_ptr]),
    'brv_memset_zero': (ctypes.c_int, [_c_ptr, _c_i64, _c_ptr]),
    'brv_mean_f32': (ctypes.c_int, [_c_ptr, _c_i64, _c_ptr, _c_ptr]),
    'brv_clip_adam_step': (ctypes.c_int, [_c_ptr, _c_ptr, _c_ptr, _c_ptr,
                                          _c_i64, _c_f32, _c_f32, _c_f32,
                                          _c_f32, _c_f32, _c_f32, _c_i64,
                                          _c_ptr, _c_ptr, _c_ptr]),
}

_lib = None


def lib():
    """Load the shared library once; fail loudly if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f'{LIB_PATH} is missing: build it with '
                '`python -c "import __graft_entry__ as g; g.build()"` or '
                '`make -C brever_amd/csrc` (needs hipcc, targets gfx950). '
                'brever_amd has no CPU/PyTorch fallback for its kernels.'
            )
        handle = ctypes.CDLL(LIB_PATH)
        for name, (restype, argtypes) in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = restype
            fn.argtypes = argtypes
        _lib = handle
    return _lib


def check(status, what):
    if status != 0:
        msg = lib().brv_last_error()
        raise RuntimeError(f'{what} failed with status {status}: '
                           f'{msg.decode() if msg else ""}')


def require_device(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError(
                'brever_amd kernels run on a ROCm device only; got a '
                f'{t.device} tensor (there is no CPU fallback)'
            )


def ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def gemm_f32(a, b, d, batch, M, N, K, lda, ldb, ldd, a_bs, b_bs, d_bs, trans_a, trans_b, kbatch, a_kbs, b_kbs,
             bias, mode):
    """``brv_gemm_f32``; a product whose output has too few tiles to fill the chip (weight gradients summed over the
    batch, convolutions at the low resolutions) goes through ``brv_gemm_f32_ws`` with scratch from the caching
    allocator: ordered reduction split, split-bf16 form where the layout allows (csrc/gemm_f32_big.hip)."""
    if M >= 32 and N >= 32:
        nbytes = lib().brv_gemm_f32_workspace_bytes(batch, M, N, K, trans_a, trans_b, kbatch)
        if nbytes > 0:
            ws = torch.empty(nbytes//4, dtype=torch.float32, device=d.device)
            check(lib().brv_gemm_f32_ws(
                ptr(a), ptr(b), ptr(d), batch, M, N, K, lda, ldb, ldd, a_bs, b_bs, d_bs, trans_a, trans_b,
                kbatch, a_kbs, b_kbs, ptr(bias), mode, ptr(ws), nbytes, stream()), 'brv_gemm_f32_ws')
            return
    check(lib().brv_gemm_f32(
        ptr(a), ptr(b), ptr(d), batch, M, N, K, lda, ldb, ldd, a_bs, b_bs, d_bs, trans_a, trans_b,
        kbatch, a_kbs, b_kbs, ptr(bias), mode, stream()), 'brv_gemm_f32')


def prof_enable(mode):
    """0: off; 1: event-time every launch of the Conv-TasNet calls made through ``launch_opts``;
    2: the same with the depthwise backward kernels labelled per dilation."""
    global _prof
    if _prof is not None:
        lib().brv_prof_destroy(_prof)
        _prof = None
    if mode:
        _prof = ctypes.c_void_p(lib().brv_prof_create(1 if mode == 2 else 0))


def profile_collect():
    """Per-label aggregate of the event-timed launches since ``prof_enable`` / the last collect:
    ``{label: dict(calls, ms, flops, bytes)}``."""
    if _prof is None:
        return {}
    n = lib().brv_prof_collect(_prof, None, 0)
    buf = ctypes.create_string_buffer(int(n) + 16)
    lib().brv_prof_collect(_prof, buf, len(buf))
    out = {}
    for line in buf.value.decode().splitlines():
        label, calls, ms, flops, nbytes = line.split()
        out[label] = dict(calls=int(calls), ms=float(ms), flops=float(flops),
                          bytes=float(nbytes))
    return out
