"""ctypes binding of libwm_hip.so (C ABI declared in include/wm_kernels.h).

The library is the product: there is NO fallback.  If it is missing or a kernel reports an error the
call raises, so a GPU run can never silently route through PyTorch eager or the CPU oracle.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libwm_hip.so")

c_f = C.c_void_p          # device float*
c_i32, c_i64, c_u64, c_fl = C.c_int32, C.c_int64, C.c_uint64, C.c_float


class WmReduceJob(C.Structure):
    _fields_ = [("src", c_f), ("pstride", c_i64), ("src_ld", c_i64), ("count", c_i32), ("rows", c_i32), ("cols", c_i32),
                ("reserved", c_i32), ("dst", c_f), ("dst_rs", c_i64), ("dst_cs", c_i64)]


class WmGemm(C.Structure):
    _fields_ = [("A", c_f), ("B", c_f), ("C", c_f),
                ("M", c_i32), ("N", c_i32), ("K", c_i32),
                ("lda", c_i64), ("ldb", c_i64), ("ldc", c_i64),
                ("a_kc", c_i32), ("b_kc", c_i32), ("a_op", c_i32), ("b_op", c_i32),
                ("bias", c_f), ("res", c_f), ("ldr", c_i64),
                ("rowscale", c_f), ("rows_per_scale", c_i32),
                ("kscale", c_f), ("k_per_scale", c_i32),
                ("gaux", c_f), ("ldg", c_i64),
                ("b_ones", c_i32), ("C2", c_f),
                ("atomic_out", c_i32), ("ksplit", c_i32), ("mode", c_i32),
                ("a_bf16", c_i32), ("b_bf16", c_i32), ("c_bf16", c_i32), ("gaux_bf16", c_i32),
                ("gaux_raw", c_i32), ("scratch", c_f), ("scratch_floats", c_i64), ("defer", C.POINTER(WmReduceJob)),
                ("ln_gamma", c_f), ("ln_beta", c_f), ("ln_mod", c_f), ("ln_out", c_f), ("ln_mean", c_f), ("ln_rstd", c_f),
                ("ln_hw", c_i32), ("ln_shift", c_i32),
                ("a_grid_n", c_i32), ("a_jump_k", c_i32), ("a_jump", c_i64), ("b_seg_cols", c_i32), ("b_seg_rows", c_i32 * 4)]


class WmConv(C.Structure):
    _fields_ = [(n, c_i32) for n in ("B", "Cin", "H", "W", "Cout", "KH", "KW", "SH", "SW", "PH", "PW", "OH", "OW",
                                      "in_nhwc", "out_nhwc", "act", "transposed")]


class WmAttack(C.Structure):
    _fields_ = [("flags", c_i32), ("snr_db", c_fl), ("amp", c_fl), ("lp_h", c_f), ("lp_n", c_i32),
                ("rs_h", c_f), ("rs_n", c_i32), ("seed", c_u64), ("step", c_f),
                ("echo_delay", c_i32), ("echo_gain", c_fl), ("jitter_count", c_i32), ("reserved", c_i32), ("lp_mfma", c_f)]


class WmWaveAttack(C.Structure):
    _fields_ = [("flags", c_i32), ("snr_db", c_fl), ("amp", c_fl), ("echo_gain", c_fl), ("echo_delay", c_i64),
                ("jitter_count", c_i32), ("reserved", c_i32), ("seed", c_u64)]


_SIGS = {
    "wm_abi_version": [],
    "wm_gemm_scratch_floats": [],
    "wm_gemm": [C.POINTER(WmGemm), c_f],
    "wm_gemm_ln_fusable": [c_i32, c_i32, c_i32, c_i32, c_i32],
    "wm_last_gemm_family": [],
    "wm_layernorm_fwd": [c_f, c_f, c_f, c_f, c_f, c_i32, c_f, c_f, c_i64, c_i32, c_i32, c_i32, c_f],
    "wm_layernorm_bwd_scratch_floats": [c_i32],
    "wm_reduce_jobs": [C.POINTER(WmReduceJob), c_i32, c_f],
    "wm_layernorm_bwd": [c_f, c_i32] + [c_f] * 11 + [C.POINTER(WmReduceJob), c_i64, c_i32, c_i32, c_i32, c_f, c_i32, c_f],
    "wm_winattn_fwd": [c_f, c_f, c_f, c_f, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_f],
    "wm_winattn_bwd_scratch_floats": [],
    "wm_winattn_bwd": [c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, C.POINTER(WmReduceJob), c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_f],
    "wm_dwconv_fwd": [c_f, c_f, c_f, c_f, c_f, c_f, c_i32, c_i32, c_i32, c_i32, c_f],
    "wm_dwconv_bwd_scratch_floats": [c_i32, c_i32, c_i32],
    "wm_dwconv_bwd": [c_f, c_f, c_f, c_f, c_f, c_f, c_f, C.POINTER(WmReduceJob), c_i32, c_i32, c_i32, c_i32, c_f],
    "wm_attn_half_fusable": [c_i32, c_i32],
    "wm_attn_half_fwd": [c_f] * 22 + [c_i32, c_i32, c_i32, c_i32, c_f],
    "wm_attn_half_bwd_scratch_floats": [c_i32, c_i32, c_i32, c_i32],
    "wm_attn_half_bwd": [c_f] * 29 + [C.POINTER(WmReduceJob), c_i32, c_i32, c_i32, c_i32, c_f],
    "wm_leff_fusable": [c_i32, c_i32],
    "wm_leff_fwd": [c_f] * 12 + [c_i32, c_i32, c_i32, c_f],
    "wm_leff_bwd_scratch_floats": [c_i32, c_i32, c_i32],
    "wm_leff_bwd": [c_f, c_i32] + [c_f] * 11 + [C.POINTER(WmReduceJob), c_i32, c_i32, c_i32, c_f],
    "wm_im2col_4x4s2": [c_f, c_f, c_i32, c_i32, c_i32, c_i32, c_f],
    "wm_col2im_4x4s2": [c_f, c_i32, c_f, c_i32, c_i32, c_i32, c_f],
    "wm_conv4x4s2_xg_rows": [c_i32, c_i32],
    "wm_conv4x4s2_dyg_rows": [c_i32, c_i32],
    "wm_conv4x4s2_grid_rows": [c_i32, c_i32],
    "wm_conv4x4s2_x_to_grid": [c_f, c_f, c_i32, c_i32, c_i32, c_f],
    "wm_conv4x4s2_dy_to_grid": [c_f, c_i32, c_f, c_i32, c_i32, c_i32, c_f],
    "wm_conv4x4s2_grid_to_dx": [c_f, c_f, c_i32, c_i32, c_i32, c_f],
    "wm_conv4x4s2_weights": [c_f, c_f, c_f, c_i32, c_f],
    "wm_conv4x4s2_dweight": [c_f, c_f, c_i32, c_f],
    "wm_pixel_shuffle2": [c_f, c_i32, c_f, c_f, c_i32, c_i32, c_i32, c_i64, c_i32, c_f],
    "wm_pixel_unshuffle2": [c_f, c_f, c_i32, c_i32, c_i32, c_i32, c_i64, c_i32, c_f],
    "wm_copy_cols": [c_f, c_i64, c_i32, c_f, c_i64, c_i32, c_i64, c_i32, c_i32, c_f],
    "wm_strided3": [c_f, c_f, c_i32, c_i32, c_i32, c_i32, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_i32, c_f],
    "wm_cast_bf16": [c_f, c_f, c_i64, c_f],
    "wm_sum_bf16_rows": [c_f, c_f, c_i32, c_i64, c_f],
    "wm_transpose_cast_jobs": [c_f, c_f, c_f, c_i32, c_i32, c_f],
    "wm_minmax_scratch_floats": [],
    "wm_minmax": [c_f, c_i64, c_f, c_f, c_f],
    "wm_range_map": [c_f, c_f, c_f, c_fl, c_fl, c_i32, c_i64, c_f],
    "wm_msg_expand_fwd": [c_f, c_f, c_i32, c_f],
    "wm_msg_expand_bwd": [c_f, c_f, c_i32, c_f],
    "wm_conv_fwd": [C.POINTER(WmConv), c_f, c_f, c_f, c_f, c_f],
    "wm_conv_bwd": [C.POINTER(WmConv), c_f, c_f, c_f, c_f, c_f, c_i32, c_f, c_f, c_f],
    "wm_maxpool_fwd": [c_f, c_f, c_i32, c_i32, c_i32, c_i32, c_i32, c_f],
    "wm_maxpool_bwd": [c_f, c_f, c_f, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_f],
    "wm_stft256_tiles_per_utterance": [c_i64],
    "wm_stft256_tiles": [c_f, c_i64, c_i32, c_f, c_fl, c_f],
    "wm_stft255_frames": [c_i64],
    "wm_stft255_tiles_per_utterance": [c_i64],
    "wm_stft255_tiles": [c_f, c_i64, c_i32, c_f, c_fl, c_f],
    "wm_istft255_stitch": [c_f, c_i32, c_i32, c_i32, c_f, c_i64, c_fl, c_f],
    "wm_wave_attack": [C.POINTER(WmWaveAttack), c_f, c_f, c_i64, c_i32, c_f, c_f],
    "wm_stft255_table_bytes": [c_i32],
    "wm_stft255_tables": [c_f, c_i32, c_f],
    "wm_wave_lowpass": [c_f, c_f, c_f, c_i64, c_i32, c_f, c_i32, c_f],
    "wm_wave_resample": [c_f, c_f, c_f, c_i64, c_i32, c_f, c_i32, c_f],
    "wm_pv_frames": [c_i64],
    "wm_pv_steps": [c_i64, C.c_double],
    "wm_time_stretch": [c_f, c_i64, c_i32, C.c_double, c_f, c_f, c_f, c_f, c_i64, c_f],
    "wm_sinc_resample": [c_f, c_i64, c_i32, C.c_double, c_f, c_f, c_i32, c_i32, c_f, c_i64, c_f],
    "wm_istft255_fwd": [c_f, c_f, c_f, c_i32, c_i32, c_f],
    "wm_stft255_fwd": [c_f, c_f, c_f, c_i32, c_i32, c_f],
    "wm_istft255_bwd": [c_f, c_f, c_f, c_i32, c_i32, c_f],
    "wm_stft255_bwd": [c_f, c_f, c_f, c_i32, c_i32, c_f],
    "wm_attack_fir_table_bytes": [],
    "wm_attack_fir_tables": [c_f, c_i32, c_f, c_f],
    "wm_attack_fwd": [C.POINTER(WmAttack), c_f, c_f, c_f, c_i32, c_i32, c_f],
    "wm_attack_bwd": [C.POINTER(WmAttack), c_f, c_f, c_f, c_f, c_i32, c_i32, c_f],
    "wm_awgn_noise": [c_u64, c_f, c_f, c_i32, c_i32, c_f],
    "wm_loss_fwd_bwd": [c_f] * 11 + [c_i32, c_f, c_f],
    "wm_charbonnier_fwd_bwd": [c_f, c_f, c_i64, c_fl, c_f, c_f, c_f, c_fl, c_f],
    "wm_tvloss_fwd_bwd": [c_f, c_i32, c_i32, c_i32, c_i32, c_fl, c_f, c_f, c_fl, c_f],
    "wm_tv_pow_fwd_bwd": [c_f, c_i32, c_i32, c_i32, c_i32, c_fl, c_fl, c_f, c_f, c_fl, c_f],
    "wm_crop2d": [c_f, c_f] + [c_i32] * 8 + [c_f],
    "wm_blend2d": [c_f, c_f, c_f, c_f, c_f] + [c_i32] * 8 + [c_f],
    "wm_bernoulli_mask": [c_f, c_i32, c_fl, c_u64, c_u64, c_f],
    "wm_resize_nearest": [c_f, c_f, c_i32, c_i32, c_i32, c_i32, c_i32, c_fl, c_fl, c_i32, c_f],
    "wm_quantization_state_bytes": [],
    "wm_quantization_fwd": [c_f, c_f, c_f, c_i64, c_f, c_f],
    "wm_quantization_bwd": [c_f, c_f, c_f, c_f, c_i64, c_f, c_f],
    "wm_jpeg_mask": [c_f, c_f, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_f],
    "wm_adamw": [c_f, c_f, c_f, c_f, c_i64, c_f, c_fl, c_fl, c_fl, c_fl, c_f, c_f, c_f],
    "wm_step_inc": [c_f, c_f],
    "wm_step_copy": [c_f, c_f, c_f],
    "wm_droppath_scales": [c_f, c_f, c_i32, c_i32, c_u64, c_f, c_f],
    "wm_zero": [c_f, c_i64, c_f],
    "wm_axpbyz": [c_f, c_fl, c_f, c_fl, c_f, c_i64, c_f],
    "wm_batchnorm2d_fwd": [c_f] * 8 + [c_i32, c_i32, c_i32, c_fl, c_fl, c_i32, c_i32, c_fl, c_f],
    "wm_batchnorm2d_bwd": [c_f] * 10 + [c_i32, c_i32, c_i32, c_i32, c_i32, c_fl, c_f],
    "wm_dropout": [c_f, c_f, c_i64, c_fl, c_u64, c_u64, c_f],
    "wm_mse_fwd_bwd": [c_f, c_f, c_i64, c_f, c_f, c_f, c_fl, c_f],
    "wm_plane_fill": [c_f, c_f, c_i32, c_i32, c_i32, c_i32, c_i32, c_fl, c_f],
    "wm_plane_sum": [c_f, c_f, c_i32, c_i32, c_i32, c_i32, c_i32, c_fl, c_f],
    "wm_bce_logits_fwd_bwd": [c_f, c_i32, c_fl, c_f, c_f, c_fl, c_f],
    "wm_linear_small_fwd": [c_f, c_f, c_f, c_f, c_i32, c_i32, c_i32, c_f],
    "wm_linear_small_bwd": [c_f, c_f, c_f, c_f, c_f, c_f, c_i32, c_i32, c_i32, c_f],
}
EXPORTS = tuple(_SIGS)


class WmKernelError(RuntimeError):
    code = None                                   # the C entry point's return value (22 = WM_EINVAL), when there was one


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build the HIP kernels first "
            "(python -c 'import __graft_entry__ as g; g.build()'  or  make -C image-in-speech-watermarking_amd/csrc). "
            "There is no PyTorch/CPU fallback for the watermark hot path.")
    lib = C.CDLL(LIB_PATH)
    for name, args in _SIGS.items():
        fn = getattr(lib, name)          # AttributeError here = ABI drift between header and library
        fn.argtypes = args
        fn.restype = C.c_int64 if name.endswith(("_floats", "_bytes", "_rows")) else C.c_int
    return lib


lib = _load()


def check(rc, what):
    if rc != 0:
        err = WmKernelError(f"{what} failed with code {rc}" + (" (unsupported shape/alignment)" if rc == 22 else ""))
        err.code = int(rc)
        raise err


def ptr(t):
    """Device pointer of a contiguous fp32/bf16/int64 CUDA tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise WmKernelError("libwm_hip kernels take device tensors only (got a CPU tensor)")
    return t.data_ptr()
