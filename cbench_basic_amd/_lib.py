"""ctypes loader of libbasic_hip.so (the C ABI declared in include/basic_hip.h).

There is NO fallback: if the shared library is missing, or no HIP device is usable when a
compute entry point is called, the caller gets an exception.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libbasic_hip.so")

OK, ERR_INVALID, ERR_NOT_INIT, ERR_HIP, ERR_OVERFLOW, ERR_NO_DEVICE = 0, -1, -2, -3, -4, -5


class BasicHipError(RuntimeError):
    pass


_lib = None

c_i32p = ctypes.c_void_p  # all array arguments are passed as raw addresses
_I, _L, _P, _F = ctypes.c_int, ctypes.c_int64, ctypes.c_void_p, ctypes.c_float

_SIGNATURES = {
    "basic_last_error": (ctypes.c_char_p, []),
    "basic_device_count": (_I, [_P]),
    "basic_set_device": (_I, [_I]),
    "basic_stream_synchronize": (_I, [_P]),
    "basic_pmf_to_quantized_cdf": (_I, [_P, _I, _I, _P]),
    "basic_rans_tables_from_freqs": (_I, [_P, _I, _I, _P, _P, _I, _I, _I, _P]),
    "basic_rans_tables_from_cdfs": (_I, [_P, _I, _I, _P, _P, _I, _I, _I, _P]),
    "basic_rans_tables_set_ar": (_I, [_P, _P, _I, _I, _I, _I]),
    "basic_rans_tables_set_ar_ops": (_I, [_P, _P, _I]),
    "basic_rans_encode_host_ex": (_I, [_P, _P, _P, _L, _P, _P, _P, _P, _P, _L, _P]),
    "basic_rans_encode_host_rows": (_I, [_P, _P, _P, _L, _P, _L, _P]),
    "basic_rans_decode_host_ex": (_I, [_P, _P, _L, _P, _L, _P, _P, _P, _P, _P]),
    "basic_rans_tables_info": (_I, [_P, _P, _P]),
    "basic_rans_tables_get_cdfs": (_I, [_P, _P, _I]),
    "basic_rans_tables_destroy": (None, [_P]),
    "basic_rans_encode_host": (_I, [_P, _P, _P, _L, _P, _P, _P, _P, _L, _P]),
    "basic_rans_encode_bound": (_L, [_L]),
    "basic_frame_streams": (_I, [_P, _P, _I, ctypes.c_uint32, ctypes.c_uint32, _P, _L, _P]),
    "basic_unframe_streams": (_I, [_P, _L, _P, _P, _P, _P, _I, _P]),
    "basic_rans_decode_host": (_I, [_P, _P, _L, _P, _L, _P, _P, _P, _P]),
    "basic_rans_stream_open": (_I, [_P, _P, _L, _P]),
    "basic_rans_stream_decode": (_I, [_P, _P, _L, _P]),
    "basic_rans_stream_close": (None, [_P]),
    "basic_rans_encode_batch_dev": (_I, [_P, _P, _P, _P, _I, _P, _L, _P, _P]),
    "basic_rans_compact_streams_dev": (_I, [_P, _L, _P, _P, _I, _P, _P]),
    "basic_rans_decode_batch_dev": (_I, [_P, _P, _P, _P, _P, _I, _P, _P, _P, _P]),
    "basic_rans_decode_batch_strided_dev": (_I, [_P, _P, _P, _P, _L, _L, _L, _I, _P, _P, _P, _P]),
    "basic_gc_quantize_index_dev": (_I, [_P, _P, _L, _P, _I, _F, _P, _P, _P, _P]),
    "basic_eb_quantize_index_dev": (_I, [_P, _P, _I, _I, _I, _P, _P, _P, _P]),
    "basic_eb_dequantize_dev": (_I, [_P, _P, _I, _I, _I, _P, _P]),
    "basic_i32_to_f32_dev": (_I, [_P, _L, _P, _P]),
    "basic_pgm_gauss_encode_group_dev": (_I, [_P, _P, _I, _I, _I, _P, _L, _P, _I, _P, _P, _L, _L, _P, _P]),
    "basic_pgm_gauss_index_group_dev": (_I, [_P, _I, _I, _I, _P, _L, _P, _I, _P, _L, _L, _P]),
    "basic_pgm_gauss_scatter_group_dev": (_I, [_P, _P, _I, _I, _I, _P, _L, _L, _L, _P, _P]),
    "basic_gauss_nll_per_image_dev": (_I, [_P, _P, _I, _I, _I, _I, _F, _F, _P, _P]),
    "basic_eb_nll_per_image_dev": (_I, [_P, _P, _I, _I, _I, _F, _P, _P]),
    "basic_conv_plan_create": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _I, _I, _P]),
    "basic_conv_plan_out_hw": (_I, [_P, _I, _I, _P, _P]),
    "basic_conv_forward_dev": (_I, [_P, _P, _I, _I, _I, _P, _P]),
    "basic_conv_plan_channels": (_I, [_P, _P, _P]),
    "basic_hp_session_create": (_I, [_P, _I, _P, _I, _P, _I, _P, _I, _P, _I, _P, _P, _I, _F, _P, _P]),
    "basic_hp_encode_bound": (_L, [_P, _I, _I, _I]),
    "basic_hp_encode_images": (_I, [_P, _P, _I, _I, _I, _I, _P, _L, _P, _P]),
    "basic_hp_encode_result": (_I, [_P, _P, _L, _P]),
    "basic_hp_decoded_shape": (_I, [_P, _P, _L, _P, _P, _P, _P]),
    "basic_hp_decode_images": (_I, [_P, _P, _L, _P, _L, _P]),
    "basic_hp_session_set_rans_waves": (_I, [_P, _I]),
    "basic_hp_session_set_transform_token": (_I, [_P, _I]),
    "basic_hp_session_destroy": (None, [_P]),
    "basic_conv_plan_destroy": (None, [_P]),
    "basic_conv_plan_flops": (_L, [_P, _I, _I, _I]),
    "basic_conv_plan_launches": (_I, [_P, _I, _I, _I]),
    "basic_mconv_plan_create": (_I, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "basic_mconv_forward_pos_dev": (_I, [_P, _P, _P, _P, _I, _I, _I, _P, _L, _P, _I, _I, _P]),
    "basic_mconv_forward_step_dev": (_I, [_P, _P, _P, _P, _I, _I, _I, _P, _L, _P, _I, _I, _I, _P, _P]),
    "basic_mconv_forward_ex_dev": (_I, [_P, _P, _P, _P, _I, _I, _I, _P, _L, _P, _I, _I, _I, _I, _P, _P, _P, _P]),
    "basic_mconv_plan_destroy": (None, [_P]),
    "basic_scanline_plan_create": (_I, [_P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P]),
    "basic_scanline_plan_info": (_I, [_P, _P, _P]),
    "basic_scanline_encode_dev": (_I, [_P, _P, _P, _I, _I, _I, _P, _I, _P, _P, _P, _P]),
    "basic_scanline_decode_dev": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _P, _I, _P, _P, _P, _P]),
    "basic_scanline_can_decode": (_I, [_P, _P, _I, _P]),
    "basic_scanline_batched_max": (_I, [_P, _I, _I, _P]),
    "basic_scanline_status": (_I, [_P, _P, _P]),
    "basic_scanline_plan_destroy": (None, [_P]),
    "basic_mse_per_image_dev": (_I, [_P, _P, _I, _L, _P, _P]),
    "basic_tans_tables_create": (_I, [_P, _I, _I, _P, _P, _I, _I, _I, _I, _P]),
    "basic_tans_tables_set_ar": (_I, [_P, _P, _I, _I, _I, _I]),
    "basic_tans_tables_get_row": (_I, [_P, _I, _P, _P, _P, _P]),
    "basic_tans_tables_destroy": (None, [_P]),
    "basic_tans_encode_host": (_I, [_P, _P, _P, _L, _P, _P, _P, _L, _P, _L, _P, _P]),
    "basic_tans_decode_host": (_I, [_P, _P, _L, _P, _L, _P, _P, _P, _P]),
    "basic_tans_encode_bound_words": (_L, [_P, _L]),
    "basic_tans_encode_batch_dev": (_I, [_P, _P, _P, _P, _I, _P, _L, _P, _P]),
    "basic_tans_decode_batch_dev": (_I, [_P, _P, _P, _P, _P, _I, _P, _P, _P]),
}


def lib():
    """Load libbasic_hip.so; raise if it has not been built (no silent fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise BasicHipError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(make -C cbench_basic_amd/csrc). cbench_basic_amd has no CPU fallback.")
        # One HIP runtime per process: PyTorch ships its own libamdhip64 and the host mirror needs torch for device
        # memory anyway, so torch is loaded FIRST and libbasic_hip.so binds to that copy.  (With the system runtime
        # loaded first, torch's later initialisation finds "No HIP GPUs".)
        import torch  # noqa: F401
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def last_error():
    return (lib().basic_last_error() or b"").decode("utf-8", "replace")


def check(rc):
    """Map a C-ABI status to the exception the reference's pybind11 layer would raise."""
    if rc == OK:
        return
    msg = last_error()
    if rc in (ERR_INVALID, ERR_NOT_INIT):
        raise ValueError(msg)  # py::value_error in the reference
    if rc == ERR_NO_DEVICE:
        raise BasicHipError("no MI355X/HIP device available: " + msg)
    if rc == ERR_OVERFLOW:
        raise BasicHipError("buffer overflow: " + msg)
    raise BasicHipError(msg or f"libbasic_hip status {rc}")


def ptr(t):
    """Device/host address of a torch tensor or numpy array (None -> NULL)."""
    if t is None:
        return None
    if hasattr(t, "data_ptr"):
        return t.data_ptr()
    return t.ctypes.data


def current_stream_ptr():
    import torch
    return torch.cuda.current_stream().cuda_stream
