"""ctypes binding of libupsparts_hip.so (the C ABI declared in include/upsparts_hip.h).

The library is built in-tree by ``__graft_entry__.build()`` / ``csrc/build.sh``.  There is NO CPU or
PyTorch fallback: if the shared object is missing or a call returns non-zero this module raises.
torch is used only to own device memory and streams; every entry point receives raw device pointers.
"""
import ctypes as C
import os

import torch

from . import switches as SW

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = SW.value("UPS_LIB") or os.path.join(_HERE, "csrc", "libupsparts_hip.so")   # UPS_LIB: A/B builds

ABI_VERSION = 4               # include/upsparts_hip.h UPS_ABI_VERSION
F32, BF16, F16 = 0, 1, 2      # F16: forward tensors of precision-critical scopes (held in torch.bfloat16 containers, see ops.py)
ACT_NONE, ACT_LRELU, ACT_RELU, ACT_ELU = 0, 1, 2, 3
ACT = {None: ACT_NONE, "leaky_relu": ACT_LRELU, "relu": ACT_RELU, "elu": ACT_ELU}


class UpsError(RuntimeError):
    pass


class ConvDesc(C.Structure):
    _fields_ = [("dtype", C.c_int32), ("n", C.c_int32), ("hi", C.c_int32), ("wi", C.c_int32), ("ci", C.c_int32),
                ("ldi", C.c_int32), ("ho", C.c_int32), ("wo", C.c_int32), ("co", C.c_int32), ("co_fill", C.c_int32),
                ("ldo", C.c_int32), ("out_h", C.c_int32), ("out_w", C.c_int32), ("out_sy", C.c_int32),
                ("out_sx", C.c_int32), ("out_oy", C.c_int32), ("out_ox", C.c_int32), ("in_sy", C.c_int32),
                ("in_sx", C.c_int32), ("ntaps", C.c_int32), ("kh", C.c_int32), ("kw", C.c_int32),
                ("tap_dy", C.c_int32 * 9), ("tap_dx", C.c_int32 * 9), ("tap_w", C.c_int32 * 9),
                ("act_in", C.c_int32), ("act_slope", C.c_float), ("out_f32", C.c_int32), ("dact_kind", C.c_int32),
                ("ldr", C.c_int32), ("ldd", C.c_int32),
                ("in_", C.c_void_p), ("w", C.c_void_p), ("out", C.c_void_p), ("bias", C.c_void_p),
                ("coord_tab", C.c_void_p), ("res", C.c_void_p), ("dact", C.c_void_p),
                ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t),
                ("mask_bits", C.c_void_p), ("mask_batch", C.c_int32), ("mask_grad", C.c_void_p), ("mask_view", C.c_void_p),
                ("f8_deq", C.c_void_p), ("f8_scale", C.c_void_p), ("f8_amax", C.c_void_p), ("f8_e5m2", C.c_int32),
                ("in_f8", C.c_void_p), ("out_f8", C.c_void_p), ("out_f8_scale", C.c_void_p), ("out_f8_amax", C.c_void_p),
                ("out_f8_act", C.c_int32), ("out_f8_e5m2", C.c_int32), ("d2s", C.c_int32),
                ("out_act", C.c_int32), ("res_act", C.c_int32), ("sign_out", C.c_void_p), ("dact_bits", C.c_void_p)]


class WgradDesc(C.Structure):
    _fields_ = [("dtype", C.c_int32), ("n", C.c_int32), ("hi", C.c_int32), ("wi", C.c_int32), ("ci", C.c_int32),
                ("ldi", C.c_int32), ("ci_log", C.c_int32), ("cin_v", C.c_int32), ("ho", C.c_int32), ("wo", C.c_int32),
                ("co", C.c_int32), ("ldo", C.c_int32), ("in_sy", C.c_int32), ("in_sx", C.c_int32), ("ntaps", C.c_int32),
                ("tap_dy", C.c_int32 * 9), ("tap_dx", C.c_int32 * 9), ("tap_w", C.c_int32 * 9),
                ("act_in", C.c_int32), ("act_slope", C.c_float), ("splitk", C.c_int32),
                ("in_", C.c_void_p), ("dout", C.c_void_p), ("grad", C.c_void_p), ("grad_bias", C.c_void_p),
                ("workspace", C.c_void_p), ("mask_bits", C.c_void_p), ("mask_batch", C.c_int32), ("in_f16", C.c_int32),
                ("dout_f8", C.c_void_p), ("dout_f8_scale", C.c_void_p), ("in_f8_scale", C.c_void_p), ("in_f8_amax", C.c_void_p)]


class PriorDesc(C.Structure):
    _fields_ = [("n", C.c_int32), ("h", C.c_int32), ("w", C.c_int32), ("P", C.c_int32), ("view", C.c_int32),
                ("entropy_ce", C.c_int32), ("gamma", C.c_float), ("half_h", C.c_int32), ("half_w", C.c_int32),
                ("ms_alpha", C.c_float), ("ms_lambda", C.c_float),
                ("w_kl", C.c_float), ("w_entropy", C.c_float), ("w_ms", C.c_float), ("w_area", C.c_float),
                ("w_patch", C.c_float), ("w_gmrf", C.c_float), ("w_var", C.c_float),
                ("l", C.c_void_p), ("l_mean", C.c_void_p), ("m", C.c_void_p), ("hard", C.c_void_p), ("px", C.c_void_p),
                ("per_np", C.c_void_p), ("sums", C.c_void_p), ("g_hard", C.c_void_p), ("dl", C.c_void_p),
                ("variant", C.c_int32), ("w_ms_logits", C.c_float), ("dl_rec", C.c_void_p)]


class PrepItem(C.Structure):
    _fields_ = [("src", C.c_void_p), ("w_fwd", C.c_void_p), ("w_dgrad", C.c_void_p), ("ctab", C.c_void_p),
                ("ntaps", C.c_int32), ("cin_v", C.c_int32), ("ci_log", C.c_int32), ("co", C.c_int32),
                ("ci_pad", C.c_int32), ("dgrad_rows", C.c_int32), ("dgrad_k", C.c_int32),
                ("kh", C.c_int32), ("kw", C.c_int32), ("in_sy", C.c_int32), ("in_sx", C.c_int32),
                ("dy", C.c_int32 * 3), ("dx", C.c_int32 * 3), ("ax", C.c_float), ("ay", C.c_float)]


class TowerLayer(C.Structure):
    """ups_tower_layer: one 1x1 layer of one critic tower (ups_towers_fwd / ups_towers_bwd)."""
    _fields_ = [("w_fwd", C.c_void_p), ("w_dgrad", C.c_void_p), ("bias", C.c_void_p), ("grad_w", C.c_void_p), ("grad_b", C.c_void_p),
                ("k", C.c_int32), ("n", C.c_int32)]


_lib = None

_I, _L, _F, _P, _Z = C.c_int32, C.c_int64, C.c_float, C.c_void_p, C.c_size_t
_SIGS = {
    "ups_abi_version": ([], C.c_int),
    "ups_struct_sizes": ([C.POINTER(C.c_int64)], None),
    "ups_conv_igemm": ([C.POINTER(ConvDesc), _P], C.c_int),
    "ups_conv_wgrad_plan": ([C.POINTER(WgradDesc), C.POINTER(_I), C.POINTER(_Z)], C.c_int),
    "ups_conv_wgrad": ([C.POINTER(WgradDesc), _P], C.c_int),
    "ups_weight_prep": ([_P, _I, _I, _I, _I, _I, _P, _I, _P, _I, _I, _P], C.c_int),
    "ups_weight_prep_f8": ([_P, _I, _I, _I, _I, _I, _P, _P, _P], C.c_int),
    "ups_weight_prep_d2s": ([_P, _I, _I, _I, _I, _I, _I, _P, _P], C.c_int),
    "ups_prep_item_blocks": ([C.POINTER(PrepItem), _I], C.c_int64),
    "ups_weight_prep_batch": ([_P, _P, _I, _L, _I, _P], C.c_int),
    "ups_coord_table": ([_P, _I, _I, _I, _I, C.POINTER(_I), C.POINTER(_I), _I, _I, _F, _F, _P, _P], C.c_int),
    "ups_batch_sum": ([_P, _I, _I, _L, _I, _I, _P, _P], C.c_int),
    "ups_coord_wgrad": ([_P, _I, _I, _I, _I, _I, _I, _I, C.POINTER(_I), C.POINTER(_I), _I, _I, _F, _F, _I, _P, _P, _P, _P], C.c_int),
    "ups_col_sum": ([_P, _I, _L, _I, _I, _P, _P, _P], C.c_int),
    "ups_bilinear2x_fwd": ([_P, _P, _I, _I, _I, _I, _I, _P], C.c_int),
    "ups_bilinear2x_bwd": ([_P, _P, _I, _I, _I, _I, _I, _P], C.c_int),
    "ups_depth_to_space": ([_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P], C.c_int),
    "ups_nearest2x": ([_P, _P, _I, _I, _I, _I, _I, _I, _P], C.c_int),
    "ups_crop_fwd": ([_P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P], C.c_int),
    "ups_crop_bwd": ([_P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P], C.c_int),
    "ups_bilinear2x_fwd_act": ([_P, _P, _I, _I, _I, _I, _I, _I, _F, _P], C.c_int),
    "ups_bilinear2x_fwd_bits": ([_P, _P, _I, _I, _I, _I, _I, _I, _F, _P, _P], C.c_int),
    "ups_sign_pack": ([_P, _I, C.c_int64, _P, _P], C.c_int),
    "ups_conv_sign_out_written": ([], C.c_int),
    "ups_bilinear2x_fwd_f8": ([_P, _P, _I, _I, _I, _I, _P, _P, _P, _I, _F, _I, _P], C.c_int),
    "ups_bilinear2x_bwd_f8": ([_P, _P, _I, _I, _I, _I, _P, _P, _P, _I, _P], C.c_int),
    "ups_act_mean_fwd": ([_P, _P, _I, _I, _I, _I, _I, _F, _P], C.c_int),
    "ups_act_mean_bwd": ([_P, _P, _P, _I, _I, _I, _I, _I, _F, _P], C.c_int),
    "ups_elu_fwd": ([_P, _P, _I, C.c_int64, _P], C.c_int),
    "ups_elu_bwd": ([_P, _P, _P, _I, C.c_int64, _P], C.c_int),
    "ups_maxpool2_fwd": ([_P, _P, _I, _I, _I, _I, _I, _P], C.c_int),
    "ups_maxpool2_fwd_f8": ([_P, _P, _I, _I, _I, _I, _P, _P, _P, _I, _F, _P], C.c_int),
    "ups_maxpool2_bwd": ([_P, _P, _P, _I, _I, _I, _I, _I, _P], C.c_int),
    "ups_copy_channels": ([_P, _I, _P, _I, _I, _L, _I, _P], C.c_int),
    "ups_add_channels": ([_P, _I, _P, _I, _I, _L, _I, _P], C.c_int),
    "ups_vgg_preprocess_fwd": ([_P, _I, _I, _P, _I, _L, _P], C.c_int),
    "ups_vgg_preprocess_bwd": ([_P, _P, _I, _I, _L, _P], C.c_int),
    "ups_l1_fwd": ([_P, _P, _I, _L, _I, _I, _I, _P, _I, _P], C.c_int),
    "ups_l1_bwd": ([_P, _P, _P, _I, _L, _I, _I, _I, _P, _F, _P], C.c_int),
    "ups_sum_scale": ([_P, _I, _F, _P, _I, _P], C.c_int),
    "ups_part_softmax_fwd": ([_P, _P, _P, _P, _P, _P, _P, _L, _I, _P], C.c_int),
    "ups_part_softmax_moments_tile": ([_I], _I),
    "ups_part_softmax_moments_ints": ([_L, _I], _Z),
    "ups_part_softmax_moments_fwd": ([_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _P, _P, _P], C.c_int),
    "ups_spatial_moments": ([_P, _I, _I, _I, _I, _F, _P, _I, _I, _P, _P], C.c_int),
    "ups_spatial_moments_kl": ([_P, _I, _I, _I, _I, _F, _P, _I, _I, _P, _P, _P], C.c_int),
    "ups_spatial_moments_floats": ([_I, _I], _Z),
    "ups_moments_to_px": ([_P, _I, _I, _I, _P, _P], C.c_int),
    "ups_draw_rect": ([_P, _I, _I, _I, _I, _I, _I, _P, _P], C.c_int),
    "ups_mask_parts_fwd": ([_P, _P, _P, _I, _I, _L, _I, _P], C.c_int),
    "ups_mask_parts_bwd": ([_P, _P, _P, _I, _I, _L, _I, _P], C.c_int),
    "ups_unpool_fwd": ([_P, _P, _P, _I, _I, _L, _I, _I, _I, _P], C.c_int),
    "ups_unpool_bwd": ([_P, _P, _P, _P, _P, _I, _I, _L, _I, _I, _I, _P], C.c_int),
    "ups_unpool_bwd_floats": ([_I, _I, _I], _Z),
    "ups_prior_sums_floats": ([_I, _I], _Z),
    "ups_prior_fwd": ([C.POINTER(PriorDesc), _P], C.c_int),
    "ups_prior_bwd": ([C.POINTER(PriorDesc), _P], C.c_int),
    "ups_randn": ([_P, _L, C.c_uint64, C.c_uint64, _P], C.c_int),
    "ups_critic_head_fwd": ([_P, _P, _I, _I, _I, _I, _P, _P, _P], C.c_int),
    "ups_critic_head_bwd": ([_P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P], C.c_int),
    "ups_state_update": ([_P, _P, _P, _F, _F, _I, _F, _F, _I, _F, _F, _F, _F, _P], C.c_int),
    "ups_towers_fwd": ([C.POINTER(TowerLayer), _I, _I, C.POINTER(_P), C.POINTER(_I), C.POINTER(_P), _I, _F, _P], C.c_int),
    "ups_towers_bwd": ([C.POINTER(TowerLayer), _I, _I, C.POINTER(_P), C.POINTER(_I), C.POINTER(_P), C.POINTER(_P), C.POINTER(_P),
                        C.POINTER(_P), C.POINTER(_I), _I, _I, _F, _P], C.c_int),
    "ups_latent_fwd": ([_P, _P, C.POINTER(_F), _I, _I, _I, _P, _P, _P], C.c_int),
    "ups_latent_bwd": ([_P, _P, C.POINTER(_F), _P, _P, _F, _I, _I, _I, _P, _P], C.c_int),
    "ups_adam": ([_P, _P, _P, _P, _L, _F, _F, _F, _F, _F, _P], C.c_int),
    "ups_adam_dev": ([_P, _P, _P, _P, _L, _P, _F, _F, _F, _F, _P], C.c_int),
    "ups_tps_warp": ([_P, _P, _P, _P, _I, _I, _I, _I, _I, _P], C.c_int),
    "ups_gauss_hm": ([_P, _P, _P, _I, _I, _I, _I, _P], C.c_int),
    "ups_gauss_hm3": ([_P, _P, _P, _I, _I, _I, _I, _P], C.c_int),
    "ups_convert": ([_P, _I, _P, _I, _L, _P], C.c_int),
    "ups_pad_convert": ([_P, _I, _P, _I, _I, _L, _P], C.c_int),
}
EXPORTS = sorted(list(_SIGS) + ["ups_last_error"])


def load():
    """Load the HIP library (fails loudly when it has not been built)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise UpsError("libupsparts_hip.so not found at {} -- run __graft_entry__.build() "
                       "(there is no CPU fallback for the product path)".format(LIB_PATH))
    lib = C.CDLL(LIB_PATH)
    for name, (args, res) in _SIGS.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = res
    lib.ups_last_error.argtypes = []
    lib.ups_last_error.restype = C.c_char_p
    if lib.ups_abi_version() != ABI_VERSION:
        raise UpsError("{}: ABI version {} (this binding speaks {}): rebuild with __graft_entry__.build()".format(
            LIB_PATH, lib.ups_abi_version(), ABI_VERSION))
    sizes = (C.c_int64 * 4)()
    lib.ups_struct_sizes(sizes)
    mine = [C.sizeof(ConvDesc), C.sizeof(WgradDesc), C.sizeof(PriorDesc), C.sizeof(PrepItem)]
    if list(sizes) != mine:
        raise UpsError("{}: descriptor struct sizes {} differ from this binding's {} (stale library?)".format(LIB_PATH, list(sizes), mine))
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        raise UpsError("{} failed ({}): {}".format(what, rc, load().ups_last_error().decode()))


def raw_stream(device=None):
    """hipStream_t (as an int) of torch's current stream on `device` (default: the current device): one C call -- a training step asks
    ~1 100 times, and torch.cuda.current_stream()'s Python path (device-index parsing, a Stream object per call) was 3 ms of its
    19 ms of host work (tools/host_overhead.py profile)."""
    if device is None or isinstance(device, int):
        idx = device
    else:
        idx = (device if isinstance(device, torch.device) else torch.device(device)).index
    if idx is None:
        idx = torch._C._cuda_getDevice()
    return torch._C._cuda_getCurrentRawStream(idx)


def stream():
    return C.c_void_p(raw_stream())


def ptr(t):
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous(), "device-resident contiguous tensors only"
    return C.c_void_p(t.data_ptr())


def dt(t):
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    raise UpsError("unsupported dtype {}".format(t.dtype))


def torch_dtype(code):
    return torch.float32 if code == F32 else (torch.float16 if code == F16 else torch.bfloat16)


def call(name, *args):
    check(getattr(load(), name)(*args), name)
