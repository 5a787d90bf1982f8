"""ctypes loader for libdasac_hip.so.  PyTorch is only the allocator / stream provider: every
call hands raw `data_ptr()`s, sizes and the current HIP stream to the C ABI of include/dasac_hip.h."""
import ctypes as C
import os

import torch

LIB_PATH = os.environ.get("DASAC_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libdasac_hip.so")

_p, _i, _l, _f, _sz = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_size_t

# name -> (restype, argtypes); must list every symbol declared in include/dasac_hip.h
PROTOTYPES = {
    "dasac_version": (_i, []),
    "dasac_last_error": (C.c_char_p, []),
    "dasac_device_info": (_i, [C.POINTER(_i), C.POINTER(_i), C.c_char_p, _sz]),
    "dasac_reserved_cus": (_i, []),
    "dasac_set_reserved_cus": (_i, [_i]),
    "dasac_pseudo_labels_workspace": (_sz, [_i, _i, _l]),
    "dasac_pseudo_labels": (_i, [_p, _p, _p, _f, _f, _i, _i, _l, _p, _p, _p, _p, _sz, _p]),
    "dasac_conv_mpad": (_i, [_i]),
    "dasac_conv_kpad": (_i, [_i]),
    "dasac_conv_table": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p, _p]),
    "dasac_conv_pack": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _i, _p, _p]),
    "dasac_relu_bits_words": (_sz, [_i, _l]),
    "dasac_conv_gemm_bits_ok": (_i, [_i, _i]),
    "dasac_conv_gemm": (_i, [_p, _p, _p, _p] + [_i] * 12 + [_p, _p, _p, _p, _p, _i, _i, _i, _i, _p, _sz, _p]),
    "dasac_conv_gemm_x3": (_i, [_p, _p, _p, _p] + [_i] * 12 + [_p, _p, _p, _p, _p, _i, _i, _i, _i, _p, _sz, _p]),
    "dasac_conv_pack_x3": (_i, [_p, _i, _i, _p, _p]),
    "dasac_conv_gemm_workspace": (_sz, []),
    "dasac_conv_gemm_schedule": (_i, [_i, _i, _i, _i, _i]),
    "dasac_conv_gemm_tail_split": (_i, [_i, _i, _i, _i, _i]),
    "dasac_conv_wgrad_workspace": (_sz, [_i, _i, _i, _i, _i]),
    "dasac_conv_wgrad": (_i, [_p, _p, _p] + [_i] * 9 + [_p, _sz, _p]),
    "dasac_conv_wgrad_x3": (_i, [_p, _p, _p] + [_i] * 9 + [_p, _sz, _p]),
    "dasac_upsample_softmax": (_i, [_p, _i, _i, _i, _i, _i, _i, _p, _p, _p, _p, _p]),
    "dasac_infer_labels": (_i, [_p, _i, _i, _i, _i, _i, _i, _p, _p, _p, _p]),
    "dasac_upsample_bwd_workspace": (_sz, [_i, _i, _i]),
    "dasac_upsample_bwd": (_i, [_p, _i, _i, _i, _i, _i, _p, _p, _p, _sz, _p]),
    "dasac_ce_loss_workspace": (_sz, [_i, _i, _l]),
    "dasac_ce_loss": (_i, [_p, _p, _p, _p, _i, _i, _l, _i, _p, _p, _p, _p, _p, _sz, _p]),
    "dasac_ce_loss_bwd_low_workspace": (_sz, [_i, _i, _i, _i]),
    "dasac_ce_loss_bwd_low": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p, _p, _p, _sz, _p]),
    "dasac_warp_affine": (_i, [_p, _p, _i, _i, _i, _i, _p, _p]),
    "dasac_warp_pool": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _i, _f, _p, _p, _p, _p]),
    "dasac_warp_back": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _p, _p]),
    "dasac_class_state": (_i, [_p, _p, _i, _l, _i, _f, _f, _i, _f, _p, _p, _p]),
    "dasac_bn_fold": (_i, [_p, _p, _p, _p, _p, _f, _i, _p, _p, _p, _p]),
    "dasac_bn_param_grads": (_i, [_p, _i, _p, _p, _p, _p, _p, _i, _p, _p, _p, _p]),
    "dasac_channel_sums": (_i, [_p, _i, _i, _l, _p, _p]),
    "dasac_maxpool_fwd": (_i, [_p, _i, _i, _i, _i, _i, _i, _i, _i, _p, _p, _p]),
    "dasac_maxpool_bwd": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p, _p]),
    "dasac_ema_chunk_elems": (_i, []),
    "dasac_ema_update": (_i, [_p, _i, _p, _i, _f, _i, _p, _p, _p]),
    "dasac_sgd_step": (_i, [_p, _i, _p, _i, _p, _p, _i, _f, _i, _p]),
    "dasac_scale_planes": (_i, [_p, _p, _l, _l, _p, _p]),
    "dasac_add": (_i, [_p, _p, _p, _l, _p]),
    "dasac_relu_mask": (_i, [_p, _p, _p, _l, _p]),
    "dasac_bn_stats_workspace": (_sz, [_i, _i, _l]),
    "dasac_bn_stats": (_i, [_p, _i, _i, _l, _p, _p, _sz, _p]),
    "dasac_bn_train_finalize": (_i, [_p, C.c_double, _p, _p, _p, _p, _p, _p, _f, _f, _i, _p, _p, _p, _p, _p]),
    "dasac_bn_apply": (_i, [_p, _p, _p, _p, _i, _i, _i, _l, _p, _p]),
    "dasac_bn_bwd_reduce": (_i, [_p, _p, _p, _p, _i, _i, _l, _p, _p, _p, _p, _sz, _p]),
    "dasac_bn_train_finalize_tiles": (_i, [_p, _i, _i, C.c_double, _p, _p, _p, _p, _p, _f, _f, _i, _p, _p, _p, _p, _p]),
    "dasac_bn_tile_stats_reduce": (_i, [_p, _i, _i, _i, _p, _p]),
    "dasac_bn_train_apply_tiles": (_i, [_p, _p, _i, _i, C.c_double, _p, _p, _p, _p, _p, _f, _f, _p, _i, _i, _i, _l, _p, _p, _p, _p]),
    "dasac_bn_bwd_fused": (_i, [_p, _p, _p, _p, _p, C.c_double, _i, _i, _l, _p, _p, _p, _p, _sz, _p]),
    "dasac_conv_gemm_stats_ok": (_i, [_i, _i]),
    "dasac_conv_gemm_stats_tiles": (_i, [_i, _i, _i]),
    "dasac_conv_gemm_stats": (_i, [_p, _p, _p, _p] + [_i] * 12 + [_p, _p, _i, _i, _i, _i, _p, _sz, _p, _p]),
    "dasac_bn_bwd_apply": (_i, [_p, _p, _p, _p, _p, _p, C.c_double, _p, _i, _i, _l, _p, _p, _p, _p]),
    "dasac_conv_pack_expanded": (_i, [_p, _i, _i, _i, _i, _i, _i, _i, _p, _p]),
    "dasac_tap_gather": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _p, _i, _i, _i, _p, _p]),
    "dasac_tap_scatter": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p, _p]),
    "dasac_conv_wgrad_finish_expanded": (_i, [_p, _i, _i, _i, _i, _i, _p, _i, _i, _i, _i, _p]),
    "dasac_pack_chunk_elems": (_i, []),
    "dasac_bn_fold_multi": (_i, [_p, _p, _i, _p]),
    "dasac_conv_pack_multi": (_i, [_p, _p, _i, _p]),
    "dasac_label_pad_mask": (_i, [_p, _p, _l, _i, _i, _p]),
    "dasac_dropout_planes": (_i, [C.c_uint64, C.c_uint64, _f, _l, _p, _p]),
    "dasac_iou_counts": (_i, [_p, _p, _i, _i, _l, _i, _p, _p]),
    "dasac_make_views_table_ints": (_i, [_i, _i]),
    "dasac_make_views": (_i, [_p, _p, _p, _i, _i, _i, _p, _p, _p, _i, _p, _p, _p, _p]),
    "dasac_view_photometric_workspace": (_sz, [_i, _i, _i]),
    "dasac_view_photometric": (_i, [_p, _p, _i, _i, _i, _p, _p, _p, _i, _p, _p, _p, _sz, _p]),
    "dasac_conv_wgrad_dot_rows": (_i, [_i, _i]),
    "dasac_conv_wgrad_finish": (_i, [_p, _i, _i, _i, _i, _i, _p, _p, _p, _p, _p, _i, _i, _i, _p]),
}


class DasacError(RuntimeError):
    pass


_lib = None


def load():
    """Returns the loaded library; raises if it was not built (run `python __graft_entry__.py`)."""
    global _lib
    if _lib is None:
        if not os.path.isfile(LIB_PATH):
            raise DasacError("libdasac_hip.so not found at {} -- build it with `python -c 'import __graft_entry__ as g; "
                             "g.build()'`; there is no CPU fallback".format(LIB_PATH))
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            if not hasattr(lib, name) and os.environ.get("DASAC_LIB"):
                continue                      # an older build of the library named by DASAC_LIB (A/B measurements): newer entry points absent
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


def check(code, what):
    if code != 0:
        raise DasacError("{} failed ({}): {}".format(what, code, load().dasac_last_error().decode()))


def stream_ptr():
    return torch.cuda.current_stream().cuda_stream


def ptr(t):
    return 0 if t is None else t.data_ptr()


_checked_devices = {}
EXPECTED_CUS, EXPECTED_ARCH = 256, "gfx950"       # csrc/common.hpp: kNumCu / kNumXcd = 256 / 8 size every grid and tile run


def check_device(index):
    """Fails loudly when the device cannot run the library at all -- anything but gfx950 with 64-wide waves: the code objects
    are compiled for that ISA only.  A gfx950 device with another CU count (a CPX / fewer-CU partition of an MI355X) RUNS every
    kernel correctly: grids and tile runs are merely sized for 256 CUs in 8 XCDs, and the one schedule that depends on the CU
    count, the persistent stream-K grid, refuses itself on the device side (`persistent_grid_fits`) and falls back to one
    block per tile.  That case is a warning, once per device."""
    if index in _checked_devices:
        if isinstance(_checked_devices[index], DasacError):
            raise _checked_devices[index]
        return _checked_devices[index]
    lib = load()
    cus, wave, arch = _i(0), _i(0), C.create_string_buffer(64)
    with torch.cuda.device(index):
        check(lib.dasac_device_info(C.byref(cus), C.byref(wave), arch, 64), "dasac_device_info")
    info = (cus.value, wave.value, arch.value.decode().split(":")[0])
    if info[1:] != (64, EXPECTED_ARCH):
        msg = "libdasac_hip.so is compiled for {} (64-wide waves) only; cuda:{} reports {}, wave {}".format(
            EXPECTED_ARCH, index, info[2], info[1])
        if os.environ.get("DASAC_ALLOW_OTHER_DEVICE", "0") != "1":
            raise DasacError(msg + " (set DASAC_ALLOW_OTHER_DEVICE=1 to try anyway)")
        import warnings
        warnings.warn(msg)
    elif info[0] != EXPECTED_CUS:
        import warnings
        warnings.warn("libdasac_hip.so sizes its grids for {} CUs; cuda:{} reports {} (partitioned device?): results are unaffected, "
                      "the persistent stream-K schedule is off and launch shapes are not tuned".format(EXPECTED_CUS, index, info[0]))
    _checked_devices[index] = info
    return info


def require_gpu(*tensors):
    for t in tensors:
        if t is not None:
            if not t.is_cuda:
                raise DasacError("dasac_hip ops run on the MI355X only (got a {} tensor); no CPU fallback".format(t.device))
            known = _checked_devices.get(t.device.index)
            if known is None:
                try:
                    check_device(t.device.index)
                except DasacError as exc:          # remember the refusal too: one dasac_device_info call per device, not per op
                    _checked_devices[t.device.index] = exc
                    raise
            elif isinstance(known, DasacError):
                raise known


_ws_cache = {}


def workspace(nbytes, device, owner=None):
    """Per-(device, stream) grow-only scratch buffer (the C ABI never allocates).  `owner` names a buffer that belongs to
    ONE kernel family and is zero-filled when (re)allocated -- the stream-K hand-off flags of dasac_conv_gemm are
    self-cleaning and must not be scribbled over by other ops' scratch data (include/dasac_hip.h)."""
    key = (device.index, torch.cuda.current_stream(device).cuda_stream, owner)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        alloc = torch.zeros if owner is not None else torch.empty
        buf = alloc(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf
