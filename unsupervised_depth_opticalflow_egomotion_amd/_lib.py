"""ctypes binding of libdfe_hip.so (the C ABI declared in include/dfe_hip.h).

The product path has no CPU fallback: if the library is missing or a tensor is not a
contiguous fp32 HIP tensor, the call raises.  PyTorch is used only for device memory and
the current HIP stream."""
from __future__ import annotations

import ctypes
import os
import re
import threading

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# DFE_HIP_LIB: another build of the same sources (the ablation / experiment variants tools/*_experiment.sh and tools/*ablate*.sh
# put under scratch/abl/), for measurements only
LIB_PATH = os.environ.get("DFE_HIP_LIB") or os.path.join(_HERE, "libdfe_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "dfe_hip.h")

_lib = None
_lock = threading.Lock()

_P = ctypes.c_void_p
_I = ctypes.c_int

# symbol -> argtypes (restype is int unless listed in _RESTYPES)
_SIGNATURES = {
    "dfe_abi_version": [],
    "dfe_error_string": [_I],
    "dfe_camera_floats": [],
    "dfe_prepare_cameras": [_P, _P, _P, _I, _I, _I, _P, _P],
    "dfe_pose_vec2mat_fwd": [_P, _P, _P, _I, _P],
    "dfe_pose_vec2mat_bwd": [_P, _P, _P, _P, _I, _P],
    "dfe_warp_flow_fwd": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "dfe_scatter_ws_bytes": [ctypes.c_long],
    "dfe_adam_chunk": [],
    "dfe_adam_step": [_P, _P, _I, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double, _P],
    "dfe_adam_step_dev": [_P, _P, _I, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double, _P, _P, _P],
    "dfe_warp_flow_bwd": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "dfe_pose_partials_floats": [_I, _I, _I],
    "dfe_inverse_warp2_fwd": [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "dfe_inverse_warp2_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "dfe_wgrad3x3_partials_floats": [_I, _I, _I, _I, _I],
    "dfe_wgrad3x3_fwd": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "dfe_wino_weight_floats": [_I, _I],
    "dfe_wino_conv3x3": [_P, _P, _P, ctypes.c_long, _P, ctypes.c_long, _I, _I, _I, _I, _I, _I, _I, _P],
    "dfe_wino_scratch_floats": [_I, _I, _I, _I, _I, _I],
    "dfe_wino_wgrad_floats": [_I, _I, _I, _I, _I, _I],
    "dfe_wino_wgrad_tune": [_I, _I, _I, _I],
    "dfe_sconv_tune": [_I, _I],
    "dfe_sconv_wgrad_floats": [_I, _I, _I, _I, _I, _I, _I, _I],
    "dfe_sconv_wgrad": [_P, ctypes.c_long, _P, ctypes.c_long, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P],
    "dfe_wino_wgrad3x3": [_P, ctypes.c_long, _P, ctypes.c_long, _P, _P, _I, _I, _I, _I, _I, _I, _P],
    "dfe_wino_conv3x3_dilated": [_P, _P, _P, ctypes.c_long, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    "dfe_wino_transform_blocks": [_I, _I],
    "dfe_wino_transform_weights_multi": [_P, _P, _I, _P],
    "dfe_wino_conv3x3_u": [_P, _P, _P, ctypes.c_long, _P, ctypes.c_long, _I, _I, _I, _I, _I, _I, _I, _P],
    "dfe_wino_conv3x3_u_act": [_P, _P, _P, ctypes.c_float, _P, ctypes.c_long, _P, ctypes.c_long, _P, ctypes.c_long, _I, _I, _I, _I, _I, _I, _I, _P],
    "dfe_conv1x1_small_supported": [_I, _I, _I, _I, _I],
    "dfe_conv1x1_small_fwd": [_P, _P, _P, ctypes.c_float, _P, _I, _I, _I, _I, _I, _P],
    "dfe_conv1x1_small_bwd": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "dfe_planeconv_supported": [_I, _I, _I, _I, _I],
    "dfe_planeconv_ws_floats": [_I, _I, _I, _I, _I],
    "dfe_planeconv_fwd": [_P, _P, _P, ctypes.c_float, _P, ctypes.c_long, _P, ctypes.c_long, _P, _I, _I, _I, _I, _I, _P],
    "dfe_planeconv_dgrad": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "dfe_planeconv_wgrad": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "dfe_disp_head_partials_floats": [_I, _I, _I, _I],
    "dfe_disp_head_fwd": [_P, _P, _P, _P, _I, _I, _I, _I, _P],
    "dfe_disp_head_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "dfe_flow_head_partials_floats": [_I, _I, _I, _I],
    "dfe_flow_head_fwd": [_P, _P, _P, _P, _I, _I, _I, _I, _P],
    "dfe_flow_head_bwd": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "dfe_bn_partials_floats": [_I, _I, _I, _I, _I],
    "dfe_bn_fwd": [_P] * 10 + [_I] * 5 + [ctypes.c_float, ctypes.c_float, _I, _P],
    "dfe_bn_bwd": [_P] * 12 + [_I] * 6 + [_P],
    "dfe_bias_act_partials_floats": [_I, _I, _I, _I],
    "dfe_bias_act_fwd": [_P, _P, _I, _I, _I, _I, ctypes.c_float, _P],
    "dfe_bias_act_bwd": [_P, _P, ctypes.c_long, _P, _P, _P, _I, _I, _I, _I, ctypes.c_float, _P],
    "dfe_bias_act_fwd2": [_P, _P, _P, ctypes.c_long, _P, ctypes.c_long, _I, _I, _I, _I, ctypes.c_float, _P],
    "dfe_bias_act_bwd2": [_P, ctypes.c_long, _P, ctypes.c_long, _P, ctypes.c_long, _P, _P, _P, _I, _I, _I, _I, ctypes.c_float, _P],
    "dfe_bias_grad_final_multi": [_P, _P, _P, _I, _I, _I, _I, _P],
    "dfe_glue_partials_floats": [_I, _I, _I, _I],
    "dfe_elu_pad_fwd": [_P, _P, _P, _I, _I, _I, _I, _I, _P],
    "dfe_elu_pad_bwd": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "dfe_elu_up2_cat_pad_fwd": [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "dfe_elu_up2_cat_pad_bwd": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "dfe_rigid_flow_fwd": [_P, _P, _P, _I, _I, _I, _P],
    "dfe_rigid_flow_bwd": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _P],
    "dfe_occ_masks": [_P] * 7 + [_I, _I, _I, _P],
    "dfe_texture_mask": [_P, _P, _P, _P, _I, _I, _I, _P],
    "dfe_dynamic_mask": [_P, _P, _P, _P, ctypes.c_float, ctypes.c_float, _I, _I, _I, _P],
    "dfe_exact_math_selftest": [_P, ctypes.c_ulonglong, _P],
    "dfe_prepare_triplets": [_P, _P, _P, _I, _I, _I, _I, _I, _P],
    "dfe_forward_splat_ones": [_P, _P, _P, _I, _I, _I, _I, _P],
    "dfe_ssim_fwd": [_P, _P, _P, _I, _I, _I, _I, _P],
    "dfe_ssim_bwd": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    "dfe_corr_fwd": [_P, _P, _P, _I, _I, _I, _I, _I, _P],
    "dfe_corr_bwd": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "dfe_maxpool3x3s2_out": [_I],
    "dfe_maxpool3x3s2_fwd": [_P, _P, _P, _I, _I, _I, _P],
    "dfe_maxpool3x3s2_bwd": [_P, _P, _P, _I, _I, _I, _P],
    "dfe_pwc_level_channels": [_I],
    "dfe_pwc_level_fwd": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "dfe_pwc_level_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "dfe_pwc_level_map_bytes": [_I, _I, _I],
    "dfe_pwc_level_fwd_map": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "dfe_pwc_level_bwd_map": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    "dfe_resize": [_P, _P, _I, _I, _I, _I, _I, _I, _P],
    "dfe_resize_bilinear_fwd": [_P, _P, _I, _I, _I, _I, _I, ctypes.c_float, _I, _P],
    "dfe_resize_bilinear_bwd": [_P, _P, _I, _I, _I, _I, _I, ctypes.c_float, _I, _P],
    "dfe_geom_workspace_floats": [_P],
    "dfe_geom_maskpack_offset_bytes": [_P, _I],
    "dfe_geom_loss_fwd": [_P, _P],
    "dfe_geom_loss_bwd": [_P, _P],
    "dfe_geom_loss_fwd_profiled": [_P, _P, _P],
    "dfe_geom_loss_bwd_profiled": [_P, _P, _P],
    "dfe_geom_loss_fwd_timed": [_P, _P, _P],
    "dfe_geom_loss_bwd_timed": [_P, _P, _P],
    "dfe_geom_timed_collect": [_P, _P],
}
_RESTYPES = {"dfe_error_string": ctypes.c_char_p, "dfe_geom_workspace_floats": ctypes.c_long,
             "dfe_bias_act_partials_floats": ctypes.c_long, "dfe_glue_partials_floats": ctypes.c_long,
             "dfe_bn_partials_floats": ctypes.c_long, "dfe_disp_head_partials_floats": ctypes.c_long,
             "dfe_flow_head_partials_floats": ctypes.c_long, "dfe_pwc_level_map_bytes": ctypes.c_long,
             "dfe_wgrad3x3_partials_floats": ctypes.c_long, "dfe_planeconv_ws_floats": ctypes.c_long, "dfe_wino_weight_floats": ctypes.c_long, "dfe_wino_scratch_floats": ctypes.c_long, "dfe_wino_wgrad_floats": ctypes.c_long, "dfe_sconv_wgrad_floats": ctypes.c_long, "dfe_wino_transform_blocks": ctypes.c_long,
             "dfe_geom_maskpack_offset_bytes": ctypes.c_long, "dfe_scatter_ws_bytes": ctypes.c_long}


class DfeError(RuntimeError):
    pass


def header_symbols(path: str = HEADER_PATH):
    """Function names declared in include/dfe_hip.h."""
    with open(path) as fh:
        text = re.sub(r"/\*.*?\*/", "", fh.read(), flags=re.S)
    return sorted(set(re.findall(r"\b(dfe_[a-z0-9_]+)\s*\(", text)))


def header_abi_version(path: str = HEADER_PATH) -> int:
    """``DFE_ABI_VERSION`` as include/dfe_hip.h states it: bumped whenever an exported signature changes or disappears, so a
    stale libdfe_hip.so is refused at load time instead of being called with shifted arguments."""
    with open(path) as fh:
        return int(re.search(r"#define\s+DFE_ABI_VERSION\s+(\d+)", fh.read()).group(1))


def get_lib():
    """Load libdfe_hip.so once; raise loudly when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise DfeError(
                    "libdfe_hip.so not found at %s: build it with `python -c 'import __graft_entry__ as g; "
                    "g.build()'` (hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)
            lib = ctypes.CDLL(LIB_PATH)
            for name in header_symbols():
                fn = getattr(lib, name)  # AttributeError if the header and the library disagree
                if name in _SIGNATURES:
                    fn.argtypes = _SIGNATURES[name]
                fn.restype = _RESTYPES.get(name, ctypes.c_int)
            if lib.dfe_abi_version() != header_abi_version():
                raise DfeError("libdfe_hip.so ABI version %d != include/dfe_hip.h's DFE_ABI_VERSION %d: a stale build; "
                               "rebuild with __graft_entry__.build()" % (lib.dfe_abi_version(), header_abi_version()))
            _lib = lib
    return _lib


def check(code: int, what: str = ""):
    if code != 0:
        msg = get_lib().dfe_error_string(code)
        raise DfeError("%s failed: %s (code %d)" % (what or "dfe call", msg.decode() if msg else "?", code))


def ptr(t, strided=False):
    """Device pointer of a contiguous fp32 HIP tensor (None -> NULL).  ``strided=True`` skips the contiguity check
    for entry points that take explicit strides."""
    if t is None:
        return None
    if not isinstance(t, torch.Tensor):
        raise DfeError("expected a torch.Tensor, got %r" % type(t))
    if not t.is_cuda:
        raise DfeError("the HIP loss stack only accepts tensors on a HIP device (got %s); there is no CPU "
                       "fallback in the product path" % t.device)
    if t.dtype not in (torch.float32, torch.uint8, torch.int32, torch.int64):
        raise DfeError("unsupported dtype %s" % t.dtype)
    if not strided and not t.is_contiguous():
        raise DfeError("tensor must be contiguous")
    return ctypes.c_void_p(t.data_ptr())


def raw_stream():
    """The current HIP stream's handle as an integer.  ``torch.cuda.current_stream()`` builds a Stream object and resolves the
    device index through four Python frames (~9 us under cProfile, ~180 calls per training step: 1.6 ms of the ~17 ms the host
    needs to enqueue a step -- scratch/host_prof.py, round 6: 17.6 -> 14.3 ms at B = 1); the raw accessor is one C call."""
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())


def stream_ptr():
    return ctypes.c_void_p(raw_stream())


def f32c(t):
    """Contiguous fp32 view/copy on the same device."""
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()
