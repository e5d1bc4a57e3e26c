"""ctypes binding of libsatrender.so (the C ABI declared in include/satrender.h).

The library is built in-tree by ``__graft_entry__.build()`` (hipcc, gfx950).  There is NO fallback: if the
shared object is missing or a call fails, an exception is raised.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SATRENDER_LIB") or os.path.join(_HERE, "csrc", "libsatrender.so")  # env override: A/B builds

MODE_BF16 = 1
MODE_F16 = 2
MODE_BF16X3 = 3
FMT16, FMT8 = 16, 8  # training workspace formats (include/satrender.h)

_vp, _i, _i64, _f, _d = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_double


class MlpInputs(C.Structure):
    _fields_ = [("org", _vp), ("org_stride", _i), ("dir", _vp), ("dir_stride", _i), ("sun", _vp), ("sun_stride", _i),
                ("z", _vp), ("temb", _vp), ("ts", _vp), ("n_points", _i64), ("n_samples", _i)]


class RenderArgs(C.Structure):
    _fields_ = [("rays", _vp), ("ray_stride", _i), ("ts", _vp), ("temb", _vp), ("n_rays", _i64), ("n_samples", _i), ("z_in", _vp), ("u", _vp),
                ("seed", C.c_uint64), ("step_counter", _vp), ("tick", _i), ("noise", _vp), ("noise_std", _f), ("sky_hidden", _i),
                ("sky_w1", _vp), ("sky_b1", _vp), ("sky_w2", _vp), ("sky_b2", _vp), ("bank_chunks", _i64)]


class RenderOutputs(C.Structure):
    _fields_ = [(k, _vp) for k in ("z_vals", "albedo", "sigma", "sun_v", "beta", "sky", "weights", "transparency", "depth", "rgb")]


class TrainArgs(C.Structure):
    _fields_ = [("target", _vp), ("sched", _vp), ("beta_min", _f), ("loss_parts", _vp), ("rgb", _vp), ("d_sigma", _vp), ("d_albedo", _vp),
                ("d_sun_v", _vp), ("g_beta", _vp), ("d_sky", _vp), ("gather_idx", _vp), ("cursor", _vp), ("batches", _i64), ("out_rays", _vp),
                ("out_rgbs", _vp), ("out_ts", _vp)]


class PackScatter(C.Structure):
    _fields_ = [("map", _vp), ("hi", _vp), ("lo", _vp), ("l0", _vp), ("n_f16", _i64), ("scales", _f * 4)]


class LinearSrc(C.Structure):
    _fields_ = [("x", _vp), ("ld", _i), ("k", _i), ("act", _i), ("w0", _f), ("row_div", _i)]


ACT_NONE, ACT_SIN, ACT_RELU = 0, 1, 2
OUT_NONE, OUT_SOFTPLUS, OUT_SIGMOID, OUT_SIGMOID_RGB = 0, 1, 2, 3

# name -> (restype, argtypes); must list every symbol of include/satrender.h (tests/test_cabi.py checks this)
SIGNATURES = {
    "sr_version": (_i, []),
    "sr_last_error": (C.c_char_p, []),
    "sr_fwd_stream_elems": (_i64, [_i, _i]),
    "sr_bwd_stream_elems": (_i64, [_i, _i]),
    "sr_act_elems_per_tile": (_i64, [_i, _i]),
    "sr_satnerf_render_fwd": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "sr_render_points_per_block": (_i, [_i, _i]),
    "sr_satnerf_render_train": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "sr_pack_stream": (_i, [_vp, _vp, _vp, _i64, _vp, _vp, _i64, _vp]),
    "sr_dpre_elems_per_tile": (_i64, [_i, _i]),
    "sr_workspace_tiles": (_i64, [_i64]),
    "sr_dpre_workspace_elems": (_i64, [_i64, _i, _i]),
    "sr_unpack_grads": (_i, [_vp, _vp, _vp, _i64, _vp, _vp, _i, _vp]),
    "sr_composite_image": (_i, [_vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _i64, _i, _vp, _vp]),
    "sr_latlonalt_from_depth": (_i, [_vp, _i, _vp, _i64, _vp, _d, _vp, _vp, _vp, _vp]),
    "sr_linear_fwd": (_i, [C.POINTER(LinearSrc), _i, _vp, _vp, _i64, _i, _i, _vp, _i, _vp]),
    "sr_linear_bwd_input": (_i, [_vp, _i, _vp, _i, _i, _vp, _i, _i, C.POINTER(LinearSrc), _i64, _i, _vp, _i, _vp]),
    "sr_linear_bwd_weight": (_i, [_vp, _i, _vp, _i, _i, C.POINTER(LinearSrc), _i, _i64, _i, _vp, _vp, _vp]),
    "sr_positional_map": (_i, [_vp, _i, _i, _i64, _i, _vp, _vp]),
    "sr_points_along": (_i, [_vp, _i, _i, _vp, _i64, _i, _vp, _vp]),
    "sr_wgrad_plan": (_i, [_vp, _i, _i64, _i, _i, _vp]),
    "sr_satnerf_mlp_bwd": (_i, [_i, _i, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "sr_satnerf_wgrad": (_i, [_i, _i, _i64, _vp, _vp, _vp, _i, _i, _vp, _vp]),
    "sr_satnerf_wgrad8": (_i, [_i, _i, _i64, _vp, _i64, _vp, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "sr_wgrad8_load_ints": (_i, []),
    "sr_sky_bwd": (_i, [_vp, _i, _i64, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "sr_embedding_bwd": (_i, [_vp, _vp, _i64, _i, _i, _vp, _vp]),
    "sr_ray_setup": (_i, [_vp, _i, _vp, _i64, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "sr_sc_loss": (_i, [_vp, _vp, _vp, _f, _vp, _i64, _i, _f, _vp, _vp, _vp]),
    "sr_depth_loss": (_i, [_vp, _vp, _i, _i, _i64, _f, _vp, _vp, _vp]),
    "sr_ray_setup_rng": (_i, [_vp, _i, C.c_uint64, _vp, _i, _i64, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "sr_render_loss": (_i, [_vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _i64, _i, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "sr_satnerf_loss": (_i, [_vp, _vp, _vp, _vp, _i64, _i, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp]),
    "sr_adam_step": (_i, [_vp, _vp, _vp, _vp, _i64, _f, _f, _f, _f, _f, _i64, _i, _vp]),
    "sr_gather_setup": (_i, [_vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _i64, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, C.c_uint64, _vp, _i, _vp]),
    "sr_gather_batch": (_i, [_vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _i64, _vp]),
    "sr_grad_tail": (_i, [_vp, _vp, _vp, _i64, _vp, _i, _vp, _i, _vp, _i, _i64, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i,
                          _vp, _vp]),
    "sr_grad_tail_adam": (_i, [_vp, _vp, _vp, _i64, _vp, _i, _vp, _i, _vp, _i, _i64, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i,
                               _vp, _vp, _vp, _vp, _vp, _i, _vp, _f, _f, _f, _f, _f, _vp, _vp]),
    "sr_adam_step_graph": (_i, [_vp, _vp, _vp, _vp, _i64, _f, _f, _f, _f, _f, _vp, _i, _vp]),
    "sr_adam_step_pack": (_i, [_vp, _vp, _vp, _vp, _i64, _f, _f, _f, _f, _f, _vp, _i, _i64, _vp, _vp]),
    "sr_pack_all": (_i, [_vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _i64, _vp]),
    "sr_gather_scale_f32": (_i, [_vp, _vp, _vp, _i64, _vp, _vp]),
    "sr_ray_sample_fwd": (_i, [_vp, _i, _vp, _i64, _i, _vp, _vp]),
    "sr_sky_fwd": (_i, [_vp, _i, _i64, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    "sr_satnerf_mlp_fwd": (_i, [C.POINTER(MlpInputs), _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "sr_composite_fwd": (_i, [_vp, _vp, _vp, _f, _vp, _vp, _vp, _i64, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "sr_composite_bwd": (_i, [_vp, _vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp,
                              _vp, _vp, _vp]),
    "sr_sample_pdf_merge": (_i, [_vp, _vp, _vp, _i64, _i, _i, _f, _vp, _vp]),
    "sr_sample_pdf": (_i, [_vp, _vp, _vp, _i64, _i, _i, _f, _vp, _vp]),
    "sr_rpc_rays": (_i, [_vp, _i, _i, _d, _d, _vp, _d, _d, _d, _vp, _vp, _vp]),
}

_lib = None


class SatRenderError(RuntimeError):
    pass


def lib():
    """Load (once) and return the ctypes handle; raises if the HIP library has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SatRenderError(
                f"{LIB_PATH} not found: build the HIP library first (python -c 'import __graft_entry__ as g; g.build()'). "
                "satnerf_amd has no CPU or PyTorch fallback.")
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if the symbol is missing
            fn.restype, fn.argtypes = res, args
        _lib = handle
    return _lib


def call(name, *args):
    """Invoke an int-returning entry point; non-zero status raises with sr_last_error()."""
    handle = lib()
    rc = getattr(handle, name)(*args)
    if rc != 0:
        raise SatRenderError(f"{name} failed (status {rc}): {handle.sr_last_error().decode(errors='replace')}")
