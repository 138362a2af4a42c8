"""ctypes binding of libpstl_hip.so (C ABI: include/pstl_hip.h).

torch tensors are only the device-memory carrier: every call passes `tensor.data_ptr()` and the current HIP
stream.  There is NO fallback: if the library is missing, import of the compute path fails loudly.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpstl_hip.so")

PSTL_FLAG_CLIP = 1
PSTL_FLAG_MAXIMIZE = 2
PSTL_FLAG_CLIP_RECT = 4
PSTL_FLAG_NO_MERGE = 8
PSTL_FLAG_RNG = 16
PSTL_FLAG_NORM_STL = 32
PSTL_FLAG_KEEP_DH1 = 64

T = 20
NSEG = 15
CTRL = 40
HID = 256
FEAT = 224
NEI_PREP = 12

ABI_VERSION = 6      # include/pstl_hip.h PSTL_ABI_VERSION
SPLIT_F16_WMAX = 63.9   # include/pstl_hip.h PSTL_SPLIT_F16_WMAX
ADAM_MAX_TENSORS = 32   # include/pstl_hip.h PSTL_ADAM_MAX_TENSORS

class PstlCfg(ctypes.Structure):
    _fields_ = [("bs", ctypes.c_int32), ("rows_per_scene", ctypes.c_int32), ("S", ctypes.c_int32),
                ("K", ctypes.c_int32), ("steps", ctypes.c_int32), ("n_shards", ctypes.c_int32),
                ("flags", ctypes.c_int32), ("chain_waves", ctypes.c_int32),
                ("tau", ctypes.c_float), ("thres", ctypes.c_float), ("w_max", ctypes.c_float),
                ("a_max", ctypes.c_float), ("dt", ctypes.c_float), ("ego_L", ctypes.c_float),
                ("ego_W", ctypes.c_float), ("reserved_f", ctypes.c_float),
                ("seed", ctypes.c_uint64), ("row_offset", ctypes.c_int64), ("dyn", ctypes.c_void_p),
                ("plan_rows", ctypes.c_int64)]


class Mlp3(ctypes.Structure):
    _fields_ = [(n, ctypes.c_void_p) for n in ("w0", "b0", "w1", "b1", "w2", "b2")]


class WeightPtrs(ctypes.Structure):
    _fields_ = [(n, Mlp3) for n in ("ego_encoder", "neighbor_encoder", "lane_encoder", "policy_net", "merge_net",
                                    "rect_net")]


# One table for the exported symbols: (name, restype, argtypes).  `lib()` installs both, so that a wrong argument count or
# a Python float where the C side expects an int is a TypeError here instead of undefined behaviour there.
# C = const pstl_cfg*, P = pointer (device pointer, host array or struct, passed as void*), I = int, F = float,
# L = int64_t, Z = size_t.
_C = ctypes.POINTER(PstlCfg)
_P, _I, _F, _L, _Z = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_int64, ctypes.c_size_t
SIGNATURES = [
    ("pstl_version", _I, []),
    ("pstl_error_string", ctypes.c_char_p, [_I]),
    ("pstl_packed_weight_floats", _Z, []),
    ("pstl_packed_status_offset", _Z, []),
    ("pstl_pack_weights", _I, [ctypes.POINTER(WeightPtrs), _P, _P]),
    ("pstl_repack_weights", _I, [ctypes.POINTER(WeightPtrs), _P, _P]),
    ("pstl_time_bias", _I, [_P, _I, _P, _P]),
    ("pstl_fill_normal", _I, [_C, _I, _P, _P]),
    ("pstl_prepare_scene", _I, [_C] + [_P] * 7),
    ("pstl_encode_scene_work_floats", _Z, [_C]),
    ("pstl_encode_scene", _I, [_C] + [_P] * 14),
    ("pstl_rollout", _I, [_C] + [_P] * 9 + [_I, _I, _I, _P, _P, _I, _P]),
    ("pstl_rollout_layout", _I, [_C, _I, _P, _P, _P]),
    ("pstl_generate_trajs", _I, [_C] + [_P] * 4),
    ("pstl_stl_forward", _I, [_C, _P, _P, _P, _I] + [_P] * 10),
    ("pstl_stl_signals", _I, [_C] + [_P] * 7),
    ("pstl_stl_backward", _I, [_C] + [_P] * 10),
    ("pstl_guidance_step", _I, [_C] + [_P] * 6 + [_F, _I, _P, _P, _F, _I] + [_P] * 5),
    ("pstl_refine", _I, [_C] + [_P] * 9),
    ("pstl_reduce_metrics", _I, [_C] + [_P] * 5),
    ("pstl_select_plan", _I, [_C] + [_P] * 5),
    ("pstl_refine_train_forward", _I, [_C] + [_P] * 12),
    ("pstl_loss_grad", _I, [_C, _P, _P, _F, _P, _P, _P]),
    ("pstl_adam_step", _I, [_I, _P, _P, _P, _P, _P, _P, _I, _P, _F, _F, _F, _F, _P]),
    ("pstl_train_work_floats", _Z, [_C]),
    ("pstl_refine_backward", _I, [_C] + [_P] * 20),
    ("pstl_diversity", _I, [_C, _P, _P, _I] + [_P] * 8),
    ("pstl_stl_program_forward", _I, [_P, _I, _P, _L, _I, _P, _F, _I, _P, _P, _P]),
    ("pstl_stl_program_backward", _I, [_P, _I, _P, _L, _I, _P, _F, _I, _P, _P, _P, _P]),
    ("pstl_trajopt", _I, [_C] + [_P] * 6 + [_F, _F, _F, _I, _P, _P, _I, _P, _P, _P, _P]),
    ("pstl_diversity_loss", _I, [_C, _P, _P, _P, _F, _F, _I, _F] + [_P] * 6),
    ("pstl_stl_signals", _I, [_C] + [_P] * 7),
    ("pstl_refinement_work_floats", _Z, [_C]),
    ("pstl_refinement", _I, [_C] + [_P] * 6 + [_F, _F, _I, _P, _P, _P, _P, _I] + [_P] * 5),
    ("pstl_encode_scene_saved", _I, [_C] + [_P] * 17),
    ("pstl_encoder_backward_work_floats", _Z, [_C]),
    ("pstl_encoder_backward", _I, [_C] + [_P] * 17),
    ("pstl_merge_backward_work_floats", _Z, [_C]),
    ("pstl_merge_backward", _I, [_C, _P, _P, _P, _I] + [_P] * 8),
]
EXPORTS = sorted(set(n for n, _, _ in SIGNATURES))


_lib = None


def lib():
    """The loaded library.  Raises (never falls back) when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "libpstl_hip.so is missing (%s). Build it with `python -m pstl_diffusion_policy_amd.build` "
                "(hipcc, gfx950). There is no CPU fallback for this path." % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)    # always the in-tree build (tools/dbg/with_lib.py re-points LIB_PATH for experiments)
        L.pstl_version.restype = ctypes.c_int
        if L.pstl_version() != ABI_VERSION:
            raise RuntimeError("libpstl_hip.so has ABI version %d, this binding expects %d: rebuild with "
                               "`python -m pstl_diffusion_policy_amd.build`" % (L.pstl_version(), ABI_VERSION))
        for name, restype, argtypes in SIGNATURES:
            fn = getattr(L, name)     # AttributeError when the library lacks a symbol the header declares
            fn.restype = restype
            fn.argtypes = argtypes
        _lib = L
    return _lib


def rollout_layout(cfg, multi_step=True):
    """(kernel, tiles_per_group, rounds) pstl_rollout picks for this batch (include/pstl_hip.h, pstl_rollout_layout)."""
    k, g, r = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_int(0)
    check(lib().pstl_rollout_layout(ctypes.byref(cfg), int(bool(multi_step)), ctypes.addressof(k), ctypes.addressof(g),
                                    ctypes.addressof(r)), "rollout_layout")
    return k.value, g.value, r.value


def check(code, what=""):
    if code != 0:
        raise RuntimeError("libpstl_hip %s failed: %s (%d)" % (what, lib().pstl_error_string(code).decode(), code))


def stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t, dtype=torch.float32):
    """Device pointer of a tensor (None -> NULL).  The tensor must be contiguous, on the GPU and of `dtype`."""
    if t is None:
        return ctypes.c_void_p(0)
    if not isinstance(t, torch.Tensor):
        raise TypeError("expected a torch tensor")
    if t.dtype != dtype or not t.is_contiguous() or not t.is_cuda:
        raise ValueError("tensor must be a contiguous %s GPU tensor (got %s, contiguous=%s, device=%s)"
                         % (dtype, t.dtype, t.is_contiguous(), t.device))
    return ctypes.c_void_p(t.data_ptr())


def make_cfg(bs, rows_per_scene, S, K, steps, hp, flags=0, chain_waves=0, seed=0, row_offset=0, dyn=None, plan_rows=0):
    """plan_rows: the row count chain_waves = 0 picks its denoiser kernel for (0 = this call's own rows; a sharded job passes
    its nominal rows per GPU everywhere: include/pstl_hip.h).
    dyn: None, or a 4-float32 device tensor holding a pstl_dyn (seed as two 32-bit words, grad_scale, 0): the kernels read
    seed / grad_scale from it at run time (HIP-graph replay with new values, include/pstl_hip.h)."""
    if hp.get("norm_stl", False):     # --norm_stl travels with the hyper-parameters: every launch of the batch sees it
        flags = int(flags) | PSTL_FLAG_NORM_STL
    return PstlCfg(bs=int(bs), rows_per_scene=int(rows_per_scene), S=int(S), K=int(K), steps=int(steps),
                   n_shards=int(hp.get("n_shards", 4)), flags=int(flags), chain_waves=int(chain_waves),
                   tau=float(hp["smoothing_factor"]), thres=float(hp["stl_nn_thres"]), w_max=float(hp["mul_w_max"]),
                   a_max=float(hp["mul_a_max"]), dt=float(hp["dt"]), ego_L=float(hp["ego_L"]),
                   ego_W=float(hp["ego_W"]), reserved_f=0.0, seed=int(seed) & (2 ** 64 - 1), row_offset=int(row_offset),
                   dyn=None if dyn is None else ptr(dyn).value, plan_rows=int(plan_rows))


def f32(x, device):
    """float32, contiguous, on `device` (no copy when already so)."""
    return x.to(device=device, dtype=torch.float32).contiguous()
