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

T = 20
NSEG = 15
CTRL = 40
HID = 256
FEAT = 224
NEI_PREP = 12

ABI_VERSION = 2      # include/pstl_hip.h PSTL_ABI_VERSION
EXPORTS = ["pstl_version", "pstl_error_string", "pstl_packed_weight_floats", "pstl_pack_weights", "pstl_time_bias",
           "pstl_fill_normal",
           "pstl_prepare_scene", "pstl_encode_scene", "pstl_rollout", "pstl_generate_trajs", "pstl_stl_forward",
           "pstl_stl_backward", "pstl_guidance_step", "pstl_refine", "pstl_reduce_metrics",
           "pstl_refine_train_forward", "pstl_loss_grad", "pstl_train_create", "pstl_train_destroy",
           "pstl_train_work_floats", "pstl_refine_backward", "pstl_diversity",
           "pstl_stl_program_forward", "pstl_stl_program_backward", "pstl_trajopt",
           "pstl_diversity_loss", "pstl_stl_signals", "pstl_refinement", "pstl_refinement_work_floats",
           "pstl_encode_scene_saved", "pstl_encoder_backward", "pstl_encoder_backward_work_floats", "pstl_merge_backward",
           "pstl_merge_backward_work_floats", "pstl_encode_scene_work_floats"]


class PstlCfg(ctypes.Structure):
    _fields_ = [("bs", ctypes.c_int32), ("rows_per_scene", ctypes.c_int32), ("S", ctypes.c_int32),
                ("K", ctypes.c_int32), ("steps", ctypes.c_int32), ("n_shards", ctypes.c_int32),
                ("flags", ctypes.c_int32), ("chain_waves", ctypes.c_int32),
                ("tau", ctypes.c_float), ("thres", ctypes.c_float), ("w_max", ctypes.c_float),
                ("a_max", ctypes.c_float), ("dt", ctypes.c_float), ("ego_L", ctypes.c_float),
                ("ego_W", ctypes.c_float), ("reserved_f", ctypes.c_float),
                ("seed", ctypes.c_uint64), ("row_offset", ctypes.c_int64)]


class Mlp3(ctypes.Structure):
    _fields_ = [(n, ctypes.c_void_p) for n in ("w0", "b0", "w1", "b1", "w2", "b2")]


class WeightPtrs(ctypes.Structure):
    _fields_ = [(n, Mlp3) for n in ("ego_encoder", "neighbor_encoder", "lane_encoder", "policy_net", "merge_net",
                                    "rect_net")]


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
        L.pstl_error_string.restype = ctypes.c_char_p
        L.pstl_error_string.argtypes = [ctypes.c_int]
        L.pstl_packed_weight_floats.restype = ctypes.c_size_t
        for name in EXPORTS[3:]:
            getattr(L, name).restype = ctypes.c_int
        L.pstl_train_work_floats.restype = ctypes.c_size_t
        L.pstl_refinement_work_floats.restype = ctypes.c_size_t
        L.pstl_encoder_backward_work_floats.restype = ctypes.c_size_t
        L.pstl_merge_backward_work_floats.restype = ctypes.c_size_t
        L.pstl_encode_scene_work_floats.restype = ctypes.c_size_t
        _lib = L
    return _lib


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


def make_cfg(bs, rows_per_scene, S, K, steps, hp, flags=0, chain_waves=0, seed=0, row_offset=0):
    return PstlCfg(bs=int(bs), rows_per_scene=int(rows_per_scene), S=int(S), K=int(K), steps=int(steps),
                   n_shards=int(hp.get("n_shards", 4)), flags=int(flags), chain_waves=int(chain_waves),
                   tau=float(hp["smoothing_factor"]), thres=float(hp["stl_nn_thres"]), w_max=float(hp["mul_w_max"]),
                   a_max=float(hp["mul_a_max"]), dt=float(hp["dt"]), ego_L=float(hp["ego_L"]),
                   ego_W=float(hp["ego_W"]), reserved_f=0.0, seed=int(seed) & (2 ** 64 - 1), row_offset=int(row_offset))


def f32(x, device):
    """float32, contiguous, on `device` (no copy when already so)."""
    return x.to(device=device, dtype=torch.float32).contiguous()
