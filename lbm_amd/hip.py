"""ctypes binding of ``liblbm_hip.so`` (C ABI in ``include/lbm_hip.h``).

PyTorch is used for device memory and streams only: every function here hands raw device pointers of
caller-owned tensors to the HIP library and enqueues on ``torch.cuda.current_stream()``.  There is no
PyTorch/CPU fallback — if the library is missing or no GPU is present the calls raise.
"""
from __future__ import annotations

import ctypes as C
import os
import re
from pathlib import Path
from typing import Dict, List, Optional, Sequence, Tuple

import torch

_HERE = Path(__file__).resolve().parent
_LIB_PATH = Path(os.environ.get("LBM_AMD_LIB", _HERE / "liblbm_hip.so"))     # LBM_AMD_LIB: A/B builds of the library (development)
_HEADER = _HERE.parent / "include" / "lbm_hip.h"
MAX_LAYERS = 8

# every symbol include/lbm_hip.h declares (checked at load time and by the CPU tests)
EXPORTS = (
    "lbm_hip_version", "lbm_hip_last_error", "lbm_hip_linear", "lbm_hip_layernorm", "lbm_hip_l2norm",
    "lbm_hip_attn_readout", "lbm_hip_pyramid", "lbm_hip_corr_maps", "lbm_hip_topk", "lbm_hip_msda_sample",
    "lbm_hip_preprocess", "lbm_hip_upcat", "lbm_hip_sample_queries", "lbm_hip_step_workspace_bytes",
    "lbm_hip_track_step", "lbm_hip_step_workspace_offset", "lbm_hip_conv2d_nhwc", "lbm_hip_maxpool3x3s2_nhwc",
    "lbm_hip_profile_begin", "lbm_hip_profile_end", "lbm_hip_split_planes", "lbm_hip_debug_set", "lbm_hip_linear_ss",
    "lbm_hip_conv2d_nhwc_ss", "lbm_hip_maxpool3x3s2_nhwc_s", "lbm_hip_upcat_s", "lbm_hip_pyramid_s",
    "lbm_hip_update_points", "lbm_hip_encoder_tail", "lbm_hip_pack_fragment_major", "lbm_hip_ffn_slabs",
    "lbm_hip_debug_ptr", "lbm_hip_update_points_static", "lbm_hip_preprocess_u8", "lbm_hip_corr_topk",
    "lbm_hip_corr_topk_workspace_bytes", "lbm_hip_preprocess_split", "lbm_hip_stem_pool",
    "lbm_hip_encoder_tail_fused", "lbm_hip_debug_active", "lbm_hip_tuning_build", "lbm_hip_attn_readout_h16",
)


class HipError(RuntimeError):
    pass


class LinearDesc(C.Structure):
    _fields_ = [("X", C.c_void_p), ("W", C.c_void_p), ("bias", C.c_void_p), ("Y", C.c_void_p), ("res", C.c_void_p),
                ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32), ("ldx", C.c_int32), ("ldw", C.c_int32),
                ("ldy", C.c_int32), ("ldres", C.c_int32), ("res_div", C.c_int32), ("res_mod", C.c_int32),
                ("res_scale", C.c_float), ("act", C.c_int32), ("batch", C.c_int32),
                ("sX", C.c_int64), ("sW", C.c_int64), ("sB", C.c_int64), ("sY", C.c_int64), ("sR", C.c_int64),
                ("ln_gamma", C.c_void_p), ("ln_beta", C.c_void_p), ("keep", C.c_void_p), ("fallback", C.c_void_p),
                ("Yn", C.c_void_p), ("W_hi", C.c_void_p), ("W_lo", C.c_void_p)]


class LinearSSDesc(C.Structure):
    _fields_ = [("A", C.c_void_p), ("B", C.c_void_p), ("a_plane", C.c_int64), ("b_plane", C.c_int64),
                ("bias", C.c_void_p), ("Y", C.c_void_p), ("Ys", C.c_void_p), ("ys_plane", C.c_int64), ("res", C.c_void_p),
                ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32), ("lda", C.c_int32), ("ldb", C.c_int32),
                ("ldy", C.c_int32), ("ldys", C.c_int32), ("ldres", C.c_int32), ("res_div", C.c_int32), ("res_mod", C.c_int32),
                ("res_scale", C.c_float), ("act", C.c_int32), ("batch", C.c_int32),
                ("sA", C.c_int64), ("sB", C.c_int64), ("sBias", C.c_int64), ("sY", C.c_int64), ("sYs", C.c_int64),
                ("sR", C.c_int64), ("planes", C.c_int32)]


class MsdaDesc(C.Structure):
    _fields_ = [("value", C.c_void_p), ("v_batch_stride", C.c_int64), ("ldv", C.c_int32),
                ("lvl_h", C.c_int32 * 4), ("lvl_w", C.c_int32 * 4), ("lvl_start", C.c_int32 * 4),
                ("B", C.c_int32), ("N", C.c_int32), ("k", C.c_int32), ("P", C.c_int32),
                ("offlog", C.c_void_p), ("ld_offlog", C.c_int32), ("logit_col", C.c_int32),
                ("ref", C.c_void_p), ("net_w", C.c_float), ("net_h", C.c_float),
                ("loc", C.c_void_p), ("attw", C.c_void_p), ("out", C.c_void_p), ("value_f16", C.c_int32)]


class StepDesc(C.Structure):
    _fields_ = [("B", C.c_int32), ("N", C.c_int32), ("M", C.c_int32), ("n_layers", C.c_int32),
                ("topk", C.c_int32 * MAX_LAYERS),
                ("net_h", C.c_int32), ("net_w", C.c_int32), ("stride", C.c_int32),
                ("n_points", C.c_int32), ("n_points_cmem", C.c_int32),
                ("ring_head", C.c_int32), ("update_state", C.c_int32),
                ("vis_threshold", C.c_float), ("img_h", C.c_float), ("img_w", C.c_float),
                ("corr_mix", C.c_float * (MAX_LAYERS * 4)), ("corr_bias", C.c_float * MAX_LAYERS),
                ("weights", C.POINTER(C.c_void_p)), ("n_weights", C.c_int32),
                ("precision", C.c_int32), ("weights_lo", C.POINTER(C.c_void_p)),
                ("tokens", C.c_void_p), ("tokens_split", C.c_void_p), ("query", C.c_void_p), ("stream_mem", C.c_void_p), ("coll_mem", C.c_void_p),
                ("vis_mask", C.c_void_p), ("mem_mask", C.c_void_p), ("queried", C.c_void_p),
                ("coords", C.c_void_p), ("vis", C.c_void_p), ("rho", C.c_void_p), ("offset", C.c_void_p),
                ("stream_out", C.c_void_p), ("coll_out", C.c_void_p),
                ("ref_points", C.c_void_p * MAX_LAYERS), ("topk_idx", C.c_void_p * MAX_LAYERS),
                ("ref_rho", C.c_void_p * MAX_LAYERS), ("corr", C.c_void_p * MAX_LAYERS),
                ("workspace", C.c_void_p), ("workspace_bytes", C.c_uint64),
                ("aux_streams", C.c_void_p * 3), ("n_aux_streams", C.c_int32),
                ("phase", C.c_int32), ("row_lo", C.c_int32), ("row_count", C.c_int32), ("sub_index", C.c_int32),
                ("query_split", C.c_void_p), ("vis_flag", C.c_void_p), ("rho_prob", C.c_void_p),
                ("query_u0", C.c_void_p), ("query_u0_valid", C.c_int32), ("inv_norm", C.c_void_p)]


class ProfileEntry(C.Structure):
    _fields_ = [("name", C.c_char * 96), ("launches", C.c_int64), ("total_ms", C.c_double), ("flops", C.c_double),
                ("bytes", C.c_double)]


_lib: Optional[C.CDLL] = None
_consts: Optional[Dict[str, int]] = None


def header_constants() -> Dict[str, int]:
    """``#define LBM_* <int>`` values of include/lbm_hip.h (the weight-table layout lives there)."""
    global _consts
    if _consts is None:
        text = _HEADER.read_text()
        _consts = {m.group(1): int(m.group(2)) for m in re.finditer(r"#define\s+(LBM_\w+)\s+(\d+)\b", text)}
    return _consts


def lib_path() -> Path:
    return _LIB_PATH


def load() -> C.CDLL:
    """Load the shared library (no GPU needed for loading); raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not _LIB_PATH.exists():
        raise HipError(f"{_LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                       f"or `make -C {_HERE / 'csrc'}`; lbm_amd has no fallback path")
    lib = C.CDLL(str(_LIB_PATH))
    for name in EXPORTS:
        if not hasattr(lib, name):
            raise HipError(f"{_LIB_PATH} does not export {name}")
    lib.lbm_hip_last_error.restype = C.c_char_p
    lib.lbm_hip_step_workspace_bytes.restype = C.c_uint64
    lib.lbm_hip_step_workspace_offset.restype = C.c_int64
    lib.lbm_hip_step_workspace_offset.argtypes = [C.POINTER(StepDesc), C.c_char_p]
    lib.lbm_hip_linear.argtypes = [C.POINTER(LinearDesc), C.c_void_p]
    lib.lbm_hip_layernorm.argtypes = [C.c_void_p] * 4 + [C.c_int32] + [C.c_void_p] * 4
    lib.lbm_hip_l2norm.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]
    lib.lbm_hip_attn_readout.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_int32,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
    lib.lbm_hip_attn_readout_h16.argtypes = lib.lbm_hip_attn_readout.argtypes
    lib.lbm_hip_pyramid.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]
    lib.lbm_hip_corr_maps.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_float), C.c_int32, C.c_int32,
                                      C.c_int32, C.c_int32, C.c_void_p]
    lib.lbm_hip_topk.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                 C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.lbm_hip_msda_sample.argtypes = [C.POINTER(MsdaDesc), C.c_void_p]
    lib.lbm_hip_preprocess.argtypes = [C.c_void_p, C.c_int64, C.c_void_p] + [C.c_int32] * 6 + [C.c_void_p]
    lib.lbm_hip_preprocess_u8.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p] + [C.c_int32] * 6 + [C.c_void_p]
    lib.lbm_hip_upcat.argtypes = [C.c_void_p] * 4 + [C.c_int32] * 3 + [C.c_void_p]
    lib.lbm_hip_sample_queries.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p] + [C.c_int32] * 6 + [C.c_void_p]
    lib.lbm_hip_pack_fragment_major.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
    lib.lbm_hip_ffn_slabs.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                      C.c_int32, C.c_int32, C.c_void_p]
    lib.lbm_hip_encoder_tail.argtypes = [C.c_void_p] * 6 + [C.c_int64] + [C.c_int32] * 3 + [C.c_void_p]
    lib.lbm_hip_update_points.argtypes = ([C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32] + [C.c_void_p] * 13)
    lib.lbm_hip_update_points_static.argtypes = ([C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32] + [C.c_void_p] * 12)
    lib.lbm_hip_step_workspace_bytes.argtypes = [C.POINTER(StepDesc)]
    lib.lbm_hip_track_step.argtypes = [C.POINTER(StepDesc), C.c_void_p]
    lib.lbm_hip_conv2d_nhwc.argtypes = [C.c_void_p] * 6 + [C.c_int32] * 11 + [C.c_void_p]
    lib.lbm_hip_maxpool3x3s2_nhwc.argtypes = [C.c_void_p, C.c_void_p] + [C.c_int32] * 4 + [C.c_void_p]
    lib.lbm_hip_conv2d_nhwc_ss.argtypes = ([C.c_void_p, C.c_int64] + [C.c_void_p] * 6 + [C.c_int64, C.c_void_p, C.c_void_p, C.c_uint64] +
                                           [C.c_int32] * 11 + [C.c_void_p])
    lib.lbm_hip_maxpool3x3s2_nhwc_s.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64] + [C.c_int32] * 4 + [C.c_void_p]
    lib.lbm_hip_upcat_s.argtypes = [C.c_void_p] * 5 + [C.c_int64] + [C.c_int32] * 3 + [C.c_void_p]
    lib.lbm_hip_pyramid_s.argtypes = [C.c_void_p, C.c_void_p, C.c_int64] + [C.c_int32] * 3 + [C.c_void_p]
    lib.lbm_hip_linear_ss.argtypes = [C.POINTER(LinearSSDesc), C.c_void_p]
    lib.lbm_hip_split_planes.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
    lib.lbm_hip_profile_end.argtypes = [C.POINTER(ProfileEntry), C.c_int32]
    lib.lbm_hip_debug_ptr.argtypes = [C.c_int32, C.c_void_p]
    lib.lbm_hip_debug_set.argtypes = [C.c_int32, C.c_int32]
    lib.lbm_hip_debug_active.argtypes = [C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_int32]
    lib.lbm_hip_debug_active.restype = C.c_int32
    lib.lbm_hip_tuning_build.restype = C.c_int32
    lib.lbm_hip_preprocess_split.argtypes = [C.c_void_p, C.c_int32, C.c_int64, C.c_int32, C.c_void_p, C.c_int64] + [C.c_int32] * 5 + [C.c_void_p]
    lib.lbm_hip_stem_pool.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                      C.c_int32, C.c_int32, C.c_int32, C.c_void_p]
    lib.lbm_hip_encoder_tail_fused.argtypes = ([C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int32] + [C.c_void_p] * 6 + [C.c_int64] +
                                               [C.c_void_p] + [C.c_int32] * 3 + [C.c_void_p])
    lib.lbm_hip_corr_topk_workspace_bytes.restype = C.c_int64
    lib.lbm_hip_corr_topk_workspace_bytes.argtypes = [C.c_int32] * 4
    lib.lbm_hip_corr_topk.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                      C.c_int32, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    _lib = lib
    return lib


def tensor_version(t: torch.Tensor):
    """The tensor's in-place version counter, or None for tensors created under torch.inference_mode() (they do not track one:
    in-place edits of such a tensor are then not detected by the caches keyed on it)."""
    return None if t.is_inference() else t._version


def frame_hw(frames: torch.Tensor, u8_layout: str = "chw") -> Tuple[int, int]:
    """(H, W) of a batch of frames: (B,3,H,W), or (B,H,W,3) for uint8 frames in an interleaved layout."""
    if frames.dtype == torch.uint8 and u8_layout != "chw":
        return int(frames.shape[-3]), int(frames.shape[-2])
    return int(frames.shape[-2]), int(frames.shape[-1])


def debug_active() -> Dict[int, int]:
    """Knobs of lbm_hip_debug_set that differ from their defaults right now: {key: value}."""
    lib = load()
    n = int(lib.lbm_hip_debug_active(None, None, 0))
    if n == 0:
        return {}
    keys, vals = (C.c_int32 * n)(), (C.c_int32 * n)()
    n = min(n, int(lib.lbm_hip_debug_active(keys, vals, n)))
    return {int(keys[i]): int(vals[i]) for i in range(n)}


def _check(rc: int, what: str) -> None:
    if rc != 0:
        raise HipError(f"{what}: {load().lbm_hip_last_error().decode()}")


def _require_gpu() -> None:
    if not torch.cuda.is_available():
        raise HipError("lbm_amd needs a ROCm GPU (gfx950); there is no CPU fallback")


def _p(t: Optional[torch.Tensor], dtype=torch.float32) -> Optional[int]:
    """Raw device pointer of a contiguous CUDA tensor of the expected dtype."""
    if t is None:
        return None
    if not t.is_cuda:
        raise HipError("expected a GPU tensor")
    if dtype is not None and t.dtype != dtype:
        if not (dtype == torch.uint8 and t.dtype == torch.bool):
            raise HipError(f"expected dtype {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise HipError("expected a contiguous tensor")
    return t.data_ptr()


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def profile_begin() -> None:
    """Bracket every kernel launch of the library with HIP events until profile_end()."""
    _check(load().lbm_hip_profile_begin(), "lbm_hip_profile_begin")


def profile_end() -> List[dict]:
    """Per-kernel totals since profile_begin(): name, launches, total_ms, algorithmic flops and bytes."""
    buf = (ProfileEntry * 128)()
    n = load().lbm_hip_profile_end(buf, 128)
    if n < 0:
        raise HipError("lbm_hip_profile_end: " + load().lbm_hip_last_error().decode())
    return [dict(name=buf[i].name.decode(), launches=int(buf[i].launches), total_ms=float(buf[i].total_ms),
                 flops=float(buf[i].flops), bytes=float(buf[i].bytes)) for i in range(min(n, 128))]


# ----------------------------------------------------------------------------------------------
# op-level wrappers (used by the module mirrors and by the parity tests)
# ----------------------------------------------------------------------------------------------

class SplitWeight:
    """A weight matrix [N, K] pre-split into two f16 planes (hi, lo * 2^11) for the split-precision GEMM."""

    def __init__(self, w: torch.Tensor):
        _require_gpu()
        w = w.detach().float().contiguous()
        if w.dim() != 2 or w.shape[1] % 8 != 0:
            raise HipError("SplitWeight needs a 2-d matrix whose row length is a multiple of 8")
        self.shape = tuple(w.shape)
        self.planes = torch.empty(2, *w.shape, device=w.device, dtype=torch.float16)
        _check(load().lbm_hip_split_planes(w.data_ptr(), self.planes[0].data_ptr(), self.planes[1].data_ptr(), w.numel(),
                                           _stream()), "lbm_hip_split_planes")

    @property
    def hi(self) -> int:
        return self.planes[0].data_ptr()

    @property
    def lo(self) -> int:
        return self.planes[1].data_ptr()


class SplitTensor:
    """An activation stored as two f16 planes (hi, lo * 2^11): `planes` has shape (2, *shape)."""

    inv_norm: Optional[torch.Tensor] = None     # (B, L) row factors of F.normalize, when the producer of the twin also wrote them

    def __init__(self, shape, device):
        self.shape = tuple(shape)
        self.planes = torch.empty((2,) + self.shape, device=device, dtype=torch.float16)

    @property
    def hi(self) -> int:
        return self.planes.data_ptr()

    @property
    def plane(self) -> int:
        return self.planes[0].numel()

    def float(self) -> torch.Tensor:
        return self.planes[0].float() + self.planes[1].float() / 2048.0

    def clone(self) -> "SplitTensor":
        t = SplitTensor.__new__(SplitTensor)
        t.planes = self.planes.clone()
        t.shape = self.shape
        return t

    def reshape(self, *shape) -> "SplitTensor":
        t = SplitTensor.__new__(SplitTensor)
        t.planes = self.planes.view(2, *shape)
        t.shape = tuple(t.planes.shape[1:])
        return t


def split_tensor(x: torch.Tensor) -> SplitTensor:
    """fp32 tensor -> its split twin (lbm_hip_split_planes)."""
    _require_gpu()
    x = x.float().contiguous()
    t = SplitTensor(tuple(x.shape), x.device)
    _check(load().lbm_hip_split_planes(x.data_ptr(), t.planes[0].data_ptr(), t.planes[1].data_ptr(), x.numel(), _stream()), "lbm_hip_split_planes")
    return t


_zeros: Dict[str, torch.Tensor] = {}


def zero_page(device) -> int:
    key = str(device)
    if key not in _zeros:
        _zeros[key] = torch.zeros(256, device=device, dtype=torch.uint8)
    return _zeros[key].data_ptr()


def conv2d_nhwc_ss(x: "SplitTensor", w: "SplitWeight", bias: torch.Tensor, kh: int, stride: int, pad: int, relu: bool,
                   res: Optional[torch.Tensor] = None, want_f32: bool = True, want_split: bool = True,
                   scratch: Optional[torch.Tensor] = None, planes: int = 2):
    """Split-precision NHWC convolution: x (B,H,W,Cin) split, packed weights split -> (y fp32 | None, ys SplitTensor | None)."""
    _require_gpu()
    B, H, W, Cin = x.shape
    Cout = w.shape[0]
    Ho = (H + 2 * pad - kh) // stride + 1
    Wo = (W + 2 * pad - kh) // stride + 1
    dev = x.planes.device
    y = torch.empty(B, Ho, Wo, Cout, device=dev, dtype=torch.float32) if want_f32 else None
    ys = SplitTensor((B, Ho, Wo, Cout), dev) if want_split else None
    _check(load().lbm_hip_conv2d_nhwc_ss(x.hi, x.plane, w.hi, w.lo if planes == 2 else None, _p(bias), _p(res), _p(y), ys.hi if ys else None,
                                         ys.plane if ys else 0, zero_page(dev), _p(scratch, None) if scratch is not None else None,
                                         scratch.numel() * scratch.element_size() if scratch is not None else 0,
                                         B, H, W, Cin, Cout, kh, kh, stride, pad, 1 if relu else 0, w.shape[1], _stream()),
           "lbm_hip_conv2d_nhwc_ss")
    return y, ys


def update_points(remove_mask: torch.Tensor, update_mask: torch.Tensor, old, new_query: torch.Tensor, M: int):
    """Device-side edit of the tracked point set (lbm_hip_update_points).  old = (query (1,N_old,C), stream, coll (1,N_old,M,C),
    vis, mem (1,N_old,M) uint8) or None; returns the new (query, stream, coll, vis, mem) for N_new = len(update_mask) points."""
    _require_gpu()
    dev = new_query.device
    N_new = int(update_mask.shape[0])
    N_old = 0 if old is None else int(old[0].shape[1])
    Cc = new_query.shape[-1]
    query = torch.empty(1, N_new, Cc, device=dev)
    stream = torch.empty(1, N_new, M, Cc, device=dev)
    coll = torch.empty(1, N_new, M, Cc, device=dev)
    vis = torch.empty(1, N_new, M, device=dev, dtype=torch.uint8)
    mem = torch.empty(1, N_new, M, device=dev, dtype=torch.uint8)
    scratch = torch.empty(N_old + 1, device=dev, dtype=torch.int32)
    rm = remove_mask.to(torch.uint8).contiguous() if N_old else None
    um = update_mask.to(torch.uint8).contiguous()
    o = old if N_old else (None,) * 5
    _check(load().lbm_hip_update_points(_p(rm, torch.uint8), N_old, _p(um, torch.uint8), N_new, M, _p(o[0]), _p(o[1]), _p(o[2]),
                                        _p(o[3], torch.uint8), _p(o[4], torch.uint8), _p(new_query.contiguous()), _p(query),
                                        _p(stream), _p(coll), _p(vis, torch.uint8), _p(mem, torch.uint8), scratch.data_ptr(),
                                        _stream()), "lbm_hip_update_points")
    return query, stream, coll, vis, mem


def fragment_major(sw: "SplitWeight") -> torch.Tensor:
    """Fragment-major copy [N/32, K/16, 2, 64, 8] (f16) of a split weight matrix (lbm_hip_pack_fragment_major)."""
    _require_gpu()
    N, K = sw.shape
    out = torch.empty(N // 32, K // 16, 2, 64, 8, device=sw.planes.device, dtype=torch.float16)
    _check(load().lbm_hip_pack_fragment_major(sw.hi, sw.lo, N, K, out.data_ptr(), _stream()), "lbm_hip_pack_fragment_major")
    return out


def ffn_slabs(xs: "SplitTensor", f1_fm: torch.Tensor, b1: torch.Tensor, f2_fm: torch.Tensor, planes: int = 2) -> torch.Tensor:
    """xs: split [M, 256] rows -> (8, M, 256) partial outputs of the fused feed-forward (lbm_hip_ffn_slabs)."""
    _require_gpu()
    M = xs.shape[0]
    slabs = torch.empty(8, M, 256, device=xs.planes.device, dtype=torch.float32)
    _check(load().lbm_hip_ffn_slabs(xs.hi, xs.plane, xs.shape[1], f1_fm.data_ptr(), _p(b1), f2_fm.data_ptr(), slabs.data_ptr(),
                                    M * 256, M, planes, _stream()), "lbm_hip_ffn_slabs")
    return slabs


def maxpool3x3s2_nhwc_s(x: torch.Tensor):
    _require_gpu()
    B, H, W, Cc = x.shape
    Ho, Wo = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
    y = torch.empty(B, Ho, Wo, Cc, device=x.device, dtype=torch.float32)
    ys = SplitTensor((B, Ho, Wo, Cc), x.device)
    _check(load().lbm_hip_maxpool3x3s2_nhwc_s(_p(x), _p(y), ys.hi, ys.plane, B, H, W, Cc, _stream()), "lbm_hip_maxpool3x3s2_nhwc_s")
    return y, ys


def upcat_s(p1, p2, p3, B, h, w) -> "SplitTensor":
    _require_gpu()
    outs = SplitTensor((B * h * w, 192), p1.device)
    _check(load().lbm_hip_upcat_s(_p(p1), _p(p2), _p(p3), None, outs.hi, outs.plane, B, h, w, _stream()), "lbm_hip_upcat_s")
    return outs


def encoder_tail(q1: torch.Tensor, q2: torch.Tensor, q3: torch.Tensor, pos: Optional[torch.Tensor], tokens: torch.Tensor,
                 tokens_s: Optional["SplitTensor"], h: int, w: int) -> None:
    """tokens level 0 = GELU(q1 + up2(q2) + up4(q3)) + pos, levels 1..3 pooled, split twin emitted (lbm_hip_encoder_tail)."""
    _require_gpu()
    _check(load().lbm_hip_encoder_tail(_p(q1), _p(q2), _p(q3), _p(pos), _p(tokens), tokens_s.hi if tokens_s is not None else None,
                                       tokens_s.plane if tokens_s is not None else 0, tokens.shape[0], h, w, _stream()),
           "lbm_hip_encoder_tail")


def encoder_tail_fused(f1: "SplitTensor", w1: "SplitWeight", b1: Optional[torch.Tensor], q2: torch.Tensor, q3: torch.Tensor,
                       pos: Optional[torch.Tensor], tokens: torch.Tensor, tokens_s: Optional["SplitTensor"], h: int, w: int,
                       inv_norm: Optional[torch.Tensor] = None) -> None:
    """encoder_tail with q1 = f1 . w1^T + b1 formed inside the kernel (lbm_hip_encoder_tail_fused); f1: split (B, h, w, 64).
    inv_norm (B, L): receives F.normalize's factor of every token row (the step's correlation maps then skip their norm pass)."""
    _require_gpu()
    assert w1.shape == (256, 64) and f1.shape[-1] == 64
    assert inv_norm is None or (inv_norm.shape == tuple(tokens.shape[:2]) and inv_norm.is_contiguous())
    _check(load().lbm_hip_encoder_tail_fused(f1.hi, f1.plane, w1.hi, w1.lo, 64, _p(b1), _p(q2), _p(q3), _p(pos), _p(tokens),
                                             tokens_s.hi if tokens_s is not None else None,
                                             tokens_s.plane if tokens_s is not None else 0, _p(inv_norm), tokens.shape[0], h, w, _stream()),
           "lbm_hip_encoder_tail_fused")


def pyramid_s(tokens: torch.Tensor, tokens_s: "SplitTensor", h: int, w: int) -> None:
    _require_gpu()
    _check(load().lbm_hip_pyramid_s(_p(tokens), tokens_s.hi, tokens_s.plane, tokens.shape[0], h, w, _stream()), "lbm_hip_pyramid_s")


def linear_ss_raw(**kw) -> None:
    """lbm_hip_linear_ss with the descriptor fields given by name (pointers as ints)."""
    _require_gpu()
    d = LinearSSDesc()
    for k, v in kw.items():
        setattr(d, k, v)
    _check(load().lbm_hip_linear_ss(C.byref(d), _stream()), "lbm_hip_linear_ss")


def linear_ss(a: "SplitWeight", b: "SplitWeight", bias=None, act: int = 0, res=None, res_scale: float = 1.0,
              want_f32: bool = True, want_split: bool = False, planes: int = 2):
    """y = act(a @ b.T + bias) + res_scale * res with both operands pre-split; returns fp32 and/or SplitWeight-like.
    planes=1 multiplies the hi planes only (plain f16 operands: the reduced-precision mode)."""
    _require_gpu()
    M, K = a.shape
    N = b.shape[0]
    d = LinearSSDesc()
    d.A, d.B, d.a_plane, d.b_plane = a.hi, b.hi, M * K, N * K
    d.bias, d.res = _p(bias), _p(res)
    y = torch.empty(M, N, device=a.planes.device, dtype=torch.float32) if want_f32 else None
    ys = torch.empty(2, M, N, device=a.planes.device, dtype=torch.float16) if want_split else None
    d.Y, d.Ys, d.ys_plane = _p(y), (ys.data_ptr() if ys is not None else None), M * N
    d.M, d.N, d.K, d.lda, d.ldb, d.ldy, d.ldys, d.ldres = M, N, K, K, K, N, N, N
    d.res_div, d.res_mod, d.res_scale, d.act, d.batch = 1, 0, res_scale, act, 1
    d.planes = planes
    _check(load().lbm_hip_linear_ss(C.byref(d), _stream()), "lbm_hip_linear_ss")
    return y, ys


def linear(x: torch.Tensor, w, bias: Optional[torch.Tensor] = None, act: int = 0,
           res: Optional[torch.Tensor] = None, res_div: int = 1, res_mod: int = 0, res_scale: float = 1.0,
           ln: Optional[Tuple[torch.Tensor, torch.Tensor]] = None, keep: Optional[torch.Tensor] = None,
           fallback: Optional[torch.Tensor] = None, want_normed: bool = False, out: Optional[torch.Tensor] = None):
    """y = act(x @ w.T + bias) + res_scale * res[(m // res_div) % res_mod]  (+ optional LayerNorm)."""
    _require_gpu()
    lib = load()
    split = isinstance(w, SplitWeight)
    assert x.dim() == 2 and len(w.shape) == 2 and x.shape[1] == w.shape[1]
    M, K = x.shape
    N = w.shape[0]
    y = out if out is not None else torch.empty(M, N, device=x.device, dtype=torch.float32)
    yn = torch.empty(M, N, device=x.device, dtype=torch.float32) if want_normed else None
    d = LinearDesc()
    d.X, d.bias, d.Y, d.res = _p(x), _p(bias), _p(y), _p(res)
    if split:
        d.W_hi, d.W_lo = w.hi, w.lo
    else:
        d.W = _p(w)
    d.M, d.N, d.K, d.ldx, d.ldw, d.ldy = M, N, K, K, K, N
    d.ldres = res.shape[1] if res is not None else 0
    d.res_div, d.res_mod, d.res_scale = res_div, res_mod, res_scale
    d.act, d.batch = act, 1
    if ln is not None:
        d.ln_gamma, d.ln_beta = _p(ln[0]), _p(ln[1])
        d.keep, d.fallback, d.Yn = _p(keep, torch.uint8), _p(fallback), _p(yn)
    _check(lib.lbm_hip_linear(C.byref(d), _stream()), "lbm_hip_linear")
    return (y, yn) if want_normed else y


def linear_batched(x: torch.Tensor, w: torch.Tensor, out: torch.Tensor, bias=None, act: int = 0,
                   res: Optional[torch.Tensor] = None, strides: Sequence[int] = (0, 0, 0, 0, 0),
                   M: int = 0, N: int = 0, K: int = 0, ldx: int = 0, ldw: int = 0, ldy: int = 0, ldres: int = 0,
                   batch: int = 1, res_mod: int = 0) -> None:
    """Raw strided/batched form (element strides sX, sW, sB, sY, sR)."""
    _require_gpu()
    d = LinearDesc()
    d.X, d.bias, d.Y, d.res = x.data_ptr(), _p(bias), out.data_ptr(), (res.data_ptr() if res is not None else None)
    if isinstance(w, SplitWeight):
        d.W_hi, d.W_lo = w.hi, w.lo
    else:
        d.W = w.data_ptr()
    d.M, d.N, d.K, d.ldx, d.ldw, d.ldy, d.ldres = M, N, K, ldx, ldw, ldy, ldres
    d.res_div, d.res_mod, d.res_scale, d.act, d.batch = 1, res_mod, 1.0, act, batch
    d.sX, d.sW, d.sB, d.sY, d.sR = strides
    _check(load().lbm_hip_linear(C.byref(d), _stream()), "lbm_hip_linear")


def layernorm(x, gamma, beta, keep=None, fallback=None, want_normed=False):
    _require_gpu()
    y = torch.empty_like(x)
    yn = torch.empty_like(x) if want_normed else None
    rows = x.numel() // 256
    _check(load().lbm_hip_layernorm(_p(x), _p(gamma), _p(beta), _p(y), rows, _p(keep, torch.uint8), _p(fallback), _p(yn),
                                    _stream()), "lbm_hip_layernorm")
    return (y, yn) if want_normed else y


def l2norm(x: torch.Tensor) -> torch.Tensor:
    _require_gpu()
    y = torch.empty_like(x)
    rows = x.numel() // 256
    _check(load().lbm_hip_l2norm(_p(x), 256, _p(y), 256, rows, _stream()), "lbm_hip_l2norm")
    return y


def attn_readout(U, seq, S, ring_head=0, temb=None, mask_a=None, mask_b=None, want_valid=False):
    """U (R,4,256); seq (R,S,256) -> (R,4,256) [and valid (R,) uint8]."""
    _require_gpu()
    R = U.shape[0]
    out = torch.empty(R, 4, 256, device=U.device, dtype=torch.float32)
    valid = torch.empty(R, device=U.device, dtype=torch.uint8) if want_valid else None
    h16 = seq.dtype == torch.float16           # the f16-storage mode's memories
    fn = load().lbm_hip_attn_readout_h16 if h16 else load().lbm_hip_attn_readout
    _check(fn(_p(U), _p(seq, torch.float16 if h16 else torch.float32), S * 256, 256, S, ring_head, _p(temb), _p(mask_a, torch.uint8),
              _p(mask_b, torch.uint8), _p(out), _p(valid, torch.uint8), R, _stream()),
           "lbm_hip_attn_readout")
    return (out, valid) if want_valid else out


def pyramid(tokens: torch.Tensor, h: int, w: int) -> None:
    _require_gpu()
    _check(load().lbm_hip_pyramid(_p(tokens), tokens.shape[0], h, w, _stream()), "lbm_hip_pyramid")


def corr_maps(tokens: torch.Tensor, corr_mix: Sequence[float], n_layers: int, h: int, w: int) -> torch.Tensor:
    _require_gpu()
    B, L, _ = tokens.shape
    inv = torch.empty(B, L, device=tokens.device, dtype=torch.float32)
    F = torch.empty(n_layers, B, h * w, 256, device=tokens.device, dtype=torch.float32)
    mix = (C.c_float * (n_layers * 4))(*[float(v) for v in corr_mix])
    _check(load().lbm_hip_corr_maps(_p(tokens), _p(inv), _p(F), mix, n_layers, B, h, w, _stream()), "lbm_hip_corr_maps")
    return F


def corr_topk(q: "SplitTensor", F: "SplitTensor", k: int, grid_w: int, stride_px: int = 4):
    """Correlation product + top-k in one pass (lbm_hip_corr_topk).  q: split planes (B, N, 256), F: split planes (B, P0, 256).
    Returns (idx int32 (B, N, k), cell centres (B, N, k, 2), scores (B, N, k))."""
    _require_gpu()
    B, N, _ = q.shape
    P0 = F.shape[1]
    dev = q.planes.device
    lib = load()
    nbytes = int(lib.lbm_hip_corr_topk_workspace_bytes(B, N, P0, k))
    if nbytes < 0:
        raise HipError("lbm_hip_corr_topk_workspace_bytes: shape not supported")
    ws = torch.empty(nbytes, device=dev, dtype=torch.uint8)
    idx = torch.empty(B, N, k, device=dev, dtype=torch.int32)
    ref = torch.empty(B, N, k, 2, device=dev, dtype=torch.float32)
    val = torch.empty(B, N, k, device=dev, dtype=torch.float32)
    _check(lib.lbm_hip_corr_topk(q.hi, q.plane, F.hi, F.plane, B, N, P0, k, grid_w, stride_px, ws.data_ptr(), nbytes,
                                 idx.data_ptr(), ref.data_ptr(), val.data_ptr(), _stream()), "lbm_hip_corr_topk")
    return idx, ref, val


def topk(corr: torch.Tensor, k: int, grid_w: int, stride_px: int = 4):
    """corr (rows, P) -> idx (rows,k) int32, ref (rows,k,2) float, val (rows,k)."""
    _require_gpu()
    rows, P = corr.shape
    idx = torch.empty(rows, k, device=corr.device, dtype=torch.int32)
    ref = torch.empty(rows, k, 2, device=corr.device, dtype=torch.float32)
    val = torch.empty(rows, k, device=corr.device, dtype=torch.float32)
    _check(load().lbm_hip_topk(_p(corr), P, P, k, rows, grid_w, stride_px, _p(idx, torch.int32), _p(ref), _p(val), _stream()),
           "lbm_hip_topk")
    return idx, ref, val


def _fill_levels(d: MsdaDesc, shapes: Sequence[Tuple[int, int]]) -> int:
    st = 0
    for l, (hl, wl) in enumerate(shapes):
        d.lvl_h[l], d.lvl_w[l], d.lvl_start[l] = hl, wl, st
        st += hl * wl
    return st


def msda_sample(value: torch.Tensor, shapes, offlog: torch.Tensor, ref: torch.Tensor, B: int, N: int, k: int, P: int,
                net_w: float, net_h: float, vcol: int = 0) -> torch.Tensor:
    """Implicit mode: value (B, L, ldv) [columns vcol..vcol+255 used], offlog (B*N, 48P), ref (B*N, k, 2) px."""
    _require_gpu()
    assert len(shapes) == 4
    d = MsdaDesc()
    L = _fill_levels(d, shapes)
    ldv = value.shape[-1]
    out = torch.empty(B * N * k, 256, device=value.device, dtype=torch.float32)
    h16 = value.dtype == torch.float16         # the f16-storage mode's value maps
    d.value = _p(value, value.dtype) + vcol * (2 if h16 else 4)
    d.value_f16 = 1 if h16 else 0
    d.v_batch_stride, d.ldv = L * ldv, ldv
    d.B, d.N, d.k, d.P = B, N, k, P
    d.offlog, d.ld_offlog, d.logit_col = _p(offlog), offlog.shape[-1], 32 * P
    d.ref, d.net_w, d.net_h = _p(ref), net_w, net_h
    d.out = _p(out)
    _check(load().lbm_hip_msda_sample(C.byref(d), _stream()), "lbm_hip_msda_sample")
    return out


def msda_sample_loc(value: torch.Tensor, shapes, loc: torch.Tensor, attw: torch.Tensor, out: torch.Tensor) -> None:
    """Explicit mode (generic operator): value (B,L,4,64), loc (B,Q,4,4,P,2), attw (B,Q,4,4,P), out (B,Q,256)."""
    _require_gpu()
    if len(shapes) != 4 or value.shape[2] != 4 or value.shape[3] != 64:
        raise HipError("the gfx950 MSDA kernel is specialised for 4 levels, 4 heads x 64 channels")
    B, L = value.shape[:2]
    Q, P = loc.shape[1], loc.shape[4]
    d = MsdaDesc()
    _fill_levels(d, shapes)
    d.value, d.v_batch_stride, d.ldv = _p(value), L * 256, 256
    d.B, d.N, d.k, d.P = B, Q, 1, P
    d.loc, d.attw, d.out = _p(loc), _p(attw), _p(out)
    _check(load().lbm_hip_msda_sample(C.byref(d), _stream()), "lbm_hip_msda_sample")


def msda_forward(mod, query, value, reference_points, shapes, identity) -> torch.Tensor:
    """MultiScaleDeformableAttention.forward (msdeformattn.py:142-185) for batch-first inputs."""
    _require_gpu()
    B, Q, Cc = query.shape
    L = value.shape[1]
    P = mod.num_points
    if Cc != 256 or mod.num_heads != 4 or mod.num_levels != 4:
        raise HipError("the gfx950 MSDA kernel is specialised for embed_dims=256, 4 heads, 4 levels")
    v = linear(value.reshape(B * L, Cc), mod.value_proj.weight.detach(), mod.value_proj.bias.detach()).view(B, L, 4, 64)
    q2 = query.reshape(B * Q, Cc)
    off = linear(q2, mod.sampling_offsets.weight.detach(), mod.sampling_offsets.bias.detach()).view(B, Q, 4, 4, P, 2)
    aw = linear(q2, mod.attention_weights.weight.detach(), mod.attention_weights.bias.detach()).view(B, Q, 4, 4 * P)
    aw = torch.softmax(aw, -1).view(B, Q, 4, 4, P).contiguous()
    norm = torch.tensor([[wl, hl] for (hl, wl) in shapes], device=query.device, dtype=torch.float32)
    loc = (reference_points[:, :, None, :, None, :] + off / norm[None, None, None, :, None, :]).contiguous()
    samp = torch.empty(B, Q, 256, device=query.device, dtype=torch.float32)
    msda_sample_loc(v, shapes, loc, aw, samp)
    out = linear(samp.view(B * Q, 256), mod.output_proj.weight.detach(), mod.output_proj.bias.detach(),
                 res=identity.reshape(B * Q, 256))
    return out.view(B, Q, 256)


U8_LAYOUTS = {"chw": 0, "hwc_rgb": 1, "hwc_bgr": 2}


def preprocess(frames: torch.Tensor, oh: int, ow: int, layout: int = 0, u8_layout: str = "chw") -> torch.Tensor:
    """frames (B,3,H,W) fp32 in [0,255] -> normalised image.  layout 0: (B,3,oh,ow); 3 / 4: pixel-major
    (B,oh,ow,layout) with channel 3 zero-filled (what the NHWC convolution stem reads).
    uint8 frames are read as they are (lbm_hip_preprocess_u8): (B,3,H,W) for u8_layout "chw", (B,H,W,3) for "hwc_rgb" /
    "hwc_bgr" (a decoder's interleaved output; BGR = OpenCV's order)."""
    _require_gpu()
    assert layout in (0, 3, 4)
    if frames.dtype == torch.uint8:
        code = U8_LAYOUTS[u8_layout]
        frames = frames.contiguous()
        if code == 0:
            B, Cc, H, W = frames.shape
        else:
            B, H, W, Cc = frames.shape
        assert Cc == 3
    else:
        B, Cc, H, W = frames.shape
        assert Cc == 3
        frames = frames.float().contiguous()
    shape = (B, 3, oh, ow) if layout == 0 else (B, oh, ow, layout)
    out = torch.empty(shape, device=frames.device, dtype=torch.float32)
    if frames.dtype == torch.uint8:
        _check(load().lbm_hip_preprocess_u8(frames.data_ptr(), 3 * H * W, code, out.data_ptr(), B, H, W, oh, ow, layout, _stream()),
               "lbm_hip_preprocess_u8")
    else:
        _check(load().lbm_hip_preprocess(frames.data_ptr(), 3 * H * W, out.data_ptr(), B, H, W, oh, ow, layout, _stream()),
               "lbm_hip_preprocess")
    return out


def preprocess_split(frames: torch.Tensor, oh: int, ow: int, u8_layout: str = "chw") -> "SplitTensor":
    """preprocess(..., layout=4) written as f16 hi / lo planes (B, oh, ow, 4): the input of stem_pool."""
    _require_gpu()
    is_u8 = frames.dtype == torch.uint8
    code = U8_LAYOUTS[u8_layout] if is_u8 else 0
    frames = frames.contiguous() if is_u8 else frames.float().contiguous()
    if code == 0:
        B, Cc, H, W = frames.shape
    else:
        B, H, W, Cc = frames.shape
    assert Cc == 3
    out = SplitTensor((B, oh, ow, 4), frames.device)
    _check(load().lbm_hip_preprocess_split(frames.data_ptr(), int(is_u8), 3 * H * W, code, out.hi, out.plane, B, H, W, oh, ow, _stream()),
           "lbm_hip_preprocess_split")
    return out


def stem_pool(xs: "SplitTensor", w: "SplitWeight", bias: torch.Tensor, want_f32: bool = True, want_split: bool = True):
    """conv 7 x 7 / 2 (4 -> 64 channels, BatchNorm folded) + ReLU + max-pool 3 x 3 / 2 in one launch; xs: split (B, H, W, 4)."""
    _require_gpu()
    B, H, W, Cc = xs.shape
    assert Cc == 4 and w.shape[0] == 64
    y = torch.empty(B, H // 4, W // 4, 64, device=xs.planes.device, dtype=torch.float32) if want_f32 else None
    ys = SplitTensor((B, H // 4, W // 4, 64), xs.planes.device) if want_split else None
    _check(load().lbm_hip_stem_pool(xs.hi, xs.plane, w.hi, w.lo, w.shape[1], _p(bias), _p(y), ys.hi if ys else None, ys.plane if ys else 0,
                                    B, H, W, _stream()), "lbm_hip_stem_pool")
    return y, ys


def upcat(p1, p2, p3, B, h, w) -> torch.Tensor:
    _require_gpu()
    out = torch.empty(B * h * w, 192, device=p1.device, dtype=torch.float32)
    _check(load().lbm_hip_upcat(_p(p1), _p(p2), _p(p3), _p(out), B, h, w, _stream()), "lbm_hip_upcat")
    return out


def sample_queries(tokens: torch.Tensor, pts: torch.Tensor, h: int, w: int, net_h: int, net_w: int) -> torch.Tensor:
    """tokens (B, rows>=h*w, 256) (a view with batch stride is fine), pts (B,N,2) network px -> (B,N,256)."""
    _require_gpu()
    B, N = pts.shape[:2]
    if tokens.stride(2) != 1 or tokens.stride(1) != 256:
        tokens = tokens.contiguous()
    out = torch.empty(B, N, 256, device=tokens.device, dtype=torch.float32)
    _check(load().lbm_hip_sample_queries(tokens.data_ptr(), tokens.stride(0), 256, _p(pts), _p(out), B, N, h, w, net_h, net_w,
                                         _stream()), "lbm_hip_sample_queries")
    return out


def conv2d_nhwc(x: torch.Tensor, w_packed: torch.Tensor, bias: torch.Tensor, kh: int, kw: int, stride: int, pad: int,
                relu: bool, res: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x (B,H,W,Cin) fp32; w_packed (Cout, kh*kw*Cin) tap-major; -> relu?(conv + bias (+ res)) as (B,Ho,Wo,Cout)."""
    _require_gpu()
    B, H, W, Cin = x.shape
    Cout = w_packed.shape[0]
    Ho = (H + 2 * pad - kh) // stride + 1
    Wo = (W + 2 * pad - kw) // stride + 1
    y = torch.empty(B, Ho, Wo, Cout, device=x.device, dtype=torch.float32)
    if isinstance(w_packed, SplitWeight):
        w_ptr, w_lo = w_packed.hi, w_packed.lo
    else:
        w_ptr, w_lo = _p(w_packed), None
    _check(load().lbm_hip_conv2d_nhwc(_p(x), w_ptr, _p(bias), _p(res), _p(y), w_lo, B, H, W, Cin, Cout, kh, kw, stride,
                                      pad, 1 if relu else 0, w_packed.shape[1], _stream()), "lbm_hip_conv2d_nhwc")
    return y


def maxpool3x3s2_nhwc(x: torch.Tensor) -> torch.Tensor:
    _require_gpu()
    B, H, W, Cc = x.shape
    Ho, Wo = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
    y = torch.empty(B, Ho, Wo, Cc, device=x.device, dtype=torch.float32)
    _check(load().lbm_hip_maxpool3x3s2_nhwc(_p(x), _p(y), B, H, W, Cc, _stream()), "lbm_hip_maxpool3x3s2_nhwc")
    return y
