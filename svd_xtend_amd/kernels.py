"""ctypes binding of libsvdx.so -- the only place the host code touches the HIP kernels.

Every function here is a 1:1 wrapper of an `extern "C"` entry declared in include/svdx.h: it passes raw
device pointers, sizes and the current HIP stream.  There is NO CPU fallback and no alternative
implementation in the product: if the shared library is missing, was not built for gfx950, or no GPU is
visible, the first call raises.  (tests/ can install an emulation backend through
`_set_backend_for_tests` to exercise the host-side orchestration on a CPU-only box; nothing in the
package does.)
"""
from __future__ import annotations

import ctypes
import os
from dataclasses import dataclass
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SVDX_LIB") or os.path.join(_HERE, "csrc", "libsvdx.so")   # SVDX_LIB: A/B a second build of the library

F16, BF16 = 0, 1
OUT_ACT, OUT_F32, OUT_F32_ATOMIC, OUT_F32_SLAB, OUT_F32_ADD = 0, 1, 2, 3, 4
EPI_NONE, EPI_GEGLU_FWD, EPI_GEGLU_BWD = 0, 1, 2
LN_PARTIAL_ROWS = 2048
GN_REPLICAS = 8
GN_STAT_FLOATS = 4             # floats of storage per (replica, sample, group) of a GroupNorm statistics buffer: two int64
GATHER_PLAIN, GATHER_CONV3X3, GATHER_CONV3X3_DGRAD2, GATHER_TEMPORAL3, GATHER_CONV3X3_PAD0 = 0, 1, 2, 3, 4
ABI_VERSION = 600              # include/svdx.h: SVDX_VERSION this binding was written against
OPT_STATE_FLOATS = 16          # include/svdx.h: layout of the optimizer / loss-scale / schedule state
SCHED_KINDS = {"constant": 0, "constant_with_warmup": 1, "linear": 2, "cosine": 3, "cosine_with_restarts": 4, "polynomial": 5, "piecewise_constant": 6}
SCHED_MAX_RULES = 8            # include/svdx.h SVDX_SCHED_MAX_RULES: step rules of piecewise_constant, stored behind the 16 state floats
OPT_STATE_ALLOC = 40           # floats the Trainer allocates: the 16 of the contract + 2 * 8 rule floats + the last multiplier, rounded up


class SvdxError(RuntimeError):
    pass


class _LinJobC(ctypes.Structure):            # include/svdx.h: svdx_lin_job
    _fields_ = [("X", ctypes.c_void_p), ("W", ctypes.c_void_p), ("bias", ctypes.c_void_p), ("Y", ctypes.c_void_p),
                ("N", ctypes.c_int), ("K", ctypes.c_int), ("ldw", ctypes.c_int), ("flags", ctypes.c_int)]


class _OuterJobC(ctypes.Structure):          # include/svdx.h: svdx_outer_job
    _fields_ = [("dY", ctypes.c_void_p), ("X", ctypes.c_void_p), ("dW", ctypes.c_void_p),
                ("N", ctypes.c_int), ("K", ctypes.c_int), ("scale", ctypes.c_float), ("reserved", ctypes.c_int)]


class _GradFinJobC(ctypes.Structure):        # include/svdx.h: svdx_gradfin_job
    _fields_ = [("acc", ctypes.c_void_p), ("dst", ctypes.c_void_p), ("colsum_slabs", ctypes.c_void_p), ("colsum_out", ctypes.c_void_p),
                ("slab_stride", ctypes.c_int64), ("count", ctypes.c_int64), ("nsplit", ctypes.c_int), ("colsum_n", ctypes.c_int),
                ("store", ctypes.c_int), ("reserved", ctypes.c_int), ("found_inf", ctypes.c_void_p)]


class _LnRedJobC(ctypes.Structure):          # include/svdx.h: svdx_lnred_job
    _fields_ = [("partial", ctypes.c_void_p), ("dgamma", ctypes.c_void_p), ("dbeta", ctypes.c_void_p),
                ("nblk", ctypes.c_int), ("C", ctypes.c_int)]


class _GatherC(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int) for n in
                ("mode", "n_img", "hi", "wi", "ho", "wo", "cin", "stride", "ups", "t", "hw", "lda")]


@dataclass(frozen=True)
class Gather:
    """Implicit-GEMM addressing of the A operand (include/svdx.h: svdx_gather)."""
    mode: int
    n_img: int = 0
    hi: int = 0
    wi: int = 0
    ho: int = 0
    wo: int = 0
    cin: int = 0
    stride: int = 1
    ups: int = 0
    t: int = 0
    hw: int = 0
    lda: int = 0

    def to_c(self) -> _GatherC:
        return _GatherC(self.mode, self.n_img, self.hi, self.wi, self.ho, self.wo, self.cin, self.stride,
                        self.ups, self.t, self.hw, self.lda)


# signature table: p void*, i int, f float, d double, l int64, z size_t
_SIGS = {
    "svdx_gemm": "ppp" "iiiiii" "p" "piii" "pi" "pp" "ifii" "ippi" "ip",
    "svdx_gemm_dual": "ppp" "iiiiii" "p" "piii" "pi" "pp" "ifi" "pp" "iiii" "ip",
    "svdx_gemm_gn": "ppp" "iiiiii" "p" "piii" "pi" "pp" "fi" "pii" "ip",
    "svdx_gemm_tn": "ppp" "iiiiii" "pp" "iii" "p" "ip",
    "svdx_gemm_finalize": "pil" "pi" "iii" "pp" "iii" "pi" "ppi" "ip",
    "svdx_gemm_finalize_gn": "pil" "p" "iii" "pp" "iii" "pi" "pii" "ip",
    "svdx_small_linear": "pppp" "iiii" "iii" "ip",
    "svdx_outer_acc": "ppp" "iii" "f" "p",
    "svdx_small_linear_batch": "p" "iii" "ip",
    "svdx_outer_acc_batch": "p" "ii" "p",
    "svdx_ln_param_reduce_batch": "p" "i" "p",
    "svdx_grad_finalize_batch": "p" "i" "p",
    "svdx_timestep_embed": "pp" "ii" "p",
    "svdx_gn_stats": "pp" "iiii" "i" "ip",
    "svdx_gn_apply": "ppppp" "iiii" "fi" "ip",
    "svdx_gn_bwd_stats": "pppppp" "iiii" "fi" "i" "ip",
    "svdx_gn_bwd_apply": "pppppppp" "iiii" "fi" "ip",
    "svdx_ln_fwd": "ppppp" "ii" "f" "ip",
    "svdx_ln_bwd": "pppppp" "f" "pppp" "iii" "ip",
    "svdx_attn_fwd": "ppppp" "iiiii" "f" "ip",
    "svdx_attn_bwd_prep": "ppp" "iii" "i" "ip",
    "svdx_attn_bwd_dkv": "pppppppp" "iiiiii" "f" "ip",
    "svdx_attn_bwd_dq": "ppppppp" "iiiiii" "f" "ip",
    "svdx_tattn_fwd": "pppp" "iiii" "ii" "f" "ip",
    "svdx_tattn_bwd": "ppppppp" "iiii" "iii" "f" "ip",
    "svdx_tsa_fwd": "ppp" "f" "ppp" "piii" "ppppp" "iiiii" "f" "ip",
    "svdx_geglu_fwd": "pp" "ii" "ip",
    "svdx_geglu_bwd": "ppp" "ii" "ip",
    "svdx_add": "ppp" "l" "ip",
    "svdx_blend": "pppp" "l" "ip",
    "svdx_blend_bwd": "pppp" "l" "ip",
    "svdx_add_rowvec": "ppp" "iiiii" "ip",
    "svdx_colsum": "pp" "iiiiiii" "p" "ip",
    "svdx_transpose": "pi" "pi" "ii" "ip",
    "svdx_concat2": "pi" "pi" "p" "i" "ip",
    "svdx_split2": "p" "pi" "pi" "i" "ip",
    "svdx_sum2x2": "pp" "iiii" "ip",
    "svdx_cast_from_f32": "pp" "l" "ip",
    "svdx_cast_transpose_from_f32": "pp" "ii" "ip",
    "svdx_nchw_to_rows": "pp" "iiiii" "f" "ip",
    "svdx_rows_to_nchw": "pp" "iiiii" "ip",
    "svdx_zero": "pzp",
    "svdx_patch_rows": "pp" "iiiiiiiiii" "i" "f" "ip",
    "svdx_softmax_rows": "pp" "iii" "ll" "f" "ip",
    "svdx_act_rows": "pp" "l" "i" "ip",
    "svdx_blur_axis": "pp" "iii" "p" "ii" "p",
    "svdx_bicubic_affine": "pp" "iiiiii" "pp" "p",
    "svdx_attn_small_fwd": "pp" "iiiii" "ll" "f" "ip",
    "svdx_stamp": "pp",
    "svdx_zero_spans": "pp" "ip",
    "svdx_edm_loss": "pi" "ppppp" "iiii" "pp" "ip",
    "svdx_check_finite": "plpp",
    "svdx_check_finite_spans": "pp" "i" "pp",
    "svdx_optim_prep": "p" "ffff" "ii" "p",
    "svdx_adamw": "pppp" "l" "dddddd" "pp" "iip",
    "svdx_adamw_tiled": "ppppp" "i" "dddddd" "ppp" "iip",
    "svdx_ema_lerp": "pp" "l" "f" "p",
    "svdx_allreduce_grads": "p" "ii" "l" "i" "p",
    "svdx_plan_begin": "",
    "svdx_plan_end": "p",
    "svdx_plan_replay": "pp",
    "svdx_plan_free": "p",
}
_CT = {"p": ctypes.c_void_p, "i": ctypes.c_int, "f": ctypes.c_float, "d": ctypes.c_double, "l": ctypes.c_int64, "z": ctypes.c_size_t}

EXPORTED_SYMBOLS = tuple(_SIGS) + ("svdx_wall_clock_khz", "svdx_version", "svdx_last_error", "svdx_device_ok", "svdx_tsa_pixels_per_band", "svdx_ln_bwd_blocks",
                                    "svdx_plan_launches", "svdx_plan_bytes")
PARAMS_F32, PARAMS_BF16_REFERENCE = 0, 1      # include/svdx.h: param_mode of svdx_adamw / svdx_adamw_tiled
TN_FLAT = 64                   # include/svdx.h SVDX_TN_FLAT: the staging of operands >= 2 GiB, on request (tests)
MAX_PEERS = 16                 # include/svdx.h SVDX_MAX_PEERS: ranks of svdx_allreduce_grads
BATCH_MAX_JOBS = 48            # include/svdx.h SVDX_BATCH_MAX_JOBS: jobs of a *_batch entry that share one launch
TSA_MAX_C, TSA_MAX_T, TSA_BAND_ROWS = 320, 16, 144


COLSUM_SLAB = 512
EDM_LOSS_SCRATCH = 1024        # floats of scratch svdx_edm_loss wants (per-workgroup partial sums)


def colsum_slabs(rows: int, rpg: int, mod: int) -> int:
    """Row slabs of svdx_colsum (its deterministic form wants float[slabs][n_groups][C] of scratch)."""
    maxcnt = -(-rows // mod) if mod else min(rpg, rows)
    return -(-maxcnt // COLSUM_SLAB)


def ln_bwd_blocks(rows: int, C: int) -> int:
    """include/svdx.h svdx_ln_bwd_blocks: partial rows the affine-gradient form of svdx_ln_bwd leaves in its scratch."""
    cc = C // 8
    lanes = 16 if cc <= 48 else (32 if cc <= 96 else 64)
    per, cap = 1, min(512, LN_PARTIAL_ROWS)                                          # csrc/norm.hip: rows per workgroup pass, cap on the partial rows
    return max(1, min(-(-rows // (per * (256 // lanes))), cap))


def tsa_pixels_per_band(T: int, HW: int) -> int:
    """include/svdx.h svdx_tsa_pixels_per_band: the largest divisor of HW with at most 144 band rows (0: unsupported T)."""
    if T <= 0 or T > TSA_MAX_T or HW <= 0:
        return 0
    return max((P for P in range(1, min(HW, TSA_BAND_ROWS // T) + 1) if HW % P == 0), default=0)


def load_library(path: str = LIB_PATH) -> ctypes.CDLL:
    """dlopen libsvdx.so and type every entry point.  Raises SvdxError when the library is absent."""
    if not os.path.exists(path):
        raise SvdxError(f"{path} not found: build it with `python __graft_entry__.py` "
                        "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    lib = ctypes.CDLL(path)
    for name, sig in _SIGS.items():
        fn = getattr(lib, name)
        fn.argtypes = [_CT[c] for c in sig]
        fn.restype = ctypes.c_int
    lib.svdx_tsa_pixels_per_band.argtypes = [ctypes.c_int, ctypes.c_int]
    lib.svdx_tsa_pixels_per_band.restype = ctypes.c_int
    lib.svdx_ln_bwd_blocks.argtypes = [ctypes.c_int, ctypes.c_int]
    lib.svdx_ln_bwd_blocks.restype = ctypes.c_int
    for name in ("svdx_plan_launches", "svdx_plan_bytes"):
        getattr(lib, name).argtypes = [ctypes.c_void_p]
        getattr(lib, name).restype = ctypes.c_int64
    lib.svdx_version.restype = ctypes.c_int
    lib.svdx_wall_clock_khz.restype = ctypes.c_int
    lib.svdx_device_ok.restype = ctypes.c_int
    lib.svdx_last_error.argtypes = [ctypes.c_char_p, ctypes.c_size_t]
    lib.svdx_last_error.restype = ctypes.c_int
    if lib.svdx_version() != ABI_VERSION:
        raise SvdxError(f"{path} was built from another revision of include/svdx.h (library {lib.svdx_version()}, binding {ABI_VERSION}): "
                        "rebuild it with `python __graft_entry__.py`")
    return lib


def _dt(t: torch.Tensor) -> int:
    if t.dtype == torch.float16:
        return F16
    if t.dtype == torch.bfloat16:
        return BF16
    raise SvdxError(f"activation dtype must be float16/bfloat16, got {t.dtype}")


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _f32(t: Optional[torch.Tensor]):
    if t is not None and t.dtype != torch.float32:
        raise SvdxError(f"expected float32 tensor, got {t.dtype}")
    return _p(t)


class LaunchPlan:
    """Handle of a recorded launch plan (svdx_plan_end).  `replay(stream)` re-issues its launches through svdx_plan_replay: ctypes -> C, no
    torch graph.  The tensors the recorded launches point into are the caller's to keep alive and in place."""

    def __init__(self, backend: "HipBackend", handle: int):
        self.be, self.handle = backend, handle

    @property
    def launches(self) -> int:
        return int(self.be.lib.svdx_plan_launches(self.handle))

    @property
    def host_bytes(self) -> int:
        return int(self.be.lib.svdx_plan_bytes(self.handle))

    def replay(self, stream: Optional[int] = None) -> None:
        self.be._call("svdx_plan_replay", self.handle, stream if stream is not None else self.be._stream())

    def free(self) -> None:
        if self.handle:
            self.be.lib.svdx_plan_free(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.free()
        except Exception:  # noqa: BLE001 -- interpreter shutdown
            pass


class HipBackend:
    """Calls into libsvdx.so on the current torch stream."""

    def __init__(self):
        self.lib = load_library()
        if not torch.cuda.is_available():
            raise SvdxError("no HIP device visible: svd_xtend_amd runs on MI355X only (no CPU fallback)")
        if not self.lib.svdx_device_ok():
            raise SvdxError("libsvdx.so reports no usable gfx950 device: " + self.last_error())
        self._zero_page = torch.zeros(1024, dtype=torch.uint8, device="cuda")
        self._log_extra = None
        self.n_calls = 0
        self.launch_log = None      # developer aid (bench.py --launch-log): a list that receives (entry, args) of every call, in order

    def last_error(self) -> str:
        buf = ctypes.create_string_buffer(512)
        self.lib.svdx_last_error(buf, 512)
        return buf.value.decode(errors="replace")

    def _call(self, name, *args):
        self.n_calls += 1                          # GraphedStep cuts a graph segment only where launches were captured since the last cut
        if self.launch_log is not None:
            self.launch_log.append((name, [a if isinstance(a, (int, float)) or a is None else "obj" for a in args], self._log_extra))
            self._log_extra = None
        rc = getattr(self.lib, name)(*args)
        if rc != 0:
            raise SvdxError(f"{name} failed ({rc}): {self.last_error()}")

    @staticmethod
    def _stream():
        return torch.cuda.current_stream().cuda_stream

    # ---- GEMM family ------------------------------------------------------------------------------
    def gemm(self, A, B, C, M, N, K, lda, ldb, ldc, bias=None, rowvec=None, rv_ld=0, rv_rpg=0, rv_mod=0,
             res=None, ldres=0, gather: Optional[Gather] = None, out_mode=OUT_ACT, alpha=1.0, split_k=1,
             variant=0, epilogue=EPI_NONE, aux_in=None, aux_out=None, aux_dim=0, dual=None, gn=None):
        """dual = (A2, B2, K2, lda2, ldb2[, a2_seg_n]): second operand pair reduced into the same accumulators (svdx_gemm_dual).
        gn = (stats, rows, cg): also leave the GroupNorm statistics of C in `stats` (svdx_gemm_gn; zeroed buffer, activation output)."""
        g = gather.to_c() if gather is not None else None
        if self.launch_log is not None and gather is not None:
            self._log_extra = [gather.mode, gather.cin, gather.stride, gather.ups]
        if gn is not None:
            assert dual is None and split_k == 1 and epilogue == EPI_NONE and out_mode == OUT_ACT
            self._call("svdx_gemm_gn", _p(A), _p(B), _p(C), M, N, K, lda, ldb, ldc, _f32(bias),
                       _f32(rowvec), rv_ld, rv_rpg, rv_mod, _p(res), ldres,
                       ctypes.cast(ctypes.pointer(g), ctypes.c_void_p) if g is not None else None,
                       _p(self._zero_page), float(alpha), variant, _f32(gn[0]), gn[1], gn[2], _dt(A), self._stream())
            return
        if dual is not None:
            A2, B2, K2, lda2, ldb2 = dual[:5]
            seg = dual[5] if len(dual) > 5 else 0
            assert split_k == 1 and epilogue == EPI_NONE and A2.dtype == A.dtype == B2.dtype
            self._call("svdx_gemm_dual", _p(A), _p(B), _p(C), M, N, K, lda, ldb, ldc, _f32(bias),
                       _f32(rowvec), rv_ld, rv_rpg, rv_mod, _p(res), ldres,
                       ctypes.cast(ctypes.pointer(g), ctypes.c_void_p) if g is not None else None,
                       _p(self._zero_page), out_mode, float(alpha), variant, _p(A2), _p(B2), K2, lda2, ldb2, seg, _dt(A), self._stream())
            return
        self._call("svdx_gemm", _p(A), _p(B), _p(C), M, N, K, lda, ldb, ldc, _f32(bias),
                   _f32(rowvec), rv_ld, rv_rpg, rv_mod, _p(res), ldres,
                   ctypes.cast(ctypes.pointer(g), ctypes.c_void_p) if g is not None else None,
                   _p(self._zero_page), out_mode, float(alpha), split_k, variant, epilogue, _p(aux_in), _p(aux_out), aux_dim,
                   _dt(A), self._stream())

    def gemm_tn(self, A, B, C, R, N, K, lda, ldb, ldc, out_mode=OUT_F32_ADD, split_k=1, a_colsum=None, stages=0, found_inf=None):
        """found_inf: one-float view (opt_state[3:4]) set to 1 when a value left in C is not finite (store / += modes)."""
        self._call("svdx_gemm_tn", _p(A), _p(B), _f32(C), R, N, K, lda, ldb, ldc, _f32(a_colsum), _p(self._zero_page), out_mode, split_k, stages,
                   _f32(found_inf), _dt(A), self._stream())

    def gemm_finalize(self, acc, nsplit, slab_stride, C, M, N, ldc, bias=None, rowvec=None, rv_ld=0, rv_rpg=0, rv_mod=0,
                      res=None, ldres=0, accumulate_f32=False, dtype=None, colsum_slabs=None, colsum_out=None, gn=None):
        """gn = (stats, rows, cg): also leave the GroupNorm statistics of C in `stats` (svdx_gemm_finalize_gn; activation output)."""
        dt = F16 if dtype == torch.float16 else BF16 if dtype == torch.bfloat16 else _dt(C)
        if gn is not None:
            assert not accumulate_f32 and colsum_slabs is None
            self._call("svdx_gemm_finalize_gn", _f32(acc), nsplit, slab_stride, _p(C), M, N, ldc, _f32(bias), _f32(rowvec), rv_ld, rv_rpg, rv_mod,
                       _p(res), ldres, _f32(gn[0]), gn[1], gn[2], dt, self._stream())
            return
        self._call("svdx_gemm_finalize", _f32(acc), nsplit, slab_stride, _p(C), int(accumulate_f32), M, N, ldc, _f32(bias),
                   _f32(rowvec), rv_ld, rv_rpg, rv_mod, _p(res), ldres, _f32(colsum_slabs), _f32(colsum_out),
                   colsum_out.numel() if colsum_out is not None else 0, dt, self._stream())

    def small_linear(self, X, W, bias, Y, M, N, K, ldw, trans=0, silu_in=0, accumulate=0):
        self._call("svdx_small_linear", _f32(X), _p(W), _f32(bias), _f32(Y), M, N, K, ldw, trans, silu_in,
                   accumulate, _dt(W), self._stream())

    def outer_acc(self, dY, X, dW, M, N, K, scale=1.0):
        self._call("svdx_outer_acc", _f32(dY), _f32(X), _f32(dW), M, N, K, float(scale), self._stream())

    def small_linear_batch(self, jobs, M, trans=0):
        """jobs: sequence of (X, W, bias, Y, N, K, ldw, silu_in, accumulate) -- each what `small_linear` takes; ONE launch per
        BATCH_MAX_JOBS jobs.  The jobs of a call must be independent of each other."""
        arr = (_LinJobC * len(jobs))(*[_LinJobC(_f32(X), _p(W), _f32(b), _f32(Y), N, Kd, ldw, int(bool(si)) | int(bool(acc)) << 1)
                                       for X, W, b, Y, N, Kd, ldw, si, acc in jobs])
        self._call("svdx_small_linear_batch", ctypes.cast(arr, ctypes.c_void_p), len(jobs), M, trans, _dt(jobs[0][1]), self._stream())

    def outer_acc_batch(self, jobs, M):
        """jobs: sequence of (dY, X or None (a column of ones: bias gradient, K = 1), dW, N, K, scale)."""
        arr = (_OuterJobC * len(jobs))(*[_OuterJobC(_f32(dY), _f32(X), _f32(dW), N, Kd, float(sc), 0) for dY, X, dW, N, Kd, sc in jobs])
        self._call("svdx_outer_acc_batch", ctypes.cast(arr, ctypes.c_void_p), len(jobs), M, self._stream())

    def grad_finalize_batch(self, jobs):
        """jobs: sequence of (slabs, nsplit, slab_stride, dst, count, colsum_slabs or None, colsum_out or None, store[, found_inf]) -- each
        what the float forms of `gemm_finalize` take (+ the one-float non-finite flag); ONE launch per BATCH_MAX_JOBS jobs, distinct destinations."""
        arr = (_GradFinJobC * len(jobs))(*[_GradFinJobC(_f32(j[0]), _f32(j[3]), _f32(j[5]), _f32(j[6]), j[2], j[4], j[1], j[6].numel() if j[6] is not None else 0,
                                                        int(bool(j[7])), 0, _f32(j[8]) if len(j) > 8 else None) for j in jobs])
        self._call("svdx_grad_finalize_batch", ctypes.cast(arr, ctypes.c_void_p), len(jobs), self._stream())

    def timestep_embed(self, t, out, n, dim):
        self._call("svdx_timestep_embed", _f32(t), _f32(out), n, dim, self._stream())

    # ---- norms ------------------------------------------------------------------------------------
    def gn_stats(self, x, stats, n_s, rows, C, G, prezeroed=0):
        self._call("svdx_gn_stats", _p(x), _f32(stats), n_s, rows, C, G, int(prezeroed), _dt(x), self._stream())

    def gn_apply(self, x, stats, gamma, beta, y, n_s, rows, C, G, eps, silu):
        self._call("svdx_gn_apply", _p(x), _f32(stats), _f32(gamma), _f32(beta), _p(y), n_s, rows, C, G,
                   float(eps), int(silu), _dt(x), self._stream())

    def gn_bwd_stats(self, dy, x, stats, gamma, beta, bstats, n_s, rows, C, G, eps, silu, prezeroed=0):
        self._call("svdx_gn_bwd_stats", _p(dy), _p(x), _f32(stats), _f32(gamma), _f32(beta), _f32(bstats),
                   n_s, rows, C, G, float(eps), int(silu), int(prezeroed), _dt(x), self._stream())

    def gn_bwd_apply(self, dy, x, stats, bstats, gamma, beta, add, dx, n_s, rows, C, G, eps, silu):
        self._call("svdx_gn_bwd_apply", _p(dy), _p(x), _f32(stats), _f32(bstats), _f32(gamma), _f32(beta),
                   _p(add), _p(dx), n_s, rows, C, G, float(eps), int(silu), _dt(x), self._stream())

    def ln_fwd(self, x, gamma, beta, y, stats, rows, C, eps):
        self._call("svdx_ln_fwd", _p(x), _f32(gamma), _f32(beta), _p(y), _f32(stats), rows, C, float(eps),
                   _dt(x), self._stream())

    def ln_bwd(self, dy, x, stats, gamma, add, dx, dgamma, dbeta, rows, C, scratch=None, add2=None, add2_scale=1.0, defer_reduce=False):
        """defer_reduce: scratch (>= ln_bwd_blocks(rows, C) * 2 * C floats) keeps the partial rows for `ln_param_reduce_batch`."""
        if scratch is not None:
            assert scratch.dtype == torch.float32 and scratch.numel() >= (ln_bwd_blocks(rows, C) if defer_reduce else LN_PARTIAL_ROWS) * 2 * C
        self._call("svdx_ln_bwd", _p(dy), _p(x), _f32(stats), _f32(gamma), _p(add), _p(add2), float(add2_scale), _p(dx),
                   _f32(dgamma), _f32(dbeta), _f32(scratch), rows, C, int(defer_reduce), _dt(x), self._stream())

    def ln_param_reduce_batch(self, jobs):
        """jobs: sequence of (partial, dgamma, dbeta, nblk, C): the deferred reductions of `ln_bwd(..., defer_reduce=True)`."""
        arr = (_LnRedJobC * len(jobs))(*[_LnRedJobC(_f32(pt), _f32(dg), _f32(db), nblk, C) for pt, dg, db, nblk, C in jobs])
        self._call("svdx_ln_param_reduce_batch", ctypes.cast(arr, ctypes.c_void_p), len(jobs), self._stream())

    # ---- attention --------------------------------------------------------------------------------
    def attn_fwd(self, q, k, v, o, lse, nb, heads, S, ld, ld_o, scale):
        self._call("svdx_attn_fwd", _p(q), _p(k), _p(v), _p(o), _f32(lse), nb, heads, S, ld, ld_o,
                   float(scale), _dt(q), self._stream())

    def attn_bwd_prep(self, o, d_o, D, nb, heads, S, ld_o):
        self._call("svdx_attn_bwd_prep", _p(o), _p(d_o), _f32(D), nb, heads, S, ld_o, _dt(o), self._stream())

    def attn_bwd_dkv(self, q, k, v, d_o, lse, D, dk, dv, nb, heads, S, ld, ld_o, ld_d, scale):
        self._call("svdx_attn_bwd_dkv", _p(q), _p(k), _p(v), _p(d_o), _f32(lse), _f32(D),
                   _p(dk), _p(dv), nb, heads, S, ld, ld_o, ld_d, float(scale), _dt(q), self._stream())

    def attn_bwd_dq(self, q, k, v, d_o, lse, D, dq, nb, heads, S, ld, ld_o, ld_d, scale):
        self._call("svdx_attn_bwd_dq", _p(q), _p(k), _p(v), _p(d_o), _f32(lse), _f32(D), _p(dq),
                   nb, heads, S, ld, ld_o, ld_d, float(scale), _dt(q), self._stream())

    def tattn_fwd(self, q, k, v, o, B, T, HW, heads, ld, ld_o, scale):
        self._call("svdx_tattn_fwd", _p(q), _p(k), _p(v), _p(o), B, T, HW, heads, ld, ld_o, float(scale),
                   _dt(q), self._stream())

    def tattn_bwd(self, q, k, v, d_o, dq, dk, dv, B, T, HW, heads, ld, ld_o, ld_d, scale):
        self._call("svdx_tattn_bwd", _p(q), _p(k), _p(v), _p(d_o), _p(dq), _p(dk), _p(dv), B, T, HW, heads,
                   ld, ld_o, ld_d, float(scale), _dt(q), self._stream())

    def tsa_fwd(self, x, gamma, beta, eps, wqkv, wo, bo, cvec, rv_ld, rv_rpg, rv_mod, n1, stats, qkv, o, h1, B, T, HW, C, heads, scale):
        self._call("svdx_tsa_fwd", _p(x), _f32(gamma), _f32(beta), float(eps), _p(wqkv), _p(wo), _f32(bo), _f32(cvec), rv_ld, rv_rpg,
                   rv_mod, _p(n1), _f32(stats), _p(qkv), _p(o), _p(h1), B, T, HW, C, heads, float(scale), _dt(x), self._stream())

    # ---- elementwise ------------------------------------------------------------------------------
    def geglu_fwd(self, pre, out, M, F):
        self._call("svdx_geglu_fwd", _p(pre), _p(out), M, F, _dt(pre), self._stream())

    def geglu_bwd(self, dout, pre, dpre, M, F):
        self._call("svdx_geglu_bwd", _p(dout), _p(pre), _p(dpre), M, F, _dt(pre), self._stream())

    def add(self, a, b, out, n):
        self._call("svdx_add", _p(a), _p(b), _p(out), n, _dt(a), self._stream())

    def blend(self, a, b, mix, out, n):
        self._call("svdx_blend", _p(a), _p(b), _f32(mix), _p(out), n, _dt(a), self._stream())

    def blend_bwd(self, dy, mix, da, db, n):
        self._call("svdx_blend_bwd", _p(dy), _f32(mix), _p(da), _p(db), n, _dt(dy), self._stream())

    def add_rowvec(self, x, vec, out, rows, C, rv_ld, rpg, mod):
        self._call("svdx_add_rowvec", _p(x), _f32(vec), _p(out), rows, C, rv_ld, rpg, mod, _dt(x), self._stream())

    def colsum(self, x, out, rows, C, ldx, n_groups, rpg, mod, accumulate=0, scratch=None):
        if scratch is not None:
            assert scratch.numel() >= colsum_slabs(rows, rpg, mod) * n_groups * C
        self._call("svdx_colsum", _p(x), _f32(out), rows, C, ldx, n_groups, rpg, mod, int(accumulate), _f32(scratch), _dt(x),
                   self._stream())

    def transpose(self, inp, ld_in, out, ld_out, rows, cols):
        self._call("svdx_transpose", _p(inp), ld_in, _p(out), ld_out, rows, cols, _dt(inp), self._stream())

    def concat2(self, a, Ca, b, Cb, out, rows):
        self._call("svdx_concat2", _p(a), Ca, _p(b), Cb, _p(out), rows, _dt(a), self._stream())

    def split2(self, inp, a, Ca, b, Cb, rows):
        self._call("svdx_split2", _p(inp), _p(a), Ca, _p(b), Cb, rows, _dt(inp), self._stream())

    def sum2x2(self, inp, out, n_img, h, w, C):
        self._call("svdx_sum2x2", _p(inp), _p(out), n_img, h, w, C, _dt(inp), self._stream())

    def cast_from_f32(self, inp, out, n):
        self._call("svdx_cast_from_f32", _f32(inp), _p(out), n, _dt(out), self._stream())

    def cast_transpose_from_f32(self, inp, out, R, Ccols):
        self._call("svdx_cast_transpose_from_f32", _f32(inp), _p(out), R, Ccols, _dt(out), self._stream())

    def nchw_to_rows(self, inp, out, n_img, C, H, W, ld, mul=1.0):
        self._call("svdx_nchw_to_rows", _f32(inp), _p(out), n_img, C, H, W, ld, float(mul), _dt(out), self._stream())

    def rows_to_nchw(self, inp, out, n_img, C, H, W, ld):
        self._call("svdx_rows_to_nchw", _p(inp), _f32(out), n_img, C, H, W, ld, _dt(inp), self._stream())

    def zero(self, t):
        self._call("svdx_zero", _p(t), t.numel() * t.element_size(), self._stream())

    # ---- frozen conditioners (VAE encoder, CLIP image tower) --------------------------------------
    def patch_rows(self, inp, out, n_img, C, H, W, kh, kw, stride, pad, ho, wo, ldk, mul=1.0):
        self._call("svdx_patch_rows", _f32(inp), _p(out), n_img, C, H, W, kh, kw, stride, pad, ho, wo, ldk, float(mul), _dt(out),
                   self._stream())

    def softmax_rows(self, inp, out, rows, cols, cols_out, ld_in, ld_out, scale):
        self._call("svdx_softmax_rows", _p(inp), _p(out), rows, cols, cols_out, ld_in, ld_out, float(scale), _dt(inp), self._stream())

    def act_rows(self, inp, out, n, act=0):
        self._call("svdx_act_rows", _p(inp), _p(out), n, int(act), _dt(inp), self._stream())

    def blur_axis(self, inp, out, planes, H, W, taps, axis):
        self._call("svdx_blur_axis", _f32(inp), _f32(out), planes, H, W, _f32(taps), taps.numel(), int(axis), self._stream())

    def bicubic_affine(self, inp, out, n_img, C, H, W, ho, wo, scale, shift):
        self._call("svdx_bicubic_affine", _f32(inp), _f32(out), n_img, C, H, W, ho, wo, _f32(scale), _f32(shift), self._stream())

    def attn_small_fwd(self, qkv, out, n_img, S, heads, d, dp, ld, ld_o, scale):
        self._call("svdx_attn_small_fwd", _p(qkv), _p(out), n_img, S, heads, d, dp, ld, ld_o, float(scale), _dt(qkv), self._stream())

    # ---- loss / optimizer -------------------------------------------------------------------------
    def edm_loss(self, pred, ld, noisy, target, sigma, loss, dpred, B, T, C, HW, opt_state, scratch=None):
        if scratch is None:
            scratch = torch.empty(EDM_LOSS_SCRATCH, dtype=torch.float32, device=pred.device)
        assert scratch.numel() >= EDM_LOSS_SCRATCH
        self._call("svdx_edm_loss", _p(pred), ld, _f32(noisy), _f32(target), _f32(sigma), _f32(loss),
                   _p(dpred), B, T, C, HW, _f32(opt_state), _f32(scratch), _dt(pred), self._stream())

    def check_finite(self, g, n, opt_state):
        self._call("svdx_check_finite", _f32(g), n, _f32(opt_state), self._stream())

    def ema_lerp(self, shadow, p, n, one_minus_decay):
        self._call("svdx_ema_lerp", _f32(shadow), _f32(p), n, float(one_minus_decay), self._stream())

    def allreduce_grads(self, peer_ptrs, rank, n, phase):
        """include/svdx.h svdx_allreduce_grads: peer_ptrs = device addresses of every rank's buffer (own included, index = rank)."""
        arr = (ctypes.c_void_p * len(peer_ptrs))(*peer_ptrs)
        self._call("svdx_allreduce_grads", ctypes.cast(arr, ctypes.c_void_p), len(peer_ptrs), int(rank), int(n), int(phase), self._stream())

    def check_finite_spans(self, g, spans, n_spans, opt_state):
        self._call("svdx_check_finite_spans", _f32(g), spans.data_ptr(), n_spans, _f32(opt_state), self._stream())

    def optim_prep(self, opt_state, beta1, beta2, growth, backoff, growth_interval, dynamic):
        assert opt_state.numel() >= OPT_STATE_FLOATS
        self._call("svdx_optim_prep", _f32(opt_state), float(beta1), float(beta2), float(growth),
                   float(backoff), int(growth_interval), int(dynamic), self._stream())

    def adamw(self, p, g, m, v, n, lr, beta1, beta2, eps, wd, grad_mul, opt_state, p_act, param_mode=PARAMS_F32):
        self._call("svdx_adamw", _f32(p), _f32(g), _f32(m), _f32(v), n, float(lr), float(beta1), float(beta2),
                   float(eps), float(wd), float(grad_mul), _f32(opt_state), _p(p_act), int(param_mode),
                   _dt(p_act) if p_act is not None else F16, self._stream())

    def stamp(self, slots, i):
        """slots: int64 device tensor; slots[i] = the device wall clock when the stream reaches this point (svdx_stamp)"""
        assert slots.dtype == torch.int64
        self._call("svdx_stamp", slots.data_ptr() + 8 * i, self._stream())

    def wall_clock_khz(self) -> int:
        return int(self.lib.svdx_wall_clock_khz())

    # ---- launch plans (include/svdx.h svdx_plan_*): the launches of one pass recorded on this thread, replayed from C ----
    def plan_begin(self) -> None:
        self._call("svdx_plan_begin")

    def plan_end(self) -> "LaunchPlan":
        h = ctypes.c_void_p()
        self._call("svdx_plan_end", ctypes.byref(h))
        return LaunchPlan(self, h.value)

    def zero_spans(self, base, spans, n_spans):
        assert spans.dtype == torch.int32 and spans.is_contiguous()
        self._call("svdx_zero_spans", _f32(base), spans.data_ptr(), n_spans, self._stream())

    def adamw_tiled(self, p, g, m, v, tiles, n_tiles, lr, beta1, beta2, eps, wd, grad_mul, opt_state, p_act, pt_act, param_mode=PARAMS_F32):
        assert tiles.dtype == torch.int32 and tiles.is_contiguous() and tiles.numel() >= 6 * n_tiles
        self._call("svdx_adamw_tiled", _f32(p), _f32(g), _f32(m), _f32(v), tiles.data_ptr(), n_tiles, float(lr), float(beta1),
                   float(beta2), float(eps), float(wd), float(grad_mul), _f32(opt_state), _p(p_act), _p(pt_act), int(param_mode),
                   _dt(p_act), self._stream())


_backend = None


def backend():
    """The HIP backend (created on first use).  Raises SvdxError when the extension or GPU is missing."""
    global _backend
    if _backend is None:
        _backend = HipBackend()
    return _backend


def _set_backend_for_tests(obj) -> None:
    """tests/ only: install an emulation of this interface so host logic can run without a GPU."""
    global _backend
    _backend = obj
