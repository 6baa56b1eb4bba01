"""ctypes binding of libjchemo_hip.so (C ABI: include/jchemo_hip.h).

No CPU fallback: if the HIP library is missing or no gfx950 device is present, every compute entry
point raises — the product path never routes through oracle/ or numpy arithmetic for n-sized work.
"""
from __future__ import annotations

import ctypes as C
import os

try:  # torch (if present) must load its bundled HIP runtime BEFORE our library resolves libamdhip64.so.7,
    import torch  # noqa: F401  so that torch tensors and this library share one runtime instance.
except Exception:  # pragma: no cover
    torch = None

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "lib", "libjchemo_hip.so")

JCH_OK, JCH_EINVAL, JCH_EHIP, JCH_ERCCL, JCH_ENOMEM, JCH_ENODEV = 0, -1, -2, -3, -4, -5
LOC_HOST, LOC_DEVICE = 0, 1
F64, BF16 = 0, 1

SYMBOLS = (
    "jch_version", "jch_ctx_create", "jch_ctx_destroy", "jch_last_error", "jch_comm_unique_id",
    "jch_ctx_comm_init", "jch_ctx_comm_info", "jch_plskern_fit", "jch_plsnipals_fit", "jch_affine_gemm",
    "jch_weighted_ss", "jch_fill_uniform", "jch_ctx_set_profiling", "jch_ctx_get_profile", "jch_lwplsr_predict",
    "jch_weighted_cov", "jch_score_sums", "jch_score_sums_lv", "jch_plssimp_fit", "jch_plsrosa_fit", "jch_plswold_fit", "jch_transform",
    "jch_predict", "jch_loopback_group_create", "jch_loopback_group_destroy", "jch_ctx_comm_init_loopback",
    "jch_ctx_p2p_export", "jch_ctx_p2p_import", "jch_ctx_p2p_enable", "jch_plskern_fit_scaled", "jch_col_stats",
    "jch_ctx_get_counter", "jch_ctx_allreduce_probe", "jch_lwplsr_prepare", "jch_lwplsr_predict_prepared", "jch_lwplsr_release", "jch_lwplsr_add_query_map",
)


class JchError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libjchemo_hip error {code}: {msg}")
        self.code = code


class PlsDesc(C.Structure):
    _fields_ = [("n", C.c_int64), ("p", C.c_int64), ("q", C.c_int64), ("nlv", C.c_int32), ("scal", C.c_int32),
                ("dtype", C.c_int32), ("loc", C.c_int32), ("inplace", C.c_int32), ("reserved", C.c_int32)]


class Profile(C.Structure):
    _fields_ = [("fit_ms", C.c_double), ("prologue_ms", C.c_double), ("sweep_ms", C.c_double),
                ("smallstate_ms", C.c_double), ("sweep_launches", C.c_int32), ("nlv", C.c_int32),
                ("sweep_bytes", C.c_double), ("collective_ms", C.c_double), ("prologue_collective_ms", C.c_double),
                ("collective_wait_ms", C.c_double), ("collective_calls", C.c_int32), ("collective_transport", C.c_int32)]


TRANSPORT_NONE, TRANSPORT_RCCL, TRANSPORT_INBOX, TRANSPORT_INBOX_FUSED, TRANSPORT_LOOPBACK = 0, 1, 2, 3, 4
COUNTER_PIVOT_REFITS, COUNTER_LOCW_REFITS, COUNTER_KNN_SCREENED, COUNTER_KNN_SCREEN_REDONE, COUNTER_XCOPY_REUSED, COUNTER_SWEEPS_TIMED = range(6)   # include/jchemo_hip.h JCH_COUNTER_*
TRANSPORT_NAMES = {0: "none", 1: "rccl", 2: "inbox kernel", 3: "inbox fused into the small-state kernel", 4: "loopback (test harness)"}


_lib = None


def load():
    """Load the shared library (raises if it has not been built: `python -c 'import __graft_entry__ as g; g.build()'`)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise JchError(JCH_ENODEV, f"{LIB_PATH} not built; run __graft_entry__.build() (hipcc --offload-arch=gfx950)")
    L = C.CDLL(LIB_PATH)
    vp, i32, i64, dp = C.c_void_p, C.c_int32, C.c_int64, C.c_void_p  # double* passed as raw addresses
    L.jch_version.restype = i32
    L.jch_ctx_create.argtypes = [C.POINTER(vp), i32, vp, C.c_uint32]
    L.jch_ctx_destroy.argtypes = [vp]
    L.jch_last_error.argtypes = [vp]
    L.jch_last_error.restype = C.c_char_p
    L.jch_comm_unique_id.argtypes = [vp]
    L.jch_ctx_comm_init.argtypes = [vp, vp, i32, i32]
    L.jch_ctx_comm_info.argtypes = [vp, C.POINTER(i32), C.POINTER(i32)]
    L.jch_loopback_group_create.argtypes = [i32, C.POINTER(vp)]
    L.jch_loopback_group_destroy.argtypes = [vp]
    L.jch_ctx_comm_init_loopback.argtypes = [vp, vp, i32]
    L.jch_ctx_p2p_export.argtypes = [vp, i32, vp]
    L.jch_ctx_p2p_import.argtypes = [vp, vp, i32, i32, C.c_uint32]
    L.jch_ctx_p2p_enable.argtypes = [vp, i32]
    fit = [vp, C.POINTER(PlsDesc), dp, i64, dp, i64, dp] + [dp] * 11 + [C.POINTER(i32)]
    L.jch_plskern_fit.argtypes = fit
    L.jch_plsnipals_fit.argtypes = fit
    L.jch_plssimp_fit.argtypes = fit
    L.jch_plsrosa_fit.argtypes = fit
    L.jch_plswold_fit.argtypes = ([vp, C.POINTER(PlsDesc), dp, i64, dp, i64, dp, C.c_double, i32] + [dp] * 11 + [dp, C.POINTER(i32)])
    L.jch_plskern_fit_scaled.argtypes = [vp, C.POINTER(PlsDesc), dp, i64, dp, i64, dp, dp, dp] + [dp] * 11 + [C.POINTER(i32)]
    L.jch_col_stats.argtypes = [vp, i32, dp, i64, i64, i64, dp, dp, dp]
    L.jch_transform.argtypes = [vp, i32, dp, i64, i64, i64, dp, dp, dp, i32, dp, i64]
    L.jch_predict.argtypes = [vp, i32, dp, i64, i64, i64, dp, dp, dp, dp, dp, dp, i64, i32, i32, dp, i64]
    L.jch_affine_gemm.argtypes = [vp, i32, dp, i64, i64, i64, dp, dp, dp, i64, dp, dp, i64]
    L.jch_weighted_ss.argtypes = [vp, i32, dp, i64, i64, i64, dp, dp, dp, C.POINTER(C.c_double)]
    L.jch_lwplsr_predict.argtypes = [vp, i32, dp, i64, i64, i64, dp, i64, i64, dp, i64, dp, i64, i64, dp, i64, i64, i32,
                                     C.c_double, C.c_double, i32, i32, i32, dp, dp, dp, dp]
    L.jch_lwplsr_prepare.argtypes = [vp, i32, dp, i64, i64, i64, dp, i64, i64, dp, i64, i64, C.POINTER(vp)]
    L.jch_lwplsr_predict_prepared.argtypes = [vp, vp, i32, dp, i64, dp, i64, i64, i32, C.c_double, C.c_double, i32, i32, i32, dp, dp, dp, dp]
    L.jch_lwplsr_release.argtypes = [vp, vp]
    L.jch_lwplsr_add_query_map.argtypes = [vp, vp, dp, dp, dp, i64, i64, dp]
    L.jch_weighted_cov.argtypes = [vp, i32, dp, i64, i64, i64, dp, dp, dp]
    L.jch_score_sums.argtypes = [vp, i32, dp, i64, i64, i64, dp, i64, i64, dp, dp]
    L.jch_score_sums_lv.argtypes = [vp, i32, dp, i64, i64, i64, dp, dp, dp, dp, i64, i64, dp, i32, i32, dp]
    L.jch_fill_uniform.argtypes = [vp, dp, i64, i64, i64, i64, i64, C.c_uint64]
    L.jch_ctx_set_profiling.argtypes = [vp, i32]
    L.jch_ctx_get_profile.argtypes = [vp, C.POINTER(Profile)]
    L.jch_ctx_get_counter.argtypes = [vp, i32, C.POINTER(i64)]
    L.jch_ctx_allreduce_probe.argtypes = [vp, i32, dp, i64, i32, C.POINTER(C.c_double)]
    for name in SYMBOLS:
        if name != "jch_last_error":
            getattr(L, name).restype = i32
    _lib = L
    return L


class Context:
    """One GPU, one stream (jch_ctx).  `stream` = a hipStream_t handle (int) or None for a private stream;
    'torch' = torch's current stream on that device."""

    def __init__(self, device: int = 0, stream=None):
        L = load()
        h = C.c_void_p()
        if stream == "torch":
            if torch is None:
                raise JchError(JCH_EINVAL, "stream='torch' needs torch")
            stream = torch.cuda.current_stream(device).cuda_stream
        st = L.jch_ctx_create(C.byref(h), device, C.c_void_p(stream or 0), 0)
        if st != JCH_OK:
            raise JchError(st, L.jch_last_error(None).decode())
        self._h = h
        self.device = device
        self.rank, self.nranks = 0, 1

    def check(self, status: int):
        if status != JCH_OK:
            raise JchError(status, load().jch_last_error(self._h).decode())

    def close(self):
        if getattr(self, "_h", None):
            load().jch_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- multi-GPU: one process per GPU, ids exchanged by the host side (torch.distributed here)
    def comm_init(self, uid: bytes, rank: int, nranks: int):
        buf = C.create_string_buffer(uid, 128)
        self.check(load().jch_ctx_comm_init(self._h, buf, rank, nranks))
        self.rank, self.nranks = rank, nranks

    def comm_init_loopback(self, group, rank: int, nranks: int):
        """Test harness: ranks = threads of this process on one GPU (include/jchemo_hip.h, loopback communicator)."""
        self.check(load().jch_ctx_comm_init_loopback(self._h, group, rank))
        self.rank, self.nranks = rank, nranks

    # ---- P2P inbox transport (include/jchemo_hip.h): export -> exchange handles -> import (self-test) -> enable
    def p2p_export(self, nranks: int) -> bytes:
        buf = C.create_string_buffer(64)
        self.check(load().jch_ctx_p2p_export(self._h, nranks, buf))
        return buf.raw

    def p2p_import(self, handles, rank: int, nranks: int) -> bool:
        """handles: the nranks 64-byte IPC handles in rank order.  Collective call (maps the peers, runs the self-test);
        returns whether THIS rank's self-test passed — agree over all ranks before p2p_enable."""
        buf = C.create_string_buffer(b"".join(handles), 64 * nranks)
        st = load().jch_ctx_p2p_import(self._h, buf, rank, nranks, 0)
        if st == JCH_OK:
            self.rank, self.nranks = rank, nranks
        else:
            self.p2p_error = load().jch_last_error(self._h).decode()
        return st == JCH_OK

    def p2p_enable(self, on: bool):
        self.check(load().jch_ctx_p2p_enable(self._h, int(on)))

    def set_profiling(self, on):
        """on = True / False; an integer N > 1 samples the plskern-shaped sweeps: events around every N-th launch only
        (include/jchemo_hip.h jch_ctx_set_profiling)."""
        self.check(load().jch_ctx_set_profiling(self._h, int(on)))

    def counter(self, which: int = 0) -> int:
        """Diagnostic counter (0 = raw-mode fits repeated on the centred copy, include/jchemo_hip.h)."""
        v = C.c_int64(0)
        self.check(load().jch_ctx_get_counter(self._h, which, C.byref(v)))
        return int(v.value)

    def allreduce_probe(self, vec, transport: int = TRANSPORT_NONE, iters: int = 20):
        """Collective diagnostic (include/jchemo_hip.h jch_ctx_allreduce_probe): returns (sum over ranks of `vec`, average
        microseconds per all-reduce through `transport`)."""
        import numpy as np
        v = np.ascontiguousarray(np.asarray(vec, dtype=np.float64)).copy()
        us = C.c_double(0.0)
        self.check(load().jch_ctx_allreduce_probe(self._h, transport, v.ctypes.data, v.size, iters, C.byref(us)))
        return v, float(us.value)

    def profile(self) -> Profile:
        pr = Profile()
        self.check(load().jch_ctx_get_profile(self._h, C.byref(pr)))
        return pr


def unique_id() -> bytes:
    buf = C.create_string_buffer(128)
    st = load().jch_comm_unique_id(buf)
    if st != JCH_OK:
        raise JchError(st, load().jch_last_error(None).decode())
    return buf.raw


_default_ctx = {}


def default_context(device: int = 0) -> Context:
    if device not in _default_ctx:
        use_torch = torch is not None and torch.cuda.is_available()
        _default_ctx[device] = Context(device, stream="torch" if use_torch else None)
    return _default_ctx[device]
