"""ctypes binding of libt1d_hip.so (include/t1d.h).  There is no CPU fallback: if the HIP
library is missing or fails to load, every product path raises."""
import ctypes as C
import os
import subprocess

_PKG = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_PKG)
LIB_PATH = os.environ.get("T1D_LIB_PATH") or os.path.join(_PKG, "libt1d_hip.so")   # override: A/B builds only
SOURCES = [os.path.join(_PKG, "csrc", "t1d_abi.hip"), os.path.join(_PKG, "csrc", "t1d_kernels.hpp"),
           os.path.join(_PKG, "csrc", "t1d_device.hpp"),
           os.path.join(_ROOT, "include", "t1d.h")]

T1D_F64, T1D_F32 = 0, 1
T1D_ST_NORMALS_EXHAUSTED, T1D_ST_NONFINITE, T1D_ST_BAD_INDEX, T1D_ST_STALL = 1, 2, 4, 8
ABI_VERSION = 4
T1D_BATCH_NO_PUMP = 2
T1D_BATCH_NO_REFILL_DUE = 4
META_PLANNED = 0x200
P_NCOLS = 45
MEAL_UNUSED = 0x7FFFFFFF
META_EATING = 0x100

EXPORTS = ("t1d_abi_version", "t1d_last_error", "t1d_ctx_create", "t1d_ctx_set_option", "t1d_ctx_destroy", "t1d_reset",
           "t1d_step", "t1d_rollout_pid", "t1d_philox_normals", "t1d_sync", "t1d_split_tables",
           "t1d_rollout_bb", "t1d_random_meals", "t1d_outcome_stats", "t1d_model_rhs")


class T1DError(RuntimeError):
    pass


class Batch(C.Structure):
    """struct t1d_batch (include/t1d.h)"""
    _fields_ = [
        ("n", C.c_int64), ("env_offset", C.c_int64), ("dtype", C.c_int32), ("n_meals", C.c_int32),
        ("n_normals", C.c_int32), ("flags", C.c_int32), ("seed", C.c_uint64),
        ("x", C.c_void_p), ("planned", C.c_void_p), ("last_qsto", C.c_void_p), ("last_food", C.c_void_p),
        ("t", C.c_void_p), ("meta", C.c_void_p), ("episode", C.c_void_p), ("next_meal", C.c_void_p),
        ("last_cgm", C.c_void_p), ("ar_e", C.c_void_p), ("pts", C.c_void_p), ("prev_risk", C.c_void_p), ("dbar", C.c_void_p),
        ("basal", C.c_void_p), ("bolus", C.c_void_p), ("cho", C.c_void_p), ("meal_time", C.c_void_p),
        ("meal_amt", C.c_void_p), ("normals", C.c_void_p), ("x0_override", C.c_void_p),
        ("cgm", C.c_void_p), ("bg", C.c_void_p), ("reward", C.c_void_p), ("done", C.c_void_p),
        ("lbgi", C.c_void_p), ("hbgi", C.c_void_p), ("risk", C.c_void_p), ("meal", C.c_void_p),
        ("insulin", C.c_void_p), ("cgm0", C.c_void_p),
    ]


class Pid(C.Structure):
    """struct t1d_pid (include/t1d.h)"""
    _fields_ = [("P", C.c_double), ("I", C.c_double), ("D", C.c_double), ("target", C.c_double),
                ("integ", C.c_void_p), ("prev", C.c_void_p), ("sum_risk", C.c_void_p),
                ("min_bg", C.c_void_p), ("max_bg", C.c_void_p), ("n_low", C.c_void_p), ("n_high", C.c_void_p),
                ("bg_trace", C.c_void_p), ("cgm_trace", C.c_void_p), ("cho_trace", C.c_void_p),
                ("insulin_trace", C.c_void_p), ("trace_row", C.c_int64)]


class Bb(C.Structure):
    """struct t1d_bb (include/t1d.h)"""
    _fields_ = [("target", C.c_double), ("basal", C.c_void_p), ("cr", C.c_void_p), ("cf", C.c_void_p),
                ("prev_meal", C.c_void_p), ("sum_risk", C.c_void_p), ("min_bg", C.c_void_p), ("max_bg", C.c_void_p),
                ("n_low", C.c_void_p), ("n_high", C.c_void_p),
                ("bg_trace", C.c_void_p), ("cgm_trace", C.c_void_p), ("cho_trace", C.c_void_p),
                ("insulin_trace", C.c_void_p), ("trace_row", C.c_int64)]


class Outcome(C.Structure):
    """struct t1d_outcome (include/t1d.h)"""
    _fields_ = [("counts", C.c_void_p), ("pct", C.c_void_p), ("zone", C.c_void_p), ("risk_trace", C.c_void_p),
                ("q_lo", C.c_double), ("q_hi", C.c_double), ("chunk", C.c_int32)]


def _stale():
    return not os.path.exists(LIB_PATH) or any(os.path.exists(s) and os.path.getmtime(s) > os.path.getmtime(LIB_PATH)
                                               for s in SOURCES)


def build(force=False, verbose=False):
    """Compile libt1d_hip.so for gfx950 in-tree with hipcc (cross-compiles without a GPU).  Safe when several
    processes (one per GPU) find the library stale at once: one builds under a file lock into a temporary
    file that is moved into place, the others wait and then find it fresh."""
    import fcntl
    if not (force or _stale()):
        return LIB_PATH
    with open(LIB_PATH + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if force or _stale():
                hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
                tmp = "%s.tmp.%d" % (LIB_PATH, os.getpid())
                # -fno-slp-vectorize: in the fp32 kernels the SLP vectoriser pairs a fifth of the arithmetic into v_pk_*_f32
                # and pays more than it gains in the register moves that line the operands up (1 Mi envs fp32: 45.5 -> 42.4 us)
                cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-fno-slp-vectorize", "-std=c++17", "-shared", "-fPIC", "-o", tmp, SOURCES[0]]
                if verbose:
                    cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
                try:
                    subprocess.check_call(cmd)
                    os.replace(tmp, LIB_PATH)
                finally:
                    if os.path.exists(tmp):
                        os.remove(tmp)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB_PATH


_lib = None


def lib():
    """Load the HIP library (building it first if the sources are newer).  Raises T1DError when
    it cannot be produced or loaded -- the product never falls back to a CPU path."""
    global _lib
    if _lib is not None:
        return _lib
    try:
        if _stale():
            build()
        L = C.CDLL(LIB_PATH)
    except (OSError, subprocess.CalledProcessError, FileNotFoundError) as e:
        raise T1DError("libt1d_hip.so (HIP/gfx950 extension) is not available: %s" % e) from e
    vp, i32, i64, u64, u32 = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64, C.c_uint32
    dp = C.POINTER(C.c_double)
    L.t1d_abi_version.restype = C.c_int
    L.t1d_last_error.restype = C.c_char_p
    L.t1d_ctx_create.argtypes = [C.c_int, dp, C.c_int, C.c_int, dp, dp, C.POINTER(vp)]
    L.t1d_ctx_destroy.argtypes = [vp]
    L.t1d_ctx_set_option.argtypes = [vp, C.c_char_p, i64]
    L.t1d_reset.argtypes = [vp, C.POINTER(Batch), vp, C.c_int, vp]
    L.t1d_step.argtypes = [vp, C.POINTER(Batch), C.c_int, C.c_int, vp]
    L.t1d_rollout_pid.argtypes = [vp, C.POINTER(Batch), C.POINTER(Pid), C.c_int, C.c_int, C.c_int, vp]
    L.t1d_rollout_bb.argtypes = [vp, C.POINTER(Batch), C.POINTER(Bb), C.c_int, C.c_int, C.c_int, vp]
    L.t1d_random_meals.argtypes = [C.c_int, u64, i64, i64, C.c_int, C.c_int, vp, C.c_int, vp, vp, vp]
    L.t1d_outcome_stats.argtypes = [C.c_int, C.c_int, i64, i64, vp, C.POINTER(Outcome), vp]
    L.t1d_philox_normals.argtypes = [vp, u64, i64, i64, u32, i32, i32, vp, vp]
    L.t1d_model_rhs.argtypes = [vp, C.c_int, i64, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp]
    L.t1d_sync.argtypes = [vp, vp, C.POINTER(i32)]
    L.t1d_split_tables.argtypes = [dp, C.c_int, C.c_int, dp, C.c_int]
    for name in EXPORTS:
        if name not in ("t1d_last_error",):
            getattr(L, name).restype = C.c_int
    if L.t1d_abi_version() != ABI_VERSION:
        raise T1DError("libt1d_hip.so ABI version mismatch")
    _lib = L
    return L


def check(rc):
    if rc != 0:
        raise T1DError("t1d error %d: %s" % (rc, lib().t1d_last_error().decode()))
