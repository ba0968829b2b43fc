"""CPU suite, part 2: libt1d_hip.so loads without a GPU and exports every function include/t1d.h
declares; argument validation that needs no device; struct layout of the ctypes mirror."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    src = open(os.path.join(ROOT, "include", "t1d.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(t1d_[a-z_0-9]+)\s*\(", src)))


def test_library_loads_and_exports_every_declared_symbol():
    from simglucose_amd import _lib
    L = _lib.lib()
    names = _declared_functions()
    assert len(names) >= 10
    assert set(names) == set(_lib.EXPORTS), (names, _lib.EXPORTS)
    for n in names:
        assert hasattr(L, n), n
    assert L.t1d_abi_version() == _lib.ABI_VERSION == 4


def test_ctx_create_rejects_bad_arguments_without_touching_a_gpu():
    from simglucose_amd import _lib, params
    L = _lib.lib()
    names, tab = params.patient_table()
    tab = np.ascontiguousarray(tab)
    dp = C.POINTER(C.c_double)
    sen = params.sensor_row("Dexcom"); pump = params.pump_row("Insulet")
    ctx = C.c_void_p()

    def create(n_cols=45, npat=30, sensor=sen):
        return L.t1d_ctx_create(0, tab.ctypes.data_as(dp), npat, n_cols, sensor.ctypes.data_as(dp),
                                pump.ctypes.data_as(dp), C.byref(ctx))
    assert create(n_cols=44) == -1 and b"n_cols" in L.t1d_last_error()
    assert create(npat=0) == -1
    assert create(npat=65) == -1
    bad = sen.copy(); bad[5] = 2.5
    assert create(sensor=bad) == -1 and b"sample_time" in L.t1d_last_error()
    bad2 = sen.copy(); bad2[5] = 151.0
    assert create(sensor=bad2) == -1
    assert L.t1d_step(None, None, 1, 4, None) == -1
    assert L.t1d_sync(None, None, None) == -1
    import torch
    if not torch.cuda.is_available():
        rc = create()
        assert rc in (-3, -2), rc                 # no device: T1D_E_NODEVICE (or a HIP error), never success
        with pytest.raises(_lib.T1DError):
            from simglucose_amd.batch_env import BatchedT1DSimEnv
            BatchedT1DSimEnv(patient="adult#001", n_envs=4)     # product path fails loudly, no CPU fallback


def test_batch_struct_layout_matches_header():
    """Field order/offsets of the ctypes mirror follow struct t1d_batch in include/t1d.h."""
    from simglucose_amd import _lib
    src = open(os.path.join(ROOT, "include", "t1d.h")).read()
    body = src[src.index("typedef struct t1d_batch {"):src.index("} t1d_batch;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S).replace("typedef struct t1d_batch {", "")
    fields = []
    for stmt in body.split(";"):
        stmt = stmt.strip()
        if not stmt or stmt.startswith("typedef"):
            continue
        fields.append(re.findall(r"([A-Za-z_0-9]+)\s*$", stmt)[0])
    assert fields == [f[0] for f in _lib.Batch._fields_]
    assert C.sizeof(_lib.Batch) == 8 + 8 + 4 * 4 + 8 + 8 * (len(fields) - 7)
    body = src[src.index("typedef struct t1d_pid {"):src.index("} t1d_pid;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    pf = []
    for stmt in body.replace("typedef struct t1d_pid {", "").split(";"):
        for part in stmt.split(","):
            m = re.findall(r"([A-Za-z_0-9]+)\s*$", part.strip())
            if m and part.strip():
                pf.append(m[0])
    assert pf == [f[0] for f in _lib.Pid._fields_]
    body = src[src.index("typedef struct t1d_bb {"):src.index("} t1d_bb;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    bf = []
    for stmt in body.replace("typedef struct t1d_bb {", "").split(";"):
        for part in stmt.split(","):
            m = re.findall(r"([A-Za-z_0-9]+)\s*$", part.strip())
            if m and part.strip():
                bf.append(m[0])
    assert bf == [f[0] for f in _lib.Bb._fields_]


@pytest.mark.parametrize("n_sub", [2, 4, 6, 8])
def test_split_tables_match_independent_matrix_exponential(n_sub):
    """t1d_split_tables (host-only: the library's own scaling-and-squaring exp and series weights) against
    the oracle's tables (scipy.linalg.expm + Gauss-Legendre quadrature) for all 30 patients; entries the
    sparse device layout leaves out must be structural zeros of the dense propagator."""
    from simglucose_amd import _lib, params
    from oracle import t1d_oracle as O
    L = _lib.lib()
    names, tab = params.patient_table()
    dense = O.split_tables(tab, n_sub)
    dp = C.POINTER(C.c_double)
    nb = 2 * n_sub                                            # propagator blocks: tau = k / nb
    rows = 14 * nb + 21
    c6, c8 = [4, 0, 1, 2, 3, 7, 8], [6, 5, 0, 1, 2, 3, 7]
    c5, c7 = [0, 1, 2, 3, 7], [5, 0, 1, 2, 3, 7]
    for ip in range(len(names)):
        out = np.zeros(rows + 8)
        row = np.ascontiguousarray(tab[ip])
        assert L.t1d_split_tables(row.ctypes.data_as(dp), 45, n_sub, out.ctypes.data_as(dp), len(out)) == 0
        phi = dense[ip, :nb * 63].reshape(nb, 7, 9)
        want = np.zeros(rows)
        used = np.zeros((7, 9), bool)
        for k in range(nb):
            want[k * 14:k * 14 + 7] = phi[k, 4, c6]
            want[k * 14 + 7:k * 14 + 14] = phi[k, 6, c8]
        t = 14 * nb
        want[t:t + 5] = phi[-1, 0, c5]; want[t + 5:t + 10] = phi[-1, 1, c5]
        want[t + 10:t + 12] = phi[-1, 2, [2, 7]]; want[t + 12:t + 15] = phi[-1, 3, [2, 3, 7]]
        want[t + 15:t + 21] = phi[-1, 5, c7]
        for r, cols in ((4, c6), (6, c8), (0, c5), (1, c5), (2, [2, 7]), (3, [2, 3, 7]), (5, c7)):
            used[r, cols] = True
        assert np.abs(phi[:, ~used]).max() < 1e-14            # what the layout drops is zero
        scale = np.maximum(np.abs(want), 1e-3)
        assert (np.abs(out[:rows] - want) / scale).max() < 1e-12, names[ip]
        assert np.abs(out[rows:] - dense[ip, nb * 63:]).max() < 1e-14, names[ip]      # x2 weights for the gut steps of levels 1 and 2
    bad = np.zeros(10)
    assert L.t1d_split_tables(np.ascontiguousarray(tab[0]).ctypes.data_as(dp), 45, 3, bad.ctypes.data_as(dp), 10) != 0
    assert L.t1d_split_tables(np.ascontiguousarray(tab[0]).ctypes.data_as(dp), 45, 4, bad.ctypes.data_as(dp), 10) != 0
