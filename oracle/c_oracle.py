"""ctypes binding of oracle/_build/libemg_oracle.so (oracle/emg_oracle.c).  TEST INFRASTRUCTURE ONLY
(tests/, __graft_entry__.smoke(), bench.py cpu_baseline)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(_HERE, "_build", "libemg_oracle.so")
_lib = None


def build(force=False):
    if force or not os.path.exists(SO) or os.path.getmtime(SO) < os.path.getmtime(os.path.join(_HERE, "emg_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(SO):
            build()
        _lib = C.CDLL(SO)
        _lib.orc_chain_score.restype = C.c_float
        _lib.orc_num_threads.restype = C.c_int
    return _lib


def _f(a):
    return a.ctypes.data_as(C.c_void_p)


def _c(a, dt):
    a = np.ascontiguousarray(a, dtype=dt)
    return a


def num_threads():
    return int(lib().orc_num_threads())


def corrupt_codes(B, eta, side, n_choices, seed, counter, entities_list=None):
    codes = np.empty(B * eta, np.int32)
    el = _c(entities_list, np.int32) if entities_list is not None else None
    lib().orc_corrupt_codes(C.c_int64(B), C.c_int32(eta), C.c_int(side), C.c_int64(n_choices),
                            _f(el) if el is not None else None, C.c_uint64(seed), C.c_uint64(counter), _f(codes))
    return codes


def train_forward(model, ent, rel, k_int, scale, pos, eta, codes):
    ent, rel, pos = _c(ent, np.float32), _c(rel, np.float32), _c(pos, np.int32)
    B = pos.shape[0]
    sp = np.empty(B, np.float32)
    sn = np.empty(B * eta, np.float32)
    codes = _c(codes, np.int32) if eta else np.zeros(1, np.int32)
    lib().orc_train_forward(C.c_int(model), _f(ent), C.c_int64(ent.shape[1]), _f(rel), C.c_int64(rel.shape[1]),
                            C.c_int32(k_int), C.c_float(scale), _f(pos), C.c_int64(B), C.c_int32(eta), _f(codes),
                            _f(sp), _f(sn))
    return sp, sn


def build_queries(model, ent, rel, k_int, scale, test, side_mode):
    ent, rel, test = _c(ent, np.float32), _c(rel, np.float32), _c(test, np.int32)
    n_q = test.shape[0]
    n_rows = 2 * n_q if side_mode >= 2 else n_q
    Q = np.zeros((n_rows, k_int), np.float32)
    pos_int = np.empty(n_rows, np.int32)
    lib().orc_build_queries(C.c_int(model), _f(ent), C.c_int64(ent.shape[1]), _f(rel), C.c_int64(rel.shape[1]),
                            C.c_int32(k_int), _f(test), C.c_int64(n_q), C.c_int(side_mode), _f(Q), C.c_int64(k_int))
    lib().orc_pos_int(C.c_int(model), _f(ent), C.c_int64(ent.shape[1]), C.c_int32(k_int), C.c_float(scale), _f(test),
                      C.c_int64(n_q), C.c_int(side_mode), _f(Q), C.c_int64(k_int), _f(pos_int))
    return Q, pos_int


def scores_dense(model, Q, ent, k_int, scale, cand=None):
    Q, ent = _c(Q, np.float32), _c(ent, np.float32)
    cand_a = _c(cand, np.int32) if cand is not None else None
    n_cand = len(cand_a) if cand_a is not None else ent.shape[0]
    S = np.empty((Q.shape[0], n_cand), np.float32)
    lib().orc_scores_dense(C.c_int(model), _f(Q), C.c_int64(Q.shape[1]), C.c_int64(Q.shape[0]), _f(ent),
                           C.c_int64(n_cand), C.c_int64(ent.shape[1]), _f(cand_a) if cand_a is not None else None,
                           C.c_int32(k_int), C.c_float(scale), _f(S), C.c_int64(n_cand))
    return S


def count(model, Q, pos_int, ent, k_int, scale, cand=None):
    Q, ent, pos_int = _c(Q, np.float32), _c(ent, np.float32), _c(pos_int, np.int32)
    cand_a = _c(cand, np.int32) if cand is not None else None
    n_cand = len(cand_a) if cand_a is not None else ent.shape[0]
    gt = np.zeros(Q.shape[0], np.int32)
    eq = np.zeros(Q.shape[0], np.int32)
    lib().orc_count(C.c_int(model), _f(Q), C.c_int64(Q.shape[1]), _f(pos_int), C.c_int64(Q.shape[0]), _f(ent),
                    C.c_int64(n_cand), C.c_int64(ent.shape[1]), _f(cand_a) if cand_a is not None else None,
                    C.c_int32(k_int), C.c_float(scale), _f(gt), _f(eq))
    return gt, eq


def filter_count(model, Q, pos_int, ent, ent_offset, k_int, scale, fptr, fidx):
    Q, ent, pos_int = _c(Q, np.float32), _c(ent, np.float32), _c(pos_int, np.int32)
    fptr, fidx = _c(fptr, np.int64), _c(fidx, np.int32)
    gt = np.zeros(Q.shape[0], np.int32)
    eq = np.zeros(Q.shape[0], np.int32)
    lib().orc_filter_count(C.c_int(model), _f(Q), C.c_int64(Q.shape[1]), _f(pos_int), C.c_int64(Q.shape[0]), _f(ent),
                           C.c_int64(ent.shape[0]), C.c_int64(ent.shape[1]), C.c_int64(ent_offset), C.c_int32(k_int),
                           C.c_float(scale), _f(fptr), _f(fidx), _f(gt), _f(eq))
    return gt, eq


# ---- the CPU baseline of bench.py (oracle/emg_cpu_fast.c): optimised, not order-pinned -------------------------------
_fast = None
FAST_SO = os.path.join(_HERE, "_build", "libemg_cpufast.so")
FAST_NATIVE_SO = os.path.join(_HERE, "_build", "libemg_cpufast_native.so")


def fast_lib(native=True):
    """the optimised CPU baseline; ``native``: try to rebuild it with -march=native for THIS host first (a few seconds of
    gcc), falling back to the portable AVX2 + FMA build"""
    global _fast
    if _fast is None:
        so = FAST_SO
        if native:
            try:
                subprocess.check_call(["make", "-C", _HERE, "-s", "native"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
                so = FAST_NATIVE_SO
            except (subprocess.CalledProcessError, OSError):
                pass
        if so == FAST_SO and not os.path.exists(so):
            subprocess.check_call(["make", "-C", _HERE, "-s"])
        _fast = C.CDLL(so)
        _fast.cpufast_num_threads.restype = C.c_int
        _fast._so = so
    return _fast


def fast_train_forward(model, ent, rel, k_int, scale, pos, eta, codes, native=True):
    ent, rel, pos = _c(ent, np.float32), _c(rel, np.float32), _c(pos, np.int32)
    B = pos.shape[0]
    sp = np.empty(B, np.float32)
    sn = np.empty(B * eta, np.float32)
    codes = _c(codes, np.int32) if eta else np.zeros(1, np.int32)
    fast_lib(native).cpufast_train_forward(C.c_int(model), _f(ent), C.c_int64(ent.shape[1]), _f(rel), C.c_int64(rel.shape[1]),
                                           C.c_int32(k_int), C.c_float(scale), _f(pos), C.c_int64(B), C.c_int32(eta), _f(codes),
                                           _f(sp), _f(sn))
    return sp, sn
