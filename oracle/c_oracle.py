"""TEST INFRASTRUCTURE ONLY: ctypes wrapper of oracle/c/librvq_oracle.so (plain-C RVQ restatement)."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "c")
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(HERE, "librvq_oracle.so")
        if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(os.path.join(HERE, "rvq_oracle.c")):
            subprocess.run(["make", "-s", "-C", HERE], check=True)
        _LIB = C.CDLL(path)
        _LIB.rvq_oracle_encode.restype = C.c_int
        _LIB.rvq_oracle_decode.restype = C.c_int
        _LIB.rvq_oracle_q0_source.restype = C.c_int
        _LIB.rvq_oracle_encode_src0.restype = C.c_int
    return _LIB


def rvq_encode(x: np.ndarray, cb: np.ndarray, nq: int):
    """x [N,D] f32, cb [n_q,K,D] f32 -> codes [nq,N] i64, quant [N,D] f32"""
    x = np.ascontiguousarray(x, np.float32)
    cb = np.ascontiguousarray(cb, np.float32)
    N, D = x.shape
    K = cb.shape[1]
    codes = np.empty((nq, N), np.int64)
    quant = np.empty((N, D), np.float32)
    rc = lib().rvq_oracle_encode(x.ctypes.data_as(C.c_void_p), C.c_int(N), C.c_int(D), C.c_int(K), C.c_int(nq),
                                 cb.ctypes.data_as(C.c_void_p), codes.ctypes.data_as(C.c_void_p),
                                 quant.ctypes.data_as(C.c_void_p))
    assert rc == 0
    return codes, quant


def q0_source(Tf: int) -> np.ndarray:
    """frame whose stage-0 result frame t receives under quantizer_conf.q0_ds_ratio > 1 (ddp_core_vq.py:396-404)"""
    src = np.empty(Tf, np.int32)
    rc = lib().rvq_oracle_q0_source(C.c_int(Tf), src.ctypes.data_as(C.c_void_p))
    assert rc == 0
    return src


def rvq_encode_q0(x: np.ndarray, cb: np.ndarray, nq: int, Tf: int):
    """x [B*Tf,D] rows of B utterances of Tf frames, first stage on the nearest-neighbour half-rate sequence"""
    x = np.ascontiguousarray(x, np.float32)
    cb = np.ascontiguousarray(cb, np.float32)
    N, D = x.shape
    assert N % Tf == 0
    src = (np.arange(N // Tf, dtype=np.int32)[:, None] * Tf + q0_source(Tf)[None]).reshape(-1).astype(np.int32)
    codes = np.empty((nq, N), np.int64)
    quant = np.empty((N, D), np.float32)
    rc = lib().rvq_oracle_encode_src0(x.ctypes.data_as(C.c_void_p), C.c_int(N), C.c_int(D), C.c_int(cb.shape[1]), C.c_int(nq),
                                      cb.ctypes.data_as(C.c_void_p), codes.ctypes.data_as(C.c_void_p),
                                      quant.ctypes.data_as(C.c_void_p), src.ctypes.data_as(C.c_void_p))
    assert rc == 0
    return codes, quant


def rvq_decode(codes: np.ndarray, cb: np.ndarray):
    """codes [N,nq] i64 -> emb [N,D]"""
    codes = np.ascontiguousarray(codes, np.int64)
    cb = np.ascontiguousarray(cb, np.float32)
    N, nq = codes.shape
    K, D = cb.shape[1], cb.shape[2]
    emb = np.empty((N, D), np.float32)
    rc = lib().rvq_oracle_decode(codes.ctypes.data_as(C.c_void_p), C.c_int(N), C.c_int(nq), C.c_int(D), C.c_int(K),
                                 cb.ctypes.data_as(C.c_void_p), emb.ctypes.data_as(C.c_void_p))
    assert rc == 0
    return emb
