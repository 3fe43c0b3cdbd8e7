"""ctypes bindings of the parity checker (oracle/liboracle.so and, when built, oracle/_ref/libdcref.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by the
product package.  `liboracle.so` is our CPU restatement (oracle/dega_oracle.c); `libdcref.so` is the real reference
library compiled from /root/reference by oracle/Makefile (present only where it was built: this container, or a
GPU box that received the prebuilt file with the snapshot).
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "liboracle.so")
REF_SO = os.path.join(HERE, "_ref", "libdcref.so")
REF_CLI = os.path.join(HERE, "_ref", "DCCLI")

NO_ERROR = 0
ERROR_INVALID_VALUE = -1
ERROR_INVALID_FORMAT = -3
ERROR_MEMORY = -6
ERROR_LIBRARY_CALL = -11


def build(ref=True):
    """(Re)build the checker with oracle/Makefile (gcc). Building the checker is not using it."""
    targets = ["oracle"] + (["ref"] if ref else [])
    subprocess.run(["make", "-s", "-C", HERE] + targets, check=True)


class _Bits(C.Structure):
    _fields_ = [("data", C.POINTER(C.c_uint8)), ("nbits", C.c_size_t), ("cap_bytes", C.c_size_t)]


_lib = None
_ref = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(ORACLE_SO):
            build(ref=False)
        L = C.CDLL(ORACLE_SO)
        L.orc_dega_worst_case_bytes.restype = C.c_size_t
        L.orc_dega_worst_case_bytes.argtypes = [C.c_size_t]
        L.orc_dega_encode_i32.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_uint64)]
        L.orc_dega_decode_i32.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
        L.orc_dega_encode_f32.argtypes = [C.c_void_p, C.c_size_t, C.c_float, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_uint64)]
        L.orc_dega_decode_f32.argtypes = [C.c_void_p, C.c_uint64, C.c_float, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
        L.orc_dega_encode_batch_tc.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
        L.orc_dega_decode_batch_tc.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]
        for name in ("orc_normalize_encode", "orc_normalize_decode"):
            getattr(L, name).argtypes = [C.POINTER(_Bits), C.POINTER(_Bits), C.c_float, C.c_uint]
        for name in ("orc_diff_encode", "orc_diff_decode", "orc_seg_encode", "orc_seg_decode"):
            getattr(L, name).argtypes = [C.POINTER(_Bits), C.POINTER(_Bits), C.c_uint]
        for name in ("orc_lzmh_encode", "orc_lzmh_decode"):
            getattr(L, name).argtypes = [C.POINTER(_Bits), C.POINTER(_Bits)]
        for name in ("orc_bac_encode", "orc_bac_decode"):
            getattr(L, name).argtypes = [C.POINTER(_Bits), C.POINTER(_Bits), C.c_int]
        L.orc_bits_assign.argtypes = [C.POINTER(_Bits), C.c_void_p, C.c_size_t]
        L.orc_bits_init.argtypes = [C.POINTER(_Bits)]
        L.orc_bits_free.argtypes = [C.POINTER(_Bits)]
        _lib = L
    return _lib


def have_ref():
    return os.path.exists(REF_SO)


def ref():
    global _ref
    if _ref is None:
        R = C.CDLL(REF_SO)
        R.dcref_run_chain.restype = C.c_int64
        R.dcref_run_chain.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_char_p), C.c_size_t,
                                      C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.POINTER(C.c_double)]
        R.dcref_free.argtypes = [C.c_void_p]
        R.dcref_dega_encode_i32.restype = C.c_int64
        R.dcref_dega_encode_i32.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_uint64), C.POINTER(C.c_double)]
        R.dcref_dega_decode_i32.restype = C.c_int64
        R.dcref_dega_decode_i32.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t), C.POINTER(C.c_double)]
        _ref = R
    return _ref


# ---------------------------------------------------------------------------------------------------------------
# stage level (bit streams are (bytes, nbits) pairs)
# ---------------------------------------------------------------------------------------------------------------

def _stage(fn, data, nbits, *args):
    L = lib()
    a, b = _Bits(), _Bits()
    L.orc_bits_init(C.byref(a))
    L.orc_bits_init(C.byref(b))
    try:
        buf = (C.c_uint8 * max(1, len(data))).from_buffer_copy(bytes(data) if len(data) else b"\0")
        ret = L.orc_bits_assign(C.byref(a), buf, nbits)
        if ret != 0:
            return ret, b"", 0
        ret = fn(C.byref(a), C.byref(b), *args)
        if ret != 0:
            return ret, b"", 0
        n = b.nbits
        out = bytes(bytearray(b.data[: (n + 7) // 8])) if n else b""
        return 0, out, n
    finally:
        L.orc_bits_free(C.byref(a))
        L.orc_bits_free(C.byref(b))


def stage(name, encode, data, nbits, valuesize=32, adaptive=0, factor=100.0):
    """Run one reference stage of the restatement: name in {normalize, diff, seg, bac}. Returns (ret, bytes, nbits)."""
    L = lib()
    fn = getattr(L, "orc_%s_%s" % (name, "encode" if encode else "decode"))
    if name == "normalize":
        return _stage(fn, data, nbits, C.c_float(factor), C.c_uint(valuesize))
    if name == "bac":
        return _stage(fn, data, nbits, C.c_int(adaptive))
    if name == "lzmh":
        return _stage(fn, data, nbits)
    return _stage(fn, data, nbits, C.c_uint(valuesize))


def file_bytes(data, nbits):
    """What a file written by the reference holds for this stream: zero padded, one 0x00 byte when empty."""
    return bytes(data[: (nbits + 7) // 8]) if nbits else b"\0"


# ---------------------------------------------------------------------------------------------------------------
# whole chain, one channel
# ---------------------------------------------------------------------------------------------------------------

def encode_i32(x, adaptive=1):
    x = np.ascontiguousarray(x, dtype=np.int32)
    L = lib()
    cap = L.orc_dega_worst_case_bytes(x.size)
    out = np.zeros(cap, dtype=np.uint8)
    nbits = C.c_uint64(0)
    ret = L.orc_dega_encode_i32(x.ctypes.data, x.size, adaptive, out.ctypes.data, cap, C.byref(nbits))
    n = nbits.value
    return ret, (out[: (n + 7) // 8].tobytes() if ret == 0 else b""), n


def decode_i32(data, nbits, max_T, adaptive=1):
    L = lib()
    buf = np.frombuffer(bytes(data) + b"\0" * 8, dtype=np.uint8).copy()
    x = np.zeros(max(1, max_T), dtype=np.int32)
    got = C.c_size_t(0)
    ret = L.orc_dega_decode_i32(buf.ctypes.data, nbits, adaptive, x.ctypes.data, max_T, C.byref(got))
    return ret, x[: got.value].copy()


def encode_f32(v, factor=100.0, adaptive=1):
    v = np.ascontiguousarray(v, dtype=np.float32)
    L = lib()
    cap = L.orc_dega_worst_case_bytes(v.size)
    out = np.zeros(cap, dtype=np.uint8)
    nbits = C.c_uint64(0)
    ret = L.orc_dega_encode_f32(v.ctypes.data, v.size, factor, adaptive, out.ctypes.data, cap, C.byref(nbits))
    n = nbits.value
    return ret, (out[: (n + 7) // 8].tobytes() if ret == 0 else b""), n


def decode_f32(data, nbits, max_T, factor=100.0, adaptive=1):
    L = lib()
    buf = np.frombuffer(bytes(data) + b"\0" * 8, dtype=np.uint8).copy()
    v = np.zeros(max(1, max_T), dtype=np.float32)
    got = C.c_size_t(0)
    ret = L.orc_dega_decode_f32(buf.ctypes.data, nbits, factor, adaptive, v.ctypes.data, max_T, C.byref(got))
    return ret, v[: got.value].copy()


# ---------------------------------------------------------------------------------------------------------------
# batch, [T][C] layout
# ---------------------------------------------------------------------------------------------------------------

def encode_batch_tc(x_tc, adaptive=1, cap=None):
    """x_tc: int32 [T, C]. Returns (out uint8 [C, cap], bits uint64 [C], err int32 [C])."""
    x_tc = np.ascontiguousarray(x_tc, dtype=np.int32)
    T, Cn = x_tc.shape
    L = lib()
    if cap is None:
        cap = L.orc_dega_worst_case_bytes(T)
    out = np.zeros((Cn, cap), dtype=np.uint8)
    bits = np.zeros(Cn, dtype=np.uint64)
    err = np.zeros(Cn, dtype=np.int32)
    L.orc_dega_encode_batch_tc(x_tc.ctypes.data, Cn, T, Cn, adaptive, out.ctypes.data, cap, bits.ctypes.data, err.ctypes.data)
    return out, bits, err


def decode_batch_tc(out, bits, T, adaptive=1):
    out = np.ascontiguousarray(out, dtype=np.uint8)
    bits = np.ascontiguousarray(bits, dtype=np.uint64)
    Cn, cap = out.shape
    x = np.zeros((T, Cn), dtype=np.int32)
    err = np.zeros(Cn, dtype=np.int32)
    lib().orc_dega_decode_batch_tc(out.ctypes.data, cap, bits.ctypes.data, Cn, T, Cn, adaptive, x.ctypes.data, err.ctypes.data)
    return x, err


# ---------------------------------------------------------------------------------------------------------------
# the real reference (oracle/_ref)
# ---------------------------------------------------------------------------------------------------------------

def ref_run_chain(data, nbits, stages):
    """Run DCCLI-style stages ("encode diff", "encode bac adaptive", ...) through the compiled reference in memory.
    Returns (ret, bytes, nbits, per-stage seconds)."""
    R = ref()
    arr = (C.c_char_p * len(stages))(*[s.encode() for s in stages])
    buf = (C.c_uint8 * max(1, len(data))).from_buffer_copy(bytes(data) if len(data) else b"\0")
    out = C.c_void_p()
    onb = C.c_uint64(0)
    secs = (C.c_double * len(stages))()
    ret = R.dcref_run_chain(buf, nbits, arr, len(stages), C.byref(out), C.byref(onb), secs)
    res = b""
    if ret == 0 and out.value:
        res = C.string_at(out.value, (onb.value + 7) // 8)
    if out.value:
        R.dcref_free(out)
    return ret, res, onb.value, list(secs)


def ref_encode_i32(x, adaptive=1):
    x = np.ascontiguousarray(x, dtype=np.int32)
    R = ref()
    cap = lib().orc_dega_worst_case_bytes(x.size)
    out = np.zeros(cap, dtype=np.uint8)
    nbits = C.c_uint64(0)
    secs = (C.c_double * 3)()
    ret = R.dcref_dega_encode_i32(x.ctypes.data, x.size, adaptive, out.ctypes.data, cap, C.byref(nbits), secs)
    n = nbits.value
    return ret, (out[: (n + 7) // 8].tobytes() if ret == 0 else b""), n, list(secs)


def ref_decode_i32(data, nbits, max_T, adaptive=1):
    R = ref()
    buf = np.frombuffer(bytes(data) + b"\0" * 8, dtype=np.uint8).copy()
    x = np.zeros(max(1, max_T), dtype=np.int32)
    got = C.c_size_t(0)
    secs = (C.c_double * 3)()
    ret = R.dcref_dega_decode_i32(buf.ctypes.data, nbits, adaptive, x.ctypes.data, max_T, C.byref(got), secs)
    return ret, x[: got.value].copy(), list(secs)
