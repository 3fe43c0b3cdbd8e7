"""data-compressor_amd -- MI355X-native DEGA encode/decode (normalize -> diff -> seg -> bac adaptive) of
CenterForSecureEnergyInformatics/data-compressor, behind a C ABI.

This Python module is only the thin ctypes binding of `libdega_hip.so` (include/dega_hip.h) that tests/ and bench.py use
to drive the library with torch-owned device memory and streams.  The product is the shared library (HIP kernels in
csrc/) and the C host layer in host/ (plugin table mirror of DCLib/inc/enc_dec.h, DCCLI-style driver).

There is NO CPU fallback: if the library is missing, or no GPU is visible, calls raise / return the reference's
ERROR_LIBRARY_INIT (-10).

The directory name contains a hyphen, so import it with
    import importlib.util, sys
    spec = importlib.util.spec_from_file_location("data_compressor_amd", "<repo>/data-compressor_amd/__init__.py")
or simply `from __graft_entry__ import load_package; dca = load_package()`.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DEGA_HIP_LIB", os.path.join(HERE, "libdega_hip.so"))  # override: diagnostic builds only
HEADER_PATH = os.path.join(os.path.dirname(HERE), "include", "dega_hip.h")

OK = 0
ERROR_INVALID_VALUE = -1
ERROR_INVALID_FORMAT = -3
ERROR_MEMORY = -6
ERROR_LIBRARY_INIT = -10
ERROR_LIBRARY_CALL = -11

_P = C.c_void_p
_Z = C.c_size_t

_SIGNATURES = {
    "dega_hip_device_count": (C.c_int, []),
    "dega_hip_create": (C.c_int, [C.c_int, C.POINTER(_P)]),
    "dega_hip_destroy": (None, [_P]),
    "dega_hip_last_error": (C.c_char_p, [_P]),
    "dega_hip_version": (C.c_char_p, []),
    "dega_hip_worst_case_bytes": (_Z, [_Z]),
    "dega_hip_encode_dev": (C.c_int, [_P, _P, _Z, _Z, _Z, C.c_int, C.c_int, _P, _Z, _P, _P, _P]),
    "dega_hip_decode_dev": (C.c_int, [_P, _P, _Z, _P, _Z, _Z, _Z, C.c_int, C.c_int, _P, _P, _P]),
    "dega_hip_decode_var_dev": (C.c_int, [_P, _P, _Z, _P, _Z, _Z, _Z, C.c_int, C.c_int, _P, _P, _P, _P]),
    "dega_hip_normalize_dev": (C.c_int, [_P, _P, _Z, _Z, _Z, C.c_float, C.c_int, _P, _P, _P]),
    "dega_hip_denormalize_dev": (C.c_int, [_P, _P, _Z, _Z, _Z, C.c_float, C.c_int, _P, _P]),
    "dega_hip_compact_offsets_dev": (C.c_int, [_P, _P, _Z, _P, _P]),
    "dega_hip_compact_gather_dev": (C.c_int, [_P, _P, _Z, _P, _Z, _P, _P]),
    "dega_hip_synth_dev": (C.c_int, [_P, _P, _Z, _Z, _Z, C.c_uint64, C.c_uint64, C.c_uint32, _P]),
    "dega_hip_encode_host": (C.c_int, [_P, _P, _Z, _Z, _Z, C.c_int, C.c_int, _P, _Z, _P, _P]),
    "dega_hip_encode_packed_host": (C.c_int, [_P, _P, _Z, _Z, _Z, C.c_int, C.c_int, _P, _Z, _P, _P, _P]),
    "dega_hip_decode_packed_host": (C.c_int, [_P, _P, _P, _P, _Z, _Z, _Z, C.c_int, C.c_int, _P, _P, _P]),
    "dega_hip_decode_host": (C.c_int, [_P, _P, _Z, _P, _Z, _Z, _Z, C.c_int, C.c_int, _P, _P]),
    "dega_hip_decode_var_host": (C.c_int, [_P, _P, _Z, _P, _Z, _Z, _Z, C.c_int, C.c_int, _P, _P, _P]),
    "dega_hip_encode_f32_host": (C.c_int, [_P, _P, _Z, _Z, _Z, C.c_float, C.c_int, C.c_int, _P, _Z, _P, _P]),
    "dega_hip_decode_f32_host": (C.c_int, [_P, _P, _Z, _P, _Z, _Z, _Z, C.c_float, C.c_int, C.c_int, _P, _P]),
    "dega_hip_decode_f32_var_host": (C.c_int, [_P, _P, _Z, _P, _Z, _Z, _Z, C.c_float, C.c_int, C.c_int, _P, _P, _P]),
    "dega_hip_worst_case_bytes64": (_Z, [_Z]),
    "dega_hip_encode64_dev": (C.c_int, [_P, _P, _Z, _Z, _Z, C.c_int, C.c_int, _P, _Z, _P, _P, _P]),
    "dega_hip_decode64_dev": (C.c_int, [_P, _P, _Z, _P, _Z, _Z, _Z, C.c_int, C.c_int, _P, _P, _P]),
    "dega_hip_decode64_var_dev": (C.c_int, [_P, _P, _Z, _P, _Z, _Z, _Z, C.c_int, C.c_int, _P, _P, _P, _P]),
    "dega_hip_encode64_host": (C.c_int, [_P, _P, _Z, _Z, _Z, C.c_int, C.c_int, _P, _Z, _P, _P]),
    "dega_hip_decode64_var_host": (C.c_int, [_P, _P, _Z, _P, _Z, _Z, _Z, C.c_int, C.c_int, _P, _P, _P]),
    "dega_hip_lzmh_worst_case_bytes": (_Z, [_Z]),
    "dega_hip_lzmh_encode_dev": (C.c_int, [_P, _P, _Z, _P, _Z, _P, _Z, _P, _P, _P]),
    "dega_hip_lzmh_decode_dev": (C.c_int, [_P, _P, _Z, _P, _Z, _P, _Z, _P, _P, _P]),
    "dega_hip_lzmh_render_dev": (C.c_int, [_P, _P, _Z, _Z, _Z, _P, _Z, _P, _P, _P]),
    "dega_hip_lzmh_encode_host": (C.c_int, [_P, _P, _Z, _P, _Z, _P, _Z, _P, _P]),
    "dega_hip_lzmh_decode_host": (C.c_int, [_P, _P, _Z, _P, _Z, _P, _Z, _P, _P]),
    "dega_hip_encode_f32_dev": (C.c_int, [_P, _P, _Z, _Z, _Z, C.c_float, C.c_int, C.c_int, _P, _Z, _P, _P, _P]),
    "dega_hip_decode_f32_dev": (C.c_int, [_P, _P, _Z, _P, _Z, _Z, _Z, C.c_float, C.c_int, C.c_int, _P, _P, _P, _P]),
    "dega_hip_encode_job_host": (C.c_int, [_P, _P, _P, _P, _Z, _P, _P, _P]),
    "dega_hip_decode_job_host": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P]),
    "dega_hip_split_channels": (C.c_int, [_Z, C.c_int, _P]),
    "dega_hip_group_create": (C.c_int, [_P, C.c_int, C.POINTER(_P)]),
    "dega_hip_group_destroy": (None, [_P]),
    "dega_hip_group_size": (C.c_int, [_P]),
    "dega_hip_group_context": (_P, [_P, C.c_int]),
    "dega_hip_group_last_error": (C.c_char_p, [_P]),
    "dega_hip_group_encode": (C.c_int, [_P, _P, _P, _P, _Z, _P, _P, _P]),
    "dega_hip_group_decode": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P]),
    "dega_hip_encode_state_bytes": (C.c_size_t, [C.c_size_t]),
    "dega_hip_encode_segment_dev": (C.c_int, [_P, _P, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int, C.c_int, _P, C.c_size_t, _P, _P, _P, C.c_uint, _P]),
    "dega_hip_group_lzmh_encode": (C.c_int, [_P, _P, C.c_size_t, _P, C.c_size_t, _P, C.c_size_t, _P, _P, _P]),
    "dega_hip_group_lzmh_decode": (C.c_int, [_P, _P, _P, _P, C.c_size_t, _P, C.c_size_t, _P, _P]),
    "dega_hip_pinned_alloc": (_P, [_Z]),
    "dega_hip_pinned_free": (None, [_P]),
    "dega_hip_profile": (C.c_int, [_P, C.c_int]),
    "dega_hip_profile_read": (C.c_int, [_P, C.c_int, C.POINTER(C.c_double), C.c_int]),
}

SAMPLES_I32, SAMPLES_BE32, SAMPLES_I64, SAMPLES_F32 = 0, 1, 2, 3


class Job(C.Structure):
    """dega_hip_job of include/dega_hip.h"""
    _fields_ = [("C", _Z), ("T", _Z), ("ld", _Z), ("adaptive", C.c_int), ("valuesize", C.c_int), ("samples", C.c_int), ("factor", C.c_float)]


_lib = None


class DegaError(RuntimeError):
    def __init__(self, code, what):
        super().__init__("%s failed with code %d" % (what, code))
        self.code = code


def library():
    """Load libdega_hip.so (once).  torch is imported first when available so that both share one HIP runtime."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise DegaError(ERROR_LIBRARY_INIT, "loading %s (not built: run __graft_entry__.build())" % LIB_PATH)
        try:
            import torch  # noqa: F401  (loads torch's libamdhip64 first; same SONAME, one runtime per process)
        except Exception:
            pass
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def exported_symbols():
    return sorted(_SIGNATURES)


def worst_case_bytes(T):
    return library().dega_hip_worst_case_bytes(T)


def lzmh_worst_case_bytes(n):
    return library().dega_hip_lzmh_worst_case_bytes(n)


def _sample_dtype(samples):
    import numpy as np
    return {SAMPLES_I32: np.dtype(np.int32), SAMPLES_BE32: np.dtype(">i4"), SAMPLES_I64: np.dtype(np.int64), SAMPLES_F32: np.dtype(np.float32)}[samples]


class _JobCalls:
    """encode_job / decode_job on numpy arrays: the packed host-pointer surface, shared by Context (one device) and Group
    (every device).  Subclasses provide _enc_fn / _dec_fn / _handle / _check."""

    def encode_job(self, x_tc, adaptive=1, valuesize=32, samples=SAMPLES_I32, factor=100.0, packed_cap=None, channels=None, packed=None):
        """x_tc: [T, ld] array of the sample type (channels = the first `channels` columns, default all).
        Returns (packed uint8 [total], offsets uint64 [C+1], bits uint64 [C], err int32 [C])."""
        import numpy as np
        if not (isinstance(x_tc, np.ndarray) and x_tc.flags.c_contiguous and x_tc.dtype == _sample_dtype(samples)):
            x_tc = np.ascontiguousarray(x_tc, dtype=_sample_dtype(samples))
        T, pitch = x_tc.shape
        Cn = pitch if channels is None else int(channels)
        job = Job(Cn, T, pitch, int(adaptive), int(valuesize), int(samples), float(factor))
        if packed_cap is None:
            packed_cap = Cn * (T * 2 + 64)  # generous for meter data; the call says so if it is not
        offsets = np.zeros(Cn + 1, dtype=np.uint64)
        bits = np.zeros(Cn, dtype=np.uint64)
        err = np.zeros(Cn, dtype=np.int32)
        ret = OK
        for _ in range(2):
            buf = packed if packed is not None and packed.size >= packed_cap else np.empty(max(1, packed_cap), dtype=np.uint8)
            ret = self._enc_fn()(self._handle(), C.byref(job), x_tc.ctypes.data, buf.ctypes.data, packed_cap, offsets.ctypes.data, bits.ctypes.data, err.ctypes.data)
            if ret != ERROR_MEMORY or int(offsets[Cn]) <= packed_cap:
                break
            packed_cap = int(offsets[Cn])
        self._check(ret, "encode_job")
        return buf[: int(offsets[Cn])], offsets, bits, err

    def decode_job(self, packed, offsets, bits, T, adaptive=1, valuesize=32, samples=SAMPLES_I32, factor=100.0, var=False, out=None):
        """The inverse.  Returns (x [T, C] of the sample type, err) or, with var=True, (x, counts, err)."""
        import numpy as np
        packed = np.ascontiguousarray(packed, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        bits = np.ascontiguousarray(bits, dtype=np.uint64)
        Cn = bits.size
        x = out if out is not None else np.zeros((T, Cn), dtype=_sample_dtype(samples))
        counts = np.zeros(Cn, dtype=np.uint64)
        err = np.zeros(Cn, dtype=np.int32)
        job = Job(Cn, T, x.shape[1] if x.ndim == 2 else Cn, int(adaptive), int(valuesize), int(samples), float(factor))
        ret = self._dec_fn()(self._handle(), C.byref(job), packed.ctypes.data, offsets.ctypes.data, bits.ctypes.data, x.ctypes.data,
                             counts.ctypes.data if var else None, err.ctypes.data)
        self._check(ret, "decode_job")
        return (x, counts, err) if var else (x, err)


class PinnedArray:
    """numpy view of pinned host memory (dega_hip_pinned_alloc): `.array`; `.free()` when done."""

    def __init__(self, shape, dtype):
        import numpy as np
        count = int(np.prod(shape))
        n = max(1, count * np.dtype(dtype).itemsize)
        self._p = library().dega_hip_pinned_alloc(n)
        if not self._p:
            raise DegaError(ERROR_MEMORY, "dega_hip_pinned_alloc(%d)" % n)
        self._buf = (C.c_uint8 * n).from_address(self._p)
        self.array = np.frombuffer(self._buf, dtype=dtype, count=count).reshape(shape)

    def free(self):
        if self._p:
            self.array = None
            self._buf = None
            library().dega_hip_pinned_free(self._p)
            self._p = None


class Group(_JobCalls):
    """dega_hip_group: every visible GPU (or `devices`), channel ranges per device, host-side concatenate."""

    def __init__(self, devices=None):
        self._h = _P()
        if devices:
            arr = (C.c_int * len(devices))(*devices)
            ret = library().dega_hip_group_create(arr, len(devices), C.byref(self._h))
        else:
            ret = library().dega_hip_group_create(None, 0, C.byref(self._h))
        if ret != OK:
            self._h = None
            raise DegaError(ret, "dega_hip_group_create")

    def size(self):
        return library().dega_hip_group_size(self._h)

    def close(self):
        if self._h:
            library().dega_hip_group_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _handle(self):
        return self._h

    def _enc_fn(self):
        return library().dega_hip_group_encode

    def _dec_fn(self):
        return library().dega_hip_group_decode

    def _check(self, ret, what):
        if ret != OK:
            raise DegaError(ret, "%s [%s]" % (what, library().dega_hip_group_last_error(self._h).decode()))

    def lzmh_encode_job(self, text, lens, packed=None):
        """text: uint8 [C, stride] host array (stride a multiple of 16; numpy or a PinnedArray's .array), lens: bytes per
        channel.  Returns (packed uint8, offsets uint64 [C + 1], bits uint64 [C], err int32 [C]); channel c's stream is
        packed[offsets[c]:offsets[c + 1]].  The host pipeline on every device of the group (dega_hip_group_lzmh_encode)."""
        import numpy as np
        Cn, stride = text.shape
        lens = np.ascontiguousarray(lens, dtype=np.uint64)
        if packed is None:
            packed = np.empty(int(lens.sum()) * 5 // 4 + 64 * Cn + 64, dtype=np.uint8)
        offsets = np.zeros(Cn + 1, dtype=np.uint64)
        bits = np.zeros(Cn, dtype=np.uint64)
        err = np.zeros(Cn, dtype=np.int32)
        ret = library().dega_hip_group_lzmh_encode(self._h, text.ctypes.data, stride, lens.ctypes.data, Cn, packed.ctypes.data, packed.size,
                                                   offsets.ctypes.data, bits.ctypes.data, err.ctypes.data)
        self._check(ret, "dega_hip_group_lzmh_encode")
        return packed, offsets, bits, err

    def lzmh_decode_job(self, packed, offsets, bits, stride, out=None):
        """The inverse: (text uint8 [C, stride], lens uint64 [C], err int32 [C])."""
        import numpy as np
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        bits = np.ascontiguousarray(bits, dtype=np.uint64)
        Cn = bits.size
        if out is None:
            out = np.zeros((Cn, stride), dtype=np.uint8)
        lens = np.zeros(Cn, dtype=np.uint64)
        err = np.zeros(Cn, dtype=np.int32)
        ret = library().dega_hip_group_lzmh_decode(self._h, packed.ctypes.data, offsets.ctypes.data, bits.ctypes.data, Cn, out.ctypes.data, stride,
                                                   lens.ctypes.data, err.ctypes.data)
        self._check(ret, "dega_hip_group_lzmh_decode")
        return out, lens, err


class Context(_JobCalls):
    """One device context (dega_hip_ctx).  Methods take torch CUDA tensors and enqueue on torch's current stream."""

    def _handle(self):
        return self._h

    def _enc_fn(self):
        return library().dega_hip_encode_job_host

    def _dec_fn(self):
        return library().dega_hip_decode_job_host

    def __init__(self, device=0):
        self._h = _P()
        self.device = device
        ret = library().dega_hip_create(device, C.byref(self._h))
        if ret != OK:
            self._h = None
            raise DegaError(ret, "dega_hip_create(device=%d)" % device)

    def close(self):
        if self._h:
            library().dega_hip_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def last_error(self):
        return library().dega_hip_last_error(self._h).decode()

    def _check(self, ret, what):
        if ret != OK:
            raise DegaError(ret, "%s [%s]" % (what, self.last_error()))

    @staticmethod
    def _stream():
        import torch
        return _P(torch.cuda.current_stream().cuda_stream)

    # ---- device-resident API (torch tensors) ---------------------------------------------------------------------
    def encode(self, x_tc, adaptive=1, cap=None, out=None, bits=None, err=None, valuesize=32):
        """x_tc: int32 CUDA tensor [T, ld>=C].  Returns (out uint8 [C, cap], bits int64 [C] (bit lengths), err int32 [C])."""
        import torch
        T, ld = x_tc.shape
        Cn = ld
        assert x_tc.dtype == torch.int32 and x_tc.is_cuda and x_tc.is_contiguous()
        if cap is None:
            cap = worst_case_bytes(T)
        if out is None:
            out = torch.zeros((Cn, cap), dtype=torch.uint8, device=x_tc.device)
        if bits is None:
            bits = torch.zeros(Cn, dtype=torch.int64, device=x_tc.device)
        if err is None:
            err = torch.zeros(Cn, dtype=torch.int32, device=x_tc.device)
        ret = library().dega_hip_encode_dev(self._h, x_tc.data_ptr(), Cn, T, ld, int(adaptive), int(valuesize), out.data_ptr(), cap,
                                            bits.data_ptr(), err.data_ptr(), self._stream())
        self._check(ret, "dega_hip_encode_dev")
        return out, bits, err

    def encode_segments(self, x_tc, cuts, adaptive=1, cap=None, valuesize=32):
        """The rows of x_tc ([T, C] int32 CUDA tensor) coded in ranges [cuts[k], cuts[k+1]), one launch each, the lanes' state
        kept in device memory in between (dega_hip_encode_segment_dev).  Returns (out, bits, err) of the whole channels."""
        import torch
        T, Cn = x_tc.shape
        assert x_tc.dtype == torch.int32 and x_tc.is_cuda and x_tc.is_contiguous() and cuts[0] == 0 and cuts[-1] == T
        if cap is None:
            cap = worst_case_bytes(T)
        out = torch.zeros((Cn, cap), dtype=torch.uint8, device=x_tc.device)
        bits = torch.zeros(Cn, dtype=torch.int64, device=x_tc.device)
        err = torch.zeros(Cn, dtype=torch.int32, device=x_tc.device)
        state = torch.zeros(library().dega_hip_encode_state_bytes(Cn), dtype=torch.uint8, device=x_tc.device)
        for k in range(len(cuts) - 1):
            flags = (1 if k > 0 else 0) | (2 if k + 2 < len(cuts) else 0)
            rows = x_tc[cuts[k]:cuts[k + 1]]
            ptr = rows.data_ptr() if cuts[k + 1] > cuts[k] else x_tc.data_ptr()
            ret = library().dega_hip_encode_segment_dev(self._h, ptr, Cn, cuts[k + 1] - cuts[k], Cn, int(adaptive), int(valuesize), out.data_ptr(), cap,
                                                        bits.data_ptr(), err.data_ptr(), state.data_ptr(), flags, self._stream())
            self._check(ret, "dega_hip_encode_segment_dev")
        return out, bits, err

    def decode(self, streams, bits, T, adaptive=1, x_tc=None, err=None, valuesize=32):
        import torch
        Cn, cap = streams.shape
        assert streams.dtype == torch.uint8 and streams.is_cuda and streams.is_contiguous() and bits.dtype == torch.int64
        if x_tc is None:
            x_tc = torch.zeros((T, Cn), dtype=torch.int32, device=streams.device)
        if err is None:
            err = torch.zeros(Cn, dtype=torch.int32, device=streams.device)
        ret = library().dega_hip_decode_dev(self._h, streams.data_ptr(), cap, bits.data_ptr(), Cn, T, x_tc.shape[1], int(adaptive), int(valuesize),
                                            x_tc.data_ptr(), err.data_ptr(), self._stream())
        self._check(ret, "dega_hip_decode_dev")
        return x_tc, err

    def encode_f32(self, v_tc, factor=100.0, adaptive=1, cap=None, valuesize=32):
        """float32 CUDA tensor [T, C] -> streams: Normalize fused into the encode kernel (one launch)."""
        import torch
        T, Cn = v_tc.shape
        assert v_tc.dtype == torch.float32 and v_tc.is_cuda and v_tc.is_contiguous()
        if cap is None:
            cap = worst_case_bytes(T) if valuesize <= 32 else library().dega_hip_worst_case_bytes64(T)
        out = torch.zeros((Cn, cap), dtype=torch.uint8, device=v_tc.device)
        bits = torch.zeros(Cn, dtype=torch.int64, device=v_tc.device)
        err = torch.zeros(Cn, dtype=torch.int32, device=v_tc.device)
        ret = library().dega_hip_encode_f32_dev(self._h, v_tc.data_ptr(), Cn, T, Cn, float(factor), int(adaptive), int(valuesize), out.data_ptr(), cap,
                                                bits.data_ptr(), err.data_ptr(), self._stream())
        self._check(ret, "dega_hip_encode_f32_dev")
        return out, bits, err

    def decode_f32(self, streams, bits, T, factor=100.0, adaptive=1, valuesize=32):
        import torch
        Cn, cap = streams.shape
        v = torch.zeros((T, Cn), dtype=torch.float32, device=streams.device)
        err = torch.zeros(Cn, dtype=torch.int32, device=streams.device)
        ret = library().dega_hip_decode_f32_dev(self._h, streams.data_ptr(), cap, bits.data_ptr(), Cn, T, Cn, float(factor), int(adaptive), int(valuesize),
                                                v.data_ptr(), None, err.data_ptr(), self._stream())
        self._check(ret, "dega_hip_decode_f32_dev")
        return v, err

    def normalize(self, v_tc, factor=100.0, valuesize=32):
        import torch
        T, Cn = v_tc.shape
        assert v_tc.dtype == torch.float32 and v_tc.is_cuda and v_tc.is_contiguous()
        x = torch.empty((T, Cn), dtype=torch.int32, device=v_tc.device)
        err = torch.zeros(Cn, dtype=torch.int32, device=v_tc.device)
        ret = library().dega_hip_normalize_dev(self._h, v_tc.data_ptr(), Cn, T, Cn, float(factor), int(valuesize), x.data_ptr(), err.data_ptr(), self._stream())
        self._check(ret, "dega_hip_normalize_dev")
        return x, err

    def denormalize(self, x_tc, factor=100.0, valuesize=32):
        import torch
        T, Cn = x_tc.shape
        v = torch.empty((T, Cn), dtype=torch.float32, device=x_tc.device)
        ret = library().dega_hip_denormalize_dev(self._h, x_tc.data_ptr(), Cn, T, Cn, float(factor), int(valuesize), v.data_ptr(), self._stream())
        self._check(ret, "dega_hip_denormalize_dev")
        return v

    def synth(self, C_, T, seed=1234, c0=0, S=50, device=None, out=None):
        import torch
        if out is None:
            out = torch.empty((T, C_), dtype=torch.int32, device=device or ("cuda:%d" % self.device))
        ret = library().dega_hip_synth_dev(self._h, out.data_ptr(), C_, T, out.shape[1], seed, c0, S, self._stream())
        self._check(ret, "dega_hip_synth_dev")
        return out

    def compact(self, streams, bits):
        """[C][cap] slabs -> (packed uint8 [total], offsets int64 [C+1])."""
        import torch
        Cn, cap = streams.shape
        offsets = torch.zeros(Cn + 1, dtype=torch.int64, device=streams.device)
        self._check(library().dega_hip_compact_offsets_dev(self._h, bits.data_ptr(), Cn, offsets.data_ptr(), self._stream()), "compact_offsets")
        total = int(offsets[-1].item())
        packed = torch.empty(max(total, 1), dtype=torch.uint8, device=streams.device)
        self._check(library().dega_hip_compact_gather_dev(self._h, streams.data_ptr(), cap, offsets.data_ptr(), Cn, packed.data_ptr(), self._stream()), "compact_gather")
        return packed[:total], offsets

    # ---- LZMH (BASELINE config 4) ---------------------------------------------------------------------------------
    def lzmh_encode(self, data, lens, cap=None, out=None, bits=None, err=None):
        """data: uint8 CUDA tensor [C, stride] (stride % 16 == 0), lens: int64 [C] bytes per channel.
        Returns (out uint8 [C, cap], bits int64 [C], err int32 [C])."""
        import torch
        Cn, stride = data.shape
        assert data.dtype == torch.uint8 and data.is_cuda and data.is_contiguous() and lens.dtype == torch.int64
        if cap is None:
            cap = lzmh_worst_case_bytes(stride)
        if out is None:
            out = torch.zeros((Cn, cap), dtype=torch.uint8, device=data.device)
        if bits is None:
            bits = torch.zeros(Cn, dtype=torch.int64, device=data.device)
        if err is None:
            err = torch.zeros(Cn, dtype=torch.int32, device=data.device)
        ret = library().dega_hip_lzmh_encode_dev(self._h, data.data_ptr(), stride, lens.data_ptr(), Cn, out.data_ptr(), cap,
                                                 bits.data_ptr(), err.data_ptr(), self._stream())
        self._check(ret, "dega_hip_lzmh_encode_dev")
        return out, bits, err

    def lzmh_decode(self, streams, bits, stride, out=None, lens=None, err=None):
        """streams: uint8 CUDA tensor [C, cap], bits int64 [C].  Returns (bytes uint8 [C, stride], lens int64 [C], err int32 [C])."""
        import torch
        Cn, cap = streams.shape
        assert streams.dtype == torch.uint8 and streams.is_cuda and streams.is_contiguous() and bits.dtype == torch.int64
        if out is None:
            out = torch.zeros((Cn, stride), dtype=torch.uint8, device=streams.device)
        if lens is None:
            lens = torch.zeros(Cn, dtype=torch.int64, device=streams.device)
        if err is None:
            err = torch.zeros(Cn, dtype=torch.int32, device=streams.device)
        ret = library().dega_hip_lzmh_decode_dev(self._h, streams.data_ptr(), cap, bits.data_ptr(), Cn, out.data_ptr(), stride,
                                                 lens.data_ptr(), err.data_ptr(), self._stream())
        self._check(ret, "dega_hip_lzmh_decode_dev")
        return out, lens, err

    def lzmh_render(self, x_tc, stride, out=None):
        """int32 channels [T, C] -> ASCII "%d.%02d\\n" lines per channel: (text uint8 [C, stride], lens int64 [C], err int32 [C])."""
        import torch
        T, Cn = x_tc.shape
        assert x_tc.dtype == torch.int32 and x_tc.is_cuda and x_tc.is_contiguous()
        if out is None:
            out = torch.zeros((Cn, stride), dtype=torch.uint8, device=x_tc.device)
        lens = torch.zeros(Cn, dtype=torch.int64, device=x_tc.device)
        err = torch.zeros(Cn, dtype=torch.int32, device=x_tc.device)
        ret = library().dega_hip_lzmh_render_dev(self._h, x_tc.data_ptr(), Cn, T, Cn, out.data_ptr(), stride, lens.data_ptr(),
                                                 err.data_ptr(), self._stream())
        self._check(ret, "dega_hip_lzmh_render_dev")
        return out, lens, err

    def lzmh_encode_host(self, strings, cap=None):
        """strings: list of bytes objects (one per channel).  Returns (out uint8 [C, cap], bits uint64 [C], err int32 [C])."""
        import numpy as np
        Cn = len(strings)
        stride = (max([len(s) for s in strings] + [1]) + 15) // 16 * 16
        data = np.zeros((Cn, stride), dtype=np.uint8)
        lens = np.zeros(Cn, dtype=np.uint64)
        for i, s in enumerate(strings):
            data[i, : len(s)] = np.frombuffer(s, dtype=np.uint8)
            lens[i] = len(s)
        if cap is None:
            cap = lzmh_worst_case_bytes(stride)
        out = np.zeros((Cn, cap), dtype=np.uint8)
        bits = np.zeros(Cn, dtype=np.uint64)
        err = np.zeros(Cn, dtype=np.int32)
        ret = library().dega_hip_lzmh_encode_host(self._h, data.ctypes.data, stride, lens.ctypes.data, Cn, out.ctypes.data, cap,
                                                  bits.ctypes.data, err.ctypes.data)
        self._check(ret, "dega_hip_lzmh_encode_host")
        return out, bits, err

    def lzmh_decode_host(self, streams, bits, stride):
        import numpy as np
        streams = np.ascontiguousarray(streams, dtype=np.uint8)
        bits = np.ascontiguousarray(bits, dtype=np.uint64)
        Cn, cap = streams.shape
        out = np.zeros((Cn, stride), dtype=np.uint8)
        lens = np.zeros(Cn, dtype=np.uint64)
        err = np.zeros(Cn, dtype=np.int32)
        ret = library().dega_hip_lzmh_decode_host(self._h, streams.ctypes.data, cap, bits.ctypes.data, Cn, out.ctypes.data, stride,
                                                  lens.ctypes.data, err.ctypes.data)
        self._check(ret, "dega_hip_lzmh_decode_host")
        return out, lens, err

    def profile(self, enable=True):
        library().dega_hip_profile(self._h, 1 if enable else 0)

    def profile_read(self, which=0, reset=True):
        avg = C.c_double(0.0)
        n = library().dega_hip_profile_read(self._h, which, C.byref(avg), 1 if reset else 0)
        return n, avg.value

    # ---- host-pointer API (numpy) ---------------------------------------------------------------------------------
    def encode_host(self, x_tc, adaptive=1, cap=None, valuesize=32):
        import numpy as np
        x_tc = np.ascontiguousarray(x_tc, dtype=np.int32)
        T, Cn = x_tc.shape
        if cap is None:
            cap = worst_case_bytes(T)
        out = np.zeros((Cn, cap), dtype=np.uint8)
        bits = np.zeros(Cn, dtype=np.uint64)
        err = np.zeros(Cn, dtype=np.int32)
        ret = library().dega_hip_encode_host(self._h, x_tc.ctypes.data, Cn, T, Cn, int(adaptive), int(valuesize), out.ctypes.data, cap, bits.ctypes.data, err.ctypes.data)
        self._check(ret, "dega_hip_encode_host")
        return out, bits, err

    def encode_packed_host(self, x_tc, adaptive=1, valuesize=32, packed_cap=None):
        """Streams returned packed: (packed uint8 [total], offsets uint64 [C+1], bits uint64 [C], err int32 [C])."""
        import numpy as np
        x_tc = np.ascontiguousarray(x_tc, dtype=np.int32)
        T, Cn = x_tc.shape
        if packed_cap is None:
            packed_cap = Cn * (T * 2 + 64)  # 2 bytes per sample: generous for meter data; the call says so if it is not
        offsets = np.zeros(Cn + 1, dtype=np.uint64)
        bits = np.zeros(Cn, dtype=np.uint64)
        err = np.zeros(Cn, dtype=np.int32)
        for _ in range(2):
            packed = np.empty(max(1, packed_cap), dtype=np.uint8)
            ret = library().dega_hip_encode_packed_host(self._h, x_tc.ctypes.data, Cn, T, Cn, int(adaptive), int(valuesize), packed.ctypes.data,
                                                        packed_cap, offsets.ctypes.data, bits.ctypes.data, err.ctypes.data)
            if ret != ERROR_MEMORY or int(offsets[Cn]) <= packed_cap:
                break
            packed_cap = int(offsets[Cn])
        self._check(ret, "dega_hip_encode_packed_host")
        return packed[: int(offsets[Cn])], offsets, bits, err

    def decode_packed_host(self, packed, offsets, bits, T, adaptive=1, valuesize=32, var=False):
        """The inverse of encode_packed_host.  Returns (x int32 [T, C], err) or, with var=True, (x, counts, err)."""
        import numpy as np
        packed = np.ascontiguousarray(packed, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        bits = np.ascontiguousarray(bits, dtype=np.uint64)
        Cn = bits.size
        x = np.zeros((T, Cn), dtype=np.int32)
        counts = np.zeros(Cn, dtype=np.uint64)
        err = np.zeros(Cn, dtype=np.int32)
        ret = library().dega_hip_decode_packed_host(self._h, packed.ctypes.data, offsets.ctypes.data, bits.ctypes.data, Cn, T, Cn, int(adaptive),
                                                    int(valuesize), x.ctypes.data, counts.ctypes.data if var else None, err.ctypes.data)
        self._check(ret, "dega_hip_decode_packed_host")
        return (x, counts, err) if var else (x, err)

    def decode_host(self, streams, bits, T, adaptive=1, valuesize=32):
        import numpy as np
        streams = np.ascontiguousarray(streams, dtype=np.uint8)
        bits = np.ascontiguousarray(bits, dtype=np.uint64)
        Cn, cap = streams.shape
        x = np.zeros((T, Cn), dtype=np.int32)
        err = np.zeros(Cn, dtype=np.int32)
        ret = library().dega_hip_decode_host(self._h, streams.ctypes.data, cap, bits.ctypes.data, Cn, T, Cn, int(adaptive), int(valuesize), x.ctypes.data, err.ctypes.data)
        self._check(ret, "dega_hip_decode_host")
        return x, err

    def decode_var_host(self, streams, bits, max_T, adaptive=1, valuesize=32):
        """Decode streams of unknown length: returns (x [max_T, C], counts uint64 [C], err)."""
        import numpy as np
        streams = np.ascontiguousarray(streams, dtype=np.uint8)
        bits = np.ascontiguousarray(bits, dtype=np.uint64)
        Cn, cap = streams.shape
        x = np.zeros((max_T, Cn), dtype=np.int32)
        counts = np.zeros(Cn, dtype=np.uint64)
        err = np.zeros(Cn, dtype=np.int32)
        ret = library().dega_hip_decode_var_host(self._h, streams.ctypes.data, cap, bits.ctypes.data, Cn, max_T, Cn, int(adaptive), int(valuesize),
                                                 x.ctypes.data, counts.ctypes.data, err.ctypes.data)
        self._check(ret, "dega_hip_decode_var_host")
        return x, counts, err

    def encode64_host(self, x_tc, valuesize, adaptive=1, cap=None):
        """valuesize 33..64: x_tc int64 [T, C] (the low valuesize bits count)."""
        import numpy as np
        x_tc = np.ascontiguousarray(x_tc, dtype=np.int64)
        T, Cn = x_tc.shape
        if cap is None:
            cap = library().dega_hip_worst_case_bytes64(T)
        out = np.zeros((Cn, cap), dtype=np.uint8)
        bits = np.zeros(Cn, dtype=np.uint64)
        err = np.zeros(Cn, dtype=np.int32)
        ret = library().dega_hip_encode64_host(self._h, x_tc.ctypes.data, Cn, T, Cn, int(adaptive), int(valuesize), out.ctypes.data, cap,
                                               bits.ctypes.data, err.ctypes.data)
        self._check(ret, "dega_hip_encode64_host")
        return out, bits, err

    def decode64_var_host(self, streams, bits, max_T, valuesize, adaptive=1):
        import numpy as np
        streams = np.ascontiguousarray(streams, dtype=np.uint8)
        bits = np.ascontiguousarray(bits, dtype=np.uint64)
        Cn, cap = streams.shape
        x = np.zeros((max_T, Cn), dtype=np.int64)
        counts = np.zeros(Cn, dtype=np.uint64)
        err = np.zeros(Cn, dtype=np.int32)
        ret = library().dega_hip_decode64_var_host(self._h, streams.ctypes.data, cap, bits.ctypes.data, Cn, max_T, Cn, int(adaptive), int(valuesize),
                                                   x.ctypes.data, counts.ctypes.data, err.ctypes.data)
        self._check(ret, "dega_hip_decode64_var_host")
        return x, counts, err

    def encode_f32_host(self, v_tc, factor=100.0, adaptive=1, cap=None, valuesize=32):
        import numpy as np
        v_tc = np.ascontiguousarray(v_tc, dtype=np.float32)
        T, Cn = v_tc.shape
        if cap is None:
            cap = worst_case_bytes(T) if valuesize <= 32 else library().dega_hip_worst_case_bytes64(T)
        out = np.zeros((Cn, cap), dtype=np.uint8)
        bits = np.zeros(Cn, dtype=np.uint64)
        err = np.zeros(Cn, dtype=np.int32)
        ret = library().dega_hip_encode_f32_host(self._h, v_tc.ctypes.data, Cn, T, Cn, float(factor), int(adaptive), int(valuesize), out.ctypes.data, cap, bits.ctypes.data, err.ctypes.data)
        self._check(ret, "dega_hip_encode_f32_host")
        return out, bits, err

    def decode_f32_host(self, streams, bits, T, factor=100.0, adaptive=1, valuesize=32):
        import numpy as np
        streams = np.ascontiguousarray(streams, dtype=np.uint8)
        bits = np.ascontiguousarray(bits, dtype=np.uint64)
        Cn, cap = streams.shape
        v = np.zeros((T, Cn), dtype=np.float32)
        err = np.zeros(Cn, dtype=np.int32)
        ret = library().dega_hip_decode_f32_host(self._h, streams.ctypes.data, cap, bits.ctypes.data, Cn, T, Cn, float(factor), int(adaptive), int(valuesize), v.ctypes.data, err.ctypes.data)
        self._check(ret, "dega_hip_decode_f32_host")
        return v, err


def synth_reference(C_, T, seed=1234, c0=0, S=50):
    """numpy restatement of dega_synth_kernel (csrc/dega_kernels.hpp) -- the workload definition of SURVEY.md 8d."""
    import numpy as np
    M = (1 << 64) - 1

    def mix(z):
        z = (z + 0x9E3779B97F4A7C15) & M
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M
        return z ^ (z >> 31)

    x = np.zeros((T, C_), dtype=np.int32)
    for c in range(C_):
        key = mix((seed ^ (((c0 + c) * 0xD1342543DE82EF95) & M)) & M)
        v = 10000 + mix(key) % 50000
        for t in range(T):
            if t > 0:
                v += mix((key + t) & M) % (2 * S + 1) - S
                v = min(max(v, 0), 2**31 - 1)
            x[t, c] = v
    return x
