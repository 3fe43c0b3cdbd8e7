"""LZMH kernel LOGIC check without a GPU: data-compressor_amd/csrc/lzmh_kernels.hpp compiled by g++ under the
thread-per-lane emulator (tests/sim/) against the fixtures made from the compiled reference and against the oracle.
A debugging aid for the build container; the parity tests proper are tests/test_gpu_lzmh.py (-m gpu)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle import orc

HERE = os.path.dirname(os.path.abspath(__file__))
SIM_DIR = os.path.join(HERE, "sim")
GOLDEN = os.path.join(HERE, "golden")


@pytest.fixture(scope="module")
def sim():
    subprocess.run(["make", "-s", "-C", SIM_DIR], check=True)
    S = C.CDLL(os.path.join(SIM_DIR, "libdega_sim.so"))
    sig = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    S.sim_lzmh_encode.argtypes = sig
    S.sim_lzmh_decode.argtypes = sig
    S.sim_lzmh_decode_half.argtypes = sig
    S.sim_lzmh_render.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    return S


def pack(strings, align):
    stride = (max([len(s) for s in strings] + [1]) + align - 1) // align * align
    data = np.zeros((len(strings), stride), dtype=np.uint8)
    lens = np.zeros(len(strings), dtype=np.uint64)
    for i, s in enumerate(strings):
        data[i, : len(s)] = np.frombuffer(s, dtype=np.uint8)
        lens[i] = len(s)
    return data, lens, stride


def sim_encode(S, strings, cap=None):
    data, lens, stride = pack(strings, 16)
    cap = cap or ((stride * 10 // 8 + 47) // 16 * 16)
    out = np.zeros((len(strings), cap), dtype=np.uint8)
    bits = np.zeros(len(strings), dtype=np.uint64)
    err = np.zeros(len(strings), dtype=np.int32)
    S.sim_lzmh_encode(data.ctypes.data, stride, lens.ctypes.data, len(strings), out.ctypes.data, cap, bits.ctypes.data, err.ctypes.data)
    return out, bits, err


def sim_decode(S, streams, stride, half=False):
    """half: the 32-channels-per-wave launch the library uses for batches of up to 64 Ki channels"""
    data, _, cap = pack([b for b, _ in streams], 4)
    bits = np.array([n for _, n in streams], dtype=np.uint64)
    out = np.zeros((len(streams), stride), dtype=np.uint8)
    lens = np.zeros(len(streams), dtype=np.uint64)
    err = np.zeros(len(streams), dtype=np.int32)
    fn = S.sim_lzmh_decode_half if half else S.sim_lzmh_decode
    fn(data.ctypes.data, cap, bits.ctypes.data, len(streams), out.ctypes.data, stride, lens.ctypes.data, err.ctypes.data)
    return out, lens, err


def test_lzmh_encode_kernel_logic_on_goldens(sim):
    z = np.load(os.path.join(GOLDEN, "lzmh.npz"))
    names = sorted(k[:-3] for k in z.files if k.endswith(".in") and z[k].size <= 2000)  # the emulator is slow; all of them run on the GPU
    strings = [z[n + ".in"].tobytes() for n in names]
    out, bits, err = sim_encode(sim, strings)
    assert (err == 0).all()
    for i, n in enumerate(names):
        nb = int(z[n + ".bits"][0])
        assert int(bits[i]) == nb, n
        assert out[i, : (nb + 7) // 8].tobytes() == z[n + ".stream"].tobytes(), n


def test_lzmh_decode_kernel_logic_on_goldens(sim):
    z = np.load(os.path.join(GOLDEN, "lzmh.npz"))
    names = sorted(k[:-3] for k in z.files if k.endswith(".in") and z[k].size <= 2000)
    streams = [(z[n + ".stream"].tobytes(), int(z[n + ".bits"][0])) for n in names]
    out, lens, err = sim_decode(sim, streams, 2008, half=True)
    assert (err == 0).all()
    for i, n in enumerate(names):
        want = z[n + ".dec"].tobytes()
        assert int(lens[i]) == len(want) and out[i, : len(want)].tobytes() == want, n


def test_lzmh_kernels_ragged_wave_vs_oracle(sim):
    rng = np.random.default_rng(21)
    strings = []
    for it in range(70):  # two waves, the second ragged; lengths around the ring size and the window reload points
        kind = it % 6
        n = int(rng.integers(0, 800)) if it % 7 else [0, 1, 2, 3, 402, 403, 404, 405, 806][it % 9]
        if kind == 0:
            s = bytes(rng.integers(0, 256, n, dtype=np.uint8))
        elif kind == 1:
            s = "".join("%.2f\n" % v for v in 230 + np.cumsum(rng.normal(0, 0.3, n // 6 + 1))).encode()[:n]
        elif kind == 2:
            s = bytes(rng.integers(48, 58, n, dtype=np.uint8))
        elif kind == 3:
            s = bytes(rng.integers(0, 3, n, dtype=np.uint8))
        elif kind == 4:
            s = (b"abcabcabd" * (n // 9 + 1))[:n]
        else:
            s = bytes(n)  # 274-byte matches: the window-reload-and-retry path
        strings.append(s)
    out, bits, err = sim_encode(sim, strings)
    assert (err == 0).all()
    streams = []
    for i, s in enumerate(strings):
        r, b, n = orc.stage("lzmh", True, s, 8 * len(s))
        assert r == 0 and int(bits[i]) == n and out[i, : (n + 7) // 8].tobytes() == b[: (n + 7) // 8], (i, len(s))
        streams.append((b[: (n + 7) // 8], n if i % 3 else 8 * ((n + 7) // 8)))  # every third one in its zero-padded file form
    for k in range(6):  # damaged streams: same status and length as the oracle (the reference itself is undefined there)
        g = bytes(rng.integers(0, 256, 40, dtype=np.uint8))
        streams.append((g, 8 * len(g) - k))
        streams[-1] = (g[:-1] + bytes([g[-1] & (0xFF00 >> ((8 - k) % 8 or 8)) & 0xFF]), 8 * len(g) - k)
    dout, lens, derr = sim_decode(sim, streams, 16384)
    assert (derr == 0).all()
    for i, (b, n) in enumerate(streams):
        r, d, dn = orc.stage("lzmh", False, b, n)
        assert r == 0 and int(lens[i]) == dn // 8 and dout[i, : dn // 8].tobytes() == d[: dn // 8], i


def test_lzmh_slab_overflow_is_reported(sim):
    rng = np.random.default_rng(2)
    s = bytes(rng.integers(0, 256, 900, dtype=np.uint8))  # incompressible: ~10 bits per byte
    out, bits, err = sim_encode(sim, [s, b"1.00\n" * 100], cap=256)
    assert err[0] == orc.ERROR_MEMORY and bits[0] == 0 and err[1] == 0
    r, b, n = orc.stage("lzmh", True, s, 8 * len(s))
    dout, lens, derr = sim_decode(sim, [(b[: (n + 7) // 8], n)], 512)
    assert derr[0] == orc.ERROR_MEMORY and lens[0] == 0


def test_lzmh_render_kernel(sim):
    rng = np.random.default_rng(4)
    x = np.abs(np.cumsum(rng.integers(-300, 301, (50, 5)), axis=0) + np.array([0, 99, 100, 23045, 2000000000])).astype(np.int32)
    out = np.zeros((5, 1024), dtype=np.uint8)
    lens = np.zeros(5, dtype=np.uint64)
    err = np.zeros(5, dtype=np.int32)
    sim.sim_lzmh_render(x.ctypes.data, 5, 50, 5, out.ctypes.data, 1024, lens.ctypes.data, err.ctypes.data)
    for c in range(5):
        want = "".join("%d.%02d\n" % (v // 100, v % 100) for v in x[:, c].tolist()).encode()
        assert err[c] == 0 and out[c, : int(lens[c])].tobytes() == want


@pytest.mark.timeout(600)
def test_lzmh_longest_match_at_every_window_alignment(sim):
    """A match of the maximum length (274) that begins at every position modulo 16: the window a lane reloads starts at
    (P - 128) rounded down to 16, and whatever the alignment it has to hold the whole match, or the step is retried for ever."""
    rng = np.random.default_rng(3)
    strings = []
    for k in range(32):
        head = bytes(rng.integers(0, 256, 130 + k, dtype=np.uint8))
        strings.append(head + bytes([65 + k]) * (700 + 3 * k) + head[:40])
    out, bits, err = sim_encode(sim, strings)
    assert (err == 0).all()
    for i, s in enumerate(strings):
        r, b, n = orc.stage("lzmh", True, s, 8 * len(s))
        assert r == 0 and int(bits[i]) == n and out[i, : (n + 7) // 8].tobytes() == b[: (n + 7) // 8], (i, len(s))
