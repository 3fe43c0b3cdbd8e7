"""Kernel LOGIC check without a GPU: the real kernel source (data-compressor_amd/csrc/dega_kernels.hpp) compiled by g++
under the thread-per-lane emulator of tests/sim/ and compared with the golden vectors.  This is a debugging aid for
the build container (ring indexing, phase control, termination); the parity tests proper are the -m gpu tests."""
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
    S.sim_encode.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    S.sim_synth.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_uint64, C.c_uint64, C.c_uint32]
    S.sim_normalize.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_float, C.c_void_p, C.c_void_p]
    S.sim_denormalize.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_float, C.c_void_p]
    return S


def sim_encode(S, x, ad, cap=None):
    x = np.ascontiguousarray(x, dtype=np.int32)
    T, Cn = x.shape
    if cap is None:
        cap = (orc.lib().orc_dega_worst_case_bytes(T) + 3) & ~3
    out = np.zeros((Cn, cap), dtype=np.uint8)
    bits = np.zeros(Cn, dtype=np.uint64)
    err = np.zeros(Cn, dtype=np.int32)
    S.sim_encode(x.ctypes.data, Cn, T, Cn, ad, out.ctypes.data, cap, bits.ctypes.data, err.ctypes.data)
    return out, bits, err


def test_encode_kernel_logic_on_golden_sets(sim):
    z = np.load(os.path.join(GOLDEN, "channels.npz"))
    for name in ("walk50", "walk300_T96", "ragged_small", "wild", "with_errors", "zeros"):
        x = z[name + ".x"]
        for ad, tag in ((1, "ad"), (0, "st")):
            out, bits, err = sim_encode(sim, x, ad)
            gs, gb, ge = z["%s.%s.stream" % (name, tag)], z["%s.%s.bits" % (name, tag)], z["%s.%s.err" % (name, tag)]
            assert (err == ge).all(), name
            ok = ge == 0
            assert (bits[ok] == gb[ok]).all(), name
            for c in np.nonzero(ok)[0]:
                nb = (int(gb[c]) + 7) // 8
                assert out[c, :nb].tobytes() == gs[c, :nb].tobytes(), (name, tag, c)


def test_encode_kernel_logic_multiwave_vs_oracle(sim):
    rng = np.random.default_rng(11)
    T, Cn = 120, 300  # two workgroups, a ragged last wave
    x = np.cumsum(rng.integers(-80, 81, (T, Cn)) * rng.integers(0, 3, Cn)[None, :], axis=0) + 40000
    x = x.astype(np.int32)
    out, bits, err = sim_encode(sim, x, 1)
    o2, b2, e2 = orc.encode_batch_tc(x, 1, cap=out.shape[1])
    assert (err == e2).all() and (bits == b2).all() and (out == o2).all()


def test_synth_and_normalize_kernels(sim):
    from __graft_entry__ import load_package
    dca = load_package()
    x = np.zeros((40, 70), dtype=np.int32)
    sim.sim_synth(x.ctypes.data, 70, 40, 70, 1234, 3, 50)
    assert (x == dca.synth_reference(70, 40, seed=1234, c0=3, S=50)).all()
    z = np.load(os.path.join(GOLDEN, "floats.npz"))
    v = z["v_100"]
    vt = np.ascontiguousarray(np.stack([v, v], axis=1))
    xi = np.zeros(vt.shape, dtype=np.int32)
    err = np.zeros(2, dtype=np.int32)
    sim.sim_normalize(vt.ctypes.data, 2, v.size, 2, 100.0, xi.ctypes.data, err.ctypes.data)
    assert (err == 0).all() and (xi[:, 0] == z["norm_100"]).all()
    back = np.zeros(vt.shape, dtype=np.float32)
    sim.sim_denormalize(xi.ctypes.data, 2, v.size, 2, 100.0, back.ctypes.data)
    assert back[:, 1].tobytes() == z["denorm_100"].tobytes()


def test_fast_word_path_equals_bit_path_on_random_and_nasty_states(sim):
    """BacEncoder::encode_word (classes FAST8, FAST4, GENERAL -- also run on lanes of a cheaper class, as happens when another
    lane of the wave needs it) against encode_bit on random encoder states: every model state (near a halving, near an
    MPS/LPS swap, skewed counts), all-ones accumulators and held-back words that overflow when a carry arrives (the
    33+-pending-bits case, settled after the word from the record the word path returns), and words made of the rare
    symbol only (the most finished bits a group of symbols can make: the sentinel of the 64-bit register must not be
    shifted out -- BacEncoder::classify's capacity precondition)."""
    sim.sim_fast_vs_slow.argtypes = [C.c_uint64, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    for ad in (1, 0):
        for seed in (2024, 5):
            words, ripples = C.c_int(), C.c_int()
            bad = sim.sim_fast_vs_slow(seed, 60000, ad, C.byref(words), C.byref(ripples))
            assert bad == 0
            assert words.value > 30000 and ripples.value > 500  # word paths and the deferred carry were both exercised


def test_decode_kernel_logic_on_golden_sets(sim):
    sim.sim_decode.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]
    z = np.load(os.path.join(GOLDEN, "channels.npz"))
    for name in ("walk300_T96", "ragged_small", "wild", "zeros"):
        x = z[name + ".x"]
        T = x.shape[0]
        for ad, tag in ((1, "ad"), (0, "st")):
            st, gb, ge = z["%s.%s.stream" % (name, tag)], z["%s.%s.bits" % (name, tag)], z["%s.%s.err" % (name, tag)]
            ok = ge == 0
            s2 = np.ascontiguousarray(st[ok])
            if s2.shape[1] % 4:
                s3 = np.zeros((s2.shape[0], (s2.shape[1] + 3) & ~3), dtype=np.uint8)
                s3[:, : s2.shape[1]] = s2
                s2 = s3
            bits = np.ascontiguousarray(gb[ok])
            y = np.zeros((T, s2.shape[0]), dtype=np.int32)
            err = np.zeros(s2.shape[0], dtype=np.int32)
            sim.sim_decode(s2.ctypes.data, s2.shape[1], bits.ctypes.data, s2.shape[0], T, s2.shape[0], ad, y.ctypes.data, err.ctypes.data)
            assert (err == 0).all() and (y == x[:, ok]).all(), (name, tag)


def test_wide_workgroup_shape_on_golden_sets(sim):
    """The wide workgroups (8 pairs of waves, 16-sample ring) the library uses for decoding batches of more than 64 Ki channels,
    forced on the small golden batches: the same samples back.  (The encoder has one workgroup shape.)"""
    sig_d = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]
    sim.sim_decode_wide.argtypes = sig_d
    z = np.load(os.path.join(GOLDEN, "channels.npz"))
    for name in ("walk300_T96", "wild", "with_errors", "ragged_small"):
        x = np.ascontiguousarray(z[name + ".x"], dtype=np.int32)
        T, Cn = x.shape
        for ad, tag in ((1, "ad"), (0, "st")):
            cap = (orc.lib().orc_dega_worst_case_bytes(T) + 3) & ~3
            out = np.zeros((Cn, cap), dtype=np.uint8)
            bits = np.zeros(Cn, dtype=np.uint64)
            err = np.zeros(Cn, dtype=np.int32)
            sim.sim_encode(x.ctypes.data, Cn, T, Cn, ad, out.ctypes.data, cap, bits.ctypes.data, err.ctypes.data)
            gs, gb, ge = z["%s.%s.stream" % (name, tag)], z["%s.%s.bits" % (name, tag)], z["%s.%s.err" % (name, tag)]
            assert (err == ge).all(), name
            ok = ge == 0
            assert (bits[ok] == gb[ok]).all(), name
            for c in np.nonzero(ok)[0]:
                nb = (int(gb[c]) + 7) // 8
                assert out[c, :nb].tobytes() == gs[c, :nb].tobytes(), (name, tag, c)
            y = np.zeros((T, Cn), dtype=np.int32)
            derr = np.zeros(Cn, dtype=np.int32)
            b2 = np.where(ok, bits, 0).astype(np.uint64)
            sim.sim_decode_wide(out.ctypes.data, cap, b2.ctypes.data, Cn, T, Cn, ad, y.ctypes.data, derr.ctypes.data)
            assert (derr[ok] == 0).all() and (y[:, ok] == x[:, ok]).all(), (name, tag)


def test_long_channels_halvings_and_rare_symbol_runs(sim):
    """Channels long enough for the counts to halve (bac.c:57) several times, at a different step in every lane -- the
    words around a halving are coded in two masked parts -- plus channels that sit still for thousands of samples and then
    jump: 32 rare symbols in a row take some 300 stream bits, more than a word path looks ahead."""
    sim.sim_decode.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(5)
    T, Cn = 4200, 70
    step = rng.integers(1, 400, Cn)
    x = np.cumsum(rng.integers(-1, 2, (T, Cn)) * rng.integers(0, 2, (T, Cn)) * step[None, :], axis=0) + 1000000
    for c in range(0, Cn, 9):  # still, then wild
        x[:, c] = 777
        x[3000:3040, c] = rng.integers(0, 2 ** 31 - 1, 40)
        x[3500:, c] = np.cumsum(rng.integers(-3, 4, T - 3500)) + 5000
    x = x.astype(np.int32)
    out, bits, err = sim_encode(sim, x, 1)
    o2, b2, e2 = orc.encode_batch_tc(x, 1, cap=out.shape[1])
    assert (err == e2).all() and (bits == b2).all() and (out == o2).all()
    y = np.zeros((T, Cn), dtype=np.int32)
    derr = np.zeros(Cn, dtype=np.int32)
    sim.sim_decode(out.ctypes.data, out.shape[1], bits.ctypes.data, Cn, T, Cn, 1, y.ctypes.data, derr.ctypes.data)
    assert (derr == 0).all() and (y == x).all()


def test_channels_coded_over_several_launches_with_saved_state(sim):
    """The rows of a batch coded in ranges, one launch per range, every lane's state (bit queue, interval, model, finished
    bits, held-back word) saved in between (EncodeArgs::seg_state): the streams must be those of one launch -- cuts inside a
    seg-bit word, right after the first row, ranges of a single row, an empty last range, channels in error."""
    sim.sim_encode_segments.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(5)
    T, Cn = 700, 130  # a ragged last wave
    x = (np.cumsum(rng.integers(-300, 301, (T, Cn)), axis=0) + 70000).astype(np.int32)
    x[300:, 7] = -5  # a channel that goes out of range in the middle (diff.c:17-18)
    x[0, 9] = -1     # ... and one that starts out of range
    cap = (orc.lib().orc_dega_worst_case_bytes(T) + 3) & ~3
    for ad in (1, 0):
        want_out, want_bits, want_err = orc.encode_batch_tc(x, ad, cap=cap)
        for cuts in ([0, 1, 2, 350, 351, 699, 700], [0, 255, 700, 700], [0, 8, 16, 400, 700]):
            out = np.zeros((Cn, cap), dtype=np.uint8)
            bits = np.zeros(Cn, dtype=np.uint64)
            err = np.zeros(Cn, dtype=np.int32)
            cu = np.array(cuts, dtype=np.uint64)
            sim.sim_encode_segments(x.ctypes.data, Cn, T, Cn, ad, cu.ctypes.data, len(cuts), out.ctypes.data, cap, bits.ctypes.data, err.ctypes.data)
            assert (err == want_err).all(), (ad, cuts)
            ok = want_err == 0
            assert (bits[ok] == want_bits[ok]).all(), (ad, cuts)
            for c in np.nonzero(ok)[0]:
                nb = (int(want_bits[c]) + 7) // 8
                assert out[c, :nb].tobytes() == want_out[c, :nb].tobytes(), (ad, cuts, c)
