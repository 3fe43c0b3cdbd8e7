"""valuesize 1..31: the option every DEGA stage takes (DCLib/src/enc_dec.c:72).  Samples are unsigned
valuesize-bit fields (diff.c:15 does not sign extend), held in int32 containers on the GPU side.
  * oracle stages against tests/golden/valuesizes.npz (made from the compiled reference, error channels included),
  * the kernel source under the emulator (a subset; CPU),
  * the HIP kernels through the C ABI (-m gpu), plus normalize/denormalize against the oracle's stage."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle import orc

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")
SIZES = (1, 2, 7, 8, 12, 15, 16, 17, 24, 31)


def pack_be(vals, vs):
    v = np.asarray(vals, dtype=np.uint64)
    bits = np.zeros(len(v) * vs, dtype=np.uint8)
    for k in range(vs):
        bits[k::vs] = (v >> np.uint64(vs - 1 - k)) & np.uint64(1)
    return np.packbits(bits).tobytes(), len(v) * vs


def unpack_be(data, nbits, vs):
    bits = np.unpackbits(np.frombuffer(data, dtype=np.uint8))[:nbits].reshape(-1, vs).astype(np.uint64)
    return (bits << np.arange(vs - 1, -1, -1, dtype=np.uint64)).sum(axis=1).astype(np.uint32)


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GOLDEN, "valuesizes.npz"))


def check_streams(gold, vs, tag, out, bits, err):
    ge, gb, gs = gold["vs%d.%s.err" % (vs, tag)], gold["vs%d.%s.bits" % (vs, tag)], gold["vs%d.%s.stream" % (vs, tag)]
    assert (np.asarray(err) == ge).all(), (vs, tag)
    for c in np.nonzero(ge == 0)[0]:
        nb = int(gb[c])
        assert int(bits[c]) == nb and bytes(out[c][: (nb + 7) // 8]) == gs[c, : (nb + 7) // 8].tobytes(), (vs, tag, int(c))


def test_oracle_stages_with_valuesize(gold):
    for vs in SIZES:
        x = gold["vs%d.x" % vs]
        for ad, tag in ((1, "ad"), (0, "st")):
            outs, bits, errs = [], [], []
            for c in range(x.shape[1]):
                data, n = pack_be(x[:, c], vs)
                r = 0
                for name in ("diff", "seg", "bac"):
                    r, data, n = orc.stage(name, True, data, n, valuesize=vs, adaptive=ad)
                    if r != 0:
                        break
                outs.append(data if r == 0 else b"")
                bits.append(n if r == 0 else 0)
                errs.append(r)
                if r == 0:
                    d, dn = data, n
                    for name in ("bac", "seg", "diff"):
                        r2, d, dn = orc.stage(name, False, d, dn, valuesize=vs, adaptive=ad)
                        assert r2 == 0
                    assert (unpack_be(d, dn, vs) == x[:, c]).all()
            check_streams(gold, vs, tag, outs, bits, errs)
    assert sum(int((gold["vs%d.ad.err" % vs] == orc.ERROR_INVALID_VALUE).sum()) for vs in SIZES) > 30  # the range check is exercised


@pytest.fixture(scope="module")
def sim():
    sim_dir = os.path.join(HERE, "sim")
    subprocess.run(["make", "-s", "-C", sim_dir], check=True)
    S = C.CDLL(os.path.join(sim_dir, "libdega_sim.so"))
    S.sim_encode_vs.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    S.sim_decode_vs.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    return S


def dirty(x, vs, rng):
    """The same samples with random bits above the value size: the kernels must ignore them."""
    junk = rng.integers(0, 1 << (32 - vs), x.shape, dtype=np.uint64) << np.uint64(vs)
    return np.ascontiguousarray((x.astype(np.uint64) | junk).astype(np.uint32).view(np.int32))


def test_kernel_logic_with_valuesize(sim, gold):
    rng = np.random.default_rng(3)
    for vs in (2, 8, 15, 17, 31):
        x = gold["vs%d.x" % vs]
        T, Cn = x.shape
        xin = dirty(x, vs, rng)
        for ad, tag in ((1, "ad"), (0, "st")):
            cap = 4 * ((T * (2 * vs + 3) // 4 + 64) // 4 + 4)
            out = np.zeros((Cn, cap), dtype=np.uint8)
            bits = np.zeros(Cn, dtype=np.uint64)
            err = np.zeros(Cn, dtype=np.int32)
            sim.sim_encode_vs(xin.ctypes.data, Cn, T, Cn, ad, vs, out.ctypes.data, cap, bits.ctypes.data, err.ctypes.data)
            check_streams(gold, vs, tag, out, bits, err)
            ok = err == 0
            y = np.zeros((T, Cn), dtype=np.int32)
            derr = np.zeros(Cn, dtype=np.int32)
            b2 = np.where(ok, bits, 0).astype(np.uint64)
            sim.sim_decode_vs(out.ctypes.data, cap, b2.ctypes.data, Cn, T, Cn, ad, vs, y.ctypes.data, derr.ctypes.data)
            assert (derr[ok] == 0).all() and (y.view(np.uint32)[:, ok] == x[:, ok]).all(), (vs, tag)


@pytest.mark.gpu
def test_gpu_valuesize_goldens(gold):
    from __graft_entry__ import load_package
    dca = load_package()
    ctx = dca.Context(0)
    rng = np.random.default_rng(4)
    for vs in SIZES:
        x = gold["vs%d.x" % vs]
        T, Cn = x.shape
        xin = dirty(x, vs, rng)
        for ad, tag in ((1, "ad"), (0, "st")):
            out, bits, err = ctx.encode_host(xin, adaptive=ad, valuesize=vs)
            check_streams(gold, vs, tag, out, bits, err)
            ok = err == 0
            y, derr = ctx.decode_host(out, np.where(ok, bits, 0).astype(np.uint64), T, adaptive=ad, valuesize=vs)
            assert (derr[ok] == 0).all() and (y.view(np.uint32)[:, ok] == x[:, ok]).all(), (vs, tag)
    with pytest.raises(dca.DegaError):
        ctx.encode_host(np.zeros((4, 4), dtype=np.int32), valuesize=33)  # int32 containers end at 32
    ctx.close()


@pytest.mark.gpu
def test_gpu_valuesize_damaged_streams_and_prefix_cap():
    """Random bits into the decoder: the status per channel and, where it is 0, the samples equal the oracle's chain
    (the zero-prefix cap valuesize + 1 of seg.c:55-56,74 is what most of these trip over)."""
    from __graft_entry__ import load_package
    dca = load_package()
    ctx = dca.Context(0)
    rng = np.random.default_rng(8)
    for vs in (3, 8, 16, 24):
        Cn, T = 128, 40
        # start from valid adaptive streams of small walks, then flip a few bits
        # the first sample is differenced against 0 (diff.c:11), so it has to stay below 2^(valuesize-1) itself
        x = (np.cumsum(rng.integers(-2, 3, (T, Cn)), axis=0) + (1 << (vs - 2))).clip(0, (1 << (vs - 1)) - 1).astype(np.int32)
        out, bits, err = ctx.encode_host(x, adaptive=1, valuesize=vs)
        assert (err == 0).all()
        for c in range(Cn):
            nb = int(bits[c])
            for _ in range(c % 4):
                k = int(rng.integers(0, nb))
                out[c, k // 8] ^= 0x80 >> (k % 8)
        room = 4096  # a damaged stream may decode to many more samples than went in, one per coded bit at most
        y, counts, derr = ctx.decode_var_host(out, bits, room, adaptive=1, valuesize=vs)
        for c in range(Cn):
            nb = int(bits[c])
            d, dn, r = out[c, : (nb + 7) // 8].tobytes(), nb, 0
            for name in ("bac", "seg", "diff"):
                r, d, dn = orc.stage(name, False, d, dn, valuesize=vs, adaptive=1)
                if r != 0:
                    break
            assert r != 0 or dn // vs <= room
            if r != 0 and derr[c] == orc.ERROR_MEMORY:
                continue  # the fused decoder ran out of room before it reached the place where the stage-wise chain fails
            assert derr[c] == r, (vs, c, derr[c], r)
            if r == 0:
                want = unpack_be(d, dn, vs)
                assert int(counts[c]) == len(want) and (y[: len(want), c].view(np.uint32) == want).all(), (vs, c)
    ctx.close()


@pytest.mark.gpu
def test_gpu_normalize_with_valuesize():
    """normalize writes the low valuesize bits after a range check against +-2^(valuesize-1) (normalize.c:21-24);
    denormalize sign extends them (normalize.c:36-38)."""
    import torch
    from __graft_entry__ import load_package
    dca = load_package()
    ctx = dca.Context(0)
    rng = np.random.default_rng(6)
    for vs in (8, 16, 26, 31):
        lim = float(1 << (vs - 1))
        v = np.concatenate([rng.uniform(-lim / 100.0, lim / 100.0, 500), [lim / 100.0 - 0.01, -lim / 100.0, 0.0, 0.004, -0.005]]).astype(np.float32)
        r, b, n = orc.stage("normalize", True, v.tobytes(), 32 * v.size, valuesize=vs, factor=100.0)
        vt = torch.from_numpy(v.reshape(-1, 1).copy()).cuda()
        x, err = ctx.normalize(vt, 100.0, valuesize=vs)
        if r == 0:
            assert int(err.item()) == 0 and (x.cpu().numpy().ravel().view(np.uint32) == unpack_be(b, n, vs)).all(), vs
            r2, fb, fn = orc.stage("normalize", False, b, n, valuesize=vs, factor=100.0)
            back = ctx.denormalize(x, 100.0, valuesize=vs).cpu().numpy().ravel()
            assert r2 == 0 and back.tobytes() == fb[: fn // 8], vs
        else:
            assert int(err.item()) == r
        big = np.array([[lim * 1.01 / 100.0 * 1.5]], dtype=np.float32)
        rb, _, _ = orc.stage("normalize", True, big.tobytes(), 32, valuesize=vs, factor=100.0)
        _, e2 = ctx.normalize(torch.from_numpy(big).cuda(), 100.0, valuesize=vs)
        assert rb == orc.ERROR_INVALID_VALUE == int(e2.item())
    ctx.close()


def _delimiter_cases():
    """seg bit strings that end inside a codeword, coded with the (pinned) oracle's bac: right after the delimiting one the
    reference takes the stump for padding (seg.c:58-62), with part of the residual present its read comes up short."""
    cases = ["01", "001", "0001", "101", "1001", "11101", "0011", "00101", "1" * 40 + "01", "010" + "1" * 31 + "0001"]
    streams = []
    for bits in cases:
        data = np.packbits(np.array([int(b) for b in bits], dtype=np.uint8)).tobytes()
        r, b, n = orc.stage("bac", True, data, len(bits), adaptive=1)
        assert r == 0
        d, dn, rr = b, n, 0
        for name in ("bac", "seg", "diff"):
            rr, d, dn = orc.stage(name, False, d, dn, adaptive=1)
            if rr != 0:
                break
        streams.append((b[: (n + 7) // 8], n, rr, np.frombuffer(d[: dn // 8], dtype=">u4").astype(np.uint32) if rr == 0 else None))
    return cases, streams


def _check_delimiter_cases(decode_var):
    cases, streams = _delimiter_cases()
    cap = 4 * ((max(len(s[0]) for s in streams) + 3) // 4 + 1)
    data = np.zeros((len(streams), cap), dtype=np.uint8)
    bits = np.zeros(len(streams), dtype=np.uint64)
    for i, (b, n, _, _) in enumerate(streams):
        data[i, : len(b)] = np.frombuffer(b, dtype=np.uint8)
        bits[i] = n
    y, counts, err = decode_var(data, bits, 64)
    for i, (_, _, r, want) in enumerate(streams):
        assert err[i] == r, (cases[i], err[i], r)
        if r == 0:
            assert int(counts[i]) == len(want) and (y[: len(want), i].view(np.uint32) == want).all(), cases[i]
    assert [s[2] for s in streams].count(orc.ERROR_LIBRARY_CALL) >= 1 and [s[2] for s in streams].count(0) >= 6  # both rules are exercised


def test_kernel_logic_stream_ending_after_a_delimiter(sim):
    sim.sim_decode_var_vs.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int, C.c_int, C.c_void_p,
                                      C.c_void_p, C.c_void_p]

    def decode_var(data, bits, room):
        Cn, cap = data.shape
        y = np.zeros((room, Cn), dtype=np.int32)
        counts = np.zeros(Cn, dtype=np.uint64)
        err = np.zeros(Cn, dtype=np.int32)
        sim.sim_decode_var_vs(data.ctypes.data, cap, bits.ctypes.data, Cn, room, Cn, 1, 32, y.ctypes.data, counts.ctypes.data, err.ctypes.data)
        return y, counts, err
    _check_delimiter_cases(decode_var)


@pytest.mark.gpu
def test_gpu_stream_ending_after_a_delimiter():
    from __graft_entry__ import load_package
    dca = load_package()
    ctx = dca.Context(0)
    _check_delimiter_cases(lambda data, bits, room: ctx.decode_var_host(data, bits, room, adaptive=1))
    ctx.close()


@pytest.mark.gpu
def test_gpu_valuesize_large_batch_wide_workgroups():
    """More than 64 Ki channels take the wide workgroup shape (8 pairs of waves); with a narrow value size that is its own instantiation
    of both kernels.  A sample of channels against the oracle's chain, all of them through the round trip."""
    import torch
    from __graft_entry__ import load_package
    dca = load_package()
    ctx = dca.Context(0)
    rng = np.random.default_rng(12)
    for vs, ad in ((16, 1), (9, 0)):
        Cn, T = 70000, 48
        half = 1 << (vs - 1)
        x = (np.cumsum(rng.integers(-(half // 16 + 1), half // 16 + 2, (T, Cn)), axis=0) + half // 2).clip(0, (1 << vs) - 1).astype(np.int32)
        xt = torch.from_numpy(x).cuda()
        out, bits, err = ctx.encode(xt, adaptive=ad, valuesize=vs)
        y, derr = ctx.decode(out, torch.where(err == 0, bits, torch.zeros_like(bits)), T, adaptive=ad, valuesize=vs)
        torch.cuda.synchronize()
        e = err.cpu().numpy()
        ok = e == 0
        assert ok.sum() > Cn // 2  # most walks stay inside +-2^(valuesize-1) per step; the clipped ones may not
        assert (derr.cpu().numpy()[ok] == 0).all() and (y.cpu().numpy()[:, ok] == x[:, ok]).all()
        ob, bb = out.cpu().numpy(), bits.cpu().numpy()
        for c in list(range(0, Cn, 997)) + [Cn - 1]:
            d, n = pack_be(x[:, c], vs)
            r = 0
            for name in ("diff", "seg", "bac"):
                r, d, n = orc.stage(name, True, d, n, valuesize=vs, adaptive=ad)
                if r != 0:
                    break
            assert e[c] == r, (vs, c)
            if r == 0:
                assert int(bb[c]) == n and ob[c, : (n + 7) // 8].tobytes() == d[: (n + 7) // 8], (vs, c)
    ctx.close()


# ---- valuesize 33..64: 64-bit containers -----------------------------------------------------------------------------------
SIZES64 = (33, 40, 48, 63, 64)


@pytest.fixture(scope="module")
def gold64():
    return np.load(os.path.join(GOLDEN, "valuesizes64.npz"))


def unpack_be64(data, nbits, vs):
    bits = np.unpackbits(np.frombuffer(data, dtype=np.uint8))[:nbits].reshape(-1, vs).astype(np.uint64)
    out = np.zeros(bits.shape[0], dtype=np.uint64)
    for k in range(vs):
        out |= bits[:, k] << np.uint64(vs - 1 - k)
    return out


def check_streams64(gold64, vs, tag, out, bits, err):
    ge, gb, gs = gold64["vs%d.%s.err" % (vs, tag)], gold64["vs%d.%s.bits" % (vs, tag)], gold64["vs%d.%s.stream" % (vs, tag)]
    assert (np.asarray(err) == ge).all(), (vs, tag)
    for c in np.nonzero(ge == 0)[0]:
        nb = int(gb[c])
        assert int(bits[c]) == nb and bytes(out[c][: (nb + 7) // 8]) == gs[c, : (nb + 7) // 8].tobytes(), (vs, tag, int(c))


def test_oracle_stages_with_valuesize_above_32(gold64):
    lossy = 0
    for vs in SIZES64:
        x = gold64["vs%d.x" % vs]
        for ad, tag in ((1, "ad"), (0, "st")):
            outs, bits, errs = [], [], []
            for c in range(x.shape[1]):
                data, n = pack_be(x[:, c], vs)
                r = 0
                for name in ("diff", "seg", "bac"):
                    r, data, n = orc.stage(name, True, data, n, valuesize=vs, adaptive=ad)
                    if r != 0:
                        break
                outs.append(data if r == 0 else b"")
                bits.append(n if r == 0 else 0)
                errs.append(r)
                if r == 0:
                    d, dn, r2 = data, n, 0
                    for name in ("bac", "seg", "diff"):
                        r2, d, dn = orc.stage(name, False, d, dn, valuesize=vs, adaptive=ad)
                        assert r2 == 0
                    back = unpack_be64(d, dn, vs)
                    assert (back == gold64["vs%d.%s.dec" % (vs, tag)][:, c]).all()
                    lossy += int((back != x[:, c]).any())
            check_streams64(gold64, vs, tag, outs, bits, errs)
    assert lossy == 2  # valuesize 64, both models: the difference of magnitude 2^63 is coded like 0 (seg.c:25-28 wraps)


def _sim64(sim):
    sim.sim_encode64.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    sim.sim_decode64.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    return sim


def _roundtrip64(gold64, sizes, encode, decode_var):
    for vs in sizes:
        x = gold64["vs%d.x" % vs]
        T, Cn = x.shape
        xin = np.ascontiguousarray(x.view(np.int64))
        for ad, tag in ((1, "ad"), (0, "st")):
            out, bits, err = encode(xin, vs, ad)
            check_streams64(gold64, vs, tag, out, bits, err)
            ok = err == 0
            y, counts, derr = decode_var(out, np.where(ok, bits, 0).astype(np.uint64), T + 3, vs, ad)
            want = gold64["vs%d.%s.dec" % (vs, tag)]
            assert (derr[ok] == gold64["vs%d.%s.decerr" % (vs, tag)][ok]).all(), (vs, tag)
            assert (counts[ok] == T).all() and (y[:T].view(np.uint64)[:, ok] == want[:, ok]).all(), (vs, tag)


def test_kernel_logic_with_valuesize_above_32(sim, gold64):
    _sim64(sim)

    def encode(xin, vs, ad):
        T, Cn = xin.shape
        cap = 4 * ((T * (2 * vs + 3) // 4 + 64) // 4 + 4)
        out = np.zeros((Cn, cap), dtype=np.uint8)
        bits = np.zeros(Cn, dtype=np.uint64)
        err = np.zeros(Cn, dtype=np.int32)
        sim.sim_encode64(xin.ctypes.data, Cn, T, Cn, ad, vs, out.ctypes.data, cap, bits.ctypes.data, err.ctypes.data)
        return out, bits, err

    def decode_var(out, bits, room, vs, ad):
        Cn, cap = out.shape
        y = np.zeros((room, Cn), dtype=np.int64)
        counts = np.zeros(Cn, dtype=np.uint64)
        derr = np.zeros(Cn, dtype=np.int32)
        sim.sim_decode64(out.ctypes.data, cap, bits.ctypes.data, Cn, room, Cn, ad, vs, y.ctypes.data, counts.ctypes.data, derr.ctypes.data)
        return y, counts, derr
    _roundtrip64(gold64, (40, 64), encode, decode_var)


def test_last_partial_word_waits_for_its_ring_slot(sim):
    """Worst-case codewords (125 bits each) up to the last row, and a coding wave that is slow to take words (emulator
    knob): the filling wave's last batch leaves the seg-bit ring full to the last slot, so the channel's final, partial
    word may only be written once the coder has freed the slot it goes to -- written at once it replaced the oldest
    queued word (found as a rare failure of the test above under load; whether a given run would hit it depends on
    the threads' timing, so this test raises the odds rather than proving the absence)."""
    _sim64(sim)
    sim.sim_set_drag.argtypes = [C.c_int, C.c_int]
    rng = np.random.default_rng(1)
    T, Cn, vs = 64, 64, 63
    x = np.zeros((T, Cn), dtype=np.uint64)
    for c in range(Cn):
        x[1::2, c] = np.uint64(2**62 - 1) - rng.integers(0, 5, T // 2).astype(np.uint64)
    xin = np.ascontiguousarray(x.view(np.int64))
    cap = 4 * ((T * 40 + 64) // 4)
    out = np.zeros((Cn, cap), dtype=np.uint8)
    bits = np.zeros(Cn, dtype=np.uint64)
    err = np.zeros(Cn, dtype=np.int32)
    sim.sim_set_drag(4, 200)  # waves 4..7 of the workgroup are the coding waves: 200 us per step
    try:
        sim.sim_encode64(xin.ctypes.data, Cn, T, Cn, 1, vs, out.ctypes.data, cap, bits.ctypes.data, err.ctypes.data)
    finally:
        sim.sim_set_drag(1 << 30, 0)
    for c in range(Cn):
        d, n = pack_be(x[:, c], vs)
        for name in ("diff", "seg", "bac"):
            r, d, n = orc.stage(name, True, d, n, valuesize=vs, adaptive=1)
            assert r == 0
        assert int(err[c]) == 0 and int(bits[c]) == n and out[c, : (n + 7) // 8].tobytes() == d[: (n + 7) // 8], c


@pytest.mark.gpu
def test_gpu_valuesize_above_32(gold64):
    from __graft_entry__ import load_package
    dca = load_package()
    ctx = dca.Context(0)
    _roundtrip64(gold64, SIZES64, lambda xin, vs, ad: ctx.encode64_host(xin, vs, adaptive=ad),
                 lambda out, bits, room, vs, ad: ctx.decode64_var_host(out, bits, room, vs, adaptive=ad))
    with pytest.raises(dca.DegaError):
        ctx.encode64_host(np.zeros((4, 4), dtype=np.int64), 32)  # 1..32 live in the int32 entry points
    ctx.close()
