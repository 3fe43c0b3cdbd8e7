"""The oracle (oracle/dega_oracle.c, our CPU restatement) against the golden vectors generated from the compiled
reference (tests/golden/, made by tests/golden/make_golden.py) and -- where oracle/_ref was built, i.e. in the build
container -- against the compiled reference itself on random streams.  Bit-exact throughout."""
import gzip
import hashlib
import json
import os

import numpy as np
import pytest

from oracle import orc

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def sha(b):
    return hashlib.sha256(b).hexdigest()


@pytest.fixture(scope="module")
def testfile():
    with open(os.path.join(GOLDEN, "testfile.json")) as f:
        meta = json.load(f)
    with gzip.open(os.path.join(GOLDEN, "input.txt.gz"), "rb") as f:
        raw = f.read()
    assert sha(raw) == meta["input_sha256"]
    v = np.array(raw.split(), dtype=np.float64).astype(np.float32)  # what `decode csv` yields (strtof of column 1)
    return meta, raw, v


def test_testfile_float_stream(testfile):
    meta, _, v = testfile
    assert v.size == 100000
    assert sha(v.tobytes()) == meta["stages"]["float32"]["sha256"]


def test_testfile_every_stage(testfile):
    """normalize -> diff -> seg -> bac adaptive, chained with exact bit lengths, equals the reference at every stage."""
    meta, _, v = testfile
    st = meta["stages"]
    ret, b, n = orc.stage("normalize", True, v.tobytes(), v.size * 32)
    assert ret == 0 and n == 3200000 and sha(b) == st["normalize"]["sha256"]
    ret, b, n = orc.stage("diff", True, b, n)
    assert ret == 0 and sha(b) == st["diff"]["sha256"]
    ret, seg, nseg = orc.stage("seg", True, b, n)
    assert ret == 0 and nseg == st["seg"]["wrote_bytes"] * 8 + st["seg"]["wrote_bits"]
    assert sha(orc.file_bytes(seg, nseg)) == st["seg"]["sha256"]
    ret, bac, nbac = orc.stage("bac", True, seg, nseg, adaptive=1)
    assert ret == 0 and nbac == st["dega_adaptive"]["wrote_bytes"] * 8 + st["dega_adaptive"]["wrote_bits"]
    with open(os.path.join(GOLDEN, "dega_adaptive.bin"), "rb") as f:
        assert orc.file_bytes(bac, nbac) == f.read()
    ret, bac0, nbac0 = orc.stage("bac", True, seg, nseg, adaptive=0)
    assert ret == 0 and sha(orc.file_bytes(bac0, nbac0)) == st["dega_nonadaptive"]["sha256"]


def test_testfile_padded_seg_is_not_canonical(testfile):
    """Feeding bac from the byte-padded seg FILE changes the tail: the chained stream is canonical."""
    meta, _, v = testfile
    ret, b, n = orc.stage("normalize", True, v.tobytes(), v.size * 32)
    ret, b, n = orc.stage("diff", True, b, n)
    ret, seg, nseg = orc.stage("seg", True, b, n)
    ret, bac, nbac = orc.stage("bac", True, seg, 8 * len(seg), adaptive=1)
    assert ret == 0 and sha(orc.file_bytes(bac, nbac)) != meta["stages"]["dega_adaptive"]["sha256"]


def test_testfile_roundtrip_from_file_bytes(testfile):
    """Decoding the FILE (zero padded) recovers the float stream exactly, like the second half of `make test`."""
    meta, raw, v = testfile
    with open(os.path.join(GOLDEN, "dega_adaptive.bin"), "rb") as f:
        data = f.read()
    ret, out = orc.decode_f32(data, 8 * len(data), v.size + 8, 100.0, 1)
    assert ret == 0 and out.size == v.size and out.tobytes() == v.tobytes()
    text = "".join("%.2f\n" % x for x in out).encode()
    assert text == raw  # encode csv num_decimal_places=2 (csv.c:46-64)
    assert meta["roundtrip_identical"] is True


def test_whole_chain_helpers_equal_stagewise(testfile):
    meta, _, v = testfile
    ret, b, n = orc.encode_f32(v, 100.0, 1)
    assert ret == 0 and sha(orc.file_bytes(b, n)) == meta["stages"]["dega_adaptive"]["sha256"]
    ret, bn, nn = orc.stage("normalize", True, v.tobytes(), v.size * 32)
    ints = np.frombuffer(bn, dtype=">i4").astype(np.int32)
    ret, b2, n2 = orc.encode_i32(ints, 1)
    assert (ret, b2, n2) == (0, b, n)


def test_kats():
    with open(os.path.join(GOLDEN, "kats.json")) as f:
        kats = json.load(f)
    assert kats["empty/adaptive"]["hex"] == "20" and kats["zero/adaptive"]["hex"] == "58"
    assert kats["zeros96/adaptive"]["hex"] == "aa9c40"
    assert kats["max_zero/adaptive"]["hex"] == "ff8796ce83b69d3ffa29dffd6e9990ab51f0"
    assert kats["sign_change/adaptive"]["ret"] == orc.ERROR_INVALID_VALUE
    assert kats["empty_diff_file"]["file_hex"] == "00"
    for name, k in kats.items():
        if "x" not in k:
            continue
        x = np.array(k["x"], dtype=np.int32)
        ret, b, n = orc.encode_i32(x, k["adaptive"])
        assert ret == k["ret"], name
        if ret == 0:
            assert b.hex() == k["hex"] and n == k["nbits"], name
            rd, y = orc.decode_i32(b, n, x.size, k["adaptive"])
            assert rd == 0 and (y == x).all(), name
            rd, y = orc.decode_i32(b, 8 * len(b), x.size + 4, k["adaptive"])  # from a zero-padded file
            assert rd == 0 and (y == x).all(), name
    ret, b, n = orc.stage("diff", True, b"", 0)
    assert ret == 0 and n == 0 and orc.file_bytes(b, n) == b"\0"


def test_channel_batches():
    z = np.load(os.path.join(GOLDEN, "channels.npz"))
    names = sorted(k[:-2] for k in z.files if k.endswith(".x"))
    assert len(names) >= 8
    for name in names:
        x = z[name + ".x"]
        T, Cn = x.shape
        for ad, tag in ((1, "ad"), (0, "st")):
            stream, bits, err = z["%s.%s.stream" % (name, tag)], z["%s.%s.bits" % (name, tag)], z["%s.%s.err" % (name, tag)]
            out, obits, oerr = orc.encode_batch_tc(x, ad)
            assert (oerr == err).all(), name
            ok = err == 0
            assert (obits[ok] == bits[ok]).all(), name
            for c in np.nonzero(ok)[0]:
                nb = (int(bits[c]) + 7) // 8
                assert out[c, :nb].tobytes() == stream[c, :nb].tobytes(), (name, ad, c)
            y, derr = orc.decode_batch_tc(stream[ok], bits[ok], T, ad)
            assert (derr == 0).all() and (y == x[:, ok]).all(), name
    assert (z["with_errors.ad.err"] != 0).sum() == 2  # the two channels holding a negative sample


def test_float_entry():
    z = np.load(os.path.join(GOLDEN, "floats.npz"))
    for factor in (100.0, 1.0, 1000.0, 0.5):
        v = z["v_%g" % factor]
        ret, b, n = orc.stage("normalize", True, v.tobytes(), v.size * 32, factor=factor)
        assert ret == 0
        ints = np.frombuffer(b, dtype=">i4").astype(np.int32)
        assert (ints == z["norm_%g" % factor]).all(), factor
        ret, b2, n2 = orc.stage("normalize", False, b, n, factor=factor)
        assert ret == 0 and b2 == z["denorm_%g" % factor].tobytes(), factor
    for e, want in zip(z["edge_v"], z["edge_ret"]):
        ret, _, _ = orc.stage("normalize", True, np.array([e], dtype=np.float32).tobytes(), 32)
        assert ret == want == orc.ERROR_INVALID_VALUE
    ret, b, n = orc.stage("normalize", True, np.array([21474836.48], dtype=np.float32).tobytes(), 32)
    assert ret == 0 and b == z["two31_hex"].tobytes() == bytes.fromhex("80000000")  # 2^31 passes the check and wraps


def test_error_paths():
    # truncated value: short read -> ERROR_LIBRARY_CALL (io_macros.h:13-27)
    assert orc.stage("diff", True, b"\x00\x00\x01", 24)[0] == orc.ERROR_LIBRARY_CALL
    # seg prefix longer than valuesize+1 -> ERROR_INVALID_FORMAT (seg.c:55-56)
    assert orc.stage("seg", False, b"\x00" * 5 + b"\x01", 48)[0] == orc.ERROR_INVALID_FORMAT
    # bac: more than 14 phantom bits needed -> ERROR_INVALID_FORMAT (bac.c:171-186)
    assert orc.stage("bac", False, b"", 0, adaptive=1)[0] == orc.ERROR_INVALID_FORMAT
    # padding zeros at the end of a seg file are swallowed (seg.c:58-62)
    ret, b, n = orc.stage("seg", False, bytes([0b10100000]), 8)
    assert ret == 0 and np.frombuffer(b, dtype=">i4").tolist() == [0, 1]


@pytest.mark.skipif(not orc.have_ref(), reason="oracle/_ref not built (needs /root/reference)")
def test_restatement_equals_compiled_reference_on_random_streams():
    rng = np.random.default_rng(99)
    for it in range(120):
        T = int(rng.integers(0, 500))
        kind = it % 6
        if kind == 0:
            x = np.cumsum(rng.integers(-50, 51, T)) + 20000
        elif kind == 1:
            x = rng.integers(0, 2**31, T)
        elif kind == 2:
            x = np.abs(np.cumsum(rng.integers(-3000, 3001, T)))
        elif kind == 3:
            x = rng.integers(0, 2, T)
        elif kind == 4:
            x = rng.integers(-5, 2**31, T)
        else:
            x = np.cumsum(rng.integers(-300, 301, T)) + 5000
        x = np.asarray(x, dtype=np.int64).clip(-2**31, 2**31 - 1).astype(np.int32)
        for ad in (0, 1):
            got = orc.encode_i32(x, ad)
            rr, rb, rn, _ = orc.ref_encode_i32(x, ad)
            assert got == (rr, rb, rn), (it, ad)
            if rr == 0:
                assert orc.decode_i32(rb, rn, T, ad)[1].tolist() == orc.ref_decode_i32(rb, rn, T, ad)[1].tolist() == x.tolist()
                a = orc.decode_i32(rb, 8 * len(rb), T + 8, ad)
                b = orc.ref_decode_i32(rb, 8 * len(rb), T + 8, ad)
                assert a[0] == b[0] and a[1].tolist() == b[1].tolist()
    # stage level, random garbage into the decoders: same error code or same output
    for it in range(200):
        nbits = int(rng.integers(0, 300))
        data = rng.integers(0, 256, (nbits + 7) // 8, dtype=np.uint8).tobytes()
        if nbits % 8:
            data = data[:-1] + bytes([data[-1] & (0xFF00 >> (nbits % 8)) & 0xFF])
        for name, spec in (("bac", "decode bac adaptive"), ("bac", "decode bac"), ("seg", "decode seg")):
            ad = 1 if spec.endswith("adaptive") else 0
            a = orc.stage(name, False, data, nbits, adaptive=ad)
            r = orc.ref_run_chain(data, nbits, [spec])
            assert a[0] == r[0], (it, spec, a[0], r[0])
            if a[0] == 0:
                assert (a[1], a[2]) == (r[1], r[2]), (it, spec)


def test_float_entry_at_other_value_sizes():
    """The restatement's normalize -> diff -> seg -> bac chain at valuesize 8..64 (io_int_t = int64 arithmetic of
    normalize.c:21-24 included) against the compiled reference's streams, error verdicts and decoded floats
    (tests/golden/floats_vs.npz)."""
    z = np.load(os.path.join(GOLDEN, "floats_vs.npz"))
    tags = sorted({k.split(".")[0] for k in z.files}, key=lambda t: int(t[2:]))
    assert len(tags) >= 10
    for tag in tags:
        vs, f = int(tag[2:]), float(z[tag + ".factor"][0])
        v = z[tag + ".v"]
        for c in range(v.shape[1]):
            col = np.ascontiguousarray(v[:, c])
            data, n, ret = col.tobytes(), col.size * 32, 0
            for name in ("normalize", "diff", "seg", "bac"):
                ret, data, n = orc.stage(name, True, data, n, valuesize=vs, adaptive=1, factor=f)
                if ret != 0:
                    break
            assert ret == int(z[tag + ".err"][c]), (tag, c)
            if ret != 0:
                continue
            nb = int(z[tag + ".bits"][c])
            assert n == nb and data[: (n + 7) // 8] == z[tag + ".stream"][c, : (nb + 7) // 8].tobytes(), (tag, c)
            for name in ("bac", "seg", "diff", "normalize"):
                r, data, n = orc.stage(name, False, data, n, valuesize=vs, adaptive=1, factor=f)
                assert r == 0, (tag, c, name)
            assert data == np.ascontiguousarray(z[tag + ".back"][:, c]).tobytes(), (tag, c)
