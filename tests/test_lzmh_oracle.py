"""The LZMH restatement (oracle/lzmh_oracle.c) against the fixtures made from the compiled reference
(tests/golden/lzmh.json, lzmh.npz) and, where oracle/_ref exists, against the compiled reference on random inputs."""
import gzip
import hashlib
import json
import os

import numpy as np
import pytest

from oracle import orc

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def fixtures():
    with open(os.path.join(GOLDEN, "lzmh.json")) as f:
        meta = json.load(f)
    return meta, np.load(os.path.join(GOLDEN, "lzmh.npz"))


def test_lzmh_testfile_golden(fixtures):
    meta = fixtures[0]["testfile"]
    with gzip.open(os.path.join(GOLDEN, "input.txt.gz"), "rb") as f:
        raw = f.read()
    ret, b, n = orc.stage("lzmh", True, raw, 8 * len(raw))
    assert ret == 0
    data = orc.file_bytes(b, n)
    assert len(data) == meta["file_bytes"] == 270896
    assert (n // 8, n % 8) == (meta["wrote_bytes"], meta["wrote_bits"])
    assert hashlib.sha256(data).hexdigest() == meta["sha256"]
    assert meta["sha256"].startswith("c493269c") and meta["roundtrip_identical"]
    ret, d, dn = orc.stage("lzmh", False, b, n)
    assert ret == 0 and dn == 8 * len(raw) and d == raw


def test_lzmh_small_goldens(fixtures):
    meta, z = fixtures
    names = sorted(k[:-3] for k in z.files if k.endswith(".in"))
    assert len(names) >= 20
    for name in names:
        data = z[name + ".in"].tobytes()
        ret, b, n = orc.stage("lzmh", True, data, 8 * len(data))
        assert ret == 0 and n == int(z[name + ".bits"][0]), name
        assert b[: (n + 7) // 8] == z[name + ".stream"].tobytes(), name
        ret, d, dn = orc.stage("lzmh", False, b, n)
        assert ret == 0 and d[: dn // 8] == z[name + ".dec"].tobytes(), name
        assert (d[: dn // 8] == data) == meta[name]["decodes_to_input"], name
    # the reference's quirks are part of the fixtures: exactly one ring of input encodes to nothing, and the decoder
    # turns an empty stream into one zero byte
    assert meta["digits_403"]["bits"] == 0 and meta["digits_0"]["decoded_bytes"] == 1


def test_lzmh_rejects_partial_byte():
    assert orc.stage("lzmh", True, b"\xff\x80", 9)[0] == orc.ERROR_LIBRARY_CALL


@pytest.mark.skipif(not orc.have_ref(), reason="oracle/_ref not built (needs /root/reference)")
def test_lzmh_restatement_equals_compiled_reference_on_random_inputs():
    rng = np.random.default_rng(3)
    for it in range(300):
        kind = it % 5
        L = int(rng.integers(0, 1500)) if it % 7 else [0, 1, 2, 3, 402, 403, 404, 405, 806][it % 9]
        if kind == 0:
            data = bytes(rng.integers(0, 256, L, dtype=np.uint8))
        elif kind == 1:
            data = ("".join("%.2f\n" % v for v in rng.uniform(0, 400, L // 6 + 1)))[:L].encode()
        elif kind == 2:
            data = bytes(rng.integers(48, 58, L, dtype=np.uint8))
        elif kind == 3:
            data = bytes(rng.integers(0, 3, L, dtype=np.uint8))
        else:
            data = (b"abcabcabd" * (L // 9 + 1))[:L]
        a = orc.stage("lzmh", True, data, 8 * len(data))
        r = orc.ref_run_chain(data, 8 * len(data), ["encode lzmh"])
        assert a == (r[0], r[1], r[2]), (it, kind, L)
        d = orc.stage("lzmh", False, a[1], a[2])
        dr = orc.ref_run_chain(a[1], a[2], ["decode lzmh"])
        assert d == (dr[0], dr[1], dr[2]), (it, kind, L)
        # garbage into the decoder: the reference reads never-written list symbols there (lzmh.c:395, :434), so only
        # the status and the length are defined
        g = bytes(rng.integers(0, 256, 40, dtype=np.uint8))
        d = orc.stage("lzmh", False, g, 8 * len(g))
        dr = orc.ref_run_chain(g, 8 * len(g), ["decode lzmh"])
        assert (d[0], d[2]) == (dr[0], dr[2]), it


@pytest.mark.skipif(not orc.have_ref(), reason="oracle/_ref not built (needs /root/reference)")
def test_lzmh_restatement_equals_compiled_reference_on_long_inputs():
    """Count saturation at 65 535 (lzmh.c:304) and back-to-back 274-byte matches: beyond what the short fixtures reach."""
    rng = np.random.default_rng(42)
    for data in (bytes(rng.integers(48, 58, 1_500_000, dtype=np.uint8)), bytes(600_000)):
        a = orc.stage("lzmh", True, data, 8 * len(data))
        r = orc.ref_run_chain(data, 8 * len(data), ["encode lzmh"])
        assert a == (r[0], r[1], r[2])
        d = orc.stage("lzmh", False, a[1], a[2])
        dr = orc.ref_run_chain(a[1], a[2], ["decode lzmh"])
        assert d == (dr[0], dr[1], dr[2]) and d[1] == data
