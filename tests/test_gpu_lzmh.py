"""GPU parity tests for the LZMH row (BASELINE config 4), through the C ABI (libdega_hip.so): the HIP kernels against
the fixtures made from the compiled reference (tests/golden/lzmh.*), the reference's own test file, and the oracle on
seeded inputs.  Bit-exact: stream bytes, exact bit lengths, decoded bytes, per-channel status."""
import gzip
import hashlib
import json
import os

import numpy as np
import pytest

from __graft_entry__ import load_package
from oracle import orc

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def dca():
    return load_package()


@pytest.fixture(scope="module")
def ctx(dca):
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU; the product has no CPU fallback"
    c = dca.Context(0)
    yield c
    c.close()


def make_strings(rng, count, max_len):
    strings = []
    for it in range(count):
        kind = it % 7
        n = int(rng.integers(0, max_len)) if it % 11 else [0, 1, 2, 3, 402, 403, 404, 405, 806][it % 9]
        if kind == 0:
            s = bytes(rng.integers(0, 256, n, dtype=np.uint8))
        elif kind in (1, 6):
            s = "".join("%.2f\n" % v for v in 230 + np.cumsum(rng.normal(0, 0.3, n // 6 + 1))).encode()[:n]
        elif kind == 2:
            s = bytes(rng.integers(48, 58, n, dtype=np.uint8))
        elif kind == 3:
            s = bytes(rng.integers(0, 3, n, dtype=np.uint8))
        elif kind == 4:
            s = (b"abcabcabd" * (n // 9 + 1))[:n]
        else:
            s = bytes(n)
        strings.append(s)
    return strings


def test_lzmh_goldens(ctx):
    z = np.load(os.path.join(GOLDEN, "lzmh.npz"))
    names = sorted(k[:-3] for k in z.files if k.endswith(".in"))
    strings = [z[n + ".in"].tobytes() for n in names]
    out, bits, err = ctx.lzmh_encode_host(strings)
    assert (err == 0).all()
    for i, n in enumerate(names):
        nb = int(z[n + ".bits"][0])
        assert int(bits[i]) == nb, n
        assert out[i, : (nb + 7) // 8].tobytes() == z[n + ".stream"].tobytes(), n
    dec, lens, derr = ctx.lzmh_decode_host(out, bits, 5008)
    assert (derr == 0).all()
    for i, n in enumerate(names):
        want = z[n + ".dec"].tobytes()  # what the reference's decoder makes of it (not always the input: see lzmh.json)
        assert int(lens[i]) == len(want) and dec[i, : len(want)].tobytes() == want, n


def test_lzmh_reference_test_file(ctx):
    """`encode lzmh` of DCCLI/testdata/input.txt: 270 896 bytes, sha256 c493269c...; decodes back."""
    with open(os.path.join(GOLDEN, "lzmh.json")) as f:
        meta = json.load(f)["testfile"]
    with gzip.open(os.path.join(GOLDEN, "input.txt.gz"), "rb") as f:
        raw = f.read()
    out, bits, err = ctx.lzmh_encode_host([raw])
    assert err[0] == 0
    n = int(bits[0])
    assert (n // 8, n % 8) == (meta["wrote_bytes"], meta["wrote_bits"])
    data = out[0, : (n + 7) // 8].tobytes()
    assert len(data) == meta["file_bytes"] == 270896 and hashlib.sha256(data).hexdigest() == meta["sha256"]
    dec, lens, derr = ctx.lzmh_decode_host(out, bits, (len(raw) + 15) // 8 * 8)
    assert derr[0] == 0 and int(lens[0]) == len(raw) and dec[0, : len(raw)].tobytes() == raw
    # the file form (zero padded to a byte) decodes to the same bytes: the padding never completes a code
    dec, lens, derr = ctx.lzmh_decode_host(out, np.array([8 * ((n + 7) // 8)], dtype=np.uint64), (len(raw) + 15) // 8 * 8)
    assert derr[0] == 0 and int(lens[0]) == len(raw) and dec[0, : len(raw)].tobytes() == raw


def test_lzmh_random_batch_vs_oracle(ctx):
    rng = np.random.default_rng(77)
    strings = make_strings(rng, 700, 3000)  # three workgroups, a ragged last wave
    out, bits, err = ctx.lzmh_encode_host(strings)
    assert (err == 0).all()
    streams = []
    for i, s in enumerate(strings):
        r, b, n = orc.stage("lzmh", True, s, 8 * len(s))
        assert r == 0 and int(bits[i]) == n and out[i, : (n + 7) // 8].tobytes() == b[: (n + 7) // 8], (i, len(s))
    dbits = bits.copy()
    dbits[::3] = 8 * ((dbits[::3] + 7) // 8)  # every third stream in its zero-padded file form
    dec, lens, derr = ctx.lzmh_decode_host(out, dbits, 3008)
    assert (derr == 0).all()
    for i, s in enumerate(strings):
        nb = int(dbits[i])
        r, d, dn = orc.stage("lzmh", False, out[i, : (nb + 7) // 8].tobytes(), nb)
        assert r == 0 and int(lens[i]) == dn // 8 and dec[i, : dn // 8].tobytes() == d[: dn // 8], i


def test_lzmh_damaged_streams_match_oracle(ctx):
    rng = np.random.default_rng(5)
    C_ = 256
    data = rng.integers(0, 256, (C_, 64), dtype=np.uint8)
    bits = rng.integers(0, 8 * 60, C_).astype(np.uint64)
    for i in range(C_):  # clear what lies beyond the bit length, like a stream written by the coder
        nb = int(bits[i])
        data[i, (nb + 7) // 8:] = 0
        if nb % 8:
            data[i, nb // 8] &= (0xFF00 >> (nb % 8)) & 0xFF
    dec, lens, derr = ctx.lzmh_decode_host(data, bits, 65536)
    for i in range(C_):
        nb = int(bits[i])
        r, d, dn = orc.stage("lzmh", False, data[i, : (nb + 7) // 8].tobytes(), nb)
        assert derr[i] == r == 0 and int(lens[i]) == dn // 8 and dec[i, : dn // 8].tobytes() == d[: dn // 8], i


def test_lzmh_slab_too_small(ctx):
    rng = np.random.default_rng(2)
    s = bytes(rng.integers(0, 256, 900, dtype=np.uint8))
    out, bits, err = ctx.lzmh_encode_host([s, b"1.00\n" * 100], cap=256)
    assert err[0] == orc.ERROR_MEMORY and bits[0] == 0 and err[1] == 0
    r, b, n = orc.stage("lzmh", True, b"1.00\n" * 100, 4000)
    assert int(bits[1]) == n and out[1, : (n + 7) // 8].tobytes() == b[: (n + 7) // 8]
    dec, lens, derr = ctx.lzmh_decode_host(out[1:2], bits[1:2], 256)
    assert derr[0] == orc.ERROR_MEMORY and lens[0] == 0


def test_lzmh_cfg4_workload_device_resident(ctx, dca):
    """BASELINE config 4 in small: synthetic channels rendered as ASCII lines on the device, encoded, decoded and compared
    on the device; a sample of channels against the oracle byte for byte."""
    import torch
    C_, T = 4096, 2000
    x = ctx.synth(C_, T, seed=1234, c0=0, S=50)
    stride = 16 * ((T * 9 + 15) // 16)
    text, lens, rerr = ctx.lzmh_render(x, stride)
    assert int((rerr != 0).sum().item()) == 0
    xs = x[:, :4].cpu().numpy()
    for c in range(4):
        want = "".join("%d.%02d\n" % (v // 100, v % 100) for v in xs[:, c].tolist()).encode()
        assert int(lens[c].item()) == len(want) and text[c, : len(want)].cpu().numpy().tobytes() == want
    out, bits, err = ctx.lzmh_encode(text, lens)
    assert int((err != 0).sum().item()) == 0
    back, blens, derr = ctx.lzmh_decode(out, bits, stride)
    torch.cuda.synchronize()
    assert int((derr != 0).sum().item()) == 0 and bool((blens == lens).all().item())
    idx = torch.arange(stride, device=text.device)[None, :] < lens[:, None]
    assert bool(((back == text) | ~idx).all().item())
    for c in range(0, C_, 257):
        n = int(lens[c].item())
        r, b, nb = orc.stage("lzmh", True, text[c, :n].cpu().numpy().tobytes(), 8 * n)
        assert r == 0 and int(bits[c].item()) == nb and out[c, : (nb + 7) // 8].cpu().numpy().tobytes() == b[: (nb + 7) // 8], c


def test_lzmh_long_channels_count_saturation_and_long_runs(ctx):
    """1.5 MB of random digits (the list counts run into their 65 535 cap, lzmh.c:304) and 600 kB of zeros (a 274-byte
    match per step: the window-reload-and-retry path all the way), next to short channels in the same wave."""
    rng = np.random.default_rng(42)
    strings = [bytes(rng.integers(48, 58, 1_500_000, dtype=np.uint8)), bytes(600_000), b"12.50\n" * 50, b""]
    out, bits, err = ctx.lzmh_encode_host(strings)
    assert (err == 0).all()
    for i, s in enumerate(strings):
        r, b, n = orc.stage("lzmh", True, s, 8 * len(s))
        assert r == 0 and int(bits[i]) == n and out[i, : (n + 7) // 8].tobytes() == b[: (n + 7) // 8], i
    dec, lens, derr = ctx.lzmh_decode_host(out, bits, 1_500_008)
    assert (derr == 0).all()
    for i, s in enumerate(strings[:3]):
        assert int(lens[i]) == len(s) and dec[i, : len(s)].tobytes() == s, i
    assert int(lens[3]) == 1  # the reference's decoder turns the empty stream into one zero byte (lzmh.json: digits_0)


def test_lzmh_group_pipeline_packs_and_splits(dca, ctx):
    """dega_hip_group_lzmh_encode / _decode: the host pipeline (chunks of channels on their own streams, packed streams
    back) on a group of one and of two members (with one visible GPU both share device 0: the partition, the threads and
    the host-side concatenate are the same code): the packed streams are the single-context slab call's, byte for byte,
    and decode to the text; pageable and pinned memory; a packed buffer too small reports the size needed."""
    rng = np.random.default_rng(321)
    strings = make_strings(rng, 1100, 900)  # more than two 512-channel chunks
    Cn = len(strings)
    stride = (max(len(s) for s in strings) + 16) // 16 * 16
    text = np.zeros((Cn, stride), dtype=np.uint8)
    lens = np.array([len(s) for s in strings], dtype=np.uint64)
    for i, s in enumerate(strings):
        text[i, : len(s)] = np.frombuffer(s, dtype=np.uint8)
    want_out, want_bits, want_err = ctx.lzmh_encode_host(strings)
    assert (want_err == 0).all()
    want_back, want_lens, want_derr = ctx.lzmh_decode_host(want_out, want_bits, stride)
    assert (want_derr == 0).all()
    os.environ["DEGA_PIPELINE_CHUNKS"] = "3"
    try:
        for devices in ([0], [0, 0]):
            g = dca.Group(devices)
            try:
                pinned = dca.PinnedArray((Cn, stride), np.uint8)
                pinned.array[:] = text
                for src in (text, pinned.array):
                    packed, offsets, bits, err = g.lzmh_encode_job(src, lens)
                    assert (err == 0).all() and (bits == want_bits).all()
                    assert int(offsets[0]) == 0 and (np.diff(offsets.astype(np.int64)) == (bits.astype(np.int64) + 7) // 8).all()
                    for c in range(Cn):
                        nb = (int(bits[c]) + 7) // 8
                        assert packed[int(offsets[c]): int(offsets[c]) + nb].tobytes() == want_out[c, :nb].tobytes(), (devices, c)
                    back, blens, berr = g.lzmh_decode_job(packed, offsets, bits, stride)
                    assert (berr == 0).all() and (blens == want_lens).all()  # (the codec's quirks included: an empty stream decodes to one byte)
                    for c in range(Cn):
                        assert back[c, : int(want_lens[c])].tobytes() == want_back[c, : int(want_lens[c])].tobytes(), (devices, c)
                        assert len(strings[c]) in (0, 403) or back[c, : len(strings[c])].tobytes() == strings[c], (devices, c)
                pinned.free()
                small = np.empty(100, dtype=np.uint8)
                with pytest.raises(dca.DegaError) as e:
                    g.lzmh_encode_job(text, lens, packed=small)
                assert e.value.code == -6
            finally:
                g.close()
    finally:
        del os.environ["DEGA_PIPELINE_CHUNKS"]
