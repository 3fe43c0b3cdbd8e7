"""GPU parity tests proper (-m gpu): the HIP path, called through the C ABI (libdega_hip.so), against
  * the golden vectors generated from the compiled reference (tests/golden/), and
  * the oracle (oracle/liboracle.so) on the same seeded inputs.
Bit-exact: stream bytes, exact bit lengths and per-channel error codes."""
import gzip
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


def assert_streams_equal(out, bits, err, want_out, want_bits, want_err, tag=""):
    assert (err == want_err).all(), (tag, err[:8], want_err[:8])
    ok = want_err == 0
    assert (bits[ok] == want_bits[ok]).all(), tag
    for c in np.nonzero(ok)[0]:
        nb = (int(want_bits[c]) + 7) // 8
        assert out[c, :nb].tobytes() == want_out[c, :nb].tobytes(), (tag, int(c))


def test_golden_channel_batches(ctx):
    z = np.load(os.path.join(GOLDEN, "channels.npz"))
    names = sorted(k[:-2] for k in z.files if k.endswith(".x"))
    for name in names:
        x = z[name + ".x"]
        for ad, tag in ((1, "ad"), (0, "st")):
            out, bits, err = ctx.encode_host(x, adaptive=ad)
            assert_streams_equal(out, bits, err, z["%s.%s.stream" % (name, tag)], z["%s.%s.bits" % (name, tag)],
                                 z["%s.%s.err" % (name, tag)], (name, tag))


def test_kats(ctx):
    with open(os.path.join(GOLDEN, "kats.json")) as f:
        kats = json.load(f)
    for name, k in kats.items():
        if "x" not in k:
            continue
        x = np.array(k["x"], dtype=np.int32).reshape(-1, 1)
        out, bits, err = ctx.encode_host(x, adaptive=k["adaptive"], cap=64 + 4 * ((x.size * 20 + 3) // 4))
        assert int(err[0]) == k["ret"], name
        if k["ret"] == 0:
            assert int(bits[0]) == k["nbits"], name
            assert out[0, : (k["nbits"] + 7) // 8].tobytes().hex() == k["hex"], name


def test_reference_test_file(ctx):
    """The reference's own test file through normalize + DEGA on the GPU equals the canonical stream of `make test`."""
    with gzip.open(os.path.join(GOLDEN, "input.txt.gz"), "rb") as f:
        v = np.array(f.read().split(), dtype=np.float64).astype(np.float32)
    with open(os.path.join(GOLDEN, "dega_adaptive.bin"), "rb") as f:
        want = f.read()
    with open(os.path.join(GOLDEN, "testfile.json")) as f:
        meta = json.load(f)["stages"]["dega_adaptive"]
    # 70 copies: more than one wave, a ragged last wave
    vt = np.repeat(v.reshape(-1, 1), 70, axis=1)
    out, bits, err = ctx.encode_f32_host(vt, factor=100.0, adaptive=1, cap=4 * ((len(want) + 64) // 4))
    assert (err == 0).all()
    assert (bits == meta["wrote_bytes"] * 8 + meta["wrote_bits"]).all()
    for c in (0, 1, 63, 64, 69):
        assert out[c, : len(want)].tobytes() == want


def test_normalize_kernel_golden(ctx):
    import torch
    z = np.load(os.path.join(GOLDEN, "floats.npz"))
    for factor in (100.0, 1.0, 1000.0, 0.5):
        v = z["v_%g" % factor]
        vt = torch.from_numpy(np.stack([v, v[::-1].copy()], axis=1)).cuda()
        x, err = ctx.normalize(vt, factor)
        torch.cuda.synchronize()
        assert (err.cpu().numpy() == 0).all()
        xn = x.cpu().numpy()
        assert (xn[:, 0] == z["norm_%g" % factor]).all() and (xn[::-1, 1] == z["norm_%g" % factor]).all(), factor
        d = ctx.denormalize(x, factor).cpu().numpy()
        assert d[:, 0].tobytes() == z["denorm_%g" % factor].tobytes(), factor
    edge = np.concatenate([z["edge_v"], np.array([21474836.48, 1.0], dtype=np.float32)])
    x, err = ctx.normalize(torch.from_numpy(edge.reshape(1, -1).copy()).cuda(), 100.0)
    torch.cuda.synchronize()
    assert err.cpu().numpy().tolist() == [-1, -1, -1, -1, 0, 0]
    assert x.cpu().numpy()[0, 4] == -2147483648 and x.cpu().numpy()[0, 5] == 100


def ctx_worst(T):
    return load_package().worst_case_bytes(T)


def test_random_batch_vs_oracle(ctx):
    rng = np.random.default_rng(42)
    for (T, Cn, S, ld_pad) in ((300, 1000, 50, 0), (96, 3000, 300, 0), (1000, 130, 5000, 7), (17, 64, 2, 0), (1, 257, 1000, 0)):
        x = np.zeros((T, Cn), dtype=np.int64)
        x[0] = rng.integers(0, 60000, Cn)
        steps = rng.integers(-S, S + 1, (T, Cn)) * rng.integers(0, 3, Cn)[None, :]
        for t in range(1, T):
            x[t] = np.clip(x[t - 1] + steps[t], 0, 2**31 - 1)
        x = x.astype(np.int32)
        for ad in (1, 0):
            cap = 4 * ((T * 16 + 67) // 4) if S < 1000 else ctx_worst(T)
            want = orc.encode_batch_tc(x, ad, cap=cap)
            if ld_pad:
                import torch
                xt = torch.zeros((T, Cn + ld_pad), dtype=torch.int32, device="cuda")
                xt[:, :Cn] = torch.from_numpy(x).cuda()
                out, bits, err = ctx.encode(xt, adaptive=ad, cap=want[0].shape[1])
                torch.cuda.synchronize()
                got = (out.cpu().numpy()[:Cn], bits.cpu().numpy().astype(np.uint64)[:Cn], err.cpu().numpy()[:Cn])
            else:
                got = ctx.encode_host(x, adaptive=ad, cap=want[0].shape[1])
            assert_streams_equal(*got, *want, tag=(T, Cn, S, ad))


def test_slab_too_small_reports_memory_error(ctx):
    x = np.cumsum(np.random.default_rng(3).integers(-50, 51, (400, 65)), axis=0).astype(np.int32) + 30000
    out, bits, err = ctx.encode_host(x, adaptive=1, cap=64)
    assert (err == -6).all()  # ERROR_MEMORY: 400 samples cannot fit 64 bytes
    out2, bits2, err2 = ctx.encode_host(x, adaptive=1)
    assert (err2 == 0).all() and (bits2 == bits).all()  # the length is still reported
    assert (out[:, :60] == out2[:, :60]).all()


def test_synth_matches_numpy_definition(ctx, dca):
    import torch
    x = ctx.synth(130, 60, seed=1234, c0=5, S=50)
    torch.cuda.synchronize()
    assert (x.cpu().numpy() == dca.synth_reference(130, 60, seed=1234, c0=5, S=50)).all()
    x = ctx.synth(70, 96, seed=99, c0=0, S=300)
    torch.cuda.synchronize()
    assert (x.cpu().numpy() == dca.synth_reference(70, 96, seed=99, c0=0, S=300)).all()


def test_compaction(ctx):
    import torch
    x = ctx.synth(300, 200, seed=7)
    out, bits, err = ctx.encode(x, adaptive=1)
    packed, offsets = ctx.compact(out, bits)
    torch.cuda.synchronize()
    o, b, p, off = out.cpu().numpy(), bits.cpu().numpy(), packed.cpu().numpy(), offsets.cpu().numpy()
    nbytes = (b + 7) // 8
    assert (off[1:] - off[:-1] == nbytes).all() and off[0] == 0
    for c in (0, 1, 63, 64, 255, 256, 299):
        assert p[off[c]: off[c + 1]].tobytes() == o[c, : nbytes[c]].tobytes()


def test_larger_batch_sampled_against_oracle(ctx):
    """8192 channels x 3000 samples (crosses the first model halvings): every 97th channel compared with the oracle,
    all lengths compared with the lengths the oracle gives for those channels, all statuses zero."""
    import torch
    Cn, T = 8192, 3000
    x = ctx.synth(Cn, T, seed=1234, S=50)
    cap = 4 * ((T * 3 + 67) // 4)
    out, bits, err = ctx.encode(x, adaptive=1, cap=cap)
    torch.cuda.synchronize()
    assert (err.cpu().numpy() == 0).all()
    sel = np.arange(0, Cn, 97)
    xs = x[:, torch.from_numpy(sel).cuda()].cpu().numpy()
    want = orc.encode_batch_tc(xs, 1, cap=cap)
    got_out = out.cpu().numpy()[sel]
    got_bits = bits.cpu().numpy().astype(np.uint64)[sel]
    assert_streams_equal(got_out, got_bits, np.zeros(len(sel), dtype=np.int32), *want, tag="large")


# ---------------------------------------------------------------------------------------------------------------------
# decode (bac -> seg -> prefix sum on the GPU)
# ---------------------------------------------------------------------------------------------------------------------

def pad4(streams):
    Cn, cap = streams.shape
    if cap % 4 == 0:
        return np.ascontiguousarray(streams)
    out = np.zeros((Cn, (cap + 3) & ~3), dtype=np.uint8)
    out[:, :cap] = streams
    return out


def test_decode_golden_channel_batches(ctx):
    """The reference's own streams (tests/golden, generated by the compiled reference) decode to the original samples,
    from exact bit lengths and from byte-padded ("file") lengths."""
    z = np.load(os.path.join(GOLDEN, "channels.npz"))
    names = sorted(k[:-2] for k in z.files if k.endswith(".x"))
    for name in names:
        x = z[name + ".x"]
        T = x.shape[0]
        for ad, tag in ((1, "ad"), (0, "st")):
            st, gb, ge = z["%s.%s.stream" % (name, tag)], z["%s.%s.bits" % (name, tag)], z["%s.%s.err" % (name, tag)]
            ok = ge == 0
            y, err = ctx.decode_host(pad4(st[ok]), gb[ok], T, adaptive=ad)
            assert (err == 0).all() and (y == x[:, ok]).all(), (name, tag)
            y, err = ctx.decode_host(pad4(st[ok]), ((gb[ok] + 7) // 8) * 8, T, adaptive=ad)
            assert (err == 0).all() and (y == x[:, ok]).all(), (name, tag, "padded")


def test_decode_reference_test_file(ctx):
    with gzip.open(os.path.join(GOLDEN, "input.txt.gz"), "rb") as f:
        v = np.array(f.read().split(), dtype=np.float64).astype(np.float32)
    with open(os.path.join(GOLDEN, "dega_adaptive.bin"), "rb") as f:
        data = f.read()
    cap = (len(data) + 3) & ~3
    st = np.zeros((66, cap), dtype=np.uint8)
    st[:, : len(data)] = np.frombuffer(data, dtype=np.uint8)
    bits = np.full(66, 8 * len(data), dtype=np.uint64)  # the file: zero padded to a byte
    out, err = ctx.decode_f32_host(st, bits, v.size, factor=100.0, adaptive=1)
    assert (err == 0).all()
    for c in (0, 1, 63, 64, 65):
        assert out[:, c].tobytes() == v.tobytes()  # decode bac # decode seg # decode diff # decode normalize


def test_decode_error_codes(ctx):
    x = (np.cumsum(np.random.default_rng(8).integers(-50, 51, (200, 70)), axis=0) + 30000).astype(np.int32)
    out, bits, err = ctx.encode_host(x, adaptive=1)
    assert (err == 0).all()
    # asking for the wrong number of samples
    y, derr = ctx.decode_host(out, bits, 199, adaptive=1)
    assert (derr == -3).all()
    y, derr = ctx.decode_host(out, bits, 201, adaptive=1)
    assert (derr == -3).all()
    # truncated streams: the decoder runs out of bits (more than 14 phantom bits) or of samples
    y, derr = ctx.decode_host(out, bits // 2, 200, adaptive=1)
    assert (derr != 0).all()
    # empty stream: ERROR_INVALID_FORMAT like the reference's decode bac on an empty file
    y, derr = ctx.decode_host(out, np.zeros(70, dtype=np.uint64), 200, adaptive=1)
    assert (derr == -3).all()
    # the oracle gives the same verdict on a damaged stream
    bad = out.copy()
    bad[:, 40] ^= 0x5A
    y, derr = ctx.decode_host(bad, bits, 200, adaptive=1)
    oy, oerr = orc.decode_batch_tc(bad, bits, 200, 1)
    assert ((derr == 0) == (oerr == 0)).all()
    same = derr == 0
    assert (y[:, same] == oy[:, same]).all()


def test_round_trip_random_batches(ctx):
    rng = np.random.default_rng(77)
    for (T, Cn, S) in ((500, 700, 50), (96, 2000, 300), (1200, 150, 20000), (3, 130, 5), (1, 64, 100000)):
        x = np.zeros((T, Cn), dtype=np.int64)
        x[0] = rng.integers(0, 60000, Cn)
        steps = rng.integers(-S, S + 1, (T, Cn)) * rng.integers(0, 3, Cn)[None, :]
        for t in range(1, T):
            x[t] = np.clip(x[t - 1] + steps[t], 0, 2**31 - 1)
        x = x.astype(np.int32)
        for ad in (1, 0):
            out, bits, err = ctx.encode_host(x, adaptive=ad)
            assert (err == 0).all()
            y, derr = ctx.decode_host(out, bits, T, adaptive=ad)
            assert (derr == 0).all() and (y == x).all(), (T, Cn, S, ad)


def test_round_trip_large_device_resident(ctx):
    """16 Ki channels x 4000 samples, encode -> decode on the device, compared on the device; checksums of the
    streams' bit lengths are compared with the oracle on a sample of channels."""
    import torch
    Cn, T = 16384, 4000
    x = ctx.synth(Cn, T, seed=4321, S=50)
    cap = 4 * ((T * 3 + 67) // 4)
    out, bits, err = ctx.encode(x, adaptive=1, cap=cap)
    y, derr = ctx.decode(out, bits, T, adaptive=1)
    torch.cuda.synchronize()
    assert int((err != 0).sum()) == 0 and int((derr != 0).sum()) == 0
    assert bool((y == x).all())


# ---------------------------------------------------------------------------------------------------------------------
# BASELINE.json shapes as parity cases
# ---------------------------------------------------------------------------------------------------------------------

def test_cfg3_short_series_high_batch(ctx):
    """configs[2]: 1 Mi channels x 96 samples (15-min granularity, S = 300).  Every stream round-trips on the device;
    every 4099th channel is compared byte for byte with the oracle."""
    import torch
    Cn, T = 1 << 20, 96
    x = ctx.synth(Cn, T, seed=1234, S=300)
    cap = 4 * ((T * 8 + 67) // 4)
    out, bits, err = ctx.encode(x, adaptive=1, cap=cap)
    y, derr = ctx.decode(out, bits, T, adaptive=1)
    torch.cuda.synchronize()
    assert int((err != 0).sum()) == 0 and int((derr != 0).sum()) == 0
    assert bool((y == x).all())
    sel = np.arange(0, Cn, 4099)
    selt = torch.from_numpy(sel).cuda()
    want = orc.encode_batch_tc(x[:, selt].cpu().numpy(), 1, cap=cap)
    assert_streams_equal(out[selt].cpu().numpy(), bits[selt].cpu().numpy().astype(np.uint64), np.zeros(len(sel), dtype=np.int32), *want, tag="cfg3")


def test_short_channel_shape_at_its_longest_and_widest(ctx):
    """The two-workgroups-per-CU encode shape (half the division table in LDS) is taken for more than 64 Ki channels of at
    most 124 samples: at exactly 124 samples of 31-bit noise (the most symbols a sample can cost) a channel counts up to
    the last table entry that shape holds; 125 samples take the ordinary shape.  Sampled channels byte for byte vs the oracle."""
    import torch
    Cn = 65536 + 320
    g = torch.Generator(device="cuda").manual_seed(77)
    for T in (124, 125):
        x = torch.randint(0, 2**31, (T, Cn), dtype=torch.int64, device="cuda", generator=g).to(torch.int32)  # (diff.c:15-18: samples are read unsigned and a difference must fit int32)
        cap = 4 * ((T * 12 + 67) // 4)
        out, bits, err = ctx.encode(x, adaptive=1, cap=cap)
        y, derr = ctx.decode(out, bits, T, adaptive=1)
        torch.cuda.synchronize()
        assert int((err != 0).sum()) == 0 and int((derr != 0).sum()) == 0
        assert bool((y == x).all())
        sel = np.concatenate([np.arange(0, Cn, 997), np.arange(Cn - 70, Cn)])
        selt = torch.from_numpy(sel).cuda()
        want = orc.encode_batch_tc(x[:, selt].cpu().numpy(), 1, cap=cap)
        assert_streams_equal(out[selt].cpu().numpy(), bits[selt].cpu().numpy().astype(np.uint64), np.zeros(len(sel), dtype=np.int32), *want, tag="short T=%d" % T)


def test_cfg5_style_streamed_batches(ctx):
    """configs[4] in miniature: a channel population larger than one resident batch is streamed through the device in
    batches (here 4 x 32 Ki channels x 1500 samples, generated per batch from the channel ids), encode + decode per
    batch, and compared with the regenerated input; the concatenation of the batches' bit lengths is checked against
    one big batch (channels are independent, so the split must not matter)."""
    import torch
    per, nb, T = 32768, 4, 1500
    cap = 4 * ((T * 3 + 67) // 4)
    all_bits = []
    for b in range(nb):
        x = ctx.synth(per, T, seed=1234, c0=b * per, S=50)
        out, bits, err = ctx.encode(x, adaptive=1, cap=cap)
        y, derr = ctx.decode(out, bits, T, adaptive=1)
        torch.cuda.synchronize()
        assert int((err != 0).sum()) == 0 and int((derr != 0).sum()) == 0 and bool((y == x).all())
        all_bits.append(bits.clone())
    xb = ctx.synth(per * nb, T, seed=1234, c0=0, S=50)
    out, bits, err = ctx.encode(xb, adaptive=1, cap=cap)
    torch.cuda.synchronize()
    assert bool((torch.cat(all_bits) == bits).all())


def test_static_model_and_float_entry_batch(ctx):
    """`bac` without `adaptive` (the reference's default) and the float entry, on a batch, vs the oracle."""
    rng = np.random.default_rng(12)
    T, Cn = 400, 200
    v = (np.abs(np.cumsum(rng.normal(0, 0.4, (T, Cn)), axis=0)) + 1.0).astype(np.float32).round(2)
    for ad in (0, 1):
        out, bits, err = ctx.encode_f32_host(v, factor=100.0, adaptive=ad)
        assert (err == 0).all()
        for c in (0, 63, 64, 199):
            ret, b, n = orc.encode_f32(np.ascontiguousarray(v[:, c]), 100.0, ad)
            assert ret == 0 and n == int(bits[c]) and out[c, : len(b)].tobytes() == b
        back, derr = ctx.decode_f32_host(out, bits, T, factor=100.0, adaptive=ad)
        ints = np.frombuffer(orc.stage("normalize", True, np.ascontiguousarray(v[:, 0]).tobytes(), T * 32)[1], dtype=">i4")
        assert (derr == 0).all() and back[:, 0].tobytes() == (ints.astype(np.float32) / np.float32(100.0)).astype(np.float32).tobytes()


def test_row_pitch_and_alignment_variants(ctx):
    """The encoder fetches rows four at a time when a wave's 64 channels all exist, the pitch is a multiple of 4 and the
    base is 16-byte aligned, else one dword per lane; the decoder writes rows with the caller's pitch.  Both paths, with
    ld > C, odd pitches, a misaligned base and a ragged last wave, against the oracle; the columns beyond C stay untouched."""
    import torch
    rng = np.random.default_rng(17)
    for Cn, ld, off in ((128, 128, 0), (128, 131, 0), (128, 132, 1), (100, 160, 0), (64, 64, 2), (320, 323, 3)):
        T = 160
        x = (np.cumsum(rng.integers(-60, 61, (T, Cn)), axis=0) + 30000).astype(np.int32)
        flat = torch.full((T * ld + 8,), -7, dtype=torch.int32, device="cuda")
        view = flat[off: off + T * ld].view(T, ld)  # base misaligned by `off` dwords
        view[:, :Cn] = torch.from_numpy(x).cuda()
        cap = 4 * ((orc.lib().orc_dega_worst_case_bytes(T) + 3) // 4)
        out = torch.zeros((Cn, cap), dtype=torch.uint8, device="cuda")
        bits = torch.zeros(Cn, dtype=torch.int64, device="cuda")
        err = torch.zeros(Cn, dtype=torch.int32, device="cuda")
        lib = ctx_lib()
        ret = lib.dega_hip_encode_dev(ctx._h, view.data_ptr(), Cn, T, ld, 1, 32, out.data_ptr(), cap, bits.data_ptr(), err.data_ptr(), None)
        assert ret == 0
        torch.cuda.synchronize()
        want_out, want_bits, want_err = orc.encode_batch_tc(x, 1, cap=cap)
        assert_streams_equal(out.cpu().numpy(), bits.cpu().numpy().astype(np.uint64), err.cpu().numpy(), want_out, want_bits, want_err, (Cn, ld, off))
        yflat = torch.full((T * ld + 8,), -9, dtype=torch.int32, device="cuda")
        yview = yflat[off: off + T * ld].view(T, ld)
        derr = torch.zeros(Cn, dtype=torch.int32, device="cuda")
        ret = lib.dega_hip_decode_dev(ctx._h, out.data_ptr(), cap, bits.data_ptr(), Cn, T, ld, 1, 32, yview.data_ptr(), derr.data_ptr(), None)
        assert ret == 0
        torch.cuda.synchronize()
        got = yview.cpu().numpy()
        assert (derr.cpu().numpy() == 0).all() and (got[:, :Cn] == x).all(), (Cn, ld, off)
        assert (got[:, Cn:] == -9).all(), "the decoder wrote outside its C columns"


def ctx_lib():
    return load_package().library()


def test_packed_host_entry_equals_slabs(ctx, dca):
    """dega_hip_encode_packed_host: the streams of encode_host, concatenated (ceil(bits/8) bytes each, channel order)."""
    rng = np.random.default_rng(23)
    T, Cn = 700, 300
    x = (np.cumsum(rng.integers(-80, 81, (T, Cn)), axis=0) + 25000).astype(np.int32)
    x[3, 7] = -1  # a channel in error (a negative sample is read as 2^32 - 1: the difference does not fit, diff.c:15-18)
    out, bits, err = ctx.encode_host(x, adaptive=1)
    packed, offsets, pbits, perr = ctx.encode_packed_host(x, adaptive=1)
    assert (pbits == bits).all() and (perr == err).all() and err[7] != 0
    assert int(offsets[0]) == 0 and int(offsets[Cn]) == len(packed) == int(((bits + 7) // 8).sum())
    for c in range(Cn):
        nb = (int(bits[c]) + 7) // 8
        assert int(offsets[c + 1] - offsets[c]) == nb and packed[int(offsets[c]): int(offsets[c]) + nb].tobytes() == out[c, :nb].tobytes(), c
    # and back through the packed decode entry (fixed count, then variable)
    good = err == 0
    y, derr = ctx.decode_packed_host(packed, offsets, np.where(good, bits, 0).astype(np.uint64), T, adaptive=1)
    assert (derr[good] == 0).all() and (y[:, good] == x[:, good]).all()
    y2, counts, derr2 = ctx.decode_packed_host(packed, offsets, np.where(good, bits, 0).astype(np.uint64), T + 5, adaptive=1, var=True)
    assert (derr2[good] == 0).all() and (counts[good] == T).all() and (y2[:T, good] == x[:, good]).all()
    # a buffer that is too small: the size needed is reported
    small, off2, _, _ = None, np.zeros(Cn + 1, dtype=np.uint64), None, None
    b2 = np.zeros(Cn, dtype=np.uint64)
    e2 = np.zeros(Cn, dtype=np.int32)
    buf = np.empty(16, dtype=np.uint8)
    ret = dca.library().dega_hip_encode_packed_host(ctx._h, x.ctypes.data, Cn, T, Cn, 1, 32, buf.ctypes.data, 16, off2.ctypes.data, b2.ctypes.data, e2.ctypes.data)
    assert ret == dca.ERROR_MEMORY and int(off2[Cn]) == len(packed) and (b2 == bits).all()
