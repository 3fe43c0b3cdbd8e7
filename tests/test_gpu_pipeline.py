"""GPU tests (-m gpu) of the host-pointer path a reference user actually takes (DCCLI/src/cli.c:447 -> plugin -> C ABI):
the chunked multi-stream pipeline, the packed sample types (int32 / big-endian / int64 / float32 with Normalize fused into
the kernels), the multi-device group, the headline channel length T = 86 400 against the oracle, and the REAL DCCLI
(built from the reference's sources by oracle/Makefile with the three table rows of INTEGRATION.md) driving the plugin."""
import gzip
import hashlib
import json
import os
import subprocess

import numpy as np
import pytest

from __graft_entry__ import load_package
from oracle import orc

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
REF_CLI_GPU = os.path.join(ROOT, "oracle", "_ref", "DCCLI_gpu")


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


def walk(rng, T, Cn, S=50, base=20000):
    x = np.cumsum(rng.integers(-S, S + 1, (T, Cn)), axis=0) + rng.integers(base, 3 * base, Cn)[None, :]
    return np.clip(x, 0, 2**31 - 1).astype(np.int32)


def stream_of(packed, offsets, c):
    return packed[int(offsets[c]): int(offsets[c + 1])].tobytes()


def test_packed_job_all_sample_types_vs_oracle(dca, ctx):
    """dega_hip_encode_job_host / _decode_job_host: every sample type gives the oracle's streams, packed in channel order;
    a row pitch larger than the channel count; decode returns the samples in the same type."""
    rng = np.random.default_rng(11)
    T, Cn = 700, 37
    x = walk(rng, T, Cn)
    x[:, 5] = rng.integers(0, 1 << 30, T)  # a noisy channel: its stream is longer than its samples
    x[3:, 9] = -5  # diff.c:17-18: ERROR_INVALID_VALUE
    want = [orc.encode_i32(np.ascontiguousarray(x[:, c]), 1) for c in range(Cn)]
    for samples, arr in ((dca.SAMPLES_I32, x), (dca.SAMPLES_BE32, x.astype(">i4"))):
        wide = np.zeros((T, Cn + 6), dtype=arr.dtype)
        wide[:, :Cn] = arr
        for src, ch in ((arr, None), (wide, Cn)):
            packed, offsets, bits, err = ctx.encode_job(src, adaptive=1, samples=samples, channels=ch)
            for c in range(Cn):
                ret, b, n = want[c]
                assert int(err[c]) == ret, (samples, c)
                if ret == 0:
                    assert int(bits[c]) == n and stream_of(packed, offsets, c) == b, (samples, c)
            assert int(offsets[0]) == 0 and int(offsets[Cn]) == packed.size
            ok = err == 0
            back, derr = ctx.decode_job(packed, offsets, bits, T, adaptive=1, samples=samples)
            assert (derr[ok] == 0).all() and (back[:, ok] == arr[:, ok]).all()
            back, counts, derr = ctx.decode_job(packed, offsets, bits, T + 9, adaptive=1, samples=samples, var=True)
            assert (counts[ok] == T).all() and (back[:T, ok] == arr[:, ok]).all()
    # static model, narrow value size through the same entry
    x12 = (x & 0xFFF).astype(np.int32)
    packed, offsets, bits, err = ctx.encode_job(x12, adaptive=0, valuesize=12)
    w_out, w_bits, w_err = ctx.encode_host(x12, adaptive=0, valuesize=12)
    assert (err == w_err).all() and (bits[w_err == 0] == w_bits[w_err == 0]).all()
    for c in np.nonzero(w_err == 0)[0]:
        assert stream_of(packed, offsets, c) == w_out[c, : (int(w_bits[c]) + 7) // 8].tobytes()


def test_fused_float_entry_against_the_reference(dca, ctx):
    """Normalize runs inside the encode kernel, Denormalize inside the decode kernel -- one launch per
    direction, valuesize 8..64.  Streams, error verdicts and the floats coming back are the compiled reference's
    (tests/golden/floats_vs.npz), through the host entry and (up to 32 bits: torch has the slab API) the device entry."""
    import torch
    z = np.load(os.path.join(GOLDEN, "floats_vs.npz"))
    for tag in sorted({k.split(".")[0] for k in z.files}, key=lambda t: int(t[2:])):
        vs, factor = int(tag[2:]), float(z[tag + ".factor"][0])
        v = np.ascontiguousarray(z[tag + ".v"])
        T, Cn = v.shape
        w_err, w_bits, w_stream, w_back = z[tag + ".err"], z[tag + ".bits"], z[tag + ".stream"], z[tag + ".back"]
        packed, offsets, bits, err = ctx.encode_job(v, adaptive=1, valuesize=vs, samples=dca.SAMPLES_F32, factor=factor)
        assert (err == w_err).all(), (tag, err, w_err)
        ok = w_err == 0
        assert (bits[ok] == w_bits[ok]).all(), tag
        for c in np.nonzero(ok)[0]:
            assert stream_of(packed, offsets, c) == w_stream[c, : (int(w_bits[c]) + 7) // 8].tobytes(), (tag, int(c))
        back, derr = ctx.decode_job(packed, offsets, bits, T, adaptive=1, valuesize=vs, samples=dca.SAMPLES_F32, factor=factor)
        assert (derr[ok] == 0).all() and back[:, ok].tobytes() == np.ascontiguousarray(w_back[:, ok]).tobytes(), tag
        # the device-pointer forms: one encode launch, one decode launch
        out, dbits, derr2 = ctx.encode_f32(torch.from_numpy(v).cuda(), factor=factor, adaptive=1, valuesize=vs)
        torch.cuda.synchronize()
        assert (derr2.cpu().numpy() == w_err).all() and (dbits.cpu().numpy().astype(np.uint64)[ok] == w_bits[ok]).all(), tag
        oh = out.cpu().numpy()
        for c in np.nonzero(ok)[0]:
            nb = (int(w_bits[c]) + 7) // 8
            assert oh[c, :nb].tobytes() == w_stream[c, :nb].tobytes(), (tag, int(c))
        vb, verr = ctx.decode_f32(out, dbits, T, factor=factor, adaptive=1, valuesize=vs)
        torch.cuda.synchronize()
        assert vb.cpu().numpy()[:, ok].tobytes() == np.ascontiguousarray(w_back[:, ok]).tobytes(), tag


def test_float_entry_is_one_launch_without_intermediate(dca, ctx):
    """The fused float entry equals normalize-kernel + encode-kernel on the same data (the two-launch form it replaces)."""
    import torch
    rng = np.random.default_rng(4)
    v = np.round(np.abs(np.cumsum(rng.normal(0, 0.4, (900, 130)), axis=0) + 40.0), 2).astype(np.float32)
    vt = torch.from_numpy(v).cuda()
    xi, nerr = ctx.normalize(vt, 100.0)
    out2, bits2, err2 = ctx.encode(xi, adaptive=1, cap=4 * 900 + 64)
    out1, bits1, err1 = ctx.encode_f32(vt, factor=100.0, adaptive=1, cap=4 * 900 + 64)
    torch.cuda.synchronize()
    assert (err1 == err2).all().item() and (bits1 == bits2).all().item() and (out1 == out2).all().item()


def test_pipeline_chunks_and_streams_equal_device_resident(dca, ctx):
    """A batch large enough to be cut into several chunks (320 MB of samples, 5 streams in flight): the packed host result
    equals the device-resident kernel's streams, and decoding it through the host pipeline returns the samples."""
    import torch
    Cn, T = 4608, 16384  # not a multiple of the chunk size: a ragged last chunk
    x = ctx.synth(Cn, T, seed=99, S=50)
    cap = 4 * T + 64
    out, bits, err = ctx.encode(x, adaptive=1, cap=cap)
    dpacked, doff = ctx.compact(out, bits)
    torch.cuda.synchronize()
    xh = x.cpu().numpy()
    packed, offsets, hbits, herr = ctx.encode_job(xh, adaptive=1)
    assert (herr == 0).all() and (hbits.astype(np.int64) == bits.cpu().numpy()).all()
    assert (offsets.astype(np.int64) == doff.cpu().numpy()).all()
    assert packed.tobytes() == dpacked.cpu().numpy().tobytes()
    back, derr = ctx.decode_job(packed, offsets, hbits, T, adaptive=1)
    assert (derr == 0).all() and (back == xh).all()
    # too little room: the call says how much it needs and still reports bits and err
    small = np.empty(1000, dtype=np.uint8)
    L = dca.library()
    import ctypes as C
    job = dca.Job(Cn, T, Cn, 1, 32, dca.SAMPLES_I32, 0.0)
    o2, b2, e2 = np.zeros(Cn + 1, dtype=np.uint64), np.zeros(Cn, dtype=np.uint64), np.zeros(Cn, dtype=np.int32)
    ret = L.dega_hip_encode_job_host(ctx._h, C.byref(job), xh.ctypes.data, small.ctypes.data, small.size, o2.ctypes.data, b2.ctypes.data, e2.ctypes.data)
    assert ret == dca.ERROR_MEMORY and int(o2[Cn]) == packed.size and (b2 == hbits).all() and (e2 == 0).all()


def test_order_of_the_enqueues_does_not_change_the_result(dca, ctx, monkeypatch):
    """How far the first stages of an encode call run ahead of the second ones (DEGA_PIPELINE_AHEAD), whether a decode
    call's uploads go first on one stream (DEGA_PIPELINE_UPLOADS_FIRST) and how many chunks there are only change WHEN the
    copies and kernels run: same packed streams, same samples back.  Pinned and pageable memory, a ragged last chunk."""
    import torch
    Cn, T = 2600, 4200
    x = ctx.synth(Cn, T, seed=5, S=80)
    out, bits, err = ctx.encode(x, adaptive=1, cap=4 * T + 64)
    dpacked, doff = ctx.compact(out, bits)
    torch.cuda.synchronize()
    want = dpacked.cpu().numpy().tobytes()
    xh = x.cpu().numpy()
    pin = dca.PinnedArray((T, Cn), np.int32)
    pin.array[:] = xh
    for chunks, ahead, first in (("5", "1", "1"), ("5", "2", "0"), ("5", "16", "1"), ("3", "2", "1"), ("1", "2", "1")):
        monkeypatch.setenv("DEGA_PIPELINE_CHUNKS", chunks)
        monkeypatch.setenv("DEGA_PIPELINE_AHEAD", ahead)
        monkeypatch.setenv("DEGA_PIPELINE_AHEAD_DECODE", ahead)
        monkeypatch.setenv("DEGA_PIPELINE_UPLOADS_FIRST", first)
        for src in (xh, pin.array):
            packed, offsets, hbits, herr = ctx.encode_job(src, adaptive=1)
            assert (herr == 0).all() and (offsets.astype(np.int64) == doff.cpu().numpy()).all() and packed.tobytes() == want, (chunks, ahead, first)
        pk = dca.PinnedArray((packed.size,), np.uint8)
        pk.array[:] = packed
        back, derr = ctx.decode_job(packed, offsets, hbits, T, adaptive=1)
        assert (derr == 0).all() and (back == xh).all(), (chunks, ahead, first)
        pin2 = dca.PinnedArray((T, Cn), np.int32)
        back2, derr2 = ctx.decode_job(pk.array, offsets, hbits, T, adaptive=1, out=pin2.array)
        assert (derr2 == 0).all() and (pin2.array == xh).all(), (chunks, ahead, first)


def test_rows_uploaded_in_bands_beside_the_running_kernel(dca, ctx, monkeypatch):
    """Few, long channels: the encode kernel starts before the samples are there and takes the rows as the bands of the
    upload arrive (EncodeArgs::rows_ready); the decode kernel's rows go home in bands while it is still running
    (DecodeArgs::rows_done).  Forced onto small batches with small bands -- many band ends, a ragged last wave, pageable and
    pinned memory, a float batch -- and compared with the device-resident kernel's streams / the input."""
    import torch
    monkeypatch.setenv("DEGA_PIPELINE_BAND_BYTES", "65536")
    for Cn, T in ((300, 5000), (64, 9001), (1030, 700)):
        x = ctx.synth(Cn, T, seed=7 + Cn, S=120)
        out, bits, err = ctx.encode(x, adaptive=1, cap=4 * T + 64)
        dpacked, doff = ctx.compact(out, bits)
        torch.cuda.synchronize()
        xh = x.cpu().numpy()
        packed, offsets, hbits, herr = ctx.encode_job(xh, adaptive=1)
        assert (herr == 0).all() and (hbits.astype(np.int64) == bits.cpu().numpy()).all(), (Cn, T)
        assert packed.tobytes() == dpacked.cpu().numpy().tobytes(), (Cn, T)
        pin = dca.PinnedArray((T, Cn), np.int32)
        pin.array[:] = xh
        p2, o2, b2, e2 = ctx.encode_job(pin.array, adaptive=1)
        assert (e2 == 0).all() and p2.tobytes() == packed.tobytes(), (Cn, T)
        # and back: the rows come home in bands while the decode kernel is still running (DecodeArgs::rows_done)
        back, derr = ctx.decode_job(packed, offsets, hbits, T, adaptive=1)
        assert (derr == 0).all() and (back == xh).all(), (Cn, T)
        pin.array[:] = 0
        back2, derr2 = ctx.decode_job(packed, offsets, hbits, T, adaptive=1, out=pin.array)
        assert (derr2 == 0).all() and (pin.array == xh).all(), (Cn, T)
    v = (np.cumsum(np.random.default_rng(3).normal(0, 0.4, (6000, 130)), axis=0) + 230.0).astype(np.float32)
    pf, of, bf, ef = ctx.encode_job(v, adaptive=1, samples=dca.SAMPLES_F32, factor=100.0)
    monkeypatch.setenv("DEGA_PIPELINE_BAND_BYTES", "0")
    pg, og, bg, eg = ctx.encode_job(v, adaptive=1, samples=dca.SAMPLES_F32, factor=100.0)
    assert (ef == 0).all() and (bf == bg).all() and pf.tobytes() == pg.tobytes()


def test_pinned_samples_read_in_place_by_the_kernel(dca, ctx, monkeypatch):
    """A chunk that is all columns of the caller's pinned array is not uploaded: the encode kernel's filling waves fetch
    the rows from host memory themselves.  Same streams as through the copy engine (DEGA_PIPELINE_IN_PLACE=0) and as
    the oracle's; a channel whose stream outgrows its samples takes the worst-case pass from the same rows; float rows."""
    rng = np.random.default_rng(77)
    for Cn, T in ((257, 4500), (64, 131), (1000, 33)):
        x = walk(rng, T, Cn)
        if T > 1000:
            x[:, 5] = rng.integers(0, 1 << 30, T)  # noise: its stream is longer than its samples
        pin = dca.PinnedArray((T, Cn), np.int32)
        pin.array[:] = x
        monkeypatch.setenv("DEGA_PIPELINE_IN_PLACE", "1")
        p1, o1, b1, e1 = ctx.encode_job(pin.array, adaptive=1)
        monkeypatch.setenv("DEGA_PIPELINE_IN_PLACE", "0")
        p0, o0, b0, e0 = ctx.encode_job(pin.array, adaptive=1)
        assert (e1 == 0).all() and (e0 == 0).all() and (b1 == b0).all() and (o1 == o0).all() and p1.tobytes() == p0.tobytes(), (Cn, T)
        for c in (0, 5, Cn - 1):
            ret, b, n = orc.encode_i32(np.ascontiguousarray(x[:, c]), 1)
            assert ret == 0 and int(b1[c]) == n and stream_of(p1, o1, c) == b, (Cn, T, c)
    v = (np.cumsum(rng.normal(0, 0.4, (5000, 130)), axis=0) + 230.0).astype(np.float32)
    pf = dca.PinnedArray(v.shape, np.float32)
    pf.array[:] = v
    monkeypatch.setenv("DEGA_PIPELINE_IN_PLACE", "1")
    a = ctx.encode_job(pf.array, adaptive=1, samples=dca.SAMPLES_F32, factor=100.0)
    monkeypatch.setenv("DEGA_PIPELINE_IN_PLACE", "0")
    b = ctx.encode_job(v, adaptive=1, samples=dca.SAMPLES_F32, factor=100.0)
    assert (a[3] == 0).all() and (a[2] == b[2]).all() and a[0].tobytes() == b[0].tobytes()


def test_streams_longer_than_their_samples_take_the_worst_case_pass(dca, ctx):
    """Noise: ~60 coded bits per 32-bit sample.  The pipeline's first attempt sizes slabs for streams no longer than their
    samples; chunks that do not fit are redone with worst-case slabs -- same streams as the oracle's, mixed with channels
    that did fit, order kept."""
    rng = np.random.default_rng(21)
    T, Cn = 3000, 1300
    x = walk(rng, T, Cn)
    noisy = [0, 7, 640, 1299]
    for c in noisy:
        x[:, c] = rng.integers(0, 1 << 30, T)
    packed, offsets, bits, err = ctx.encode_job(x, adaptive=1)
    assert (err == 0).all()
    for c in noisy + [1, 8, 641, 1298]:
        ret, b, n = orc.encode_i32(np.ascontiguousarray(x[:, c]), 1)
        assert ret == 0 and int(bits[c]) == n and stream_of(packed, offsets, c) == b, c
    assert all(int(bits[c]) > 32 * T for c in noisy)
    back, derr = ctx.decode_job(packed, offsets, bits, T, adaptive=1)
    assert (derr == 0).all() and (back == x).all()
    # slab form of the same call: a slab too small for its stream is that channel's ERROR_MEMORY
    out, sbits, serr = ctx.encode_host(x[:, :16], adaptive=1, cap=4 * T + 64)
    assert serr[0] == dca.ERROR_MEMORY and serr[7] == dca.ERROR_MEMORY and (np.delete(serr, [0, 7]) == 0).all()


def test_group_splits_channels_and_concatenates_on_the_host(dca, ctx):
    """dega_hip_group_*: with one visible GPU the group IS the single-context call; a group with two members (here: two
    contexts, as on a two-GPU node -- on a one-GPU box both sit on device 0) splits the channels into contiguous ranges,
    codes them on separate host threads and concatenates the packed streams on the host.  Same bytes either way."""
    import torch
    rng = np.random.default_rng(8)
    T, Cn = 2000, 2600
    x = walk(rng, T, Cn)
    x[:, 1400] = rng.integers(0, 1 << 30, T)  # one channel needs the worst-case pass, in the second member's range
    x[5:, 33] = -1
    ref = ctx.encode_job(x, adaptive=1)
    g_all = dca.Group()
    assert g_all.size() == torch.cuda.device_count()
    n_dev = torch.cuda.device_count()
    members = [[0], [0, 0], [0, 0, 0]] + ([list(range(n_dev))] if n_dev > 1 else [])
    for devs in members:
        g = dca.Group(devs)
        assert g.size() == len(devs)
        got = g.encode_job(x, adaptive=1)
        for a, b in zip(ref, got):
            assert a.tobytes() == b.tobytes(), devs
        back, counts, derr = g.decode_job(got[0], got[1], got[2], T, adaptive=1, var=True)
        ok = ref[3] == 0
        assert (derr[ok] == 0).all() and (counts[ok] == T).all() and (back[:, ok] == x[:, ok]).all(), devs
        # float samples through the group as well
        v = (x[:, :1100].astype(np.float32) / 100.0)
        pf = g.encode_job(v, adaptive=1, samples=dca.SAMPLES_F32, factor=100.0)
        cf = ctx.encode_job(v, adaptive=1, samples=dca.SAMPLES_F32, factor=100.0)
        for a, b in zip(cf, pf):
            assert a.tobytes() == b.tobytes(), devs
        g.close()
    g_all.close()


def test_pinned_host_memory_round_trip(dca, ctx):
    rng = np.random.default_rng(2)
    T, Cn = 512, 700
    pin = dca.PinnedArray((T, Cn), np.int32)
    pin.array[:] = walk(rng, T, Cn)
    packed, offsets, bits, err = ctx.encode_job(pin.array, adaptive=1)
    out = dca.PinnedArray((T, Cn), np.int32)
    back, derr = ctx.decode_job(packed, offsets, bits, T, adaptive=1, out=out.array)
    assert (err == 0).all() and (derr == 0).all() and (back == pin.array).all()
    pin.free()
    out.free()


def test_headline_channel_length_vs_oracle(dca, ctx):
    """BASELINE configs[1] at its real channel length: T = 86 400 samples (one day at 1 s; ~110 model halvings per channel,
    bac.c:57-67), 256 channels of the bench's synthetic workload, adaptive model against the oracle byte for byte; 64 of
    them with the static model too; and decoded back."""
    import torch
    T, Cn = 86400, 256
    x = ctx.synth(Cn, T, seed=1234, S=50)
    xh = x.cpu().numpy()
    assert (xh[:100, :4] == dca.synth_reference(4, 100, seed=1234, S=50)).all()  # the workload definition
    packed, offsets, bits, err = ctx.encode_job(xh, adaptive=1)
    assert (err == 0).all()
    for c in range(Cn):
        ret, b, n = orc.encode_i32(np.ascontiguousarray(xh[:, c]), 1)
        assert ret == 0 and int(bits[c]) == n and stream_of(packed, offsets, c) == b, c
    back, derr = ctx.decode_job(packed, offsets, bits, T, adaptive=1)
    assert (derr == 0).all() and (back == xh).all()
    ps, os_, bs, es = ctx.encode_job(xh[:, :64], adaptive=0, channels=64)
    for c in range(64):
        ret, b, n = orc.encode_i32(np.ascontiguousarray(xh[:, c]), 0)
        assert ret == 0 and int(bs[c]) == n and stream_of(ps, os_, c) == b, c
    back, derr = ctx.decode_job(ps, os_, bs, T, adaptive=0)
    assert (derr == 0).all() and (back == xh[:, :64]).all()


# ---- the real caller ---------------------------------------------------------------------------------------------------

def ref_cli(args, env=None):
    return subprocess.run([REF_CLI_GPU] + args, capture_output=True, text=True, env=env)


@pytest.fixture(scope="module")
def input_txt(tmp_path_factory):
    p = tmp_path_factory.mktemp("ref") / "input.txt"
    with gzip.open(os.path.join(GOLDEN, "input.txt.gz"), "rb") as f:
        p.write_bytes(f.read())
    return p


def test_reference_dccli_runs_its_make_test_chain_through_the_plugin(input_txt, tmp_path):
    """The reference's own DCCLI (sources from /root/reference, compiled in the build container by oracle/Makefile; only
    DCLib/src/enc_dec.c gets the three table rows of INTEGRATION.md section 2) calls this project's plugin at its one
    call site (DCCLI/src/cli.c:447): the `make test` chain (DCCLI/build/gcc/Makefile:75-77) with the eight middle
    stages replaced by `encode fdega adaptive # decode fdega adaptive` returns the input text; the compressed file is
    byte-identical to the canonical file of the all-CPU chain."""
    if not os.path.exists(REF_CLI_GPU):
        pytest.skip("oracle/_ref/DCCLI_gpu is built in the container that holds /root/reference (make -C oracle)")
    out = tmp_path / "output.txt"
    p = ref_cli([str(input_txt), str(out), "decode", "csv", "#", "encode", "fdega", "adaptive", "#", "decode", "fdega", "adaptive", "#", "encode", "csv"])
    assert p.returncode == 0, p.stdout + p.stderr
    assert out.read_bytes() == input_txt.read_bytes()  # the reference's own test: diff input output
    enc = tmp_path / "out.dega"
    p = ref_cli([str(input_txt), str(enc), "decode", "csv", "#", "encode", "fdega", "adaptive"])
    assert p.returncode == 0, p.stdout + p.stderr
    with open(os.path.join(GOLDEN, "dega_adaptive.bin"), "rb") as f:
        assert enc.read_bytes() == f.read()
    # mixed chains: the reference's CPU stages on one side, the GPU codec on the other
    p = ref_cli([str(enc), str(out), "decode", "bac", "adaptive", "#", "decode", "seg", "#", "decode", "diff", "#", "decode", "normalize", "#", "encode", "csv"])
    assert p.returncode == 0 and out.read_bytes() == input_txt.read_bytes()
    p = ref_cli([str(input_txt), str(enc), "decode", "csv", "#", "encode", "normalize", "#", "encode", "dega", "adaptive", "#", "decode", "dega", "adaptive",
                 "#", "decode", "normalize", "#", "encode", "csv"])
    assert p.returncode == 0 and enc.read_bytes() == input_txt.read_bytes()
    # the second codec: `encode glzmh` writes the file `encode lzmh` writes
    with open(os.path.join(GOLDEN, "lzmh.json")) as f:
        meta = json.load(f)["testfile"]
    p = ref_cli([str(input_txt), str(enc), "encode", "glzmh"])
    assert p.returncode == 0 and hashlib.sha256(enc.read_bytes()).hexdigest() == meta["sha256"]
    p = ref_cli([str(enc), str(out), "decode", "lzmh"])  # decoded by the reference's CPU decoder
    assert p.returncode == 0 and out.read_bytes() == input_txt.read_bytes()
    # an option the row does not list is refused by the reference's own parser
    assert ref_cli([str(input_txt), str(enc), "decode", "csv", "#", "encode", "dega", "normalization_factor=3"]).returncode != 0


@pytest.mark.gpu
def test_reference_dccli_codes_a_batch_of_channels(tmp_path):
    """The batch dimension through the reference's OWN DCCLI: built from the reference's sources with the num_channels
    edit of INTEGRATION.md 2b (`size_t num_channels` appended to options_t, OPTION_NUM_CHANNELS, its sorted
    option_descriptions[] row and default -- applied to scratch copies by oracle/Makefile), its option parser accepts
    `num_channels=4096`, its stage loop (DCCLI/src/cli.c:447) hands the plugin 4 096 interleaved channels, and every
    inner stream of the container it writes is the oracle's for that channel; `decode dega ... num_channels=4096` returns
    the input.  The same for `glzmh` (pieces of a text), decoded piece by piece by the oracle."""
    if not os.path.exists(REF_CLI_GPU):
        pytest.skip("oracle/_ref/DCCLI_gpu is built in the container that holds /root/reference (make -C oracle)")
    rng = np.random.default_rng(17)
    T, Cn = 200, 4096
    x = (np.cumsum(rng.integers(-60, 61, (T, Cn)), axis=0) + 30000).astype(">i4")
    src, enc, dec = tmp_path / "batch.be32", tmp_path / "batch.degb", tmp_path / "batch.out"
    src.write_bytes(x.tobytes())  # sample-major interleaving = [T][C]
    p = ref_cli([str(src), str(enc), "encode", "dega", "adaptive", "num_channels=%d" % Cn])
    assert p.returncode == 0, p.stdout + p.stderr
    blob = enc.read_bytes()
    assert blob[:4] == b"DEGB" and int.from_bytes(blob[8:16], "big") == Cn and int.from_bytes(blob[16:24], "big") == T
    lens = [int.from_bytes(blob[24 + 8 * c: 32 + 8 * c], "big") for c in range(Cn)]
    off = 24 + 8 * Cn
    want_out, want_bits, want_err = orc.encode_batch_tc(x.astype(np.int32), 1, cap=(orc.lib().orc_dega_worst_case_bytes(T) + 3) & ~3)
    assert (want_err == 0).all()
    for c in range(Cn):
        nb = (lens[c] + 7) // 8
        assert lens[c] == int(want_bits[c]) and blob[off: off + nb] == want_out[c, :nb].tobytes(), c
        off += nb
    assert off == len(blob)
    p = ref_cli([str(enc), str(dec), "decode", "dega", "adaptive", "num_channels=%d" % Cn])
    assert p.returncode == 0, p.stdout + p.stderr
    assert dec.read_bytes() == x.tobytes()
    # chained inside one invocation: the exact bit length goes from stage to stage in memory
    p = ref_cli([str(src), str(dec), "encode", "dega", "adaptive", "num_channels=%d" % Cn, "#", "decode", "dega", "adaptive", "num_channels=%d" % Cn])
    assert p.returncode == 0 and dec.read_bytes() == x.tobytes()
    # the float entry: csv text of 8 interleaved channels -> fdega -> back (the reference's csv codec on both ends)
    vals = (np.cumsum(rng.integers(-40, 41, (300, 8)), axis=0) + 20000) / 100.0
    txt = tmp_path / "in.txt"
    txt.write_text("".join("%.2f\n" % v for v in vals.reshape(-1)))
    p = ref_cli([str(txt), str(dec), "decode", "csv", "#", "encode", "fdega", "adaptive", "num_channels=8", "#", "decode", "fdega", "adaptive", "num_channels=8",
                 "#", "encode", "csv"])
    assert p.returncode == 0 and dec.read_bytes() == txt.read_bytes()
    # glzmh: the text cut into 64 pieces, each the stream `encode lzmh` writes for that piece
    text = "".join("%d.%02d\n" % (v // 100, v % 100) for v in x[:, :40].astype(np.int64).reshape(-1).tolist()).encode()
    tsrc, tenc = tmp_path / "t.txt", tmp_path / "t.lzmb"
    tsrc.write_bytes(text)
    p = ref_cli([str(tsrc), str(tenc), "encode", "glzmh", "num_channels=64"])
    assert p.returncode == 0, p.stdout + p.stderr
    blob = tenc.read_bytes()
    assert blob[:4] == b"LZMB" and int.from_bytes(blob[8:16], "big") == 64
    piece = (len(text) + 63) // 64
    off = 16 + 16 * 64
    for c in range(64):
        nbytes, nbits = int.from_bytes(blob[16 + 16 * c: 24 + 16 * c], "big"), int.from_bytes(blob[24 + 16 * c: 32 + 16 * c], "big")
        part = text[c * piece: (c + 1) * piece]
        assert nbytes == len(part)
        r, b, n = orc.stage("lzmh", True, part, 8 * len(part))
        assert r == 0 and n == nbits and blob[off: off + (n + 7) // 8] == b[: (n + 7) // 8], c
        off += (nbits + 7) // 8
    p = ref_cli([str(tenc), str(dec), "decode", "glzmh", "num_channels=64"])
    assert p.returncode == 0 and dec.read_bytes() == text
    # the reference's parser knows the option now -- and still refuses it on a row that does not list it
    assert ref_cli([str(src), str(enc), "encode", "diff", "num_channels=4"]).returncode != 0


@pytest.mark.gpu
def test_channels_coded_over_several_launches(ctx):
    """dega_hip_encode_segment_dev: the rows of a batch in ranges, one launch per range, the lanes' state saved in device
    memory in between -- what the host pipeline does with the bands of a few-long-channels batch.  The streams are those
    of one launch over all rows (= the oracle's): cuts inside a seg-bit word, single-row ranges, an empty last range,
    channels out of range from the start and from the middle, both models, a narrow value size, a ragged wave."""
    import torch
    rng = np.random.default_rng(77)
    T, Cn = 5000, 700
    x = (np.cumsum(rng.integers(-300, 301, (T, Cn)), axis=0) + 1000000).astype(np.int32)
    x[2000:, 5] = -7
    x[0, 6] = -1
    xd = torch.from_numpy(x).cuda()
    for ad, vs in ((1, 32), (0, 32), (1, 24)):
        xs = x if vs == 32 else (x & ((1 << vs) - 1))
        want_out, want_bits, want_err = ctx.encode(torch.from_numpy(xs).cuda(), adaptive=ad, valuesize=vs)
        torch.cuda.synchronize()
        if vs == 32:  # the one-launch streams are the oracle's (pinned elsewhere for the other value sizes)
            o2, b2, e2 = orc.encode_batch_tc(x, ad, cap=want_out.shape[1])
            assert (want_err.cpu().numpy() == e2).all() and (want_bits.cpu().numpy().astype(np.uint64)[e2 == 0] == b2[e2 == 0]).all()
        for cuts in ([0, 1, 2, 2500, 2501, 4999, 5000], [0, 1024, 2048, 3072, 4096, 5000, 5000], [0, 5000]):
            out, bits, err = ctx.encode_segments(torch.from_numpy(xs).cuda() if vs != 32 else xd, cuts, adaptive=ad, cap=want_out.shape[1], valuesize=vs)
            torch.cuda.synchronize()
            assert (err == want_err).all(), (ad, vs, cuts)
            ok = (want_err == 0)
            assert (bits[ok] == want_bits[ok]).all(), (ad, vs, cuts)
            nb = ((want_bits + 7) // 8)
            idx = torch.arange(want_out.shape[1], device=out.device)[None, :] < nb[:, None]
            assert bool((((out == want_out) | ~idx)[ok]).all()), (ad, vs, cuts)
