"""The C host layer (data-compressor_amd/host): plugin/option tables with the reference's lookup semantics
(DCLib/src/enc_dec.c), the bit I/O layer with the reference's stream format (DCIOLib/src/bit_file_buffer.c), and the
DCCLI-style driver.  CPU-only checks here; the GPU-backed codecs are exercised under -m gpu."""
import ctypes as C
import gzip
import json
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "data-compressor_amd", "host")
GOLDEN = os.path.join(ROOT, "tests", "golden")
CLI = os.path.join(HOST, "dccli_amd")


class Options(C.Structure):  # DCLib/inc/enc_dec.h:28-41 + num_channels
    _fields_ = [("error_log_file", C.c_void_p), ("encode", C.c_int), ("encoder_decoder", C.c_void_p),
                ("block_size_bits", C.c_size_t), ("value_size_bits", C.c_size_t), ("adaptive", C.c_int),
                ("column", C.c_size_t), ("separator_char", C.c_char), ("num_decimal_places", C.c_size_t),
                ("normalization_factor", C.c_float), ("num_values", C.c_size_t), ("num_channels", C.c_size_t)]


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__
    if not os.path.exists(os.path.join(ROOT, "data-compressor_amd", "libdega_hip.so")):
        __graft_entry__.build()
    subprocess.run(["make", "-s", "-C", HOST], check=True)
    L = C.CDLL(os.path.join(HOST, "libdclib_amd.so"))
    L.GetEncoder.restype = C.c_void_p
    L.GetEncoder.argtypes = [C.c_char_p]
    L.GetEncoderDescription.restype = C.c_char_p
    L.GetEncoderDescription.argtypes = [C.c_char_p]
    L.GetNumberOfEncoders.restype = C.c_size_t
    L.GetNumberOfOptions.restype = C.c_size_t
    L.OptionNameExists.argtypes = [C.c_char_p]
    L.EncoderSupportsOption.argtypes = [C.c_char_p, C.c_char_p]
    L.EncoderFromFunctionSupportsOption.argtypes = [C.c_void_p, C.c_int, C.c_char_p]
    L.GetOptionType.argtypes = [C.c_char_p]
    for n, t in (("Bool", C.c_int), ("Size", C.c_size_t), ("Float", C.c_float), ("Char", C.c_char)):
        getattr(L, "SetOptionValue" + n).argtypes = [C.POINTER(Options), C.c_char_p, t]
        getattr(L, "GetOptionValue" + n).argtypes = [C.POINTER(Options), C.c_char_p, C.POINTER(t)]
    L.AllocateFileBuffer.restype = C.c_void_p
    L.AllocateBitFileBuffer.restype = C.c_void_p
    L.InitFileBufferInMemory.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
    L.InitBitFileBuffer.argtypes = [C.c_void_p, C.c_void_p]
    L.WriteSingleValueToBitFileBuffer.restype = C.c_int64
    L.WriteSingleValueToBitFileBuffer.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_size_t]
    L.ReadSingleValueFromBitFileBuffer.restype = C.c_int64
    L.ReadSingleValueFromBitFileBuffer.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_size_t]
    L.ReadBitFileBuffer.restype = C.c_int64
    L.ReadBitFileBuffer.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.SetBitFileBufferMode.argtypes = [C.c_void_p, C.c_int]
    L.EndOfBitFileBuffer.argtypes = [C.c_void_p]
    return L


def test_codec_table_lookup_semantics(lib):
    names = (C.c_char_p * lib.GetNumberOfEncoders())()
    lib.GetEncoderNames(names)
    listed = [n.decode() for n in names]
    assert listed == sorted(listed) and {"copy", "dega", "fdega"} <= set(listed)  # sorted: bsearch depends on it
    assert lib.GetEncoder(b"dega") and lib.GetEncoder(b"fdega") and lib.GetEncoder(b"copy")
    assert lib.GetEncoder(b"dega") != lib.GetEncoder(b"fdega")
    # prefix semantics of the comparator (enc_dec.c:89-93): a key that starts with a registered name finds it ...
    assert lib.GetEncoder(b"degaX") == lib.GetEncoder(b"dega")
    # ... a shorter key does not, and unknown names give NULL
    assert not lib.GetEncoder(b"deg") and not lib.GetEncoder(b"bac") and not lib.GetEncoder(b"zzz")
    assert b"GPU" in lib.GetEncoderDescription(b"dega")


def test_option_table_and_defaults(lib):
    names = (C.c_char_p * lib.GetNumberOfOptions())()
    lib.GetOptionNames(names)
    listed = [n.decode() for n in names]
    assert listed == sorted(listed)
    for ref_name in ("adaptive", "blocksize", "column", "normalization_factor", "num_decimal_places", "num_values", "separator_char", "valuesize"):
        assert ref_name in listed  # every option of the reference (enc_dec.c:64-73) keeps its name
    o = Options()
    lib.SetDefaultOptions(C.byref(o))  # enc_dec.c:187-197
    assert (o.adaptive, o.block_size_bits, o.column, o.num_decimal_places, o.value_size_bits, o.num_values) == (0, 8, 1, 2, 32, 2)
    assert o.normalization_factor == 100.0 and o.separator_char == b"," and o.num_channels == 1
    assert lib.SetOptionValueBool(C.byref(o), b"adaptive", 1) == 0 and o.adaptive == 1
    assert lib.SetOptionValueSize(C.byref(o), b"valuesize", 16) == 0 and o.value_size_bits == 16
    assert lib.SetOptionValueFloat(C.byref(o), b"normalization_factor", 10.0) == 0 and o.normalization_factor == 10.0
    assert lib.SetOptionValueSize(C.byref(o), b"nonsense", 1) == -1
    v = C.c_size_t()
    assert lib.GetOptionValueSize(C.byref(o), b"valuesize", C.byref(v)) == 0 and v.value == 16
    assert lib.EncoderSupportsOption(b"dega", b"adaptive") and lib.EncoderSupportsOption(b"fdega", b"normalization_factor")
    assert not lib.EncoderSupportsOption(b"dega", b"normalization_factor") and not lib.EncoderSupportsOption(b"copy", b"adaptive")


def test_option_support_by_function_pointer(lib):
    """EncoderFromFunctionSupportsOption (DCLib/inc/enc_dec.h:59, DCLib/src/enc_dec.c:216-225; used by DCCLI/src/cli.c:305):
    the option mask of the row a codec FUNCTION belongs to, looked up by encoder or by decoder pointer."""
    class EncDec(C.Structure):
        _fields_ = [("encoder", C.c_void_p), ("decoder", C.c_void_p)]
    dega = EncDec.from_address(lib.GetEncoder(b"dega"))
    fdega = EncDec.from_address(lib.GetEncoder(b"fdega"))
    copy = EncDec.from_address(lib.GetEncoder(b"copy"))
    f = lib.EncoderFromFunctionSupportsOption
    assert f(dega.encoder, 1, b"adaptive") and f(dega.decoder, 0, b"valuesize") and f(dega.encoder, 1, b"num_channels")
    assert not f(dega.encoder, 1, b"normalization_factor") and f(fdega.decoder, 0, b"normalization_factor")
    assert not f(dega.encoder, 0, b"adaptive")  # an encoder pointer is not a decoder (enc_dec.c:221)
    assert f(copy.encoder, 1, b"blocksize") and not f(copy.encoder, 1, b"adaptive")
    assert not f(dega.encoder, 1, b"nonsense") and not f(None, 1, b"adaptive")


def test_bit_stream_format(lib):
    fb, bb = lib.AllocateFileBuffer(), lib.AllocateBitFileBuffer()
    assert lib.InitFileBufferInMemory(fb, 1, 16) == 0
    lib.InitBitFileBuffer(bb, fb)
    for value, n in ((0b101, 3), (0x1234ABCD, 32), (1, 1), (0, 0), (0x3FF, 10)):
        v = C.c_uint64(value)
        assert lib.WriteSingleValueToBitFileBuffer(bb, C.byref(v), n) == n
    assert lib.SetBitFileBufferMode(bb, 0) == 0  # write -> read keeps the exact 46 bits
    got = (C.c_uint8 * 8)()
    assert lib.ReadBitFileBuffer(bb, got, 46) == 46 and lib.EndOfBitFileBuffer(bb) == 1
    bits = "101" + format(0x1234ABCD, "032b") + "1" + format(0x3FF, "010b")
    want = int(bits.ljust(48, "0"), 2).to_bytes(6, "big")  # MSB first, last byte zero padded
    assert bytes(got[:6]) == want
    v = C.c_uint64()
    assert lib.ReadSingleValueFromBitFileBuffer(bb, C.byref(v), 8) == 0  # nothing left: short read


def run_cli(args):
    return subprocess.run([CLI] + args, capture_output=True, text=True)


def test_cli_copy_chain_and_file_padding(lib, tmp_path):
    src = tmp_path / "in.bin"
    data = bytes(range(1, 40))
    src.write_bytes(data)
    out = tmp_path / "out.bin"
    p = run_cli([str(src), str(out), "encode", "copy", "blocksize=3", "#", "decode", "copy", "blocksize=7", "#", "encode", "copy"])
    assert p.returncode == 0 and out.read_bytes() == data
    assert "Executing encoder copy (1 of 3 total)" in p.stdout and "Total time elapsed" in p.stdout
    empty = tmp_path / "empty.bin"
    empty.write_bytes(b"")
    p = run_cli([str(empty), str(out), "encode", "copy"])
    assert p.returncode == 0 and out.read_bytes() == b"\x00"  # an empty output still gets one zero byte
    assert run_cli([str(src), str(out), "encode", "nonsense"]).returncode != 0
    assert run_cli([str(src), str(out), "encode", "copy", "adaptive"]).returncode != 0  # option not supported by codec
    assert run_cli([str(src), str(out), "frobnicate", "copy"]).returncode != 0


def test_cli_gpu_codec_fails_loudly_without_gpu(lib, tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    src = tmp_path / "in.bin"
    src.write_bytes(np.arange(10, dtype=">i4").tobytes())
    p = run_cli([str(src), str(tmp_path / "o.bin"), "encode", "dega", "adaptive"])
    assert p.returncode == 245  # ERROR_LIBRARY_CALL as an exit status, like DCCLI (cli.c:447-453)
    assert "Error initializing library" in p.stderr


# ---- GPU: the driver runs the DEGA path end to end ---------------------------------------------------------------------

@pytest.mark.gpu
def test_cli_dega_on_reference_test_file(lib, tmp_path):
    """`decode csv # encode normalize` (reference, CPU) hands float32 / big-endian int32 to our stages; the GPU output is
    byte-identical to the reference's canonical DEGA file, and decodes back to the same bytes."""
    with gzip.open(os.path.join(GOLDEN, "input.txt.gz"), "rb") as f:
        v = np.array(f.read().split(), dtype=np.float64).astype(np.float32)
    with open(os.path.join(GOLDEN, "dega_adaptive.bin"), "rb") as f:
        want = f.read()
    fl = tmp_path / "in.f32"
    fl.write_bytes(v.tobytes())
    out = tmp_path / "out.dega"
    p = run_cli([str(fl), str(out), "encode", "fdega", "adaptive", "normalization_factor=100"])
    assert p.returncode == 0, p.stderr
    assert out.read_bytes() == want
    assert "Wrote 169326 bytes and 5 bits" in p.stdout  # exact bit length, as DCCLI reports it
    back = tmp_path / "back.f32"
    p = run_cli([str(out), str(back), "decode", "fdega", "adaptive", "normalization_factor=100"])
    assert p.returncode == 0, p.stderr
    assert back.read_bytes() == v.tobytes()
    # integer entry: big-endian int32 in (what `encode normalize` emits)
    ints = np.frombuffer(orc_normalize(v), dtype=np.uint8)
    bi = tmp_path / "in.be32"
    bi.write_bytes(ints.tobytes())
    p = run_cli([str(bi), str(out), "encode", "dega", "adaptive", "#", "decode", "dega", "adaptive"])
    assert p.returncode == 0, p.stderr
    assert out.read_bytes() == ints.tobytes()
    p = run_cli([str(bi), str(out), "encode", "dega", "adaptive"])
    assert p.returncode == 0 and out.read_bytes() == want


def orc_normalize(v):
    from oracle import orc
    ret, b, n = orc.stage("normalize", True, v.tobytes(), v.size * 32)
    assert ret == 0
    return b


@pytest.mark.gpu
def test_cli_batch_container_round_trip(lib, tmp_path):
    rng = np.random.default_rng(5)
    T, Cn = 300, 37
    x = (np.cumsum(rng.integers(-40, 41, (T, Cn)), axis=0) + 20000).astype(">i4")
    src = tmp_path / "batch.be32"
    src.write_bytes(x.tobytes())  # sample-major interleaving = [T][C]
    enc = tmp_path / "batch.degb"
    dec = tmp_path / "batch.out"
    p = run_cli([str(src), str(enc), "encode", "dega", "adaptive", "num_channels=%d" % Cn])
    assert p.returncode == 0, p.stderr
    blob = enc.read_bytes()
    assert blob[:4] == b"DEGB" and int.from_bytes(blob[8:16], "big") == Cn and int.from_bytes(blob[16:24], "big") == T
    p = run_cli([str(enc), str(dec), "decode", "dega", "adaptive", "num_channels=%d" % Cn])
    assert p.returncode == 0, p.stderr
    assert dec.read_bytes() == x.tobytes()
    # every channel's stream inside the container is the reference chain's stream for that channel
    from oracle import orc
    lens = [int.from_bytes(blob[24 + 8 * c: 32 + 8 * c], "big") for c in range(Cn)]
    off = 24 + 8 * Cn
    for c in range(Cn):
        ret, b, n = orc.encode_i32(x[:, c].astype(np.int32), 1)
        assert ret == 0 and n == lens[c] and blob[off: off + len(b)] == b
        off += (lens[c] + 7) // 8


@pytest.mark.gpu
def test_cli_glzmh_on_reference_test_file(lib, tmp_path):
    """`encode glzmh` of the reference's test file is the file `encode lzmh` writes (270 896 bytes, sha256 c493269c...,
    SURVEY.md Appendix B) and `decode glzmh` restores the text."""
    import hashlib
    import json
    with open(os.path.join(GOLDEN, "lzmh.json")) as f:
        meta = json.load(f)["testfile"]
    with gzip.open(os.path.join(GOLDEN, "input.txt.gz"), "rb") as f:
        raw = f.read()
    src = tmp_path / "input.txt"
    src.write_bytes(raw)
    out = tmp_path / "out.lzmh"
    p = run_cli([str(src), str(out), "encode", "glzmh"])
    assert p.returncode == 0, p.stderr
    data = out.read_bytes()
    assert len(data) == meta["file_bytes"] and hashlib.sha256(data).hexdigest() == meta["sha256"]
    assert "Wrote %d bytes and %d bits" % (meta["wrote_bytes"], meta["wrote_bits"]) in p.stdout
    back = tmp_path / "back.txt"
    p = run_cli([str(out), str(back), "decode", "glzmh"])  # from the zero-padded file, as DCCLI would
    assert p.returncode == 0, p.stderr
    assert back.read_bytes() == raw
    p = run_cli([str(src), str(back), "encode", "glzmh", "#", "decode", "glzmh"])  # chained: exact bit length hand-off
    assert p.returncode == 0 and back.read_bytes() == raw


@pytest.mark.gpu
def test_cli_dega_with_valuesize(lib, tmp_path):
    """`valuesize=n` through the plugin: n-bit big-endian fields in, the stream of the reference's three stages with the
    same option out (here: the oracle's, which tests/test_valuesize.py pins to the reference), and back."""
    from oracle import orc
    rng = np.random.default_rng(9)
    for vs in (12, 16, 7, 40, 64):  # 1..32 in int32 containers, 33..64 in int64 ones
        T = 504  # whole bytes for every one of the sizes
        half = 1 << (vs - 1)
        step = min(half // 32, 1 << 40)
        x = (np.cumsum(rng.integers(-(step + 1), step + 2, T)) + half // 2).clip(0, half - 1).astype(np.uint64)
        bits = np.zeros(T * vs, dtype=np.uint8)
        for k in range(vs):
            bits[k::vs] = (x >> np.uint64(vs - 1 - k)) & np.uint64(1)
        packed = np.packbits(bits).tobytes()
        assert (T * vs) % 8 == 0
        d, n = packed, T * vs
        for name in ("diff", "seg", "bac"):
            r, d, n = orc.stage(name, True, d, n, valuesize=vs, adaptive=1)
            assert r == 0
        src = tmp_path / ("in%d.bin" % vs)
        src.write_bytes(packed)
        enc = tmp_path / ("out%d.dega" % vs)
        p = run_cli([str(src), str(enc), "encode", "dega", "adaptive", "valuesize=%d" % vs])
        assert p.returncode == 0, p.stderr
        assert enc.read_bytes() == orc.file_bytes(d, n), vs
        back = tmp_path / ("back%d.bin" % vs)
        p = run_cli([str(enc), str(back), "decode", "dega", "adaptive", "valuesize=%d" % vs])
        assert p.returncode == 0, p.stderr
        assert back.read_bytes() == packed, vs
    p = run_cli([str(src), str(enc), "encode", "fdega", "valuesize=65"])  # 1..64, as the reference's option table (enc_dec.c:72)
    assert p.returncode != 0


@pytest.mark.gpu
def test_cli_fdega_valuesizes_against_the_reference(lib, tmp_path):
    """`fdega valuesize=n` (Normalize inside the encode kernel, Denormalize inside the decode kernel) for n = 8..64: the
    stream and the floats coming back are the compiled reference's (tests/golden/floats_vs.npz, made by make_golden.py)."""
    z = np.load(os.path.join(GOLDEN, "floats_vs.npz"))
    for tag in sorted({k.split(".")[0] for k in z.files}, key=lambda t: int(t[2:])):
        vs, factor = int(tag[2:]), float(z[tag + ".factor"][0])
        v = z[tag + ".v"]
        for c in (0, 1, 3, 2):  # 2 is the channel that fails the range check of normalize.c:21
            src = tmp_path / "v.f32"
            src.write_bytes(np.ascontiguousarray(v[:, c]).tobytes())
            enc = tmp_path / "v.dega"
            p = run_cli([str(src), str(enc), "encode", "fdega", "adaptive", "valuesize=%d" % vs, "normalization_factor=%r" % factor])
            if int(z[tag + ".err"][c]) != 0:
                assert p.returncode == 245 and "Invalid value" in p.stderr, (tag, c)
                continue
            assert p.returncode == 0, (tag, c, p.stderr)
            nb = int(z[tag + ".bits"][c])
            assert enc.read_bytes() == z[tag + ".stream"][c, : (nb + 7) // 8].tobytes(), (tag, c)
            back = tmp_path / "back.f32"
            p = run_cli([str(enc), str(back), "decode", "fdega", "adaptive", "valuesize=%d" % vs, "normalization_factor=%r" % factor])
            assert p.returncode == 0, (tag, c, p.stderr)
            assert back.read_bytes() == np.ascontiguousarray(z[tag + ".back"][:, c]).tobytes(), (tag, c)


@pytest.mark.gpu
def test_cli_long_single_stream_round_trip(lib, tmp_path):
    """A bare (headerless) stream larger than 2 MiB: the decoder's first guess at the sample count is clamped to what one
    library call takes (2^25) instead of being rejected, so encode -> decode round-trips (the reference has no such limit)."""
    rng = np.random.default_rng(77)
    T = 1 << 20
    x = rng.integers(0, 1 << 30, T).astype(">i4")  # ~60 coded bits per sample: a stream of ~7.5 MiB
    src = tmp_path / "noise.be32"
    src.write_bytes(x.tobytes())
    enc = tmp_path / "noise.dega"
    p = run_cli([str(src), str(enc), "encode", "dega", "adaptive"])
    assert p.returncode == 0, p.stderr
    assert enc.stat().st_size > (2 << 20)
    back = tmp_path / "noise.back"
    p = run_cli([str(enc), str(back), "decode", "dega", "adaptive"])
    assert p.returncode == 0, p.stderr
    assert back.read_bytes() == x.tobytes()


@pytest.mark.gpu
def test_cli_damaged_container_headers_are_rejected(lib, tmp_path):
    """Nothing of a DEGB header is trusted: channel / sample counts and bit lengths that do not fit the bytes that follow
    (including lengths near 2^64, whose rounding to bytes would wrap) give ERROR_INVALID_FORMAT, not a kernel launch."""
    rng = np.random.default_rng(3)
    T, Cn = 64, 5
    x = (np.cumsum(rng.integers(-9, 10, (T, Cn)), axis=0) + 500).astype(">i4")
    src = tmp_path / "b.be32"
    src.write_bytes(x.tobytes())
    enc = tmp_path / "b.degb"
    assert run_cli([str(src), str(enc), "encode", "dega", "adaptive", "num_channels=%d" % Cn]).returncode == 0
    good = bytearray(enc.read_bytes())
    out = tmp_path / "b.out"

    def decode(blob):
        bad = tmp_path / "bad.degb"
        bad.write_bytes(bytes(blob))
        return run_cli([str(bad), str(out), "decode", "dega", "adaptive", "num_channels=%d" % Cn])

    assert decode(good).returncode == 0 and out.read_bytes() == x.tobytes()
    for at, value in ((24, (1 << 64) - 3), (24 + 8, (1 << 63)), (24 + 16, 8 * len(good)), (16, 1 << 40), (16, (1 << 64) - 1), (8, Cn + 1), (8, 1 << 61)):
        blob = bytearray(good)
        blob[at: at + 8] = value.to_bytes(8, "big")
        p = decode(blob)
        assert p.returncode == 245 and "Invalid format" in p.stderr, (at, value, p.stderr)
    assert decode(good[:30]).returncode == 245
    # a stream cut short inside a channel is that channel's decode error, not a crash
    p = decode(good[: len(good) - 7])
    assert p.returncode == 245


@pytest.mark.gpu
def test_cli_glzmh_batch_container(lib, tmp_path):
    """`glzmh num_channels=n`: the input cut into n pieces, each a stream of its own in one launch ("LZMB" container); every
    inner stream is what the reference's `encode lzmh` makes of that piece, and decode restores the text."""
    from oracle import orc
    with gzip.open(os.path.join(GOLDEN, "input.txt.gz"), "rb") as f:
        raw = f.read()[:120000]
    src = tmp_path / "text.txt"
    src.write_bytes(raw)
    enc, back = tmp_path / "text.lzmb", tmp_path / "text.back"
    n = 7
    p = run_cli([str(src), str(enc), "encode", "glzmh", "num_channels=%d" % n])
    assert p.returncode == 0, p.stderr
    blob = enc.read_bytes()
    assert blob[:4] == b"LZMB" and int.from_bytes(blob[8:16], "big") == n
    piece = (len(raw) + n - 1) // n
    off = 16 + 16 * n
    for c in range(n):
        want_len, nbits = int.from_bytes(blob[16 + 16 * c: 24 + 16 * c], "big"), int.from_bytes(blob[24 + 16 * c: 32 + 16 * c], "big")
        part = raw[c * piece: (c + 1) * piece]
        assert want_len == len(part)
        r, b, nb = orc.stage("lzmh", True, part, 8 * len(part))
        assert r == 0 and nb == nbits and blob[off: off + (nb + 7) // 8] == b[: (nb + 7) // 8], c
        off += (nbits + 7) // 8
    p = run_cli([str(enc), str(back), "decode", "glzmh", "num_channels=%d" % n])
    assert p.returncode == 0, p.stderr
    assert back.read_bytes() == raw
    bad = bytearray(blob)
    bad[24:32] = ((1 << 64) - 1).to_bytes(8, "big")
    (tmp_path / "bad.lzmb").write_bytes(bytes(bad))
    assert run_cli([str(tmp_path / "bad.lzmb"), str(back), "decode", "glzmh", "num_channels=%d" % n]).returncode == 245
