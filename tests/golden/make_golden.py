#!/usr/bin/env python3
"""Generates the golden fixtures in tests/golden/ from the COMPILED REFERENCE (oracle/_ref, built by oracle/Makefile
from /root/reference).  Run in the build container only (the reference does not exist on the GPU box):

    make -C oracle && python tests/golden/make_golden.py

Fixtures are data only (inputs + the reference's outputs):
  input.txt.gz         the reference's own test file DCCLI/testdata/input.txt (100 000 lines), gzip'ed
  testfile.json        size / bit length / sha256 of every stage of the reference's `make test` chain on it
                       (DCCLI/build/gcc/Makefile:75-77), plus first/last bytes
  dega_adaptive.bin    the canonical DEGA stream of the test file (chained: decode csv # encode normalize #
                       encode diff # encode seg # encode bac adaptive), as the file DCCLI writes
  kats.json            known-answer vectors re-generated through the reference library
  channels.npz         small int32 channel batches [T][C] + the reference's per-channel DEGA streams
  floats.npz           float32 edge cases + the reference's normalize / denormalize results
  floats_vs.npz        float32 channels through the whole chain at valuesize 8..64 (the float entry at other value sizes)
  valuesizes64.npz     the same for valuesize 33..64 (uint64 samples), including differences the decoder cannot take back
  valuesizes.npz       channel batches for valuesize 1..31 (unsigned valuesize-bit samples) + the reference's streams of
                       `encode diff valuesize=n # encode seg valuesize=n # encode bac [adaptive]`, error channels included
  lzmh.json / lzmh.npz the reference's `encode lzmh` of the test file (size, bits, sha256) and of small byte strings
                       (meter CSV text, digits, binary, periodic; lengths around the 403-byte ring size)
"""
import gzip
import hashlib
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import orc  # noqa: E402

REF_INPUT = "/root/reference/DataCompressor/DCCLI/testdata/input.txt"


def sha(b):
    return hashlib.sha256(b).hexdigest()


def dccli(infile, stages):
    """Run the reference CLI; returns (file bytes, (bytes, bits) it reports having written)."""
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "out.bin")
        args = [orc.REF_CLI, infile, out]
        for i, s in enumerate(stages):
            if i:
                args.append("#")
            args += s.split()
        p = subprocess.run(args, capture_output=True, text=True)
        if p.returncode != 0:
            return None, p.returncode
        wrote = [ln for ln in p.stdout.splitlines() if "Wrote" in ln][-1]
        nbytes = int(wrote.split("(")[1].split()[0])
        nbits = int(wrote.split("and")[1].split()[0])
        with open(out, "rb") as f:
            return f.read(), (nbytes, nbits)


def testfile():
    with open(REF_INPUT, "rb") as f:
        raw = f.read()
    with gzip.GzipFile(os.path.join(HERE, "input.txt.gz"), "wb", compresslevel=9, mtime=0) as g:
        g.write(raw)
    chain = ["decode csv", "encode normalize", "encode diff", "encode seg", "encode bac adaptive"]
    names = ["float32", "normalize", "diff", "seg", "dega_adaptive"]
    meta = {"input_sha256": sha(raw), "input_bytes": len(raw), "stages": {}}
    for i, name in enumerate(names):
        data, wrote = dccli(REF_INPUT, chain[: i + 1])
        meta["stages"][name] = {
            "chain": " # ".join(chain[: i + 1]), "file_bytes": len(data), "wrote_bytes": wrote[0], "wrote_bits": wrote[1],
            "sha256": sha(data), "head": data[:8].hex(), "tail": data[-8:].hex(),
        }
        if name == "dega_adaptive":
            with open(os.path.join(HERE, "dega_adaptive.bin"), "wb") as f:
                f.write(data)
    data, wrote = dccli(REF_INPUT, chain[:4] + ["encode bac"])
    meta["stages"]["dega_nonadaptive"] = {"chain": " # ".join(chain[:4] + ["encode bac"]), "file_bytes": len(data),
                                          "wrote_bytes": wrote[0], "wrote_bits": wrote[1], "sha256": sha(data),
                                          "head": data[:8].hex(), "tail": data[-8:].hex()}
    # the full round trip of `make test`
    full = chain + ["decode bac adaptive", "decode seg", "decode diff", "decode normalize", "encode csv"]
    data, _ = dccli(REF_INPUT, full)
    meta["roundtrip_identical"] = (data == raw)
    with open(os.path.join(HERE, "testfile.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print("testfile:", {k: v["sha256"][:12] for k, v in meta["stages"].items()}, "roundtrip", meta["roundtrip_identical"])


def kats():
    cases = {
        "empty": [], "zero": [0], "zeros96": [0] * 96, "max": [2147483647], "max_zero": [2147483647, 0],
        "sign_change": [5, -3, 7], "one": [1], "ramp": list(range(0, 64)), "neg_first": [-1],
        "big_swing": [0, 2147483647, 0, 2147483647], "alternating": [1000, 1001] * 40,
    }
    res = {}
    for name, x in cases.items():
        for ad in (1, 0):
            ret, b, n, _ = orc.ref_encode_i32(np.array(x, dtype=np.int32), ad)
            res["%s/%s" % (name, "adaptive" if ad else "static")] = {"x": x, "adaptive": ad, "ret": int(ret), "hex": b.hex(), "nbits": int(n)}
    # the empty stream through `encode diff` alone, written to a FILE, is one 0x00 byte (flush quirk, Appendix A.0)
    with tempfile.NamedTemporaryFile() as tf:
        data, wrote = dccli(tf.name, ["encode diff"])
    res["empty_diff_file"] = {"file_hex": data.hex(), "wrote": list(wrote)}
    with open(os.path.join(HERE, "kats.json"), "w") as f:
        json.dump(res, f, indent=1, sort_keys=True)
    print("kats:", len(res))


def channels():
    rng = np.random.default_rng(20241004)
    sets = {}

    def walk(T, Cn, S, base_lo=10000, base_hi=60000):
        x = np.zeros((T, Cn), dtype=np.int64)
        if T:
            x[0] = rng.integers(base_lo, base_hi, Cn)
            steps = rng.integers(-S, S + 1, (T, Cn))
            for t in range(1, T):
                x[t] = np.clip(x[t - 1] + steps[t], 0, 2**31 - 1)
        return x.astype(np.int32)

    sets["walk50"] = walk(700, 24, 50)                       # 1-s granularity style (cfg2 shape, shortened)
    sets["walk300_T96"] = walk(96, 80, 300)                  # 15-min granularity style (cfg3 shape)
    sets["ragged_small"] = walk(5, 7, 3, 0, 4)               # tiny values: exercises the MPS/LPS swap path
    wild = rng.integers(0, 2**31, (200, 16)).astype(np.int64)
    wild[::7] = 0
    wild[3::11] = 2**31 - 1
    sets["wild"] = wild.astype(np.int32)                     # full-range jumps: 63-bit codewords, long pending runs
    bad = walk(50, 8, 50)
    bad[10, 2] = -5                                          # negative sample -> ERROR_INVALID_VALUE from diff
    bad[0, 5] = -1
    sets["with_errors"] = bad
    sets["zeros"] = np.zeros((300, 4), dtype=np.int32)
    sets["const"] = np.full((300, 4), 123456, dtype=np.int32)
    # long channel crossing many model halvings (first at 16 380 coded bits, then every ~8 190)
    sets["long"] = walk(6000, 3, 50)
    out = {}
    for name, x in sets.items():
        T, Cn = x.shape
        for ad in (1, 0):
            streams, bits, errs = [], [], []
            for c in range(Cn):
                ret, b, n, _ = orc.ref_encode_i32(np.ascontiguousarray(x[:, c]), ad)
                streams.append(b)
                bits.append(n)
                errs.append(ret)
                if ret == 0:
                    rd, y, _ = orc.ref_decode_i32(b, n, T, ad)
                    assert rd == 0 and (y == x[:, c]).all()
            cap = max(1, max(len(s) for s in streams))
            arr = np.zeros((Cn, cap), dtype=np.uint8)
            for c, s in enumerate(streams):
                arr[c, : len(s)] = np.frombuffer(s, dtype=np.uint8)
            tag = "%s.%s" % (name, "ad" if ad else "st")
            out[tag + ".stream"] = arr
            out[tag + ".bits"] = np.array(bits, dtype=np.uint64)
            out[tag + ".err"] = np.array(errs, dtype=np.int32)
        out[name + ".x"] = x
    np.savez_compressed(os.path.join(HERE, "channels.npz"), **out)
    print("channels:", {k: v.shape for k, v in sets.items()})


def floats():
    rng = np.random.default_rng(7)
    v = np.concatenate([
        np.array([0.0, -0.0, 0.004, 0.005, 0.0050001, -0.005, 0.015, 0.025, 1.005, 2.675, 301.87, 327.67, 0.12,
                  -1.0, -0.994, -0.995, -0.996, 1e-30, -1e-30, 21474836.0, 21474836.47, -21474836.48, 1e6 + 0.01,
                  8388607.5, 8388608.0, 16777216.0, 123456.789], dtype=np.float32),
        rng.uniform(0, 400, 2000).astype(np.float32).round(2),
        rng.uniform(-5e4, 5e4, 500).astype(np.float32),
    ]).astype(np.float32)
    out = {"v": v}
    for factor in (100.0, 1.0, 1000.0, 0.5):
        vf = v[np.abs(v.astype(np.float64) * factor) < 2.0e9]  # in-range subset for this factor
        out["v_%g" % factor] = vf
        ret, b, n, _ = orc.ref_run_chain(vf.tobytes(), vf.size * 32, ["encode normalize normalization_factor=%r" % factor])
        assert ret == 0, (factor, ret)
        ints = np.frombuffer(b, dtype=">i4").astype(np.int32)
        out["norm_%g" % factor] = ints
        ret, b2, n2, _ = orc.ref_run_chain(b, n, ["decode normalize normalization_factor=%r" % factor])
        assert ret == 0
        out["denorm_%g" % factor] = np.frombuffer(b2, dtype=np.float32).copy()
    # out-of-range values make the reference fail with ERROR_INVALID_VALUE (normalize.c:21-22); 2^31 exactly passes
    edge = np.array([21474836.48 * 1.0001, -21474836.48 * 1.001, 3e9, -3e9], dtype=np.float32)
    rets = []
    for e in edge:
        ret, _, _, _ = orc.ref_run_chain(np.array([e], dtype=np.float32).tobytes(), 32, ["encode normalize"])
        rets.append(ret)
    out["edge_v"] = edge
    out["edge_ret"] = np.array(rets, dtype=np.int64)
    ret, b, n, _ = orc.ref_run_chain(np.array([21474836.48], dtype=np.float32).tobytes(), 32, ["encode normalize"])
    out["two31_ret"] = np.array([ret], dtype=np.int64)
    out["two31_hex"] = np.frombuffer(b, dtype=np.uint8).copy()
    np.savez_compressed(os.path.join(HERE, "floats.npz"), **out)
    print("floats:", v.size, "edge rets", rets, "2^31:", ret, b.hex())


def floats_vs():
    """The float entry at other value sizes (fdega valuesize=n): per channel the reference's
    `encode normalize normalization_factor=f valuesize=n # encode diff valuesize=n # encode seg valuesize=n # encode bac adaptive`
    stream, and what the inverse chain gives back as float32 -- 1..64 bits, error channels included."""
    rng = np.random.default_rng(2026)
    out = {}
    sets = []
    for vs, factor in ((8, 1.0), (12, 10.0), (16, 100.0), (24, 100.0), (31, 1000.0), (32, 0.5), (33, 100.0), (40, 1000.0), (48, 100.0), (63, 1.0), (64, 100.0)):
        T, Cn = 96, 8
        half = float(2 ** (vs - 1))
        v = np.zeros((T, Cn), dtype=np.float32)
        for c in range(Cn):
            kind = c % 4
            # normalized values must stay non-negative and steps within +-2^(vs-1) (diff.c:15-18)
            scale = min(half / factor, 3.0e6)
            if kind == 0:
                col = np.abs(np.cumsum(rng.normal(0, scale / 400, T)) + scale / 4)
            elif kind == 1:
                col = rng.uniform(0, scale / 2.5, T)
            elif kind == 2:
                col = np.abs(np.cumsum(rng.normal(0, scale / 4000, T)) + scale / 8)
                col[T // 2] = half * 4.0 / factor  # out of range: ERROR_INVALID_VALUE (normalize.c:21-22)
            else:
                col = np.round(rng.uniform(0, min(scale / 3, 400.0), T), 2)
            v[:, c] = col.astype(np.float32)
        tag = "vs%d" % vs
        sets.append(tag)
        out[tag + ".v"] = v
        out[tag + ".factor"] = np.array([factor], dtype=np.float32)
        opt = " valuesize=%d" % vs
        streams, bits, errs, backs = [], [], [], []
        for c in range(Cn):
            col = np.ascontiguousarray(v[:, c])
            ret, b, nb, _ = orc.ref_run_chain(col.tobytes(), col.size * 32, ["encode normalize normalization_factor=%r" % factor + opt, "encode diff" + opt,
                                                                              "encode seg" + opt, "encode bac adaptive"])
            streams.append(b if ret == 0 else b"")
            bits.append(nb if ret == 0 else 0)
            errs.append(ret)
            back = np.zeros(T, dtype=np.float32)
            if ret == 0:
                rd, d, dn, _ = orc.ref_run_chain(b, nb, ["decode bac adaptive", "decode seg" + opt, "decode diff" + opt,
                                                         "decode normalize normalization_factor=%r" % factor + opt])
                assert rd == 0 and dn == 32 * T, (vs, c, rd, dn)
                back = np.frombuffer(d, dtype=np.float32).copy()
            backs.append(back)
        cap = max(1, max(len(s_) for s_ in streams))
        arr = np.zeros((Cn, cap), dtype=np.uint8)
        for c, s_ in enumerate(streams):
            arr[c, : len(s_)] = np.frombuffer(s_, dtype=np.uint8)
        out[tag + ".stream"] = arr
        out[tag + ".bits"] = np.array(bits, dtype=np.uint64)
        out[tag + ".err"] = np.array(errs, dtype=np.int32)
        out[tag + ".back"] = np.stack(backs, axis=1)
    np.savez_compressed(os.path.join(HERE, "floats_vs.npz"), **out)
    print("floats_vs:", {t: int((out[t + ".err"] != 0).sum()) for t in sets}, "(channels in error)")


def pack_be(vals, vs):
    """unsigned values as vs-bit big-endian fields -> (bytes, nbits): what a stage reads with valuesize=vs"""
    v = np.asarray(vals, dtype=np.uint64)
    bits = np.zeros(len(v) * vs, dtype=np.uint8)
    for k in range(vs):
        bits[k::vs] = (v >> np.uint64(vs - 1 - k)) & np.uint64(1)
    return np.packbits(bits).tobytes(), len(v) * vs


def valuesizes():
    rng = np.random.default_rng(31)
    out = {}
    for vs in (1, 2, 7, 8, 12, 15, 16, 17, 24, 31):
        T, Cn = 160, 12
        top = (1 << vs) - 1
        x = np.zeros((T, Cn), dtype=np.int64)
        for c in range(Cn):
            kind = c % 4
            if kind == 0:
                col = np.clip(np.cumsum(rng.integers(-3, 4, T)) + top // 2, 0, top)
            elif kind == 1:
                col = rng.integers(0, top + 1, T)  # jumps beyond +-2^(vs-1): ERROR_INVALID_VALUE (diff.c:17-18)
            elif kind == 2:
                col = np.clip(np.cumsum(rng.integers(-(top // 8 + 1), top // 8 + 2, T)) + top // 2, 0, top)
            else:
                col = np.full(T, top // 3)
            x[:, c] = col
        out["vs%d.x" % vs] = x.astype(np.uint32)
        for ad in (1, 0):
            streams, bits, errs = [], [], []
            opt = " valuesize=%d" % vs
            for c in range(Cn):
                data, n = pack_be(x[:, c], vs)
                ret, b, nb, _ = orc.ref_run_chain(data, n, ["encode diff" + opt, "encode seg" + opt, "encode bac adaptive" if ad else "encode bac"])
                streams.append(b if ret == 0 else b"")
                bits.append(nb if ret == 0 else 0)
                errs.append(ret)
                if ret == 0:
                    rd, d, dn, _ = orc.ref_run_chain(b, nb, ["decode bac adaptive" if ad else "decode bac", "decode seg" + opt, "decode diff" + opt])
                    assert rd == 0 and (d[: (dn + 7) // 8], dn) == (data, n), (vs, c)
            cap = max(1, max(len(s_) for s_ in streams))
            arr = np.zeros((Cn, cap), dtype=np.uint8)
            for c, s_ in enumerate(streams):
                arr[c, : len(s_)] = np.frombuffer(s_, dtype=np.uint8)
            tag = "vs%d.%s" % (vs, "ad" if ad else "st")
            out[tag + ".stream"] = arr
            out[tag + ".bits"] = np.array(bits, dtype=np.uint64)
            out[tag + ".err"] = np.array(errs, dtype=np.int32)
    np.savez_compressed(os.path.join(HERE, "valuesizes.npz"), **out)
    print("valuesizes:", {k[:-7]: int((v != 0).sum()) for k, v in out.items() if k.endswith(".ad.err")}, "(channels in error)")


def valuesizes64():
    rng = np.random.default_rng(64)
    out = {}
    for vs in (33, 40, 48, 63, 64):
        T, Cn = 80, 10
        top = (1 << vs) - 1
        x = np.zeros((T, Cn), dtype=np.uint64)
        for c in range(Cn):
            kind = c % 5
            if kind == 0:
                col = [int(v) for v in np.clip(np.cumsum(rng.integers(-1000, 1001, T)) + 10**6, 0, None)]
            elif kind == 1:
                col = [int(rng.integers(0, 2**62)) * 4 % (top + 1) for _ in range(T)]  # jumps: range errors below 64 bits
            elif kind == 2:
                col, cur = [], top // 4
                for _ in range(T):
                    cur = min(max(cur + int(rng.integers(-(2 ** (vs - 3)), 2 ** (vs - 3))), 0), top // 2)
                    col.append(cur)
            elif kind == 3:
                col = [top // 3] * T
            elif vs == 64 and c == 9:
                col = [0, 1 << 63, (1 << 63) + 5, 5, 0] * (T // 5)  # a difference of magnitude 2^63: encodes, does not decode (seg.c:74)
            else:
                col = [0, top // 2, 0, 1, top // 2 - 1] * (T // 5)
            x[:, c] = np.array(col, dtype=np.uint64)
        out["vs%d.x" % vs] = x
        for ad in (1, 0):
            streams, bits, errs, decs = [], [], [], []
            decoded = np.zeros((T, Cn), dtype=np.uint64)
            opt = " valuesize=%d" % vs
            for c in range(Cn):
                data, n = pack_be(x[:, c], vs)
                ret, b, nb, _ = orc.ref_run_chain(data, n, ["encode diff" + opt, "encode seg" + opt, "encode bac adaptive" if ad else "encode bac"])
                streams.append(b if ret == 0 else b"")
                bits.append(nb if ret == 0 else 0)
                errs.append(ret)
                rd = 0
                if ret == 0:  # what the reference's own decoder makes of it: the input, except after a difference of 2^63
                    rd, d, dn, _ = orc.ref_run_chain(b, nb, ["decode bac adaptive" if ad else "decode bac", "decode seg" + opt, "decode diff" + opt])
                    if rd == 0:
                        assert dn == n, (vs, c)
                        dbits = np.unpackbits(np.frombuffer(d, dtype=np.uint8))[:dn].reshape(-1, vs).astype(np.uint64)
                        for k in range(vs):
                            decoded[:, c] |= dbits[:, k] << np.uint64(vs - 1 - k)
                        assert (d[: (dn + 7) // 8] == data) or (vs == 64 and c == 9), (vs, c)
                decs.append(rd)
            cap = max(1, max(len(s_) for s_ in streams))
            arr = np.zeros((Cn, cap), dtype=np.uint8)
            for c, s_ in enumerate(streams):
                arr[c, : len(s_)] = np.frombuffer(s_, dtype=np.uint8)
            tag = "vs%d.%s" % (vs, "ad" if ad else "st")
            out[tag + ".stream"] = arr
            out[tag + ".bits"] = np.array(bits, dtype=np.uint64)
            out[tag + ".err"] = np.array(errs, dtype=np.int32)
            out[tag + ".decerr"] = np.array(decs, dtype=np.int32)
            out[tag + ".dec"] = decoded
    np.savez_compressed(os.path.join(HERE, "valuesizes64.npz"), **out)
    print("valuesizes64:", {k[:-7]: int((v != 0).sum()) for k, v in out.items() if k.endswith(".ad.err")}, "(channels in error)",
          "lossy columns:", [k for k, v in out.items() if k.endswith(".ad.dec") and (v != out[k[:-7] + ".x"]).any()])


def lzmh_inputs():
    """Deterministic byte strings for the LZMH fixtures (name -> bytes)."""
    rng = np.random.default_rng(11)
    out = {}
    for L in (0, 1, 2, 3, 255, 256, 257, 402, 403, 404, 405, 806, 1209):
        out["digits_%d" % L] = bytes(rng.integers(48, 58, L, dtype=np.uint8))
    for i, L in enumerate((700, 1500, 4000)):
        walk = 230.0 + np.cumsum(rng.normal(0, 0.4, L // 7 + 2))
        out["csv_%d" % L] = "".join("%.2f\n" % v for v in walk).encode()[:L]
    out["binary_2000"] = bytes(rng.integers(0, 256, 2000, dtype=np.uint8))
    out["few_symbols_3000"] = bytes(rng.integers(0, 3, 3000, dtype=np.uint8))
    out["periodic_1000"] = (b"abcabcabd" * 112)[:1000]
    out["zeros_900"] = bytes(900)
    out["many_symbols_5000"] = bytes((rng.integers(0, 60, 5000) + 32).astype(np.uint8))
    return out


def lzmh():
    with open(REF_INPUT, "rb") as f:
        raw = f.read()
    data, wrote = dccli(REF_INPUT, ["encode lzmh"])
    back, _ = dccli(REF_INPUT, ["encode lzmh", "decode lzmh"])
    meta = {"testfile": {"chain": "encode lzmh", "file_bytes": len(data), "wrote_bytes": wrote[0], "wrote_bits": wrote[1],
                         "sha256": sha(data), "head": data[:8].hex(), "tail": data[-8:].hex(),
                         "roundtrip_identical": back == raw}}
    arrays = {}
    for name, b in lzmh_inputs().items():
        ret, s, n, _ = orc.ref_run_chain(b, 8 * len(b), ["encode lzmh"])
        assert ret == 0, (name, ret)
        ret2, d, dn, _ = orc.ref_run_chain(s, n, ["decode lzmh"])
        assert ret2 == 0
        arrays[name + ".in"] = np.frombuffer(b, dtype=np.uint8).copy()
        arrays[name + ".stream"] = np.frombuffer(s[: (n + 7) // 8], dtype=np.uint8).copy()
        arrays[name + ".bits"] = np.array([n], dtype=np.int64)
        arrays[name + ".dec"] = np.frombuffer(d[: dn // 8], dtype=np.uint8).copy()  # NOT always the input (decoder quirks)
        meta[name] = {"in_bytes": len(b), "bits": n, "decoded_bytes": dn // 8, "decodes_to_input": d[: dn // 8] == b}
    with open(os.path.join(HERE, "lzmh.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    np.savez_compressed(os.path.join(HERE, "lzmh.npz"), **arrays)
    print("lzmh:", meta["testfile"]["file_bytes"], meta["testfile"]["sha256"][:12],
          "not round-tripping:", [k for k, v in meta.items() if k != "testfile" and not v["decodes_to_input"]])


if __name__ == "__main__":
    assert orc.have_ref() and os.path.exists(orc.REF_CLI), "build oracle/_ref first: make -C oracle"
    testfile()
    kats()
    only = sys.argv[1:]
    if only == ["lzmh"]:
        lzmh()
        sys.exit(0)
    if only == ["floats_vs"]:
        floats_vs()
        sys.exit(0)
    if only == ["valuesizes"]:
        valuesizes()
        valuesizes64()
        sys.exit(0)
    channels()
    floats()
    floats_vs()
    valuesizes()
    valuesizes64()
    lzmh()
